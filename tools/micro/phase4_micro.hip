// Developer microbenchmark (GPU box): where the cycles of phase 4 (W^T W on v_mfma_f32_16x16x4_f32 + the value gradient) go.
// One workgroup of 512 threads, the panel in LDS, REPS repetitions per variant, shader-clock cycles per repetition printed.
//   hipcc -O3 -w --offload-arch=gfx950 tools/micro/phase4_micro.hip -o gpurun_out/phase4_micro && gpurun_out/phase4_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#define RLD 36
#define PLD 39
#define NPAN 46
#define NXA 39
#define NS 15
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512, 1) void k(long long* out, int reps)
{
    __shared__ __attribute__((aligned(16))) float Pan[48 * RLD + 64];
    __shared__ float P[NXA * PLD + 8];
    __shared__ float Qb[NS * 16];
    __shared__ double pv[40];
    const int tid = threadIdx.x;
    for (int e = tid; e < 48 * RLD + 64; e += 512) Pan[e] = 0.001f * (float)((e * 7) % 97);
    for (int e = tid; e < NS * 16; e += 512) Qb[e] = 1.f;
    __syncthreads();
    const int wv = tid >> 6, ln = tid & 63, m4 = ln & 15, kq = ln >> 4;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if (wv < 6) {
            const int t = wv;
            const int I = t >= 3 ? 2 : (t >= 1 ? 1 : 0), J = t - I * (I + 1) / 2;
            const int i0 = 16 * I + 4 * kq, jj = 16 * J + m4;
            const float* ra = Pan + (16 * I + m4) * RLD + 4 * kq;
            const float* rb = Pan + (16 * J + m4) * RLD + 4 * kq;
            const float4 a0 = *reinterpret_cast<const float4*>(ra), a1 = *reinterpret_cast<const float4*>(ra + 16);
            const float4 b0 = *reinterpret_cast<const float4*>(rb), b1 = *reinterpret_cast<const float4*>(rb + 16);
            float bs[4];
            const int jc = jj < NS ? jj : NS;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ii = i0 + i;
                const float qb = Qb[(ii < NS ? ii : NS - 1) * 16 + jc];
                bs[i] = ii < NS ? qb : (ii == jj ? 20.f : 0.f);
            }
            v4f c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
            if (MODE >= 1) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, c1, 0, 0, 0);
            } else { c0[0] = a0.x + b0.x + a1.y + b1.y; c0[1] = a0.z + b1.w; c0[2] = a1.x; c0[3] = b0.y; }
            if (MODE >= 2 || MODE == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ii = i0 + i;
                    if (jj <= ii && ii < NXA) {
                        const float rr = bs[i] - (c0[i] + c1[i]);
                        P[ii * PLD + jj] = rr;
                        P[jj * PLD + ii] = rr;
                    }
                }
            } else if (c0[0] + c1[1] + bs[2] == 12345.f) P[0] = 1.f;
        }
        if (MODE >= 3 && tid >= 448 && tid < 448 + NXA) {
            const int i = tid - 448;
            const float4* ri = reinterpret_cast<const float4*>(Pan + i * RLD);
            const float4* rl = reinterpret_cast<const float4*>(Pan + (NPAN - 1) * RLD);
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int q4 = 0; q4 < 8; ++q4) {
                const float4 x = ri[q4], y = rl[q4];
                s0 += x.x * y.x + x.y * y.y;
                if (q4 < 7) s1 += x.z * y.z + x.w * y.w;
            }
            pv[i] = (double)Qb[i % 200] - (double)(s0 + s1);
        }
        if (MODE != 5) __syncthreads();
        if (MODE >= 4) { Pan[(tid * 5) % (46 * RLD)] += 1e-9f; __syncthreads(); }   // (something between repetitions, like the next phase)
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[0] = t1 - t0;
    if (P[tid % 100] == 42.f && pv[tid % 39] == 1.0) out[1] = 1;
}
int main()
{
    long long* d; hipMalloc(&d, 16);
    const int reps = 2000;
    const char* names[] = {"0 loads + epilogue, no MFMA", "1 loads + MFMA, no stores", "2 loads + MFMA + epilogue", "3 + value gradient on wave 7", "4 + a write phase and second barrier", "5 as 3 without the barrier"};
#define RUN(M) { hipLaunchKernelGGL(k<M>, dim3(1), dim3(512), 0, 0, d, reps); hipDeviceSynchronize(); long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("%-45s %8.1f cycles per repetition\n", names[M], (double)h[0] / reps); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    RUN(0) RUN(1) RUN(2) RUN(3)
    return 0;
}
