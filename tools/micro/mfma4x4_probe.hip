// Developer probe (GPU box): v_mfma_f32_4x4x1_16b_f32 with cbsz:4 abid:g as a cross-lane outer-product engine.
// Claim to verify: with cbsz = 4 the A values of block `abid` (lanes 4 abid .. 4 abid + 3) are used by all 16 blocks, so
//     acc_l[i] += A[lane 4 abid + i] * B[lane l]          (l = 0..63, i = 0..3)
// i.e. one instruction applies a rank-1 update to four columns of a row-per-lane matrix whose column entries live in lanes 4 abid + i --
// the trailing update of the fused Cholesky without a single v_readlane.  Also times it (s_memtime, one wave).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma4x4_probe.hip -o gpurun_out/mfma4x4_probe && gpurun_out/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int ABID>
__global__ void probe(const float* a, const float* b, const float* c0, float* out)
{
    const int l = threadIdx.x;
    v4f c = {c0[4 * l], c0[4 * l + 1], c0[4 * l + 2], c0[4 * l + 3]};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 4, ABID, 0);
    for (int i = 0; i < 4; ++i) out[4 * l + i] = c[i];
}

// timing: REP rounds of a rank-3 update of NG four-column groups (3 NG MFMAs per round, k-major: independent across groups)
template <int NG>
__global__ void time_mfma(const float* a, float* out, long long* cyc, int rep)
{
    const int l = threadIdx.x;
    v4f acc[NG];
    for (int g = 0; g < NG; ++g) acc[g] = v4f{0.f, 0.f, 0.f, 0.f};
    float x0 = a[l], x1 = a[64 + l], x2 = a[128 + l];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
#define G3(g) if (g < NG) { acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x0, acc[g], 4, g, 0); }
#define H3(g) if (g < NG) { acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(x1, x1, acc[g], 4, g, 0); }
#define I3(g) if (g < NG) { acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(x2, x2, acc[g], 4, g, 0); }
        G3(0) G3(1) G3(2) G3(3) G3(4) G3(5) G3(6) G3(7)
        H3(0) H3(1) H3(2) H3(3) H3(4) H3(5) H3(6) H3(7)
        I3(0) I3(1) I3(2) I3(3) I3(4) I3(5) I3(6) I3(7)
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2));
    }
    float s = 0.f;
    for (int g = 0; g < NG; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[l] = s;
    if (l == 0) cyc[0] = t1 - t0;
}
// the v_readlane / v_pk_fma form of the same update (what the kernel does today), for the same NG groups = 2 NG column pairs
typedef float float2v __attribute__((ext_vector_type(2)));
template <int NG>
__global__ void time_readlane(const float* a, float* out, long long* cyc, int rep)
{
    const int l = threadIdx.x;
    float2v vv[2 * NG];
    for (int g = 0; g < 2 * NG; ++g) vv[g] = float2v{0.f, 0.f};
    float x0 = a[l], x1 = a[64 + l], x2 = a[128 + l];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rep; ++r) {
        const float2v nx0 = {-x0, -x0}, nx1 = {-x1, -x1}, nx2 = {-x2, -x2};
#pragma unroll
        for (int pp = 0; pp < 2 * NG; ++pp) {
            const int cc = 2 * pp;
            const float2v s0 = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(x0), cc)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x0), cc + 1))};
            const float2v s1 = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(x1), cc)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x1), cc + 1))};
            const float2v s2 = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(x2), cc)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x2), cc + 1))};
            float2v acc = vv[pp];
            acc = __builtin_elementwise_fma(nx0, s0, acc);
            acc = __builtin_elementwise_fma(nx1, s1, acc);
            acc = __builtin_elementwise_fma(nx2, s2, acc);
            vv[pp] = acc;
            asm volatile("" : "+v"(vv[pp]));
        }
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2));
    }
    float s = 0.f;
    for (int g = 0; g < 2 * NG; ++g) s += vv[g][0] + vv[g][1];
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[l] = s;
    if (l == 0) cyc[0] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

template <int ABID>
static int check(const std::vector<float>& ha, const std::vector<float>& hb, const std::vector<float>& hc, float* da, float* db, float* dc, float* dout)
{
    std::vector<float> ho(256);
    hipLaunchKernelGGL(probe<ABID>, dim3(1), dim3(64), 0, 0, da, db, dc, dout);
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpy(ho.data(), dout, 256 * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    int bad = 0;
    double worst = 0.0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const float want = fmaf(ha[4 * ABID + i], hb[l], hc[4 * l + i]);
            const double e = fabs((double)ho[4 * l + i] - (double)want);
            if (e > worst) worst = e;
            if (ho[4 * l + i] != want) ++bad;
        }
    printf("abid %2d: entries not bit-equal to fmaf(A[4 abid + i], B[l], C) : %d of 256, worst abs diff %.3g\n", ABID, bad, worst);
    return worst > 1e-5 ? 1 : 0;
}

int main()
{
    std::vector<float> ha(192), hb(64), hc(256);
    srand(7);
    for (auto& v : ha) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : hb) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : hc) v = (float)rand() / RAND_MAX - 0.5f;
    float *da, *db, *dc, *dout;
    long long* dcyc;
    CK(hipMalloc(&da, 192 * 4)); CK(hipMalloc(&db, 64 * 4)); CK(hipMalloc(&dc, 256 * 4)); CK(hipMalloc(&dout, 256 * 4)); CK(hipMalloc(&dcyc, 8));
    CK(hipMemcpy(da, ha.data(), 192 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), 64 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, hc.data(), 256 * 4, hipMemcpyHostToDevice));
    int rc = 0;
    rc |= check<0>(ha, hb, hc, da, db, dc, dout);
    rc |= check<3>(ha, hb, hc, da, db, dc, dout);
    rc |= check<7>(ha, hb, hc, da, db, dc, dout);
    rc |= check<15>(ha, hb, hc, da, db, dc, dout);
    const int rep = 2000;
    long long c;
#define TIME(K, NG) do { hipLaunchKernelGGL((K<NG>), dim3(1), dim3(64), 0, 0, da, dout, dcyc, rep); CK(hipDeviceSynchronize()); \
        hipLaunchKernelGGL((K<NG>), dim3(1), dim3(64), 0, 0, da, dout, dcyc, rep); CK(hipDeviceSynchronize()); \
        CK(hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost)); printf("%-14s groups %d: %.1f cycles per rank-3 update (%.2f per 4 columns)\n", #K, NG, (double)c / rep, (double)c / rep / NG); } while (0)
    TIME(time_mfma, 1); TIME(time_mfma, 2); TIME(time_mfma, 4); TIME(time_mfma, 8);
    TIME(time_readlane, 1); TIME(time_readlane, 2); TIME(time_readlane, 4); TIME(time_readlane, 7);
    return rc;
}
