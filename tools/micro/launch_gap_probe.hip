// Developer probe (GPU box): what back-to-back launches of a 256 x 512-thread, 147 KB-LDS kernel cost on one stream -- empty, with a private segment,
// and with tens of MB of fire-and-forget stores per launch (dirty L2 lines that the end-of-kernel release has to write back).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/launch_gap_probe.hip -o gpurun_out/launch_gap_probe && gpurun_out/launch_gap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

// spins ~`cycles` shader clocks, then optionally writes `words` floats per thread to `junk`
template <int PRIV>
__global__ __launch_bounds__(512) void k(float* out, float* junk, int words, long long cycles)
{
    extern __shared__ float lds[];
    volatile float priv[PRIV > 0 ? PRIV : 1];
    if (PRIV > 0) { priv[threadIdx.x % PRIV] = threadIdx.x; priv[(threadIdx.x + 7) % PRIV] = 1.f; }   // (the array must live in scratch; two words of it are touched)
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    const size_t base = ((size_t)blockIdx.x * 512 + threadIdx.x);
    for (int w = 0; w < words; ++w) junk[base + (size_t)w * 256 * 512] = w;
    float s = lds[(threadIdx.x + 1) & 511];
    if (PRIV > 0) s += priv[(threadIdx.x >> 3) % PRIV];
    if (s == -1.f) out[0] = s;
}

template <int PRIV>
static int run(const char* name, float* out, float* junk, int words, long long cycles, int lds)
{
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<PRIV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<PRIV>, dim3(256), dim3(512), lds, 0, out, junk, words, cycles);
    CK(hipDeviceSynchronize());
    const int K = 50;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k<PRIV>, dim3(256), dim3(512), lds, 0, out, junk, words, cycles);
    CK(hipDeviceSynchronize());
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
    printf("%-46s %8.1f us per launch\n", name, us);
    return 0;
}

int main()
{
    float *out, *junk;
    CK(hipMalloc(&out, 4));
    CK(hipMalloc(&junk, (size_t)256 * 512 * 4 * 512));   // up to 512 words per thread = 268 MB
    const long long cyc = 100 * 1000 * 2;   // ~100 us at 100 MHz s_memtime?  (s_memtime counts at a fixed 100 MHz on gfx9: 10 ns ticks)
    const long long ticks = 20000;          // 200 us if 100 MHz
    (void)cyc;
    int rc = 0;
    rc |= run<0>("no private segment", out, junk, 0, 0, 147 * 1024);
    rc |= run<4>("16 B private", out, junk, 0, 0, 147 * 1024);
    rc |= run<16>("64 B private", out, junk, 0, 0, 147 * 1024);
    rc |= run<32>("128 B private", out, junk, 0, 0, 147 * 1024);
    rc |= run<64>("256 B private", out, junk, 0, 0, 147 * 1024);
    rc |= run<68>("272 B private", out, junk, 0, 0, 147 * 1024);
    rc |= run<96>("384 B private", out, junk, 0, 0, 147 * 1024);
    rc |= run<130>("520 B private", out, junk, 0, 0, 147 * 1024);
    rc |= run<256>("1024 B private", out, junk, 0, 0, 147 * 1024);
    rc |= run<0>("no private segment, 64 MB of stores", out, junk, 128, 0, 147 * 1024);
    rc |= run<0>("no private segment, 128 MB of stores", out, junk, 256, 0, 147 * 1024);
    return rc;
}
