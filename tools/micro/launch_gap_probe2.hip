// Developer probe (GPU box): the idle time between back-to-back launches on one stream as a function of what the kernel did -- its duration, the dirty data it
// leaves behind, its private segment.  Every launch stamps the 100 MHz wall clock (s_memrealtime) when its first wave starts and when its last wave ends.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/launch_gap_probe2.hip -o /tmp/p2 && /tmp/p2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

template <int PRIV>
__global__ __launch_bounds__(512) void k(unsigned long long* stamps, float* junk, int words, long long spin_us, int every)
{
    extern __shared__ float lds[];
    volatile float priv[PRIV > 0 ? PRIV : 1];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) atomicMin(&stamps[0], t0);
    lds[threadIdx.x] = threadIdx.x;
    const size_t base = ((size_t)blockIdx.x * 512 + threadIdx.x);
    int w = 0;
    // spin, writing `words` floats per thread spread over the duration (`every` writes per 10 us)
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100ull) {
        for (int q = 0; q < every && w < words; ++q, ++w) junk[base + (size_t)w * 256 * 512] = w;
        if (PRIV > 0) priv[(w + threadIdx.x) % PRIV] = w;
        __builtin_amdgcn_s_sleep(32);
    }
    for (; w < words; ++w) junk[base + (size_t)w * 256 * 512] = w;
    float s = lds[(threadIdx.x + 1) & 511];
    if (PRIV > 0) s += priv[(threadIdx.x >> 3) % PRIV];
    if (s == -1.f) junk[0] = s;
    if (threadIdx.x == 0) atomicMax(&stamps[1], __builtin_amdgcn_s_memrealtime());
}

template <int PRIV>
static int run(const char* name, unsigned long long* st, float* junk, int words, long long spin_us, int every)
{
    const int lds = 147 * 1024, K = 24;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<PRIV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    std::vector<unsigned long long> init(2 * K);
    for (int i = 0; i < K; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
    CK(hipMemcpy(st, init.data(), 16 * K, hipMemcpyHostToDevice));
    for (int i = 0; i < K; ++i) hipLaunchKernelGGL(k<PRIV>, dim3(256), dim3(512), lds, 0, st + 2 * i, junk, words, spin_us, every);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(init.data(), st, 16 * K, hipMemcpyDeviceToHost));
    std::vector<double> gaps, durs;
    for (int i = 4; i < K; ++i) { gaps.push_back((double)(init[2 * i] - init[2 * i - 1]) / 100.0); durs.push_back((double)(init[2 * i + 1] - init[2 * i]) / 100.0); }
    std::sort(gaps.begin(), gaps.end()); std::sort(durs.begin(), durs.end());
    printf("%-58s kernel %8.1f us, idle between launches %6.1f us (median of %d)\n", name, durs[durs.size() / 2], gaps[gaps.size() / 2], (int)gaps.size());
    return 0;
}

int main()
{
    unsigned long long* st; float* junk;
    CK(hipMalloc(&st, 16 * 64));
    CK(hipMalloc(&junk, (size_t)256 * 512 * 4 * 512));
    int rc = 0;
    rc |= run<0>("empty", st, junk, 0, 0, 0);
    rc |= run<0>("850 us spin", st, junk, 0, 850, 0);
    rc |= run<130>("850 us spin, 520 B private (touched every ~3 us)", st, junk, 0, 850, 0);
    rc |= run<0>("850 us spin, 64 MB of stores spread over it", st, junk, 128, 850, 2);
    rc |= run<0>("850 us spin, 128 MB of stores spread over it", st, junk, 256, 850, 4);
    rc |= run<0>("850 us spin, 64 MB of stores at the end", st, junk, 128, 850, 0);
    rc |= run<0>("100 us spin", st, junk, 0, 100, 0);
    return rc;
}
