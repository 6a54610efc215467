// A Monte-Carlo driver in host C++ over the C ABI (include/cmpc.h), one process per GPU: the shape of the reference's main() -- build the blocks, run them
// (src/centroidal-mpc-walking/src/Main.cpp:90-134) -- for a sharded batch instead of one robot.  Rank r of W takes the contiguous shard
// [r B/W, (r+1) B/W) of BASELINE config 4's 65 536 problems (here: perturbed-CoM standing problems, made by the class-shaped setters and read back with
// cmpc_get_parameters), solves it on its GPU (cmpc_solve_device: operands resident in HBM), packs the compact record of every problem
// (cmpc_compact_output_device) and gathers all ranks' records with RCCL over xGMI through cmpc_allgather_compact_device.  No data-path collective before the
// gather: the problems are independent.
//
//   hipcc -std=c++17 -I include examples/montecarlo_allgather.cpp -L <pkg> -lcmpc_hip -lrccl -Wl,-rpath,<pkg> -o montecarlo_allgather
//   for r in 0 1 .. W-1: ./montecarlo_allgather $r W /tmp/cmpc_nccl_id [total] &      (rank 0 writes the ncclUniqueId to the file, the others read it)
//
// With W = 1 it runs on a one-GPU box (tests/test_gpu_facade.py builds and runs it that way); W > 1 needs W GPUs and is NOT measured on this pool's one-GPU
// lease -- the gather path itself is the one tests/test_gpu_setters.py checks with a one-rank communicator.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "cmpc.h"

#define CK(call)                                                                                              \
    do {                                                                                                      \
        if ((call) != 0) { std::fprintf(stderr, "%s failed: %s\n", #call, cmpc_last_error(nullptr)); return 1; } \
    } while (0)
#define HK(call)                                                                                              \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; }    \
    } while (0)

int main(int argc, char** argv)
{
    const int rank = argc > 1 ? std::atoi(argv[1]) : 0, world = argc > 2 ? std::atoi(argv[2]) : 1;
    const char* idfile = argc > 3 ? argv[3] : "/tmp/cmpc_nccl_id";
    const int total = argc > 4 ? std::atoi(argv[4]) : 65536;
    if (world < 1 || rank < 0 || rank >= world || total % world) { std::fprintf(stderr, "usage: rank world idfile [total divisible by world]\n"); return 2; }
    const int B = total / world;                                  // equal shards (pad a ragged one)
    int ndev = 0;
    HK(hipGetDeviceCount(&ndev));
    const int dev = rank % ndev;
    HK(hipSetDevice(dev));
    // the communicator: one rank per process and GPU
    ncclUniqueId id;
    if (rank == 0) {
        if (ncclGetUniqueId(&id) != ncclSuccess) return 3;
        FILE* f = std::fopen(idfile, "wb");
        if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) return 3;
        std::fclose(f);
    } else {
        for (int tries = 0;; ++tries) {
            FILE* f = std::fopen(idfile, "rb");
            if (f && std::fread(&id, sizeof(id), 1, f) == 1) { std::fclose(f); break; }
            if (f) std::fclose(f);
            if (tries > 600) return 3;
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
    }
    ncclComm_t comm;
    if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess) { std::fprintf(stderr, "ncclCommInitRank failed\n"); return 3; }

    // initialize(): ergoCubGazeboV1/centroidal_mpc.ini
    cmpc_config cfg;
    cmpc_default_config(&cfg);
    cmpc_handle h = nullptr;
    CK(cmpc_create(&cfg, B, dev, &h));
    int nx, np, ng, nj, nh;
    CK(cmpc_dims(cfg.horizon, &nx, &np, &ng, &nj, &nh));
    const int N = cfg.horizon;
    // this rank's shard of the problem set through the class-shaped setters: problem b of the shard is global problem rank B + b
    std::vector<float> state((size_t)B * 9), ref((size_t)B * 3 * (N + 1)), href((size_t)B * 3 * (N + 1), 0.f);
    std::vector<float> R((size_t)B * 2 * N * 9, 0.f), up((size_t)B * 2 * N * 3), lo((size_t)B * 2 * N * 3), en((size_t)B * 2 * N, 1.f),
        nom((size_t)B * 2 * (N + 1) * 3), cur((size_t)B * 2 * 3);
    const float bu[2][3] = {{0.01f, 0.05f, 0.f}, {0.01f, 0.f, 0.f}}, bl[2][3] = {{-0.01f, 0.f, 0.f}, {-0.01f, -0.05f, 0.f}};
    for (int b = 0; b < B; ++b) {
        unsigned s = 2654435761u * (unsigned)(rank * B + b + 1);                   // the global problem number seeds its disturbance
        auto u = [&](float a) { s = s * 1664525u + 1013904223u; return a * ((float)(s >> 8) / 8388608.f - 1.f); };
        float* st = &state[(size_t)b * 9];
        st[0] = u(0.02f); st[1] = u(0.02f); st[2] = 0.7f + u(0.02f);
        for (int i = 3; i < 6; ++i) st[i] = u(0.1f);
        for (int i = 6; i < 9; ++i) st[i] = u(0.05f);
        for (int k = 0; k <= N; ++k) { float* c = &ref[((size_t)b * (N + 1) + k) * 3]; c[0] = 0.f; c[1] = 0.f; c[2] = 0.7f; }
        for (int ct = 0; ct < 2; ++ct) {
            const float y = ct == 0 ? 0.08f : -0.08f;
            for (int k = 0; k < N; ++k) {
                float* Rk = &R[(((size_t)b * 2 + ct) * N + k) * 9];
                Rk[0] = Rk[4] = Rk[8] = 1.f;
                for (int i = 0; i < 3; ++i) { up[(((size_t)b * 2 + ct) * N + k) * 3 + i] = bu[ct][i]; lo[(((size_t)b * 2 + ct) * N + k) * 3 + i] = bl[ct][i]; }
            }
            for (int k = 0; k <= N; ++k) { float* p = &nom[(((size_t)b * 2 + ct) * (N + 1) + k) * 3]; p[0] = 0.f; p[1] = y; p[2] = 0.f; }
            float* c = &cur[((size_t)b * 2 + ct) * 3]; c[0] = 0.f; c[1] = y; c[2] = 0.f;
        }
    }
    CK(cmpc_set_state(h, state.data(), nullptr));
    CK(cmpc_set_reference(h, ref.data(), href.data()));
    CK(cmpc_set_contacts(h, R.data(), up.data(), lo.data(), en.data(), nom.data(), cur.data()));
    // what the setters wrote, as one parameter matrix P[B][n_p]; and the cold start of SURVEY 8d
    std::vector<float> P((size_t)B * np), X0((size_t)B * nx, 0.f);
    CK(cmpc_get_parameters(h, P.data()));
    for (int b = 0; b < B; ++b) {
        float* x = &X0[(size_t)b * nx];
        for (int k = 0; k <= N; ++k)
            for (int i = 0; i < 3; ++i) x[3 * k + i] = state[(size_t)b * 9 + i];
        for (int ct = 0; ct < 2; ++ct) {
            float* pos = x + 9 * (N + 1) + ct * (18 * N + 3);
            std::memcpy(pos, &nom[((size_t)b * 2 + ct) * (N + 1) * 3], sizeof(float) * 3 * (N + 1));
            for (int j = 0; j < 4; ++j)
                for (int k = 0; k < N; ++k) pos[3 * (N + 1) + 3 * N + j * 3 * N + 3 * k + 2] = (float)(cfg.gravity / 8.0);
        }
    }
    const int W = 3 * (N + 1) + 38;
    float *dP, *dX0, *dX, *dInfo, *dLocal, *dAll;
    HK(hipMalloc(&dP, sizeof(float) * P.size())); HK(hipMalloc(&dX0, sizeof(float) * X0.size())); HK(hipMalloc(&dX, sizeof(float) * X0.size()));
    HK(hipMalloc(&dInfo, sizeof(float) * CMPC_INFO * B)); HK(hipMalloc(&dLocal, sizeof(float) * (size_t)B * W));
    HK(hipMalloc(&dAll, sizeof(float) * (size_t)world * B * W));
    HK(hipMemcpy(dP, P.data(), sizeof(float) * P.size(), hipMemcpyHostToDevice));
    HK(hipMemcpy(dX0, X0.data(), sizeof(float) * X0.size(), hipMemcpyHostToDevice));
    // advance() for the shard, then the gather: three asynchronous calls on the handle's stream
    hipStream_t st = (hipStream_t)cmpc_stream(h);
    const auto t0 = std::chrono::steady_clock::now();
    CK(cmpc_solve_device(h, dP, dX0, dX, dInfo, nullptr));
    CK(cmpc_compact_output_device(h, dX, dInfo, dLocal, nullptr));
    CK(cmpc_allgather_compact_device(h, comm, world, dLocal, dAll, nullptr));
    HK(hipStreamSynchronize(st));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::vector<float> all((size_t)world * B * W);
    HK(hipMemcpy(all.data(), dAll, sizeof(float) * all.size(), hipMemcpyDeviceToHost));
    int bad = 0;
    double fz = 0.0;
    for (size_t b = 0; b < (size_t)world * B; ++b) {
        const float* rec = &all[b * W];
        if (rec[W - 1] != 0.f) ++bad;                               // status column
        for (int j = 0; j < 8; ++j) fz += rec[3 * (N + 1) + 3 * j + 2];
    }
    std::printf("rank %d of %d: %d problems per rank, %d gathered, %d not converged, mean vertical force per problem %.4f (gravity %.4f), solve + gather %.2f ms (kernel %.2f ms)\n",
                rank, world, B, world * B, bad, fz / ((double)world * B), cfg.gravity, ms, cmpc_last_solve_ms(h));
    ncclCommDestroy(comm);
    (void)hipFree(dP); (void)hipFree(dX0); (void)hipFree(dX); (void)hipFree(dInfo); (void)hipFree(dLocal); (void)hipFree(dAll);
    cmpc_destroy(h);
    return bad ? 4 : 0;
}
