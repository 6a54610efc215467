// Batched contact-schedule kernels: one thread per (problem, foot).  The lists are a few dozen
// bytes per foot and the work is a handful of comparisons per knot: these kernels exist so that a Monte-Carlo
// roll-out (solve -> plant -> merge -> sample -> shift, every tick) never leaves HBM, not because they are hot.
// The logic itself is in cmpc_contacts.h, shared with the host entry points of the C ABI.
#include "cmpc_contacts.h"

namespace {

__global__ __launch_bounds__(128) void cmpc_contacts_merge_kernel(int B, int M, double now, const double* __restrict__ plan_t,
                                                                  const float* __restrict__ plan_pose, const int* __restrict__ plan_n,
                                                                  const double* __restrict__ mpc_t, const float* __restrict__ mpc_pose,
                                                                  const int* __restrict__ mpc_n, double* __restrict__ out_t,
                                                                  float* __restrict__ out_pose, int* __restrict__ out_n, int* __restrict__ ok)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;   // problem * 2 + foot
    if (e >= 2 * B) return;
    const size_t o = (size_t)e * M;
    // list lengths outside 0..M (the host entry point rejects them with CMPC_ERR_ARG): nothing is read, the merged list is empty, ok = 0
    const bool sane = plan_n[e] >= 0 && plan_n[e] <= M && mpc_n[e] >= 0 && mpc_n[e] <= M;
    if (!sane) out_n[e] = 0;
    const bool good = sane && cmpc_merge_foot(now, plan_t + 2 * o, plan_pose + 7 * o, plan_n[e], mpc_t + 2 * o, mpc_pose + 7 * o, mpc_n[e], M,
                                              out_t + 2 * o, out_pose + 7 * o, out_n + e);
    if (ok && !good) atomicAnd(ok + (e >> 1), 0);   // (ok[] starts at 1: cmpc_launch_contacts_merge fills it)
}

__global__ __launch_bounds__(128) void cmpc_fill_int_kernel(int n, int v, int* __restrict__ dst)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) dst[e] = v;
}

// one workgroup (one wave) per problem, one thread per (foot, stage); the landing knots from the stages' contact flags in LDS.  (One thread per foot walking its N
// stages through global-memory latency took 33 us per launch whatever the batch: profiles/r04_rollout_tick_overhead.txt.)
__global__ __launch_bounds__(64) void cmpc_contacts_sample_kernel(int B, int N, int M, double dt, double now, const double* __restrict__ t,
                                                                  const float* __restrict__ pose, const int* __restrict__ n,
                                                                  const float* __restrict__ box /* upper[6] | lower[6] */,
                                                                  float* __restrict__ P, int* __restrict__ land)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const CmpcIdx L{N};
    __shared__ unsigned char acts[2][CMPC_NMAX];
    // An empty list (cmpc_merge_foot leaves one where the reference's updateContactPhaseList returns false, CentroidalMPCBlock.cpp:70-77,
    // and the reference then aborts the tick, :603-607) or a length beyond M has no owner to sample: the foot's blocks of P are left
    // as they are and the landing knot reads -2 (the host entry point returns CMPC_ERR_ARG for the same input).
    for (int e2 = tid; e2 < 2 * N; e2 += 64) {
        const int c = e2 / N, k = e2 - c * N, e = 2 * b + c;
        const size_t o = (size_t)e * M;
        if (n[e] >= 1 && n[e] <= M)
            acts[c][k] = cmpc_sample_stage(N, dt, now, c, k, t + 2 * o, pose + 7 * o, n[e], box, box + 6, P + (size_t)b * L.np()) ? 1 : 0;
    }
    __syncthreads();
    if (tid < 2 && land) {
        const int e = 2 * b + tid;
        land[e] = (n[e] < 1 || n[e] > M) ? -2 : cmpc_landing_knot(N, [&](int k) { return acts[tid][k] != 0; });
    }
}

// step adjustment: the next contact of every foot that lands inside the horizon takes the optimised landing position
__global__ __launch_bounds__(128) void cmpc_contacts_adjust_kernel(int B, int N, int M, double now, const float* __restrict__ X,
                                                                   const int* __restrict__ land, const double* __restrict__ t,
                                                                   float* __restrict__ pose, const int* __restrict__ n)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * B) return;
    const int b = e >> 1, c = e & 1;
    const int lk = land[e];
    if (lk < 0 || lk > N || n[e] < 1 || n[e] > M) return;   // (no landing inside the horizon, or a list that was not sampled)
    const CmpcIdx L{N};
    const size_t o = (size_t)e * M;
    const int nx = cmpc_next_contact(t + 2 * o, n[e], now);
    if (nx < 0) return;
    const float* x = X + (size_t)b * L.nx() + L.oPos(c) + 3 * lk;
    for (int i = 0; i < 3; ++i) pose[7 * (o + nx) + i] = x[i];
}

// measured state (and external wrench) into the parameter rows of every problem: setState on the device
__global__ __launch_bounds__(128) void cmpc_write_state_kernel(int B, int N, const float* __restrict__ state, const float* __restrict__ wrench,
                                                               float* __restrict__ P)
{
    const int b = blockIdx.x;
    const CmpcIdx L{N};
    float* p = P + (size_t)b * L.np();
    for (int e = threadIdx.x; e < 9; e += blockDim.x) p[L.pCom0() + e] = state[9 * (size_t)b + e];
    if (wrench)
        for (int e = threadIdx.x; e < 3 * N; e += blockDim.x) {
            const int k = e / 3, i = e % 3;
            p[L.pFext() + e] = wrench[((size_t)b * N + k) * 6 + i];
            p[L.pText() + e] = wrench[((size_t)b * N + k) * 6 + 3 + i];
        }
}

// 8f-3 on the device: comRef / hRef of every problem from the planner's trajectories, one thread per knot (cmpc_resample_reference_knot)
__global__ __launch_bounds__(64) void cmpc_reference_from_planner_kernel(int B, int N, int n_in, double dt, double in_dt, double t_offset, double robot_mass,
                                                                         double com_height, const float* __restrict__ com_in, const float* __restrict__ h_in,
                                                                         float* __restrict__ P)
{
    const int b = blockIdx.x;
    const CmpcIdx L{N};
    float* p = P + (size_t)b * L.np();
    for (int k = threadIdx.x; k <= N; k += 64)
        cmpc_resample_reference_knot(com_in + (size_t)b * n_in * 3, h_in + (size_t)b * n_in * 3, n_in, in_dt, t_offset, dt, k, robot_mass, com_height,
                                     p + L.pComref() + 3 * k, p + L.pHref() + 3 * k);
}

}  // namespace

extern "C" int cmpc_launch_reference_from_planner(int B, int N, int n_in, double dt, double in_dt, double t_offset, double robot_mass, double com_height,
                                                  const float* com_in, const float* h_in, float* P, hipStream_t stream)
{
    hipLaunchKernelGGL(cmpc_reference_from_planner_kernel, dim3(B), dim3(64), 0, stream, B, N, n_in, dt, in_dt, t_offset, robot_mass, com_height, com_in, h_in, P);
    return (int)hipGetLastError();
}

extern "C" int cmpc_launch_contacts_merge(int B, int M, double now, const double* plan_t, const float* plan_pose, const int* plan_n,
                                          const double* mpc_t, const float* mpc_pose, const int* mpc_n, double* out_t, float* out_pose,
                                          int* out_n, int* ok, hipStream_t stream)
{
    if (ok) hipLaunchKernelGGL(cmpc_fill_int_kernel, dim3((B + 127) / 128), dim3(128), 0, stream, B, 1, ok);
    hipLaunchKernelGGL(cmpc_contacts_merge_kernel, dim3((2 * B + 127) / 128), dim3(128), 0, stream, B, M, now, plan_t, plan_pose, plan_n,
                       mpc_t, mpc_pose, mpc_n, out_t, out_pose, out_n, ok);
    return (int)hipGetLastError();
}

extern "C" int cmpc_launch_contacts_sample(int B, int N, int M, double dt, double now, const double* t, const float* pose, const int* n,
                                           const float* box, float* P, int* land, hipStream_t stream)
{
    hipLaunchKernelGGL(cmpc_contacts_sample_kernel, dim3(B), dim3(64), 0, stream, B, N, M, dt, now, t, pose, n, box, P, land);
    return (int)hipGetLastError();
}

extern "C" int cmpc_launch_contacts_adjust(int B, int N, int M, double now, const float* X, const int* land, const double* t, float* pose,
                                           const int* n, hipStream_t stream)
{
    hipLaunchKernelGGL(cmpc_contacts_adjust_kernel, dim3((2 * B + 127) / 128), dim3(128), 0, stream, B, N, M, now, X, land, t, pose, n);
    return (int)hipGetLastError();
}

extern "C" int cmpc_launch_write_state(int B, int N, const float* state, const float* wrench, float* P, hipStream_t stream)
{
    hipLaunchKernelGGL(cmpc_write_state_kernel, dim3(B), dim3(128), 0, stream, B, N, state, wrench, P);
    return (int)hipGetLastError();
}
