// C ABI of libcmpc_hip.so (include/cmpc.h): handle management, host<->device plumbing, launches.
#include "../../include/cmpc.h"
#include "cmpc_contacts.h"
#include "cmpc_device.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <vector>

extern "C" size_t cmpc_solver_lds_bytes(int N, int factors_global);
extern "C" int cmpc_launch_solver(const CmpcParams* prm, size_t lds_bytes, hipStream_t stream);
extern "C" int cmpc_prepare_solver(int N, int factors_global, size_t lds_bytes);
extern "C" int cmpc_launch_nlp_eval(const CmpcParams* prm, const float* dX, const float* dP, const float* dLamG,
                                    float lam_f, float* dF, float* dG, float* dGradF, float* dJac, float* dHess,
                                    hipStream_t stream);
extern "C" int cmpc_launch_nlp_grad(const CmpcParams* prm, const float* dX, const float* dP, const float* dLamG, float lam_f, float* dGradX,
                                    float* dGradP, hipStream_t stream);
extern "C" int cmpc_launch_warm_shift(const CmpcParams* prm, const float* dXprev, float* dX0, hipStream_t stream);
extern "C" int cmpc_launch_contacts_merge(int B, int M, double now, const double* plan_t, const float* plan_pose, const int* plan_n,
                                          const double* mpc_t, const float* mpc_pose, const int* mpc_n, double* out_t, float* out_pose,
                                          int* out_n, int* ok, hipStream_t stream);
extern "C" int cmpc_launch_contacts_sample(int B, int N, int M, double dt, double now, const double* t, const float* pose, const int* n,
                                           const float* box, float* P, int* land, hipStream_t stream);
extern "C" int cmpc_launch_contacts_adjust(int B, int N, int M, double now, const float* X, const int* land, const double* t, float* pose,
                                           const int* n, hipStream_t stream);
extern "C" int cmpc_launch_write_state(int B, int N, const float* state, const float* wrench, float* P, hipStream_t stream);
extern "C" int cmpc_launch_compact(int N, int B, const float* dX, const float* dInfo, float* dOut, hipStream_t stream);
extern "C" int cmpc_launch_reference_from_planner(int B, int N, int n_in, double dt, double in_dt, double t_offset, double robot_mass, double com_height,
                                                  const float* com_in, const float* h_in, float* P, hipStream_t stream);
extern "C" int cmpc_launch_tick_pre(int B, int N, int M, double dt, double now, int merge, const double* plan_t, const float* plan_pose, const int* plan_n,
                                    const double* prev_t, const float* prev_pose, const int* prev_n, double* list_t, float* list_pose, int* list_n, int* ok,
                                    int* land, const float* box, const float* state, const float* wrench, float* P, const float* Xprev, float* X0,
                                    const float* plan_com, const float* plan_h, int plan_knots, double plan_dt, double plan_t_offset, double robot_mass,
                                    double com_height, hipStream_t stream);
extern "C" int cmpc_launch_tick_post(int B, int N, int M, double now, float grav, const float* dCorners, const float* dX, const float* dP,
                                     const float* dStateIn, float* dStateOut, float* dZmp, float h, int nsub, float zx, float zy, const int* land,
                                     const double* t, float* pose, const int* n, hipStream_t stream);
extern "C" int cmpc_launch_plant_step(int N, int B, float grav, const float* dCorners, const float* dX, const float* dP,
                                      const float* dStateIn, float* dStateOut, float* dZmp, float h, int nsub, float zx, float zy,
                                      hipStream_t stream);

struct cmpc_handle_s {
    cmpc_config cfg;
    CmpcLayout L;
    int B = 0, device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    bool timing = true;          // record the event pair around every solve launch (cmpc_set_timing)
    float* dP = nullptr;
    float* dX0 = nullptr;
    float* dX = nullptr;
    float* dInfo = nullptr;
    CmpcConsts* dConsts = nullptr;
    float* dScratch = nullptr;   // factor storage when the horizon's LDS image exceeds 160 KiB
    float* dBox = nullptr;       // bounding-box limits upper[2][3] | lower[2][3] of the schedule sampler
    float* dDuals = nullptr;     // costates, slacks, multipliers of the last solve (warm start with duals: allocated by cmpc_create when the developer knob CMPC_WARM_DUALS is set)
    int warm_duals = 0;          // 0: primal shift only (default); 1: + costates; 2: + multipliers
    float hBox[12] = {0};
    bool box_set = false;
    long long scratch_stride = 0;
    std::vector<float> hP, hX0;  // host staging for the class-shaped setters
    bool have_solution = false, x0_set = false;
    // pinned host mirror of the handle's solution and status words, filled by cmpc_advance in the same synchronisation as the solve (the class surface's
    // getOutput / cmpc_get_solution then cost no GPU call: a pageable device-to-host copy of their own was 25 us of a 0.83 ms tick at B = 1)
    float* hXpin = nullptr;
    float* hInfoPin = nullptr;
    bool hX_valid = false;
    bool warm = false;           // class path only (cmpc_set_initial_guess(.., 1) -> cmpc_advance): the handle's own dX0 is a shifted previous solution
    double mu_warm = 1e-2, floor_warm = 1e-2;  // measured: 1e-2 saves 35 % (standing) / 15 % (walking) of the iterations; 1e-4 can stall
    int warm_budget = 14, warm_no_restart = 0;  // cmpc_set_warm_policy (CmpcParams); 14: measured on the walking roll-out (profiles/r03_walking_rollout.txt)
    bool force_warm = false;     // developer knob CMPC_FORCE_WARM (read once, at cmpc_create)
    float mu_adapt = 3.5f;       // cold starts: mu0 = clamp(mu_adapt ep0^2, 0.03, 0.5) (developer knob CMPC_MU_ADAPT, read once)
    size_t lds = 0;
    std::string err;
};

static thread_local std::string g_err;

static int fail(cmpc_handle h, int code, const std::string& msg)
{
    if (h) h->err = msg;
    g_err = msg;
    return code;
}
#define HIPCHK(h, call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(h, CMPC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));  \
    } while (0)

static void fill_consts(cmpc_handle h, CmpcConsts& q);

extern "C" {

void cmpc_default_config(cmpc_config* c)
{
    // config/robots/ergoCubGazeboV1/centroidal_mpc.ini:3-42
    std::memset(c, 0, sizeof(*c));
    c->horizon = 20;
    c->sampling_time = 0.06;
    c->friction_coefficient = 0.33;
    c->gravity = 9.80665;
    c->com_weight[0] = 10; c->com_weight[1] = 10; c->com_weight[2] = 200;
    c->angular_momentum_weight = 1e2;
    c->contact_position_weight = 2e3;
    c->force_rate_of_change_weight[0] = c->force_rate_of_change_weight[1] = c->force_rate_of_change_weight[2] = 10;
    c->contact_force_symmetry_weight = 100;
    const double cr[4][3] = {{0.08, 0.01, 0}, {0.08, -0.01, 0}, {-0.08, -0.01, 0}, {-0.08, 0.01, 0}};
    for (int ct = 0; ct < 2; ++ct)
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 3; ++i) c->corners[ct][j][i] = cr[j][i];
    c->max_iterations = 40;
    c->tolerance = 1e-6;
    c->step_tolerance = 1e-4;
    c->mu_init = 0.0;  // <= 0: chosen per problem from its initial infeasibility
    c->mu_min = 5e-8;
    c->exact_hessian = 1;
    c->final_extrapolation = 1;
    c->tail_stages = 3;
    c->tail_iterations = 2;
    c->tail_trigger = 2e-5;
}

int cmpc_dims(int N, int* nx, int* np, int* ng, int* nnzj, int* nnzh)
{
    if (N < 1) return CMPC_ERR_ARG;
    if (nx) *nx = 45 * N + 15;
    if (np) *np = 50 * N + 27;
    if (ng) *ng = 53 * N + 15;
    if (nnzj) *nnzj = 243 * N + 15;
    if (nnzh) *nnzh = 348 * N - 36;
    return CMPC_OK;
}

const char* cmpc_last_error(cmpc_handle h) { return h ? h->err.c_str() : g_err.c_str(); }
int cmpc_batch(cmpc_handle h) { return h ? h->B : 0; }
void* cmpc_stream(cmpc_handle h) { return h ? (void*)h->stream : nullptr; }

int cmpc_create(const cmpc_config* cfg, int batch, int device, cmpc_handle* out)
{
    if (!cfg || !out || batch < 1) return fail(nullptr, CMPC_ERR_ARG, "cmpc_create: null argument or batch < 1");
    if (cfg->horizon < 2 || cfg->horizon > CMPC_NMAX)
        return fail(nullptr, CMPC_ERR_ARG, "cmpc_create: horizon must be in [2, " + std::to_string(CMPC_NMAX) + "]");
    if (!(cfg->sampling_time > 0) || !(cfg->friction_coefficient > 0))
        return fail(nullptr, CMPC_ERR_ARG, "cmpc_create: sampling_time and friction_coefficient must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, CMPC_ERR_HIP, "cmpc_create: no HIP device (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(nullptr, CMPC_ERR_ARG, "cmpc_create: bad device index");
    cmpc_handle h = new cmpc_handle_s();
    h->cfg = *cfg;
    if (h->cfg.max_iterations <= 0) h->cfg.max_iterations = 40;
    // Default tolerance by horizon: 1e-6 up to N = 20; 3e-7 beyond (the step tolerance follows it).  The error of the far-horizon CoM velocity is fed by the
    // complementarity products that still lag above the barrier floor at termination (max t z <= tolerance), and it grows with the number of stages behind the
    // first knots.  Config 5 (N = 30), worst of 5 unseen seeds x 512 problems against the float64 oracle (profiles/r04_accuracy_sweep.txt), barrier floor 5e-8:
    // tolerance 1e-6 -> 1.05e-4 (round 3), 5e-7 -> 9.3e-5, 4e-7 -> 9.3e-5, 3e-7 -> 6.8e-5 at 8.57 / 8.95 / 9.08 / 9.24 iterations on the mean.
    if (!(h->cfg.tolerance > 0)) h->cfg.tolerance = h->cfg.horizon > 20 ? 3e-7 : 1e-6;
    if (!(h->cfg.step_tolerance > 0)) h->cfg.step_tolerance = 100.0 * h->cfg.tolerance;
    // 0.05 x tolerance: the same iteration counts as tolerance / 10 (the barrier decreases superlinearly at the end)
    // at 0.7 x the sqrt(mu) bias of the nearly degenerate rows; float32 factorisations start to fail at 2e-8 (8 of 512
    // problems of config 5), 1e-8 loses most of config 3
    // ... so the default floor never goes below 5e-8, whatever the tolerance (N > 20: tolerance 5e-7, floor 5e-8)
    if (!(h->cfg.mu_min > 0)) h->cfg.mu_min = std::max(0.05 * h->cfg.tolerance, 5e-8);
    if (!(h->cfg.gravity > 0)) h->cfg.gravity = 9.80665;
    h->B = batch;
    h->device = device;
#ifdef CMPC_PROFILE
    // developer knobs exist in the diagnostic build only (-DCMPC_PROFILE), read once, here: the shipped library reads no environment variable
    if (const char* e = std::getenv("CMPC_WARM_DUALS")) h->warm_duals = std::atoi(e);
    if (const char* e = std::getenv("CMPC_MU_WARM")) { h->mu_warm = std::atof(e); h->floor_warm = std::min(1e-2, h->mu_warm); }
    h->force_warm = std::getenv("CMPC_FORCE_WARM") != nullptr;
    if (const char* e = std::getenv("CMPC_MU_ADAPT")) h->mu_adapt = (float)std::atof(e);
#endif
    if (h->cfg.tail_stages < 0 || h->cfg.tail_stages >= h->cfg.horizon) h->cfg.tail_stages = 0;
    if (h->cfg.tail_iterations < 0) h->cfg.tail_iterations = 0;
    if (!(h->cfg.tail_trigger > 0)) h->cfg.tail_trigger = 2e-5;
    if (!h->cfg.final_extrapolation) h->cfg.tail_stages = 0;   // (the polish hangs off the extrapolation step)
    cmpc_layout_init(h->L, cfg->horizon);
    h->lds = cmpc_solver_lds_bytes(cfg->horizon, 0);
    // factors in HBM scratch when the LDS image would not fit -- or, by choice, to halve the image so that
    // two workgroups share a CU (CMPC_FACTORS=hbm|lds overrides; default: LDS whenever it fits)
    bool fg = h->lds > 160 * 1024;
    {
        // more problems than CUs: the batch is throughput-bound, and two 67 KB workgroups per CU overlap
        // each other's single-wave phases (measured 1.42x at B = 4096); at B <= #CU latency rules: LDS
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && batch > prop.multiProcessorCount) fg = true;
    }
    if (h->cfg.factor_storage == CMPC_FACTORS_HBM) fg = true;
    if (h->cfg.factor_storage == CMPC_FACTORS_LDS && h->lds <= 160 * 1024) fg = false;
    if (fg) {
        h->lds = cmpc_solver_lds_bytes(cfg->horizon, 1);
        h->scratch_stride = ((long long)CMPC_REC_N + 2 * CMPC_NI) * cfg->horizon;   // factor records | slacks | multipliers
    }
    if (h->lds > 160 * 1024) {
        const int n = cfg->horizon;
        delete h;
        return fail(nullptr, CMPC_ERR_ARG, "cmpc_create: horizon " + std::to_string(n) + " needs more than 160 KiB of LDS per problem");
    }
    // a failure below releases whatever the handle already owns (cmpc_destroy copes with a partly built handle)
#define HIPCHK_CREATE(call)                                                                                   \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            const std::string m_ = std::string("cmpc_create: " #call ": ") + hipGetErrorString(e_);           \
            (void)hipGetLastError(); /* the runtime keeps the error until it is read: a later launch must not see it */ \
            cmpc_destroy(h);                                                                                  \
            return fail(nullptr, CMPC_ERR_HIP, m_);                                                           \
        }                                                                                                     \
    } while (0)
    HIPCHK_CREATE(hipSetDevice(device));
    HIPCHK_CREATE((hipError_t)cmpc_prepare_solver(cfg->horizon, fg ? 1 : 0, h->lds));   // (the kernel variant's dynamic-LDS limit: once per handle, not per launch)
    if (fg) {
        // the factor records rely on never-written zero blocks (layout: cmpc_solver.hip): zero once
        HIPCHK_CREATE(hipMalloc(&h->dScratch, sizeof(float) * (size_t)h->scratch_stride * (size_t)batch));
        HIPCHK_CREATE(hipMemset(h->dScratch, 0, sizeof(float) * (size_t)h->scratch_stride * (size_t)batch));
    }
    HIPCHK_CREATE(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPCHK_CREATE(hipEventCreate(&h->ev0));
    HIPCHK_CREATE(hipEventCreate(&h->ev1));
    HIPCHK_CREATE(hipMalloc(&h->dInfo, sizeof(float) * CMPC_INFO_N * (size_t)batch));
    HIPCHK_CREATE(hipMalloc(&h->dConsts, sizeof(CmpcConsts)));
    HIPCHK_CREATE(hipMalloc(&h->dBox, sizeof(float) * 12));
    if (h->warm_duals) {
        const size_t nd = (size_t)batch * (CMPC_NS * (cfg->horizon + 1) + 2 * CMPC_NI * cfg->horizon);
        HIPCHK_CREATE(hipMalloc(&h->dDuals, sizeof(float) * nd));
        HIPCHK_CREATE(hipMemset(h->dDuals, 0, sizeof(float) * nd));
    }
    {
        CmpcConsts q;
        fill_consts(h, q);
        HIPCHK_CREATE(hipMemcpy(h->dConsts, &q, sizeof(q), hipMemcpyHostToDevice));
    }
    // the memset and the copy above ran on the null stream; solves run on a non-blocking stream that does not
    // wait for it (a first solve racing the tail of a 350 MB memset was observed to fail): drain the device once
    HIPCHK_CREATE(hipDeviceSynchronize());
#undef HIPCHK_CREATE
    *out = h;
    return CMPC_OK;
}

int cmpc_destroy(cmpc_handle h)
{
    if (!h) return CMPC_OK;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->hXpin) hipHostFree(h->hXpin);
    if (h->hInfoPin) hipHostFree(h->hInfoPin);
    hipFree(h->dP); hipFree(h->dX0); hipFree(h->dX); hipFree(h->dInfo); hipFree(h->dConsts); hipFree(h->dScratch); hipFree(h->dBox); hipFree(h->dDuals);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return CMPC_OK;
}

static void fill_consts(cmpc_handle h, CmpcConsts& q)
{
    const cmpc_config& c = h->cfg;
    std::memset(&q, 0, sizeof(q));
    q.N = c.horizon; q.max_iter = c.max_iterations;
    q.exact_hessian = c.exact_hessian; q.final_extrap = c.final_extrapolation;
    q.tail_stages = c.tail_stages; q.tail_iters = c.tail_iterations; q.tail_trigger = (float)c.tail_trigger;
    q.dt = (float)c.sampling_time; q.mu_fr = (float)c.friction_coefficient; q.grav = (float)c.gravity;
    q.w_com0 = (float)c.com_weight[0]; q.w_com1 = (float)c.com_weight[1];
    q.w_h = (float)c.angular_momentum_weight; q.w_pos = (float)c.contact_position_weight;
    q.w_sym = (float)c.contact_force_symmetry_weight;
    for (int i = 0; i < 3; ++i) q.D[i] = (float)(2.0 * c.force_rate_of_change_weight[i]);
    for (int ct = 0; ct < 2; ++ct)
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 3; ++i) q.corners[12 * ct + 3 * j + i] = (float)c.corners[ct][j][i];
    for (int k = 0; k <= c.horizon; ++k) {
        const double wz = 0.5 * c.com_weight[2] * (1.0 + std::exp(-(double)k));
        q.wz2[k] = (float)(2.0 * wz * wz);
    }
    q.tol = (float)c.tolerance; q.step_tol = (float)c.step_tolerance; q.mu_init = (float)c.mu_init; q.mu_min = (float)c.mu_min;
    // Levenberg shift: 5e-5 of the smallest cost curvature (2 w_rate).  It does not move the fixed
    // point (the right-hand side is exact); it keeps the stage Hessians factorisable in float32 along
    // directions the cost does not see (measured on MI355X: 1e-5..1e-2 all converge in the same
    // number of iterations, 1e-1 doubles it)
    {
        double dmin = 2.0 * c.force_rate_of_change_weight[0];
        for (int i = 1; i < 3; ++i) dmin = std::min(dmin, 2.0 * c.force_rate_of_change_weight[i]);
        q.reg = (float)std::max(1e-5, 5e-5 * dmin);
#ifdef CMPC_PROFILE
        if (const char* e = std::getenv("CMPC_REG")) q.reg = (float)std::atof(e);
#endif
    }
    // Mehrotra's sigma = (mu_aff/mu)^3 can ask for a 1000-fold barrier decrease in one step; the linearisation
    // does not hold that far and the blocked step costs the problem 3-6 extra iterations.  A floor of 0.03
    // costs +0.3 iterations on the mean of config 2 and removes the tail (max 11 -> 8 of 4096 problems; config 3:
    // mean 8.60 -> 8.49, max 14 -> 12), and a batch is as slow as its slowest problem.
    q.sigma_min = 0.03f;
#ifdef CMPC_PROFILE
    if (const char* e = std::getenv("CMPC_SIGMA_MIN")) q.sigma_min = (float)std::atof(e);
    if (const char* e = std::getenv("CMPC_HWID_PROBE")) q.hwid_probe = std::atoi(e);
#endif
}

static void fill_params(cmpc_handle h, CmpcParams& p)
{
    std::memset(&p, 0, sizeof(p));
    p.kc = h->dConsts; p.N = h->cfg.horizon; p.B = h->B;
    p.scratch = h->dScratch; p.scratch_stride = h->scratch_stride;
    // cold start: a fixed initial barrier parameter if the configuration names one, else per problem
    // mu0 = clamp(3.5 ep0^2, 0.03, 0.5) from the initial primal infeasibility ep0 (measured on 4096-problem batches:
    // config 2 wants ~0.03, config 3 ~0.3; the rule cuts the slowest problem of a 256-batch by ~0.7 iterations)
    p.mu_init = h->cfg.mu_init > 0 ? (float)h->cfg.mu_init : 0.1f;
    p.mu_adapt = h->cfg.mu_init > 0 ? 0.f : h->mu_adapt;
    p.t_floor = 1e-2f;
    p.warm_budget = h->warm_budget; p.warm_no_restart = h->warm_no_restart;
    p.duals = h->dDuals; p.warm_duals = h->warm_duals;
}

static int solve_device_impl(cmpc_handle h, const float* dP, const float* dX0, float* dX, float* dInfo, void* stream, bool warm)
{
    if (!h || !dP || !dX0 || !dX) return fail(h, CMPC_ERR_ARG, "cmpc_solve_device: null argument");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    CmpcParams p;
    fill_params(h, p);
    p.P = dP; p.X0 = dX0; p.X = dX; p.info = dInfo ? dInfo : h->dInfo;
    if (warm || h->force_warm) {  // dX0 is a previous solution shifted by one knot: start near the central path
        p.mu_init = (float)h->mu_warm; p.mu_adapt = 0.f; p.t_floor = (float)h->floor_warm; p.warm = 1;
    }
    if (h->timing) HIPCHK(h, hipEventRecord(h->ev0, st));
    int rc = cmpc_launch_solver(&p, h->lds, st);
    if (rc != 0) return fail(h, CMPC_ERR_HIP, std::string("solver launch: ") + hipGetErrorString((hipError_t)rc));
    if (h->timing) HIPCHK(h, hipEventRecord(h->ev1, st));
    h->timed = h->timing;
    return CMPC_OK;
}

int cmpc_solve_device(cmpc_handle h, const float* dP, const float* dX0, float* dX, float* dInfo, void* stream)
{
    return solve_device_impl(h, dP, dX0, dX, dInfo, stream, false);
}

int cmpc_set_warm_policy(cmpc_handle h, int warm_budget, int restart_in_kernel)
{
    if (!h || warm_budget < 0) return fail(h, CMPC_ERR_ARG, "cmpc_set_warm_policy: bad argument");
    h->warm_budget = warm_budget;
    h->warm_no_restart = restart_in_kernel ? 0 : 1;
    return CMPC_OK;
}

int cmpc_solve_device_warm(cmpc_handle h, const float* dP, const float* dX0, float* dX, float* dInfo, void* stream)
{
    return solve_device_impl(h, dP, dX0, dX, dInfo, stream, true);
}

namespace {
__global__ __launch_bounds__(256) void poison_lds_kernel(int words, unsigned* sink)
{
    extern __shared__ unsigned pl[];
    for (int e = threadIdx.x; e < words; e += 256) pl[e] = 0x7fc00000u | (unsigned)(e & 0xffff);
    __syncthreads();
    if (sink && threadIdx.x == 0 && pl[(blockIdx.x * 97) % words] == 1u) *sink = 1u;  // keep the stores alive
}
}  // namespace

int cmpc_test_poison_lds(cmpc_handle h)
{
    if (!h) return CMPC_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    const int bytes = 160 * 1024;
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    // one workgroup per CU at a time (160 KiB each); several waves of them so that every CU is visited
    hipLaunchKernelGGL(poison_lds_kernel, dim3(2048), dim3(256), bytes, h->stream, bytes / 4, reinterpret_cast<unsigned*>(h->dInfo));
    HIPCHK(h, hipGetLastError());
    return CMPC_OK;
}

int cmpc_set_timing(cmpc_handle h, int enabled)
{
    if (!h) return CMPC_ERR_ARG;
    h->timing = enabled != 0;
    if (!h->timing) h->timed = false;
    return CMPC_OK;
}

float cmpc_last_solve_ms(cmpc_handle h)
{
    if (!h || !h->timed) return -1.f;
    float ms = -1.f;
    if (hipEventSynchronize(h->ev1) != hipSuccess) return -1.f;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) != hipSuccess) return -1.f;
    return ms;
}

static int ensure_buffers(cmpc_handle h)
{
    const size_t nP = (size_t)h->B * h->L.np, nX = (size_t)h->B * h->L.nx;
    if (!h->dP) HIPCHK(h, hipMalloc(&h->dP, sizeof(float) * nP));
    if (!h->dX0) HIPCHK(h, hipMalloc(&h->dX0, sizeof(float) * nX));
    if (!h->dX) HIPCHK(h, hipMalloc(&h->dX, sizeof(float) * nX));
    if (h->hP.empty()) h->hP.assign(nP, 0.f);
    return CMPC_OK;
}

static int check_status(cmpc_handle h, const std::vector<float>& info)
{
    int bad = 0, first = -1;
    for (int b = 0; b < h->B; ++b)
        if (info[(size_t)b * CMPC_INFO_N + 5] != 0.f) { if (first < 0) first = b; ++bad; }
    if (bad) {
        char buf[160];
        std::snprintf(buf, sizeof(buf), "%d of %d problems did not converge (first: %d, status %d, kkt %.3g)", bad, h->B, first,
                      (int)info[(size_t)first * CMPC_INFO_N + 5], info[(size_t)first * CMPC_INFO_N + 1]);
        return fail(h, CMPC_ERR_NOT_CONVERGED, buf);
    }
    return CMPC_OK;
}

int cmpc_solve(cmpc_handle h, const float* P, const float* X0, float* X, float* info)
{
    if (!h || !P || !X0 || !X) return fail(h, CMPC_ERR_ARG, "cmpc_solve: null argument");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = ensure_buffers(h);
    if (rc) return rc;
    const size_t nP = (size_t)h->B * h->L.np, nX = (size_t)h->B * h->L.nx;
    HIPCHK(h, hipMemcpyAsync(h->dP, P, sizeof(float) * nP, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->dX0, X0, sizeof(float) * nX, hipMemcpyHostToDevice, h->stream));
    h->hX_valid = false;   // (dX is about to change behind the pinned mirror of cmpc_advance)
    rc = cmpc_solve_device(h, h->dP, h->dX0, h->dX, h->dInfo, nullptr);
    if (rc) return rc;
    std::vector<float> hinfo((size_t)h->B * CMPC_INFO_N);
    HIPCHK(h, hipMemcpyAsync(X, h->dX, sizeof(float) * nX, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(hinfo.data(), h->dInfo, sizeof(float) * hinfo.size(), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (info) std::memcpy(info, hinfo.data(), sizeof(float) * hinfo.size());
    h->have_solution = true;
    return check_status(h, hinfo);
}

// ---------------- class-shaped setters ----------------
int cmpc_set_state(cmpc_handle h, const float* state, const float* wrench)
{
    if (!h || !state) return fail(h, CMPC_ERR_ARG, "cmpc_set_state: null argument");
    int rc = ensure_buffers(h);
    if (rc) return rc;
    const int N = h->cfg.horizon;
    for (int b = 0; b < h->B; ++b) {
        float* p = h->hP.data() + (size_t)b * h->L.np;
        std::memcpy(p + h->L.p_com0, state + 9 * (size_t)b, sizeof(float) * 9);
        for (int k = 0; k < N; ++k)
            for (int i = 0; i < 3; ++i) {
                p[h->L.p_fext + 3 * k + i] = wrench ? wrench[((size_t)b * N + k) * 6 + i] : 0.f;
                p[h->L.p_text + 3 * k + i] = wrench ? wrench[((size_t)b * N + k) * 6 + 3 + i] : 0.f;
            }
    }
    return CMPC_OK;
}

int cmpc_set_reference(cmpc_handle h, const float* com_ref, const float* h_ref)
{
    if (!h || !com_ref || !h_ref) return fail(h, CMPC_ERR_ARG, "cmpc_set_reference: null argument");
    int rc = ensure_buffers(h);
    if (rc) return rc;
    const size_t n = 3 * (size_t)(h->cfg.horizon + 1);
    for (int b = 0; b < h->B; ++b) {
        float* p = h->hP.data() + (size_t)b * h->L.np;
        std::memcpy(p + h->L.p_comref, com_ref + n * b, sizeof(float) * n);
        std::memcpy(p + h->L.p_href, h_ref + n * b, sizeof(float) * n);
    }
    return CMPC_OK;
}

int cmpc_set_contacts(cmpc_handle h, const float* R, const float* upper, const float* lower, const float* enabled,
                      const float* nominal, const float* current)
{
    if (!h || !R || !upper || !lower || !enabled || !nominal || !current)
        return fail(h, CMPC_ERR_ARG, "cmpc_set_contacts: null argument");
    int rc = ensure_buffers(h);
    if (rc) return rc;
    const int N = h->cfg.horizon;
    for (int b = 0; b < h->B; ++b) {
        float* p = h->hP.data() + (size_t)b * h->L.np;
        for (int ct = 0; ct < 2; ++ct) {
            const size_t bc = (size_t)b * 2 + ct;
            for (int k = 0; k < N; ++k) {
                const float* Rk = R + (bc * N + k) * 9;  // row-major
                for (int r = 0; r < 3; ++r)
                    for (int cc = 0; cc < 3; ++cc) p[h->L.p_R[ct] + 9 * k + 3 * cc + r] = Rk[3 * r + cc];
                const float e = enabled[bc * N + k];
                if (e != 0.f && e != 1.f) return fail(h, CMPC_ERR_ARG, "cmpc_set_contacts: enabled must be 0 or 1");
                p[h->L.p_gam[ct] + k] = e;
                for (int i = 0; i < 3; ++i) {
                    p[h->L.p_up[ct] + 3 * k + i] = upper[(bc * N + k) * 3 + i];
                    p[h->L.p_lo[ct] + 3 * k + i] = lower[(bc * N + k) * 3 + i];
                    if (p[h->L.p_up[ct] + 3 * k + i] < p[h->L.p_lo[ct] + 3 * k + i])
                        return fail(h, CMPC_ERR_ARG, "cmpc_set_contacts: bounding box upper < lower");
                }
            }
            std::memcpy(p + h->L.p_nom[ct], nominal + bc * 3 * (N + 1), sizeof(float) * 3 * (N + 1));
            std::memcpy(p + h->L.p_cur[ct], current + bc * 3, sizeof(float) * 3);
        }
    }
    return CMPC_OK;
}

static void cold_start(cmpc_handle h)
{
    const int N = h->cfg.horizon;
    h->hX0.assign((size_t)h->B * h->L.nx, 0.f);
    for (int b = 0; b < h->B; ++b) {
        const float* p = h->hP.data() + (size_t)b * h->L.np;
        float* x = h->hX0.data() + (size_t)b * h->L.nx;
        for (int k = 0; k <= N; ++k)
            for (int i = 0; i < 3; ++i) x[h->L.o_com + 3 * k + i] = p[h->L.p_com0 + i];
        for (int ct = 0; ct < 2; ++ct) {
            std::memcpy(x + h->L.o_pos[ct], p + h->L.p_nom[ct], sizeof(float) * 3 * (N + 1));
            for (int j = 0; j < 4; ++j)
                for (int k = 0; k < N; ++k) x[h->L.o_f[ct][j] + 3 * k + 2] = (float)(h->cfg.gravity / 8.0);
        }
    }
}

int cmpc_set_initial_guess(cmpc_handle h, const float* x0, int shift_previous)
{
    if (!h) return fail(h, CMPC_ERR_ARG, "cmpc_set_initial_guess: null handle");
    int rc = ensure_buffers(h);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t nX = (size_t)h->B * h->L.nx;
    h->warm = false;   // only a shifted previous solution starts the barrier at mu_warm
    if (x0) {
        HIPCHK(h, hipMemcpyAsync(h->dX0, x0, sizeof(float) * nX, hipMemcpyHostToDevice, h->stream));
    } else if (shift_previous && h->have_solution) {
        CmpcParams p;
        fill_params(h, p);
        int r = cmpc_launch_warm_shift(&p, h->dX, h->dX0, h->stream);
        if (r != 0) return fail(h, CMPC_ERR_HIP, "warm-start shift launch failed");
        h->warm = true;
    } else {
        cold_start(h);
        HIPCHK(h, hipMemcpyAsync(h->dX0, h->hX0.data(), sizeof(float) * nX, hipMemcpyHostToDevice, h->stream));
    }
    h->x0_set = true;
    return CMPC_OK;
}

int cmpc_advance(cmpc_handle h)
{
    if (!h) return fail(h, CMPC_ERR_ARG, "cmpc_advance: null handle");
    int rc = ensure_buffers(h);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->x0_set) { rc = cmpc_set_initial_guess(h, nullptr, 0); if (rc) return rc; }
    HIPCHK(h, hipMemcpyAsync(h->dP, h->hP.data(), sizeof(float) * h->hP.size(), hipMemcpyHostToDevice, h->stream));
    {
        // a shifted previous solution starts close to the optimum: start the barrier at mu_warm (not 0.1) with a
        // smaller slack floor, i.e. near the central path where the previous solve passed through
        CmpcParams p;
        fill_params(h, p);
        p.P = h->dP; p.X0 = h->dX0; p.X = h->dX; p.info = h->dInfo;
        if (h->warm) { p.mu_init = (float)h->mu_warm; p.mu_adapt = 0.f; p.t_floor = (float)h->floor_warm; p.warm = 1; }
        if (h->timing) HIPCHK(h, hipEventRecord(h->ev0, h->stream));
        int lrc = cmpc_launch_solver(&p, h->lds, h->stream);
        if (lrc != 0) return fail(h, CMPC_ERR_HIP, std::string("solver launch: ") + hipGetErrorString((hipError_t)lrc));
        if (h->timing) HIPCHK(h, hipEventRecord(h->ev1, h->stream));
        h->timed = h->timing;
        h->warm = false;
    }
    const size_t nXo = (size_t)h->B * h->L.nx, nIo = (size_t)h->B * CMPC_INFO_N;
    h->hX_valid = false;
    if (!h->hXpin) HIPCHK(h, hipHostMalloc((void**)&h->hXpin, sizeof(float) * nXo, hipHostMallocDefault));
    if (!h->hInfoPin) HIPCHK(h, hipHostMalloc((void**)&h->hInfoPin, sizeof(float) * nIo, hipHostMallocDefault));
    HIPCHK(h, hipMemcpyAsync(h->hXpin, h->dX, sizeof(float) * nXo, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->hInfoPin, h->dInfo, sizeof(float) * nIo, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_solution = true;
    h->hX_valid = true;
    h->x0_set = false;  // the next tick picks its own start unless told otherwise
    const std::vector<float> hinfo(h->hInfoPin, h->hInfoPin + nIo);
    return check_status(h, hinfo);
}

int cmpc_get_parameters(cmpc_handle h, float* P)
{
    if (!h || !P) return fail(h, CMPC_ERR_ARG, "cmpc_get_parameters: null argument");
    int rc = ensure_buffers(h);
    if (rc) return rc;
    std::memcpy(P, h->hP.data(), sizeof(float) * h->hP.size());
    return CMPC_OK;
}

int cmpc_get_parameters_device(cmpc_handle h, const float** dP)
{
    if (!h || !dP) return fail(h, CMPC_ERR_ARG, "cmpc_get_parameters_device: null argument");
    int rc = ensure_buffers(h);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(h->dP, h->hP.data(), sizeof(float) * h->hP.size(), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *dP = h->dP;
    return CMPC_OK;
}

int cmpc_get_solution(cmpc_handle h, float* X, float* info)
{
    if (!h || !h->have_solution) return fail(h, CMPC_ERR_ARG, "cmpc_get_solution: no solution yet");
    if (h->hX_valid) {   // (the last writer of dX was cmpc_advance: its pinned mirror is the solution)
        if (X) std::memcpy(X, h->hXpin, sizeof(float) * (size_t)h->B * h->L.nx);
        if (info) std::memcpy(info, h->hInfoPin, sizeof(float) * (size_t)h->B * CMPC_INFO_N);
        return CMPC_OK;
    }
    HIPCHK(h, hipSetDevice(h->device));
    if (X) HIPCHK(h, hipMemcpyAsync(X, h->dX, sizeof(float) * (size_t)h->B * h->L.nx, hipMemcpyDeviceToHost, h->stream));
    if (info) HIPCHK(h, hipMemcpyAsync(info, h->dInfo, sizeof(float) * (size_t)h->B * CMPC_INFO_N, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CMPC_OK;
}

int cmpc_get_output(cmpc_handle h, float* forces0, float* pos0, float* next_pos, int* next_knot)
{
    if (!h || !h->have_solution) return fail(h, CMPC_ERR_ARG, "cmpc_get_output: no solution yet");
    std::vector<float> Xcopy;
    const float* Xh = h->hXpin;
    if (!h->hX_valid) {
        Xcopy.resize((size_t)h->B * h->L.nx);
        int rc = cmpc_get_solution(h, Xcopy.data(), nullptr);
        if (rc) return rc;
        Xh = Xcopy.data();
    }
    const int N = h->cfg.horizon;
    for (int b = 0; b < h->B; ++b) {
        const float* x = Xh + (size_t)b * h->L.nx;
        const float* p = h->hP.data() + (size_t)b * h->L.np;
        for (int ct = 0; ct < 2; ++ct) {
            for (int j = 0; j < 4; ++j)
                for (int i = 0; i < 3; ++i)
                    if (forces0) forces0[(((size_t)b * 2 + ct) * 4 + j) * 3 + i] = x[h->L.o_f[ct][j] + i];
            int land = -1;  // first knot after a swing stage whose successor is in contact (or the horizon end)
            for (int k = 0; k < N; ++k)
                if (p[h->L.p_gam[ct] + k] < 0.5f && (k + 1 == N || p[h->L.p_gam[ct] + k + 1] > 0.5f)) { land = k + 1; break; }
            for (int i = 0; i < 3; ++i) {
                if (pos0) pos0[((size_t)b * 2 + ct) * 3 + i] = x[h->L.o_pos[ct] + i];
                if (next_pos) next_pos[((size_t)b * 2 + ct) * 3 + i] = land >= 0 ? x[h->L.o_pos[ct] + 3 * land + i] : x[h->L.o_pos[ct] + i];
            }
            if (next_knot) next_knot[(size_t)b * 2 + ct] = land;
        }
    }
    return CMPC_OK;
}

int cmpc_eval_nlp_device(cmpc_handle h, const float* dX, const float* dP, const float* dLamG, float lam_f, float* dF,
                         float* dG, float* dGradF, float* dJac, float* dHess, void* stream)
{
    if (!h || !dX || !dP) return fail(h, CMPC_ERR_ARG, "cmpc_eval_nlp_device: null argument");
    if (dHess && !dLamG) return fail(h, CMPC_ERR_ARG, "cmpc_eval_nlp_device: the Hessian needs lam_g");
    HIPCHK(h, hipSetDevice(h->device));
    CmpcParams p;
    fill_params(h, p);
    int rc = cmpc_launch_nlp_eval(&p, dX, dP, dLamG, lam_f, dF, dG, dGradF, dJac, dHess, stream ? (hipStream_t)stream : h->stream);
    if (rc != 0) return fail(h, CMPC_ERR_HIP, std::string("nlp eval launch: ") + hipGetErrorString((hipError_t)rc));
    return CMPC_OK;
}

int cmpc_eval_nlp_grad_device(cmpc_handle h, const float* dX, const float* dP, const float* dLamG, float lam_f, float* dGradX, float* dGradP,
                              void* stream)
{
    if (!h || !dX || !dP || !dLamG) return fail(h, CMPC_ERR_ARG, "cmpc_eval_nlp_grad_device: null argument");
    HIPCHK(h, hipSetDevice(h->device));
    CmpcParams p;
    fill_params(h, p);
    int rc = cmpc_launch_nlp_grad(&p, dX, dP, dLamG, lam_f, dGradX, dGradP, stream ? (hipStream_t)stream : h->stream);
    if (rc != 0) return fail(h, CMPC_ERR_HIP, std::string("nlp grad launch: ") + hipGetErrorString((hipError_t)rc));
    return CMPC_OK;
}

// ---- 8f-3: planner references -> MPC knots (CentroidalMPCBlock.cpp:525-577): angular momentum / mass, CoM height
// override, linear spline from the planner's knots (period in_dt, first knot t_offset in the past) to the N+1 MPC
// knots; clamped at both ends.  Host-side like the reference's LinearSpline; fills the handle's parameter staging. ----
int cmpc_set_reference_from_planner(cmpc_handle h, const float* com_in, const float* h_in, int n_in, double in_dt, double t_offset,
                                    double robot_mass, double com_height)
{
    if (!h || !com_in || !h_in || n_in < 2 || !(in_dt > 0) || !(robot_mass > 0))
        return fail(h, CMPC_ERR_ARG, "cmpc_set_reference_from_planner: bad argument");
    int rc = ensure_buffers(h);
    if (rc) return rc;
    const int N = h->cfg.horizon;
    for (int b = 0; b < h->B; ++b) {
        float* p = h->hP.data() + (size_t)b * h->L.np;
        const float* ci = com_in + (size_t)b * n_in * 3;
        const float* hi = h_in + (size_t)b * n_in * 3;
        for (int k = 0; k <= N; ++k)
            cmpc_resample_reference_knot(ci, hi, n_in, in_dt, t_offset, h->cfg.sampling_time, k, robot_mass, com_height, p + h->L.p_comref + 3 * k,
                                         p + h->L.p_href + 3 * k);
    }
    return CMPC_OK;
}

// the same on the device, into the caller's dP (one thread per problem and knot)
int cmpc_write_reference_from_planner_device(cmpc_handle h, const float* dComIn, const float* dHIn, int n_in, double in_dt, double t_offset, double robot_mass,
                                             double com_height, float* dP, void* stream)
{
    if (!h || !dComIn || !dHIn || !dP || n_in < 2 || !(in_dt > 0) || !(robot_mass > 0))
        return fail(h, CMPC_ERR_ARG, "cmpc_write_reference_from_planner_device: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = cmpc_launch_reference_from_planner(h->B, h->cfg.horizon, n_in, h->cfg.sampling_time, in_dt, t_offset, robot_mass, com_height, dComIn, dHIn, dP,
                                                stream ? (hipStream_t)stream : h->stream);
    if (rc != 0) return fail(h, CMPC_ERR_HIP, std::string("reference resampling launch: ") + hipGetErrorString((hipError_t)rc));
    return CMPC_OK;
}

// ---- 8f-4: plant step on the device (see cmpc_plant_step_kernel) ----
int cmpc_plant_step_device(cmpc_handle h, const float* dX, const float* dP, const float* dStateIn, float* dStateOut, float* dZmp,
                           double step, int substeps, double zmp_half_x, double zmp_half_y, void* stream)
{
    if (!h || !dX || !dP || !dStateIn || !dStateOut || !(step > 0) || substeps < 1)
        return fail(h, CMPC_ERR_ARG, "cmpc_plant_step_device: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    const float* corners = h->dConsts->corners;  // device pointer arithmetic only
    int rc = cmpc_launch_plant_step(h->cfg.horizon, h->B, (float)h->cfg.gravity, corners, dX, dP, dStateIn, dStateOut, dZmp, (float)step,
                                    substeps, (float)zmp_half_x, (float)zmp_half_y, stream ? (hipStream_t)stream : h->stream);
    if (rc != 0) return fail(h, CMPC_ERR_HIP, std::string("plant step launch: ") + hipGetErrorString((hipError_t)rc));
    return CMPC_OK;
}

// ---- 8e: compact per-problem output for the all-gather (see cmpc_compact_kernel) ----
int cmpc_compact_output_device(cmpc_handle h, const float* dX, const float* dInfo, float* dOut, void* stream)
{
    if (!h || !dX || !dInfo || !dOut) return fail(h, CMPC_ERR_ARG, "cmpc_compact_output_device: null pointer");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = cmpc_launch_compact(h->cfg.horizon, h->B, dX, dInfo, dOut, stream ? (hipStream_t)stream : h->stream);
    if (rc != 0) return fail(h, CMPC_ERR_HIP, std::string("compact output launch: ") + hipGetErrorString((hipError_t)rc));
    return CMPC_OK;
}

// ---- 8f-1: contact schedules, batched (logic: cmpc_contacts.h) ----
// ---- 8e: the gather of compact solutions across the GPUs of a node, through RCCL (ncclAllGather over xGMI).  librccl is opened on first use, so that the
// library itself carries no link-time dependency on it: a single-GPU caller never loads it. ----
namespace {
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int /*ncclDataType_t*/, void* /*ncclComm_t*/, hipStream_t);
nccl_allgather_fn load_allgather(std::string& why)
{
    static nccl_allgather_fn fn = nullptr;
    static bool tried = false;
    if (tried) { if (!fn) why = "librccl.so: ncclAllGather not available"; return fn; }
    tried = true;
    void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { why = std::string("dlopen(librccl.so): ") + dlerror(); return nullptr; }
    fn = reinterpret_cast<nccl_allgather_fn>(dlsym(lib, "ncclAllGather"));
    if (!fn) why = "librccl.so has no ncclAllGather";
    return fn;
}
}  // namespace

int cmpc_allgather_compact_device(cmpc_handle h, void* nccl_comm, int world_size, const float* dLocal, float* dAll, void* stream)
{
    if (!h || !nccl_comm || world_size < 1 || !dLocal || !dAll) return fail(h, CMPC_ERR_ARG, "cmpc_allgather_compact_device: bad argument");
    std::string why;
    nccl_allgather_fn ag = load_allgather(why);
    if (!ag) return fail(h, CMPC_ERR_HIP, "cmpc_allgather_compact_device: " + why);
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    const size_t count = (size_t)h->B * (size_t)(3 * (h->cfg.horizon + 1) + 38);   // floats of this rank's compact records (cmpc_compact_output_device)
    const int rc = ag(dLocal, dAll, count, 7 /* ncclFloat32 */, nccl_comm, st);
    if (rc != 0) return fail(h, CMPC_ERR_HIP, "ncclAllGather failed with code " + std::to_string(rc));
    return CMPC_OK;
}

int cmpc_contacts_merge(int batch, int max_contacts, double now, const double* plan_t, const float* plan_pose, const int* plan_n,
                        const double* mpc_t, const float* mpc_pose, const int* mpc_n, double* out_t, float* out_pose, int* out_n, int* ok)
{
    if (batch < 1 || max_contacts < 1 || !plan_t || !plan_pose || !plan_n || !mpc_t || !mpc_pose || !mpc_n || !out_t || !out_pose || !out_n)
        return fail(nullptr, CMPC_ERR_ARG, "cmpc_contacts_merge: bad argument");
    const int M = max_contacts;
    int all = CMPC_OK;
    for (int b = 0; b < batch; ++b) {
        bool good = true;
        for (int c = 0; c < 2; ++c) {
            const size_t e = (size_t)b * 2 + c, o = e * M;
            if (plan_n[e] < 0 || plan_n[e] > M || mpc_n[e] < 0 || mpc_n[e] > M) return fail(nullptr, CMPC_ERR_ARG, "cmpc_contacts_merge: list length out of range");
            good = cmpc_merge_foot(now, plan_t + 2 * o, plan_pose + 7 * o, plan_n[e], mpc_t + 2 * o, mpc_pose + 7 * o, mpc_n[e], M,
                                   out_t + 2 * o, out_pose + 7 * o, out_n + e) && good;
        }
        if (ok) ok[b] = good ? 1 : 0;
        if (!good) all = CMPC_ERR_ARG;
    }
    if (all != CMPC_OK) return fail(nullptr, CMPC_ERR_ARG, "cmpc_contacts_merge: the planner has no active contact where the MPC list has one");
    return CMPC_OK;
}

int cmpc_contacts_sample(int horizon, double dt, int batch, int max_contacts, double now, const double* t, const float* pose, const int* n,
                         const float* box_upper, const float* box_lower, float* P, int* land)
{
    if (horizon < 1 || !(dt > 0) || batch < 1 || max_contacts < 1 || !t || !pose || !n || !box_upper || !box_lower || !P)
        return fail(nullptr, CMPC_ERR_ARG, "cmpc_contacts_sample: bad argument");
    const CmpcIdx L{horizon};
    for (int b = 0; b < batch; ++b)
        for (int c = 0; c < 2; ++c) {
            const size_t e = (size_t)b * 2 + c, o = e * max_contacts;
            if (n[e] < 1 || n[e] > max_contacts) return fail(nullptr, CMPC_ERR_ARG, "cmpc_contacts_sample: every foot needs 1..max_contacts contacts");
            const int lk = cmpc_sample_foot(horizon, dt, now, c, t + 2 * o, pose + 7 * o, n[e], box_upper, box_lower, P + (size_t)b * L.np());
            if (land) land[e] = lk;
        }
    return CMPC_OK;
}

int cmpc_contacts_adjust(int horizon, int batch, int max_contacts, double now, const float* X, const int* land, const double* t, float* pose, const int* n)
{
    if (horizon < 1 || batch < 1 || max_contacts < 1 || !X || !land || !t || !pose || !n) return fail(nullptr, CMPC_ERR_ARG, "cmpc_contacts_adjust: bad argument");
    const CmpcIdx L{horizon};
    for (int b = 0; b < batch; ++b)
        for (int c = 0; c < 2; ++c) {
            const size_t e = (size_t)b * 2 + c, o = e * max_contacts;
            if (land[e] < 0 || land[e] > horizon) continue;
            if (n[e] < 1 || n[e] > max_contacts) return fail(nullptr, CMPC_ERR_ARG, "cmpc_contacts_adjust: every foot needs 1..max_contacts contacts");
            const int nx = cmpc_next_contact(t + 2 * o, n[e], now);
            if (nx < 0) continue;
            for (int i = 0; i < 3; ++i) pose[7 * (o + nx) + i] = X[(size_t)b * L.nx() + L.oPos(c) + 3 * land[e] + i];
        }
    return CMPC_OK;
}

int cmpc_contacts_merge_device(cmpc_handle h, int max_contacts, double now, const double* dPlanT, const float* dPlanPose, const int* dPlanN,
                               const double* dMpcT, const float* dMpcPose, const int* dMpcN, double* dOutT, float* dOutPose, int* dOutN,
                               int* dOk, void* stream)
{
    if (!h || max_contacts < 1 || !dPlanT || !dPlanPose || !dPlanN || !dMpcT || !dMpcPose || !dMpcN || !dOutT || !dOutPose || !dOutN)
        return fail(h, CMPC_ERR_ARG, "cmpc_contacts_merge_device: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = cmpc_launch_contacts_merge(h->B, max_contacts, now, dPlanT, dPlanPose, dPlanN, dMpcT, dMpcPose, dMpcN, dOutT, dOutPose, dOutN, dOk,
                                        stream ? (hipStream_t)stream : h->stream);
    if (rc != 0) return fail(h, CMPC_ERR_HIP, std::string("contact merge launch: ") + hipGetErrorString((hipError_t)rc));
    return CMPC_OK;
}

// the bounding boxes of the two contacts on the device (uploaded when they change)
static int upload_box(cmpc_handle h, const float* box_upper, const float* box_lower, hipStream_t st)
{
    float box[12];
    std::memcpy(box, box_upper, sizeof(float) * 6);
    std::memcpy(box + 6, box_lower, sizeof(float) * 6);
    if (!h->box_set || std::memcmp(box, h->hBox, sizeof(box)) != 0) {
        std::memcpy(h->hBox, box, sizeof(box));
        HIPCHK(h, hipMemcpyAsync(h->dBox, h->hBox, sizeof(box), hipMemcpyHostToDevice, st));
        h->box_set = true;
    }
    return CMPC_OK;
}

int cmpc_contacts_sample_device(cmpc_handle h, int max_contacts, double now, const double* dT, const float* dPose, const int* dN,
                                const float* box_upper, const float* box_lower, float* dP, int* dLand, void* stream)
{
    if (!h || max_contacts < 1 || !dT || !dPose || !dN || !box_upper || !box_lower || !dP) return fail(h, CMPC_ERR_ARG, "cmpc_contacts_sample_device: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    int rc = upload_box(h, box_upper, box_lower, st);
    if (rc != CMPC_OK) return rc;
    rc = cmpc_launch_contacts_sample(h->B, h->cfg.horizon, max_contacts, h->cfg.sampling_time, now, dT, dPose, dN, h->dBox, dP, dLand, st);
    if (rc != 0) return fail(h, CMPC_ERR_HIP, std::string("contact sampling launch: ") + hipGetErrorString((hipError_t)rc));
    return CMPC_OK;
}

int cmpc_contacts_adjust_device(cmpc_handle h, int max_contacts, double now, const float* dX, const int* dLand, const double* dT, float* dPose,
                                const int* dN, void* stream)
{
    if (!h || max_contacts < 1 || !dX || !dLand || !dT || !dPose || !dN) return fail(h, CMPC_ERR_ARG, "cmpc_contacts_adjust_device: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = cmpc_launch_contacts_adjust(h->B, h->cfg.horizon, max_contacts, now, dX, dLand, dT, dPose, dN, stream ? (hipStream_t)stream : h->stream);
    if (rc != 0) return fail(h, CMPC_ERR_HIP, std::string("contact adjustment launch: ") + hipGetErrorString((hipError_t)rc));
    return CMPC_OK;
}

int cmpc_write_state_device(cmpc_handle h, const float* dState, const float* dWrench, float* dP, void* stream)
{
    if (!h || !dState || !dP) return fail(h, CMPC_ERR_ARG, "cmpc_write_state_device: null argument");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = cmpc_launch_write_state(h->B, h->cfg.horizon, dState, dWrench, dP, stream ? (hipStream_t)stream : h->stream);
    if (rc != 0) return fail(h, CMPC_ERR_HIP, std::string("state write launch: ") + hipGetErrorString((hipError_t)rc));
    return CMPC_OK;
}

int cmpc_shift_solution_device(cmpc_handle h, const float* dXprev, float* dX0, void* stream)
{
    if (!h || !dXprev || !dX0) return fail(h, CMPC_ERR_ARG, "cmpc_shift_solution_device: null argument");
    HIPCHK(h, hipSetDevice(h->device));
    CmpcParams p;
    fill_params(h, p);
    int rc = cmpc_launch_warm_shift(&p, dXprev, dX0, stream ? (hipStream_t)stream : h->stream);
    if (rc != 0) return fail(h, CMPC_ERR_HIP, "warm-start shift launch failed");
    return CMPC_OK;   // (no state on the handle: the caller solves from dX0 with cmpc_solve_device_warm)
}

// ---- 8f-1 .. 8f-4 chained: one tick of the receding-horizon loop in one call (include/cmpc.h): the steps of the entry points above in the reference's order, with the
// same argument checks, as THREE launches -- everything in front of the solve, the solve, everything behind it (cmpc_tick_pre_kernel / cmpc_tick_post_kernel call the same
// per-problem functions as the single kernels; results identical to the last bit). ----
int cmpc_rollout_tick_device(cmpc_handle h, int max_contacts, double now, int warm, const cmpc_tick_io* io, void* stream)
{
    if (!h || !io) return fail(h, CMPC_ERR_ARG, "cmpc_rollout_tick_device: null argument");
    if (!io->dLand || !io->dInfo) return fail(h, CMPC_ERR_ARG, "cmpc_rollout_tick_device: dLand and dInfo are needed");
    const bool merge = io->dPrevT || io->dPrevPose || io->dPrevN;
    if (merge && (io->dPrevT == io->dListT || io->dPrevPose == io->dListPose || io->dPrevN == io->dListN))
        return fail(h, CMPC_ERR_ARG, "cmpc_rollout_tick_device: the merged lists must not alias the previous tick's");
    if (merge && (!io->dPlanT || !io->dPlanPose || !io->dPlanN || !io->dPrevT || !io->dPrevPose || !io->dPrevN))
        return fail(h, CMPC_ERR_ARG, "cmpc_rollout_tick_device: the merge needs the planner's and the previous tick's lists");
    if (max_contacts < 1 || !io->dListT || !io->dListPose || !io->dListN || !io->box_upper || !io->box_lower || !io->dState || !io->dP || !io->dX0 || !io->dX ||
        !io->dStateOut || !(io->plant_step > 0) || io->plant_substeps < 1)
        return fail(h, CMPC_ERR_ARG, "cmpc_rollout_tick_device: bad argument");
    if ((io->dPlanCom || io->dPlanH) && (!io->dPlanCom || !io->dPlanH || io->plan_knots < 2 || !(io->plan_dt > 0) || !(io->robot_mass > 0)))
        return fail(h, CMPC_ERR_ARG, "cmpc_rollout_tick_device: bad planner trajectory");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    int rc = upload_box(h, io->box_upper, io->box_lower, st);
    if (rc != CMPC_OK) return rc;
    int lrc = cmpc_launch_tick_pre(h->B, h->cfg.horizon, max_contacts, h->cfg.sampling_time, now, merge ? 1 : 0, io->dPlanT, io->dPlanPose, io->dPlanN, io->dPrevT,
                                   io->dPrevPose, io->dPrevN, io->dListT, io->dListPose, io->dListN, io->dOk, io->dLand, h->dBox, io->dState, io->dWrench, io->dP,
                                   warm ? io->dX : nullptr, io->dX0, io->dPlanCom, io->dPlanH, io->plan_knots, io->plan_dt, io->plan_t_offset, io->robot_mass,
                                   io->com_height, st);
    if (lrc != 0) return fail(h, CMPC_ERR_HIP, std::string("tick (front) launch: ") + hipGetErrorString((hipError_t)lrc));
    rc = solve_device_impl(h, io->dP, io->dX0, io->dX, io->dInfo, stream, warm != 0);
    if (rc != CMPC_OK) return rc;
    lrc = cmpc_launch_tick_post(h->B, h->cfg.horizon, max_contacts, now, (float)h->cfg.gravity, h->dConsts->corners, io->dX, io->dP, io->dState, io->dStateOut, io->dZmp,
                                (float)io->plant_step, io->plant_substeps, (float)io->zmp_half_x, (float)io->zmp_half_y, io->dLand, io->dListT, io->dListPose,
                                io->dListN, st);
    if (lrc != 0) return fail(h, CMPC_ERR_HIP, std::string("tick (back) launch: ") + hipGetErrorString((hipError_t)lrc));
    return CMPC_OK;
}

// the handle's own contact blocks from contact lists (what the class facade's setContactPhaseList calls)
int cmpc_set_contact_lists(cmpc_handle h, int max_contacts, double now, const double* t, const float* pose, const int* n,
                           const float* box_upper, const float* box_lower, int* land)
{
    if (!h) return fail(h, CMPC_ERR_ARG, "cmpc_set_contact_lists: null handle");
    int rc = ensure_buffers(h);
    if (rc) return rc;
    for (int c = 0; box_upper && box_lower && c < 6; ++c)
        if (box_upper[c] < box_lower[c]) return fail(h, CMPC_ERR_ARG, "cmpc_set_contact_lists: bounding box upper < lower");
    rc = cmpc_contacts_sample(h->cfg.horizon, h->cfg.sampling_time, h->B, max_contacts, now, t, pose, n, box_upper, box_lower, h->hP.data(), land);
    if (rc) return fail(h, rc, g_err);
    return CMPC_OK;
}

}  // extern "C"
