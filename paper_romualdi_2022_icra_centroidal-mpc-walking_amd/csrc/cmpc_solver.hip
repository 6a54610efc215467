// Batched centroidal-MPC interior-point solver for gfx950 (MI355X): one workgroup per problem,
// the whole problem (iterate, multipliers, per-stage Riccati factors) resident in LDS.
//
// Replaces CentroidalMPC::advance() of the reference (call site
// src/centroidal-mpc-walking/src/CentroidalMPCBlock.cpp:615; CasADi Opti -> IPOPT) for a batch of
// problems.  The NLP is the one the reference ships as generated code
// (config/robots/ergoCubGazeboV1/tmp.c: nlp_fg :12430, nlp_jac_fg :71962, nlp_hess_l :58926);
// x and p use its index layout (cmpc_device.h).  The algorithm -- Mehrotra predictor-corrector
// primal-dual interior point, stage-wise Riccati factorisation with the previous force as extra
// state, contact velocities eliminated -- is stated once in oracle/ipm_ref.c (CPU, float64), which
// is test infrastructure; this file is the product and shares no code with it.
//
// Arithmetic: float32 storage and matrix work (linearisation, value-function updates, triangular
// solves); float64 where cancellation decides the answer: dynamics defects, right-hand-side /
// costate recursions, and the 30x30 stage Hessian Quu + its Cholesky (barrier terms z/t span
// 1e-6..1e9 next to cost curvature of order 10).
#include "cmpc_device.h"

#define NS CMPC_NS
#define NF CMPC_NF
#define NQ CMPC_NQ
#define NU CMPC_NU
#define NXA CMPC_NXA
#define NI CMPC_NI
#define LP CMPC_LP
#define PLD 39   // leading dim of P (odd: column walks are conflict-free)
#define QLD 47   // leading dim of the solve panel [Qus (15) | I (30) | pad]
#define QCOLS 46  // 15 (Qus) + 30 (identity -> L^{-1}) + 1 (qu -> lq)
#define QULD 31  // leading dim of Quu (doubles)

namespace {

#ifdef CMPC_PROFILE
// diagnostic build only: per-phase shader-clock sums of workgroup 0 -> prm.scratch[0..31]
__device__ long long g_prof[32];
#define PROF_DECL long long pt_ = __builtin_amdgcn_s_memtime()
#define PROF(slot) do { long long n_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x == 0) g_prof[slot] += n_ - pt_; pt_ = n_; } while (0)
#else
#define PROF_DECL
#define PROF(slot)
#endif

struct Ctx {
    // problem
    CmpcLayout L;
    int N;
    float dt;
    const float* sp;     // parameter vector in LDS
    // iterate
    float *S, *U, *T, *Z;
    double* LAM;
    // step
    float *dS, *dU, *dT, *dZ, *d;
    // factors
    float *Lf, *Ws, *lqs;
    // stage workspace
    float *P0, *P1, *Qx, *A, *Bm, *PA, *PB, *PpB, *PpA, *arow, *geo;
    double *Quu, *pv, *pn, *qu, *qs, *Pd, *lqd, *sig, *gco, *redd;
    float* red;
    int* flag;
};

__device__ inline float gam_of(const Ctx& c, int ct, int k) { return c.sp[c.L.p_gam[ct] + k]; }
// reference stores vec(R) column-major: R(r,cc) = R[3*cc + r]
__device__ inline float Rm(const float* R, int r, int cc) { return R[3 * cc + r]; }

__device__ inline bool qfree(const Ctx& c, int k, int m)
{
    const int ct = m / 3, i = m % 3;
    const float lo = c.sp[c.L.p_lo[ct] + 3 * k + i], hi = c.sp[c.L.p_up[ct] + 3 * k + i];
    return gam_of(c, ct, k) < 0.5f && (hi - lo) > 1e-9f;
}
__device__ inline float qlo(const Ctx& c, int k, int m) { return c.sp[c.L.p_lo[m / 3] + 3 * k + m % 3]; }
__device__ inline float qhi(const Ctx& c, int k, int m) { return c.sp[c.L.p_up[m / 3] + 3 * k + m % 3]; }

// friction row i (0..31) of stage k: a = R (sx, sy, -mu)^T, acting on corner i/4
__device__ inline void fric_row(const Ctx& c, const CmpcParams& prm, int k, int i, float a[3])
{
    const int ct = i >> 4, face = i & 3;
    const float* R = c.sp + c.L.p_R[ct] + 9 * k;
    const float sx = (face == 0 || face == 3) ? 1.f : -1.f;
    const float sy = (face < 2) ? 1.f : -1.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) a[r] = sx * Rm(R, r, 0) + sy * Rm(R, r, 1) - prm.mu_fr * Rm(R, r, 2);
}

__device__ inline bool row_active(const Ctx& c, int k, int i) { return i < 32 ? true : qfree(c, k, (i - 32) % 6); }

// a_i^T u - b_i   (<= 0 feasible)
__device__ inline float row_val(const Ctx& c, const CmpcParams& prm, int k, int i, const float* u)
{
    if (i < 32) {
        float a[3];
        fric_row(c, prm, k, i, a);
        const float* f = u + 3 * (i >> 2);
        return a[0] * f[0] + a[1] * f[1] + a[2] * f[2];
    }
    if (i < 38) return u[24 + i - 32] - qhi(c, k, i - 32);
    return qlo(c, k, i - 38) - u[24 + i - 38];
}
__device__ inline float row_dot(const Ctx& c, const CmpcParams& prm, int k, int i, const float* du)
{
    if (i < 32) {
        float a[3];
        fric_row(c, prm, k, i, a);
        const float* f = du + 3 * (i >> 2);
        return a[0] * f[0] + a[1] * f[1] + a[2] * f[2];
    }
    if (i < 38) return du[24 + i - 32];
    return -du[24 + i - 38];
}

__device__ inline float qdiag(const CmpcParams& prm, int k, int i)
{
    if (i == 0) return 2.f * prm.w_com0;
    if (i == 1) return 2.f * prm.w_com1;
    if (i == 2) return prm.wz2[k];
    if (i < 6) return 0.f;
    if (i < 9) return 2.f * prm.w_h;
    return 2.f * prm.w_pos;
}

// gradient of the tracking cost w.r.t. component i of s_k
__device__ inline double grad_track(const Ctx& c, const CmpcParams& prm, int k, int i)
{
    const float* s = c.S + NS * k;
    if (i < 3) return (double)qdiag(prm, k, i) * ((double)s[i] - (double)c.sp[c.L.p_comref + 3 * k + i]);
    if (i < 6) return 0.0;
    if (i < 9) return 2.0 * prm.w_h * ((double)s[i] - (double)c.sp[c.L.p_href + 3 * k + i - 6]);
    const int ct = (i - 9) / 3, a = (i - 9) % 3;
    return 2.0 * prm.w_pos * ((double)s[i] - (double)c.sp[c.L.p_nom[ct] + 3 * k + a]);
}

// gradient of the force-symmetry cost w.r.t. force component m (0..23) of stage k
__device__ inline double grad_sym(const Ctx& c, const CmpcParams& prm, int k, int m)
{
    const int ct = m / 12, i = m % 3;
    const float* u = c.U + NU * k + 12 * ct;
    const double gam = gam_of(c, ct, k);
    const double mean = 0.25 * ((double)u[i] + (double)u[3 + i] + (double)u[6 + i] + (double)u[9 + i]);
    const double esum = 4.0 * mean * (1.0 - gam);
    const double e = (double)c.U[NU * k + m] - gam * mean;
    return 2.0 * prm.w_sym * (e - 0.25 * gam * esum);
}

template <int NT>
__device__ inline float block_max(float v, float* red, int tid)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) r = fmaxf(r, red[w]);
    return r;
}
template <int NT>
__device__ inline float block_min(float v, float* red, int tid) { return -block_max<NT>(-v, red, tid); }
template <int NT>
__device__ inline double block_sum(double v, double* red, int tid)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double r = red[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) r += red[w];
    return r;
}

// ---- geometry of stage k: r_cj = R c_j + pos_c - com (8x3), Fc (2x3, plain sums), Fsum (gam-weighted) ----
// geo layout: r[24] | Fc[6] | Fsum[3]
__device__ inline void stage_geo(const Ctx& c, const CmpcParams& prm, int k, int tid)
{
    const float* s = c.S + NS * k;
    const float* u = c.U + NU * k;
    if (tid < 24) {
        const int ct = tid / 12, j = (tid % 12) / 3, i = tid % 3;
        const float* R = c.sp + c.L.p_R[ct] + 9 * k;
        const float* cn = prm.corners + 12 * ct + 3 * j;
        c.geo[tid] = Rm(R, i, 0) * cn[0] + Rm(R, i, 1) * cn[1] + Rm(R, i, 2) * cn[2] + s[9 + 3 * ct + i] - s[i];
    } else if (tid < 30) {
        const int ct = (tid - 24) / 3, i = (tid - 24) % 3;
        const float* f = u + 12 * ct;
        c.geo[tid] = f[i] + f[3 + i] + f[6 + i] + f[9 + i];
    } else if (tid < 33) {
        const int i = tid - 30;
        float v = 0.f;
        for (int ct = 0; ct < 2; ++ct) {
            const float* f = u + 12 * ct;
            v += gam_of(c, ct, k) * (f[i] + f[3 + i] + f[6 + i] + f[9 + i]);
        }
        c.geo[tid] = v;
    }
}

// dynamics defect component i of stage k in float64: phi_k(s_k,u_k)[i] - s_{k+1}[i]
__device__ inline double defect(const Ctx& c, const CmpcParams& prm, int k, int i)
{
    const float* s = c.S + NS * k;
    const float* u = c.U + NU * k;
    const float* sn = c.S + NS * (k + 1);
    const double dt = prm.dt;
    if (i < 3) return (double)s[i] + dt * (double)s[3 + i] - (double)sn[i];
    if (i < 6) {
        const int a = i - 3;
        double acc = (double)c.sp[c.L.p_fext + 3 * k + a] - (a == 2 ? (double)prm.grav : 0.0);
        for (int ct = 0; ct < 2; ++ct) {
            const float* f = u + 12 * ct;
            acc += (double)gam_of(c, ct, k) * ((double)f[a] + (double)f[3 + a] + (double)f[6 + a] + (double)f[9 + a]);
        }
        return (double)s[i] + dt * acc - (double)sn[i];
    }
    if (i < 9) {
        const int a = i - 6, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
        double tor = (double)c.sp[c.L.p_text + 3 * k + a];
        for (int ct = 0; ct < 2; ++ct) {
            const float* R = c.sp + c.L.p_R[ct] + 9 * k;
            const double gam = gam_of(c, ct, k);
            double t = 0.0;
            for (int j = 0; j < 4; ++j) {
                const float* cn = prm.corners + 12 * ct + 3 * j;
                const float* f = u + 12 * ct + 3 * j;
                const double r1 = (double)Rm(R, a1, 0) * cn[0] + (double)Rm(R, a1, 1) * cn[1] + (double)Rm(R, a1, 2) * cn[2]
                                  + (double)s[9 + 3 * ct + a1] - (double)s[a1];
                const double r2 = (double)Rm(R, a2, 0) * cn[0] + (double)Rm(R, a2, 1) * cn[1] + (double)Rm(R, a2, 2) * cn[2]
                                  + (double)s[9 + 3 * ct + a2] - (double)s[a2];
                t += r1 * (double)f[a2] - r2 * (double)f[a1];
            }
            tor += gam * t;
        }
        return (double)s[i] + dt * tor - (double)sn[i];
    }
    {
        const int ct = (i - 9) / 3, a = (i - 9) % 3;
        const float* R = c.sp + c.L.p_R[ct] + 9 * k;
        const double gam = gam_of(c, ct, k);
        double land = (double)c.sp[c.L.p_nom[ct] + 3 * (k + 1) + a];
        for (int m = 0; m < 3; ++m) land += (double)Rm(R, a, m) * (double)u[24 + 3 * ct + m];
        return gam * (double)s[i] + (1.0 - gam) * land - (double)sn[i];
    }
}

// (B_k^T v)[m] for a 15-vector v (double), closed form (needs stage_geo of stage k)
__device__ inline double Bt_vec(const Ctx& c, const CmpcParams& prm, int k, int m, const double* v)
{
    if (m < 24) {
        const int ct = m / 12, i = m % 3;
        const float* r = c.geo + 3 * (m / 3);
        const int a1 = (i + 1) % 3, a2 = (i + 2) % 3;
        // (lam_h x r)_i = lh[a1] r[a2] - lh[a2] r[a1]
        return (double)prm.dt * (double)gam_of(c, ct, k) * (v[3 + i] + v[6 + a1] * (double)r[a2] - v[6 + a2] * (double)r[a1]);
    }
    const int q = m - 24, ct = q / 3, a = q % 3;
    if (!qfree(c, k, q)) return 0.0;
    const float* R = c.sp + c.L.p_R[ct] + 9 * k;
    const double g1 = 1.0 - (double)gam_of(c, ct, k);
    return g1 * ((double)Rm(R, 0, a) * v[9 + 3 * ct] + (double)Rm(R, 1, a) * v[10 + 3 * ct] + (double)Rm(R, 2, a) * v[11 + 3 * ct]);
}

// (A_k^T v)[i] for a 15-vector v (double), closed form (needs stage_geo of stage k)
__device__ inline double At_vec(const Ctx& c, const CmpcParams& prm, int k, int i, const double* v)
{
    const double dt = prm.dt;
    if (i < 3) {  // com: v_com + dt (v_h x Fsum)_i
        const int a1 = (i + 1) % 3, a2 = (i + 2) % 3;
        const float* Fs = c.geo + 30;
        return v[i] + dt * (v[6 + a1] * (double)Fs[a2] - v[6 + a2] * (double)Fs[a1]);
    }
    if (i < 6) return dt * v[i - 3] + v[i];
    if (i < 9) return v[i];
    const int ct = (i - 9) / 3, a = (i - 9) % 3, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
    const float* Fc = c.geo + 24 + 3 * ct;
    const double gam = gam_of(c, ct, k);
    // gam v_pos + dt gam (Fc x v_h)_a
    return gam * (v[i] + dt * ((double)Fc[a1] * v[6 + a2] - (double)Fc[a2] * v[6 + a1]));
}

// (A_k ds + B_k du)[i]  (float, closed form; needs stage_geo of stage k)
__device__ inline float AB_step(const Ctx& c, const CmpcParams& prm, int k, int i, const float* ds, const float* du)
{
    const float dt = prm.dt;
    if (i < 3) return ds[i] + dt * ds[3 + i];
    if (i < 6) {
        const int a = i - 3;
        float acc = 0.f;
        for (int ct = 0; ct < 2; ++ct) {
            const float* f = du + 12 * ct;
            acc += gam_of(c, ct, k) * (f[a] + f[3 + a] + f[6 + a] + f[9 + a]);
        }
        return ds[i] + dt * acc;
    }
    if (i < 9) {
        const int a = i - 6, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
        float tor = 0.f;
        for (int ct = 0; ct < 2; ++ct) {
            float t = 0.f;
            const float e1 = ds[9 + 3 * ct + a1] - ds[a1], e2 = ds[9 + 3 * ct + a2] - ds[a2];
            for (int j = 0; j < 4; ++j) {
                const float* r = c.geo + 12 * ct + 3 * j;
                const float* df = du + 12 * ct + 3 * j;
                const float* f = c.U + NU * k + 12 * ct + 3 * j;
                t += r[a1] * df[a2] - r[a2] * df[a1] + e1 * f[a2] - e2 * f[a1];
            }
            tor += gam_of(c, ct, k) * t;
        }
        return ds[i] + dt * tor;
    }
    const int ct = (i - 9) / 3, a = (i - 9) % 3;
    const float* R = c.sp + c.L.p_R[ct] + 9 * k;
    const float gam = gam_of(c, ct, k);
    float land = 0.f;
    for (int m = 0; m < 3; ++m)
        if (qfree(c, k, 3 * ct + m)) land += Rm(R, a, m) * du[24 + 3 * ct + m];
    return gam * ds[i] + (1.f - gam) * land;
}

// ---- dense A (15x15) and B (15x30) of stage k into LDS (needs stage_geo) ----
template <int NT>
__device__ inline void build_AB(const Ctx& c, const CmpcParams& prm, int k, int tid)
{
    for (int e = tid; e < NS * NS + NS * NU; e += NT) {
        if (e < NS * NS) c.A[e] = 0.f;
        else c.Bm[e - NS * NS] = 0.f;
    }
    __syncthreads();
    const float dt = prm.dt;
    if (tid < 8) {  // corner columns of B
        const int ct = tid >> 2, col = 3 * tid;
        const float g = dt * gam_of(c, ct, k);
        const float* r = c.geo + 3 * tid;
        for (int i = 0; i < 3; ++i) c.Bm[(3 + i) * NU + col + i] = g;
        c.Bm[6 * NU + col + 1] = -g * r[2]; c.Bm[6 * NU + col + 2] = g * r[1];
        c.Bm[7 * NU + col + 0] = g * r[2];  c.Bm[7 * NU + col + 2] = -g * r[0];
        c.Bm[8 * NU + col + 0] = -g * r[1]; c.Bm[8 * NU + col + 1] = g * r[0];
    } else if (tid < 10) {  // contact blocks
        const int ct = tid - 8, col = 9 + 3 * ct;
        const float gam = gam_of(c, ct, k);
        const float g = -dt * gam;
        const float* Fc = c.geo + 24 + 3 * ct;
        const float* R = c.sp + c.L.p_R[ct] + 9 * k;
        c.A[6 * NS + col + 1] = -g * Fc[2]; c.A[6 * NS + col + 2] = g * Fc[1];
        c.A[7 * NS + col + 0] = g * Fc[2];  c.A[7 * NS + col + 2] = -g * Fc[0];
        c.A[8 * NS + col + 0] = -g * Fc[1]; c.A[8 * NS + col + 1] = g * Fc[0];
        for (int i = 0; i < 3; ++i) {
            c.A[(col + i) * NS + col + i] = gam;
            for (int a = 0; a < 3; ++a)
                if (qfree(c, k, 3 * ct + a)) c.Bm[(col + i) * NU + 24 + 3 * ct + a] = (1.f - gam) * Rm(R, i, a);
        }
    } else if (tid == 10) {
        const float* Fs = c.geo + 30;
        for (int i = 0; i < 9; ++i) c.A[i * NS + i] = 1.f;
        for (int i = 0; i < 3; ++i) c.A[i * NS + 3 + i] = dt;
        c.A[6 * NS + 1] = -dt * Fs[2]; c.A[6 * NS + 2] = dt * Fs[1];
        c.A[7 * NS + 0] = dt * Fs[2];  c.A[7 * NS + 2] = -dt * Fs[0];
        c.A[8 * NS + 0] = -dt * Fs[1]; c.A[8 * NS + 1] = dt * Fs[0];
    }
    __syncthreads();
}

__device__ inline int lpk(int i, int j) { return i * (i + 1) / 2 + j; }  // packed lower (i >= j)

// ---- Riccati backward sweep (matrices + right-hand side).  dZ holds the per-row complementarity
// target (0 for the affine step).  Returns (uniformly) 0 ok, 1 non-positive pivot. ----
template <int NT>
__device__ int riccati_backward(const Ctx& c, const CmpcParams& prm, int tid, bool use_exact, bool affine)
{
    const int N = c.N;
    float* Pcur = c.P0;  // value function of stage k+1
    float* Pnew = c.P1;
    bool havep = false;
    // terminal value
    for (int e = tid; e < NXA * PLD; e += NT) Pcur[e] = 0.f;
    __syncthreads();
    if (tid < NS) {
        Pcur[tid * PLD + tid] = qdiag(prm, N, tid);
        c.pv[tid] = grad_track(c, prm, N, tid);
    } else if (tid < NXA) c.pv[tid] = 0.0;
    __syncthreads();

    for (int k = N - 1; k >= 0; --k) {
        const bool pk = k > 0;
        const float* u = c.U + NU * k;
        PROF_DECL;
        stage_geo(c, prm, k, tid);
        __syncthreads();
        build_AB<NT>(c, prm, k, tid);
        PROF(0);
        // ---- per-row barrier coefficients ----
        if (tid < NI) {
            double sg = 0.0, gc = 0.0;
            if (row_active(c, k, tid)) {
                const double t = c.T[NI * k + tid], z = c.Z[NI * k + tid];
                const double cmu = affine ? 0.0 : (double)c.dZ[NI * k + tid];
                const double r = (double)row_val(c, prm, k, tid, u) + t;
                sg = z / t;
                gc = cmu / t + sg * r;
            }
            c.sig[tid] = sg; c.gco[tid] = gc;
            if (tid < 32) {
                float a[3];
                fric_row(c, prm, k, tid, a);
                c.arow[3 * tid] = a[0]; c.arow[3 * tid + 1] = a[1]; c.arow[3 * tid + 2] = a[2];
            }
        }
        // ---- PB = Pss B, PA = Pss A, PpB = Pps B, PpA = Pps A ----
        for (int e = tid; e < NS * NU + NS * NS + (havep ? NF * NU + NF * NS : 0); e += NT) {
            if (e < NS * NU) {
                const int i = e / NU, j = e % NU;
                const int a0 = j < 24 ? 3 : 9 + 3 * ((j - 24) / 3), a1 = j < 24 ? 9 : a0 + 3;
                float v = 0.f;
                for (int a = a0; a < a1; ++a) v += Pcur[i * PLD + a] * c.Bm[a * NU + j];
                c.PB[e] = v;
            } else if (e < NS * NU + NS * NS) {
                const int e2 = e - NS * NU, i = e2 / NS, j = e2 % NS;
                float v = 0.f;
                for (int a = 0; a < NS; ++a) v += Pcur[i * PLD + a] * c.A[a * NS + j];
                c.PA[e2] = v;
            } else if (e < NS * NU + NS * NS + NF * NU) {
                const int e2 = e - NS * NU - NS * NS, i = e2 / NU, j = e2 % NU;
                const int a0 = j < 24 ? 3 : 9 + 3 * ((j - 24) / 3), a1 = j < 24 ? 9 : a0 + 3;
                float v = 0.f;
                for (int a = a0; a < a1; ++a) v += Pcur[(NS + i) * PLD + a] * c.Bm[a * NU + j];
                c.PpB[e2] = v;
            } else {
                const int e2 = e - NS * NU - NS * NS - NF * NU, i = e2 / NS, j = e2 % NS;
                float v = 0.f;
                for (int a = 0; a < NS; ++a) v += Pcur[(NS + i) * PLD + a] * c.A[a * NS + j];
                c.PpA[e2] = v;
            }
        }
        // Pd = P [d; 0] + pv  (float64, 39)
        if (tid < NXA) {
            double acc = c.pv[tid];
            if (tid < NS || havep)
                for (int a = 0; a < NS; ++a) acc += (double)Pcur[tid * PLD + a] * (double)c.d[NS * k + a];
            c.Pd[tid] = acc;
        }
        __syncthreads();
        PROF(1);
        // ---- Quu (float64) ----
        for (int e = tid; e < NU * NU; e += NT) {
            const int i = e / NU, j = e % NU;
            double v = 0.0;
            if (i < NF && j < NF) {
                const int ci = i / 12, cj = j / 12;
                if (ci == cj && (i % 3) == (j % 3)) {
                    const double gam = gam_of(c, ci, k);
                    v = 2.0 * prm.w_sym * ((i == j ? 1.0 : 0.0) - 0.25 * gam * (2.0 - gam));
                }
                if (i / 3 == j / 3) {  // same corner: friction barrier
                    const int r0 = 4 * (i / 3);
#pragma unroll
                    for (int f = 0; f < 4; ++f)
                        v += c.sig[r0 + f] * (double)c.arow[3 * (r0 + f) + i % 3] * (double)c.arow[3 * (r0 + f) + j % 3];
                }
                if (i == j && pk) v += (double)prm.D[i % 3];
                if (havep) v += (double)Pcur[(NS + i) * PLD + NS + j];
            } else if (i == j) {  // q diagonal
                v = qfree(c, k, i - 24) ? c.sig[32 + i - 24] + c.sig[38 + i - 24] : 1.0;
            }
            {   // B^T P B
                const int a0 = i < 24 ? 3 : 9 + 3 * ((i - 24) / 3), a1 = i < 24 ? 9 : a0 + 3;
                double acc = 0.0;
                for (int a = a0; a < a1; ++a) acc += (double)c.Bm[a * NU + i] * (double)c.PB[a * NU + j];
                v += acc;
            }
            if (havep) {
                if (i < NF) v += (double)c.PpB[i * NU + j];
                if (j < NF) v += (double)c.PpB[j * NU + i];
            }
            c.Quu[i * QULD + j] = v;
        }
        PROF(2);
        // ---- solve panel Qx = [Qus | I] ----
        for (int e = tid; e < NU * QCOLS; e += NT) {
            const int i = e / QCOLS, j = e % QCOLS;
            float v;
            if (j < NS) {
                const int a0 = i < 24 ? 3 : 9 + 3 * ((i - 24) / 3), a1 = i < 24 ? 9 : a0 + 3;
                v = 0.f;
                for (int a = a0; a < a1; ++a) v += c.Bm[a * NU + i] * c.PA[a * NS + j];
                if (havep && i < NF) v += c.PpA[i * NS + j];
                if (use_exact && i < NF) {
                    // S[f_cj, pos_c] = dt gam [lam_h]x ; S[f_cj, com] = -dt gam [lam_h]x
                    const int ct = i / 12, a = i % 3;
                    int b = -1;
                    float sgn = 0.f;
                    if (j < 3) { b = j; sgn = -1.f; }
                    else if (j >= 9 + 3 * ct && j < 12 + 3 * ct) { b = j - 9 - 3 * ct; sgn = 1.f; }
                    if (b >= 0 && b != a) {
                        // [l]x(a,b): (0,1)=-l2 (0,2)=l1 (1,0)=l2 (1,2)=-l0 (2,0)=-l1 (2,1)=l0
                        const int o = 3 - a - b;
                        const float lv = (float)c.LAM[NS * (k + 1) + 6 + o];
                        const float sk = ((b - a + 3) % 3 == 1) ? -lv : lv;
                        v += sgn * prm.dt * gam_of(c, ct, k) * sk;
                    }
                }
            } else if (j < NS + NU) {
                v = (j - NS == i) ? 1.f : 0.f;
            } else continue;  // column 45 (qu) is written with the right-hand sides below
            c.Qx[i * QLD + j] = v;
        }
        PROF(3);
        // ---- right-hand sides (float64) ----
        if (tid < NU) {
            double g;
            if (tid < NF) {
                g = grad_sym(c, prm, k, tid);
                const int r0 = 4 * (tid / 3);
#pragma unroll
                for (int f = 0; f < 4; ++f) g += c.gco[r0 + f] * (double)c.arow[3 * (r0 + f) + tid % 3];
                if (pk) g += (double)prm.D[tid % 3] * ((double)u[tid] - (double)c.U[NU * (k - 1) + tid]);
                if (havep) g += c.Pd[NS + tid];
            } else {
                const int q = tid - 24;
                g = qfree(c, k, q) ? c.gco[32 + q] - c.gco[38 + q] : 0.0;
            }
            g += Bt_vec(c, prm, k, tid, c.Pd);
            // qu -> 0 at convergence, so float keeps its relative accuracy through the solve
            c.Qx[tid * QLD + NS + NU] = (float)g;
        } else if (tid >= 64 && tid < 64 + NS) {
            const int i = tid - 64;
            c.qs[i] = grad_track(c, prm, k, i) + At_vec(c, prm, k, i, c.Pd);
        }
        __syncthreads();
        PROF(4);
        // ---- Qss = A^T PA + Q  -> Pnew ss block (W^T W subtracted below) ----
        for (int e = tid; e < NS * NS; e += NT) {
            const int i = e / NS, j = e % NS;
            float v = (i == j) ? qdiag(prm, k, i) : 0.f;
            for (int a = 0; a < NS; ++a) v += c.A[a * NS + i] * c.PA[a * NS + j];
            Pnew[i * PLD + j] = v;
        }
        PROF(5);
        // ---- Cholesky of Quu in float64 (right-looking; L columns land in place, scaled) ----
        for (int j = 0; j < NU; ++j) {
            const double piv = c.Quu[j * QULD + j];
            if (!(piv > 0.0)) return 1;  // uniform: every thread reads the same LDS word
            const double rinv = 1.0 / sqrt(piv);
            __syncthreads();  // everyone has read the pivot / column before it is rescaled
            // trailing update uses the unscaled column j: A[i][c] -= A[i][j] A[c][j] / piv
            const int n = NU - 1 - j;
            for (int e = tid; e < n * n; e += NT) {
                const int i = j + 1 + e / n, cc = j + 1 + e % n;
                if (cc <= i) c.Quu[i * QULD + cc] -= c.Quu[i * QULD + j] * c.Quu[cc * QULD + j] * (rinv * rinv);
            }
            __syncthreads();
            if (tid >= j && tid < NU) c.Quu[tid * QULD + j] *= rinv;
        }
        __syncthreads();
        PROF(6);
        // ---- W = L^{-1} [Qus | I]  (one column per thread, float32), lq = L^{-1} qu (float64) ----
        if (tid < QCOLS) {
            const int j = tid;
            const int i0 = (j < NS || j == NS + NU) ? 0 : j - NS;  // identity columns start at their own row
            for (int i = i0; i < NU; ++i) {
                float v = c.Qx[i * QLD + j];
                for (int a = i0; a < i; ++a) v -= (float)c.Quu[i * QULD + a] * c.Qx[a * QLD + j];
                c.Qx[i * QLD + j] = v / (float)c.Quu[i * QULD + i];
            }
        }
        __syncthreads();
        if (tid < NU) c.lqd[tid] = (double)c.Qx[tid * QLD + NS + NU];
        PROF(7);
        // ---- store factors: Linv (packed lower), Ws, lq ----
        {
            float* Lf = c.Lf + (size_t)LP * k;
            float* Ws = c.Ws + (size_t)(NU * NS) * k;
            for (int e = tid; e < NU * NU; e += NT) {
                const int i = e / NU, j = e % NU;
                if (j <= i) Lf[lpk(i, j)] = c.Qx[i * QLD + NS + j];
            }
            for (int e = tid; e < NU * NS; e += NT) Ws[e] = c.Qx[(e / NS) * QLD + e % NS];
            if (tid < NU) c.lqs[NU * k + tid] = c.Qx[tid * QLD + NS + NU];
        }
        __syncthreads();
        PROF(8);
        // ---- value function of stage k:  P = [Qss 0; 0 D] - W^T W,  W = [Ws | Wp], Wp = -Linv[:, :24] D ----
        {
            const int ncol = pk ? NXA : NS;
            for (int e = tid; e < ncol * ncol; e += NT) {
                const int i = e / ncol, j = e % ncol;
                if (j > i) continue;
                float v = 0.f;
                const int ci = i < NS ? i : NS + (i - NS), cj = j < NS ? j : NS + (j - NS);
                const int a0 = i < NS ? 0 : i - NS;  // Linv[a][i-NS] = 0 for a < i-NS
                for (int a = a0; a < NU; ++a) v += c.Qx[a * QLD + ci] * c.Qx[a * QLD + cj];
                if (i >= NS) v *= prm.D[(i - NS) % 3];
                if (j >= NS) v *= prm.D[(j - NS) % 3];
                if ((i >= NS) != (j >= NS)) v = -v;  // one factor of (-D)
                // (-D)(-D) = +, so ss and pp blocks get -W^T W
                float base = 0.f;
                if (i < NS) base = Pnew[i * PLD + j];       // Qss
                else if (i == j) base = prm.D[(i - NS) % 3];
                float r;
                if ((i >= NS) != (j >= NS)) r = base - v;    // sp block: -Ws^T Wp = +Ws^T Linv D -> v already negated
                else r = base - v;
                Pnew[i * PLD + j] = r;
                Pnew[j * PLD + i] = r;
            }
            // gradient of the value function (float64)
            if (tid >= 64 && tid < 64 + ncol) {
                const int i = tid - 64;
                double v = 0.0;
                if (i < NS) {
                    for (int a = 0; a < NU; ++a) v += (double)c.Qx[a * QLD + i] * c.lqd[a];
                    c.pn[i] = c.qs[i] - v;
                } else {
                    const int m = i - NS;
                    for (int a = m; a < NU; ++a) v += (double)c.Qx[a * QLD + NS + m] * c.lqd[a];
                    // qp = -D (u - u_prev);  Wp^T lq = -D Linv[:,m]^T lq
                    c.pn[i] = -(double)prm.D[m % 3] * ((double)u[m] - (double)c.U[NU * (k - 1) + m]) + (double)prm.D[m % 3] * v;
                }
            }
        }
        __syncthreads();
        if (tid < NXA) c.pv[tid] = (tid < (pk ? NXA : NS)) ? c.pn[tid] : 0.0;
        {   // swap
            float* t = Pcur; Pcur = Pnew; Pnew = t;
        }
        havep = pk;
        __syncthreads();
        PROF(9);
    }
    return 0;
}

// ---- vector-only backward sweep for the corrector: the row coefficients change by DG = CMU/t
// (dZ holds CMU); updates lq in place ----
template <int NT>
__device__ void riccati_delta(const Ctx& c, const CmpcParams& prm, int tid)
{
    const int N = c.N;
    bool havep = false;
    if (tid < NXA) c.pv[tid] = 0.0;
    __syncthreads();
    for (int k = N - 1; k >= 0; --k) {
        const bool pk = k > 0;
        const float* Lf = c.Lf + (size_t)LP * k;
        const float* Ws = c.Ws + (size_t)(NU * NS) * k;
        stage_geo(c, prm, k, tid);
        if (tid >= 64 && tid < 64 + NI) {
            const int i = tid - 64;
            c.gco[i] = row_active(c, k, i) ? (double)c.dZ[NI * k + i] / (double)c.T[NI * k + i] : 0.0;
            if (i < 32) {
                float a[3];
                fric_row(c, prm, k, i, a);
                c.arow[3 * i] = a[0]; c.arow[3 * i + 1] = a[1]; c.arow[3 * i + 2] = a[2];
            }
        }
        __syncthreads();
        if (tid < NU) {
            double g;
            if (tid < NF) {
                const int r0 = 4 * (tid / 3);
                g = 0.0;
#pragma unroll
                for (int f = 0; f < 4; ++f) g += c.gco[r0 + f] * (double)c.arow[3 * (r0 + f) + tid % 3];
                if (havep) g += c.pv[NS + tid];
            } else {
                const int q = tid - 24;
                g = qfree(c, k, q) ? c.gco[32 + q] - c.gco[38 + q] : 0.0;
            }
            c.qu[tid] = g + Bt_vec(c, prm, k, tid, c.pv);
        }
        __syncthreads();
        if (tid < NU) {  // dl = Linv dq (lower-triangular mat-vec)
            double v = 0.0;
            for (int a = 0; a <= tid; ++a) v += (double)Lf[lpk(tid, a)] * c.qu[a];
            c.lqd[tid] = v;
            c.lqs[NU * k + tid] += (float)v;
        }
        __syncthreads();
        if (tid < NXA) {
            double v;
            if (tid < NS) {
                v = At_vec(c, prm, k, tid, c.pv);
                for (int a = 0; a < NU; ++a) v -= (double)Ws[a * NS + tid] * c.lqd[a];
            } else if (pk) {
                const int m = tid - NS;
                v = 0.0;
                for (int a = m; a < NU; ++a) v += (double)Lf[lpk(a, m)] * c.lqd[a];
                v *= (double)prm.D[m % 3];
            } else v = 0.0;
            c.pn[tid] = v;
        }
        __syncthreads();
        if (tid < NXA) c.pv[tid] = c.pn[tid];
        havep = pk;
        __syncthreads();
    }
}

// ---- forward sweep: dS, dU, then dT and dZ (dZ holds the complementarity target on entry
// unless affine) ----
template <int NT>
__device__ void riccati_forward(const Ctx& c, const CmpcParams& prm, int tid, bool affine)
{
    const int N = c.N;
    float* y = c.red + 8;  // 30 floats of scratch
    if (tid < NS) c.dS[tid] = 0.f;
    __syncthreads();
    for (int k = 0; k < N; ++k) {
        const float* Lf = c.Lf + (size_t)LP * k;
        const float* Ws = c.Ws + (size_t)(NU * NS) * k;
        stage_geo(c, prm, k, tid);
        if (tid >= 64 && tid < 64 + NU) {
            const int i = tid - 64;
            float v = c.lqs[NU * k + i];
            for (int a = 0; a < NS; ++a) v += Ws[i * NS + a] * c.dS[NS * k + a];
            if (k > 0) {  // Wp dp = -Linv[:, :24] D dp
                const float* dp = c.dU + NU * (k - 1);
                const int na = i < NF ? i + 1 : NF;
                for (int a = 0; a < na; ++a) v -= Lf[lpk(i, a)] * prm.D[a % 3] * dp[a];
            }
            y[i] = -v;
        }
        __syncthreads();
        if (tid < NU) {  // du = Linv^T y
            float v = 0.f;
            for (int a = tid; a < NU; ++a) v += Lf[lpk(a, tid)] * y[a];
            c.dU[NU * k + tid] = v;
        }
        __syncthreads();
        if (tid < NS) c.dS[NS * (k + 1) + tid] = AB_step(c, prm, k, tid, c.dS + NS * k, c.dU + NU * k) + c.d[NS * k + tid];
        __syncthreads();
    }
    for (int e = tid; e < N * NI; e += NT) {
        const int k = e / NI, i = e % NI;
        float dt_ = 0.f, dz_ = 0.f;
        if (row_active(c, k, i)) {
            const float t = c.T[e], z = c.Z[e];
            const float cmu = affine ? 0.f : c.dZ[e];
            const float r = row_val(c, prm, k, i, c.U + NU * k) + t;
            dt_ = -r - row_dot(c, prm, k, i, c.dU + NU * k);
            dz_ = (cmu - z * t) / t - (z / t) * dt_;
        }
        c.dT[e] = dt_; c.dZ[e] = dz_;
    }
    __syncthreads();
}

// largest step lengths keeping t, z positive (fraction tau to the boundary)
template <int NT>
__device__ void step_lengths(const Ctx& c, int tid, float tau, float& ap, float& ad)
{
    float a_p = 1.f, a_d = 1.f;
    for (int e = tid; e < c.N * NI; e += NT) {
        const float dt_ = c.dT[e], dz_ = c.dZ[e];
        if (dt_ < 0.f) a_p = fminf(a_p, -tau * c.T[e] / dt_);
        if (dz_ < 0.f) a_d = fminf(a_d, -tau * c.Z[e] / dz_);
    }
    ap = block_min<NT>(a_p, c.red, tid);
    ad = block_min<NT>(a_d, c.red, tid);
}

// new costates (backward), blended into LAM with step ap:  lam_k = gs_k + Q_k ds_k + S_k^T du_k + A_k^T lam_{k+1}
template <int NT>
__device__ void costate_update(const Ctx& c, const CmpcParams& prm, int tid, float ap, bool use_exact)
{
    const int N = c.N;
    // pv <- full-step lam_{k+1}
    if (tid < NS) c.pv[tid] = grad_track(c, prm, N, tid) + (double)qdiag(prm, N, tid) * (double)c.dS[NS * N + tid];
    __syncthreads();
    if (tid < NS) c.LAM[NS * N + tid] += (double)ap * (c.pv[tid] - c.LAM[NS * N + tid]);
    for (int k = N - 1; k >= 1; --k) {
        stage_geo(c, prm, k, tid);
        // dFc (gam-weighted force-step sums) for the S^T du term, using the OLD lam_{k+1} (the one the Hessian used)
        if (tid >= 64 && tid < 70) {
            const int ct = (tid - 64) / 3, i = (tid - 64) % 3;
            const float* df = c.dU + NU * k + 12 * ct;
            c.gco[tid - 64] = (double)gam_of(c, ct, k) * ((double)df[i] + df[3 + i] + df[6 + i] + df[9 + i]);
        }
        __syncthreads();
        if (tid < NS) {
            double v = grad_track(c, prm, k, tid) + (double)qdiag(prm, k, tid) * (double)c.dS[NS * k + tid] + At_vec(c, prm, k, tid, c.pv);
            if (use_exact && (tid < 3 || tid >= 9)) {
                // Sx = dt [lh]x with lh = lam_h,k+1 used in the Hessian (= c.qs scratch holds it)
                const double* lh = c.qs;  // 3 doubles, old lam_h of stage k+1
                const int i = tid < 3 ? tid : (tid - 9) % 3;
                const int a1 = (i + 1) % 3, a2 = (i + 2) % 3;
                double F[3];
                if (tid < 3) { F[0] = c.gco[0] + c.gco[3]; F[1] = c.gco[1] + c.gco[4]; F[2] = c.gco[2] + c.gco[5]; }
                else { const int ct = (tid - 9) / 3; F[0] = c.gco[3 * ct]; F[1] = c.gco[3 * ct + 1]; F[2] = c.gco[3 * ct + 2]; }
                // (Sx^T F)_i = dt (F x lh)_i ... Sx^T = -dt[lh]x  -> (Sx^T F) = -dt (lh x F) = dt (F x lh)
                const double sxtf = (double)prm.dt * (F[a1] * lh[a2] - F[a2] * lh[a1]);
                v += (tid < 3) ? -sxtf : sxtf;
            }
            c.pn[tid] = v;
        }
        __syncthreads();
        if (tid < NS) {
            c.pv[tid] = c.pn[tid];
            if (tid >= 6 && tid < 9) c.qs[tid - 6] = c.LAM[NS * k + tid];  // old lam_h,k for the next (k-1) stage
            c.LAM[NS * k + tid] += (double)ap * (c.pn[tid] - c.LAM[NS * k + tid]);
        }
        __syncthreads();
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void cmpc_solve_kernel(CmpcParams prm)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int N = prm.N;
    const long long t_start = __builtin_amdgcn_s_memtime();
    Ctx c;
    cmpc_layout_init(c.L, N);
    c.N = N; c.dt = prm.dt;
    // ---- carve LDS: doubles first ----
    double* dp = reinterpret_cast<double*>(smem);
    c.LAM = dp; dp += NS * (N + 1);
    c.Quu = dp; dp += NU * QULD;
    c.pv = dp; dp += 40; c.pn = dp; dp += 40; c.qu = dp; dp += 30; c.qs = dp; dp += 16; c.Pd = dp; dp += 40;
    c.lqd = dp; dp += 30; c.sig = dp; dp += NI; c.gco = dp; dp += NI; c.redd = dp; dp += 8;
    float* fp = reinterpret_cast<float*>(dp);
    float* spw = fp; fp += (c.L.np + 3) & ~3;
    c.sp = spw;
    c.S = fp; fp += NS * (N + 1); c.U = fp; fp += NU * N; c.T = fp; fp += NI * N; c.Z = fp; fp += NI * N;
    c.dS = fp; fp += NS * (N + 1); c.dU = fp; fp += NU * N; c.dT = fp; fp += NI * N; c.dZ = fp; fp += NI * N;
    c.d = fp; fp += NS * N;
    c.lqs = fp; fp += NU * N;
    c.P0 = fp; fp += NXA * PLD; c.P1 = fp; fp += NXA * PLD;
    c.Qx = fp; fp += NU * QLD;
    c.A = fp; fp += NS * NS; c.Bm = fp; fp += NS * NU; c.PA = fp; fp += NS * NS; c.PB = fp; fp += NS * NU;
    c.PpB = fp; fp += NF * NU; c.PpA = fp; fp += NF * NS;
    c.arow = fp; fp += 96; c.geo = fp; fp += 36; c.red = fp; fp += 48;
    c.flag = reinterpret_cast<int*>(fp); fp += 4;
    c.Lf = fp; fp += (size_t)LP * N;
    c.Ws = fp; fp += (size_t)(NU * NS) * N;

    // ---- load the parameter vector (coalesced) ----
    {
        const float* gp = prm.P + (size_t)b * c.L.np;
        for (int e = tid; e < c.L.np; e += NT) spw[e] = gp[e];
    }
    __syncthreads();
    // ---- initial iterate from x0 ----
    {
        const float* x0 = prm.X0 + (size_t)b * c.L.nx;
        for (int e = tid; e < NS * (N + 1); e += NT) {
            const int k = e / NS, i = e % NS;
            float v;
            if (k == 0) {  // initial-condition rows of g hold exactly
                if (i < 9) v = c.sp[c.L.p_com0 + i];
                else v = c.sp[c.L.p_cur[(i - 9) / 3] + (i - 9) % 3];
            } else if (i < 9) v = x0[c.L.o_com + 3 * (N + 1) * (i / 3) + 3 * k + i % 3];
            else v = x0[c.L.o_pos[(i - 9) / 3] + 3 * k + (i - 9) % 3];
            c.S[e] = v;
            c.LAM[e] = 0.0;
        }
        for (int e = tid; e < NU * N; e += NT) {
            const int k = e / NU, m = e % NU;
            float v = 0.f;
            if (m < NF) v = x0[c.L.o_f[m / 12][(m % 12) / 3] + 3 * k + m % 3];
            else {
                const int q = m - 24, ct = q / 3, i = q % 3;
                if (qfree(c, k, q)) {
                    const float* R = c.sp + c.L.p_R[ct] + 9 * k;
                    const float lo = qlo(c, k, q), hi = qhi(c, k, q), push = 0.01f * (hi - lo);
                    for (int a = 0; a < 3; ++a)
                        v += Rm(R, a, i) * (x0[c.L.o_pos[ct] + 3 * (k + 1) + a] - c.sp[c.L.p_nom[ct] + 3 * (k + 1) + a]);
                    v = fminf(fmaxf(v, lo + push), hi - push);
                } else if (gam_of(c, ct, k) < 0.5f) v = qlo(c, k, q);
            }
            c.U[e] = v;
        }
        __syncthreads();
        for (int e = tid; e < NI * N; e += NT) {
            const int k = e / NI, i = e % NI;
            float t = 1.f, z = 0.f;
            if (row_active(c, k, i)) {
                t = -row_val(c, prm, k, i, c.U + NU * k);
                if (i < 32) t = fmaxf(t, 1e-2f);
                z = prm.mu_init / t;
            }
            c.T[e] = t; c.Z[e] = z;
        }
        __syncthreads();
    }
    int nrow = 0;
    for (int k = 0; k < N; ++k)
        for (int i = 32; i < NI; ++i) nrow += row_active(c, k, i) ? 1 : 0;
    nrow += 32 * N;

    int it = 0, status = 1, gn = 0;
    float err = 0.f, ep = 0.f, mu_cur = 0.f, es_out = 0.f;
    bool finishing = false;
    for (it = 0; it < prm.max_iter + 1; ++it) {
        if (it == prm.max_iter && !finishing) break;
        // ---- residuals of the current iterate ----
        PROF_DECL;
        float l_ep = 0.f, l_ec = 0.f, l_es = 0.f;
        double l_mu = 0.0;
        for (int e = tid; e < NS * N; e += NT) {
            const double dv = defect(c, prm, e / NS, e % NS);
            c.d[e] = (float)dv;
            l_ep = fmaxf(l_ep, fabsf((float)dv));
        }
        for (int e = tid; e < NI * N; e += NT) {
            const int k = e / NI, i = e % NI;
            if (row_active(c, k, i)) {
                const float t = c.T[e], z = c.Z[e];
                l_ep = fmaxf(l_ep, fabsf(row_val(c, prm, k, i, c.U + NU * k) + t));
                l_ec = fmaxf(l_ec, t * z);
                l_mu += (double)t * z;
            }
        }
        // stationarity (float64), one stage at a time through the closed forms
        for (int k = 0; k <= N; ++k) {
            if (k < N) stage_geo(c, prm, k, tid);
            __syncthreads();
            if (k < N && tid < NU) {
                if (tid < NF || qfree(c, k, tid - 24)) {
                    double r = Bt_vec(c, prm, k, tid, c.LAM + NS * (k + 1));
                    if (tid < NF) {
                        r += grad_sym(c, prm, k, tid);
                        const float* u = c.U + NU * k;
                        if (k > 0) r += (double)prm.D[tid % 3] * ((double)u[tid] - (double)c.U[NU * (k - 1) + tid]);
                        if (k + 1 < N) r -= (double)prm.D[tid % 3] * ((double)c.U[NU * (k + 1) + tid] - (double)u[tid]);
                        const int r0 = 4 * (tid / 3);
                        for (int f = 0; f < 4; ++f) {
                            float a[3];
                            fric_row(c, prm, k, r0 + f, a);
                            r += (double)c.Z[NI * k + r0 + f] * (double)a[tid % 3];
                        }
                    } else {
                        const int q = tid - 24;
                        r += (double)c.Z[NI * k + 32 + q] - (double)c.Z[NI * k + 38 + q];
                    }
                    l_es = fmaxf(l_es, fabsf((float)r));
                }
            } else if (k > 0 && tid >= 64 && tid < 64 + NS) {
                const int i = tid - 64;
                double r = grad_track(c, prm, k, i) - c.LAM[NS * k + i];
                if (k < N) r += At_vec(c, prm, k, i, c.LAM + NS * (k + 1));
                l_es = fmaxf(l_es, fabsf((float)r));
            }
            __syncthreads();
        }
        ep = block_max<NT>(l_ep, c.red, tid);
        const float ec = block_max<NT>(l_ec, c.red, tid);
        const float es = block_max<NT>(l_es, c.red, tid);
        mu_cur = (float)(block_sum<NT>(l_mu, c.redd, tid) / (double)nrow);
        es_out = es;
        PROF(10);
        // ---- predictor (affine scaling) ----
        bool exact = prm.exact_hessian != 0;
        int fail = riccati_backward<NT>(c, prm, tid, exact, true);
        if (fail) {
            __syncthreads();
            ++gn; exact = false;
            fail = riccati_backward<NT>(c, prm, tid, false, true);
        }
        if (fail) { status = 2; break; }
        PROF(11);
        riccati_forward<NT>(c, prm, tid, true);
        PROF(12);
        float ap, ad;
        if (finishing) {
            // last step: affine-scaling extrapolation of the central path to mu = 0 (primal only)
            step_lengths<NT>(c, tid, 0.999f, ap, ad);
            for (int e = tid; e < NS * (N + 1); e += NT) c.S[e] += ap * c.dS[e];
            for (int e = tid; e < NU * N; e += NT) c.U[e] += ap * c.dU[e];
            __syncthreads();
            ++it;
            break;
        }
        step_lengths<NT>(c, tid, 1.f, ap, ad);
        double l_aff = 0.0;
        for (int e = tid; e < NI * N; e += NT)
            if (row_active(c, e / NI, e % NI)) l_aff += (double)(c.T[e] + ap * c.dT[e]) * (double)(c.Z[e] + ad * c.dZ[e]);
        const float mu_aff = (float)(block_sum<NT>(l_aff, c.redd, tid) / (double)nrow);
        float sigma = mu_aff / mu_cur;
        sigma = sigma * sigma * sigma;
        const float mu_t = fmaxf(sigma * mu_cur, prm.mu_min);
        // ---- corrector ----
        for (int e = tid; e < NI * N; e += NT)
            c.dZ[e] = row_active(c, e / NI, e % NI) ? mu_t - c.dT[e] * c.dZ[e] : 0.f;  // complementarity target
        __syncthreads();
        PROF(13);
        riccati_delta<NT>(c, prm, tid);
        PROF(14);
        riccati_forward<NT>(c, prm, tid, false);
        PROF(15);
        step_lengths<NT>(c, tid, fmaxf(0.99f, 1.f - mu_t), ap, ad);
        // ---- costates, then the iterate ----
        if (tid < 3) c.qs[tid] = c.LAM[NS * N + 6 + tid];
        __syncthreads();
        costate_update<NT>(c, prm, tid, ap, exact);
        PROF(16);
        for (int e = tid; e < NS * (N + 1); e += NT) c.S[e] += ap * c.dS[e];
        for (int e = tid; e < NU * N; e += NT) c.U[e] += ap * c.dU[e];
        for (int e = tid; e < NI * N; e += NT) { c.T[e] += ap * c.dT[e]; c.Z[e] += ad * c.dZ[e]; }
        // ---- convergence: the Newton step itself is the error estimate.  (The stationarity residual
        // `es` of a float32-stored iterate cannot go below ~1e-3: one ulp of com_z moves its
        // gradient by 2 w_z^2 ulp ~ 5e-3; it is reported, not tested.) ----
        float l_st = 0.f;
        for (int e = tid; e < NS * (N + 1); e += NT) l_st = fmaxf(l_st, fabsf(c.dS[e]));
        for (int e = tid; e < NU * N; e += NT) l_st = fmaxf(l_st, fabsf(c.dU[e]));
        const float step = ap * block_max<NT>(l_st, c.red, tid);
        err = fmaxf(fmaxf(0.1f * step, ep), ec);
        PROF(17);
        if (err <= prm.tol) {
            status = 0;
            if (!prm.final_extrap) { ++it; break; }
            finishing = true;
        }
    }
    // ---- export x in the reference layout ----
    {
        float* x = prm.X + (size_t)b * c.L.nx;
        for (int e = tid; e < NS * (N + 1); e += NT) {
            const int k = e / NS, i = e % NS;
            if (i < 9) x[c.L.o_com + 3 * (N + 1) * (i / 3) + 3 * k + i % 3] = c.S[e];
            else x[c.L.o_pos[(i - 9) / 3] + 3 * k + (i - 9) % 3] = c.S[e];
        }
        for (int e = tid; e < NU * N; e += NT) {
            const int k = e / NU, m = e % NU;
            if (m < NF) x[c.L.o_f[m / 12][(m % 12) / 3] + 3 * k + m % 3] = c.U[e];
            else {
                const int q = m - 24, ct = q / 3, i = q % 3;
                const float v = gam_of(c, ct, k) < 0.5f ? (c.S[NS * (k + 1) + 9 + q] - c.S[NS * k + 9 + q]) / prm.dt : 0.f;
                x[c.L.o_vel[ct] + 3 * k + i] = v;
            }
        }
        if (prm.info && tid == 0) {
            float* inf = prm.info + (size_t)b * CMPC_INFO_N;
            inf[0] = (float)it; inf[1] = err; inf[2] = mu_cur; inf[3] = (float)gn; inf[4] = ep; inf[5] = (float)status;
            inf[6] = (float)(__builtin_amdgcn_s_memtime() - t_start); inf[7] = es_out;
        }
    }
}

}  // namespace

// LDS bytes the kernel needs for horizon N
extern "C" size_t cmpc_solver_lds_bytes(int N)
{
    CmpcLayout L;
    cmpc_layout_init(L, N);
    size_t dbl = (size_t)CMPC_NS * (N + 1) + NU * QULD + 40 + 40 + 30 + 16 + 40 + 30 + NI + NI + 8;
    size_t flt = ((L.np + 3) & ~3) + 2 * ((size_t)NS * (N + 1) + (size_t)NU * N + 2 * (size_t)NI * N) + (size_t)NS * N
                 + (size_t)NU * N + 2 * NXA * PLD + NU * QLD + 2 * NS * NS + 2 * NS * NU + NF * NU + NF * NS + 96 + 36 + 48 + 4
                 + (size_t)LP * N + (size_t)NU * NS * N;
    return dbl * 8 + flt * 4;
}

#ifdef CMPC_PROFILE
extern "C" int cmpc_profile_read(long long* out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(long long) * 32);
    if (reset) { long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
    return (int)e;
}
#endif

extern "C" int cmpc_launch_solver(const CmpcParams* prm, size_t lds_bytes, hipStream_t stream)
{
    constexpr int NT = 256;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cmpc_solve_kernel<NT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cmpc_solve_kernel<NT>, dim3(prm->B), dim3(NT), lds_bytes, stream, *prm);
    return (int)hipGetLastError();
}
