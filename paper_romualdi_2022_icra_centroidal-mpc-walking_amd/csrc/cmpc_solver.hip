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
// Arithmetic: float32 storage and matrix work; float64 where cancellation decides the answer:
// dynamics defects, right-hand-side / costate recursions, and the 3x3 diagonal blocks of the stage
// Hessian Quu through its Cholesky (barrier terms z/t span 1e-6..1e9 inside one corner's block
// next to cost curvature of order 10).
//
// Structure of one stage of the backward sweep (256 threads = 4 waves, ~5 barriers):
//   1. G = P [B;E], T1 = Pss A           sparse: every column of A, B has <= 3 non-zeros
//   2. Quu, Qus, Qss, right-hand sides    (Quu diag blocks in float64)
//   3. fused Cholesky + panel solve       one wave, matrix rows in registers: lanes 0-29 hold the
//      rows of Quu, lanes 30-63 hold rows of [Qus | I | qu]^T, so L^{-1}[Qus | I | qu] falls out of
//      the same rank-1 updates (v_readlane broadcasts, no LDS traffic, no barriers)
//   4. P = [Qss 0; 0 D] - W^T W           30-term dot products on 16-byte LDS reads
// The vector sweeps (forward, corrector right-hand side, costates) run on one wave without
// workgroup barriers.
#include "cmpc_device.h"

#define NS CMPC_NS
#define NF CMPC_NF
#define NQ CMPC_NQ
#define NU CMPC_NU
#define NXA CMPC_NXA
#define NI CMPC_NI
#define LP CMPC_LP
#define PLD 39     // leading dim of P (odd: column walks are conflict-free)
#define GLD 30     // G = P [B;E]  (39 x 30)
#define RLD 36     // leading dim of the row-major float panels (16-byte aligned rows)
#define NPAN 46    // panel rows: 15 (Qus^T) + 30 (I) + 1 (qu)
#define GEO 36     // floats of stage geometry: r[24] | Fc[6] | Fsum[3] | pad
#define NTRI 780   // lower-triangular entries of a 39x39

namespace {

#ifdef CMPC_PROFILE
// diagnostic build only: per-phase shader-clock sums of workgroup 0
__device__ long long g_prof[32];
__device__ float g_trace[64 * 8];  // per iteration of workgroup 0: mu, ep, ec, step, ap, ad, sigma, mu_t
#define PROF_DECL long long pt_ = __builtin_amdgcn_s_memtime()
#define PROF(slot) do { long long n_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x == 0) g_prof[slot] += n_ - pt_; pt_ = n_; } while (0)
#define PROF2_DECL long long pt2_ = __builtin_amdgcn_s_memtime()
#define PROF2(slot) do { long long n_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x == 0) g_prof[slot] += n_ - pt2_; pt2_ = n_; } while (0)
#else
#define PROF_DECL
#define PROF(slot)
#define PROF2_DECL
#define PROF2(slot)
#endif

struct Ctx {
    CmpcIdx L;
    int N;
    const float* sp;     // parameter vector in LDS
    float *S, *U, *T, *Z;
    double* LAM;
    float *dS, *dU, *dT, *dZ, *d;
    float *Lf, *Ws, *lqs;  // per-stage factors: L^{-1} (packed lower), Ws = L^{-1} Qus, lq = L^{-1} qu
    float *geoA;           // N x GEO
    float *P0, *P1, *G, *T1, *QuuF, *Pan, *Bval, *Aval, *arow, *ybuf, *fpv, *fpn;
    int *Brow, *Arow, *qmask;
    unsigned short* tri;
    double *QuuD, *pv, *pn, *qs, *Pd, *sig, *gco, *redd;
    float* red;
    int* flag;
};

__device__ inline void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ inline float gam_of(const Ctx& c, int ct, int k) { return c.sp[c.L.pGam(ct) + k]; }
// reference stores vec(R) column-major: R(r,cc) = R[3*cc + r]
__device__ inline float Rm(const float* R, int r, int cc) { return R[3 * cc + r]; }

// a landing-offset component is a free variable in a swing stage whose box row is not an equality;
// the 6-bit mask per stage is computed once per solve (the schedule is data)
__device__ inline bool qfree_compute(const Ctx& c, int k, int m)
{
    const int ct = m / 3, i = m % 3;
    const float lo = c.sp[c.L.pLo(ct) + 3 * k + i], hi = c.sp[c.L.pUp(ct) + 3 * k + i];
    return gam_of(c, ct, k) < 0.5f && (hi - lo) > 1e-9f;
}
__device__ inline bool qfree(const Ctx& c, int k, int m) { return (c.qmask[k] >> m) & 1; }
__device__ inline float qlo(const Ctx& c, int k, int m) { return c.sp[c.L.pLo(m / 3) + 3 * k + m % 3]; }
__device__ inline float qhi(const Ctx& c, int k, int m) { return c.sp[c.L.pUp(m / 3) + 3 * k + m % 3]; }

// friction row i (0..31) of stage k: a = R (sx, sy, -mu)^T, acting on corner i/4
__device__ inline void fric_row(const Ctx& c, const CmpcConsts& prm, int k, int i, float& a0, float& a1, float& a2)
{
    const int ct = i >> 4, face = i & 3;
    const float* R = c.sp + c.L.pR(ct) + 9 * k;
    const float sx = (face == 0 || face == 3) ? 1.f : -1.f;
    const float sy = (face < 2) ? 1.f : -1.f;
    a0 = sx * Rm(R, 0, 0) + sy * Rm(R, 0, 1) - prm.mu_fr * Rm(R, 0, 2);
    a1 = sx * Rm(R, 1, 0) + sy * Rm(R, 1, 1) - prm.mu_fr * Rm(R, 1, 2);
    a2 = sx * Rm(R, 2, 0) + sy * Rm(R, 2, 1) - prm.mu_fr * Rm(R, 2, 2);
}

__device__ inline bool row_active(const Ctx& c, int k, int i) { return i < 32 ? true : qfree(c, k, (i - 32) % 6); }

// a_i^T u - b_i   (<= 0 feasible)
__device__ inline float row_val(const Ctx& c, const CmpcConsts& prm, int k, int i, const float* u)
{
    if (i < 32) {
        float a0, a1, a2;
        fric_row(c, prm, k, i, a0, a1, a2);
        const float* f = u + 3 * (i >> 2);
        return a0 * f[0] + a1 * f[1] + a2 * f[2];
    }
    if (i < 38) return u[24 + i - 32] - qhi(c, k, i - 32);
    return qlo(c, k, i - 38) - u[24 + i - 38];
}
__device__ inline float row_dot(const Ctx& c, const CmpcConsts& prm, int k, int i, const float* du)
{
    if (i < 32) {
        float a0, a1, a2;
        fric_row(c, prm, k, i, a0, a1, a2);
        const float* f = du + 3 * (i >> 2);
        return a0 * f[0] + a1 * f[1] + a2 * f[2];
    }
    if (i < 38) return du[24 + i - 32];
    return -du[24 + i - 38];
}

__device__ inline float qdiag(const CmpcConsts& prm, int k, int i)
{
    if (i == 0) return 2.f * prm.w_com0;
    if (i == 1) return 2.f * prm.w_com1;
    if (i == 2) return prm.wz2[k];
    if (i < 6) return 0.f;
    if (i < 9) return 2.f * prm.w_h;
    return 2.f * prm.w_pos;
}

// gradient of the tracking cost w.r.t. component i of s_k
__device__ inline double grad_track(const Ctx& c, const CmpcConsts& prm, int k, int i)
{
    const float* s = c.S + NS * k;
    if (i < 3) return (double)qdiag(prm, k, i) * ((double)s[i] - (double)c.sp[c.L.pComref() + 3 * k + i]);
    if (i < 6) return 0.0;
    if (i < 9) return 2.0 * prm.w_h * ((double)s[i] - (double)c.sp[c.L.pHref() + 3 * k + i - 6]);
    const int ct = (i - 9) / 3, a = (i - 9) % 3;
    return 2.0 * prm.w_pos * ((double)s[i] - (double)c.sp[c.L.pNom(ct) + 3 * k + a]);
}

// gradient of the force-symmetry cost w.r.t. force component m (0..23) of stage k
__device__ inline double grad_sym(const Ctx& c, const CmpcConsts& prm, int k, int m)
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

// ---- geometry of every stage for the current iterate: r_cj = R c_j + pos_c - com (8x3), Fc (2x3,
// plain corner sums), Fsum (gam-weighted sum) ----
template <int NT>
__device__ inline void all_geo(const Ctx& c, const CmpcConsts& prm, int tid)
{
    for (int e = tid; e < c.N * 33; e += NT) {
        const int k = e / 33, t = e % 33;
        const float* s = c.S + NS * k;
        const float* u = c.U + NU * k;
        float v;
        if (t < 24) {
            const int ct = t / 12, j = (t % 12) / 3, i = t % 3;
            const float* R = c.sp + c.L.pR(ct) + 9 * k;
            const float* cn = prm.corners + 12 * ct + 3 * j;
            v = Rm(R, i, 0) * cn[0] + Rm(R, i, 1) * cn[1] + Rm(R, i, 2) * cn[2] + s[9 + 3 * ct + i] - s[i];
        } else if (t < 30) {
            const int ct = (t - 24) / 3, i = (t - 24) % 3;
            const float* f = u + 12 * ct;
            v = f[i] + f[3 + i] + f[6 + i] + f[9 + i];
        } else {
            const int i = t - 30;
            v = gam_of(c, 0, k) * (u[i] + u[3 + i] + u[6 + i] + u[9 + i])
                + gam_of(c, 1, k) * (u[12 + i] + u[15 + i] + u[18 + i] + u[21 + i]);
        }
        c.geoA[GEO * k + t] = v;
    }
}

// dynamics defect component i of stage k in float64: phi_k(s_k,u_k)[i] - s_{k+1}[i]
__device__ inline double defect(const Ctx& c, const CmpcConsts& prm, int k, int i)
{
    const float* s = c.S + NS * k;
    const float* u = c.U + NU * k;
    const float* sn = c.S + NS * (k + 1);
    const double dt = prm.dt;
    if (i < 3) return (double)s[i] + dt * (double)s[3 + i] - (double)sn[i];
    if (i < 6) {
        const int a = i - 3;
        double acc = (double)c.sp[c.L.pFext() + 3 * k + a] - (a == 2 ? (double)prm.grav : 0.0);
        for (int ct = 0; ct < 2; ++ct) {
            const float* f = u + 12 * ct;
            acc += (double)gam_of(c, ct, k) * ((double)f[a] + (double)f[3 + a] + (double)f[6 + a] + (double)f[9 + a]);
        }
        return (double)s[i] + dt * acc - (double)sn[i];
    }
    if (i < 9) {
        const int a = i - 6, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
        double tor = (double)c.sp[c.L.pText() + 3 * k + a];
        for (int ct = 0; ct < 2; ++ct) {
            const float* R = c.sp + c.L.pR(ct) + 9 * k;
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
        const float* R = c.sp + c.L.pR(ct) + 9 * k;
        const double gam = gam_of(c, ct, k);
        double land = (double)c.sp[c.L.pNom(ct) + 3 * (k + 1) + a];
        for (int m = 0; m < 3; ++m) land += (double)Rm(R, a, m) * (double)u[24 + 3 * ct + m];
        return gam * (double)s[i] + (1.0 - gam) * land - (double)sn[i];
    }
}

// (B_k^T v)[m], closed form.  T = float or double
template <typename T>
__device__ inline T Bt_vec(const Ctx& c, const CmpcConsts& prm, int k, int m, const T* v)
{
    const float* geo = c.geoA + GEO * k;
    if (m < 24) {
        const int ct = m / 12, i = m % 3;
        const float* r = geo + 3 * (m / 3);
        const int a1 = (i + 1) % 3, a2 = (i + 2) % 3;
        return (T)prm.dt * (T)gam_of(c, ct, k) * (v[3 + i] + v[6 + a1] * (T)r[a2] - v[6 + a2] * (T)r[a1]);
    }
    const int q = m - 24, ct = q / 3, a = q % 3;
    if (!qfree(c, k, q)) return (T)0;
    const float* R = c.sp + c.L.pR(ct) + 9 * k;
    const T g1 = (T)1 - (T)gam_of(c, ct, k);
    return g1 * ((T)Rm(R, 0, a) * v[9 + 3 * ct] + (T)Rm(R, 1, a) * v[10 + 3 * ct] + (T)Rm(R, 2, a) * v[11 + 3 * ct]);
}

// (A_k^T v)[i], closed form
template <typename T>
__device__ inline T At_vec(const Ctx& c, const CmpcConsts& prm, int k, int i, const T* v)
{
    const float* geo = c.geoA + GEO * k;
    const T dt = (T)prm.dt;
    if (i < 3) {
        const int a1 = (i + 1) % 3, a2 = (i + 2) % 3;
        const float* Fs = geo + 30;
        return v[i] + dt * (v[6 + a1] * (T)Fs[a2] - v[6 + a2] * (T)Fs[a1]);
    }
    if (i < 6) return dt * v[i - 3] + v[i];
    if (i < 9) return v[i];
    const int ct = (i - 9) / 3, a = (i - 9) % 3, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
    const float* Fc = geo + 24 + 3 * ct;
    const T gam = (T)gam_of(c, ct, k);
    return gam * (v[i] + dt * ((T)Fc[a1] * v[6 + a2] - (T)Fc[a2] * v[6 + a1]));
}

// (A_k ds + B_k du)[i]  (float, closed form)
__device__ inline float AB_step(const Ctx& c, const CmpcConsts& prm, int k, int i, const float* ds, const float* du)
{
    const float* geo = c.geoA + GEO * k;
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
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* r = geo + 12 * ct + 3 * j;
                const float* df = du + 12 * ct + 3 * j;
                const float* f = c.U + NU * k + 12 * ct + 3 * j;
                t += r[a1] * df[a2] - r[a2] * df[a1] + e1 * f[a2] - e2 * f[a1];
            }
            tor += gam_of(c, ct, k) * t;
        }
        return ds[i] + dt * tor;
    }
    const int ct = (i - 9) / 3, a = (i - 9) % 3;
    const float* R = c.sp + c.L.pR(ct) + 9 * k;
    const float gam = gam_of(c, ct, k);
    float land = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m)
        if (qfree(c, k, 3 * ct + m)) land += Rm(R, a, m) * du[24 + 3 * ct + m];
    return gam * ds[i] + (1.f - gam) * land;
}

__device__ inline int lpk(int i, int j) { return i * (i + 1) / 2 + j; }  // packed lower (i >= j)

__device__ inline float readlane_f(float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }
__device__ inline double readlane_d(double x, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// ---- fused Cholesky + panel solve, one wave, rows in registers ----
// lanes 0..29: row `lane` of the symmetric matrix (v[c], c <= lane, float; its own 3x3 diagonal
// block in dd[], float64).  Lanes >= 30: a row of the panel [Qus | I | qu]^T (30 floats).
// On exit v[] holds the row of L (lanes < 30) or of (L^{-1} [Qus | I | qu])^T.  Returns true if a
// pivot was not positive (uniform across the wave).
__device__ inline bool chol_solve_fused(float (&v)[NU], double (&dd)[3], int lane, int fixedmask)
{
    // Lanes < 30 enter with v[c] = Quu[lane][c] for c outside their own 3x3 diagonal block and 0 inside it; the
    // diagonal block itself is in dd (float64).  Updates of a diagonal block by earlier columns are moderate
    // numbers and accumulate in those zeroed float slots; they are folded into dd when the block becomes the
    // pivot block.  Inside the pivot block everything is float64 (partners by v_readlane of compile-time lanes).
    bool bad = false;
    const int myblk = lane / 3;
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        const int b = j / 3, jm = j % 3;
        const bool inblk = (myblk == b);
        if (jm == 0 && inblk) {
            dd[0] += (double)v[3 * b];
            dd[1] += (double)v[3 * b + 1];
            dd[2] += (double)v[3 * b + 2];
        }
        if (j >= NF && ((fixedmask >> (j - NF)) & 1)) continue;  // identity row/column (fixed q): nothing to do
        double piv = readlane_d(dd[jm], j);
        if (!(piv > 0.0)) { bad = true; piv = 1.0; }
        double rinv = (double)rsqrtf((float)piv);
        rinv = rinv * (1.5 - 0.5 * piv * rinv * rinv);  // one Newton step in float64: ~1e-14 relative
        const float rinvf = (float)rinv;
        const double ld = inblk ? dd[jm] * rinv : (double)(v[j] * rinvf);
        const float lf = (float)ld;
        v[j] = lf;
        if (jm < 2) {
            const double dA = readlane_d(ld, j + 1);
            if (inblk && lane > j) dd[jm + 1] -= ld * dA;
            if (jm < 1) {
                const double dB = readlane_d(ld, j + 2);
                if (inblk && lane > j + 1) dd[2] -= ld * dB;
            }
        }
        // everything else (float32): v[c] -= l_ij * l_cj
#pragma unroll
        for (int cc = j + 1; cc < NU; ++cc) v[cc] -= lf * readlane_f(lf, cc);
    }
    return bad;
}

// phase 3 of a backward stage, kept out of line so that its ~40 VGPRs of matrix rows and its
// stream of v_readlane broadcasts get a register allocation of their own
__device__ inline void stage_factor(const float* QuuF, const double* QuuD, float* Pan, float* Lf, float* Ws, float* lq,
                                          float D0, float D1, float D2, int* flag, int tid, int fixedmask)
{
    const int lane = tid & 63, wv = tid >> 6;
    const int prow = lane - 30 + 34 * wv;  // panel row of lanes >= 30
    const bool isL = lane < NU;
    const bool active = isL || prow < NPAN;
    float v[NU];
    double dd[3] = {0.0, 0.0, 0.0};
    const bool idrow = !isL && prow >= NS && prow < NS + NU;
    {
        const float* src = isL ? QuuF + lane * RLD : Pan + ((active && !idrow) ? prow : 0) * RLD;
#pragma unroll
        for (int q4 = 0; q4 < 8; ++q4) {
            const float4 t4 = *reinterpret_cast<const float4*>(src + 4 * q4);
            if (4 * q4 + 0 < NU) v[4 * q4 + 0] = t4.x;
            if (4 * q4 + 1 < NU) v[4 * q4 + 1] = t4.y;
            if (4 * q4 + 2 < NU) v[4 * q4 + 2] = t4.z;
            if (4 * q4 + 3 < NU) v[4 * q4 + 3] = t4.w;
        }
        if (idrow) {
#pragma unroll
            for (int cc = 0; cc < NU; ++cc) v[cc] = (cc == prow - NS) ? 1.f : 0.f;
        }
        if (isL) {
            dd[0] = QuuD[9 * (lane / 3) + 3 * (lane % 3) + 0];
            dd[1] = QuuD[9 * (lane / 3) + 3 * (lane % 3) + 1];
            dd[2] = QuuD[9 * (lane / 3) + 3 * (lane % 3) + 2];
        }
    }
    const bool bad = chol_solve_fused(v, dd, lane, fixedmask);
    if (bad && tid == 0) *flag = 1;
    if (!isL && active) {
        if (prow < NS) {
#pragma unroll
            for (int a = 0; a < NU; ++a) { Ws[a * NS + prow] = v[a]; Pan[prow * RLD + a] = v[a]; }
        } else if (prow < NS + NU) {
            const int m = prow - NS;
            const float dm = (m % 3 == 0) ? D0 : ((m % 3 == 1) ? D1 : D2);
            const float sc = m < NF ? -dm : 0.f;
#pragma unroll
            for (int a = 0; a < NU; ++a)
                if (a >= m) { Lf[lpk(a, m)] = v[a]; Pan[prow * RLD + a] = sc * v[a]; }
        } else {
#pragma unroll
            for (int a = 0; a < NU; ++a) { lq[a] = v[a]; Pan[prow * RLD + a] = v[a]; }
        }
    }
}

// ---- Riccati backward sweep (matrices + right-hand side of the affine step).
// Returns (uniformly) 0 ok, 1 non-positive pivot. ----
template <int NT>
__device__ int riccati_backward(const Ctx& c, const CmpcConsts& prm, int tid, bool use_exact, float reg, float cmu)
{
    const int N = c.N;
    float* Pcur = c.P0;  // value function of stage k+1
    float* Pnew = c.P1;
    bool havep = false;
    for (int e = tid; e < NXA * PLD; e += NT) Pcur[e] = 0.f;
    if (tid == 0) *c.flag = 0;
    __syncthreads();
    if (tid < NS) {
        Pcur[tid * PLD + tid] = qdiag(prm, N, tid);
        c.pv[tid] = grad_track(c, prm, N, tid);
    } else if (tid < NXA) c.pv[tid] = 0.0;
    __syncthreads();

    for (int k = N - 1; k >= 0; --k) {
        const bool pk = k > 0;
        const float* u = c.U + NU * k;
        const float* geo = c.geoA + GEO * k;
        PROF_DECL;
        // ---- phase 0: column descriptors of A and B, barrier coefficients ----
        if (tid < NU) {
            int r0, r1, r2;
            float v0, v1, v2;
            if (tid < NF) {
                const int ct = tid / 12, a = tid % 3, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
                const float g = prm.dt * gam_of(c, ct, k);
                const float* r = geo + 3 * (tid / 3);
                r0 = 3 + a; v0 = g;
                r1 = 6 + a1; v1 = g * r[a2];
                r2 = 6 + a2; v2 = -g * r[a1];
            } else {
                const int q = tid - 24, ct = q / 3, m = q % 3;
                const float* R = c.sp + c.L.pR(ct) + 9 * k;
                const float g1 = qfree(c, k, q) ? 1.f - gam_of(c, ct, k) : 0.f;
                r0 = 9 + 3 * ct; r1 = r0 + 1; r2 = r0 + 2;
                v0 = g1 * Rm(R, 0, m); v1 = g1 * Rm(R, 1, m); v2 = g1 * Rm(R, 2, m);
            }
            c.Brow[3 * tid] = r0; c.Brow[3 * tid + 1] = r1; c.Brow[3 * tid + 2] = r2;
            c.Bval[3 * tid] = v0; c.Bval[3 * tid + 1] = v1; c.Bval[3 * tid + 2] = v2;
        } else if (tid >= 32 && tid < 32 + NS) {
            const int j = tid - 32;
            int r0 = j, r1 = j, r2 = j;
            float v0 = 1.f, v1 = 0.f, v2 = 0.f;
            if (j < 3) {
                const float* Fs = geo + 30;
                r1 = 6 + (j + 1) % 3; v1 = prm.dt * Fs[(j + 2) % 3];
                r2 = 6 + (j + 2) % 3; v2 = -prm.dt * Fs[(j + 1) % 3];
            } else if (j < 6) {
                r0 = j - 3; v0 = prm.dt; r1 = j; v1 = 1.f;
            } else if (j >= 9) {
                const int ct = (j - 9) / 3, cc = (j - 9) % 3;
                const float gam = gam_of(c, ct, k);
                const float* Fc = geo + 24 + 3 * ct;
                v0 = gam;
                r1 = 6 + (cc + 1) % 3; v1 = -prm.dt * gam * Fc[(cc + 2) % 3];
                r2 = 6 + (cc + 2) % 3; v2 = prm.dt * gam * Fc[(cc + 1) % 3];
            }
            c.Arow[3 * j] = r0; c.Arow[3 * j + 1] = r1; c.Arow[3 * j + 2] = r2;
            c.Aval[3 * j] = v0; c.Aval[3 * j + 1] = v1; c.Aval[3 * j + 2] = v2;
        } else if (tid >= 64 && tid < 64 + NI) {
            const int i = tid - 64;
            double sg = 0.0, gc = 0.0;
            if (row_active(c, k, i)) {
                const double t = c.T[NI * k + i], z = c.Z[NI * k + i];
                const double r = (double)row_val(c, prm, k, i, u) + t;
                sg = z / t;
                gc = (double)cmu / t + sg * r;  // complementarity target cmu (0: affine-scaling predictor)
            }
            c.sig[i] = sg; c.gco[i] = gc;
            if (i < 32) {
                float a0, a1, a2;
                fric_row(c, prm, k, i, a0, a1, a2);
                c.arow[3 * i] = a0; c.arow[3 * i + 1] = a1; c.arow[3 * i + 2] = a2;
            }
        } else if (tid >= 112 && tid < 121) {
            // exact-Hessian block dt [lam_h]x (zero for the Gauss-Newton Hessian), row-major 3x3
            const int a = (tid - 112) / 3, b = (tid - 112) % 3;
            float sv = 0.f;
            if (use_exact && a != b) {
                const float lv = (float)c.LAM[NS * (k + 1) + 6 + (3 - a - b)];
                sv = prm.dt * (((b - a + 3) % 3 == 1) ? -lv : lv);
            }
            c.arow[96 + tid - 112] = sv;
        } else if (tid >= 128 && tid < 128 + NXA) {
            // Pd = P [d; 0] + pv  (float64)
            const int r = tid - 128;
            double acc = c.pv[r];
            if (r < NS || havep) {
#pragma unroll
                for (int a = 0; a < NS; ++a) acc += (double)Pcur[r * PLD + a] * (double)c.d[NS * k + a];
            }
            c.Pd[r] = acc;
        }
        __syncthreads();
        PROF(0);
        // ---- phase 1: G = P [B;E] (39 x 30), T1 = Pss A (15 x 15): thread <-> column, its three non-zeros in
        // registers, rows strided over 8 (G) / all 15 (T1) ----
        {
            const int nrow = havep ? NXA : NS;
            if (tid < 8 * NU) {
                const int i = tid % NU, r0 = tid / NU;
                const int b0 = c.Brow[3 * i], b1 = c.Brow[3 * i + 1], b2 = c.Brow[3 * i + 2];
                const float w0 = c.Bval[3 * i], w1 = c.Bval[3 * i + 1], w2 = c.Bval[3 * i + 2];
                const bool addp = havep && i < NF;
#pragma unroll
                for (int rr = 0; rr < 5; ++rr) {
                    const int r = r0 + 8 * rr;
                    if (r < nrow) {
                        const float* Pr = Pcur + r * PLD;
                        float v = Pr[b0] * w0 + Pr[b1] * w1 + Pr[b2] * w2;
                        if (addp) v += Pr[NS + i];
                        c.G[r * GLD + i] = v;
                    }
                }
            }
            if (tid < NS * NS) {
                const int i = tid / NS, j = tid % NS;
                const float* Pr = Pcur + i * PLD;
                c.T1[tid] = Pr[c.Arow[3 * j]] * c.Aval[3 * j] + Pr[c.Arow[3 * j + 1]] * c.Aval[3 * j + 1] + Pr[c.Arow[3 * j + 2]] * c.Aval[3 * j + 2];
            }
        }
        __syncthreads();
        PROF(1);
        // ---- phase 2: Quu (lower triangle; float, its 3x3 diagonal blocks in float64), panel [Qus | I | qu]^T,
        // Qss, qs ----
        for (int e = tid; e < LP; e += NT) {
            const unsigned short ij = c.tri[e];
            const int i = ij >> 8, j = ij & 255;
            const int b0 = c.Brow[3 * i], b1 = c.Brow[3 * i + 1], b2 = c.Brow[3 * i + 2];
            // [B;E]^T P [B;E]
            float vf = c.Bval[3 * i] * c.G[b0 * GLD + j] + c.Bval[3 * i + 1] * c.G[b1 * GLD + j] + c.Bval[3 * i + 2] * c.G[b2 * GLD + j];
            if (havep && i < NF) vf += c.G[(NS + i) * GLD + j];
            if (i / 3 == j / 3) {  // diagonal block: cost + barrier terms in float64
                double v = (double)vf;
                if (i < NF) {
                    if ((i % 3) == (j % 3)) {  // same corner, same axis => i == j here
                        const double gam = gam_of(c, i / 12, k);
                        v += 2.0 * prm.w_sym * (1.0 - 0.25 * gam * (2.0 - gam));
                        if (pk) v += (double)prm.D[i % 3];
                    }
                    const int r0 = 4 * (i / 3);
#pragma unroll
                    for (int f = 0; f < 4; ++f)
                        v += c.sig[r0 + f] * (double)c.arow[3 * (r0 + f) + i % 3] * (double)c.arow[3 * (r0 + f) + j % 3];
                    if (i == j) v += (double)reg;
                } else if (i == j) {
                    const bool fr = qfree(c, k, i - 24);
                    v = fr ? v + c.sig[32 + i - 24] + c.sig[38 + i - 24] + (double)reg : 1.0;  // fixed q: exact identity row
                }
                c.QuuD[9 * (i / 3) + 3 * (i % 3) + j % 3] = v;
                vf = 0.f;  // the float copy of a diagonal block collects the updates by earlier columns
            } else if (i < NF && (i / 12) == (j / 12) && (i % 3) == (j % 3)) {
                // another corner of the same foot, same axis: symmetry-cost coupling
                const float gam = gam_of(c, i / 12, k);
                vf -= 2.f * prm.w_sym * 0.25f * gam * (2.f - gam);
            }
            c.QuuF[i * RLD + j] = vf;
        }
        for (int e = tid; e < NS * NU; e += NT) {  // identity rows of the panel are made in registers (phase 3)
            const int r = e / NU, i = e % NU;  // panel row r (column of Qus), entry i
            float v;
            {
                const int j = r;
                v = c.G[c.Arow[3 * j] * GLD + i] * c.Aval[3 * j] + c.G[c.Arow[3 * j + 1] * GLD + i] * c.Aval[3 * j + 1]
                    + c.G[c.Arow[3 * j + 2] * GLD + i] * c.Aval[3 * j + 2];
                if (i < NF) {
                    // S[f_cj, pos_c] = gam Sx ; S[f_cj, com] = -gam Sx   (Sx = dt [lam_h]x, zero diagonal)
                    const int ct = i / 12, a = i % 3;
                    const int b = j < 3 ? j : j - 9 - 3 * ct;
                    const float sgn = j < 3 ? -1.f : ((b >= 0 && b < 3) ? 1.f : 0.f);
                    v += sgn * gam_of(c, ct, k) * c.arow[96 + 3 * a + ((b >= 0 && b < 3) ? b : 0)];
                }
            }
            c.Pan[r * RLD + i] = v;
        }
        if (tid >= 192 && tid < 192 + NU) {  // qu on wave 3 (float64 -> float: it vanishes at convergence, so float keeps its relative accuracy)
            const int iq = tid - 192;
            double g;
            if (iq < NF) {
                g = grad_sym(c, prm, k, iq);
                const int r0 = 4 * (iq / 3);
#pragma unroll
                for (int f = 0; f < 4; ++f) g += c.gco[r0 + f] * (double)c.arow[3 * (r0 + f) + iq % 3];
                if (pk) g += (double)prm.D[iq % 3] * ((double)u[iq] - (double)c.U[NU * (k - 1) + iq]);
                if (havep) g += c.Pd[NS + iq];
            } else {
                const int q = iq - 24;
                g = qfree(c, k, q) ? c.gco[32 + q] - c.gco[38 + q] : 0.0;
            }
            g += Bt_vec<double>(c, prm, k, iq, c.Pd);
            c.Pan[(NPAN - 1) * RLD + iq] = (float)g;
        } else if (tid >= 64 && tid < 64 + NS) {
            const int i = tid - 64;
            c.qs[i] = grad_track(c, prm, k, i) + At_vec<double>(c, prm, k, i, c.Pd);
        }
        for (int e = tid + 128; e < 128 + NS * NS; e += NT) {  // Qss = A^T T1 + Q
            const int e2 = e - 128, i = e2 / NS, j = e2 % NS;
            float v = (i == j) ? qdiag(prm, k, i) : 0.f;
            v += c.Aval[3 * i] * c.T1[c.Arow[3 * i] * NS + j] + c.Aval[3 * i + 1] * c.T1[c.Arow[3 * i + 1] * NS + j]
                 + c.Aval[3 * i + 2] * c.T1[c.Arow[3 * i + 2] * NS + j];
            Pnew[i * PLD + j] = v;
        }
        __syncthreads();
        PROF(2);
        // ---- phase 3: fused Cholesky + panel solve (waves 0 and 1; each repeats the factorisation) ----
        if (tid < 128) {
            const int fixedmask = (~c.qmask[k]) & 63;
            stage_factor(c.QuuF, c.QuuD, c.Pan, c.Lf + (size_t)LP * k, c.Ws + (size_t)(NU * NS) * k, c.lqs + NU * k,
                         prm.D[0], prm.D[1], prm.D[2], c.flag, tid, fixedmask);
        }
        __syncthreads();
        PROF(3);
        if (*c.flag) return 1;
        // ---- phase 4: P = [Qss 0; 0 D] - W^T W  (rows of the panel are W^T rows; p rows pre-scaled by -D) ----
        {
            // 2x2 output tiles: thread <-> tile (bi, bj), bj <= bi, of the 39x39 (or 15x15) lower triangle
            const int nb = pk ? 20 : 8;
            if (tid < nb * (nb + 1) / 2) {
                const unsigned short t = c.tri[tid];
                const int bi = t >> 8, bj = t & 255;
                const int i0 = 2 * bi, j0 = 2 * bj, ncol = pk ? NXA : NS;
                const float4* ri0 = reinterpret_cast<const float4*>(c.Pan + i0 * RLD);
                const float4* ri1 = reinterpret_cast<const float4*>(c.Pan + (i0 + 1) * RLD);
                const float4* rj0 = reinterpret_cast<const float4*>(c.Pan + j0 * RLD);
                const float4* rj1 = reinterpret_cast<const float4*>(c.Pan + (j0 + 1) * RLD);
                float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
#pragma unroll
                for (int q4 = 0; q4 < 8; ++q4) {
                    const float4 x0 = ri0[q4], x1 = ri1[q4], y0 = rj0[q4], y1 = rj1[q4];
                    a00 += x0.x * y0.x + x0.y * y0.y; a01 += x0.x * y1.x + x0.y * y1.y;
                    a10 += x1.x * y0.x + x1.y * y0.y; a11 += x1.x * y1.x + x1.y * y1.y;
                    if (q4 < 7) {  // columns 30, 31 are padding
                        a00 += x0.z * y0.z + x0.w * y0.w; a01 += x0.z * y1.z + x0.w * y1.w;
                        a10 += x1.z * y0.z + x1.w * y0.w; a11 += x1.z * y1.z + x1.w * y1.w;
                    }
                }
                auto put = [&](int i, int j, float acc) {
                    if (j > i || i >= ncol) return;
                    float base = 0.f;
                    if (i < NS) base = Pnew[i * PLD + j];
                    else if (i == j) base = prm.D[(i - NS) % 3];
                    const float r = base - acc;
                    Pnew[i * PLD + j] = r;
                    Pnew[j * PLD + i] = r;
                };
                put(i0, j0, a00); put(i0, j0 + 1, a01); put(i0 + 1, j0, a10); put(i0 + 1, j0 + 1, a11);
            }
            // gradient of the value function (float64)
            const int ncol = pk ? NXA : NS;
            if (tid >= 216 && tid < 216 + NXA) {  // threads beyond the 210 tile owners
                const int i = tid - 216;
                double v = 0.0;
                if (i < ncol) {
                    // W^T lq: lq vanishes at convergence, so the dot product itself is fine in float32
                    const float4* ri = reinterpret_cast<const float4*>(c.Pan + i * RLD);
                    const float4* rl = reinterpret_cast<const float4*>(c.Pan + (NPAN - 1) * RLD);
                    float s0 = 0.f, s1 = 0.f;
#pragma unroll
                    for (int q4 = 0; q4 < 8; ++q4) {
                        const float4 x = ri[q4], y = rl[q4];
                        s0 += x.x * y.x + x.y * y.y;
                        if (q4 < 7) s1 += x.z * y.z + x.w * y.w;
                    }
                    v = (double)(s0 + s1);
                    if (i < NS) v = c.qs[i] - v;
                    else {
                        const int m = i - NS;
                        v = -(double)prm.D[m % 3] * ((double)u[m] - (double)c.U[NU * (k - 1) + m]) - v;
                    }
                }
                c.pn[i] = v;
            }
        }
        __syncthreads();
        PROF(4);
        if (tid < NXA) c.pv[tid] = c.pn[tid];
        {
            float* t = Pcur; Pcur = Pnew; Pnew = t;
        }
        havep = pk;
        __syncthreads();
    }
    return 0;
}

// ---- forward sweep on wave 0: dS, dU.  All threads then compute dT, dZ (dZ holds the per-row
// complementarity target on entry unless affine). ----
template <int NT>
__device__ void riccati_forward(const Ctx& c, const CmpcConsts& prm, int tid, bool affine)
{
    const int N = c.N;
    if (tid < 64) {
        // two lanes per row (lane i and lane i+32 each sum half of the terms, then one __shfl_xor)
        const int half = tid >> 5, i = (tid & 31) < NU ? (tid & 31) : 0;
        const bool row = (tid & 31) < NU;
        const float D0 = prm.D[0], D1 = prm.D[1], D2 = prm.D[2];
        if (tid < NS) c.dS[tid] = 0.f;
        wave_lds_sync();
        PROF2_DECL;
        for (int k = 0; k < N; ++k) {
            const float* Lf = c.Lf + (size_t)LP * k;
            const float* Ws = c.Ws + (size_t)(NU * NS) * k;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            PROF2(20);
            if (half == 0) {  // lq + Ws ds
                s0 = c.lqs[NU * k + i];
#pragma unroll
                for (int a = 0; a < NS; a += 3) {
                    s0 += Ws[i * NS + a] * c.dS[NS * k + a];
                    s1 += Ws[i * NS + a + 1] * c.dS[NS * k + a + 1];
                    s2 += Ws[i * NS + a + 2] * c.dS[NS * k + a + 2];
                }
            } else if (k > 0) {  // Wp dp = -Linv[:, :24] D dp  (loads past the row end are masked, not skipped)
                const float* dp = c.dU + NU * (k - 1);
                const float* Li = Lf + lpk(i, 0);
#pragma unroll
                for (int a = 0; a < NF; a += 3) {
                    const float l0 = Li[a], l1 = Li[a + 1], l2 = Li[a + 2];
                    s0 -= (a <= i ? l0 : 0.f) * D0 * dp[a];
                    s1 -= (a + 1 <= i ? l1 : 0.f) * D1 * dp[a + 1];
                    s2 -= (a + 2 <= i ? l2 : 0.f) * D2 * dp[a + 2];
                }
            }
            float v = s0 + s1 + s2;
            v += __shfl_xor(v, 32);
            if (tid < NU) c.ybuf[tid] = -v;
            wave_lds_sync();
            PROF2(21);
            {   // du = Linv^T y : half 0 sums a = 0..14, half 1 a = 15..29
                float t0 = 0.f, t1 = 0.f, t2 = 0.f;
                const int a0 = 15 * half;
#pragma unroll
                for (int aa = 0; aa < 15; aa += 3) {
                    const int a = a0 + aa;
                    const float l0 = Lf[lpk(a, 0) + (i <= a ? i : 0)], l1 = Lf[lpk(a + 1, 0) + (i <= a + 1 ? i : 0)],
                                l2 = Lf[lpk(a + 2, 0) + (i <= a + 2 ? i : 0)];
                    t0 += (a >= i ? l0 : 0.f) * c.ybuf[a];
                    t1 += (a + 1 >= i ? l1 : 0.f) * c.ybuf[a + 1];
                    t2 += (a + 2 >= i ? l2 : 0.f) * c.ybuf[a + 2];
                }
                float w = t0 + t1 + t2;
                w += __shfl_xor(w, 32);
                if (tid < NU) c.dU[NU * k + tid] = w;
            }
            wave_lds_sync();
            PROF2(22);
            if (tid < NS) c.dS[NS * (k + 1) + tid] = AB_step(c, prm, k, tid, c.dS + NS * k, c.dU + NU * k) + c.d[NS * k + tid];
            wave_lds_sync();
            PROF2(23);
            (void)row;
        }
    }
    __syncthreads();
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

// ---- corrector right-hand side on wave 0: the row coefficients change by CMU/t (dZ holds CMU);
// updates lq in place through the stored factors ----
__device__ void riccati_delta(const Ctx& c, const CmpcConsts& prm, int tid)
{
    const int N = c.N;
    if (tid < 64) {
        bool havep = false;
        if (tid < NXA) c.fpv[tid] = 0.f;
        wave_lds_sync();
        PROF2_DECL;
        for (int k = N - 1; k >= 0; --k) {
            const bool pk = k > 0;
            const float* Lf = c.Lf + (size_t)LP * k;
            const float* Ws = c.Ws + (size_t)(NU * NS) * k;
            PROF2(24);
            if (tid < NU) {
                float g;
                if (tid < NF) {
                    const int r0 = 4 * (tid / 3);
                    g = 0.f;
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        float a0, a1, a2;
                        fric_row(c, prm, k, r0 + f, a0, a1, a2);
                        const float av = (tid % 3 == 0) ? a0 : ((tid % 3 == 1) ? a1 : a2);
                        g += c.dZ[NI * k + r0 + f] / c.T[NI * k + r0 + f] * av;
                    }
                    if (havep) g += c.fpv[NS + tid];
                } else {
                    const int q = tid - 24;
                    g = qfree(c, k, q) ? c.dZ[NI * k + 32 + q] / c.T[NI * k + 32 + q] - c.dZ[NI * k + 38 + q] / c.T[NI * k + 38 + q] : 0.f;
                }
                c.ybuf[tid] = g + Bt_vec<float>(c, prm, k, tid, c.fpv);
            }
            wave_lds_sync();
            PROF2(25);
            {   // dl = Linv dq : two lanes per row
                const int half = tid >> 5, i = (tid & 31) < NU ? (tid & 31) : 0;
                const float* Li = Lf + lpk(i, 0);
                float t0 = 0.f, t1 = 0.f, t2 = 0.f;
                const int a0 = 15 * half;
#pragma unroll
                for (int aa = 0; aa < 15; aa += 3) {
                    const int a = a0 + aa;
                    const float l0 = Li[a <= i ? a : 0], l1 = Li[a + 1 <= i ? a + 1 : 0], l2 = Li[a + 2 <= i ? a + 2 : 0];
                    t0 += (a <= i ? l0 : 0.f) * c.ybuf[a];
                    t1 += (a + 1 <= i ? l1 : 0.f) * c.ybuf[a + 1];
                    t2 += (a + 2 <= i ? l2 : 0.f) * c.ybuf[a + 2];
                }
                float v = t0 + t1 + t2;
                v += __shfl_xor(v, 32);
                if (tid < NU) {
                    c.ybuf[32 + tid] = v;
                    c.lqs[NU * k + tid] += v;
                }
            }
            wave_lds_sync();
            PROF2(26);
            if (tid < NXA) {
                float v;
                const float* dl = c.ybuf + 32;
                if (tid < NS) {
                    float t0 = At_vec<float>(c, prm, k, tid, c.fpv), t1 = 0.f, t2 = 0.f;
#pragma unroll
                    for (int a = 0; a < NU; a += 3) {
                        t0 -= Ws[a * NS + tid] * dl[a];
                        t1 -= Ws[(a + 1) * NS + tid] * dl[a + 1];
                        t2 -= Ws[(a + 2) * NS + tid] * dl[a + 2];
                    }
                    v = t0 + t1 + t2;
                } else if (pk) {
                    const int m = tid - NS;
                    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
                    for (int a = 0; a < NU; a += 3) {
                        const float l0 = Lf[lpk(a, 0) + (m <= a ? m : 0)], l1 = Lf[lpk(a + 1, 0) + (m <= a + 1 ? m : 0)],
                                    l2 = Lf[lpk(a + 2, 0) + (m <= a + 2 ? m : 0)];
                        t0 += (a >= m ? l0 : 0.f) * dl[a];
                        t1 += (a + 1 >= m ? l1 : 0.f) * dl[a + 1];
                        t2 += (a + 2 >= m ? l2 : 0.f) * dl[a + 2];
                    }
                    v = (t0 + t1 + t2) * prm.D[m % 3];
                } else v = 0.f;
                c.fpn[tid] = v;
            }
            wave_lds_sync();
            if (tid < NXA) c.fpv[tid] = c.fpn[tid];
            havep = pk;
            wave_lds_sync();
            PROF2(27);
        }
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

// new costates (backward, wave 0), blended into LAM with step ap:
//   lam_k = gs_k + Q_k ds_k + S_k^T du_k + A_k^T lam_{k+1}
__device__ void costate_update(const Ctx& c, const CmpcConsts& prm, int tid, float ap, bool use_exact)
{
    const int N = c.N;
    if (tid < 64) {
        if (tid < 3) c.qs[tid] = c.LAM[NS * N + 6 + tid];  // lam_h,N the Hessian used
        if (tid < NS) c.pv[tid] = grad_track(c, prm, N, tid) + (double)qdiag(prm, N, tid) * (double)c.dS[NS * N + tid];
        wave_lds_sync();
        if (tid < NS) c.LAM[NS * N + tid] += (double)ap * (c.pv[tid] - c.LAM[NS * N + tid]);
        for (int k = N - 1; k >= 1; --k) {
            if (tid < NS) {
                double v = grad_track(c, prm, k, tid) + (double)qdiag(prm, k, tid) * (double)c.dS[NS * k + tid] + At_vec<double>(c, prm, k, tid, c.pv);
                if (use_exact && (tid < 3 || tid >= 9)) {
                    const double* lh = c.qs;  // old lam_h of stage k+1
                    const int i = tid < 3 ? tid : (tid - 9) % 3;
                    const int a1 = (i + 1) % 3, a2 = (i + 2) % 3;
                    const float* du = c.dU + NU * k;
                    const double g0 = gam_of(c, 0, k), g1 = gam_of(c, 1, k);
                    double F1, F2;  // components a1, a2 of the gam-weighted force-step sum
                    if (tid < 3) {
                        F1 = g0 * ((double)du[a1] + du[3 + a1] + du[6 + a1] + du[9 + a1]) + g1 * ((double)du[12 + a1] + du[15 + a1] + du[18 + a1] + du[21 + a1]);
                        F2 = g0 * ((double)du[a2] + du[3 + a2] + du[6 + a2] + du[9 + a2]) + g1 * ((double)du[12 + a2] + du[15 + a2] + du[18 + a2] + du[21 + a2]);
                    } else {
                        const int ct = (tid - 9) / 3;
                        const float* df = du + 12 * ct;
                        const double g = ct ? g1 : g0;
                        F1 = g * ((double)df[a1] + df[3 + a1] + df[6 + a1] + df[9 + a1]);
                        F2 = g * ((double)df[a2] + df[3 + a2] + df[6 + a2] + df[9 + a2]);
                    }
                    // (S^T du) = dt (F x lam_h) on pos rows, minus that on com rows
                    const double sxtf = (double)prm.dt * (F1 * lh[a2] - F2 * lh[a1]);
                    v += (tid < 3) ? -sxtf : sxtf;
                }
                c.pn[tid] = v;
            }
            wave_lds_sync();
            if (tid < NS) {
                c.pv[tid] = c.pn[tid];
                if (tid >= 6 && tid < 9) c.qs[tid - 6] = c.LAM[NS * k + tid];
                c.LAM[NS * k + tid] += (double)ap * (c.pn[tid] - c.LAM[NS * k + tid]);
            }
            wave_lds_sync();
        }
    }
    __syncthreads();
}

// NC > 0: horizon known at compile time (every LDS offset becomes an immediate); NC == 0: runtime N
// FG: the per-stage factors (Linv, Ws: 915 floats per stage) live in global scratch instead of LDS
// (horizons whose LDS image would exceed 160 KiB)
template <int NT, int NC, bool FG>
__global__ __launch_bounds__(NT, FG ? 2 : 1) void cmpc_solve_kernel(CmpcParams kp)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int N = NC > 0 ? NC : kp.N;
    // constants first in LDS
    CmpcConsts& prmw = *reinterpret_cast<CmpcConsts*>(smem);
    {
        const int* src = reinterpret_cast<const int*>(kp.kc);
        int* dst = reinterpret_cast<int*>(smem);
        for (int e = tid; e < (int)(sizeof(CmpcConsts) / 4); e += NT) dst[e] = src[e];
    }
    const CmpcConsts& prm = prmw;
    constexpr int KBYTES = (sizeof(CmpcConsts) + 15) & ~15;
    const long long t_start = __builtin_amdgcn_s_memtime();
    Ctx c;
    c.L.N = N;
    c.N = N;
    // ---- carve LDS: doubles first, then 16-byte aligned float panels ----
    double* dp = reinterpret_cast<double*>(smem + KBYTES);
    c.LAM = dp; dp += NS * (N + 1) + ((NS * (N + 1)) & 1);
    c.QuuD = dp; dp += 90;
    c.pv = dp; dp += 40; c.pn = dp; dp += 40; c.qs = dp; dp += 16; c.Pd = dp; dp += 40;
    c.sig = dp; dp += NI; c.gco = dp; dp += NI; c.redd = dp; dp += 8;
    float* fp = reinterpret_cast<float*>(dp);
    c.QuuF = fp; fp += NU * RLD;
    c.Pan = fp; fp += NPAN * RLD;
    float* spw = fp; fp += (c.L.np() + 3) & ~3;
    c.sp = spw;
    c.S = fp; fp += NS * (N + 1); c.U = fp; fp += NU * N; c.T = fp; fp += NI * N; c.Z = fp; fp += NI * N;
    c.dS = fp; fp += NS * (N + 1); c.dU = fp; fp += NU * N; c.dT = fp; fp += NI * N; c.dZ = fp; fp += NI * N;
    c.d = fp; fp += NS * N;
    c.lqs = fp; fp += NU * N;
    c.geoA = fp; fp += GEO * N;
    c.P0 = fp; fp += NXA * PLD; c.P1 = fp; fp += NXA * PLD;
    c.G = fp; fp += NXA * GLD; c.T1 = fp; fp += NS * NS;
    c.Bval = fp; fp += 3 * NU; c.Aval = fp; fp += 3 * NS + 3;
    c.arow = fp; fp += 96 + 12; c.ybuf = fp; fp += 64; c.fpv = fp; fp += 40; c.fpn = fp; fp += 40; c.red = fp; fp += 8;
    c.Brow = reinterpret_cast<int*>(fp); fp += 3 * NU;
    c.Arow = reinterpret_cast<int*>(fp); fp += 3 * NS + 3;
    c.flag = reinterpret_cast<int*>(fp); fp += 4;
    c.qmask = reinterpret_cast<int*>(fp); fp += CMPC_NMAX;
    c.tri = reinterpret_cast<unsigned short*>(fp); fp += NTRI / 2;
    if (FG) {
        c.Lf = kp.scratch + (size_t)b * kp.scratch_stride;
        c.Ws = c.Lf + (size_t)LP * N;
    } else {
        c.Lf = fp; fp += (size_t)LP * N;
        c.Ws = fp; fp += (size_t)(NU * NS) * N;
    }

    // ---- one-off tables, parameter vector (coalesced) ----
    for (int e = tid; e < NTRI; e += NT) {
        int i = (int)((sqrtf(8.f * (float)e + 1.f) - 1.f) * 0.5f);
        while ((i + 1) * (i + 2) / 2 <= e) ++i;
        while (i * (i + 1) / 2 > e) --i;
        c.tri[e] = (unsigned short)((i << 8) | (e - i * (i + 1) / 2));
    }
    for (int e = tid; e < NPAN * RLD; e += NT) c.Pan[e] = 0.f;
    for (int e = tid; e < NU * RLD; e += NT) c.QuuF[e] = 0.f;
    if (tid < 90) c.QuuD[tid] = 0.0;
    {
        const float* gp = kp.P + (size_t)b * c.L.np();
        for (int e = tid; e < c.L.np(); e += NT) spw[e] = gp[e];
    }
    __syncthreads();
    if (tid < N) {
        int m = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) m |= qfree_compute(c, tid, q) ? (1 << q) : 0;
        c.qmask[tid] = m;
    }
    __syncthreads();
    // ---- initial iterate from x0 ----
    {
        const float* x0 = kp.X0 + (size_t)b * c.L.nx();
        for (int e = tid; e < NS * (N + 1); e += NT) {
            const int k = e / NS, i = e % NS;
            float v;
            if (k == 0) {  // initial-condition rows of g hold exactly
                if (i < 9) v = c.sp[c.L.pCom0() + i];
                else v = c.sp[c.L.pCur((i - 9) / 3) + (i - 9) % 3];
            } else if (i < 9) v = x0[c.L.oCom() + 3 * (N + 1) * (i / 3) + 3 * k + i % 3];
            else v = x0[c.L.oPos((i - 9) / 3) + 3 * k + (i - 9) % 3];
            c.S[e] = v;
            c.LAM[e] = 0.0;
        }
        for (int e = tid; e < NU * N; e += NT) {
            const int k = e / NU, m = e % NU;
            float v = 0.f;
            if (m < NF) v = x0[c.L.oF(m / 12, (m % 12) / 3) + 3 * k + m % 3];
            else {
                const int q = m - 24, ct = q / 3, i = q % 3;
                if (qfree(c, k, q)) {
                    const float* R = c.sp + c.L.pR(ct) + 9 * k;
                    const float lo = qlo(c, k, q), hi = qhi(c, k, q), push = 0.01f * (hi - lo);
                    for (int a = 0; a < 3; ++a)
                        v += Rm(R, a, i) * (x0[c.L.oPos(ct) + 3 * (k + 1) + a] - c.sp[c.L.pNom(ct) + 3 * (k + 1) + a]);
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
                if (i < 32) t = fmaxf(t, kp.t_floor);
                z = kp.mu_init / t;
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
    float err = 0.f, ep = 0.f, mu_cur = 0.f, step_out = 0.f;
    bool finishing = false;
    for (it = 0; it < prm.max_iter + 1; ++it) {
        if (it == prm.max_iter && !finishing) break;
        // ---- residuals of the current iterate ----
        PROF_DECL;
        float l_ep = 0.f, l_ec = 0.f;
        double l_mu = 0.0;
        all_geo<NT>(c, prm, tid);
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
        ep = block_max<NT>(l_ep, c.red, tid);
        const float ec = block_max<NT>(l_ec, c.red, tid);
        mu_cur = (float)(block_sum<NT>(l_mu, c.redd, tid) / (double)nrow);
        PROF(10);
        // ---- predictor (affine scaling): factorise; on a non-positive pivot fall back to the
        // Gauss-Newton Hessian, then to a larger Levenberg shift ----
        // once the complementarity products sit at the barrier floor the predictor has nothing to predict
        // (sigma ~ 0, second-order term ~ 0): take plain centring Newton steps, one sweep pair instead of three
        const bool centring = !finishing && mu_cur <= 1.5f * prm.mu_min && ec <= 4.f * prm.mu_min;
        bool exact = prm.exact_hessian != 0;
        float reg = prm.reg;
        int fail = 1;
        for (int attempt = 0; attempt < 4; ++attempt) {
            fail = riccati_backward<NT>(c, prm, tid, exact, reg, centring ? prm.mu_min : 0.f);
            if (!fail) break;
            __syncthreads();
            ++gn; exact = false;
            if (attempt > 0) reg *= 1e3f;
        }
        if (fail) { status = 2; break; }
        PROF(11);
        float ap, ad, sigma = 0.f, mu_t = prm.mu_min;
        if (centring) {
            for (int e = tid; e < NI * N; e += NT) c.dZ[e] = row_active(c, e / NI, e % NI) ? prm.mu_min : 0.f;
            __syncthreads();
            riccati_forward<NT>(c, prm, tid, false);
            PROF(15);
        } else {
        riccati_forward<NT>(c, prm, tid, true);
        PROF(12);
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
        sigma = mu_aff / mu_cur;
        sigma = sigma * sigma * sigma;
        mu_t = fmaxf(sigma * mu_cur, prm.mu_min);
        // ---- corrector ----
        for (int e = tid; e < NI * N; e += NT)
            c.dZ[e] = row_active(c, e / NI, e % NI) ? mu_t - c.dT[e] * c.dZ[e] : 0.f;  // complementarity target
        __syncthreads();
        PROF(13);
        riccati_delta(c, prm, tid);
        PROF(14);
        riccati_forward<NT>(c, prm, tid, false);
        PROF(15);
        }
        step_lengths<NT>(c, tid, fmaxf(0.99f, 1.f - mu_t), ap, ad);
        // ---- costates, then the iterate ----
        costate_update(c, prm, tid, ap, exact);
        PROF(16);
        for (int e = tid; e < NS * (N + 1); e += NT) c.S[e] += ap * c.dS[e];
        for (int e = tid; e < NU * N; e += NT) c.U[e] += ap * c.dU[e];
        for (int e = tid; e < NI * N; e += NT) { c.T[e] += ap * c.dT[e]; c.Z[e] += ad * c.dZ[e]; }
        // ---- convergence: the Newton step itself is the error estimate.  Flat directions of the cost
        // (e.g. the internal force along the line joining the feet) are kept quiet by the Levenberg
        // shift `reg`.  The stationarity residual of a float32-stored iterate cannot go below ~1e-3
        // (one ulp of com_z moves its gradient by 2 w_z^2 ulp ~ 5e-3), so it is not the test. ----
        // The step is measured on what the cost and the dynamics see: the states, the deviation of
        // each corner force from its foot's mean, the force rate, the landing offsets.  A constant
        // internal force along the line joining two stance feet changes none of them (the NLP does
        // not determine it; it only drifts slowly towards the barrier's analytic centre).
        float l_st = 0.f;
        for (int e = tid; e < NS * (N + 1); e += NT) l_st = fmaxf(l_st, fabsf(c.dS[e]));
        for (int e = tid; e < NU * N; e += NT) {
            const int k = e / NU, m = e % NU;
            const float du = c.dU[e];
            if (m < NF) {
                const float* f = c.dU + NU * k + 12 * (m / 12) + m % 3;
                const float mean = 0.25f * (f[0] + f[3] + f[6] + f[9]);
                l_st = fmaxf(l_st, fabsf(du - gam_of(c, m / 12, k) * mean));
                if (k > 0) l_st = fmaxf(l_st, fabsf(du - c.dU[e - NU]));
            } else l_st = fmaxf(l_st, fabsf(du));
        }
        const float step = ap * block_max<NT>(l_st, c.red, tid);
        step_out = step;
        err = fmaxf(ep, ec);
#ifdef CMPC_PROFILE
        if (tid == 0 && b == 0 && it < 64) {
            float* tr = g_trace + 8 * it;
            tr[0] = mu_cur; tr[1] = ep; tr[2] = ec; tr[3] = step; tr[4] = ap; tr[5] = ad; tr[6] = sigma; tr[7] = mu_t;
        }
#endif
        PROF(17);
        if (err <= prm.tol && step <= prm.step_tol) {
            status = 0;
            if (!prm.final_extrap) { ++it; break; }
            finishing = true;
        }
    }
    // ---- export x in the reference layout ----
    {
        float* x = kp.X + (size_t)b * c.L.nx();
        for (int e = tid; e < NS * (N + 1); e += NT) {
            const int k = e / NS, i = e % NS;
            if (i < 9) x[c.L.oCom() + 3 * (N + 1) * (i / 3) + 3 * k + i % 3] = c.S[e];
            else x[c.L.oPos((i - 9) / 3) + 3 * k + (i - 9) % 3] = c.S[e];
        }
        for (int e = tid; e < NU * N; e += NT) {
            const int k = e / NU, m = e % NU;
            if (m < NF) x[c.L.oF(m / 12, (m % 12) / 3) + 3 * k + m % 3] = c.U[e];
            else {
                const int q = m - 24, ct = q / 3, i = q % 3;
                const float v = gam_of(c, ct, k) < 0.5f ? (c.S[NS * (k + 1) + 9 + q] - c.S[NS * k + 9 + q]) / prm.dt : 0.f;
                x[c.L.oVel(ct) + 3 * k + i] = v;
            }
        }
        if (kp.info && tid == 0) {
            float* inf = kp.info + (size_t)b * CMPC_INFO_N;
            inf[0] = (float)it; inf[1] = err; inf[2] = mu_cur; inf[3] = (float)gn; inf[4] = ep; inf[5] = (float)status;
            inf[6] = (float)(__builtin_amdgcn_s_memtime() - t_start); inf[7] = step_out;
        }
    }
}

}  // namespace

// LDS bytes the kernel needs for horizon N
extern "C" size_t cmpc_solver_lds_bytes(int N, int factors_global)
{
    CmpcLayout L;
    cmpc_layout_init(L, N);
    const size_t nlam = (size_t)NS * (N + 1) + (((size_t)NS * (N + 1)) & 1);
    const size_t dbl = nlam + 90 + 40 + 40 + 16 + 40 + NI + NI + 8;
    const size_t flt = (size_t)NU * RLD + (size_t)NPAN * RLD + ((L.np + 3) & ~3)
                       + 2 * ((size_t)NS * (N + 1) + (size_t)NU * N + 2 * (size_t)NI * N) + (size_t)NS * N + (size_t)NU * N
                       + (size_t)GEO * N + 2 * NXA * PLD + NXA * GLD + NS * NS + 3 * NU + 3 * NS + 3 + 96 + 12 + 64 + 40 + 40 + 8
                       + 3 * NU + 3 * NS + 3 + 4 + CMPC_NMAX + NTRI / 2 + (factors_global ? 0 : (size_t)LP * N + (size_t)NU * NS * N);
    return ((sizeof(CmpcConsts) + 15) & ~(size_t)15) + dbl * 8 + flt * 4;
}

#ifdef CMPC_PROFILE
extern "C" int cmpc_profile_read(long long* out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(long long) * 32);
    if (reset) { long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
    return (int)e;
}
extern "C" int cmpc_trace_read(float* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(float) * 64 * 8); }
#endif

extern "C" int cmpc_launch_solver(const CmpcParams* prm, size_t lds_bytes, hipStream_t stream)
{
    constexpr int NT = 256;
    void (*kern)(CmpcParams) = nullptr;
#ifdef CMPC_ONLY_N20  // developer builds: one instantiation
    kern = cmpc_solve_kernel<NT, 20, false>;
#else
    if (prm->scratch) {
        kern = prm->N == 20 ? cmpc_solve_kernel<NT, 20, true> : cmpc_solve_kernel<NT, 0, true>;
    } else {
        switch (prm->N) {  // horizons of the shipped configurations get compile-time layouts
            case 10: kern = cmpc_solve_kernel<NT, 10, false>; break;
            case 12: kern = cmpc_solve_kernel<NT, 12, false>; break;
            case 13: kern = cmpc_solve_kernel<NT, 13, false>; break;  // ergoCubSN000
            case 15: kern = cmpc_solve_kernel<NT, 15, false>; break;  // iCubGazeboV3
            case 20: kern = cmpc_solve_kernel<NT, 20, false>; break;  // ergoCubGazeboV1
            case 22: kern = cmpc_solve_kernel<NT, 22, false>; break;  // ergoCubSN001
            default: kern = cmpc_solve_kernel<NT, 0, false>; break;
        }
    }
#endif
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(prm->B), dim3(NT), lds_bytes, stream, *prm);
    return (int)hipGetLastError();
}
