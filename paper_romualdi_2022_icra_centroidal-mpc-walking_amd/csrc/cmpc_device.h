// Shared host/device definitions of the centroidal-MPC kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#define CMPC_NS 15   // stage state: com, dcom, h, posL, posR
#define CMPC_NF 24   // corner forces (contact c, corner j, axis i -> 12c+3j+i)
#define CMPC_NQ 6    // foot-frame landing offsets (3c+i)
#define CMPC_NU 30
#define CMPC_NXA 39  // NS + NF: the previous force rides along as state (force-rate cost)
#define CMPC_NI 44   // inequality rows per stage: 32 friction + 6 q upper + 6 q lower
#define CMPC_REC_N 1088  // floats of one stage's factor record (layout: cmpc_solver.hip)
#define CMPC_NMAX 40 // largest horizon the kernels are built for
#define CMPC_TZ_LDS_NMAX 20  // HBM-factor variant: slacks / multipliers stay in LDS up to this horizon (three workgroups per CU still fit)
#define CMPC_INFO_N 8

// x / p index layout of the reference's generated NLP (tmp.c:62-67)
struct CmpcLayout {
    int N;
    int p_R[2], p_up[2], p_lo[2], p_gam[2], p_nom[2], p_cur[2];
    int p_com0, p_dcom0, p_h0, p_comref, p_href, p_fext, p_text, np;
    int o_com, o_dcom, o_h, o_pos[2], o_vel[2], o_f[2][4], nx, ng;
};

__host__ __device__ inline void cmpc_layout_init(CmpcLayout& L, int N)
{
    int o = 0;
    L.N = N;
    for (int c = 0; c < 2; ++c) {
        L.p_R[c] = o; o += 9 * N;
        L.p_up[c] = o; o += 3 * N;
        L.p_lo[c] = o; o += 3 * N;
        L.p_gam[c] = o; o += N;
        L.p_nom[c] = o; o += 3 * (N + 1);
        L.p_cur[c] = o; o += 3;
    }
    L.p_com0 = o; o += 3; L.p_dcom0 = o; o += 3; L.p_h0 = o; o += 3;
    L.p_comref = o; o += 3 * (N + 1); L.p_href = o; o += 3 * (N + 1);
    L.p_fext = o; o += 3 * N; L.p_text = o; o += 3 * N;
    L.np = o;
    o = 0;
    L.o_com = o; o += 3 * (N + 1); L.o_dcom = o; o += 3 * (N + 1); L.o_h = o; o += 3 * (N + 1);
    for (int c = 0; c < 2; ++c) {
        L.o_pos[c] = o; o += 3 * (N + 1);
        L.o_vel[c] = o; o += 3 * N;
        for (int j = 0; j < 4; ++j) { L.o_f[c][j] = o; o += 3 * N; }
    }
    L.nx = o;
    L.ng = 53 * N + 15;
}

// problem-independent constants; the kernels copy them to LDS once (kernel arguments live in
// SGPRs, and ~100 of them would be spilled and re-read all through the sweeps)
struct CmpcConsts {
    int N, max_iter, exact_hessian, final_extrap;
    int tail_stages, tail_iters;  // tail polish: stages re-solved after convergence (0: off), Newton steps of the re-solve
    float tail_trigger;          // ... when the extrapolation step of those stages exceeds this fraction of the largest force
    float dt, mu_fr, grav;
    float w_com0, w_com1, w_h, w_pos, w_sym;
    float D[3];                  // 2 * force_rate_of_change_weight
    float tol, step_tol, mu_init, mu_min;
    float reg;                   // Levenberg shift on the diagonal of every stage Hessian Quu
    float sigma_min;             // lower bound of Mehrotra's centring parameter (caps the barrier decrease per iteration)
    float corners[24];           // [c][j][3]
    float wz2[CMPC_NMAX + 1];    // 2 w_z(k)^2, w_z(k) = (w_cz/2)(1+exp(-k))
    int hwid_probe;              // diagnostic build only: phase_export overwrites x[0..7] with the HW_ID of each wave
};

// The same layout as closed-form index functions.  Device code uses these: indexing the offset arrays
// of CmpcLayout with a run-time contact/corner number forces the whole struct into scratch memory
// (private arrays cannot be register-indexed), which turns every LDS access into a scratch load first.
struct CmpcIdx {
    int N;
    __host__ __device__ int pR(int c) const { return c * (19 * N + 6); }
    __host__ __device__ int pUp(int c) const { return pR(c) + 9 * N; }
    __host__ __device__ int pLo(int c) const { return pR(c) + 12 * N; }
    __host__ __device__ int pGam(int c) const { return pR(c) + 15 * N; }
    __host__ __device__ int pNom(int c) const { return pR(c) + 16 * N; }
    __host__ __device__ int pCur(int c) const { return pR(c) + 19 * N + 3; }
    __host__ __device__ int pCom0() const { return 2 * (19 * N + 6); }
    __host__ __device__ int pDcom0() const { return pCom0() + 3; }
    __host__ __device__ int pH0() const { return pCom0() + 6; }
    __host__ __device__ int pComref() const { return pCom0() + 9; }
    __host__ __device__ int pHref() const { return pComref() + 3 * (N + 1); }
    __host__ __device__ int pFext() const { return pHref() + 3 * (N + 1); }
    __host__ __device__ int pText() const { return pFext() + 3 * N; }
    __host__ __device__ int np() const { return pText() + 3 * N; }
    __host__ __device__ int oCom() const { return 0; }
    __host__ __device__ int oDcom() const { return 3 * (N + 1); }
    __host__ __device__ int oH() const { return 6 * (N + 1); }
    __host__ __device__ int oPos(int c) const { return 9 * (N + 1) + c * (18 * N + 3); }
    __host__ __device__ int oVel(int c) const { return oPos(c) + 3 * (N + 1); }
    __host__ __device__ int oF(int c, int j) const { return oVel(c) + 3 * N + j * 3 * N; }
    __host__ __device__ int nx() const { return 45 * N + 15; }
    __host__ __device__ int ng() const { return 53 * N + 15; }
};

// kernel parameters (passed by value)
struct CmpcParams {
    const CmpcConsts* kc;        // device memory
    int N, B;
    const float* P;              // [B][np]
    const float* X0;             // [B][nx]
    float* X;                    // [B][nx]
    float* info;                 // [B][CMPC_INFO_N] or null
    float mu_init, t_floor;      // starting barrier parameter and slack floor of this solve (cold: 0.1 / 1e-2)
    float mu_adapt;              // > 0: mu_init is replaced per problem by clamp(mu_adapt * ep0^2, 0.03, 0.5)
    int warm;                    // the initial guess is a shifted previous solution: a problem that fails is restarted cold
    int warm_budget;             // warm starts: iteration budget of the warm-started pass (0: max_iter)
    int warm_no_restart;         // warm starts: 1 = a pass that exhausts its budget returns status 1 (the caller re-solves it), 0 = it is started again
                                 //  from the cold start inside the kernel
    float* duals;                // [B][NS (N+1) + 2 NI N] costates | slacks | multipliers of the last solve (written at exit; read, shifted
    int warm_duals;              //  by one knot, at the start of a warm solve if warm_duals != 0); may be null
    float* scratch;              // per-problem factor storage when it does not fit in LDS, else null
    long long scratch_stride;    // floats per problem
    int lds_words;               // 4-byte words of dynamic LDS the launch was given (set by cmpc_launch_solver)
};
