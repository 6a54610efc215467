/* TEST INFRASTRUCTURE ONLY -- stage-structured primal-dual interior-point solver for the
 * reference's centroidal-MPC NLP: the CPU restatement ("port") of the algorithm the HIP kernels
 * run (csrc/cmpc_solver.hip follows this file function by function).
 *
 * What it replaces: BipedalLocomotion::ReducedModelControllers::CentroidalMPC::advance(), called
 * at /root/reference/src/centroidal-mpc-walking/src/CentroidalMPCBlock.cpp:615 (CasADi Opti ->
 * IPOPT).  Problem data = the NLP parameter vector p and decision vector x in the layout of the
 * reference's generated code (config/robots/ergoCubGazeboV1/tmp.c:62-67; see nlp_ref.c).
 *
 * Algorithm (IPOPT's barrier method, stage-structured linear algebra):
 *   - contact velocities are eliminated (they carry no cost): in a swing stage (Gamma=0) the next
 *     foot position is set by a foot-frame offset q inside the bounding box, pos+ = nominal+ + R q;
 *     in a stance stage pos+ = pos.  Bounding-box rows with lower==upper fix that q component.
 *     Stance-stage bounding-box rows are constants / duplicates of the landing row and dropped.
 *   - friction rows and q bounds get slacks t>0 and multipliers z>0 (t z = mu), eliminated into
 *     the stage Hessian; dynamics are kept as stage equalities (multiple shooting, defects d_k).
 *   - the Newton/KKT system is solved by a Riccati recursion over the N stages with the previous
 *     force as extra state (the force-rate cost couples f_k and f_{k-1}): nx = 15+24, nu = 24+6.
 *   - exact Lagrangian Hessian (bilinear momentum term) with Gauss-Newton fallback when a stage
 *     Cholesky meets a non-positive pivot.
 *   - a converged solve ends with one affine-scaling step (mu -> 0, primal only); optionally (opts.tail_stages, what the HIP
 *     kernel ships) the last stages are then re-solved on their own with per-row barrier targets: tail_polish() below.
 * Compiled twice: REAL=double (oracle / CPU baseline) and REAL=float (precision study).
 */
#include "cmpc_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL double   /* vectors: iterate, residuals, right-hand sides, steps */
#define MREAL double  /* matrices: linearisation, stage Hessians, Riccati factors */
#define PREAL double  /* problem I/O: p, x0, x */
#define CREAL double  /* stage Hessian Quu and its Cholesky (barrier terms z/t span 1e-6..1e9) */
#define FN(n) n
#endif

#define NS 15  /* stage state: com, dcom, h, posL, posR */
#define NF 24  /* corner forces */
#define NQ 6   /* foot-frame offsets */
#define NU 30
#define NXA 39 /* NS + NF (previous force) */
#define NI 44  /* inequality rows per stage: 32 friction + 6 q upper + 6 q lower */
#define NMAX 64

typedef struct {
    int N;
    int p_R[2], p_up[2], p_lo[2], p_gam[2], p_nom[2], p_cur[2];
    int p_com0, p_dcom0, p_h0, p_comref, p_href, p_fext, p_text, np;
    int o_com, o_dcom, o_h, o_pos[2], o_vel[2], o_f[2][4], nx;
} lay;

static void lay_init(lay* L, int N)
{
    int o = 0, c, j;
    L->N = N;
    for (c = 0; c < 2; ++c) {
        L->p_R[c] = o; o += 9 * N;
        L->p_up[c] = o; o += 3 * N;   /* limA: upper (BLF declares upper before lower; unpinned) */
        L->p_lo[c] = o; o += 3 * N;   /* limB: lower */
        L->p_gam[c] = o; o += N;
        L->p_nom[c] = o; o += 3 * (N + 1);
        L->p_cur[c] = o; o += 3;
    }
    L->p_com0 = o; o += 3; L->p_dcom0 = o; o += 3; L->p_h0 = o; o += 3;
    L->p_comref = o; o += 3 * (N + 1); L->p_href = o; o += 3 * (N + 1);
    L->p_fext = o; o += 3 * N; L->p_text = o; o += 3 * N;
    L->np = o;
    o = 0;
    L->o_com = o; o += 3 * (N + 1); L->o_dcom = o; o += 3 * (N + 1); L->o_h = o; o += 3 * (N + 1);
    for (c = 0; c < 2; ++c) {
        L->o_pos[c] = o; o += 3 * (N + 1);
        L->o_vel[c] = o; o += 3 * N;
        for (j = 0; j < 4; ++j) { L->o_f[c][j] = o; o += 3 * N; }
    }
    L->nx = o;
}

typedef struct {
    const cmpc_nlp_cfg* cfg;
    lay L;
    int N;
    const PREAL* p;           /* parameter vector (reference layout) */
    REAL wz2[NMAX + 1];       /* 2 w_z(k)^2 */
    REAL D[3];                /* 2 w_rate */
    /* per-stage constants */
    MREAL arow[NMAX][96];      /* friction rows, world frame: a = R (sx, sy, -mu)^T */
    REAL qlo[NMAX][6], qhi[NMAX][6];
    int qfree[NMAX][6];
    /* iterate */
    REAL S[NMAX + 1][NS], U[NMAX][NU], LAM[NMAX + 1][NS], T[NMAX][NI], Z[NMAX][NI];
    /* step */
    REAL dS[NMAX + 1][NS], dU[NMAX][NU], LAMn[NMAX + 1][NS], dT[NMAX][NI], dZ[NMAX][NI];
    /* linearisation + factors */
    MREAL A[NMAX][NS * NS], B[NMAX][NS * NU];
    REAL d[NMAX][NS];
    MREAL Lc[NMAX][NU * NU], W[NMAX][NU * NXA];
    REAL lq[NMAX][NU];
    REAL gs[NMAX + 1][NS];    /* tracking gradient */
    MREAL Sx[NMAX][9];        /* exact-Hessian skew block dt*[lam_h]x of the stage (0 if GN) */
    REAL CMU[NMAX][NI];       /* per-row complementarity target (0: affine step; mu - dt_a dz_a: corrector) */
    REAL DG[NMAX][NI];        /* change of the row coefficient between predictor and corrector */
} ws;

static const REAL SXr[4] = {1, -1, -1, 1};
static const REAL SYr[4] = {1, 1, -1, -1};
#define RM(R, r, c) ((R)[3 * (c) + (r)]) /* reference stores vec(R) column-major */

static inline void crossr(const REAL* a, const REAL* b, REAL* o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

static inline REAL gam_of(const ws* w, int c, int k) { return w->p[w->L.p_gam[c] + k]; }

/* ---------------- problem set-up ---------------- */
static void setup(ws* w, const cmpc_nlp_cfg* cfg, const PREAL* p)
{
    const int N = cfg->N;
    int k, c, j, i, a;
    w->cfg = cfg; w->N = N; w->p = p;
    lay_init(&w->L, N);
    for (k = 0; k <= N; ++k) {
        double wzk = 0.5 * cfg->w_com[2] * (1.0 + exp(-(double)k));
        w->wz2[k] = (REAL)(2.0 * wzk * wzk);
    }
    for (i = 0; i < 3; ++i) w->D[i] = (REAL)(2.0 * cfg->w_rate[i]);
    for (k = 0; k < N; ++k)
        for (c = 0; c < 2; ++c) {
            const PREAL* R = p + w->L.p_R[c] + 9 * k;
            const REAL gam = gam_of(w, c, k);
            for (j = 0; j < 4; ++j)
                for (i = 0; i < 4; ++i)
                    for (a = 0; a < 3; ++a)
                        w->arow[k][3 * (16 * c + 4 * j + i) + a] =
                            (MREAL)(SXr[i] * (REAL)RM(R, a, 0) + SYr[i] * (REAL)RM(R, a, 1) - (REAL)cfg->mu * (REAL)RM(R, a, 2));
            for (i = 0; i < 3; ++i) {
                REAL lo = p[w->L.p_lo[c] + 3 * k + i], hi = p[w->L.p_up[c] + 3 * k + i];
                w->qlo[k][3 * c + i] = lo; w->qhi[k][3 * c + i] = hi;
                w->qfree[k][3 * c + i] = (gam < (REAL)0.5) && (hi - lo > (REAL)1e-9);
            }
        }
}

/* ---------------- stage functions ---------------- */
/* dynamics s+ = phi_k(s,u); optionally the Jacobians A (15x15), B (15x30), row-major */
static void dyn(const ws* w, int k, const REAL* s, const REAL* u, REAL* sn, MREAL* A, MREAL* B)
{
    const cmpc_nlp_cfg* cfg = w->cfg;
    const lay* L = &w->L;
    const REAL dt = (REAL)cfg->dt;
    const PREAL* p = w->p;
    REAL acc[3], tor[3], Fsum[3] = {0, 0, 0};
    int c, j, i, a;
    if (A) { memset(A, 0, sizeof(MREAL) * NS * NS); memset(B, 0, sizeof(MREAL) * NS * NU); }
    for (i = 0; i < 3; ++i) { acc[i] = p[L->p_fext + 3 * k + i]; tor[i] = p[L->p_text + 3 * k + i]; }
    acc[2] -= (REAL)cfg->gravity;
    for (c = 0; c < 2; ++c) {
        const PREAL* R = p + L->p_R[c] + 9 * k;
        const REAL gam = gam_of(w, c, k);
        const REAL* pos = s + 9 + 3 * c;
        REAL Fc[3] = {0, 0, 0};
        for (j = 0; j < 4; ++j) {
            const REAL* f = u + 12 * c + 3 * j;
            const double* cn = cfg->corners[c][j];
            REAL r[3], t[3];
            for (i = 0; i < 3; ++i)
                r[i] = RM(R, i, 0) * (REAL)cn[0] + RM(R, i, 1) * (REAL)cn[1] + RM(R, i, 2) * (REAL)cn[2] + pos[i] - s[i];
            crossr(r, f, t);
            for (i = 0; i < 3; ++i) { acc[i] += gam * f[i]; tor[i] += gam * t[i]; Fc[i] += f[i]; }
            if (A) {
                const int col = 12 * c + 3 * j;
                const REAL g = dt * gam;
                for (i = 0; i < 3; ++i) B[(3 + i) * NU + col + i] = g;
                B[6 * NU + col + 1] = -g * r[2]; B[6 * NU + col + 2] = g * r[1];
                B[7 * NU + col + 0] = g * r[2];  B[7 * NU + col + 2] = -g * r[0];
                B[8 * NU + col + 0] = -g * r[1]; B[8 * NU + col + 1] = g * r[0];
            }
        }
        for (i = 0; i < 3; ++i) Fsum[i] += gam * Fc[i];
        for (i = 0; i < 3; ++i) {
            REAL land = p[L->p_nom[c] + 3 * (k + 1) + i];
            for (a = 0; a < 3; ++a) land += RM(R, i, a) * u[24 + 3 * c + a];
            sn[9 + 3 * c + i] = gam * pos[i] + ((REAL)1 - gam) * land;
        }
        if (A) {
            const int col = 9 + 3 * c;
            const REAL g = -dt * gam; /* d h+/d pos_c = -dt gam [Fc]x */
            A[6 * NS + col + 1] = -g * Fc[2]; A[6 * NS + col + 2] = g * Fc[1];
            A[7 * NS + col + 0] = g * Fc[2];  A[7 * NS + col + 2] = -g * Fc[0];
            A[8 * NS + col + 0] = -g * Fc[1]; A[8 * NS + col + 1] = g * Fc[0];
            for (i = 0; i < 3; ++i) {
                A[(col + i) * NS + col + i] = gam;
                for (a = 0; a < 3; ++a)
                    if (w->qfree[k][3 * c + a]) B[(col + i) * NU + 24 + 3 * c + a] = ((REAL)1 - gam) * RM(R, i, a);
            }
        }
    }
    for (i = 0; i < 3; ++i) {
        sn[i] = s[i] + dt * s[3 + i];
        sn[3 + i] = s[3 + i] + dt * acc[i];
        sn[6 + i] = s[6 + i] + dt * tor[i];
    }
    if (A) {
        for (i = 0; i < 9; ++i) A[i * NS + i] = 1;
        for (i = 0; i < 3; ++i) A[i * NS + 3 + i] = dt;
        /* d h+/d com = dt [Fsum]x */
        A[6 * NS + 1] = -dt * Fsum[2]; A[6 * NS + 2] = dt * Fsum[1];
        A[7 * NS + 0] = dt * Fsum[2];  A[7 * NS + 2] = -dt * Fsum[0];
        A[8 * NS + 0] = -dt * Fsum[1]; A[8 * NS + 1] = dt * Fsum[0];
    }
}

static void grad_track(const ws* w, int k, const REAL* s, REAL* gs)
{
    const cmpc_nlp_cfg* cfg = w->cfg;
    const lay* L = &w->L;
    const PREAL* p = w->p;
    int i, c;
    gs[0] = 2 * (REAL)cfg->w_com[0] * (s[0] - p[L->p_comref + 3 * k + 0]);
    gs[1] = 2 * (REAL)cfg->w_com[1] * (s[1] - p[L->p_comref + 3 * k + 1]);
    gs[2] = w->wz2[k] * (s[2] - p[L->p_comref + 3 * k + 2]);
    for (i = 0; i < 3; ++i) {
        gs[3 + i] = 0;
        gs[6 + i] = 2 * (REAL)cfg->w_h * (s[6 + i] - p[L->p_href + 3 * k + i]);
        for (c = 0; c < 2; ++c)
            gs[9 + 3 * c + i] = 2 * (REAL)cfg->w_pos * (s[9 + 3 * c + i] - p[L->p_nom[c] + 3 * k + i]);
    }
}

static inline REAL qdiag(const ws* w, int k, int i)
{
    if (i == 0) return 2 * (REAL)w->cfg->w_com[0];
    if (i == 1) return 2 * (REAL)w->cfg->w_com[1];
    if (i == 2) return w->wz2[k];
    if (i < 6) return 0;
    if (i < 9) return 2 * (REAL)w->cfg->w_h;
    return 2 * (REAL)w->cfg->w_pos;
}

/* gradient of the force-symmetry cost w.r.t. f_k */
static void grad_sym(const ws* w, int k, const REAL* u, REAL* gu)
{
    int c, j, i;
    for (c = 0; c < 2; ++c) {
        const REAL gam = gam_of(w, c, k);
        for (i = 0; i < 3; ++i) {
            REAL mean = 0, esum = 0, e[4];
            for (j = 0; j < 4; ++j) mean += (REAL)0.25 * u[12 * c + 3 * j + i];
            for (j = 0; j < 4; ++j) { e[j] = u[12 * c + 3 * j + i] - gam * mean; esum += e[j]; }
            for (j = 0; j < 4; ++j)
                gu[12 * c + 3 * j + i] = 2 * (REAL)w->cfg->w_sym * (e[j] - (REAL)0.25 * gam * esum);
        }
    }
    for (i = 0; i < NQ; ++i) gu[24 + i] = 0;
}

/* inequality rows of stage k.  0..31 friction (corner cj = i/4, face i%4): a^T f_cj <= 0;
 * 32..37: q_m - hi_m <= 0 ; 38..43: lo_m - q_m <= 0 (only for free q components). */
static inline int row_active(const ws* w, int k, int i) { return i < 32 ? 1 : w->qfree[k][(i - 32) % 6]; }

static inline REAL row_val(const ws* w, int k, int i, const REAL* u)
{
    if (i < 32) {
        const MREAL* a = w->arow[k] + 3 * i;
        const REAL* f = u + 3 * (i / 4);
        return a[0] * f[0] + a[1] * f[1] + a[2] * f[2];
    }
    if (i < 38) return u[24 + i - 32] - w->qhi[k][i - 32];
    return w->qlo[k][i - 38] - u[24 + i - 38];
}

static inline REAL row_dot(const ws* w, int k, int i, const REAL* du)
{
    if (i < 32) {
        const MREAL* a = w->arow[k] + 3 * i;
        const REAL* f = du + 3 * (i / 4);
        return a[0] * f[0] + a[1] * f[1] + a[2] * f[2];
    }
    if (i < 38) return du[24 + i - 32];
    return -du[24 + i - 38];
}

/* add sum_i a_i coef_i to a 30-vector */
static inline void row_axpy(const ws* w, int k, int i, REAL coef, REAL* g)
{
    if (i < 32) {
        const MREAL* a = w->arow[k] + 3 * i;
        REAL* f = g + 3 * (i / 4);
        f[0] += a[0] * coef; f[1] += a[1] * coef; f[2] += a[2] * coef;
    } else if (i < 38) g[24 + i - 32] += coef;
    else g[24 + i - 38] -= coef;
}

static int chol(CREAL* A, int n)
{
    int i, j, k;
    for (j = 0; j < n; ++j) {
        CREAL d = A[j * n + j];
        for (k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0)) return 1;
        d = (CREAL)sqrt((double)d);
        A[j * n + j] = d;
        for (i = j + 1; i < n; ++i) {
            CREAL v = A[i * n + j];
            for (k = 0; k < j; ++k) v -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = v / d;
        }
    }
    return 0;
}

/* ---------------- linearise: A,B,d, tracking gradients ---------------- */
static void linearise(ws* w)
{
    const int N = w->N;
    int k, i;
    for (k = 0; k < N; ++k) {
        REAL sn[NS];
        dyn(w, k, w->S[k], w->U[k], sn, w->A[k], w->B[k]);
        for (i = 0; i < NS; ++i) w->d[k][i] = sn[i] - w->S[k + 1][i];
        grad_track(w, k, w->S[k], w->gs[k]);
    }
    grad_track(w, N, w->S[N], w->gs[N]);
}

/* KKT residuals of the current iterate (needs linearise() first).
 * stat: stationarity, prim: dynamics defects + inequality residuals, comp(mu): |t z - mu| */
static void kkt_error(const ws* w, REAL mu, REAL* stat, REAL* prim, REAL* comp, REAL* zsum)
{
    const int N = w->N;
    REAL es = 0, ep = 0, ec = 0, zs = 0;
    int k, i, a, nz = 0;
    for (k = 0; k < N; ++k) {
        REAL ru[NU];
        grad_sym(w, k, w->U[k], ru);
        for (i = 0; i < NF; ++i) {
            if (k > 0) ru[i] += w->D[i % 3] * (w->U[k][i] - w->U[k - 1][i]);
            if (k + 1 < N) ru[i] -= w->D[i % 3] * (w->U[k + 1][i] - w->U[k][i]);
        }
        for (i = 0; i < NU; ++i)
            for (a = 0; a < NS; ++a) ru[i] += w->B[k][a * NU + i] * w->LAM[k + 1][a];
        for (i = 0; i < NI; ++i) {
            REAL r;
            if (!row_active(w, k, i)) continue;
            row_axpy(w, k, i, w->Z[k][i], ru);
            r = row_val(w, k, i, w->U[k]) + w->T[k][i];
            if (fabs((double)r) > ep) ep = (REAL)fabs((double)r);
            r = w->T[k][i] * w->Z[k][i] - mu;
            if (fabs((double)r) > ec) ec = (REAL)fabs((double)r);
            zs += w->Z[k][i]; ++nz;
        }
        for (i = 0; i < NU; ++i) {
            if (i >= 24 && !w->qfree[k][i - 24]) continue;
            if (fabs((double)ru[i]) > es) es = (REAL)fabs((double)ru[i]);
        }
        for (i = 0; i < NS; ++i)
            if (fabs((double)w->d[k][i]) > ep) ep = (REAL)fabs((double)w->d[k][i]);
        if (k > 0)
            for (i = 0; i < NS; ++i) {
                REAL r = w->gs[k][i] - w->LAM[k][i];
                for (a = 0; a < NS; ++a) r += w->A[k][a * NS + i] * w->LAM[k + 1][a];
                if (fabs((double)r) > es) es = (REAL)fabs((double)r);
            }
    }
    for (i = 0; i < NS; ++i) {
        REAL r = w->gs[N][i] - w->LAM[N][i];
        if (fabs((double)r) > es) es = (REAL)fabs((double)r);
    }
    *stat = es; *prim = ep; *comp = ec; *zsum = nz ? zs / (REAL)nz : 0;
}

/* ---------------- Riccati backward sweep ---------------- */
/* builds and factorises the stage QPs; returns 0 ok / 1 non-positive pivot at some stage */
static int riccati_backward(ws* w, int use_exact, int k0)
{
    const cmpc_nlp_cfg* cfg = w->cfg;
    const int N = w->N;
    const REAL dt = (REAL)cfg->dt;
    MREAL Pm[NXA * NXA], Qux[NU * NXA], Qss[NS * NS], PA[NS * NS], PB[NS * NU];
    CREAL Quu[NU * NU];
    REAL pv[NXA], qu[NU], qs[NS], Pd[NS];
    int k, i, j, a, b, c, havep = 0;

    memset(Pm, 0, sizeof(Pm)); memset(pv, 0, sizeof(pv));
    for (i = 0; i < NS; ++i) { Pm[i * NXA + i] = qdiag(w, N, i); pv[i] = w->gs[N][i]; }

    for (k = N - 1; k >= k0; --k) {
        const MREAL* A = w->A[k];
        const MREAL* B = w->B[k];
        const REAL* d = w->d[k];
        const REAL* u = w->U[k];
        MREAL* Lc = w->Lc[k];
        MREAL* W = w->W[k];
        REAL* lq = w->lq[k];
        REAL gu[NU];
        const int pk = (k > 0);
        const int ncol = pk ? NXA : NS;
        /* ---- R0, gu0: symmetry + barrier ---- */
        memset(Quu, 0, sizeof(Quu));
        grad_sym(w, k, u, gu);
        for (c = 0; c < 2; ++c) {
            const REAL gam = gam_of(w, c, k);
            const REAL offd = -(REAL)0.25 * gam * ((REAL)2 - gam);
            for (j = 0; j < 4; ++j)
                for (b = 0; b < 4; ++b)
                    for (i = 0; i < 3; ++i)
                        Quu[(12 * c + 3 * j + i) * NU + 12 * c + 3 * b + i] = 2 * (REAL)cfg->w_sym * ((j == b ? (REAL)1 : (REAL)0) + offd);
        }
        for (i = 0; i < NI; ++i) {
            REAL t, z, sig, r;
            if (!row_active(w, k, i)) continue;
            t = w->T[k][i]; z = w->Z[k][i]; sig = z / t;
            r = row_val(w, k, i, u) + t;
            row_axpy(w, k, i, w->CMU[k][i] / t + sig * r, gu);
            if (i < 32) {
                const MREAL* ar = w->arow[k] + 3 * i;
                const int o = 3 * (i / 4);
                for (a = 0; a < 3; ++a)
                    for (b = 0; b < 3; ++b) Quu[(o + a) * NU + o + b] += (CREAL)sig * (CREAL)ar[a] * (CREAL)ar[b];
            } else {
                const int o = 24 + (i - 32) % 6;
                Quu[o * NU + o] += sig;
            }
        }
        for (i = 0; i < NQ; ++i)
            if (!w->qfree[k][i]) { Quu[(24 + i) * NU + 24 + i] = 1; gu[24 + i] = 0; }
        /* rate pair (k-1,k) */
        if (pk)
            for (i = 0; i < NF; ++i) {
                Quu[i * NU + i] += w->D[i % 3];
                gu[i] += w->D[i % 3] * (u[i] - w->U[k - 1][i]);
            }
        /* ---- value-function terms ---- */
        for (i = 0; i < NS; ++i) {
            REAL acc = pv[i];
            for (a = 0; a < NS; ++a) acc += Pm[i * NXA + a] * d[a];
            Pd[i] = acc;
            for (j = 0; j < NU; ++j) {
                REAL v = 0;
                for (a = 0; a < NS; ++a) v += Pm[i * NXA + a] * B[a * NU + j];
                PB[i * NU + j] = v;
            }
            for (j = 0; j < NS; ++j) {
                REAL v = 0;
                for (a = 0; a < NS; ++a) v += Pm[i * NXA + a] * A[a * NS + j];
                PA[i * NS + j] = v;
            }
        }
        for (i = 0; i < NU; ++i) {
            REAL v = gu[i];
            for (a = 0; a < NS; ++a) v += B[a * NU + i] * Pd[a];
            qu[i] = v;
            for (j = 0; j < NU; ++j) {
                REAL v2 = 0;
                for (a = 0; a < NS; ++a) v2 += B[a * NU + i] * PB[a * NU + j];
                Quu[i * NU + j] += v2;
            }
        }
        memset(Qux, 0, sizeof(Qux));
        for (i = 0; i < NU; ++i)
            for (j = 0; j < NS; ++j) {
                REAL v = 0;
                for (a = 0; a < NS; ++a) v += B[a * NU + i] * PA[a * NS + j];
                Qux[i * NXA + j] = v;
            }
        if (havep) { /* V_{k+1} depends on delta f_k */
            for (i = 0; i < NF; ++i) {
                REAL v = pv[NS + i];
                for (a = 0; a < NS; ++a) v += Pm[(NS + i) * NXA + a] * d[a];
                qu[i] += v;
                for (j = 0; j < NF; ++j) Quu[i * NU + j] += Pm[(NS + i) * NXA + NS + j];
                for (j = 0; j < NU; ++j) {
                    REAL v2 = 0;
                    for (a = 0; a < NS; ++a) v2 += Pm[(NS + i) * NXA + a] * B[a * NU + j];
                    Quu[i * NU + j] += v2;
                    Quu[j * NU + i] += v2;
                }
                for (j = 0; j < NS; ++j) {
                    REAL v2 = 0;
                    for (a = 0; a < NS; ++a) v2 += Pm[(NS + i) * NXA + a] * A[a * NS + j];
                    Qux[i * NXA + j] += v2;
                }
            }
        }
        memset(w->Sx[k], 0, sizeof(MREAL) * 9);
        if (use_exact) {
            /* S[f_cj, pos_c] = dt gam [lam_h]x ; S[f_cj, com] = -dt gam [lam_h]x (lam multiplies phi - s+) */
            const REAL* lh = w->LAM[k + 1] + 6;
            MREAL* Sx = w->Sx[k];
            Sx[1] = -dt * lh[2]; Sx[2] = dt * lh[1];
            Sx[3] = dt * lh[2];  Sx[5] = -dt * lh[0];
            Sx[6] = -dt * lh[1]; Sx[7] = dt * lh[0];
            for (c = 0; c < 2; ++c) {
                const REAL gam = gam_of(w, c, k);
                for (j = 0; j < 4; ++j)
                    for (a = 0; a < 3; ++a)
                        for (b = 0; b < 3; ++b) {
                            Qux[(12 * c + 3 * j + a) * NXA + 9 + 3 * c + b] += gam * Sx[3 * a + b];
                            Qux[(12 * c + 3 * j + a) * NXA + b] -= gam * Sx[3 * a + b];
                        }
            }
        }
        if (pk) for (i = 0; i < NF; ++i) Qux[i * NXA + NS + i] = -w->D[i % 3];
        for (i = 0; i < NS; ++i) {
            REAL v = w->gs[k][i];
            for (a = 0; a < NS; ++a) v += A[a * NS + i] * Pd[a];
            qs[i] = v;
            for (j = 0; j < NS; ++j) {
                REAL v2 = 0;
                for (a = 0; a < NS; ++a) v2 += A[a * NS + i] * PA[a * NS + j];
                Qss[i * NS + j] = v2;
            }
            Qss[i * NS + i] += qdiag(w, k, i);
        }
        /* ---- factorise ---- */
        if (chol(Quu, NU)) return 1;
        for (i = 0; i < NU * NU; ++i) Lc[i] = (MREAL)Quu[i];
        memset(W, 0, sizeof(MREAL) * NU * NXA);
        for (j = 0; j < ncol; ++j)
            for (i = 0; i < NU; ++i) {
                REAL v = Qux[i * NXA + j];
                for (a = 0; a < i; ++a) v -= Lc[i * NU + a] * W[a * NXA + j];
                W[i * NXA + j] = v / Lc[i * NU + i];
            }
        for (i = 0; i < NU; ++i) {
            REAL v = qu[i];
            for (a = 0; a < i; ++a) v -= Lc[i * NU + a] * lq[a];
            lq[i] = v / Lc[i * NU + i];
        }
        /* ---- value function of stage k ---- */
        memset(Pm, 0, sizeof(Pm));
        for (i = 0; i < ncol; ++i) {
            REAL v = 0;
            for (a = 0; a < NU; ++a) v += W[a * NXA + i] * lq[a];
            if (i < NS) pv[i] = qs[i] - v;
            else pv[i] = -w->D[(i - NS) % 3] * (u[i - NS] - w->U[k - 1][i - NS]) - v;
            for (j = 0; j <= i; ++j) {
                REAL v2 = 0;
                for (a = 0; a < NU; ++a) v2 += W[a * NXA + i] * W[a * NXA + j];
                if (i < NS) v2 = Qss[i * NS + j] - v2; /* j <= i < NS */
                else if (i == j) v2 = w->D[(i - NS) % 3] - v2;
                else v2 = -v2;
                Pm[i * NXA + j] = v2; Pm[j * NXA + i] = v2;
            }
        }
        for (i = ncol; i < NXA; ++i) pv[i] = 0;
        havep = pk;
    }
    return 0;
}

/* vector-only backward sweep for a changed right-hand side: the row coefficients change by DG
 * (corrector of the predictor-corrector step); updates lq in place using the stored factors */
static void riccati_delta(ws* w, int k0)
{
    const int N = w->N;
    REAL dp[NXA], dq[NU], dl[NU];
    int k, i, a, havep = 0;
    memset(dp, 0, sizeof(dp));
    for (k = N - 1; k >= k0; --k) {
        const MREAL* A = w->A[k];
        const MREAL* B = w->B[k];
        const MREAL* Lc = w->Lc[k];
        const MREAL* W = w->W[k];
        const int pk = (k > 0);
        memset(dq, 0, sizeof(dq));
        for (i = 0; i < NI; ++i)
            if (row_active(w, k, i)) row_axpy(w, k, i, w->DG[k][i], dq);
        for (i = 0; i < NU; ++i) {
            REAL v = dq[i];
            for (a = 0; a < NS; ++a) v += B[a * NU + i] * dp[a];
            if (havep && i < NF) v += dp[NS + i];
            dq[i] = v;
        }
        for (i = 0; i < NU; ++i) {
            REAL v = dq[i];
            for (a = 0; a < i; ++a) v -= Lc[i * NU + a] * dl[a];
            dl[i] = v / Lc[i * NU + i];
            w->lq[k][i] += dl[i];
        }
        {
            REAL np_[NXA];
            for (i = 0; i < NS; ++i) {
                REAL v = 0;
                for (a = 0; a < NS; ++a) v += A[a * NS + i] * dp[a];
                for (a = 0; a < NU; ++a) v -= W[a * NXA + i] * dl[a];
                np_[i] = v;
            }
            for (i = NS; i < NXA; ++i) {
                REAL v = 0;
                if (pk) for (a = 0; a < NU; ++a) v -= W[a * NXA + i] * dl[a];
                np_[i] = v;
            }
            memcpy(dp, np_, sizeof(dp));
        }
        havep = pk;
    }
}

/* forward sweep: dS, dU; then new costates LAMn backward; then dT, dZ */
static void riccati_forward(ws* w, int k0)
{
    const int N = w->N;
    int k, i, a;
    memset(w->dS[k0], 0, sizeof(REAL) * NS);   /* the state the sweep starts from is held (k0 = 0: the measured state) */
    for (k = k0; k < N; ++k) {
        const MREAL* W = w->W[k];
        const MREAL* Lc = w->Lc[k];
        REAL y[NU];
        for (i = 0; i < NU; ++i) {
            REAL v = w->lq[k][i];
            for (a = 0; a < NS; ++a) v += W[i * NXA + a] * w->dS[k][a];
            if (k > k0) for (a = 0; a < NF; ++a) v += W[i * NXA + NS + a] * w->dU[k - 1][a];   /* (the force before stage k0 is held too) */
            y[i] = -v;
        }
        for (i = NU - 1; i >= 0; --i) {
            REAL v = y[i];
            for (a = i + 1; a < NU; ++a) v -= Lc[a * NU + i] * w->dU[k][a];
            w->dU[k][i] = v / Lc[i * NU + i];
        }
        for (i = 0; i < NS; ++i) {
            REAL v = w->d[k][i];
            for (a = 0; a < NS; ++a) v += w->A[k][i * NS + a] * w->dS[k][a];
            for (a = 0; a < NU; ++a) v += w->B[k][i * NU + a] * w->dU[k][a];
            w->dS[k + 1][i] = v;
        }
    }
    /* costates: lam_N = Q_N ds_N + gs_N ; lam_k = gs_k + Q_k ds_k + S_k^T du_k + A_k^T lam_{k+1} */
    for (i = 0; i < NS; ++i) w->LAMn[N][i] = w->gs[N][i] + qdiag(w, N, i) * w->dS[N][i];
    for (k = N - 1; k >= 1 && k0 == 0; --k) {
        REAL Fsum[3] = {0, 0, 0}, Fc[2][3] = {{0, 0, 0}, {0, 0, 0}};
        int c, j;
        for (c = 0; c < 2; ++c) {
            const REAL gam = gam_of(w, c, k);
            for (j = 0; j < 4; ++j)
                for (i = 0; i < 3; ++i) Fc[c][i] += gam * w->dU[k][12 * c + 3 * j + i];
            for (i = 0; i < 3; ++i) Fsum[i] += Fc[c][i];
        }
        for (i = 0; i < NS; ++i) {
            REAL v = w->gs[k][i] + qdiag(w, k, i) * w->dS[k][i];
            for (a = 0; a < NS; ++a) v += w->A[k][a * NS + i] * w->LAMn[k + 1][a];
            w->LAMn[k][i] = v;
        }
        /* S^T du: S[f, pos_c] = gam Sx, S[f, com] = -gam Sx  ->  (S^T du)[pos_c] = Sx^T Fc, [com] = -Sx^T Fsum */
        for (i = 0; i < 3; ++i) {
            const MREAL* Sx = w->Sx[k];
            w->LAMn[k][i] -= Sx[0 * 3 + i] * Fsum[0] + Sx[1 * 3 + i] * Fsum[1] + Sx[2 * 3 + i] * Fsum[2];
            for (c = 0; c < 2; ++c)
                w->LAMn[k][9 + 3 * c + i] += Sx[0 * 3 + i] * Fc[c][0] + Sx[1 * 3 + i] * Fc[c][1] + Sx[2 * 3 + i] * Fc[c][2];
        }
    }
    memset(w->LAMn[0], 0, sizeof(REAL) * NS);
    for (k = k0; k < N; ++k)
        for (i = 0; i < NI; ++i) {
            REAL t, z, r, dt_;
            if (!row_active(w, k, i)) { w->dT[k][i] = 0; w->dZ[k][i] = 0; continue; }
            t = w->T[k][i]; z = w->Z[k][i];
            r = row_val(w, k, i, w->U[k]) + t;
            dt_ = -r - row_dot(w, k, i, w->dU[k]);
            w->dT[k][i] = dt_;
            w->dZ[k][i] = (w->CMU[k][i] - z * t) / t - (z / t) * dt_;
        }
}

/* ---------------- initialisation from x0 (reference layout) ---------------- */
static void init_iterate(ws* w, const PREAL* x0, REAL mu0)
{
    const lay* L = &w->L;
    const int N = w->N;
    const PREAL* p = w->p;
    int k, c, j, i, a;
    for (k = 0; k <= N; ++k)
        for (i = 0; i < 3; ++i) {
            w->S[k][i] = x0[L->o_com + 3 * k + i];
            w->S[k][3 + i] = x0[L->o_dcom + 3 * k + i];
            w->S[k][6 + i] = x0[L->o_h + 3 * k + i];
            for (c = 0; c < 2; ++c) w->S[k][9 + 3 * c + i] = x0[L->o_pos[c] + 3 * k + i];
        }
    for (i = 0; i < 3; ++i) { /* initial-condition rows (g rows 0..14) hold exactly */
        w->S[0][i] = p[L->p_com0 + i]; w->S[0][3 + i] = p[L->p_dcom0 + i]; w->S[0][6 + i] = p[L->p_h0 + i];
        for (c = 0; c < 2; ++c) w->S[0][9 + 3 * c + i] = p[L->p_cur[c] + i];
    }
    for (k = 0; k < N; ++k) {
        for (c = 0; c < 2; ++c) {
            const PREAL* R = p + L->p_R[c] + 9 * k;
            for (j = 0; j < 4; ++j)
                for (i = 0; i < 3; ++i) w->U[k][12 * c + 3 * j + i] = x0[L->o_f[c][j] + 3 * k + i];
            for (i = 0; i < 3; ++i) {
                const int m = 3 * c + i;
                REAL q = 0;
                if (w->qfree[k][m]) {
                    const REAL lo = w->qlo[k][m], hi = w->qhi[k][m], push = (REAL)0.01 * (hi - lo);
                    for (a = 0; a < 3; ++a) q += RM(R, a, i) * (x0[L->o_pos[c] + 3 * (k + 1) + a] - p[L->p_nom[c] + 3 * (k + 1) + a]);
                    if (q < lo + push) q = lo + push;
                    if (q > hi - push) q = hi - push;
                } else if (gam_of(w, c, k) < (REAL)0.5) {
                    q = w->qlo[k][m];
                }
                w->U[k][24 + m] = q;
            }
        }
        for (i = 0; i < NI; ++i) {
            REAL t = 1, z = 0;
            if (row_active(w, k, i)) {
                t = -row_val(w, k, i, w->U[k]);
                if (i < 32) { if (t < (REAL)1e-2) t = (REAL)1e-2; }
                z = mu0 / t;
            }
            w->T[k][i] = t; w->Z[k][i] = z;
        }
    }
    memset(w->LAM, 0, sizeof(w->LAM));
}

static void export_x(const ws* w, PREAL* x)
{
    const lay* L = &w->L;
    const int N = w->N;
    const REAL dt = (REAL)w->cfg->dt;
    int k, c, j, i;
    for (k = 0; k <= N; ++k)
        for (i = 0; i < 3; ++i) {
            x[L->o_com + 3 * k + i] = w->S[k][i];
            x[L->o_dcom + 3 * k + i] = w->S[k][3 + i];
            x[L->o_h + 3 * k + i] = w->S[k][6 + i];
            for (c = 0; c < 2; ++c) x[L->o_pos[c] + 3 * k + i] = w->S[k][9 + 3 * c + i];
        }
    for (k = 0; k < N; ++k)
        for (c = 0; c < 2; ++c) {
            const REAL gam = gam_of(w, c, k);
            for (i = 0; i < 3; ++i)
                x[L->o_vel[c] + 3 * k + i] = (PREAL)(gam < (REAL)0.5 ? (w->S[k + 1][9 + 3 * c + i] - w->S[k][9 + 3 * c + i]) / dt : 0);
            for (j = 0; j < 4; ++j)
                for (i = 0; i < 3; ++i) x[L->o_f[c][j] + 3 * k + i] = w->U[k][12 * c + 3 * j + i];
        }
}

/* ---------------- driver ---------------- */
static void step_lengths_range(const ws* w, REAL tau, int k0, int k1, REAL* ap_out, REAL* ad_out);
static void step_lengths(const ws* w, REAL tau, int k0, REAL* ap_out, REAL* ad_out) { step_lengths_range(w, tau, k0, w->N, ap_out, ad_out); }
/* rows of stages k0 .. k1-1 */
static void step_lengths_range(const ws* w, REAL tau, int k0, int k1, REAL* ap_out, REAL* ad_out)
{
    REAL ap = 1, ad = 1;
    int k, i;
    for (k = k0; k < k1; ++k)
        for (i = 0; i < NI; ++i) {
            if (!row_active(w, k, i)) continue;
            if (w->dT[k][i] < 0) { REAL a = -tau * w->T[k][i] / w->dT[k][i]; if (a < ap) ap = a; }
            if (w->dZ[k][i] < 0) { REAL a = -tau * w->Z[k][i] / w->dZ[k][i]; if (a < ad) ad = a; }
        }
    *ap_out = ap; *ad_out = ad;
}

/* ---------------- tail polish: the last stages re-solved with the state entering them held ----------------
 * Stages k0..N-1 form a small problem of their own once s_k0 and f_{k0-1} are fixed: tail_iters Newton steps on it with
 * per-row complementarity targets mu_i = mu_min min(1, z_i^2) (never below 1e-4 mu_min) -- a row keeps z_i / t_i = z_i^2 / mu_i
 * <= 1 / mu_min, the conditioning the main loop already lives with, while rows whose multiplier vanishes are followed
 * further down their path -- then one affine-scaling step (primal only), as at the end of the main loop. */
static void tail_polish(ws* w, const cmpc_ipm_opts* opt, int k0)
{
    const int N = w->N;
    int pi, k, i, last = 0, blocked = 1;
    /* tail_iters Newton steps at least; while a step was blocked (a row on its way to becoming active: the multipliers need their
     * iterations) up to six more, then the affine-scaling step */
    for (pi = 0; pi <= opt->tail_iters + 6 && !last; ++pi) {
        REAL ap, ad;
        int fail;
        last = (pi >= opt->tail_iters + 6) || (pi >= opt->tail_iters && !blocked);
        linearise(w);
        for (k = k0; k < N; ++k)
            for (i = 0; i < NI; ++i) {
                REAL z = w->Z[k][i], m = (REAL)opt->mu_min * (z < 1 ? z * z : (REAL)1);
                if (m < (REAL)1e-4 * (REAL)opt->mu_min) m = (REAL)1e-4 * (REAL)opt->mu_min;
                w->CMU[k][i] = (last || !row_active(w, k, i)) ? 0 : m;
            }
        fail = riccati_backward(w, opt->exact_hessian, k0);
        if (fail) fail = riccati_backward(w, 0, k0);
        if (fail) return;
        riccati_forward(w, k0);
        step_lengths(w, last ? (REAL)0.999 : (REAL)0.99, k0, &ap, &ad);
        blocked = ap < (REAL)0.9 || ad < (REAL)0.9;
        for (k = k0 + 1; k <= N; ++k) for (i = 0; i < NS; ++i) w->S[k][i] += ap * w->dS[k][i];
        for (k = k0; k < N; ++k) {
            for (i = 0; i < NU; ++i) w->U[k][i] += ap * w->dU[k][i];
            if (!last) for (i = 0; i < NI; ++i) { w->T[k][i] += ap * w->dT[k][i]; w->Z[k][i] += ad * w->dZ[k][i]; }
        }
    }
}

/* info: [0]=iterations [1]=kkt error [2]=final mu [3]=#GN fallbacks [4]=primal inf [5]=status */
int FN(cmpc_ref_solve_one)(const cmpc_nlp_cfg* cfg, const cmpc_ipm_opts* opt, const PREAL* p, const PREAL* x0,
                           PREAL* x, double* info)
{
    ws* w = (ws*)malloc(sizeof(ws));
    const int N = cfg->N;
    int it, k, i, gn = 0, status = 1, nrow = 0;
    REAL err = 0, es = 0, ep = 0, ec = 0, zavg = 0, mu_cur = 0;
    if (!w || N > NMAX) { free(w); return -1; }
    setup(w, cfg, p);
    init_iterate(w, x0, (REAL)opt->mu_init);
    for (k = 0; k < N; ++k) for (i = 0; i < NI; ++i) nrow += row_active(w, k, i);
    for (it = 0; it < opt->max_iter; ++it) {
        REAL ap, ad, tau, lmax = 1, mu_aff = 0, sigma, mu_t;
        int fail;
        linearise(w);
        kkt_error(w, 0, &es, &ep, &ec, &zavg);
        for (k = 1; k <= N; ++k)
            for (i = 0; i < NS; ++i) { REAL a = (REAL)fabs((double)w->LAM[k][i]); if (a > lmax) lmax = a; }
        mu_cur = 0;
        for (k = 0; k < N; ++k) for (i = 0; i < NI; ++i) if (row_active(w, k, i)) mu_cur += w->T[k][i] * w->Z[k][i];
        mu_cur /= (REAL)nrow;
        /* scaled like IPOPT's E_0 with s_d = s_max = 100: stationarity is judged at 100*tol */
        (void)lmax;
        err = es * (REAL)0.01; if (ep > err) err = ep; if (ec > err) err = ec;
        if (opt->verbose)
            printf("it %3d mu %.2e stat %.2e (rel %.2e) prim %.2e comp %.2e\n", it, (double)mu_cur, (double)es, (double)(es / lmax), (double)ep, (double)ec);
        if (err <= (REAL)opt->tol) {
            /* converged on the central path at mu ~ mu_min: one last affine-scaling (mu -> 0) step
             * extrapolates the path to its end point x(0) = x* without going through the
             * ill-conditioned small-mu systems (first-order path following, error O(mu^2)) */
            status = 0;
            if (opt->verbose >= 0) {
                memset(w->CMU, 0, sizeof(w->CMU));
                fail = riccati_backward(w, opt->exact_hessian, 0);
                if (fail) fail = riccati_backward(w, 0, 0);
                if (!fail) {
                    int k0 = N, trig = 0;
                    riccati_forward(w, 0);
                    step_lengths(w, (REAL)0.999, 0, &ap, &ad);
                    if (opt->tail_stages > 0 && opt->tail_stages < N) {
                        /* The extrapolation is exact to first order where the central path is smooth in mu.  Rows that are
                         * (nearly) degenerate -- slack and multiplier both -> 0, e.g. the friction rows of an unloaded corner --
                         * follow sqrt(mu) and the step covers half of their distance.  That matters in the last stages only,
                         * whose forces the cost barely sees (no cost on the CoM velocity, nothing after them): bias sqrt(mu / curvature).
                         * A large extrapolation step there is the symptom: then those stages are re-solved on their own. */
                        REAL fm = 1, tail = 0;
                        k0 = N - opt->tail_stages;
                        for (k = 0; k < N; ++k) for (i = 0; i < NF; ++i) { REAL a = (REAL)fabs((double)w->U[k][i]); if (a > fm) fm = a; }
                        for (k = k0; k < N; ++k) for (i = 0; i < NF; ++i) { REAL a = (REAL)fabs((double)w->dU[k][i]); if (a > tail) tail = a; }
                        trig = tail > (REAL)opt->tail_trigger * fm;   /* (the full step: a nearly degenerate row is what blocks ap) */
                        if (!trig) k0 = N;
                        /* the tail goes its own way: the step of the stages before it is limited by their own rows only */
                        else step_lengths_range(w, (REAL)0.999, 0, k0, &ap, &ad);
                    }
                    for (k = 0; k <= (trig ? k0 : N); ++k) for (i = 0; i < NS; ++i) w->S[k][i] += ap * w->dS[k][i];
                    for (k = 0; k < (trig ? k0 : N); ++k) for (i = 0; i < NU; ++i) w->U[k][i] += ap * w->dU[k][i];
                    if (trig) { tail_polish(w, opt, k0); gn += 100; }   /* (info[3] >= 100 marks a polished tail) */
                }
            }
            break;
        }
        /* predictor (affine scaling) */
        memset(w->CMU, 0, sizeof(w->CMU));
        fail = riccati_backward(w, opt->exact_hessian, 0);
        if (fail) { ++gn; fail = riccati_backward(w, 0, 0); }
        if (fail) { status = 2; break; }
        riccati_forward(w, 0);
        step_lengths(w, 1, 0, &ap, &ad);
        for (k = 0; k < N; ++k)
            for (i = 0; i < NI; ++i)
                if (row_active(w, k, i)) mu_aff += (w->T[k][i] + ap * w->dT[k][i]) * (w->Z[k][i] + ad * w->dZ[k][i]);
        mu_aff /= (REAL)nrow;
        sigma = mu_aff / mu_cur; sigma = sigma * sigma * sigma;
        mu_t = sigma * mu_cur;
        if (mu_t < (REAL)opt->mu_min) mu_t = (REAL)opt->mu_min;
        /* corrector */
        for (k = 0; k < N; ++k)
            for (i = 0; i < NI; ++i) {
                if (!row_active(w, k, i)) continue;
                w->CMU[k][i] = mu_t - w->dT[k][i] * w->dZ[k][i];
                w->DG[k][i] = w->CMU[k][i] / w->T[k][i];
            }
        riccati_delta(w, 0);
        riccati_forward(w, 0);
        tau = 1 - mu_t; if (tau < (REAL)0.99) tau = (REAL)0.99;
        step_lengths(w, tau, 0, &ap, &ad);
        for (k = 0; k <= N; ++k)
            for (i = 0; i < NS; ++i) {
                w->S[k][i] += ap * w->dS[k][i];
                w->LAM[k][i] += ap * (w->LAMn[k][i] - w->LAM[k][i]);
            }
        for (k = 0; k < N; ++k) {
            for (i = 0; i < NU; ++i) w->U[k][i] += ap * w->dU[k][i];
            for (i = 0; i < NI; ++i) { w->T[k][i] += ap * w->dT[k][i]; w->Z[k][i] += ad * w->dZ[k][i]; }
        }
        if (opt->verbose) printf("        sigma %.2e mu_t %.2e ap %.3f ad %.3f\n", (double)sigma, (double)mu_t, (double)ap, (double)ad);
    }
    export_x(w, x);
    if (info) { info[0] = it; info[1] = (double)err; info[2] = (double)mu_cur; info[3] = gn; info[4] = (double)ep; info[5] = status; }
    free(w);
    return status;
}

/* batch driver, OpenMP over problems: P[B][np], X0[B][nx] -> X[B][nx], info[B][6] */
int FN(cmpc_ref_solve_batch)(const cmpc_nlp_cfg* cfg, const cmpc_ipm_opts* opt, int B, const PREAL* P, const PREAL* X0,
                             PREAL* X, double* info, int nthreads)
{
    lay L;
    int b, bad = 0;
    lay_init(&L, cfg->N);
#pragma omp parallel for schedule(dynamic) num_threads(nthreads) reduction(+ : bad)
    for (b = 0; b < B; ++b) {
        int st = FN(cmpc_ref_solve_one)(cfg, opt, P + (size_t)b * L.np, X0 + (size_t)b * L.nx, X + (size_t)b * L.nx,
                                        info ? info + 6 * (size_t)b : NULL);
        bad += (st != 0);
    }
    return bad;
}
