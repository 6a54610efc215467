/* TEST INFRASTRUCTURE ONLY -- float64 restatement of the reference's centroidal-MPC NLP.
 *
 * Follows the CasADi-generated code the reference ships,
 *   /root/reference/src/centroidal-mpc-walking/config/robots/ergoCubGazeboV1/tmp.c
 *     f,g       nlp_fg      :12430-24715
 *     grad      nlp_grad    :24791-58842
 *     hess L    nlp_hess_l  :58926-71884   (CCS table casadi_s4 :66)
 *     f,gf,g,J  nlp_jac_fg  :71962-93971   (CCS table casadi_s5 :67)
 * with N, dt, weights, corners, mu as runtime parameters instead of baked constants.
 * Variable / parameter / constraint layout: see layout_init() below (decoded from the generated
 * code; tests/test_oracle_nlp.py pins it against oracle/_ref at N=12 for both weight sets).
 */
#include "cmpc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int N;
    /* x */
    int o_com, o_dcom, o_h, o_pos[CMPC_NC], o_vel[CMPC_NC], o_f[CMPC_NC][CMPC_NCORN], nx;
    /* p */
    int p_R[CMPC_NC], p_limA[CMPC_NC], p_limB[CMPC_NC], p_gam[CMPC_NC], p_nom[CMPC_NC],
        p_cur[CMPC_NC];
    int p_com0, p_dcom0, p_h0, p_comref, p_href, p_fext, p_text, np;
    /* g */
    int g_init, g_com, g_dcom, g_h, g_pos[CMPC_NC], g_bbox[CMPC_NC], g_fric[CMPC_NC], ng;
} layout;

static void layout_init(layout* L, int N)
{
    int o = 0, c, j;
    L->N = N;
    L->o_com = o; o += 3 * (N + 1);
    L->o_dcom = o; o += 3 * (N + 1);
    L->o_h = o; o += 3 * (N + 1);
    for (c = 0; c < CMPC_NC; ++c) {
        L->o_pos[c] = o; o += 3 * (N + 1);
        L->o_vel[c] = o; o += 3 * N;
        for (j = 0; j < CMPC_NCORN; ++j) { L->o_f[c][j] = o; o += 3 * N; }
    }
    L->nx = o;
    o = 0;
    for (c = 0; c < CMPC_NC; ++c) {
        L->p_R[c] = o; o += 9 * N;
        L->p_limA[c] = o; o += 3 * N;
        L->p_limB[c] = o; o += 3 * N;
        L->p_gam[c] = o; o += N;
        L->p_nom[c] = o; o += 3 * (N + 1);
        L->p_cur[c] = o; o += 3;
    }
    L->p_com0 = o; o += 3;
    L->p_dcom0 = o; o += 3;
    L->p_h0 = o; o += 3;
    L->p_comref = o; o += 3 * (N + 1);
    L->p_href = o; o += 3 * (N + 1);
    L->p_fext = o; o += 3 * N;
    L->p_text = o; o += 3 * N;
    L->np = o;
    o = 0;
    L->g_init = o; o += 15;
    L->g_com = o; o += 3 * N;
    L->g_dcom = o; o += 3 * N;
    L->g_h = o; o += 3 * N;
    for (c = 0; c < CMPC_NC; ++c) { L->g_pos[c] = o; o += 3 * N; }
    for (c = 0; c < CMPC_NC; ++c) {
        L->g_bbox[c] = o; o += 3 * N;
        L->g_fric[c] = o; o += 4 * CMPC_NCORN * N;
    }
    L->ng = o;
}

void cmpc_nlp_dims(const cmpc_nlp_cfg* c, int* nx, int* np, int* ng, int* nnzj, int* nnzh)
{
    layout L;
    layout_init(&L, c->N);
    if (nx) *nx = L.nx;
    if (np) *np = L.np;
    if (ng) *ng = L.ng;
    if (nnzj) *nnzj = 243 * c->N + 15;
    if (nnzh) *nnzh = 348 * c->N - 36;
}

static inline void cross(const double* a, const double* b, double* o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* friction pyramid rows: A = [[1,1,-mu],[-1,1,-mu],[-1,-1,-mu],[1,-1,-mu]] */
static const double FR_SX[4] = {1, -1, -1, 1};
static const double FR_SY[4] = {1, 1, -1, -1};

/* weight inside the square for the CoM height: (w_cz/2)(1+exp(-i)) */
static inline double wz(const cmpc_nlp_cfg* c, int i) { return 0.5 * c->w_com[2] * (1.0 + exp(-(double)i)); }

void cmpc_nlp_fg(const cmpc_nlp_cfg* cfg, const double* x, const double* p, double* f_out, double* g)
{
    layout L;
    const int N = cfg->N;
    const double dt = cfg->dt;
    int c, j, k, i;
    double f = 0.0;
    layout_init(&L, N);

    /* ---- cost ---- */
    for (k = 0; k <= N; ++k) {
        const double* com = x + L.o_com + 3 * k;
        const double* h = x + L.o_h + 3 * k;
        const double* cr = p + L.p_comref + 3 * k;
        const double* hr = p + L.p_href + 3 * k;
        double ez = wz(cfg, k) * (com[2] - cr[2]);
        for (i = 0; i < 3; ++i) f += cfg->w_h * (h[i] - hr[i]) * (h[i] - hr[i]);
        f += cfg->w_com[0] * (com[0] - cr[0]) * (com[0] - cr[0]);
        f += cfg->w_com[1] * (com[1] - cr[1]) * (com[1] - cr[1]);
        f += ez * ez;
    }
    for (c = 0; c < CMPC_NC; ++c) {
        for (k = 0; k <= N; ++k)
            for (i = 0; i < 3; ++i) {
                double e = x[L.o_pos[c] + 3 * k + i] - p[L.p_nom[c] + 3 * k + i];
                f += cfg->w_pos * e * e;
            }
        for (k = 0; k < N; ++k) {
            double gam = p[L.p_gam[c] + k];
            double mean[3] = {0, 0, 0};
            for (j = 0; j < CMPC_NCORN; ++j)
                for (i = 0; i < 3; ++i) mean[i] += 0.25 * x[L.o_f[c][j] + 3 * k + i];
            for (j = 0; j < CMPC_NCORN; ++j)
                for (i = 0; i < 3; ++i) {
                    double e = x[L.o_f[c][j] + 3 * k + i] - gam * mean[i];
                    f += cfg->w_sym * e * e;
                    if (k + 1 < N) {
                        double d = x[L.o_f[c][j] + 3 * (k + 1) + i] - x[L.o_f[c][j] + 3 * k + i];
                        f += cfg->w_rate[i] * d * d;
                    }
                }
        }
    }
    if (f_out) *f_out = f;
    if (!g) return;

    /* ---- constraints ---- */
    for (i = 0; i < 3; ++i) {
        g[L.g_init + i] = x[L.o_com + i];
        g[L.g_init + 3 + i] = x[L.o_dcom + i];
        g[L.g_init + 6 + i] = x[L.o_h + i];
        g[L.g_init + 9 + i] = x[L.o_pos[0] + i];
        g[L.g_init + 12 + i] = x[L.o_pos[1] + i];
    }
    for (k = 0; k < N; ++k) {
        const double* com = x + L.o_com + 3 * k;
        const double* dcom = x + L.o_dcom + 3 * k;
        const double* h = x + L.o_h + 3 * k;
        double acc[3] = {0, 0, -cfg->gravity};
        double tor[3];
        for (i = 0; i < 3; ++i) {
            acc[i] += p[L.p_fext + 3 * k + i];
            tor[i] = p[L.p_text + 3 * k + i];
        }
        for (c = 0; c < CMPC_NC; ++c) {
            const double* R = p + L.p_R[c] + 9 * k; /* col-major 3x3 */
            const double* pos = x + L.o_pos[c] + 3 * k;
            const double* posn = x + L.o_pos[c] + 3 * (k + 1);
            const double* vel = x + L.o_vel[c] + 3 * k;
            const double* nomn = p + L.p_nom[c] + 3 * (k + 1);
            double gam = p[L.p_gam[c] + k];
            double d[3];
            for (j = 0; j < CMPC_NCORN; ++j) {
                const double* fc = x + L.o_f[c][j] + 3 * k;
                const double* cn = cfg->corners[c][j];
                double r[3], t[3], fl[3];
                for (i = 0; i < 3; ++i)
                    r[i] = R[i] * cn[0] + R[3 + i] * cn[1] + R[6 + i] * cn[2] + pos[i] - com[i];
                cross(r, fc, t);
                for (i = 0; i < 3; ++i) {
                    acc[i] += gam * fc[i];
                    tor[i] += gam * t[i];
                }
                /* friction rows on R^T f */
                for (i = 0; i < 3; ++i) fl[i] = R[3 * i] * fc[0] + R[3 * i + 1] * fc[1] + R[3 * i + 2] * fc[2];
                for (i = 0; i < 4; ++i)
                    g[L.g_fric[c] + 16 * k + 4 * j + i] = FR_SX[i] * fl[0] + FR_SY[i] * fl[1] - cfg->mu * fl[2];
            }
            for (i = 0; i < 3; ++i) {
                g[L.g_pos[c] + 3 * k + i] = posn[i] - (pos[i] + dt * (1.0 - gam) * vel[i]);
                d[i] = posn[i] - nomn[i];
            }
            for (i = 0; i < 3; ++i)
                g[L.g_bbox[c] + 3 * k + i] = R[3 * i] * d[0] + R[3 * i + 1] * d[1] + R[3 * i + 2] * d[2];
        }
        for (i = 0; i < 3; ++i) {
            g[L.g_com + 3 * k + i] = x[L.o_com + 3 * (k + 1) + i] - (com[i] + dt * dcom[i]);
            g[L.g_dcom + 3 * k + i] = x[L.o_dcom + 3 * (k + 1) + i] - (dcom[i] + dt * acc[i]);
            g[L.g_h + 3 * k + i] = x[L.o_h + 3 * (k + 1) + i] - (h[i] + dt * tor[i]);
        }
    }
}

void cmpc_nlp_grad_f(const cmpc_nlp_cfg* cfg, const double* x, const double* p, double* gf)
{
    layout L;
    const int N = cfg->N;
    int c, j, k, i;
    layout_init(&L, N);
    memset(gf, 0, sizeof(double) * (size_t)L.nx);
    for (k = 0; k <= N; ++k) {
        double w = wz(cfg, k);
        for (i = 0; i < 3; ++i)
            gf[L.o_h + 3 * k + i] = 2.0 * cfg->w_h * (x[L.o_h + 3 * k + i] - p[L.p_href + 3 * k + i]);
        for (i = 0; i < 2; ++i)
            gf[L.o_com + 3 * k + i] = 2.0 * cfg->w_com[i] * (x[L.o_com + 3 * k + i] - p[L.p_comref + 3 * k + i]);
        gf[L.o_com + 3 * k + 2] = 2.0 * w * w * (x[L.o_com + 3 * k + 2] - p[L.p_comref + 3 * k + 2]);
    }
    for (c = 0; c < CMPC_NC; ++c) {
        for (k = 0; k <= N; ++k)
            for (i = 0; i < 3; ++i)
                gf[L.o_pos[c] + 3 * k + i] = 2.0 * cfg->w_pos * (x[L.o_pos[c] + 3 * k + i] - p[L.p_nom[c] + 3 * k + i]);
        for (k = 0; k < N; ++k) {
            double gam = p[L.p_gam[c] + k];
            for (i = 0; i < 3; ++i) {
                double mean = 0, esum = 0, e[CMPC_NCORN];
                for (j = 0; j < CMPC_NCORN; ++j) mean += 0.25 * x[L.o_f[c][j] + 3 * k + i];
                for (j = 0; j < CMPC_NCORN; ++j) {
                    e[j] = x[L.o_f[c][j] + 3 * k + i] - gam * mean;
                    esum += e[j];
                }
                for (j = 0; j < CMPC_NCORN; ++j) {
                    int ix = L.o_f[c][j] + 3 * k + i;
                    gf[ix] += 2.0 * cfg->w_sym * (e[j] - 0.25 * gam * esum);
                    if (k + 1 < N) {
                        double d = x[ix + 3] - x[ix];
                        gf[ix] -= 2.0 * cfg->w_rate[i] * d;
                        gf[ix + 3] += 2.0 * cfg->w_rate[i] * d;
                    }
                }
            }
        }
    }
}

#define PUT(r, cc, v) do { row[n] = (r); col[n] = (cc); val[n] = (v); ++n; } while (0)

int cmpc_nlp_jac(const cmpc_nlp_cfg* cfg, const double* x, const double* p, int* row, int* col, double* val)
{
    layout L;
    const int N = cfg->N;
    const double dt = cfg->dt;
    int c, j, k, i, a, n = 0;
    layout_init(&L, N);
    for (i = 0; i < 3; ++i) {
        PUT(L.g_init + i, L.o_com + i, 1.0);
        PUT(L.g_init + 3 + i, L.o_dcom + i, 1.0);
        PUT(L.g_init + 6 + i, L.o_h + i, 1.0);
        PUT(L.g_init + 9 + i, L.o_pos[0] + i, 1.0);
        PUT(L.g_init + 12 + i, L.o_pos[1] + i, 1.0);
    }
    for (k = 0; k < N; ++k) {
        const double* com = x + L.o_com + 3 * k;
        double Fsum[3] = {0, 0, 0}; /* sum_c gam_c sum_j f_cj */
        for (i = 0; i < 3; ++i) {
            PUT(L.g_com + 3 * k + i, L.o_com + 3 * (k + 1) + i, 1.0);
            PUT(L.g_com + 3 * k + i, L.o_com + 3 * k + i, -1.0);
            PUT(L.g_com + 3 * k + i, L.o_dcom + 3 * k + i, -dt);
            PUT(L.g_dcom + 3 * k + i, L.o_dcom + 3 * (k + 1) + i, 1.0);
            PUT(L.g_dcom + 3 * k + i, L.o_dcom + 3 * k + i, -1.0);
            PUT(L.g_h + 3 * k + i, L.o_h + 3 * (k + 1) + i, 1.0);
            PUT(L.g_h + 3 * k + i, L.o_h + 3 * k + i, -1.0);
        }
        for (c = 0; c < CMPC_NC; ++c) {
            const double* R = p + L.p_R[c] + 9 * k;
            const double* pos = x + L.o_pos[c] + 3 * k;
            double gam = p[L.p_gam[c] + k];
            double Fc[3] = {0, 0, 0};
            for (j = 0; j < CMPC_NCORN; ++j) {
                const double* fc = x + L.o_f[c][j] + 3 * k;
                const double* cn = cfg->corners[c][j];
                double r[3];
                for (i = 0; i < 3; ++i) {
                    r[i] = R[i] * cn[0] + R[3 + i] * cn[1] + R[6 + i] * cn[2] + pos[i] - com[i];
                    Fc[i] += fc[i];
                    PUT(L.g_dcom + 3 * k + i, L.o_f[c][j] + 3 * k + i, -dt * gam);
                }
                /* d(g_h)/d f = -dt gam [r]x ; [r]x = [[0,-r2,r1],[r2,0,-r0],[-r1,r0,0]] */
                PUT(L.g_h + 3 * k + 0, L.o_f[c][j] + 3 * k + 1, -dt * gam * (-r[2]));
                PUT(L.g_h + 3 * k + 0, L.o_f[c][j] + 3 * k + 2, -dt * gam * (r[1]));
                PUT(L.g_h + 3 * k + 1, L.o_f[c][j] + 3 * k + 0, -dt * gam * (r[2]));
                PUT(L.g_h + 3 * k + 1, L.o_f[c][j] + 3 * k + 2, -dt * gam * (-r[0]));
                PUT(L.g_h + 3 * k + 2, L.o_f[c][j] + 3 * k + 0, -dt * gam * (-r[1]));
                PUT(L.g_h + 3 * k + 2, L.o_f[c][j] + 3 * k + 1, -dt * gam * (r[0]));
                /* friction rows */
                for (i = 0; i < 4; ++i)
                    for (a = 0; a < 3; ++a) /* d/d f_a of sx*(R^T f)_0 + sy*(R^T f)_1 - mu (R^T f)_2 ; (R^T f)_m = sum_a R[a,m] f_a = R[3m+a] f_a */
                        PUT(L.g_fric[c] + 16 * k + 4 * j + i, L.o_f[c][j] + 3 * k + a,
                            FR_SX[i] * R[a] + FR_SY[i] * R[3 + a] - cfg->mu * R[6 + a]);
            }
            /* d(g_h)/d pos_c = +dt gam [Fc]x */
            PUT(L.g_h + 3 * k + 0, L.o_pos[c] + 3 * k + 1, dt * gam * (-Fc[2]));
            PUT(L.g_h + 3 * k + 0, L.o_pos[c] + 3 * k + 2, dt * gam * (Fc[1]));
            PUT(L.g_h + 3 * k + 1, L.o_pos[c] + 3 * k + 0, dt * gam * (Fc[2]));
            PUT(L.g_h + 3 * k + 1, L.o_pos[c] + 3 * k + 2, dt * gam * (-Fc[0]));
            PUT(L.g_h + 3 * k + 2, L.o_pos[c] + 3 * k + 0, dt * gam * (-Fc[1]));
            PUT(L.g_h + 3 * k + 2, L.o_pos[c] + 3 * k + 1, dt * gam * (Fc[0]));
            for (i = 0; i < 3; ++i) Fsum[i] += gam * Fc[i];
            for (i = 0; i < 3; ++i) {
                PUT(L.g_pos[c] + 3 * k + i, L.o_pos[c] + 3 * (k + 1) + i, 1.0);
                PUT(L.g_pos[c] + 3 * k + i, L.o_pos[c] + 3 * k + i, -1.0);
                PUT(L.g_pos[c] + 3 * k + i, L.o_vel[c] + 3 * k + i, -dt * (1.0 - gam));
                for (a = 0; a < 3; ++a) /* (R^T d)_i = sum_a R[a,i] d_a */
                    PUT(L.g_bbox[c] + 3 * k + i, L.o_pos[c] + 3 * (k + 1) + a, R[3 * i + a]);
            }
        }
        /* d(g_h)/d com = -dt [Fsum]x */
        PUT(L.g_h + 3 * k + 0, L.o_com + 3 * k + 1, -dt * (-Fsum[2]));
        PUT(L.g_h + 3 * k + 0, L.o_com + 3 * k + 2, -dt * (Fsum[1]));
        PUT(L.g_h + 3 * k + 1, L.o_com + 3 * k + 0, -dt * (Fsum[2]));
        PUT(L.g_h + 3 * k + 1, L.o_com + 3 * k + 2, -dt * (-Fsum[0]));
        PUT(L.g_h + 3 * k + 2, L.o_com + 3 * k + 0, -dt * (-Fsum[1]));
        PUT(L.g_h + 3 * k + 2, L.o_com + 3 * k + 1, -dt * (Fsum[0]));
    }
    return n;
}

int cmpc_nlp_hess(const cmpc_nlp_cfg* cfg, const double* x, const double* p, double lam_f,
                  const double* lam_g, int* row, int* col, double* val)
{
    layout L;
    const int N = cfg->N;
    const double dt = cfg->dt;
    int c, j, l, k, i, n = 0;
    (void)x;
    layout_init(&L, N);
    for (k = 0; k <= N; ++k) {
        double w = wz(cfg, k);
        PUT(L.o_com + 3 * k, L.o_com + 3 * k, lam_f * 2.0 * cfg->w_com[0]);
        PUT(L.o_com + 3 * k + 1, L.o_com + 3 * k + 1, lam_f * 2.0 * cfg->w_com[1]);
        PUT(L.o_com + 3 * k + 2, L.o_com + 3 * k + 2, lam_f * 2.0 * w * w);
        for (i = 0; i < 3; ++i) PUT(L.o_h + 3 * k + i, L.o_h + 3 * k + i, lam_f * 2.0 * cfg->w_h);
        for (c = 0; c < CMPC_NC; ++c)
            for (i = 0; i < 3; ++i) PUT(L.o_pos[c] + 3 * k + i, L.o_pos[c] + 3 * k + i, lam_f * 2.0 * cfg->w_pos);
    }
    for (k = 0; k < N; ++k) {
        const double* lh = lam_g + L.g_h + 3 * k;
        for (c = 0; c < CMPC_NC; ++c) {
            double gam = p[L.p_gam[c] + k];
            double offd = -0.25 * gam * (2.0 - gam);
            for (j = 0; j < CMPC_NCORN; ++j) {
                int fj = L.o_f[c][j] + 3 * k;
                /* symmetry + rate, diagonal and cross-corner */
                for (l = 0; l < CMPC_NCORN; ++l)
                    for (i = 0; i < 3; ++i) {
                        double v = 2.0 * cfg->w_sym * ((j == l ? 1.0 : 0.0) + offd);
                        if (j == l) {
                            int nr = (k > 0) + (k + 1 < N);
                            v += 2.0 * cfg->w_rate[i] * nr;
                        }
                        PUT(fj + i, L.o_f[c][l] + 3 * k + i, lam_f * v);
                    }
                if (k + 1 < N)
                    for (i = 0; i < 3; ++i) {
                        PUT(fj + i, fj + 3 + i, -lam_f * 2.0 * cfg->w_rate[i]);
                        PUT(fj + 3 + i, fj + i, -lam_f * 2.0 * cfg->w_rate[i]);
                    }
                /* bilinear momentum term: H[f, pos] = -dt gam [lh]x, H[f, com] = +dt gam [lh]x */
                {
                    const double S[3][3] = {{0, -lh[2], lh[1]}, {lh[2], 0, -lh[0]}, {-lh[1], lh[0], 0}};
                    int a, b;
                    for (a = 0; a < 3; ++a)
                        for (b = 0; b < 3; ++b) {
                            if (a == b) continue;
                            PUT(fj + a, L.o_pos[c] + 3 * k + b, -dt * gam * S[a][b]);
                            PUT(L.o_pos[c] + 3 * k + b, fj + a, -dt * gam * S[a][b]);
                            PUT(fj + a, L.o_com + 3 * k + b, dt * gam * S[a][b]);
                            PUT(L.o_com + 3 * k + b, fj + a, dt * gam * S[a][b]);
                        }
                }
            }
        }
    }
    return n;
}

/* nlp_grad (tmp.c:24791-58842): gradient of gamma = lam_f f + lam_g^T g with respect to x and to p.
 *   grad_gamma_x = lam_f grad f + J^T lam_g                        (from the two functions above)
 *   grad_gamma_p: closed forms per parameter block.  limA, limB, currentPos, com0, dcom0, h0 do not enter f or g
 *   (CasADi's Opti turns them into bounds), so their entries are zero. */
void cmpc_nlp_grad(const cmpc_nlp_cfg* cfg, const double* x, const double* p, double lam_f, const double* lam_g,
                   double* gx, double* gp)
{
    layout L;
    const int N = cfg->N;
    const double dt = cfg->dt;
    int c, j, k, i, a, m, n;
    layout_init(&L, N);
    if (gx) {
        const int nnzj = 243 * N + 15;
        int* row = (int*)malloc(sizeof(int) * (size_t)nnzj * 2);
        int* col = row + nnzj;
        double* val = (double*)malloc(sizeof(double) * (size_t)nnzj);
        cmpc_nlp_grad_f(cfg, x, p, gx);
        for (i = 0; i < L.nx; ++i) gx[i] *= lam_f;
        n = cmpc_nlp_jac(cfg, x, p, row, col, val);
        for (i = 0; i < n; ++i) gx[col[i]] += val[i] * lam_g[row[i]];
        free(row);
        free(val);
    }
    if (!gp) return;
    memset(gp, 0, sizeof(double) * (size_t)L.np);
    /* cost: references and nominal positions enter as (x - p)^2 */
    for (k = 0; k <= N; ++k) {
        double w = wz(cfg, k);
        for (i = 0; i < 3; ++i)
            gp[L.p_href + 3 * k + i] = -lam_f * 2.0 * cfg->w_h * (x[L.o_h + 3 * k + i] - p[L.p_href + 3 * k + i]);
        for (i = 0; i < 2; ++i)
            gp[L.p_comref + 3 * k + i] = -lam_f * 2.0 * cfg->w_com[i] * (x[L.o_com + 3 * k + i] - p[L.p_comref + 3 * k + i]);
        gp[L.p_comref + 3 * k + 2] = -lam_f * 2.0 * w * w * (x[L.o_com + 3 * k + 2] - p[L.p_comref + 3 * k + 2]);
        for (c = 0; c < CMPC_NC; ++c)
            for (i = 0; i < 3; ++i)
                gp[L.p_nom[c] + 3 * k + i] = -lam_f * 2.0 * cfg->w_pos * (x[L.o_pos[c] + 3 * k + i] - p[L.p_nom[c] + 3 * k + i]);
    }
    for (k = 0; k < N; ++k) {
        const double* com = x + L.o_com + 3 * k;
        const double* ld = lam_g + L.g_dcom + 3 * k;
        const double* lh = lam_g + L.g_h + 3 * k;
        for (i = 0; i < 3; ++i) {
            gp[L.p_fext + 3 * k + i] = -dt * ld[i];
            gp[L.p_text + 3 * k + i] = -dt * lh[i];
        }
        for (c = 0; c < CMPC_NC; ++c) {
            const double* R = p + L.p_R[c] + 9 * k; /* col-major: R(row, col) = R[3 col + row] */
            const double* pos = x + L.o_pos[c] + 3 * k;
            const double* posn = x + L.o_pos[c] + 3 * (k + 1);
            const double* vel = x + L.o_vel[c] + 3 * k;
            const double* nomn = p + L.p_nom[c] + 3 * (k + 1);
            const double* lp = lam_g + L.g_pos[c] + 3 * k;
            const double* lb = lam_g + L.g_bbox[c] + 3 * k;
            const double gam = p[L.p_gam[c] + k];
            double* gR = gp + L.p_R[c] + 9 * k;
            double dgam = 0.0, mean[3] = {0, 0, 0}, esum[3] = {0, 0, 0}, d[3];
            for (j = 0; j < CMPC_NCORN; ++j)
                for (i = 0; i < 3; ++i) mean[i] += 0.25 * x[L.o_f[c][j] + 3 * k + i];
            for (j = 0; j < CMPC_NCORN; ++j) {
                const double* fc = x + L.o_f[c][j] + 3 * k;
                const double* cn = cfg->corners[c][j];
                const double* lf = lam_g + L.g_fric[c] + 16 * k + 4 * j;
                double r[3], t[3], fxl[3], coef[3] = {0, 0, 0};
                for (i = 0; i < 3; ++i) {
                    r[i] = R[i] * cn[0] + R[3 + i] * cn[1] + R[6 + i] * cn[2] + pos[i] - com[i];
                    esum[i] += fc[i] - gam * mean[i];
                }
                cross(r, fc, t);
                cross(fc, lh, fxl);
                /* rows g_dcom, g_h: -dt gam (f, r x f) */
                for (i = 0; i < 3; ++i) dgam += -dt * (ld[i] * fc[i] + lh[i] * t[i]);
                /* lam_h . (r x f) = r . (f x lam_h), r = R cn + ...: d/dR(row, col) = cn[col] (f x lam_h)[row] */
                for (m = 0; m < 3; ++m)
                    for (a = 0; a < 3; ++a) gR[3 * m + a] += -dt * gam * cn[m] * fxl[a];
                /* friction rows: sum_face lam (sx, sy, -mu)_m (R^T f)_m, (R^T f)_m = sum_a R(a, m) f_a */
                for (i = 0; i < 4; ++i) {
                    coef[0] += lf[i] * FR_SX[i];
                    coef[1] += lf[i] * FR_SY[i];
                    coef[2] += -lf[i] * cfg->mu;
                }
                for (m = 0; m < 3; ++m)
                    for (a = 0; a < 3; ++a) gR[3 * m + a] += coef[m] * fc[a];
            }
            /* symmetry cost: sum_j sum_i w_sym (f_ji - gam mean_i)^2 */
            for (i = 0; i < 3; ++i) dgam += -lam_f * 2.0 * cfg->w_sym * mean[i] * esum[i];
            /* foot dynamics row: pos+ - (pos + dt (1 - gam) vel) */
            for (i = 0; i < 3; ++i) dgam += dt * lp[i] * vel[i];
            gp[L.p_gam[c] + k] = dgam;
            /* bounding-box rows: (R^T d)_i, d = pos+ - nominal+ */
            for (i = 0; i < 3; ++i) d[i] = posn[i] - nomn[i];
            for (i = 0; i < 3; ++i)
                for (a = 0; a < 3; ++a) {
                    gR[3 * i + a] += lb[i] * d[a];
                    gp[L.p_nom[c] + 3 * (k + 1) + a] += -lb[i] * R[3 * i + a];
                }
        }
    }
}
