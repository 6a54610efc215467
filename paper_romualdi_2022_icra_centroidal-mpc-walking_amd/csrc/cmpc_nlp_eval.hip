// Batched evaluation of the reference NLP's callbacks on gfx950: f, g, grad f, jac g, hess L.
//
// These are the five functions IPOPT calls through the reference's CasADi-generated code
// (src/centroidal-mpc-walking/config/robots/ergoCubGazeboV1/tmp.c: nlp_fg :12430, nlp_grad
// :24791, nlp_hess_l :58926, nlp_jac_fg :71962), for a batch and for any horizon N.  Jacobian and
// Hessian non-zeros come out in the reference's CCS order (casadi_s5 / casadi_s4, tmp.c:66-67):
// cmpc_nlp_sparsity() builds that order (sort by column, then row) from the structural pattern.
//
// One workgroup per problem: x, p and lam_g are staged in LDS (12 KB at N=20), every output
// element is one thread's closed-form expression, output rows are written coalesced.  The path is
// HBM-write bound (~45 KB out per problem when the Hessian is requested).
#include "cmpc_contacts.h"
#include "cmpc_device.h"

#include <algorithm>
#include <vector>

namespace {

// ---- non-zero descriptors: kind | k<<4 | c<<10 | j<<11 | a<<13 | b<<15 | face<<17 ----
enum : int {
    J_ONE = 0, J_MONE, J_MDT, J_DCOM_F, J_H_F, J_H_POS, J_H_COM, J_POS_VEL, J_BBOX, J_FRIC,
    H_COMX = 0, H_COMY, H_COMZ, H_H, H_POS, H_FF, H_RATE, H_F_POS, H_F_COM
};
__host__ __device__ inline int mk(int kind, int k, int c = 0, int j = 0, int a = 0, int b = 0, int face = 0)
{
    return kind | (k << 4) | (c << 10) | (j << 11) | (a << 13) | (b << 15) | (face << 17);
}

struct GLay {  // g-row offsets (SURVEY 8a-NLP 'Constraints')
    int g_init, g_com, g_dcom, g_h, g_pos[2], g_bbox[2], g_fric[2];
};
__host__ __device__ inline void glay_init(GLay& G, int N)
{
    int o = 0;
    G.g_init = o; o += 15;
    G.g_com = o; o += 3 * N; G.g_dcom = o; o += 3 * N; G.g_h = o; o += 3 * N;
    G.g_pos[0] = o; o += 3 * N; G.g_pos[1] = o; o += 3 * N;
    for (int c = 0; c < 2; ++c) { G.g_bbox[c] = o; o += 3 * N; G.g_fric[c] = o; o += 16 * N; }
}

struct Trip { int row, col, desc; };

// skew-matrix entry [v]x(a,b), a != b
__device__ inline float skew(const float* v, int a, int b)
{
    const int o = 3 - a - b;
    return ((b - a + 3) % 3 == 1) ? -v[o] : v[o];
}

__global__ __launch_bounds__(256) void cmpc_nlp_eval_kernel(CmpcParams kp, const float* __restrict__ X, const float* __restrict__ P,
                                                            const float* __restrict__ LamG, float lam_f, float* __restrict__ F,
                                                            float* __restrict__ Gout, float* __restrict__ GradF, float* __restrict__ Jac,
                                                            float* __restrict__ Hess, const int* __restrict__ jdesc,
                                                            const int* __restrict__ hdesc, int nnzj, int nnzh)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, NT = 256;
    const int b = blockIdx.x;
    const int N = kp.N;
    CmpcConsts& K = *reinterpret_cast<CmpcConsts*>(smem);
    {
        const int* src = reinterpret_cast<const int*>(kp.kc);
        int* dst = reinterpret_cast<int*>(smem);
        for (int e = tid; e < (int)(sizeof(CmpcConsts) / 4); e += NT) dst[e] = src[e];
    }
    const CmpcIdx L{N};
    GLay G;
    glay_init(G, N);
    float* x = reinterpret_cast<float*>(smem + ((sizeof(CmpcConsts) + 15) & ~15));
    float* p = x + ((L.nx() + 3) & ~3);
    float* lam = p + ((L.np() + 3) & ~3);
    float* red = lam + ((L.ng() + 3) & ~3);
    for (int e = tid; e < L.nx(); e += NT) x[e] = X[(size_t)b * L.nx() + e];
    for (int e = tid; e < L.np(); e += NT) p[e] = P[(size_t)b * L.np() + e];
    if (LamG) for (int e = tid; e < L.ng(); e += NT) lam[e] = LamG[(size_t)b * L.ng() + e];
    __syncthreads();
    const float dt = K.dt;

    auto gam = [&](int c, int k) { return p[L.pGam(c) + k]; };
    auto rvec = [&](int c, int j, int k, float* r) {
        const float* R = p + L.pR(c) + 9 * k;
        const float* cn = K.corners + 12 * c + 3 * j;
        for (int i = 0; i < 3; ++i)
            r[i] = R[i] * cn[0] + R[3 + i] * cn[1] + R[6 + i] * cn[2] + x[L.oPos(c) + 3 * k + i] - x[L.oCom() + 3 * k + i];
    };
    auto fcsum = [&](int c, int k, float* Fc) {
        for (int i = 0; i < 3; ++i)
            Fc[i] = x[L.oF(c, 0) + 3 * k + i] + x[L.oF(c, 1) + 3 * k + i] + x[L.oF(c, 2) + 3 * k + i] + x[L.oF(c, 3) + 3 * k + i];
    };

    // ---------------- f ----------------
    if (F) {
        float acc = 0.f;
        for (int e = tid; e < 3 * (N + 1); e += NT) {
            const int k = e / 3, i = e % 3;
            const float ec = x[L.oCom() + e] - p[L.pComref() + e];
            acc += (i == 0 ? K.w_com0 : (i == 1 ? K.w_com1 : 0.5f * K.wz2[k])) * ec * ec;
            const float eh = x[L.oH() + e] - p[L.pHref() + e];
            acc += K.w_h * eh * eh;
            for (int c = 0; c < 2; ++c) {
                const float ep = x[L.oPos(c) + e] - p[L.pNom(c) + e];
                acc += K.w_pos * ep * ep;
            }
        }
        for (int e = tid; e < 2 * N * 3; e += NT) {
            const int c = e / (3 * N), k = (e % (3 * N)) / 3, i = e % 3;
            const float g = gam(c, k);
            float mean = 0.f;
            for (int j = 0; j < 4; ++j) mean += 0.25f * x[L.oF(c, j) + 3 * k + i];
            for (int j = 0; j < 4; ++j) {
                const float fv = x[L.oF(c, j) + 3 * k + i];
                const float es = fv - g * mean;
                acc += K.w_sym * es * es;
                if (k + 1 < N) {
                    const float d = x[L.oF(c, j) + 3 * (k + 1) + i] - fv;
                    acc += 0.5f * K.D[i] * d * d;
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if ((tid & 63) == 0) red[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) F[b] = red[0] + red[1] + red[2] + red[3];
    }
    // ---------------- g ----------------
    if (Gout) {
        float* g = Gout + (size_t)b * L.ng();
        for (int r = tid; r < L.ng(); r += NT) {
            float v;
            if (r < 15) {
                const int i = r % 3;
                v = r < 3 ? x[L.oCom() + i] : r < 6 ? x[L.oDcom() + i] : r < 9 ? x[L.oH() + i] : r < 12 ? x[L.oPos(0) + i] : x[L.oPos(1) + i];
            } else if (r < G.g_dcom) {
                const int e = r - G.g_com, k = e / 3;
                v = x[L.oCom() + e + 3] - (x[L.oCom() + e] + dt * x[L.oDcom() + e]);
                (void)k;
            } else if (r < G.g_h) {
                const int e = r - G.g_dcom, k = e / 3, i = e % 3;
                float acc = p[L.pFext() + e] - (i == 2 ? K.grav : 0.f);
                for (int c = 0; c < 2; ++c) {
                    float Fc[3];
                    fcsum(c, k, Fc);
                    acc += gam(c, k) * Fc[i];
                }
                v = x[L.oDcom() + e + 3] - (x[L.oDcom() + e] + dt * acc);
            } else if (r < G.g_pos[0]) {
                const int e = r - G.g_h, k = e / 3, i = e % 3, a1 = (i + 1) % 3, a2 = (i + 2) % 3;
                float tor = p[L.pText() + e];
                for (int c = 0; c < 2; ++c) {
                    float t = 0.f;
                    for (int j = 0; j < 4; ++j) {
                        float rr[3];
                        rvec(c, j, k, rr);
                        const float* f = x + L.oF(c, j) + 3 * k;
                        t += rr[a1] * f[a2] - rr[a2] * f[a1];
                    }
                    tor += gam(c, k) * t;
                }
                v = x[L.oH() + e + 3] - (x[L.oH() + e] + dt * tor);
            } else if (r < G.g_bbox[0]) {
                const int c = r < G.g_pos[1] ? 0 : 1, e = r - G.g_pos[c], k = e / 3;
                v = x[L.oPos(c) + e + 3] - (x[L.oPos(c) + e] + dt * (1.f - gam(c, k)) * x[L.oVel(c) + e]);
            } else {
                const int c = r < G.g_bbox[1] ? 0 : 1;
                if (r < G.g_fric[c]) {
                    const int e = r - G.g_bbox[c], k = e / 3, i = e % 3;
                    const float* R = p + L.pR(c) + 9 * k;
                    v = 0.f;
                    for (int a = 0; a < 3; ++a) v += R[3 * i + a] * (x[L.oPos(c) + 3 * (k + 1) + a] - p[L.pNom(c) + 3 * (k + 1) + a]);
                } else {
                    const int e = r - G.g_fric[c], k = e / 16, j = (e % 16) / 4, face = e % 4;
                    const float* R = p + L.pR(c) + 9 * k;
                    const float* f = x + L.oF(c, j) + 3 * k;
                    const float sx = (face == 0 || face == 3) ? 1.f : -1.f, sy = face < 2 ? 1.f : -1.f;
                    float fl[3];
                    for (int m = 0; m < 3; ++m) fl[m] = R[3 * m] * f[0] + R[3 * m + 1] * f[1] + R[3 * m + 2] * f[2];
                    v = sx * fl[0] + sy * fl[1] - K.mu_fr * fl[2];
                }
            }
            g[r] = v;
        }
    }
    // ---------------- grad f ----------------
    if (GradF) {
        float* gf = GradF + (size_t)b * L.nx();
        for (int e = tid; e < L.nx(); e += NT) {
            float v = 0.f;
            if (e < L.oDcom()) {
                const int k = e / 3, i = e % 3;
                v = (i == 0 ? 2.f * K.w_com0 : (i == 1 ? 2.f * K.w_com1 : K.wz2[k])) * (x[e] - p[L.pComref() + e]);
            } else if (e < L.oH()) {
                v = 0.f;
            } else if (e < L.oPos(0)) {
                v = 2.f * K.w_h * (x[e] - p[L.pHref() + e - L.oH()]);
            } else {
                const int c = e < L.oPos(1) ? 0 : 1;
                const int e2 = e - L.oPos(c);
                if (e2 < 3 * (N + 1)) v = 2.f * K.w_pos * (x[e] - p[L.pNom(c) + e2]);
                else if (e2 < 3 * (N + 1) + 3 * N) v = 0.f;
                else {
                    const int e3 = e2 - 3 * (N + 1) - 3 * N, j = e3 / (3 * N), k = (e3 % (3 * N)) / 3, i = e3 % 3;
                    const float g = gam(c, k);
                    float mean = 0.f;
                    for (int l = 0; l < 4; ++l) mean += 0.25f * x[L.oF(c, l) + 3 * k + i];
                    const float es = x[e] - g * mean, esum = 4.f * mean * (1.f - g);
                    v = 2.f * K.w_sym * (es - 0.25f * g * esum);
                    if (k > 0) v += K.D[i] * (x[e] - x[e - 3]);
                    if (k + 1 < N) v -= K.D[i] * (x[e + 3] - x[e]);
                    (void)j;
                }
            }
            gf[e] = v;
        }
    }
    // ---------------- jac g (CCS order) ----------------
    if (Jac) {
        float* jo = Jac + (size_t)b * nnzj;
        for (int e = tid; e < nnzj; e += NT) {
            const int d = jdesc[e];
            const int kind = d & 15, k = (d >> 4) & 63, c = (d >> 10) & 1, j = (d >> 11) & 3, a = (d >> 13) & 3, bb = (d >> 15) & 3,
                      face = (d >> 17) & 3;
            float v;
            switch (kind) {
                case J_ONE: v = 1.f; break;
                case J_MONE: v = -1.f; break;
                case J_MDT: v = -dt; break;
                case J_DCOM_F: v = -dt * gam(c, k); break;
                case J_H_F: {
                    float rr[3];
                    rvec(c, j, k, rr);
                    v = -dt * gam(c, k) * skew(rr, a, bb);
                } break;
                case J_H_POS: {
                    float Fc[3];
                    fcsum(c, k, Fc);
                    v = dt * gam(c, k) * skew(Fc, a, bb);
                } break;
                case J_H_COM: {
                    float F0[3], F1[3], Fs[3];
                    fcsum(0, k, F0);
                    fcsum(1, k, F1);
                    for (int i = 0; i < 3; ++i) Fs[i] = gam(0, k) * F0[i] + gam(1, k) * F1[i];
                    v = -dt * skew(Fs, a, bb);
                } break;
                case J_POS_VEL: v = -dt * (1.f - gam(c, k)); break;
                case J_BBOX: v = p[L.pR(c) + 9 * k + 3 * a + bb]; break;  // a = bbox row i, bb = pos component
                default: {  // J_FRIC: bb = force component
                    const float* R = p + L.pR(c) + 9 * k;
                    const float sx = (face == 0 || face == 3) ? 1.f : -1.f, sy = face < 2 ? 1.f : -1.f;
                    v = sx * R[bb] + sy * R[3 + bb] - K.mu_fr * R[6 + bb];
                } break;
            }
            jo[e] = v;
        }
    }
    // ---------------- hess L (CCS order, full symmetric) ----------------
    if (Hess) {
        GLay Gl = G;
        float* ho = Hess + (size_t)b * nnzh;
        for (int e = tid; e < nnzh; e += NT) {
            const int d = hdesc[e];
            const int kind = d & 15, k = (d >> 4) & 63, c = (d >> 10) & 1, j = (d >> 11) & 3, a = (d >> 13) & 3, bb = (d >> 15) & 3,
                      l = (d >> 17) & 3;
            float v;
            switch (kind) {
                case H_COMX: v = lam_f * 2.f * K.w_com0; break;
                case H_COMY: v = lam_f * 2.f * K.w_com1; break;
                case H_COMZ: v = lam_f * K.wz2[k]; break;
                case H_H: v = lam_f * 2.f * K.w_h; break;
                case H_POS: v = lam_f * 2.f * K.w_pos; break;
                case H_FF: {
                    const float g = gam(c, k);
                    v = 2.f * K.w_sym * ((j == l ? 1.f : 0.f) - 0.25f * g * (2.f - g));
                    if (j == l) v += K.D[a] * (float)((k > 0) + (k + 1 < N));
                    v *= lam_f;
                } break;
                case H_RATE: v = -lam_f * K.D[a]; break;
                default: {
                    const float* lh = lam + Gl.g_h + 3 * k;
                    const float s = dt * gam(c, k) * skew(lh, a, bb);
                    v = (kind == H_F_POS) ? -s : s;
                } break;
            }
            ho[e] = v;
        }
    }
}

// nlp_grad (tmp.c:24791-58842): gradient of gamma = lam_f f + lam_g^T g with respect to x and to p, batched.  One
// workgroup per problem, x / p / lam_g staged in LDS, every output element one thread's closed form (J^T lam_g is
// written out per x-block instead of scattering the Jacobian's non-zeros).  Parameter blocks that do not enter f or g
// (limA, limB, currentPos, com0, dcom0, h0: CasADi's Opti turns them into bounds) get zeros, as in the reference.
__global__ __launch_bounds__(256) void cmpc_nlp_grad_kernel(CmpcParams kp, const float* __restrict__ X, const float* __restrict__ P,
                                                            const float* __restrict__ LamG, float lam_f, float* __restrict__ GradX,
                                                            float* __restrict__ GradP)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, NT = 256;
    const int b = blockIdx.x;
    const int N = kp.N;
    CmpcConsts& K = *reinterpret_cast<CmpcConsts*>(smem);
    {
        const int* src = reinterpret_cast<const int*>(kp.kc);
        int* dst = reinterpret_cast<int*>(smem);
        for (int e = tid; e < (int)(sizeof(CmpcConsts) / 4); e += NT) dst[e] = src[e];
    }
    const CmpcIdx L{N};
    GLay G;
    glay_init(G, N);
    float* x = reinterpret_cast<float*>(smem + ((sizeof(CmpcConsts) + 15) & ~15));
    float* p = x + ((L.nx() + 3) & ~3);
    float* lam = p + ((L.np() + 3) & ~3);
    for (int e = tid; e < L.nx(); e += NT) x[e] = X[(size_t)b * L.nx() + e];
    for (int e = tid; e < L.np(); e += NT) p[e] = P[(size_t)b * L.np() + e];
    for (int e = tid; e < L.ng(); e += NT) lam[e] = LamG[(size_t)b * L.ng() + e];
    __syncthreads();
    const float dt = K.dt;
    auto gam = [&](int c, int k) { return p[L.pGam(c) + k]; };
    auto rvec = [&](int c, int j, int k, float* r) {
        const float* R = p + L.pR(c) + 9 * k;
        const float* cn = K.corners + 12 * c + 3 * j;
        for (int i = 0; i < 3; ++i)
            r[i] = R[i] * cn[0] + R[3 + i] * cn[1] + R[6 + i] * cn[2] + x[L.oPos(c) + 3 * k + i] - x[L.oCom() + 3 * k + i];
    };
    auto fcsum = [&](int c, int k, float* Fc) {
        for (int i = 0; i < 3; ++i)
            Fc[i] = x[L.oF(c, 0) + 3 * k + i] + x[L.oF(c, 1) + 3 * k + i] + x[L.oF(c, 2) + 3 * k + i] + x[L.oF(c, 3) + 3 * k + i];
    };
    auto crossc = [](const float* a, const float* bb, int i) { return a[(i + 1) % 3] * bb[(i + 2) % 3] - a[(i + 2) % 3] * bb[(i + 1) % 3]; };
    // d f / d (force component): symmetry + rate terms (as in the grad f branch of cmpc_nlp_eval_kernel)
    auto gradf_force = [&](int c, int k, int i, int e) {
        const float g = gam(c, k);
        float mean = 0.f;
        for (int l = 0; l < 4; ++l) mean += 0.25f * x[L.oF(c, l) + 3 * k + i];
        const float es = x[e] - g * mean, esum = 4.f * mean * (1.f - g);
        float v = 2.f * K.w_sym * (es - 0.25f * g * esum);
        if (k > 0) v += K.D[i] * (x[e] - x[e - 3]);
        if (k + 1 < N) v -= K.D[i] * (x[e + 3] - x[e]);
        return v;
    };

    if (GradX) {
        float* gx = GradX + (size_t)b * L.nx();
        for (int e = tid; e < L.nx(); e += NT) {
            float v;
            if (e < L.oPos(0)) {  // com | dcom | h : 3 x (N+1) each
                const int blk = e / (3 * (N + 1)), e2 = e % (3 * (N + 1)), k = e2 / 3, i = e2 % 3;
                const int grow = blk == 0 ? G.g_com : (blk == 1 ? G.g_dcom : G.g_h);
                v = k == 0 ? lam[G.g_init + 3 * blk + i] : lam[grow + 3 * (k - 1) + i];
                if (k < N) v -= lam[grow + 3 * k + i];
                if (blk == 0) {
                    v += lam_f * (i == 0 ? 2.f * K.w_com0 : (i == 1 ? 2.f * K.w_com1 : K.wz2[k])) * (x[e] - p[L.pComref() + e2]);
                    if (k < N) {  // rows g_h: -dt [Fsum]x  ->  -dt (lam_h x Fsum)
                        float F0[3], F1[3], Fs[3];
                        fcsum(0, k, F0);
                        fcsum(1, k, F1);
                        for (int a = 0; a < 3; ++a) Fs[a] = gam(0, k) * F0[a] + gam(1, k) * F1[a];
                        v -= dt * crossc(lam + G.g_h + 3 * k, Fs, i);
                    }
                } else if (blk == 1) {
                    if (k < N) v -= dt * lam[G.g_com + 3 * k + i];
                } else v += lam_f * 2.f * K.w_h * (x[e] - p[L.pHref() + e2]);
            } else {
                const int c = e < L.oPos(1) ? 0 : 1;
                const int e2 = e - L.oPos(c);
                if (e2 < 3 * (N + 1)) {  // pos
                    const int k = e2 / 3, a = e2 % 3;
                    v = lam_f * 2.f * K.w_pos * (x[e] - p[L.pNom(c) + e2]);
                    if (k == 0) v += lam[G.g_init + 9 + 3 * c + a];
                    else {
                        const float* R = p + L.pR(c) + 9 * (k - 1);
                        const float* lb = lam + G.g_bbox[c] + 3 * (k - 1);
                        v += lam[G.g_pos[c] + 3 * (k - 1) + a] + lb[0] * R[a] + lb[1] * R[3 + a] + lb[2] * R[6 + a];
                    }
                    if (k < N) {
                        float Fc[3];
                        fcsum(c, k, Fc);
                        v += -lam[G.g_pos[c] + 3 * k + a] + dt * gam(c, k) * crossc(lam + G.g_h + 3 * k, Fc, a);
                    }
                } else if (e2 < 3 * (N + 1) + 3 * N) {  // vel
                    const int e3 = e2 - 3 * (N + 1), k = e3 / 3;
                    v = -dt * (1.f - gam(c, k)) * lam[G.g_pos[c] + e3];
                } else {  // corner forces
                    const int e3 = e2 - 3 * (N + 1) - 3 * N, j = e3 / (3 * N), k = (e3 % (3 * N)) / 3, a = e3 % 3;
                    const float g = gam(c, k);
                    const float* R = p + L.pR(c) + 9 * k;
                    const float* lf = lam + G.g_fric[c] + 16 * k + 4 * j;
                    float rr[3];
                    rvec(c, j, k, rr);
                    v = lam_f * gradf_force(c, k, a, e) - dt * g * (lam[G.g_dcom + 3 * k + a] + crossc(lam + G.g_h + 3 * k, rr, a));
                    const float cx = lf[0] - lf[1] - lf[2] + lf[3], cy = lf[0] + lf[1] - lf[2] - lf[3], cz = -K.mu_fr * (lf[0] + lf[1] + lf[2] + lf[3]);
                    v += cx * R[a] + cy * R[3 + a] + cz * R[6 + a];
                }
            }
            gx[e] = v;
        }
    }
    if (GradP) {
        float* gp = GradP + (size_t)b * L.np();
        for (int e = tid; e < L.np(); e += NT) {
            float v = 0.f;
            if (e < L.pCom0()) {
                const int c = e < L.pR(1) ? 0 : 1;
                const int e2 = e - L.pR(c);
                if (e2 < 9 * N) {  // R(a, m) at 9 k + 3 m + a
                    const int k = e2 / 9, m = (e2 % 9) / 3, a = e2 % 3;
                    const float g = gam(c, k);
                    const float* lh = lam + G.g_h + 3 * k;
                    for (int j = 0; j < 4; ++j) {
                        const float* f = x + L.oF(c, j) + 3 * k;
                        const float* lf = lam + G.g_fric[c] + 16 * k + 4 * j;
                        const float coef = m == 0 ? (lf[0] - lf[1] - lf[2] + lf[3]) : (m == 1 ? (lf[0] + lf[1] - lf[2] - lf[3]) : -K.mu_fr * (lf[0] + lf[1] + lf[2] + lf[3]));
                        v += -dt * g * K.corners[12 * c + 3 * j + m] * crossc(f, lh, a) + coef * f[a];
                    }
                    v += lam[G.g_bbox[c] + 3 * k + m] * (x[L.oPos(c) + 3 * (k + 1) + a] - p[L.pNom(c) + 3 * (k + 1) + a]);
                } else if (e2 < 15 * N) {
                    v = 0.f;   // limA, limB
                } else if (e2 < 16 * N) {  // Gamma
                    const int k = e2 - 15 * N;
                    const float g = gam(c, k);
                    const float* ld = lam + G.g_dcom + 3 * k;
                    const float* lh = lam + G.g_h + 3 * k;
                    float mean[3] = {0.f, 0.f, 0.f}, esum[3];
                    for (int j = 0; j < 4; ++j)
                        for (int i = 0; i < 3; ++i) mean[i] += 0.25f * x[L.oF(c, j) + 3 * k + i];
                    for (int i = 0; i < 3; ++i) esum[i] = 4.f * mean[i] * (1.f - g);
                    for (int j = 0; j < 4; ++j) {
                        const float* f = x + L.oF(c, j) + 3 * k;
                        float rr[3];
                        rvec(c, j, k, rr);
                        for (int i = 0; i < 3; ++i) v -= dt * (ld[i] * f[i] + lh[i] * crossc(rr, f, i));
                    }
                    for (int i = 0; i < 3; ++i)
                        v += -lam_f * 2.f * K.w_sym * mean[i] * esum[i] + dt * lam[G.g_pos[c] + 3 * k + i] * x[L.oVel(c) + 3 * k + i];
                } else if (e2 < 16 * N + 3 * (N + 1)) {  // nominalPos
                    const int e3 = e2 - 16 * N, k = e3 / 3, a = e3 % 3;
                    v = -lam_f * 2.f * K.w_pos * (x[L.oPos(c) + e3] - p[e]);
                    if (k > 0) {
                        const float* R = p + L.pR(c) + 9 * (k - 1);
                        const float* lb = lam + G.g_bbox[c] + 3 * (k - 1);
                        v -= lb[0] * R[a] + lb[1] * R[3 + a] + lb[2] * R[6 + a];
                    }
                }   // currentPos: 0
            } else if (e >= L.pComref() && e < L.pHref()) {
                const int e2 = e - L.pComref(), k = e2 / 3, i = e2 % 3;
                v = -lam_f * (i == 0 ? 2.f * K.w_com0 : (i == 1 ? 2.f * K.w_com1 : K.wz2[k])) * (x[L.oCom() + e2] - p[e]);
            } else if (e >= L.pHref() && e < L.pFext()) {
                v = -lam_f * 2.f * K.w_h * (x[L.oH() + e - L.pHref()] - p[e]);
            } else if (e >= L.pFext() && e < L.pText()) {
                v = -dt * lam[G.g_dcom + e - L.pFext()];
            } else if (e >= L.pText()) {
                v = -dt * lam[G.g_h + e - L.pText()];
            }   // com0, dcom0, h0: 0
            gp[e] = v;
        }
    }
}

// warm start: previous solution shifted by one knot (last knot repeated); is_warm_start_enabled of
// the reference (ergoCubGazeboV1/centroidal_mpc.ini:9)
// (one problem: xp -> x0, thread tid of nt)
__device__ inline void warm_shift_problem(int N, const float* __restrict__ xp, float* __restrict__ x0, int tid, int nt)
{
    CmpcLayout L;
    cmpc_layout_init(L, N);
    for (int e = tid; e < L.nx; e += nt) {
        // every block of x is 3 x (N+1) or 3 x N, column = knot: find block start and length
        int start, len;
        if (e < L.o_pos[0]) { start = (e / (3 * (N + 1))) * 3 * (N + 1); len = 3 * (N + 1); }
        else {
            const int c = e < L.o_pos[1] ? 0 : 1, e2 = e - L.o_pos[c];
            if (e2 < 3 * (N + 1)) { start = L.o_pos[c]; len = 3 * (N + 1); }
            else { start = L.o_pos[c] + 3 * (N + 1) + ((e2 - 3 * (N + 1)) / (3 * N)) * 3 * N; len = 3 * N; }
        }
        const int off = e - start;
        const int src = off + 3 < len ? e + 3 : e;  // shift by one knot, repeat the last
        x0[e] = xp[src];
    }
}
__global__ __launch_bounds__(256) void cmpc_warm_shift_kernel(int N, int B, const float* __restrict__ Xp, float* __restrict__ X0)
{
    CmpcLayout L;
    cmpc_layout_init(L, N);
    const int b = blockIdx.x;
    warm_shift_problem(N, Xp + (size_t)b * L.nx, X0 + (size_t)b * L.nx, threadIdx.x, 256);
}

void build_sparsity(int N, std::vector<Trip>& J, std::vector<Trip>& H)
{
    CmpcLayout L;
    cmpc_layout_init(L, N);
    GLay G;
    glay_init(G, N);
    J.clear(); H.clear();
    for (int i = 0; i < 3; ++i) {
        J.push_back({G.g_init + i, L.o_com + i, mk(J_ONE, 0)});
        J.push_back({G.g_init + 3 + i, L.o_dcom + i, mk(J_ONE, 0)});
        J.push_back({G.g_init + 6 + i, L.o_h + i, mk(J_ONE, 0)});
        J.push_back({G.g_init + 9 + i, L.o_pos[0] + i, mk(J_ONE, 0)});
        J.push_back({G.g_init + 12 + i, L.o_pos[1] + i, mk(J_ONE, 0)});
    }
    for (int k = 0; k < N; ++k) {
        for (int i = 0; i < 3; ++i) {
            J.push_back({G.g_com + 3 * k + i, L.o_com + 3 * (k + 1) + i, mk(J_ONE, k)});
            J.push_back({G.g_com + 3 * k + i, L.o_com + 3 * k + i, mk(J_MONE, k)});
            J.push_back({G.g_com + 3 * k + i, L.o_dcom + 3 * k + i, mk(J_MDT, k)});
            J.push_back({G.g_dcom + 3 * k + i, L.o_dcom + 3 * (k + 1) + i, mk(J_ONE, k)});
            J.push_back({G.g_dcom + 3 * k + i, L.o_dcom + 3 * k + i, mk(J_MONE, k)});
            J.push_back({G.g_h + 3 * k + i, L.o_h + 3 * (k + 1) + i, mk(J_ONE, k)});
            J.push_back({G.g_h + 3 * k + i, L.o_h + 3 * k + i, mk(J_MONE, k)});
        }
        for (int c = 0; c < 2; ++c) {
            for (int j = 0; j < 4; ++j) {
                for (int i = 0; i < 3; ++i) J.push_back({G.g_dcom + 3 * k + i, L.o_f[c][j] + 3 * k + i, mk(J_DCOM_F, k, c, j)});
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b)
                        if (a != b) J.push_back({G.g_h + 3 * k + a, L.o_f[c][j] + 3 * k + b, mk(J_H_F, k, c, j, a, b)});
                for (int face = 0; face < 4; ++face)
                    for (int b = 0; b < 3; ++b)
                        J.push_back({G.g_fric[c] + 16 * k + 4 * j + face, L.o_f[c][j] + 3 * k + b, mk(J_FRIC, k, c, j, 0, b, face)});
            }
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b)
                    if (a != b) J.push_back({G.g_h + 3 * k + a, L.o_pos[c] + 3 * k + b, mk(J_H_POS, k, c, 0, a, b)});
            for (int i = 0; i < 3; ++i) {
                J.push_back({G.g_pos[c] + 3 * k + i, L.o_pos[c] + 3 * (k + 1) + i, mk(J_ONE, k)});
                J.push_back({G.g_pos[c] + 3 * k + i, L.o_pos[c] + 3 * k + i, mk(J_MONE, k)});
                J.push_back({G.g_pos[c] + 3 * k + i, L.o_vel[c] + 3 * k + i, mk(J_POS_VEL, k, c)});
                for (int a = 0; a < 3; ++a) J.push_back({G.g_bbox[c] + 3 * k + i, L.o_pos[c] + 3 * (k + 1) + a, mk(J_BBOX, k, c, 0, i, a)});
            }
        }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                if (a != b) J.push_back({G.g_h + 3 * k + a, L.o_com + 3 * k + b, mk(J_H_COM, k, 0, 0, a, b)});
    }
    for (int k = 0; k <= N; ++k) {
        H.push_back({L.o_com + 3 * k, L.o_com + 3 * k, mk(H_COMX, k)});
        H.push_back({L.o_com + 3 * k + 1, L.o_com + 3 * k + 1, mk(H_COMY, k)});
        H.push_back({L.o_com + 3 * k + 2, L.o_com + 3 * k + 2, mk(H_COMZ, k)});
        for (int i = 0; i < 3; ++i) {
            H.push_back({L.o_h + 3 * k + i, L.o_h + 3 * k + i, mk(H_H, k)});
            for (int c = 0; c < 2; ++c) H.push_back({L.o_pos[c] + 3 * k + i, L.o_pos[c] + 3 * k + i, mk(H_POS, k)});
        }
    }
    for (int k = 0; k < N; ++k)
        for (int c = 0; c < 2; ++c)
            for (int j = 0; j < 4; ++j) {
                const int fj = L.o_f[c][j] + 3 * k;
                for (int l = 0; l < 4; ++l)
                    for (int i = 0; i < 3; ++i) H.push_back({fj + i, L.o_f[c][l] + 3 * k + i, mk(H_FF, k, c, j, i, 0, l)});
                if (k + 1 < N)
                    for (int i = 0; i < 3; ++i) {
                        H.push_back({fj + i, fj + 3 + i, mk(H_RATE, k, c, j, i)});
                        H.push_back({fj + 3 + i, fj + i, mk(H_RATE, k, c, j, i)});
                    }
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) {
                        if (a == b) continue;
                        H.push_back({fj + a, L.o_pos[c] + 3 * k + b, mk(H_F_POS, k, c, j, a, b)});
                        H.push_back({L.o_pos[c] + 3 * k + b, fj + a, mk(H_F_POS, k, c, j, a, b)});
                        H.push_back({fj + a, L.o_com + 3 * k + b, mk(H_F_COM, k, c, j, a, b)});
                        H.push_back({L.o_com + 3 * k + b, fj + a, mk(H_F_COM, k, c, j, a, b)});
                    }
            }
    auto ccs = [](const Trip& u, const Trip& v) { return u.col != v.col ? u.col < v.col : u.row < v.row; };
    std::sort(J.begin(), J.end(), ccs);
    std::sort(H.begin(), H.end(), ccs);
}

struct DescCache {
    int N = -1, device = -1;
    int *dJ = nullptr, *dH = nullptr;
    int nnzj = 0, nnzh = 0;
};
DescCache g_cache;

}  // namespace

extern "C" int cmpc_nlp_sparsity(int N, int* jac_row, int* jac_col, int* hess_row, int* hess_col)
{
    if (N < 1 || N > CMPC_NMAX) return -1;
    std::vector<Trip> J, H;
    build_sparsity(N, J, H);
    for (size_t i = 0; i < J.size(); ++i) {
        if (jac_row) jac_row[i] = J[i].row;
        if (jac_col) jac_col[i] = J[i].col;
    }
    for (size_t i = 0; i < H.size(); ++i) {
        if (hess_row) hess_row[i] = H[i].row;
        if (hess_col) hess_col[i] = H[i].col;
    }
    return 0;
}

extern "C" int cmpc_launch_nlp_eval(const CmpcParams* prm, const float* dX, const float* dP, const float* dLamG, float lam_f,
                                    float* dF, float* dG, float* dGradF, float* dJac, float* dHess, hipStream_t stream)
{
    const int N = prm->N;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (g_cache.N != N || g_cache.device != dev) {
        std::vector<Trip> J, H;
        build_sparsity(N, J, H);
        std::vector<int> dj(J.size()), dh(H.size());
        for (size_t i = 0; i < J.size(); ++i) dj[i] = J[i].desc;
        for (size_t i = 0; i < H.size(); ++i) dh[i] = H[i].desc;
        if (g_cache.dJ) (void)hipFree(g_cache.dJ);
        if (g_cache.dH) (void)hipFree(g_cache.dH);
        if ((e = hipMalloc(&g_cache.dJ, dj.size() * sizeof(int))) != hipSuccess) return (int)e;
        if ((e = hipMalloc(&g_cache.dH, dh.size() * sizeof(int))) != hipSuccess) return (int)e;
        if ((e = hipMemcpy(g_cache.dJ, dj.data(), dj.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return (int)e;
        if ((e = hipMemcpy(g_cache.dH, dh.data(), dh.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return (int)e;
        g_cache.N = N; g_cache.device = dev; g_cache.nnzj = (int)dj.size(); g_cache.nnzh = (int)dh.size();
    }
    CmpcLayout L;
    cmpc_layout_init(L, N);
    const size_t lds = ((sizeof(CmpcConsts) + 15) & ~(size_t)15) + 4 * (size_t)(((L.nx + 3) & ~3) + ((L.np + 3) & ~3) + ((L.ng + 3) & ~3) + 8);
    hipLaunchKernelGGL(cmpc_nlp_eval_kernel, dim3(prm->B), dim3(256), lds, stream, *prm, dX, dP, dLamG, lam_f, dF, dG, dGradF, dJac, dHess,
                       g_cache.dJ, g_cache.dH, g_cache.nnzj, g_cache.nnzh);
    return (int)hipGetLastError();
}

extern "C" int cmpc_launch_nlp_grad(const CmpcParams* prm, const float* dX, const float* dP, const float* dLamG, float lam_f, float* dGradX,
                                    float* dGradP, hipStream_t stream)
{
    CmpcLayout L;
    cmpc_layout_init(L, prm->N);
    const size_t lds = ((sizeof(CmpcConsts) + 15) & ~(size_t)15) + 4 * (size_t)(((L.nx + 3) & ~3) + ((L.np + 3) & ~3) + ((L.ng + 3) & ~3) + 8);
    hipLaunchKernelGGL(cmpc_nlp_grad_kernel, dim3(prm->B), dim3(256), lds, stream, *prm, dX, dP, dLamG, lam_f, dGradX, dGradP);
    return (int)hipGetLastError();
}

extern "C" int cmpc_launch_warm_shift(const CmpcParams* prm, const float* dXprev, float* dX0, hipStream_t stream)
{
    hipLaunchKernelGGL(cmpc_warm_shift_kernel, dim3(prm->B), dim3(256), 0, stream, prm->N, prm->B, dXprev, dX0);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Closed-loop plant between two MPC ticks: what WholeBodyQPBlock does with the MPC
// output -- centroidal dynamics driven by the first-knot corner forces of the active contacts plus
// the external wrench, RK4 with the forces held (src/centroidal-mpc-walking/src/
// WholeBodyQPBlock.cpp:1083-1084, 1150, 1259-1262), and the desired ZMP from the corner forces
// (:805-873: per-foot local ZMP = (-tau_y, tau_x)/f_z clamped to the sole, f_z-weighted mean in the
// world frame).  Everything mass-normalised like the MPC.  One thread per problem (a few hundred
// flops): the kernel is HBM/launch bound and exists so that Monte-Carlo roll-outs never leave HBM.
namespace {

// (problem b, one thread.  state_in / state_out may alias: everything is read before anything is written)
__device__ inline void plant_step_problem(int N, int b, float grav, const float* __restrict__ corners, const float* __restrict__ X,
                                          const float* __restrict__ P, const float* state_in, float* state_out, float* __restrict__ zmp, float h,
                                          int nsub, float zx, float zy)
{
    CmpcIdx L{N};
    const float* x = X + (size_t)b * L.nx();
    const float* p = P + (size_t)b * L.np();
    double com[3], v[3], hm[3], fsum[3] = {0, 0, 0}, ext_t[3];
    for (int i = 0; i < 3; ++i) {
        com[i] = state_in[(size_t)b * 9 + i]; v[i] = state_in[(size_t)b * 9 + 3 + i]; hm[i] = state_in[(size_t)b * 9 + 6 + i];
        ext_t[i] = p[L.pText() + i];
    }
    // contact points (world) and forces of the active contacts at knot 0
    double cp[8][3], cf[8][3];
    double zw[2] = {0, 0}, ztot = 0;
    for (int c = 0; c < 2; ++c) {
        const float* R = p + L.pR(c);  // col-major vec of knot 0
        const bool on = p[L.pGam(c)] > 0.5f;
        double F[3] = {0, 0, 0}, T[3] = {0, 0, 0};
        for (int j = 0; j < 4; ++j) {
            const float* cn = corners + 12 * c + 3 * j;
            double fl[3];
            for (int i = 0; i < 3; ++i) {
                cp[4 * c + j][i] = (double)x[L.oPos(c) + i] + (double)R[i] * cn[0] + (double)R[3 + i] * cn[1] + (double)R[6 + i] * cn[2];
                const float fv = x[L.oF(c, j) + i];   // (loaded whatever `on` is: a load behind the contact flag is a second global-memory round trip on this one-thread chain)
                cf[4 * c + j][i] = on ? (double)fv : 0.0;
                fsum[i] += cf[4 * c + j][i];
            }
            for (int i = 0; i < 3; ++i) fl[i] = (double)R[3 * i] * cf[4 * c + j][0] + (double)R[3 * i + 1] * cf[4 * c + j][1] + (double)R[3 * i + 2] * cf[4 * c + j][2];
            for (int i = 0; i < 3; ++i) F[i] += cf[4 * c + j][i];
            T[0] += cn[1] * fl[2] - cn[2] * fl[1];
            T[1] += cn[2] * fl[0] - cn[0] * fl[2];
            T[2] += cn[0] * fl[1] - cn[1] * fl[0];
        }
        if (F[2] > 0.001) {
            double lx = fmin((double)zx, fmax(-(double)zx, -T[1] / F[2]));
            double ly = fmin((double)zy, fmax(-(double)zy, T[0] / F[2]));
            ztot += F[2];
            zw[0] += F[2] * ((double)x[L.oPos(c)] + (double)R[0] * lx + (double)R[3] * ly);
            zw[1] += F[2] * ((double)x[L.oPos(c) + 1] + (double)R[1] * lx + (double)R[4] * ly);
        }
    }
    // The forces are held over the step, so the torque about the CoM, sum_q (p_q - c) x f_q, is tau0 - c x fsum with tau0 = sum_q p_q x f_q formed once: one cross
    // product per Runge-Kutta stage instead of eight (this one-thread-per-problem kernel is a dependent chain of float64 operations: 18 us of a tick before,
    // profiles/r04_rollout_tick_overhead.txt).
    double tau0[3] = {ext_t[0], ext_t[1], ext_t[2]}, acc[3];
    for (int q = 0; q < 8; ++q) {
        tau0[0] += cp[q][1] * cf[q][2] - cp[q][2] * cf[q][1];
        tau0[1] += cp[q][2] * cf[q][0] - cp[q][0] * cf[q][2];
        tau0[2] += cp[q][0] * cf[q][1] - cp[q][1] * cf[q][0];
    }
    for (int i = 0; i < 3; ++i) acc[i] = fsum[i] + (double)p[L.pFext() + i] - (i == 2 ? (double)grav : 0.0);
    auto deriv = [&](const double* cm, const double* vv, double* dcm, double* dv, double* dh) {
        for (int i = 0; i < 3; ++i) { dcm[i] = vv[i]; dv[i] = acc[i]; }
        dh[0] = tau0[0] - (cm[1] * fsum[2] - cm[2] * fsum[1]);
        dh[1] = tau0[1] - (cm[2] * fsum[0] - cm[0] * fsum[2]);
        dh[2] = tau0[2] - (cm[0] * fsum[1] - cm[1] * fsum[0]);
    };
    for (int s = 0; s < nsub; ++s) {
        double k1c[3], k1v[3], k1h[3], k2c[3], k2v[3], k2h[3], k3c[3], k3v[3], k3h[3], k4c[3], k4v[3], k4h[3], tc[3], tv[3];
        deriv(com, v, k1c, k1v, k1h);
        for (int i = 0; i < 3; ++i) { tc[i] = com[i] + 0.5 * h * k1c[i]; tv[i] = v[i] + 0.5 * h * k1v[i]; }
        deriv(tc, tv, k2c, k2v, k2h);
        for (int i = 0; i < 3; ++i) { tc[i] = com[i] + 0.5 * h * k2c[i]; tv[i] = v[i] + 0.5 * h * k2v[i]; }
        deriv(tc, tv, k3c, k3v, k3h);
        for (int i = 0; i < 3; ++i) { tc[i] = com[i] + h * k3c[i]; tv[i] = v[i] + h * k3v[i]; }
        deriv(tc, tv, k4c, k4v, k4h);
        for (int i = 0; i < 3; ++i) {
            com[i] += h / 6.0 * (k1c[i] + 2 * k2c[i] + 2 * k3c[i] + k4c[i]);
            v[i] += h / 6.0 * (k1v[i] + 2 * k2v[i] + 2 * k3v[i] + k4v[i]);
            hm[i] += h / 6.0 * (k1h[i] + 2 * k2h[i] + 2 * k3h[i] + k4h[i]);
        }
    }
    for (int i = 0; i < 3; ++i) {
        state_out[(size_t)b * 9 + i] = (float)com[i]; state_out[(size_t)b * 9 + 3 + i] = (float)v[i]; state_out[(size_t)b * 9 + 6 + i] = (float)hm[i];
    }
    if (zmp) {
        zmp[(size_t)b * 2] = ztot > 0.001 ? (float)(zw[0] / ztot) : nanf("");
        zmp[(size_t)b * 2 + 1] = ztot > 0.001 ? (float)(zw[1] / ztot) : nanf("");
    }
}

__global__ __launch_bounds__(256) void cmpc_plant_step_kernel(int N, int B, float grav, const float* __restrict__ corners,
                                                              const float* __restrict__ X, const float* __restrict__ P,
                                                              const float* state_in, float* state_out,
                                                              float* __restrict__ zmp, float h, int nsub, float zx, float zy)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    plant_step_problem(N, b, grav, corners, X, P, state_in, state_out, zmp, h, nsub, zx, zy);
}

// ---- the two ends of a roll-out tick as ONE launch each (cmpc_rollout_tick_device).  At B <= 256 a tick is a 0.66 ms solve between nine launches of a
// few microseconds of work each, and the dispatch of a kernel behind another costs as much as they do (tools/gpu_rollout_tick_overhead.py): the steps in
// front of the solve touch disjoint entries of P and X0 (contact blocks / state rows / the shifted solution) and so do the two behind it (the lists' poses /
// the state), so each group is one grid whose threads call the SAME per-problem functions as the single kernels -- results identical to the last bit
// (tests/test_gpu_rollout.py).
// pre: one workgroup per problem.  All threads: setState and the warm-start shift; threads 0, 1: merge (updateContactPhaseList) of foot 0 / 1; then one thread per
// (foot, stage): sampling (setContactPhaseList) of the merged lists; threads 0, 1: the landing knots.
__global__ __launch_bounds__(256) void cmpc_tick_pre_kernel(int B, int N, int M, double dt, double now, int merge, const double* plan_t, const float* plan_pose,
                                                            const int* plan_n, const double* prev_t, const float* prev_pose, const int* prev_n, double* list_t,
                                                            float* list_pose, int* list_n, int* ok, int* land, const float* __restrict__ box,
                                                            const float* __restrict__ state, const float* __restrict__ wrench, float* P,
                                                            const float* __restrict__ Xprev, float* __restrict__ X0, const float* __restrict__ plan_com,
                                                            const float* __restrict__ plan_h, int plan_knots, double plan_dt, double plan_t_offset,
                                                            double robot_mass, double com_height)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const CmpcIdx L{N};
    float* p = P + (size_t)b * L.np();
    __shared__ int okw;
    if (tid == 0) okw = 1;
    __syncthreads();
    // setState and the warm-start shift first (they depend on nothing the kernel computes: their memory traffic runs under the merge of threads 0, 1)
    for (int e = tid; e < 9; e += 256) p[L.pCom0() + e] = state[9 * (size_t)b + e];
    if (wrench)
        for (int e = tid; e < 3 * N; e += 256) {
            const int k = e / 3, i = e % 3;
            p[L.pFext() + e] = wrench[((size_t)b * N + k) * 6 + i];
            p[L.pText() + e] = wrench[((size_t)b * N + k) * 6 + 3 + i];
        }
    if (Xprev) warm_shift_problem(N, Xprev + (size_t)b * L.nx(), X0 + (size_t)b * L.nx(), tid, 256);
    if (plan_com)   // setReferenceTrajectory from the planner's trajectories (8f-3), one thread per knot, from the far end of the workgroup
        for (int k = 255 - tid; k <= N; k += 256)
            cmpc_resample_reference_knot(plan_com + (size_t)b * plan_knots * 3, plan_h + (size_t)b * plan_knots * 3, plan_knots, plan_dt, plan_t_offset, dt, k,
                                         robot_mass, com_height, p + L.pComref() + 3 * k, p + L.pHref() + 3 * k);
    // the merge scans its lists entry by entry: from LDS (the workgroup fetches both feet's lists of the planner and of the previous tick in one round trip) when they
    // fit, else from global memory as the single kernel does -- the same function on the same values either way
    constexpr int MS = 16;
    __shared__ double st_[2][2 * 2 * MS];      // [planner | previous][foot][M][2]
    __shared__ float sp_[2][2 * 7 * MS];
    const bool staged = merge && M <= MS;
    if (staged) {
        const size_t o2 = (size_t)(2 * b) * M;
        for (int e = tid; e < 4 * M; e += 256) { st_[0][e] = plan_t[2 * o2 + e]; st_[1][e] = prev_t[2 * o2 + e]; }
        for (int e = tid; e < 14 * M; e += 256) { sp_[0][e] = plan_pose[7 * o2 + e]; sp_[1][e] = prev_pose[7 * o2 + e]; }
        __syncthreads();
    }
    // this tick's lists live in LDS too (so_*): the merge writes them there, the sampling threads scan them there, and the workgroup copies them out
    __shared__ double so_t[2 * 2 * MS];
    __shared__ float so_p[2 * 7 * MS];
    __shared__ int so_n[2];
    const bool lstaged = M <= MS;
    const size_t ob = (size_t)(2 * b) * M;
    if (lstaged && !merge) {   // (first tick: the caller filled the lists)
        for (int e = tid; e < 4 * M; e += 256) so_t[e] = list_t[2 * ob + e];
        for (int e = tid; e < 14 * M; e += 256) so_p[e] = list_pose[7 * ob + e];
        if (tid < 2) so_n[tid] = list_n[2 * b + tid];
    }
    if (tid < 2 && merge) {
        const int e = 2 * b + tid;
        const size_t o = (size_t)e * M;
        const int pn = plan_n[e], mn = prev_n[e];
        const bool sane = pn >= 0 && pn <= M && mn >= 0 && mn <= M;
        int* on = lstaged ? &so_n[tid] : list_n + e;
        if (!sane) *on = 0;
        const double* pt = staged ? st_[0] + 2 * M * tid : plan_t + 2 * o;
        const double* mt = staged ? st_[1] + 2 * M * tid : prev_t + 2 * o;
        const float* pq = staged ? sp_[0] + 7 * M * tid : plan_pose + 7 * o;
        const float* mq = staged ? sp_[1] + 7 * M * tid : prev_pose + 7 * o;
        const bool good = sane && cmpc_merge_foot(now, pt, pq, pn, mt, mq, mn, M, lstaged ? so_t + 2 * M * tid : list_t + 2 * o,
                                                  lstaged ? so_p + 7 * M * tid : list_pose + 7 * o, on);
        if (!good) atomicAnd(&okw, 0);
    }
    __syncthreads();   // (the merged lists of the two feet are visible to the workgroup: in LDS, or in global memory when they do not fit)
    if (lstaged && merge) {   // out to global memory, the entries in use (what the single merge kernel writes)
        for (int e = tid; e < 4 * M; e += 256) { const int c = e / (2 * M); if (e - c * 2 * M < 2 * so_n[c]) list_t[2 * ob + e] = so_t[e]; }
        for (int e = tid; e < 14 * M; e += 256) { const int c = e / (7 * M); if (e - c * 7 * M < 7 * so_n[c]) list_pose[7 * ob + e] = so_p[e]; }
        if (tid < 2) list_n[2 * b + tid] = so_n[tid];
    }
    // sampling: one thread per (foot, stage) -- a single thread walking the N stages of a foot one global-memory round trip at a time was 33 of this kernel's
    // 39 us (rocprofv3, profiles/r04_rollout_tick_overhead.txt); the landing knot then comes from the stages' contact flags in LDS
    __shared__ unsigned char acts[2][CMPC_NMAX];
    for (int e2 = tid; e2 < 2 * N; e2 += 256) {
        const int cft = e2 / N, k = e2 - cft * N, e = 2 * b + cft;
        const size_t o = (size_t)e * M;
        const int n = lstaged ? so_n[cft] : list_n[e];
        if (n >= 1 && n <= M)
            acts[cft][k] = cmpc_sample_stage(N, dt, now, cft, k, lstaged ? so_t + 2 * M * cft : list_t + 2 * o, lstaged ? so_p + 7 * M * cft : list_pose + 7 * o, n, box,
                                             box + 6, p) ? 1 : 0;
    }
    __syncthreads();
    if (tid < 2) {
        const int e = 2 * b + tid, n = lstaged ? so_n[tid] : list_n[e];
        land[e] = (n < 1 || n > M) ? -2 : cmpc_landing_knot(N, [&](int k) { return acts[tid][k] != 0; });
    }
    if (merge && ok && tid == 0) ok[b] = okw;
}

// post: one thread per problem -- the plant step, then the step adjustment of its two feet (getOutput().contactPhaseList)
__global__ __launch_bounds__(256) void cmpc_tick_post_kernel(int B, int N, int M, double now, float grav, const float* __restrict__ corners,
                                                             const float* __restrict__ X, const float* __restrict__ P, const float* state_in, float* state_out,
                                                             float* __restrict__ zmp, float h, int nsub, float zx, float zy, const int* __restrict__ land,
                                                             const double* __restrict__ t, float* __restrict__ pose, const int* __restrict__ n)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    plant_step_problem(N, b, grav, corners, X, P, state_in, state_out, zmp, h, nsub, zx, zy);
    const CmpcIdx L{N};
    for (int c = 0; c < 2; ++c) {
        const int e = 2 * b + c, lk = land[e];
        if (lk < 0 || lk > N || n[e] < 1 || n[e] > M) continue;
        const size_t o = (size_t)e * M;
        const int nx = cmpc_next_contact(t + 2 * o, n[e], now);
        if (nx < 0) continue;
        const float* x = X + (size_t)b * L.nx() + L.oPos(c) + 3 * lk;
        for (int i = 0; i < 3; ++i) pose[7 * (o + nx) + i] = x[i];
    }
}

}  // namespace

extern "C" int cmpc_launch_tick_pre(int B, int N, int M, double dt, double now, int merge, const double* plan_t, const float* plan_pose, const int* plan_n,
                                    const double* prev_t, const float* prev_pose, const int* prev_n, double* list_t, float* list_pose, int* list_n, int* ok,
                                    int* land, const float* box, const float* state, const float* wrench, float* P, const float* Xprev, float* X0,
                                    const float* plan_com, const float* plan_h, int plan_knots, double plan_dt, double plan_t_offset, double robot_mass,
                                    double com_height, hipStream_t stream)
{
    hipLaunchKernelGGL(cmpc_tick_pre_kernel, dim3(B), dim3(256), 0, stream, B, N, M, dt, now, merge, plan_t, plan_pose, plan_n, prev_t, prev_pose, prev_n,
                       list_t, list_pose, list_n, ok, land, box, state, wrench, P, Xprev, X0, plan_com, plan_h, plan_knots, plan_dt, plan_t_offset, robot_mass,
                       com_height);
    return (int)hipGetLastError();
}

extern "C" int cmpc_launch_tick_post(int B, int N, int M, double now, float grav, const float* dCorners, const float* dX, const float* dP,
                                     const float* dStateIn, float* dStateOut, float* dZmp, float h, int nsub, float zx, float zy, const int* land,
                                     const double* t, float* pose, const int* n, hipStream_t stream)
{
    hipLaunchKernelGGL(cmpc_tick_post_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, B, N, M, now, grav, dCorners, dX, dP, dStateIn, dStateOut, dZmp,
                       h, nsub, zx, zy, land, t, pose, n);
    return (int)hipGetLastError();
}

extern "C" int cmpc_launch_plant_step(int N, int B, float grav, const float* dCorners, const float* dX, const float* dP,
                                      const float* dStateIn, float* dStateOut, float* dZmp, float h, int nsub, float zx, float zy,
                                      hipStream_t stream)
{
    hipLaunchKernelGGL(cmpc_plant_step_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, N, B, grav, dCorners, dX, dP, dStateIn,
                       dStateOut, dZmp, h, nsub, zx, zy);
    return (int)hipGetLastError();
}

// ---- compact per-problem output for the all-gather of the multi-GPU path: CoM trajectory 3(N+1), first-knot
// corner forces 24, knot-0 and knot-1 foot positions 12, iterations, status  ->  out[B][3(N+1) + 38] ----
namespace {
__global__ __launch_bounds__(128) void cmpc_compact_kernel(int N, const float* __restrict__ X, const float* __restrict__ info,
                                                           float* __restrict__ out)
{
    const CmpcIdx L{N};
    const int b = blockIdx.x, ncom = 3 * (N + 1), W = ncom + 38;
    const float* x = X + (size_t)b * L.nx();
    for (int c = threadIdx.x; c < W; c += 128) {
        float v;
        if (c < ncom) v = x[L.oCom() + c];
        else if (c < ncom + 24) { const int f = c - ncom; v = x[L.oF(f / 12, (f % 12) / 3) + f % 3]; }
        else if (c < ncom + 36) { const int p = c - ncom - 24; v = x[L.oPos(p / 6) + p % 6]; }
        else v = info[(size_t)b * CMPC_INFO_N + (c == ncom + 36 ? 0 : 5)];
        out[(size_t)b * W + c] = v;
    }
}
}  // namespace

extern "C" int cmpc_launch_compact(int N, int B, const float* dX, const float* dInfo, float* dOut, hipStream_t stream)
{
    hipLaunchKernelGGL(cmpc_compact_kernel, dim3(B), dim3(128), 0, stream, N, dX, dInfo, dOut);
    return (int)hipGetLastError();
}
