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
// dynamics defects, the right-hand-side recursion, and the 3x3 diagonal blocks of the stage
// Hessian Quu through its Cholesky (barrier terms z/t span 1e-6..1e9 inside one corner's block
// next to cost curvature of order 10).  Costates are float32: they only enter the exact-Hessian term.
//
// Structure.  The kernel body is control flow over a dozen scalars; EVERY pass lives in an out-of-line device function (a register
// allocation of its own) handed the LDS base, from which it rebuilds the LDS map: set-up, initial iterate, residuals, multiplier
// rescaling, the backward sweep, one per sweep, step lengths, corrector targets, update + step norms, the last step (extrapolation + tail
// polish), export.  Two variants: "resident" (512 threads, per-stage factor records in LDS, one workgroup per CU, B <= #CU) and
// "HBM-factor" (256 threads, records in global scratch, three workgroups per CU).
// What bounds every single-wave phase is INSTRUCTION COUNT: a wave issues at most one instruction of any kind -- scalar bookkeeping and
// exec-mask juggling included -- per four cycles (float64, conversions and transcendentals eight).  Hence, since round 4: the trailing updates
// of the fused factorisation on the matrix pipe (v_mfma_f32_4x4x1 with A-block broadcast: no v_readlane at all), the consumers of the
// streaming stage and the vector sweeps in "lean" form (everything fixed over a pass decoded once into per-lane plans / opaque LDS pointers,
// straight-line stage bodies on clamped indices, unmasked stores).
// Backward sweep of the resident variants: the streaming square-root stage (sq_factor_loop, sq_consume_loop) -- wave 0 factorises and
// publishes its columns block by block, waves 1-7 assemble the next stage's Z-independent part meanwhile and subtract Z^T Z from it one
// MFMA tile per wave as the blocks come; one barrier per stage, one call per pass and role.
// Backward sweep of the HBM-factor variants: two calls per stage (stage_mid: phase 3; stage_post_pre: phase 4 and phases 1-2 of the next
// stage), four barriers:
//   1. G = P [B;E]                        sparse: every column of A, B has <= 3 non-zeros
//   2. Quu, Qus (waves 0-1, float32), Quu diagonal blocks (wave 2, float64), Pd and qu (wave 3, float64)
//   3. fused Cholesky + panel solve       wave 0, matrix rows in registers: lanes 0-29 hold the rows of Quu, lanes 30-63 rows of
//      [Qus | I | qu]^T (the identity rows of the last columns take over the lanes of finished L rows), so L^{-1}[Qus | I | qu] falls
//      out of the same rank-3 updates (3x3 pivot blocks in float64, trailing updates on the matrix pipe, no LDS traffic);
//      meanwhile waves 2-3 build Qss, qs and the column descriptors of the next stage
//   4. P = [Qss 0; 0 D] - W^T W           30-term dot products on 16-byte LDS reads
// The vector sweeps (forward, corrector right-hand side, costates) run on one wave without
// workgroup barriers; the per-stage factors they read are stored transposed, a row per lane.
#include "cmpc_device.h"

#include <cstdlib>
#include <type_traits>
#include <map>
#include <mutex>

#define NS CMPC_NS
#define NF CMPC_NF
#define NQ CMPC_NQ
#define NU CMPC_NU
#define NXA CMPC_NXA
#define NI CMPC_NI
#define PLD 39     // leading dim of P (odd: column walks are conflict-free)
#define GLD 30     // G = P [B;E]  (39 x 30)
#define RLD 36     // leading dim of the row-major float panels (16-byte aligned rows)
#define NPAN 46    // panel rows: 15 (Qus^T) + 30 (I) + 1 (qu)
#define GEO 36     // floats of stage geometry: r[24] | Fc[6] | Fsum[3] | pad
// Two backward stages.  HBM-factor variants (four waves, three workgroups per CU): the value-function stage, phases 1-4 above.  Resident variants (eight waves, one
// workgroup per CU): the streaming SQUARE-ROOT stage -- the value function P = [Qss 0; 0 D] - W^T W is never formed.  With T = [B~ A~] (39 x 45, <= 4 non-zeros per
// column) and Z = W T, the next stage's [Quu Qus; Qsu Qss], qu, qs are  (cost / barrier terms + T^T [Qss 0; 0 D] T)  -  Z^T Z : the bracket needs nothing of this
// stage's factorisation and is built by the other waves WHILE it runs; what is left behind it is Z^T Z, one 16 x 16 tile per wave on the matrix cores as the pivot
// blocks come out (see "The streaming square-root stage" below).
#define SQ_TILE_WAVES 6        // consumer waves of the streaming stage that own a tile (and a share of the assembly): the two completion counts advance by this per stage
#define SQ_SPIN_MAX (1 << 20)  // looks at a progress word or a count before a waiting wave gives up (raises the failure flag: the pass ends as a failed factorisation)
#define SQ_PUB_FLOATS (10 * NPAN * 4 + 128 * 4)   // published W^T: ten pivot blocks x 46 panel rows x float4, and a slot per lane of two waves for lanes without a panel row
#define ZLD 36     // leading dim of Z^T (48 rows: 30 u-columns, 15 s-columns, the gradient column, 2 zero rows)
#define MSET (NU * RLD + NPAN * RLD + NS * 16)   // floats of one set QuuF | Pan | Qb (the resident variants hold two: stage k is factorised from set k & 1)
#define NTRI 256   // entries of the lower-triangle index table (the users need 210: 2x2 tiles of a 39x39)
// Rows of the 30 x 30 identity as 16-byte aligned 32-float windows of four constant strips: strip s = 68 floats with a 1 at position 32 + s; row m of the identity
// (32 floats) is strip m % 4 from float 32 + m % 4 - m on.  The factorisation's identity panel rows are LOADED like every other row (8 x ds_read_b128) instead of
// being built entry by entry: 120 instructions of the critical wave per stage.
#define ID_STRIP_LEN 68
#define ID_STRIPS (4 * ID_STRIP_LEN)
// Per-stage factor record (floats), in LDS or -- FG kernels -- in HBM scratch.  Phase 3 leaves column m of
// L^{-1} and column j of Ws = L^{-1} Qus in the registers of one lane, so both are stored transposed, one
// 16-byte-aligned row per lane:
//   [REC_UB, +576)  U = L^{-T}, upper triangle in 4x4 blocks: row m (block-row I = m / 4) holds columns 4I..31,
//                   4 (8 - I) contiguous floats.  Entries below the diagonal inside a diagonal block, columns
//                   30, 31 and rows 30, 31 are zero; row 31 is the zero block (REC_ZERO) that out-of-triangle
//                   reads are pointed at.
//   [REC_WT, +512)  16 rows x 32: row j < 15 = Ws[:, j], row 15 = lq = L^{-1} qu; the float4 index inside a row
//                   is XOR-swizzled with j & 7 (row reads of 16 consecutive j stay 2-way conflict-free)
// Column descriptors and barrier coefficients of a stage exist twice: stage k uses set k & 1, so that the set of
// stage k-1 can be written while stage k is being factorised
#define DSET_F (3 * NU + 3 * NS + 3 + 108)   // floats: Bval | Aval | arow
#define DSET_I (3 * NU + 3 * NS + 3)         // ints:   Brow | Arow
#define REC_UB 0
#define REC_WT 576
#define REC_ZERO 572
#define REC_N CMPC_REC_N
#ifndef CMPC_DELTA_UNROLL
#define CMPC_DELTA_UNROLL 4   // the same for the corrector sweep of the resident variants (2 -> 4: 354.9 k -> 356.2 k solves/s at B = 256, three rounds)
#endif
#ifndef CMPC_SWEEP_UNROLL
#define CMPC_SWEEP_UNROLL 4   // stages per trip of the sweep loops in the resident variants (A/B: 1 / 2 / 4 / 10 -> 184.4 / 186.1 / 189.8 / 188.0 k solves/s)
#endif
// The HBM-factor variants are compiled for three workgroups per CU (168 registers per lane): their LDS image fits three times into
// a CU up to N = 30 (52 KB at N = 20 and at N = 30).  Measured at N = 12, B = 8192, where 2, 3 and 4 all fit: 22.4 / 17.1 /
// 17.8 ms -- at four (128 registers) the spills of the factorisation and the sweeps cost more than the fourth workgroup brings.

namespace {

#ifdef CMPC_PROFILE
// diagnostic build only: per-phase shader-clock sums of workgroup 0
__device__ long long g_prof[128];
__device__ float g_trace[64 * 8];  // per iteration of workgroup 0: mu, ep, ec, step, ap, ad, sigma, mu_t
#define PROF_DECL long long pt_ = __builtin_amdgcn_s_memtime()
#define PROF(slot) do { long long n_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x == 0) g_prof[slot] += n_ - pt_; pt_ = n_; } while (0)
#define PROF2_DECL long long pt2_ = __builtin_amdgcn_s_memtime()
// (first lane of the wave that builds qu / the diagonal blocks in phase 2: the wave numbers depend on the thread count NT)
#define PROF3(slot) do { long long n_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 192 && blockIdx.x == 0) g_prof[slot] += n_ - pt2_; pt2_ = n_; } while (0)
#define PROF4(slot) do { long long n_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 128 && blockIdx.x == 0) g_prof[slot] += n_ - pt2_; pt2_ = n_; } while (0)
#define PROF2(slot) do { long long n_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x == 0) g_prof[slot] += n_ - pt2_; pt2_ = n_; } while (0)
// (first lane of the value-gradient wave of phase 4)
#define PROF5(slot) do { long long n_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 216 && blockIdx.x == 0) g_prof[slot] += n_ - pt2_; pt2_ = n_; } while (0)
// fire-and-forget stamp of the consumer waves (no wait on the atomic): slot 32 + 5 * wave + i
#define CPROF(i) do { if (ln == 0 && blockIdx.x == 0 && k < N) atomicAdd(reinterpret_cast<unsigned long long*>(&g_prof[32 + 5 * wv + (i)]), (unsigned long long)(__builtin_amdgcn_s_memtime() - pc0_)); } while (0)
// finer stamps of the consumers (slots 64 + 8 * wave + i)
#define CPROF2(i) do { if (ln == 0 && blockIdx.x == 0 && k < N) atomicAdd(reinterpret_cast<unsigned long long*>(&g_prof[64 + 8 * wv + (i)]), (unsigned long long)(__builtin_amdgcn_s_memtime() - pc0_)); } while (0)
// -DCMPC_PROFILE_LIGHT: only the stamps of the driver and the one per backward pass (the stamps inside a stage cost ~10 % of it)
#ifdef CMPC_PROFILE_LIGHT
#undef PROF2_DECL
#undef PROF2
#undef PROF3
#undef PROF4
#undef PROF5
#undef CPROF
#undef CPROF2
#define CPROF2(i)
#define PROF2_DECL
#define PROF2(slot)
#define PROF3(slot)
#define PROF4(slot)
#define PROF5(slot)
#define CPROF(i)
#define SQPROF_DECL
#define SQPROF(slot)
#else
#define SQPROF_DECL PROF_DECL
#define SQPROF(slot) PROF(slot)
#endif
#else
#define SQPROF_DECL
#define SQPROF(slot)
#define CPROF(i)
#define CPROF2(i)
#define PROF_DECL
#define PROF(slot)
#define PROF2_DECL
#define PROF2(slot)
#define PROF3(slot)
#define PROF4(slot)
#define PROF5(slot)
#endif

struct Ctx {
    CmpcIdx L;
    int N;
    const float* sp;     // parameter vector in LDS
    float *S, *U, *T, *Z;
    float* LAM;            // costates (they only enter the exact-Hessian term and the exported duals: float)
    float *dS, *dU, *dT, *dZ, *d;
    float *Lf;             // per-stage factor records (REC_N floats each)
    float *geoA;           // N x GEO
    float *P0, *Qb, *G, *QuuF, *Pan, *Bval, *Aval, *arow, *ybuf, *fpv, *fpn;
    float *idstrip;        // four strips of 68 floats, a single 1 in each: rows of the identity for the fused factorisation's panel (id_row)
    float *ZT;             // (resident variants) the published W^T of the streaming stage; there QuuF | Pan | Qb exist twice (stride MSET)
    double *QuuD1, *qs1;   // (resident variants) second set of the float64 diagonal blocks and of qs
    int *Brow, *Arow, *qmask;
    unsigned short* tri;
    double *QuuD, *pv, *pn, *qs, *Pd, *sig, *gco, *redd;
    float* red;
    int* flag;
    int* prog;             // (resident variants) progress words of the two factorising waves, one copy per lane
};

// LDS operations of one wave execute in issue order, so a write followed by reads of other lanes of the
// SAME wave needs no wait, only a fence for the compiler
__device__ inline void wave_lds_sync() { asm volatile("" ::: "memory"); }
// LDS addresses as 32-bit address-space-3 pointers, made opaque to the optimiser (see forward_sweep): a load through one is ds_read vaddr offset:imm
typedef __attribute__((address_space(3))) const float* ldsf_t;
typedef __attribute__((address_space(3))) float* ldsw_t;
__device__ inline ldsf_t lds_opaque(const float* p) { ldsf_t q = (ldsf_t)p; asm volatile("" : "+v"(q)); return q; }
__device__ inline ldsw_t lds_opaque_w(float* p) { ldsw_t q = (ldsw_t)p; asm volatile("" : "+v"(q)); return q; }
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ inline float4 lds_ld4(ldsf_t p)
{
    const v4f v = *reinterpret_cast<__attribute__((address_space(3))) const v4f*>(p);
    return make_float4(v[0], v[1], v[2], v[3]);
}

// ---- access to the per-stage factor records.  LDS in the resident variants.  In the HBM-factor variants they sit in global
// scratch and go through a buffer descriptor built once per phase from the wave-uniform base (a pointer argument of an
// out-of-line phase arrives in VGPRs: v_readfirstlane makes it provably uniform): buffer_load v, v_offset, s[rsrc], s_stage --
// the lane's offset stays a loop-invariant VGPR and the stage offset an SGPR, no per-access VALU address arithmetic.  (A generic
// pointer made every access a flat_ instruction on a 64-bit VALU-computed address: -2.3 % on configs 3 and 5; plain global-address-space
// accesses on the uniform base measure 1 % below the descriptor form.) ----
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <bool G>
struct RecRef;
template <>
struct RecRef<false> {
    float* p;
    __device__ RecRef(float* base, int, int k) : p(base + (size_t)REC_N * k) {}
    __device__ float ld(unsigned off) const { return p[off]; }
    __device__ float4 ld4(unsigned off) const { return *reinterpret_cast<const float4*>(p + off); }
    __device__ void st(unsigned off, float v) const { p[off] = v; }
    __device__ void st4(unsigned off, const float4& w) const { *reinterpret_cast<float4*>(p + off) = w; }
};
template <>
struct RecRef<true> {
    typedef __attribute__((address_space(1))) unsigned gu32_t;
    typedef __attribute__((address_space(1))) v4u gv4u_t;
    __amdgpu_buffer_rsrc_t r;
    unsigned so;   // byte offset of the stage (SGPR)
    char* gb;      // wave-uniform address of the stage's record, for the stores
    __device__ RecRef(float* base, int N, int k)
    {
        const unsigned long long a = (unsigned long long)base;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        r = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 4 * REC_N * N, 0x00020000);
        so = 4u * REC_N * (unsigned)k;
        gb = (char*)(((unsigned long long)hi << 32) | lo) + so;
    }
    __device__ float ld(unsigned off) const { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 4u * off, so, 0)); }
    __device__ float4 ld4(unsigned off) const
    {
        const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, 4u * off, so, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
    // Stores are plain global stores on the uniform address, NOT buffer stores: a 16-byte buffer_store with the stage offset in an
    // SGPR `soffset` delivered single wrong dwords (the data registers are re-used right behind the store; with a register soffset
    // the compiler inserts no wait state for that) -- not in every build: the shipped one happened to be right, the instrumented
    // build and a variant with the whole stage behind one call left 10-50 % of the problems of some batches unconverged.  Found by
    // dumping the records of both builds: identical value functions, a handful of differing record entries (profiles/r02_experiments_not_kept.txt).
    __device__ void st(unsigned off, float v) const { *(gu32_t*)(gb + 4u * off) = __float_as_uint(v); }
    __device__ void st4(unsigned off, const float4& w) const
    {
        const v4u v = {__float_as_uint(w.x), __float_as_uint(w.y), __float_as_uint(w.z), __float_as_uint(w.w)};
        *(gv4u_t*)(gb + 4u * off) = v;
    }
};

// carve the workgroup's LDS image (constants first, then doubles, then 16-byte aligned float panels).
// Pure address arithmetic: every phase function rebuilds it instead of receiving it, so that the
// out-of-line phases have a register allocation of their own.
template <bool FG>
__device__ inline void make_ctx(Ctx& c, char* smem, int N, float* fg_base)
{
    constexpr int KBYTES = (sizeof(CmpcConsts) + 15) & ~15;
    c.L.N = N;
    c.N = N;
    double* dp = reinterpret_cast<double*>(smem + KBYTES);
    c.QuuD = dp; dp += 90;
    c.pv = dp; dp += 40; c.pn = dp; dp += 40; c.qs = dp; dp += 16; c.Pd = dp; dp += 40;
    c.sig = dp; dp += NI; c.gco = dp; dp += NI; dp += 2 * NI;  // (second descriptor set)
    c.redd = dp; dp += 8;
    c.QuuD1 = c.QuuD; c.qs1 = c.qs;
    if (!FG) { c.QuuD1 = dp; dp += 90; c.qs1 = dp; dp += 16; }
    float* fp = reinterpret_cast<float*>(dp);
    // workspace of the backward sweep: QuuF | Pan | G | P0 | Qb (5667 floats).  The step arrays dS | dU | dT | dZ live in
    // the same bytes: they are written by the forward sweep and dead again when the next factorisation starts, while the
    // workspace is dead as soon as riccati_backward returns (10.7 KB at N = 20: what brings the HBM-factor image of N = 20
    // under a third of a CU's LDS).  NS (N+1) + NU N + 2 NI N <= 5335 floats for N <= CMPC_NMAX.
    c.QuuF = fp; fp += NU * RLD;
    c.Pan = fp; fp += NPAN * RLD;
    c.ZT = nullptr;
    if (!FG) {
        // resident variants: QuuF0 | Pan0 | Qb0 | QuuF1 | Pan1 | Qb1 | published W^T -- one stride (MSET) for everything a stage is assembled in, so that the set is an
        // immediate offset in the consumers' LDS instructions
        c.G = nullptr; c.P0 = nullptr;
        c.Qb = fp; fp += NS * 16;
        fp += MSET;
        c.ZT = fp; fp += SQ_PUB_FLOATS;   // the published W^T of the streaming stage, [block][panel row][4]
    } else {
    c.G = fp; fp += (NXA * GLD + 3) & ~3;                   // (sizes rounded to 16 bytes: ybuf and the LDS factor records
    c.P0 = fp; fp += (NXA * PLD + 3) & ~3; c.Qb = fp; fp += NS * 16;   //  behind them are read with ds_read_b128)
    }
    {
        float* vp = c.QuuF;
        c.dS = vp; vp += NS * (N + 1); c.dU = vp; vp += NU * N; c.dT = vp; vp += NI * N; c.dZ = vp;
    }
    c.ybuf = fp; fp += 96;          // 16-byte aligned vector staging of the sweeps
    c.idstrip = fp; fp += ID_STRIPS;
    if (!FG) { c.Lf = fp; fp += (size_t)REC_N * N; } else c.Lf = fg_base;
    c.sp = fp; fp += (c.L.np() + 3) & ~3;
    c.S = fp; fp += NS * (N + 1); c.U = fp; fp += NU * N;
    // slacks and multipliers: only ever touched by element-wise loops over all rows and by the descriptor builder that runs
    // under the factorisation, never on a critical path -> with the factors in HBM they live there too (behind the records)
    // whenever the horizon is longer than CMPC_TZ_LDS_NMAX: 10.5 KB of LDS less at N = 30, which is what lets three workgroups
    // of that horizon share a CU.  Up to N = 20 the image fits three times with them in LDS (52 KB), which is 1.5 % faster.
    if (!FG || N <= CMPC_TZ_LDS_NMAX) { c.T = fp; fp += NI * N; c.Z = fp; fp += NI * N; }
    else { c.T = fg_base + (size_t)REC_N * N; c.Z = c.T + NI * N; }
    c.d = fp; fp += NS * N;
    c.LAM = fp; fp += NS * (N + 1);
    c.geoA = fp; fp += GEO * N;
    c.Bval = fp; fp += 3 * NU; c.Aval = fp; fp += 3 * NS + 3;
    c.arow = fp; fp += 96 + 12; fp += DSET_F;  // (second descriptor set)
    c.fpv = fp; fp += 40; c.fpn = fp; fp += 40; c.red = fp; fp += 24;
    c.Brow = reinterpret_cast<int*>(fp); fp += 3 * NU;
    c.Arow = reinterpret_cast<int*>(fp); fp += 3 * NS + 3; fp += DSET_I;  // (second descriptor set)
    c.flag = reinterpret_cast<int*>(fp); fp += 4;
    c.prog = c.flag;
    if (!FG) { c.prog = reinterpret_cast<int*>(fp); fp += 128; }
    c.qmask = reinterpret_cast<int*>(fp); fp += CMPC_NMAX;
    c.tri = reinterpret_cast<unsigned short*>(fp); fp += NTRI / 2;   // (i, j) of the first NTRI lower-triangle entries
}

__device__ inline void use_desc_set(Ctx& c, int s)
{
    c.Bval += s * DSET_F; c.Aval += s * DSET_F; c.arow += s * DSET_F;
    c.Brow += s * DSET_I; c.Arow += s * DSET_I;
    c.sig += s * 2 * NI; c.gco += s * 2 * NI;
}

// U[m][a] (a >= 4 (m / 4)) lives at ub_row(m) + a - 4 (m / 4)
__device__ inline int ub_row(int m) { const int I = m >> 2; return REC_UB + 16 * (8 * I - I * (I - 1) / 2) + (m & 3) * 4 * (8 - I); }
__device__ inline int wt_idx(int j, int a) { return REC_WT + 32 * j + 4 * ((a >> 2) ^ (j & 7)) + (a & 3); }
// sum of the two 32-lane halves of a wave, in every lane (gfx950 v_permlane32_swap)
__device__ inline float half_sum(float v)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sum over aligned groups of 8 lanes, in every lane of the group (DPP, no LDS)
__device__ inline float oct_sum(float v)
{
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    return v;
}
__device__ inline float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

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
    // Explicit fused multiply-adds, here and in row_val / row_dot: these are inlined in half a dozen places (initial slacks, residuals, the barrier gradient z/t (a u - b + t)
    // of the descriptors, the slack step of the forward sweep) whose roundings must AGREE -- the gradient multiplies the residual by z/t, up to 1e9, and the forward
    // sweep subtracts the same residual again.  Left to the compiler, one inlined copy contracted a*b + c differently from the others (a variant of the descriptor
    // builder, round 3): one ulp of a u, times z/t, is a spurious gradient, and walking problems needed 12-40 iterations instead of 9.
    const float nmu = -prm.mu_fr;
    a0 = __builtin_fmaf(nmu, Rm(R, 0, 2), __builtin_fmaf(sy, Rm(R, 0, 1), sx * Rm(R, 0, 0)));
    a1 = __builtin_fmaf(nmu, Rm(R, 1, 2), __builtin_fmaf(sy, Rm(R, 1, 1), sx * Rm(R, 1, 0)));
    a2 = __builtin_fmaf(nmu, Rm(R, 2, 2), __builtin_fmaf(sy, Rm(R, 2, 1), sx * Rm(R, 2, 0)));
}

__device__ inline bool row_active(const Ctx& c, int k, int i) { return i < 32 ? true : qfree(c, k, (i - 32) % 6); }

// a_i^T u - b_i   (<= 0 feasible)
__device__ inline float row_val(const Ctx& c, const CmpcConsts& prm, int k, int i, const float* u)
{
    if (i < 32) {
        float a0, a1, a2;
        fric_row(c, prm, k, i, a0, a1, a2);
        const float* f = u + 3 * (i >> 2);
        return __builtin_fmaf(a2, f[2], __builtin_fmaf(a1, f[1], a0 * f[0]));
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
        return __builtin_fmaf(a2, f[2], __builtin_fmaf(a1, f[1], a0 * f[0]));
    }
    if (i < 38) return du[24 + i - 32];
    return -du[24 + i - 38];
}

// ---- The inequality rows of all stages as per-thread SLOTS, a kind at a time (compile-time horizons): slots 0 .. RF-1 are friction rows (32 per stage: stage and row are a
// shift and a mask, always in use), slots RF .. RF+RB-1 box rows of the landing offsets (12 per stage, in use where the offset is free).  With the 44 rows of a stage in one
// index space (e / NI, e % NI, row_val's three-way branch) every wave of the row passes walked through all three row formulas on every trip.  Branch-free on clamped
// indices: a slot beyond the last row computes row 0 of its kind and is masked by `in`.  Same arithmetic as row_val / row_dot (explicit fused multiply-adds: see fric_row).
//   e: index of the row in T, Z, dT, dZ;  val = a^T u - b;  dot = a^T du (DOT);  returns in | 2 * in-use ----
template <int NTR, int NC>
struct RowSlots {
    static constexpr int RF = (32 * (NC > 0 ? NC : 1) + NTR - 1) / NTR, RB = (12 * (NC > 0 ? NC : 1) + NTR - 1) / NTR, R = RF + RB;
};
template <int NTR, int NC, int Q, bool DOT>
__device__ __forceinline__ int row_slot(const Ctx& c, const CmpcConsts& prm, int tr, int& e, float& val, float& dot)
{
    constexpr int RF = RowSlots<NTR, NC>::RF;
    if constexpr (Q < RF) {
        const int ef = tr + NTR * Q;
        const bool in = ef < 32 * NC;
        const int efc = in ? ef : 0;
        const int k = efc >> 5, i = efc & 31;
        e = NI * k + i;
        float a0, a1, a2;
        fric_row(c, prm, k, i, a0, a1, a2);
        const float* f = c.U + NU * k + 3 * (i >> 2);
        val = __builtin_fmaf(a2, f[2], __builtin_fmaf(a1, f[1], a0 * f[0]));
        if (DOT) {
            const float* g = c.dU + NU * k + 3 * (i >> 2);
            dot = __builtin_fmaf(a2, g[2], __builtin_fmaf(a1, g[1], a0 * g[0]));
        }
        return in ? 3 : 0;
    } else {
        const int eb = tr + NTR * (Q - RF);
        const bool in = eb < 12 * NC;
        const int ebc = in ? eb : 0;
        const int k = ebc / 12, ib = ebc - 12 * k;
        const bool upper = ib < 6;
        const int m = upper ? ib : ib - 6;
        e = NI * k + 32 + ib;
        const int ct = m >= 3 ? 1 : 0;
        const float bound = c.sp[(upper ? c.L.pUp(ct) : c.L.pLo(ct)) + 3 * k + m - 3 * ct];
        const float u = c.U[NU * k + 24 + m];
        val = upper ? u - bound : bound - u;
        if (DOT) { const float du = c.dU[NU * k + 24 + m]; dot = upper ? du : -du; }
        return in ? (qfree(c, k, m) ? 3 : 1) : 0;
    }
}
// (a compile-time loop over the slots: F(integral_constant<int, Q>) for Q = 0 .. R-1)
template <int Q0, int Q1, typename F>
__device__ __forceinline__ void for_slots(F&& f)
{
    if constexpr (Q0 < Q1) {
        f(std::integral_constant<int, Q0>{});
        for_slots<Q0 + 1, Q1>(f);
    }
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

__device__ inline float readlane_f(float x, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }
// a value every lane holds, moved to an SGPR: the results of the block reductions (residuals, step lengths, barrier parameter) live across
// the out-of-line phase calls of the driver -- as VGPRs they were spilled to scratch memory there, as SGPRs they cost a v_writelane at worst
__device__ inline float uniform_f(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ inline double uniform_d(double x)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
__device__ inline double readlane_d(double x, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// ---- reductions over the workgroup.  Inside a wave: DPP (quad swaps, half-row and row mirrors leave every row of 16 lanes
// reduced, v_readlane joins the four rows) -- no LDS crossbar, ~12 instructions; __shfl_xor is six dependent ds_bpermute,
// twelve for a double.  Across waves: one LDS slot per wave and value; values that are needed together share the two barriers. ----
template <int CTRL>
__device__ inline float dpp_f(float v) { return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, true)); }
template <int CTRL>
__device__ inline double dpp_d(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// running maximum of |v| that a NaN turns into +inf (fmaxf alone drops a NaN operand: a non-finite residual or step would
// read as zero and pass the termination test)
__device__ inline float amax_nan(float m, float v) { return (v == v) ? fmaxf(m, fabsf(v)) : INFINITY; }
// complementarity target of one row in the tail polish: mu_min for rows with multiplier >= 1 (z / t = z^2 / mu stays <= what the main
// loop lives with), mu_min z^2 below -- rows whose multiplier vanishes are followed further down their path -- never below 1e-4 mu_min
__device__ inline float row_target(float z, float mu_min) { return fmaxf(mu_min * fminf(1.f, z * z), 1e-4f * mu_min); }
__device__ inline float wave_max(float v)
{
    v = fmaxf(v, dpp_f<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_f<0x4E>(v));    // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_f<0x141>(v));   // row_half_mirror
    v = fmaxf(v, dpp_f<0x140>(v));   // row_mirror
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}
__device__ inline double wave_sum(double v)
{
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x141>(v);
    v += dpp_d<0x140>(v);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
// maxima of up to three values at once (red: 3 slots per wave)
template <int NT, int NV>
__device__ inline void block_maxn(float (&v)[NV], float* red, int tid)
{
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = wave_max(v[q]);
    __syncthreads();
    if ((tid & 63) == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) red[3 * (tid >> 6) + q] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        float r = red[q];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) r = fmaxf(r, red[3 * w + q]);
        v[q] = uniform_f(r);
    }
}
template <int NT>
__device__ inline double block_sum(double v, double* red, int tid)
{
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double r = red[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) r += red[w];
    return uniform_d(r);
}
// two maxima and one float64 sum behind one pair of barriers (the residual pass)
template <int NT>
__device__ inline void block_max2_sum(float& m0, float& m1, double& sm, float* red, double* redd, int tid)
{
    m0 = wave_max(m0); m1 = wave_max(m1); sm = wave_sum(sm);
    __syncthreads();
    if ((tid & 63) == 0) { red[3 * (tid >> 6)] = m0; red[3 * (tid >> 6) + 1] = m1; redd[tid >> 6] = sm; }
    __syncthreads();
    float r0 = red[0], r1 = red[1];
    double rs = redd[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) { r0 = fmaxf(r0, red[3 * w]); r1 = fmaxf(r1, red[3 * w + 1]); rs += redd[w]; }
    m0 = uniform_f(r0); m1 = uniform_f(r1); sm = uniform_d(rs);
}

// ---- geometry of every stage for the current iterate: r_cj = R c_j + pos_c - com (8x3), Fc (2x3,
// plain corner sums), Fsum (gam-weighted sum) ----
template <int NT>
__device__ inline void all_geo(const Ctx& c, const CmpcConsts& prm, int tid)
{
    // (a loop per kind: with the 33 entries of a stage in one index space every wave walked through all three formulas on both of its trips)
    for (int e = tid; e < c.N * 24; e += NT) {
        const int k = e / 24, t = e - 24 * k;
        const float* s = c.S + NS * k;
        const int ct = t / 12, j = (t % 12) / 3, i = t % 3;
        const float* R = c.sp + c.L.pR(ct) + 9 * k;
        const float* cn = prm.corners + 12 * ct + 3 * j;
        c.geoA[GEO * k + t] = Rm(R, i, 0) * cn[0] + Rm(R, i, 1) * cn[1] + Rm(R, i, 2) * cn[2] + s[9 + 3 * ct + i] - s[i];
    }
    for (int e = tid; e < c.N * 9; e += NT) {
        const int k = e / 9, t = e - 9 * k, i = t % 3;
        const float* u = c.U + NU * k;
        const float f0 = u[i] + u[3 + i] + u[6 + i] + u[9 + i], f1 = u[12 + i] + u[15 + i] + u[18 + i] + u[21 + i];
        // t = 0..2: Fc of foot 0, 3..5: Fc of foot 1 (plain corner sums), 6..8: Fsum (gam-weighted)
        c.geoA[GEO * k + 24 + t] = t < 3 ? f0 : (t < 6 ? f1 : gam_of(c, 0, k) * f0 + gam_of(c, 1, k) * f1);
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


// ---- fused Cholesky + panel solve, one wave, rows in registers, 3x3 pivot blocks ----
// lanes 0..29: row `lane` of the symmetric matrix (v[c], c <= lane, float; its own 3x3 diagonal
// block in dd[], float64).  Lanes >= 30: a row of the panel [Qus | I | qu]^T (30 floats).
// On exit v[] holds the row of L (lanes < 30) or of (L^{-1} [Qus | I | qu])^T.  Returns true if a
// pivot was not positive (uniform across the wave).
//
// Lanes < 30 enter with v[c] = Quu[lane][c] for c outside their own 3x3 diagonal block and 0 inside it; the
// diagonal block itself is in dd (float64).  Updates of a diagonal block by earlier blocks are moderate
// numbers and accumulate in those zeroed float slots; they are folded into dd when the block becomes the
// pivot block.  The pivot block is broadcast (v_readlane), factorised in float64 by every lane, and applied:
// three columns of L per step, then one rank-3 update of the trailing columns.
// 1 / x in float64 from the v_rcp_f32 seed and one Newton step (seed error 1e-7 -> 1e-14).  Pivots are far from the
// float32 range limits: no scaling.
__device__ inline double rcp_d(double x)
{
    const double r = (double)__builtin_amdgcn_rcpf((float)x);
    return r * (2.0 - x * r);
}
// The row lives in eight 4-column groups (acc[g][i] = entry 4 g + i; columns 30, 31 are padding): the accumulators of the trailing update.
//
// Trailing update on the matrix pipe (round 4).  v_mfma_f32_4x4x1_16b_f32 computes sixteen independent 4 x 4 outer products, block q on lanes
// 4 q .. 4 q + 3: acc_l[i] += A[lane 4 q + i] B[lane l].  With cbsz:4 abid:g the A values of block g serve ALL sixteen blocks:
//     acc_l[i] += A[lane 4 g + i] * B[lane l]        for every lane l, i = 0..3
// (checked bit for bit against fmaf on MI355X: tools/micro/mfma4x4_probe.hip, profiles/r04_mfma4x4_probe.txt).  The multipliers of a pivot
// column, L[c][j], are the x of lanes c = j0 + 3 .. 29 -- the L rows live in the lanes of their row number -- so with A = -x (zero outside those
// lanes) and B = the lane's own x, ONE instruction applies a pivot column to four trailing columns of all 64 rows: three per group and pivot block,
// 126 per factorisation at 8 cycles each, where rounds 1-3 issued 6 v_readlane + 3 v_pk_fma_f32 per column PAIR (525 + 260 instructions, a third of
// the phase).  Measured in isolation: 27 cycles per four columns and pivot block against 81.  f32-input MFMA is an exact fmaf chain (MI355X_MICROARCH.md),
// and the order per entry (pivot columns j0, j0 + 1, j0 + 2) is the one the packed FMAs had: the factors are bit-identical.
#define VE(j) acc[(j) >> 2][(j) & 3]
// ONE: the whole factorisation on one wave.  76 rows do not fit 64 lanes, but the L rows of the first blocks are finished
// (and never read again: multipliers only come from rows below the pivot block, and L itself is not kept) long before the
// identity rows of the last columns start to differ from unit vectors: at block LATE_B the lanes of L rows 0..NLATE-1 are
// re-used for the identity rows LATE_M0..NU-1, which is exact because such a row is e_m until block m / 3.
#define LATE_B 6
#define LATE_M0 18
#define NLATE (NU - LATE_M0)
// where row m of the identity starts in the strips (float offset, a multiple of 4)
__device__ inline int id_row(int m) { return ID_STRIP_LEN * (m & 3) + 32 + (m & 3) - m; }
// rank-1 update of the column groups G0 .. 7 (av: -x of the L rows below the pivot block, 0 elsewhere; bv: the lane's own x)
template <int G0, int GE = 8>
__device__ __forceinline__ void trail_rank1(v4f (&acc)[8], float av, float bv)
{
#define CMPC_R1(G) if constexpr (G >= G0 && G < GE) acc[G] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[G], 4, G, 0);
    CMPC_R1(0) CMPC_R1(1) CMPC_R1(2) CMPC_R1(3) CMPC_R1(4) CMPC_R1(5) CMPC_R1(6) CMPC_R1(7)
#undef CMPC_R1
}
// PUB (streaming square-root stage): after every pivot block the finished entries x0..x2 of this lane's panel row go to pub[(46 b + prow) * 4 ..]
// and the wave's progress word *pflag is set to seq0 + b + 1.  LDS operations of one wave execute in issue order: whoever sees the flag sees the data.
struct CholPub { float* pub; int stride; int* pflag; int seq0; float* publate; const float* idstrip; };
// Where a lane's finished row goes: one row per lane, 16-byte stores -- the record row for the sweeps and (HBM-factor variants) the panel row for phase 4.
// The four columns of chunk Q are final as soon as the pivot block holding the last of them is done, so the stores are issued from inside the block loop and
// drain under the remaining blocks (round 4: behind the last block they were 1.3 k cycles of the critical wave's stage); chunks 6, 7 follow the loop.
// Float4s left of an identity row's diagonal block go to row 30 of U, a row whose only readers are lanes that discard what they compute (REC_ZERO is row 31).
template <bool G, bool PUB>
struct RowStore {
    static constexpr bool EARLY = !G;   // (measured at B = 256: 247.2 k solves/s with the early stores, 244.8 k without; records in HBM: see chol_block)
    const RecRef<G>& rec;
    float* prow_p;      // panel row (not PUB)
    float sc;           // its scale: the p rows of the panel are kept pre-multiplied by -D
    unsigned rrow;
    int I, sw;
    bool on;
    const float* Dp;    // the force-rate weights D[3] (in LDS: an indexed load -- three members and a select chain were turned into an indexed private array, i.e. scratch memory)
    float* Pan;
    __device__ __forceinline__ RowStore(const RecRef<G>& r, float* Pan_, const float* Dp_) : rec(r), prow_p(Pan_), sc(1.f), rrow(0), I(0), sw(0), on(false), Dp(Dp_), Pan(Pan_) {}
    // this lane stores panel row srow (Qus^T rows 0..14, identity rows NS..NS+29 = columns of L^{-1}, the lq row NPAN-1)
    __device__ __forceinline__ void set_row(int srow)
    {
        const int m = srow - NS;                       // identity rows: column m of L^{-1}
        const bool isId = srow >= NS && srow < NS + NU;
        sc = 1.f;
        if (!PUB && isId) sc = m < NF ? -Dp[m % 3] : 0.f;
        const int jw = srow < NS ? srow : 15;          // Ws column j, or the lq row
        I = isId ? (m >> 2) : 0;
        prow_p = Pan + srow * RLD;
        rrow = isId ? ub_row(m) - 4 * I : REC_WT + 32 * jw;
        sw = isId ? 0 : (jw & 7);
        on = true;
    }
    template <int Q>
    __device__ __forceinline__ void chunk(const v4f& a) const
    {
        if (!on) return;
        float4 w;
        w.x = a[0]; w.y = a[1];
        w.z = Q < 7 ? a[2] : 0.f;   // (columns 30, 31 are padding: stored as zeros)
        w.w = Q < 7 ? a[3] : 0.f;
        if (!PUB) *reinterpret_cast<float4*>(prow_p + 4 * Q) = make_float4(sc * w.x, sc * w.y, sc * w.z, sc * w.w);   // (PUB: nobody reads the panel copy)
        rec.st4(Q >= I ? rrow + 4 * (Q ^ sw) : ub_row(30), w);
    }
};
// pivot block B (columns 3 B .. 3 B + 2).  Returns true if a pivot was not positive.
// FIX: every landing offset of the stage is fixed (double stance: fixedmask == 63, decided once per stage by the caller).  Their rows and columns are identity rows
// and columns: blocks 8, 9 have nothing to do (that much is also tested at run time below), AND the trailing updates of column groups 6, 7 (columns 24..29 + padding) add
// exact zeros -- the update of column 24 + i is (-x of L row 24 + i) x (the lane's x), and the x of an identity row is zero in every block: 48 of the stage's 126
// matrix-pipe instructions, left out at compile time.  (Left out behind a branch per block the same saving was a loss: profiles/r04_experiments_not_kept.txt, item 18.)
template <int B, bool PUB, bool FIX, typename ST>
__device__ __forceinline__ bool chol_block(v4f (&acc)[8], double (&dd)[3], int lane, int fixedmask, CholPub& pb, ST& st)
{
    constexpr int j0 = 3 * B;
    if constexpr (FIX && B >= 8) return false;
    if (B == LATE_B) {
        if (lane < NLATE) {
            const float* idr = pb.idstrip + id_row(LATE_M0 + lane);
#pragma unroll
            for (int q4 = 0; q4 < 8; ++q4) {
                const float4 t4 = *reinterpret_cast<const float4*>(idr + 4 * q4);
                acc[q4] = v4f{t4.x, t4.y, t4.z, t4.w};
            }
            if (PUB) { pb.pub = pb.publate; pb.stride = NPAN * 4; }   // (the identity row this lane now holds: its entries of the earlier blocks are zeros, kept zero in the buffer)
            if constexpr (ST::EARLY) st.set_row(NS + LATE_M0 + lane);   // (its chunks 0..3, left of the diagonal block, were never stored: they belong to the discarded row anyway)
        }
    }
    // (opaque copy: the lane predicates are recomputed per block instead of living in hoisted, spilled SGPR pairs)
    int ln = lane;
    asm volatile("" : "+v"(ln));
    if ((unsigned)(ln - j0) < 3u) {
        dd[0] += (double)VE(j0);
        dd[1] += (double)VE(j0 + 1);
        dd[2] += (double)VE(j0 + 2);
    }
    // a stance foot's landing offsets are identity rows and columns: nothing to do
    if (j0 >= NF && ((fixedmask >> (j0 - NF)) & 7) == 7) return false;
    const double d00 = readlane_d(dd[0], j0);
    const double d10 = readlane_d(dd[0], j0 + 1), d11 = readlane_d(dd[1], j0 + 1);
    const double d20 = readlane_d(dd[0], j0 + 2), d21 = readlane_d(dd[1], j0 + 2), d22 = readlane_d(dd[2], j0 + 2);
    // Float64 only where cancellation decides the answer: the Schur pivots p1, p2 of the block (barrier terms ~1e9 next to
    // cost curvature ~20) through two accurate reciprocals.  The entries of L are float32 numbers anyway (rows are stored
    // and applied in float32): they come from the float32 casts of the pivots and v_rsq_f32.
    // (Elimination order on purpose.  The division-free form -- p1 = n1 / a, p2 = det / n1 from the minors n1 = a c - b^2 and det -- halves the dependent chain
    // and was 1 % faster, but a corner with ONE active friction row has a rank-one 1e9 term in its block: the determinant then cancels twice, 1e27 against
    // 4e11, and p2 loses every digit; the push goldens went from 1.6e-5 to 1.06e-4 on the forces.  profiles/r04_experiments_not_kept.txt, item 13.)
    const double i00 = rcp_d(d00);
    const double m10 = d10 * i00, m20 = d20 * i00;
    const double p1 = d11 - m10 * d10;
    const double t21 = d21 - m10 * d20;
    const double i11 = rcp_d(p1);
    const double m21 = t21 * i11;
    const double p2 = (d22 - m20 * d20) - m21 * t21;
    const float p0f = (float)d00, p1f = (float)p1, p2f = (float)p2;
    const float r00 = __builtin_amdgcn_rsqf(p0f), r11 = __builtin_amdgcn_rsqf(p1f), r22 = __builtin_amdgcn_rsqf(p2f);
    const float l10 = (float)d10 * r00, l20 = (float)d20 * r00, l21 = (float)t21 * r11;
    // a non-positive pivot: v_rsq_f32 of a negative number is NaN, of zero inf; p1 and p2 inherit a bad d00 through i00
    // -- which then is NaN in every x of the block, in every trailing column and in every later pivot: tested where the factorisation ends (blocks 7..9; a stance
    // foot's blocks 8, 9 are skipped), not ten times
    const bool bad = B >= 7 && (!(p0f > 0.f) || !(p1f > 0.f) || !(p2f > 0.f));
    // rows below the block and panel rows: x = v[j0..j0+2] L_bb^{-T}
    const float x0 = VE(j0) * r00;
    const float x1 = (VE(j0 + 1) - x0 * l10) * r11;
    const float x2 = (VE(j0 + 2) - x0 * l20 - x1 * l21) * r22;
    // (the block's own three lanes get meaningless x here -- their slots of the block are zero -- and that is fine: L itself
    // is not kept, multipliers only come from rows below the pivot block, so a lane's registers are dead once its block is done)
    VE(j0) = x0; VE(j0 + 1) = x1; VE(j0 + 2) = x2;
    if (PUB) {
        // (no lane predicates here: a lane without a panel row writes to a slot of its own behind the blocks, and every lane keeps its own copy of the progress word)
        *reinterpret_cast<float4*>(pb.pub + pb.stride * B) = make_float4(x0, x1, x2, 0.f);
        // The progress word is a RELAXED store behind a compiler fence, on purpose.  A release store is an s_waitcnt lgkmcnt(0) in front of it -- the critical
        // wave parked until its own 16-byte write has landed, every pivot block: measured 1.4 % of the headline (236.6 k against 233.4 k solves/s, one gpurun
        // call).  What makes the relaxed form correct on this hardware: the DS instructions of one wave are executed by the LDS unit in issue order, so the
        // word cannot become visible before the data written just before it; the readers load it with acquire order (lds_peek).
        asm volatile("" ::: "memory");
        __hip_atomic_store(pb.pflag, pb.seq0 + B + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler sinks the two stores behind two thirds of the block's matrix-pipe instructions: 100+ cycles later for every reader)
    }
    // rank-3 update of the trailing columns j0 + 3 .. 29 on the matrix pipe (see above): multipliers from the lanes of the L rows below the block.
    // Pivot-column-major over the groups: the three updates of one group are a dependent chain (20 cycles apart alone, 8 when interleaved).
    if constexpr (j0 + 3 < NU) {
        const bool below = (unsigned)(ln - (j0 + 3)) < (unsigned)(NU - (j0 + 3));
        const float a0 = below ? -x0 : 0.f, a1 = below ? -x1 : 0.f, a2 = below ? -x2 : 0.f;
        constexpr int G0 = (j0 + 3) >> 2;
        constexpr int GE = FIX ? 6 : 8;
        trail_rank1<G0, GE>(acc, a0, x0);
        trail_rank1<G0, GE>(acc, a1, x1);
        trail_rank1<G0, GE>(acc, a2, x2);
    }
    // chunks whose last column this block finished (behind the matrix-pipe instructions: the stores drain while those execute).  LDS records only.  With the
    // records in HBM the same early stores are SLOWER (config 3 412.4 k -> 405.8 k solves/s, config 5 266.6 k -> 261.6 k: global_store_dwordx4 from inside the
    // critical wave's block loop, three workgroups per CU competing for the memory pipeline), and the panel copy of an identity row that takes over a lane at
    // LATE_B would need its chunks 0..3 written as zeros (phase 4 reads them; a first attempt left them unwritten and delivered wrong rows -- not, as first
    // thought, a register hazard): those variants store after the last block.
    if constexpr (ST::EARLY) {
        if constexpr (B == 1) st.template chunk<0>(acc[0]);
        if constexpr (B == 2) st.template chunk<1>(acc[1]);
        if constexpr (B == 3) st.template chunk<2>(acc[2]);
        if constexpr (B == 5) st.template chunk<3>(acc[3]);
        if constexpr (B == 6) st.template chunk<4>(acc[4]);
        if constexpr (B == 7) st.template chunk<5>(acc[5]);
    }
    return bad;
}
// ---- fused Cholesky + panel solve, one wave, rows in registers, 3x3 pivot blocks (see the comment block above rcp_d) ----
template <bool PUB, bool FIX, typename ST>
__device__ __forceinline__ bool chol_solve_fused(v4f (&acc)[8], double (&dd)[3], int lane, int fixedmask, ST& st, const float* idstrip, float* pub = nullptr,
                                                 int pubstride = 0, int* pflag = nullptr, int seq0 = 0, float* publate = nullptr)
{
    CholPub pb{pub, pubstride, pflag, seq0, publate, idstrip};
    bool bad = false;
    bad |= chol_block<0, PUB, FIX>(acc, dd, lane, fixedmask, pb, st);
    bad |= chol_block<1, PUB, FIX>(acc, dd, lane, fixedmask, pb, st);
    bad |= chol_block<2, PUB, FIX>(acc, dd, lane, fixedmask, pb, st);
    bad |= chol_block<3, PUB, FIX>(acc, dd, lane, fixedmask, pb, st);
    bad |= chol_block<4, PUB, FIX>(acc, dd, lane, fixedmask, pb, st);
    bad |= chol_block<5, PUB, FIX>(acc, dd, lane, fixedmask, pb, st);
    bad |= chol_block<6, PUB, FIX>(acc, dd, lane, fixedmask, pb, st);
    bad |= chol_block<7, PUB, FIX>(acc, dd, lane, fixedmask, pb, st);
    bad |= chol_block<8, PUB, FIX>(acc, dd, lane, fixedmask, pb, st);
    bad |= chol_block<9, PUB, FIX>(acc, dd, lane, fixedmask, pb, st);
    if constexpr (ST::EARLY) {
        st.template chunk<6>(acc[6]);
        st.template chunk<7>(acc[7]);
    }
    return bad;
}

// phase 3 of a backward stage, kept out of line so that its ~40 VGPRs of matrix rows and its
// stream of v_readlane broadcasts get a register allocation of their own
// FIXSEL: which copy of the ten blocks (chol_block's FIX): 1 / 0 chosen by the caller at compile time (stage_mid of the HBM-factor variants exists in both forms and its
// caller branches: inside ONE function the second copy cost 17 callee-saved registers at 168), -1: chosen here, once per stage, by a scalar branch (streaming stage)
template <bool G, bool PUB = false, int FIXSEL = -1>
__device__ __forceinline__ void stage_factor(const float* QuuF, const double* QuuD, float* Pan, const RecRef<G>& rec, const float* idstrip,
                                          const float* Dp, int* flag, int tid, int fixedmask, float* pub = nullptr, int* pflag = nullptr, int seq0 = 0)
{
    const int lane = tid & 63, wv = tid >> 6;
    // panel row of lanes >= 30.  Two waves: rows 0..33 on wave 0, 34..45 on wave 1 (both repeat the L rows).  One wave: Qus rows
    // 0..14, the qu row (NPAN - 1), identity rows m = 0..17; identity rows 18..29 take over lanes 0..11 at block LATE_B.
    const int pi = lane - 30;
    const int prow = pi < NS ? pi : (pi == NS ? NPAN - 1 : pi - 1);
    const bool isL = lane < NU;
    const bool active = isL || prow < NPAN;
    v4f acc[8];
    double dd[3] = {0.0, 0.0, 0.0};
    PROF2_DECL;
    const bool idrow = !isL && prow >= NS && prow < NS + NU;
    {
        const float* src = isL ? QuuF + lane * RLD : (idrow ? idstrip + id_row(prow - NS) : Pan + (active ? prow : 0) * RLD);
#pragma unroll
        for (int q4 = 0; q4 < 8; ++q4) {
            const float4 t4 = *reinterpret_cast<const float4*>(src + 4 * q4);
            acc[q4] = v4f{t4.x, t4.y, t4.z, t4.w};
        }
        if (isL) {
            dd[0] = QuuD[9 * (lane / 3) + 3 * (lane % 3) + 0];
            dd[1] = QuuD[9 * (lane / 3) + 3 * (lane % 3) + 1];
            dd[2] = QuuD[9 * (lane / 3) + 3 * (lane % 3) + 2];
        }
    }
    PROF2(28);
    // PUB: where this lane's entries of a pivot block go: its panel row in block 0 (+ NPAN * 4 * b per block), or -- lanes without a panel row -- a slot of
    // their own behind the ten blocks (block stride 0)
    float* pubp = pub;
    if (PUB) pubp = (!isL && active) ? pub + prow * 4 : pub + (10 * NPAN + 64 * wv + lane) * 4;
    RowStore<G, PUB> st(rec, Pan, Dp);
    typedef RowStore<G, PUB> ST;
    if (ST::EARLY && !isL && active) st.set_row(prow);
    // (two copies of the ten blocks, chosen once per stage by a scalar branch: see chol_block)
    float* const publate = PUB ? pub + (NS + LATE_M0 + (lane < NLATE ? lane : 0)) * 4 : nullptr;
    const int pubstride = (!isL && active) ? NPAN * 4 : 0;
    bool bad;
    if constexpr (FIXSEL == 1) bad = chol_solve_fused<PUB, true>(acc, dd, lane, fixedmask, st, idstrip, pubp, pubstride, pflag, seq0, publate);
    else if constexpr (FIXSEL == 0) bad = chol_solve_fused<PUB, false>(acc, dd, lane, fixedmask, st, idstrip, pubp, pubstride, pflag, seq0, publate);
    else {
        const bool allfixed = (__builtin_amdgcn_readfirstlane(fixedmask) & 63) == 63;   // (wave-uniform by construction; made provably so: a scalar branch)
        bad = allfixed ? chol_solve_fused<PUB, true>(acc, dd, lane, fixedmask, st, idstrip, pubp, pubstride, pflag, seq0, publate)
                       : chol_solve_fused<PUB, false>(acc, dd, lane, fixedmask, st, idstrip, pubp, pubstride, pflag, seq0, publate);
    }
    PROF2(29);
    if (bad && tid == 0) *flag = 1;
    if constexpr (!ST::EARLY) {
        // (records in HBM: every chunk after the last block, and nothing of the store's addressing alive across the block loop -- the 168-register
        //  variants have no room for it: held there it pushed stage_mid into the callee-saved registers and their scratch saves)
        const bool late = lane < NLATE;                    // this lane now holds identity row LATE_M0 + lane
        if (late || (!isL && active)) {
            st.set_row(late ? NS + LATE_M0 + lane : prow);
            st.template chunk<0>(acc[0]); st.template chunk<1>(acc[1]); st.template chunk<2>(acc[2]); st.template chunk<3>(acc[3]);
            st.template chunk<4>(acc[4]); st.template chunk<5>(acc[5]); st.template chunk<6>(acc[6]); st.template chunk<7>(acc[7]);
        }
    }
    PROF2(30);
}

// ---- one stage of the backward sweep, as inline bodies that the out-of-line entry points `stage_mid` and `stage_post_pre` string together (the
// stage must stay out of the kernel body: inlined into its loops it needs > 512 VGPRs and spills to scratch):
//   stage_pre_body   phases 1-2: G = P [B;E], then Quu / Qus / Pd / qu
//   stage_factor     phase 3 on wave 0 (waves 0-1 in the resident variants): fused Cholesky + panel solve
//   stage_qss_body   Qss, qs of this stage and (stage_desc_body) the descriptors of the next on the other waves meanwhile
//   stage_post_body  phase 4: P <- [Qss 0; 0 D] - W^T W, value gradient ----
// ---- column descriptors of A_k and B_k (closed forms; <= 3 non-zeros per column), barrier coefficients z/t and the
// friction rows, the exact-Hessian block: 128 threads (t = 0..127), into the descriptor set selected in c ----
// (the four roles of stage_desc_body -- t ranges 0..29, 32..46, 48..91, 96..104 -- can be given a wave each: role r of lane l is t = base_r + l)
__device__ inline int desc_role_t(int role, int lane)
{
    const int base = role == 0 ? 0 : (role == 1 ? 32 : (role == 2 ? 48 : 96));
    const int n = role == 0 ? NU : (role == 1 ? NS : (role == 2 ? NI : 9));
    return lane < n ? base + lane : 127;   // (127: no role)
}
// the roles of 128 threads (two waves) packed so that each formula is walked through by ONE wave: B columns, A columns and the exact-Hessian block on the first
// (30 + 15 + 9 lanes), the 44 inequality rows on the second.  (With t = the thread's number the rows straddled both waves -- t = 48..91 -- and both executed the
// row formulas, float64 divisions included.)
__device__ inline int desc_pack_t(int tl)
{
    return tl < NU ? tl : (tl < NU + NS ? 32 + tl - NU : (tl < NU + NS + 9 ? 96 + tl - (NU + NS) : ((tl >= 64 && tl < 64 + NI) ? 48 + tl - 64 : 127)));
}
__device__ inline void stage_desc_body(const Ctx& c, const CmpcConsts& prm, int t, int k, bool use_exact, float cmu)
{
    const float* u = c.U + NU * k;
    const float* geo = c.geoA + GEO * k;
        // ---- phase 0: column descriptors of A and B, barrier coefficients ----
        if (t < NU) {
            int r0, r1, r2;
            float v0, v1, v2;
            if (t < NF) {
                const int ct = t / 12, a = t % 3, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
                const float g = prm.dt * gam_of(c, ct, k);
                const float* r = geo + 3 * (t / 3);
                r0 = 3 + a; v0 = g;
                r1 = 6 + a1; v1 = g * r[a2];
                r2 = 6 + a2; v2 = -g * r[a1];
            } else {
                const int q = t - 24, ct = q / 3, m = q % 3;
                const float* R = c.sp + c.L.pR(ct) + 9 * k;
                const float g1 = qfree(c, k, q) ? 1.f - gam_of(c, ct, k) : 0.f;
                r0 = 9 + 3 * ct; r1 = r0 + 1; r2 = r0 + 2;
                v0 = g1 * Rm(R, 0, m); v1 = g1 * Rm(R, 1, m); v2 = g1 * Rm(R, 2, m);
            }
            c.Brow[3 * t] = r0; c.Brow[3 * t + 1] = r1; c.Brow[3 * t + 2] = r2;
            c.Bval[3 * t] = v0; c.Bval[3 * t + 1] = v1; c.Bval[3 * t + 2] = v2;
        } else if (t >= 32 && t < 32 + NS) {
            const int j = t - 32;
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
        } else if (t >= 48 && t < 48 + NI) {
            const int i = t - 48;
            double sg = 0.0, gc = 0.0;
            if (row_active(c, k, i)) {
                const double t = c.T[NI * k + i], z = c.Z[NI * k + i];
                const double rv = (double)row_val(c, prm, k, i, u);
                sg = z / t;
                const float cm = cmu < 0.f ? row_target((float)z, -cmu) : cmu;   // (cmu < 0: per-row targets of the tail polish)
                gc = (double)cm / t + sg * (rv + t);  // complementarity target cm (0: affine-scaling predictor)
            }
            c.sig[i] = sg; c.gco[i] = gc;
            if (i < 32) {
                float a0, a1, a2;
                fric_row(c, prm, k, i, a0, a1, a2);
                c.arow[3 * i] = a0; c.arow[3 * i + 1] = a1; c.arow[3 * i + 2] = a2;
            }
        } else if (t >= 96 && t < 105) {
            // exact-Hessian block dt [lam_h]x (zero for the Gauss-Newton Hessian), row-major 3x3
            const int a = (t - 96) / 3, b = (t - 96) % 3;
            float sv = 0.f;
            if (use_exact && a != b) {
                const float lv = c.LAM[NS * (k + 1) + 6 + (3 - a - b)];
                sv = prm.dt * (((b - a + 3) % 3 == 1) ? -lv : lv);
            }
            c.arow[96 + t - 96] = sv;
        }
}

// ---- phase 2, float32 part: 285 triples of consecutive entries -- 135 of Quu (row i, block column bj < bi), 150 of
// Qus^T (row j, the xyz of one corner / one foot's offset) -- by one branch-free formula
//   out_c = sum_a w_a G[row_a][col0 + c] + ew E[c es]          (descriptor (row_a, w_a): column i of B or j of A)
// R rounds per thread (ids id0, id0 + 128, ...): addresses, then descriptor loads, then G loads, then arithmetic, then
// stores, so that the rounds overlap ----
template <int R>
__device__ inline void quu_qus_triples(const Ctx& c, const CmpcConsts& prm, int k, bool havep, int id0, int tpk)
{
    const float* Gp = c.G;
    int dsc[R], col0[R], sto[R], eo[R], es[R], symc[R];
    float ew[R], symw[R];
    bool ok[R];
#pragma unroll
    for (int rd = 0; rd < R; ++rd) {
        int id = id0 + 128 * rd;
        ok[rd] = id < 285;
        id = ok[rd] ? id : 0;
        const bool kind = id < 135;  // true: Quu
        const int idq = kind ? id : 0, idp = kind ? 0 : id - 135;
        const int ij = (tpk >> (16 * rd)) & 0xffff;   // tri[idq / 3], looked up once per backward pass (riccati_backward)
        const int i = 3 * ((ij >> 8) + 1) + idq % 3, j0 = 3 * (ij & 255);
        const int jr = idp / 10, i0 = 3 * (idp % 10);
        dsc[rd] = kind ? 3 * i : 3 * NU + 3 * jr;
        col0[rd] = kind ? j0 : i0;
        sto[rd] = kind ? (int)(c.QuuF - c.Pan) + i * RLD + j0 : jr * RLD + i0;
        // extra term: Quu rows of a force get the E^T P block; Qus^T entries of a force get -/+ gam Sx
        const int ct = i0 / 12;
        const int bb = jr < 3 ? jr : jr - 9 - 3 * ct;
        const bool bin = bb >= 0 && bb < 3;
        const float sgn = jr < 3 ? -1.f : (bin ? 1.f : 0.f);
        eo[rd] = kind ? (NS + (i < NF ? i : 0)) * GLD + j0 : (int)(c.arow - c.G) + 96 + (bin ? bb : 0);
        es[rd] = kind ? 1 : 3;
        ew[rd] = kind ? ((havep && i < NF) ? 1.f : 0.f) : (i0 < NF ? sgn * gam_of(c, ct, k) : 0.f);
        // another corner of the same foot, same axis: symmetry-cost coupling (Quu only)
        const bool sy = kind && i < NF && (i / 12) == (j0 / 12);
        const float gq = gam_of(c, i < 12 ? 0 : 1, k);
        symc[rd] = i % 3;
        symw[rd] = sy ? 2.f * prm.w_sym * 0.25f * gq * (2.f - gq) : 0.f;
    }
    int r0[R], r1[R], r2[R];
    float w0[R], w1[R], w2[R];
#pragma unroll
    for (int rd = 0; rd < R; ++rd) {
        r0[rd] = c.Brow[dsc[rd]]; r1[rd] = c.Brow[dsc[rd] + 1]; r2[rd] = c.Brow[dsc[rd] + 2];
        w0[rd] = c.Bval[dsc[rd]]; w1[rd] = c.Bval[dsc[rd] + 1]; w2[rd] = c.Bval[dsc[rd] + 2];
    }
    float o[R][3];
#pragma unroll
    for (int rd = 0; rd < R; ++rd) {
        const float* g0 = Gp + r0[rd] * GLD + col0[rd];
        const float* g1 = Gp + r1[rd] * GLD + col0[rd];
        const float* g2 = Gp + r2[rd] * GLD + col0[rd];
        const float* E = Gp + eo[rd];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            float v = w0[rd] * g0[cc] + w1[rd] * g1[cc] + w2[rd] * g2[cc] + ew[rd] * E[cc * es[rd]];
            if (cc == symc[rd]) v -= symw[rd];
            o[rd][cc] = v;
        }
    }
#pragma unroll
    for (int rd = 0; rd < R; ++rd)
        if (ok[rd]) {
            float* dst = c.Pan + sto[rd];   // QuuF sits NU * RLD floats below Pan
            dst[0] = o[rd][0]; dst[1] = o[rd][1]; dst[2] = o[rd][2];
        }
}

template <int NT, int NC, bool FG>
__device__ inline void stage_pre_body(const Ctx& c, const CmpcConsts& prm, int tid, int k, const float* Pcur, bool havep,
                                      bool use_exact, float reg, float cmu, int tpk)
{
    const bool pk = k > 0;
    const float* u = c.U + NU * k;
    PROF_DECL;
        // ---- phase 1: G = P [B;E] (39 x 30): thread <-> column, its three non-zeros in registers, rows strided over RG = 8
        // row groups ----
        {
            constexpr int RG = NT / 32, RR = (NXA + RG - 1) / RG;
            const int nrow = havep ? NXA : NS;
            if (tid < RG * NU) {
                const int i = tid % NU, r0 = tid / NU;
                const int b0 = c.Brow[3 * i], b1 = c.Brow[3 * i + 1], b2 = c.Brow[3 * i + 2];
                const float w0 = c.Bval[3 * i], w1 = c.Bval[3 * i + 1], w2 = c.Bval[3 * i + 2];
                const bool addp = havep && i < NF;
#pragma unroll
                for (int rr = 0; rr < RR; ++rr) {
                    const int r = r0 + RG * rr;
                    if (r < nrow) {
                        const float* Pr = Pcur + r * PLD;
                        float v = Pr[b0] * w0 + Pr[b1] * w1 + Pr[b2] * w2;
                        if (addp) v += Pr[NS + i];
                        c.G[r * GLD + i] = v;
                    }
                }
            }
        }
        __syncthreads();
        PROF(1);
        // ---- phase 2: what the factorisation needs.  Waves 0, 1: the float32 entries -- the 405 entries of Quu outside
        // its 3x3 diagonal blocks and the 450 of the panel rows Qus^T.  Wave 2: the diagonal blocks of Quu in float64
        // (cost, barrier and Levenberg terms).  Wave 3: qu in float64. ----
        PROF2_DECL;
        constexpr int T2 = 128;    // threads on the float32 triples
        if (tid < T2) {
            quu_qus_triples<2>(c, prm, k, havep, tid, tpk);   // triples 0..255; the next wave takes 256..284 after its float64 blocks
            PROF2(5);
        } else if (tid < T2 + 64) {
          // ---- the ten 3x3 diagonal blocks of Quu in float64 (60 lower entries).  Branch-free on clamped indices so that all
          // loads of a dependency level are in flight together: level 0 the descriptor of column i, the barrier coefficients
          // and friction rows of the corner, the E^T P entry; level 1 the three G entries the descriptor points at ----
          const int t = tid - T2;
          const int tc = t < 60 ? t : 59;
          // lower entry w of diagonal block b: (row, col) = (0,0) (1,0) (1,1) (2,0) (2,1) (2,2)
          const int b = tc / 6, w = tc - 6 * b;
          const int rr = w >= 3 ? 2 : (w >= 1 ? 1 : 0), cc = w - rr * (rr + 1) / 2;
          const int i = 3 * b + rr, j = 3 * b + cc;
          const bool isF = i < NF, dg = i == j;
          const int b0 = c.Brow[3 * i], b1 = c.Brow[3 * i + 1], b2 = c.Brow[3 * i + 2];
          const float w0 = c.Bval[3 * i], w1 = c.Bval[3 * i + 1], w2 = c.Bval[3 * i + 2];
          const float ge = c.G[(NS + (isF ? i : 0)) * GLD + j];
          const int r0 = isF ? 4 * b : 0;             // friction rows of the corner
          const int iq = isF ? 0 : i - 24;            // landing-offset component
          double sg[4], ar[4], ac[4];
#pragma unroll
          for (int f = 0; f < 4; ++f) { sg[f] = c.sig[r0 + f]; ar[f] = (double)c.arow[3 * (r0 + f) + rr]; ac[f] = (double)c.arow[3 * (r0 + f) + cc]; }
          const double slo = c.sig[32 + iq], shi = c.sig[38 + iq];
          const bool fr = qfree(c, k, iq);
          const double gam = gam_of(c, isF ? i / 12 : 0, k);
          const float g0 = c.G[b0 * GLD + j], g1 = c.G[b1 * GLD + j], g2 = c.G[b2 * GLD + j];
          float vf = w0 * g0 + w1 * g1 + w2 * g2;
          if (havep && isF) vf += ge;
          double v = (double)vf;
          double vF = v;
          if (dg) {
              vF += 2.0 * prm.w_sym * (1.0 - 0.25 * gam * (2.0 - gam));
              if (pk) vF += (double)prm.D[i % 3];
              vF += (double)reg;
          }
#pragma unroll
          for (int f = 0; f < 4; ++f) vF += sg[f] * ar[f] * ac[f];
          const double vQ = dg ? (fr ? v + slo + shi + (double)reg : 1.0) : v;   // fixed q: exact identity row
          if (t < 60) {
            c.QuuD[9 * b + 3 * rr + cc] = isF ? vF : vQ;
            c.QuuF[i * RLD + j] = 0.f;  // the float copy of a diagonal block collects the updates by earlier blocks
            PROF4(7);
          }
          if (t < 285 - 256) quu_qus_triples<1>(c, prm, k, havep, 256 + t, 0);
        } else if (tid < T2 + 128) {
          // ---- Pd = P [d; 0] + pv (float64), then qu, which reads it (same wave: LDS order suffices).  Every lane of the wave
          // computes both on clamped indices, branch-free: two levels of loads instead of a chain of small dependent ones ----
          constexpr int T3 = T2 + 64;
          const int l = tid - T3;
          const int r = l < NXA ? l : NXA - 1, iq = l < NU ? l : NU - 1;
          float pc[NS], dc[NS];
#pragma unroll
          for (int a = 0; a < NS; ++a) { pc[a] = Pcur[a * PLD + r]; dc[a] = c.d[NS * k + a]; }   // (P is stored with both triangles:
          // row a, column r is the same number as row r, column a, and consecutive lanes read consecutive banks)
          const double pvr = c.pv[r];
          const bool isF = iq < NF;
          const int m = isF ? iq : 0, ct = m / 12, ax = m % 3;   // force component (clamped)
          const int q = isF ? 0 : iq - 24;                         // landing-offset component (clamped)
          const float* uf = u + 12 * ct + ax;
          const float u0 = uf[0], u1 = uf[3], u2 = uf[6], u3 = uf[9], um = u[m];
          const float up = c.U[NU * (pk ? k - 1 : 0) + m];
          const double gam = gam_of(c, ct, k);
          const int r0 = 4 * (m / 3);
          double gc[4], ar[4];
#pragma unroll
          for (int f = 0; f < 4; ++f) { gc[f] = c.gco[r0 + f]; ar[f] = (double)c.arow[3 * (r0 + f) + ax]; }
          const double glo = c.gco[32 + q], ghi = c.gco[38 + q];
          const bool fr = qfree(c, k, q);
          const int b0 = c.Brow[3 * iq], b1 = c.Brow[3 * iq + 1], b2 = c.Brow[3 * iq + 2];
          const double w0 = c.Bval[3 * iq], w1 = c.Bval[3 * iq + 1], w2 = c.Bval[3 * iq + 2];
          {
            double acc = 0.0;
#pragma unroll
            for (int a = 0; a < NS; ++a) acc += (double)pc[a] * (double)dc[a];
            if (l < NXA) c.Pd[r] = (r < NS || havep) ? pvr + acc : pvr;   // (rows >= NS of the terminal P do not exist: discarded)
          }
          // gradient of the symmetry cost (grad_sym), barrier terms, force-rate term
          const double mean = 0.25 * ((double)u0 + (double)u1 + (double)u2 + (double)u3);
          const double esum = 4.0 * mean * (1.0 - gam);
          double gF = 2.0 * prm.w_sym * (((double)um - gam * mean) - 0.25 * gam * esum);
#pragma unroll
          for (int f = 0; f < 4; ++f) gF += gc[f] * ar[f];
          if (pk) gF += (double)prm.D[ax] * ((double)um - (double)up);
          const double gQ = fr ? glo - ghi : 0.0;
          wave_lds_sync();
          // B^T Pd through the column descriptor of B (the same float coefficients the Hessian blocks are built from)
          const double pe = c.Pd[NS + m], p0 = c.Pd[b0], p1 = c.Pd[b1], p2 = c.Pd[b2];
          double g = isF ? (havep ? gF + pe : gF) : gQ;
          g += w0 * p0 + w1 * p1 + w2 * p2;
          if (l < NU) {
            // qu (float64 -> float: it vanishes at convergence, so float keeps its relative accuracy)
            c.Pan[(NPAN - 1) * RLD + iq] = (float)g;
            PROF3(8);
          }
        }
        __syncthreads();
        PROF(2);
}

template <int NT>
__device__ inline void stage_qss_body(const Ctx& c, const CmpcConsts& prm, int tid, int k, const float* Pcur, float* Qb, int tqp)
{
            const int t = tid - 128;
            if (t < 120) {
                const int ij = tqp & 0xffff;   // tri[t]
                const int i = ij >> 8, j = ij & 255;
                const float ai0 = c.Aval[3 * i], ai1 = c.Aval[3 * i + 1], ai2 = c.Aval[3 * i + 2];
                const float aj0 = c.Aval[3 * j], aj1 = c.Aval[3 * j + 1], aj2 = c.Aval[3 * j + 2];
                const float* P0r = Pcur + c.Arow[3 * i] * PLD;
                const float* P1r = Pcur + c.Arow[3 * i + 1] * PLD;
                const float* P2r = Pcur + c.Arow[3 * i + 2] * PLD;
                const int c0 = c.Arow[3 * j], c1 = c.Arow[3 * j + 1], c2 = c.Arow[3 * j + 2];
                float v = (i == j) ? qdiag(prm, k, i) : 0.f;
                v += ai0 * (aj0 * P0r[c0] + aj1 * P0r[c1] + aj2 * P0r[c2]) + ai1 * (aj0 * P1r[c0] + aj1 * P1r[c1] + aj2 * P1r[c2])
                     + ai2 * (aj0 * P2r[c0] + aj1 * P2r[c1] + aj2 * P2r[c2]);
                Qb[i * 16 + j] = v;
            }
            if (t < NS) c.qs[t] = grad_track(c, prm, k, t) + At_vec<double>(c, prm, k, t, c.Pd);
}

template <int NT, bool PACKED>
__device__ inline void stage_post_body(const Ctx& c, const CmpcConsts& prm, int tid, int k, float* Pnew, const float* Qb, int tqp)
{
    const bool pk = k > 0;
    const float* u = c.U + NU * k;
    PROF_DECL;
        // ---- phase 4: P = [Qss 0; 0 D] - W^T W  (rows of the panel are W^T rows; p rows pre-scaled by -D) ----
        {
            // 2x2 output tiles: thread <-> tile (bi, bj), bj <= bi, of the 39x39 (or 15x15) lower triangle.
            const int nb = pk ? 20 : 8;
            const int tile = tid;
            if (tile < nb * (nb + 1) / 2) {
                const int t = (tqp >> 16) & 0xffff;   // tri[tile]
                const int bi = t >> 8, bj = t & 255;
                const int i0 = 2 * bi, j0 = 2 * bj, ncol = pk ? NXA : NS;
                const float4* ri0 = reinterpret_cast<const float4*>(c.Pan + i0 * RLD);
                const float4* ri1 = reinterpret_cast<const float4*>(c.Pan + (i0 + 1) * RLD);
                const float4* rj0 = reinterpret_cast<const float4*>(c.Pan + j0 * RLD);
                const float4* rj1 = reinterpret_cast<const float4*>(c.Pan + (j0 + 1) * RLD);
                // what the products are subtracted from (Qss or D), fetched before the dot products: off their dependency chain
                auto base_of = [&](int i, int j) {
                    const float qb = Qb[(i < NS ? i : 0) * 16 + (j < NS ? j : 0)];
                    return i < NS ? qb : (i == j ? prm.D[(i - NS) % 3] : 0.f);
                };
                const float b00 = base_of(i0, j0), b01 = base_of(i0, j0 + 1), b10 = base_of(i0 + 1, j0), b11 = base_of(i0 + 1, j0 + 1);
                float a00, a01, a10, a11;
                if constexpr (PACKED) {
                    // Packed multiply-add chains over the pairs (x, y), (z, w) of the 16-byte operands, the two halves added once at the end: 2 instructions per four
                    // products and accumulator, where the sum written as (xx + yy) + (zz + ww) added to the accumulator costs 4 (config 3 456.4 k -> 464.5 k solves/s).
                    // Horizons up to 20 only: a 16-term chain rounds worse than eight 4-term trees, and at N = 30 (tolerance 3e-7) the value function's rounding
                    // shows in the tail -- soak of 40 960 config-5 problems: 57 instead of 34 need 14 iterations or more, and one straggler of the bench batch (29
                    // iterations against 18) cost that batch 5 % (profiles/r04_experiments_not_kept.txt, item 28).
                    v2f p00 = {0.f, 0.f}, p01 = {0.f, 0.f}, p10 = {0.f, 0.f}, p11 = {0.f, 0.f};
#pragma unroll 1
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int q4 = 4 * h + q;
                            const float4 x0 = ri0[q4], x1 = ri1[q4], y0 = rj0[q4], y1 = rj1[q4];
                            const v2f x0a = {x0.x, x0.y}, x0b = {x0.z, x0.w}, x1a = {x1.x, x1.y}, x1b = {x1.z, x1.w};
                            const v2f y0a = {y0.x, y0.y}, y0b = {y0.z, y0.w}, y1a = {y1.x, y1.y}, y1b = {y1.z, y1.w};
                            p00 = __builtin_elementwise_fma(x0b, y0b, __builtin_elementwise_fma(x0a, y0a, p00));
                            p01 = __builtin_elementwise_fma(x0b, y1b, __builtin_elementwise_fma(x0a, y1a, p01));
                            p10 = __builtin_elementwise_fma(x1b, y0b, __builtin_elementwise_fma(x1a, y0a, p10));
                            p11 = __builtin_elementwise_fma(x1b, y1b, __builtin_elementwise_fma(x1a, y1a, p11));
                        }
                    }
                    a00 = p00[0] + p00[1]; a01 = p01[0] + p01[1]; a10 = p10[0] + p10[1]; a11 = p11[0] + p11[1];
                } else {
                    a00 = a01 = a10 = a11 = 0.f;
                    // The 168-register variants (three workgroups per CU) take the 32 row loads in two rounds: hoisted all
                    // at once they need the callee-saved registers, and saving those at every call of this function was
                    // 18 GB of scratch traffic per 4096-problem batch (columns 30, 31 of the panel are stored as zeros).
#pragma unroll 1
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int q4 = 4 * h + q;
                            const float4 x0 = ri0[q4], x1 = ri1[q4], y0 = rj0[q4], y1 = rj1[q4];
                            a00 += (x0.x * y0.x + x0.y * y0.y) + (x0.z * y0.z + x0.w * y0.w);
                            a01 += (x0.x * y1.x + x0.y * y1.y) + (x0.z * y1.z + x0.w * y1.w);
                            a10 += (x1.x * y0.x + x1.y * y0.y) + (x1.z * y0.z + x1.w * y0.w);
                            a11 += (x1.x * y1.x + x1.y * y1.y) + (x1.z * y1.z + x1.w * y1.w);
                        }
                    }
                }
                auto put = [&](int i, int j, float base, float acc) {
                    if (j > i || i >= ncol) return;
                    const float r = base - acc;
                    Pnew[i * PLD + j] = r;
                    Pnew[j * PLD + i] = r;
                };
                put(i0, j0, b00, a00); put(i0, j0 + 1, b01, a01);
                put(i0 + 1, j0, b10, a10); put(i0 + 1, j0 + 1, b11, a11);
            }
        }
        {
            PROF2_DECL;
            // gradient of the value function (float64)
            const int ncol = pk ? NXA : NS;
            constexpr int TG = 216;    // threads beyond the 210 tile owners
            if (tid >= TG && tid < TG + NXA) {
                const int i = tid - TG;
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
                c.pv[i] = v;   // (the value gradient of stage k+1 was last read in phase 2: written in place, one barrier less)
                PROF5(19);
            }
        }
        __syncthreads();
        PROF(4);
}

// ---- out-of-line phases: each rebuilds the LDS map from the LDS base it is handed (an address-space-3
// pointer, so the callee needs no dynamic-LDS table lookup and still addresses LDS with ds_ instructions) ----
typedef __attribute__((address_space(3))) char* lds_t;
#define CMPC_PHASE_PROLOGUE                                                            \
    char* smem = (char*)lds;                                                           \
    const int N = NC > 0 ? NC : __builtin_amdgcn_readfirstlane(Nrt);                   \
    Ctx c;                                                                             \
    make_ctx<FG>(c, smem, N, fg_base);                                                 \
    const CmpcConsts& prm = *reinterpret_cast<const CmpcConsts*>(smem);                \
    const int tid = threadIdx.x

template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) void stage_pre(lds_t lds, int Nrt, float* fg_base, int k_in, bool havep_in, bool exact_in, float reg, float cmu, int tpk)
{
    CMPC_PHASE_PROLOGUE;
    const int k = __builtin_amdgcn_readfirstlane(k_in);   // (arguments arrive in VGPRs: a uniform copy keeps the stage's address arithmetic on the SALU)
    const bool havep = __builtin_amdgcn_readfirstlane((int)havep_in) != 0, use_exact = __builtin_amdgcn_readfirstlane((int)exact_in) != 0;
    use_desc_set(c, k & 1);
    stage_pre_body<NT, NC, FG>(c, prm, tid, k, c.P0, havep, use_exact, reg, cmu, tpk);
}
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) void stage_desc(lds_t lds, int Nrt, float* fg_base, int k_in, bool exact_in, float cmu)
{
    CMPC_PHASE_PROLOGUE;
    const int k = __builtin_amdgcn_readfirstlane(k_in);   // (arguments arrive in VGPRs: a uniform copy keeps the stage's address arithmetic on the SALU)
    const bool use_exact = __builtin_amdgcn_readfirstlane((int)exact_in) != 0;
    use_desc_set(c, k & 1);
    stage_desc_body(c, prm, desc_pack_t(tid - 128), k, use_exact, cmu);
}


// A backward stage is two calls.  stage_mid: phase 3 of stage k -- the factorisation on wave 0 (waves 0-1 in the resident
// variants), Qss and the descriptors of stage k-1 on the others -- and the barrier behind it.  stage_post_pre: phase 4 of stage k,
// then phases 1-2 of stage k-1.  (Four calls per stage cost 2 % in prologues and stage addresses; all of it behind ONE call is
// within +-0.5 % of two.  The one-call variant is how the buffer-store problem described at RecRef<true> was found.)
template <int NT, int NC, bool FG, bool FIX = false>
__device__ __attribute__((noinline)) void stage_mid(lds_t lds, int Nrt, float* fg_base, int k_in, bool exact_in, float cmu, int tqp)
{
    CMPC_PHASE_PROLOGUE;
    const int k = __builtin_amdgcn_readfirstlane(k_in);
    const bool use_exact = __builtin_amdgcn_readfirstlane((int)exact_in) != 0;
    PROF_DECL;
    if (tid < 64) {
        const int fixedmask = (~c.qmask[k]) & 63;
        stage_factor<FG, false, FIX ? 1 : 0>(c.QuuF, c.QuuD, c.Pan, RecRef<FG>(c.Lf, N, k), c.idstrip, prm.D, c.flag, tid, fixedmask);
    } else if (tid >= 128) {
        use_desc_set(c, k & 1);
        stage_qss_body<NT>(c, prm, tid, k, c.P0, c.Qb, tqp);
        if (k > 0) {
            use_desc_set(c, ((k - 1) & 1) - (k & 1));
            stage_desc_body(c, prm, desc_pack_t(tid - 128), k - 1, use_exact, cmu);
        }
    }
    __syncthreads();
    PROF(3);
}
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) void stage_post_pre(lds_t lds, int Nrt, float* fg_base, int k_in, bool exact_in, float reg, float cmu, int tpk, int tqp)
{
    CMPC_PHASE_PROLOGUE;
    const int k = __builtin_amdgcn_readfirstlane(k_in);
    const bool use_exact = __builtin_amdgcn_readfirstlane((int)exact_in) != 0;
    stage_post_body<NT, (NC > 0 && NC <= 20)>(c, prm, tid, k, c.P0, c.Qb, tqp);
    if (k > 0) {
        use_desc_set(c, (k - 1) & 1);
        stage_pre_body<NT, NC, FG>(c, prm, tid, k - 1, c.P0, true, use_exact, reg, cmu, tpk);
    }
}

// =====================================================================================================================
// The square-root backward stage (resident eight-wave variants, CMPC_SQ; see the note at CMPC_SQRT_BACKWARD).
// Index space of the assembled stage matrix M: [u (30) | s (15) | g (1)].  Stage j is assembled in set j & 1:
//   uu  lower triangle, float, in QuuF (the slots inside the 3x3 diagonal blocks hold only the -Z^T Z part; their cost / barrier part is float64 in QuuD)
//   su  Pan rows 0..14 (row s, 30 columns),  gu  Pan row NPAN-1 (qu),  ss  Qb (both triangles),  gs  qs (float64)
// =====================================================================================================================
// With T = [B~ A~] (39 x 45, <= 4 non-zeros per column), W = L^{-1}[Qus | -D] of stage k+1 and Z = W T, stage k's matrix is
//   (cost, barrier, Levenberg terms + T^T [Qss 0; 0 D] T)  -  Z^T Z:
// the bracket needs nothing of stage k+1's factorisation and is assembled while that runs, through Y = Qss [B A] (15 x 45): a sandwich entry T_i^T Qss T_col is then
// three loads, sum_a w_a(i) Y[r_a(i)][col], instead of nine.

// ---- The streaming square-root stage: ONE barrier per stage, one call per backward pass and role.  The factorisation runs on wave 0 (the one-wave scheme: 76 rows
// in 64 lanes, trailing updates on the matrix pipe) and publishes the three finished columns of W^T after every pivot block (chol_block<.., PUB>).  Waves 1, 2, 3, 5, 6, 7
// first assemble the Z-independent part of the next stage -- Y = Qss [B A] on waves 5-7; the float64 right-hand sides q_u (wave 2) and q_s (wave 3), which need no Y
// and start at once; the float64 diagonal blocks (wave 1) and 360 float32 triples (all six) behind Y -- then follow the factorisation two blocks at a time: each owns
// one 16 x 16 tile of Z^T Z and, per block, forms the two operand entries its lanes feed to the matrix core -- sparse combinations of the published rows -- and issues
// one v_mfma_f32_16x16x4_f32 (K = the block's three columns and a zero).  When the last block is out, what remains is one tile update and the read-modify-write of the
// tile's entries.  Wave 4 shares its SIMD with the factorising wave: it builds the descriptors of the stage after next and then carries the gradient column in three
// batches of pivot blocks (sq_gradient_role) -- few instructions, mostly waiting, and its last batch runs while wave 0 stores its record rows.
// Synchronisation inside the stage is by LDS words: the progress of wave 0 (monotonic over the backward pass: 16 ord + block + 1; one copy per lane, so that the store
// needs no lane predicate), a count of the waves done with Y (3 per stage), and a count of those whose part of the assembly is complete (6 per stage).  Every wait is
// bounded: a wave that gives up raises the failure flag AND the give-up word c.flag[1], which the driver reports in its own digit of info[3] (include/cmpc.h) -- a
// stalled hand-off must never read as a bad pivot.
// BARRIER INVARIANT: sq_factor_loop and sq_consume_loop each execute exactly 1 + (N - k0) workgroup barriers per backward pass (sq_pass_barriers), one per stage plus
// the one behind the terminal assembly, with no early exit on either side; tests/test_host_logic.py holds the two counts together. ----
#define SQ_Y_WAVES 3           // consumer waves that write Y (waves 5, 6, 7): the Y count advances by this per stage
// the stage loop of a backward pass: BOTH roles iterate with this header and hold exactly one barrier per trip (plus SQ_PRE_BARRIERS before the loop)
#define SQ_STAGE_LOOP(k, N, k0) for (int k = (N) - 1; k >= (k0); --k)
#define SQ_PRE_BARRIERS 1
__host__ __device__ constexpr int sq_pass_barriers(int N, int k0) { return SQ_PRE_BARRIERS + (N - k0); }
// Hand-off words: release on the publishing side, acquire on the reading side, workgroup scope (relaxed accesses leaned on in-order LDS issue of one wave,
// which is how the hardware behaves and not what the memory model promises; on LDS-only traffic the stronger orders cost an s_waitcnt lgkmcnt(0) the code had anyway).
__device__ inline int lds_peek(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void lds_count(int* p) { __hip_atomic_fetch_add(p, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// bounded wait until *p >= want; false: gave up (the caller raises the flags)
__device__ inline bool lds_wait_ge(const int* p, int want)
{
    int spins = 0;
    // (the word is one address for the whole wave: made provably uniform, so that the loop is scalar control flow and not an exec-mask waterfall)
    while (__builtin_amdgcn_readfirstlane(lds_peek(p)) < want) {
        if (++spins > SQ_SPIN_MAX) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}
__device__ inline void sq_give_up(const Ctx& c) { c.flag[0] = 1; c.flag[1] = 1; }
// The factorising wave's side of a backward pass: ONE call for all stages (the stage loop and its barriers inside: a call per stage cost the critical wave the
// rebuilding of the LDS map every stage).
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) void sq_factor_loop(lds_t lds, int Nrt, float* fg_base, int k0_in)
{
    CMPC_PHASE_PROLOGUE;
    const int k0 = __builtin_amdgcn_readfirstlane(k0_in);
    __syncthreads();                                   // (the consumers assemble stage N-1 from the terminal cost meanwhile)
    int ord = 0;                                       // ordinal of the stage within the backward pass
    __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
    SQ_STAGE_LOOP(k, N, k0) {                          // (sq_pass_barriers: one barrier per trip, no early exit)
        ++ord;
        lds_t lk = lds;                                // (an opaque copy of the LDS base per stage: see sq_consume_loop)
        asm volatile("" : "+v"(lk));
        char* smk = (char*)lk;
        Ctx c;
        make_ctx<FG>(c, smk, N, fg_base);
        const CmpcConsts& prm = *reinterpret_cast<const CmpcConsts*>(smk);
        const int s = k & 1;
        const int fixedmask = (~c.qmask[k]) & 63;
        SQPROF_DECL;
        // (one wave: its second SIMD's worth of issue slots goes to the consumers -- the eight-wave shape is bound by what waves 1-7 can issue under the factorisation)
        stage_factor<FG, true>(c.QuuF + s * MSET, s ? c.QuuD1 : c.QuuD, c.Pan + s * MSET, RecRef<FG>(c.Lf, N, k), c.idstrip, prm.D, c.flag, tid,
                                     fixedmask, c.ZT, c.prog + tid, 16 * ord);
        SQPROF(9);         // (wave 0: the factorisation alone)
        __syncthreads();
        SQPROF(3);
    }
    __builtin_amdgcn_s_setprio(0);
}
// =====================================================================================================================
// The consumers of the streaming stage, lean form (round 4).  One wave executes at most one instruction of ANY kind per four cycles, so what a
// consumer wave costs is its instruction count, scalar bookkeeping and exec-mask juggling included: the first streaming version spent ~1 800
// instructions per wave and stage, two thirds of them index decoding, clamping and divergent control flow that does not depend on the stage.
// Here everything that is fixed over a backward pass -- which entries a lane owns, the rows its sparse columns point at, where results go -- is
// decoded ONCE per pass into a per-lane plan of indices and 0/1 float masks (ConsPlan), the stage body is straight-line code on clamped indices
// (results of lanes without a task go to words nobody reads), and a set is ONE stride (QuuF | Pan | Qb: that is why Qss lives inside the set) -- first with the
// set as a template parameter, every set offset an immediate of the LDS instruction; since the consumers wait for the factorising wave anyway, now as a runtime value:
// one copy of the stage body, the set offsets in scalar registers (see sq_consume_loop).
// =====================================================================================================================
struct ConsPlan {
    // the thread's float32 triple: out_c = w . Y[:, col + c] + ew E[3 c] + m_c (qd - symw)
    int t_dst, t_w, t_y0, t_y1, t_y2, t_e;
    float t_symk, t_fsel, t_esgn, t_efsel, t_qdc, t_zsel, t_m0, t_m1, t_m2;
    // the lane's two tile operands: published rows (float index kq + 4 r into the publication buffer) and where their weights are
    int pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3, wa, wb;
    float wa3, wb3;
    // where the four results of the lane go (float index from QuuF of the set); their mirror images (Qss is kept with both triangles: tiles (2,1), (2,2) and the
    // second tile of the pair) live in ri[9..12] of the waves that own such a tile -- role slots those waves do not use
    int d0, d1, d2, d3;
    // role of the wave: waves 1, 2, 3 the float64 parts (diagonal blocks, q_u, q_s), waves 5, 6, 7 Y; wave 7 carries the gradient column and no tile
    int ri[14];
    float rf[6];
};
#define SQ_TRASH (32)   // float index, from QuuF of a set, of four words nobody reads (columns 32..35 of row 0: the rows are read as 32 floats)
// rows of column `col` of [B | A] (fixed sparsity pattern; read from a descriptor set that has been built)
__device__ inline void plan_rows(const int* rows, int col, int& r0, int& r1, int& r2) { r0 = rows[3 * col]; r1 = rows[3 * col + 1]; r2 = rows[3 * col + 2]; }
__device__ inline void cons_plan_build(ConsPlan& pl, const Ctx& c, const CmpcConsts& prm, int w7, int ln, const int* rows)
{
    const int wv = w7 < 4 ? w7 - 1 : w7 - 2;       // 0..5 (wave 4 has no plan: -1 here)
    const int m4 = ln & 15, kq = ln >> 4;
    // ---- triple ----
    {
        const int id = (wv >= 0 && w7 != 4) ? 64 * wv + ln : 360;
        const bool val = id < 360;
        const int idc = val ? id : 0;
        const bool isuu = idc < 135, isss = idc >= 285;
        const int p = isuu ? idc / 3 : 0;
        const int bi = 1 + (p >= 1) + (p >= 3) + (p >= 6) + (p >= 10) + (p >= 15) + (p >= 21) + (p >= 28) + (p >= 36);
        const int bj = p - bi * (bi - 1) / 2;
        const int idp = isuu ? 0 : (isss ? idc - 285 : idc - 135);
        const int pr = isss ? idp / 5 : idp / 10, pc = isss ? idp % 5 : idp % 10;
        const int i = isuu ? 3 * bi + idc % 3 : pr;
        const int col0 = isuu ? 3 * bj : 3 * pc;
        const int drow = isuu ? i : NU + pr;
        const int dcol = isss ? NU + col0 : col0;
        int r0, r1, r2;
        plan_rows(rows, drow, r0, r1, r2);
        pl.t_w = 3 * drow;
        pl.t_y0 = 16 * dcol + r0; pl.t_y1 = 16 * dcol + r1; pl.t_y2 = 16 * dcol + r2;
        pl.t_dst = !val ? SQ_TRASH : (isuu ? i * RLD + col0 : (isss ? NU * RLD + NPAN * RLD + 16 * pr + col0 : NU * RLD + pr * RLD + col0));
        const bool sy = isuu && i < NF && (i / 12) == (col0 / 12);
        pl.t_symk = sy ? 0.5f * prm.w_sym : 0.f;
        pl.t_fsel = (isuu && i >= 12) ? 1.f : 0.f;
        const int ct = col0 / 12;
        const int bb = pr < 3 ? pr : pr - 9 - 3 * ct;
        const bool bin = bb >= 0 && bb < 3;
        const bool su = !isuu && !isss;
        pl.t_esgn = (su && col0 < NF) ? (pr < 3 ? -1.f : (bin ? 1.f : 0.f)) : 0.f;
        pl.t_efsel = (ct == 1) ? 1.f : 0.f;
        pl.t_e = 96 + (bin ? bb : 0);
        const int symc = isuu ? i % 3 : -1, qc = isss ? pr - col0 : -1;
        const int mc = isuu ? symc : qc;             // component that carries the extra term (-symw for Quu, +qd for Qss)
        pl.t_m0 = mc == 0 ? 1.f : 0.f; pl.t_m1 = mc == 1 ? 1.f : 0.f; pl.t_m2 = mc == 2 ? 1.f : 0.f;
        pl.t_qdc = (isss && pr != 2) ? qdiag(prm, 0, pr) : 0.f;
        pl.t_zsel = (isss && pr == 2) ? 1.f : 0.f;
    }
    // ---- tile ----
    int d[4], e[4];
    {
        // tiles of the lower triangle of the 3 x 3 tile grid on consumer waves 0..4: (0,0) | (2,0) | (2,1) | (1,0) and (1,1) | (2,2); wave 5 (the gradient column) has none.
        // The pair shares its operand rows 16..31 -- the second tile costs one more matrix-pipe instruction per block -- and sits on a Y wave: those are done with
        // the assembly first, and their plan has room for the second tile's destinations (ri[5..12]).
        const int I = wv == 0 ? 0 : (wv == 3 ? 1 : 2), J = wv == 2 ? 1 : (wv == 4 ? 2 : 0);
        auto operand = [&](int row, int& p0, int& p1, int& p2, int& p3, int& w, float& w3) {
            const bool ok = row < NU + NS;
            int r0 = 0, r1 = 0, r2 = 0;
            if (ok) plan_rows(rows, row, r0, r1, r2);
            p0 = kq + 4 * r0; p1 = kq + 4 * r1; p2 = kq + 4 * r2; p3 = kq + 4 * (row < NF ? NS + row : 0);
            w = ok ? 3 * row : 3 * NU + 3 * NS + 3 + 105;      // (arow[105..107]: never written, zero since the kernel started)
            w3 = row < NF ? -prm.D[row % 3] : 0.f;             // W_p = -L^{-1}[:, :24] D: the identity rows are published unscaled
        };
        operand(16 * I + m4, pl.pa0, pl.pa1, pl.pa2, pl.pa3, pl.wa, pl.wa3);
        operand(16 * J + m4, pl.pb0, pl.pb1, pl.pb2, pl.pb3, pl.wb, pl.wb3);
        auto results = [&](int TI, int TJ, bool owned) {
            const int jj = 16 * TJ + m4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ii = 16 * TI + 4 * kq + i;
                const bool ok = owned && jj <= ii && ii < NU + NS;
                int p = ii * RLD + jj, p2 = p;
                if (ii >= NU) {
                    if (jj < NU) { p = NU * RLD + (ii - NU) * RLD + jj; p2 = p; }
                    else { p = NU * RLD + NPAN * RLD + 16 * (ii - NU) + (jj - NU); p2 = NU * RLD + NPAN * RLD + 16 * (jj - NU) + (ii - NU); }
                }
                d[i] = ok ? p : SQ_TRASH + i; e[i] = ok ? p2 : SQ_TRASH + i;
            }
        };
        results(I, J, wv >= 0 && wv < 5);
        pl.d0 = d[0]; pl.d1 = d[1]; pl.d2 = d[2]; pl.d3 = d[3];
        if (wv == 3) results(1, 1, true);  // (d, e kept until the role part below: waves 2 and 4 store e, wave 3 the second tile's d and e)
    }
    // ---- role ----
#pragma unroll
    for (int q = 0; q < 14; ++q) pl.ri[q] = 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) pl.rf[q] = 0.f;
    if (wv == 0) {
        // the ten 3x3 diagonal blocks of Quu in float64: lane <-> lower entry w of block b: (row, col) = (0,0) (1,0) (1,1) (2,0) (2,1) (2,2)
        const int tc = ln < 60 ? ln : 59;
        const int b = tc / 6, w = tc - 6 * b;
        const int rr = w >= 3 ? 2 : (w >= 1 ? 1 : 0), cc = w - rr * (rr + 1) / 2;
        const int i = 3 * b + rr, j = 3 * b + cc;
        const bool isF = i < NF;
        int r0, r1, r2;
        plan_rows(rows, i, r0, r1, r2);
        const int f0 = isF ? 4 * b : 0;
        pl.ri[0] = 3 * i; pl.ri[1] = 16 * j + r0; pl.ri[2] = 16 * j + r1; pl.ri[3] = 16 * j + r2;
        pl.ri[4] = f0; pl.ri[5] = 3 * f0 + rr; pl.ri[6] = 3 * f0 + cc; pl.ri[7] = isF ? 0 : i - 24;
        pl.ri[8] = 9 * b + 3 * rr + cc;                           // (lanes 60..63 repeat lane 59: the same value to the same word)
        pl.ri[9] = i * RLD + j;
        pl.rf[0] = isF ? 1.f : 0.f; pl.rf[1] = rr == cc ? 1.f : 0.f; pl.rf[2] = (isF && i >= 12) ? 1.f : 0.f; pl.rf[3] = prm.D[i % 3];
    } else if (wv == 1) {
        // q_u: lane l <-> row l of hb (39) and entry l of q_u (30)
        const int r = ln < NXA ? ln : NXA - 1, iq = ln < NU ? ln : NU - 1;
        const bool isF = iq < NF;
        const int m = isF ? iq : 0, ct = m / 12, ax = m % 3, q = isF ? 0 : iq - 24;
        const int mr = r >= NS ? r - NS : 0;
        int r0, r1, r2;
        plan_rows(rows, iq, r0, r1, r2);
        pl.ri[0] = r < NS ? r : NS - 1; pl.ri[1] = mr; pl.ri[2] = 12 * ct + ax; pl.ri[3] = m; pl.ri[4] = 4 * (m / 3); pl.ri[5] = 12 * (m / 3) + ax;
        pl.ri[6] = q; pl.ri[7] = 3 * iq; pl.ri[8] = r0; pl.ri[9] = r1; pl.ri[10] = r2; pl.ri[11] = iq;
        pl.rf[0] = isF ? 1.f : 0.f; pl.rf[1] = ct ? 1.f : 0.f; pl.rf[2] = prm.D[mr % 3]; pl.rf[3] = prm.D[ax]; pl.rf[4] = r < NS ? 1.f : 0.f;
    } else if (wv == 2) {
        // q_s: lane l <-> state component l
        const int j = ln < NS ? ln : NS - 1;
        const int ja = j < 3 ? j : (j >= 9 ? (j - 9) % 3 : 0), ja1 = (ja + 1) % 3, ja2 = (ja + 2) % 3;
        const int jct = j >= 12 ? 1 : 0;
        const int gfo = j < 3 ? 30 : 24 + 3 * jct;
        pl.ri[0] = j;
        pl.ri[1] = j < 3 ? c.L.pComref() + j : (j < 6 ? c.L.pComref() : (j < 9 ? c.L.pHref() + j - 6 : c.L.pNom(jct) + ja));
        pl.ri[2] = gfo + ja1; pl.ri[3] = gfo + ja2; pl.ri[4] = (j >= 3 && j < 6) ? j - 3 : 0; pl.ri[5] = 6 + ja1; pl.ri[6] = 6 + ja2;
        pl.rf[0] = j != 2 ? qdiag(prm, 0, j) : 0.f; pl.rf[1] = j == 2 ? 1.f : 0.f; pl.rf[2] = jct ? 1.f : 0.f;
        pl.rf[3] = j < 3 ? 1.f : 0.f; pl.rf[4] = (j >= 3 && j < 6) ? 1.f : 0.f; pl.rf[5] = j >= 9 ? 1.f : 0.f;
        pl.ri[9] = e[0]; pl.ri[10] = e[1]; pl.ri[11] = e[2]; pl.ri[12] = e[3];         // mirror images of tile (2, 1)
    } else if (wv >= 3) {
        // Y^T = (Qss [B A])^T: one float4 per lane (180 of the 192)
        const int t = 64 * (wv - 3) + ln, col = t >> 2, q = t & 3;
        const int cl = col < NU + NS ? col : NU + NS - 1;
        int r0, r1, r2;
        plan_rows(rows, cl, r0, r1, r2);
        pl.ri[0] = 3 * cl; pl.ri[1] = 16 * r0 + 4 * q; pl.ri[2] = 16 * r1 + 4 * q; pl.ri[3] = 16 * r2 + 4 * q;
        pl.ri[4] = col < NU + NS ? NS * RLD + 16 * col + 4 * q : -1;     // float index from Pan of the set, or none
        if (wv == 3) {
            // the second tile's results and their mirror images
            pl.ri[5] = d[0]; pl.ri[6] = d[1]; pl.ri[7] = d[2]; pl.ri[8] = d[3];
            pl.ri[9] = e[0]; pl.ri[10] = e[1]; pl.ri[11] = e[2]; pl.ri[12] = e[3];
        } else if (wv == 4) {
            pl.ri[9] = e[0]; pl.ri[10] = e[1]; pl.ri[11] = e[2]; pl.ri[12] = e[3];     // mirror images of tile (2, 2)
        }
        if (wv == 5) {
            // gradient column: z_g of a pair of pivot blocks by eight lanes per entry (term gt and gt + 8 of the sixteen; entry ge = 3 block + component in lanes 8 ge ..),
            // then lane <-> published row
            const int ge = ln >> 3, gt = ln & 7;
            const int gcomp = ge % 3, gblk = ge < 3 ? 0 : 1;
            const int grow2 = gt == 7 ? NPAN - 1 : gt + 8;
            pl.ri[5] = (NPAN * gblk + gt) * 4 + gcomp; pl.ri[6] = (NPAN * gblk + grow2) * 4 + gcomp;
            pl.ri[7] = gt; pl.ri[8] = gt == 7 ? 0 : gt + 8;
            pl.ri[10] = 4 * (ln < NS ? ln : (ln == NS ? NPAN - 1 : (ln < NPAN ? ln - 1 : 0)));   // the lane's published row: v of panel row r lives in lane r (r < NS) or r + 1
            // last step: M[45][j] -= sum_a w_a(j) v[r_a(j)], lane j < 45: the four v are fetched from their lanes (ds_bpermute: byte address 4 lane)
            const int jr = ln < NU + NS ? ln : NU + NS - 1;
            plan_rows(rows, jr, r0, r1, r2);
            auto vlane = [](int r) { return r < NS ? r : r + 1; };
            pl.ri[11] = vlane(r0) | (vlane(r1) << 8) | (vlane(r2) << 16) | (vlane(jr < NF ? NS + jr : 0) << 24);
            pl.ri[12] = 3 * jr;
            pl.rf[0] = gt == 7 ? 1.f : 0.f; pl.rf[1] = gblk ? 1.f : 0.f; pl.rf[2] = jr < NF ? -prm.D[jr % 3] : 0.f;
            pl.rf[3] = ln < NU ? 1.f : 0.f; pl.rf[4] = ln < NU + NS ? 1.f : 0.f;
        }
    }
}
// One stage of the consumers.  S: the set stage kb is assembled in (kb & 1); k: stage being factorised meanwhile (N: none -- the terminal call: only stage N-1 is
// assembled, no previous-force block: havep false); kd: stage whose descriptors wave 4 builds (-1: none); ord: ordinal of the stage within the pass.  All uniform.
__device__ __forceinline__ void sq_consume_stage(const Ctx& c, const CmpcConsts& prm, const ConsPlan& pl, int tid, int N, int k, int kb, int kd, int ord, bool havep,
                                                 bool use_exact, float reg, float cmu)
{
    const int S = kb & 1;   // (a runtime value: one copy of the body for both sets, the set offsets in scalar registers -- see sq_consume_loop)
    const int w7 = tid >> 6, ln = tid & 63;
    const int wv = w7 < 4 ? w7 - 1 : w7 - 2;
    if (w7 == 4) {
        // the factorising wave's SIMD mate: descriptors of stage kd, which nobody reads before the next stage (it is starved of issue slots: nothing else lives here)
        if (kd >= 0) {
            Ctx cd = c;
            use_desc_set(cd, kd & 1);
#pragma unroll 1
            for (int r = 0; r < 4; ++r) stage_desc_body(cd, prm, desc_role_t(r, ln), kd, use_exact, cmu);
        }
        return;
    }
#ifdef CMPC_PROFILE
    const long long pc0_ = __builtin_amdgcn_s_memtime();
#endif
    CPROF(0);
    // the sets: stage kb is assembled in set S (QuuF | Pan | Qb, float64 diagonal blocks, qs) from Qss, qs of stage kb + 1 in set S ^ 1 and the descriptors of stage kb (set S)
    float* Mn = c.QuuF + S * MSET;
    float* Pann = Mn + NU * RLD;
    const float* Qc = c.Qb + (S ^ 1) * MSET;
    double* qsn = S ? c.qs1 : c.qs;
    const double* qsc = S ? c.qs : c.qs1;
    double* QuuDn = S ? c.QuuD1 : c.QuuD;
    const float* Bval = c.Bval + S * DSET_F;       // Bval | Aval | arow contiguous: index 3 col for a column of [B | A], 3 NU + 3 NS + 3 + e for arow[e]
    const float* arow = c.arow + S * DSET_F;
    const double* sig = c.sig + S * 2 * NI;
    const double* gco = c.gco + S * 2 * NI;
    float* YT = Pann + NS * RLD;
    const float gam0 = c.sp[c.L.pGam(0) + kb], gam1 = c.sp[c.L.pGam(1) + kb];
    const float dgam = gam1 - gam0;
    const float* Wb = c.ZT;                        // published W^T, [block][panel row][4]
    // the tile's operand weights: descriptor values of stage kb, there since the stage before -- fetched now, used after the assembly
    const float wa0 = Bval[pl.wa], wa1 = Bval[pl.wa + 1], wa2 = Bval[pl.wa + 2];
    const float wb0 = Bval[pl.wb], wb1 = Bval[pl.wb + 1], wb2 = Bval[pl.wb + 2];
    bool gaveup = false;
    // ---- what needs no Y ----
    if (wv >= 3) {
        // Y^T = (Qss [B A])^T in the panel rows of the set that hold nothing in the streaming stage
        const float w0 = Bval[pl.ri[0]], w1 = Bval[pl.ri[0] + 1], w2 = Bval[pl.ri[0] + 2];
        const float4 x0 = *reinterpret_cast<const float4*>(Qc + pl.ri[1]);
        const float4 x1 = *reinterpret_cast<const float4*>(Qc + pl.ri[2]);
        const float4 x2 = *reinterpret_cast<const float4*>(Qc + pl.ri[3]);
        if (pl.ri[4] >= 0)
            *reinterpret_cast<float4*>(Pann + pl.ri[4]) = make_float4(w0 * x0.x + w1 * x1.x + w2 * x2.x, w0 * x0.y + w1 * x1.y + w2 * x2.y,
                                                                      w0 * x0.z + w1 * x1.z + w2 * x2.z, w0 * x0.w + w1 * x1.w + w2 * x2.w);
        CPROF2(0);
        if (ln == 0) lds_count(c.flag + 2);
    } else if (wv == 1) {
        // hb = [Qss d + qs; -D (u_kb+1 - u_kb)] (float64), then q_u = gradient of the symmetry cost + barrier terms + force-rate term + B~^T hb
        const bool pk = kb > 0;
        double* hb = c.Pd;
        const float* u = c.U + NU * kb;
        {
            const float* qcol = Qc + pl.ri[0];
            const float* dk = c.d + NS * kb;
            double acc = 0.0;
#pragma unroll 5     // (fully unrolled, its thirty loads in flight were the register peak of the whole consumer loop: 30 more callee-saved registers saved per wave and pass)
            for (int a = 0; a < NS; ++a) acc += (double)qcol[16 * a] * (double)dk[a];
            const double hs = qsc[pl.ri[0]] + acc;
            const float du = c.U[NU * (havep ? kb + 1 : kb) + pl.ri[1]] - u[pl.ri[1]];
            const double hr = -(double)pl.rf[2] * (double)du;
            if (ln < NXA) hb[ln] = pl.rf[4] != 0.f ? hs : hr;
        }
        wave_lds_sync();
        const int m = pl.ri[3], q = pl.ri[6];
        const float* uf = u + pl.ri[2];
        const float u0 = uf[0], u1 = uf[3], u2 = uf[6], u3 = uf[9], um = u[m];
        const float up = c.U[NU * (pk ? kb - 1 : 0) + m];
        const double gam = (double)fmaf(pl.rf[1], dgam, gam0);
        const double* gcp = gco + pl.ri[4];
        const float* arp = arow + pl.ri[5];
        double gF;
        {
            const double mean = 0.25 * ((double)u0 + (double)u1 + (double)u2 + (double)u3);
            const double esum = 4.0 * mean * (1.0 - gam);
            gF = 2.0 * prm.w_sym * (((double)um - gam * mean) - 0.25 * gam * esum);
#pragma unroll
            for (int f = 0; f < 4; ++f) gF += gcp[f] * (double)arp[3 * f];
            if (pk) gF += (double)pl.rf[3] * ((double)um - (double)up);
            if (havep) gF += hb[NS + m];
        }
        const bool fr = (c.qmask[kb] >> q) & 1;
        const double gQ = fr ? gco[32 + q] - gco[38 + q] : 0.0;
        double g = pl.rf[0] != 0.f ? gF : gQ;
        g += (double)Bval[pl.ri[7]] * hb[pl.ri[8]] + (double)Bval[pl.ri[7] + 1] * hb[pl.ri[9]] + (double)Bval[pl.ri[7] + 2] * hb[pl.ri[10]];
        if (ln < NU) {
            c.pv[ln] = g;                                          // float64 until -Z^T z_g has been subtracted (the gradient role)
            if (!havep) Pann[(NPAN - 1) * RLD + ln] = (float)g;   // (the terminal stage has no Z)
        }
    } else if (wv == 2) {
        // q_s = gradient of the tracking cost + A^T hb[0..14], its own copy of those rows of hb in c.pn
        double* hs = c.pn;
        {
            const float* qcol = Qc + pl.ri[0];
            const float* dk = c.d + NS * kb;
            double acc = 0.0;
#pragma unroll 5     // (fully unrolled, its thirty loads in flight were the register peak of the whole consumer loop: 30 more callee-saved registers saved per wave and pass)
            for (int a = 0; a < NS; ++a) acc += (double)qcol[16 * a] * (double)dk[a];
            if (ln < NS) hs[ln] = qsc[pl.ri[0]] + acc;
        }
        wave_lds_sync();
        const int j = pl.ri[0];
        const double wj = (double)fmaf(pl.rf[1], prm.wz2[kb], pl.rf[0]);
        const float* geo = c.geoA + GEO * kb;
        const double gam = (double)fmaf(pl.rf[2], dgam, gam0);
        const double dt = prm.dt;
        const double sj = pl.rf[5] != 0.f ? gam : 1.0;
        const double ce = (double)pl.rf[4] * dt;
        const double cg = (double)pl.rf[3] * dt - (double)pl.rf[5] * dt * gam;
        const double sv = (double)c.S[NS * kb + j], rv = (double)c.sp[pl.ri[1] + 3 * kb];
        const double F1 = (double)geo[pl.ri[2]], F2 = (double)geo[pl.ri[3]];
        const double out = wj * (sv - rv) + sj * hs[j] + ce * hs[pl.ri[4]] + cg * (hs[pl.ri[5]] * F2 - hs[pl.ri[6]] * F1);
        if (ln < NS) qsn[j] = out;
    }
    CPROF2(1);
    gaveup = !lds_wait_ge(c.flag + 2, SQ_Y_WAVES * (ord + 1));
    CPROF2(2);
    // ---- behind Y ----
    if (wv == 0) {
        // the ten 3x3 diagonal blocks of Quu in float64 (cost, barrier and Levenberg terms + the sandwich entry)
        const bool pk = kb > 0;
        const float w0 = Bval[pl.ri[0]], w1 = Bval[pl.ri[0] + 1], w2 = Bval[pl.ri[0] + 2];
        const double v = (double)(w0 * YT[pl.ri[1]] + w1 * YT[pl.ri[2]] + w2 * YT[pl.ri[3]]);
        const double* sg = sig + pl.ri[4];
        const float* ar = arow + pl.ri[5];
        const float* ac = arow + pl.ri[6];
        const double gam = (double)fmaf(pl.rf[2], dgam, gam0);
        double vF = v;
        if (pl.rf[1] != 0.f) {
            vF += 2.0 * prm.w_sym * (1.0 - 0.25 * gam * (2.0 - gam));
            if (pk) vF += (double)pl.rf[3];       // own force-rate cost
            if (havep) vF += (double)pl.rf[3];    // E^T D E: the next stage's force-rate cost
            vF += (double)reg;
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) vF += sg[f] * (double)ar[3 * f] * (double)ac[3 * f];
        const int iq = pl.ri[7];
        const bool fr = (c.qmask[kb] >> iq) & 1;
        const double vQ = pl.rf[1] != 0.f ? (fr ? v + sig[32 + iq] + sig[38 + iq] + (double)reg : 1.0) : v;   // fixed q: exact identity row
        QuuDn[pl.ri[8]] = pl.rf[0] != 0.f ? vF : vQ;
        Mn[pl.ri[9]] = 0.f;   // the float copy of a diagonal block collects -Z^T Z and the updates by earlier blocks
    }
    {
        // the thread's triple
        const float w0 = Bval[pl.t_w], w1 = Bval[pl.t_w + 1], w2 = Bval[pl.t_w + 2];
        const float* y0 = YT + pl.t_y0;
        const float* y1 = YT + pl.t_y1;
        const float* y2 = YT + pl.t_y2;
        const float* E = arow + pl.t_e;
        const float gq = fmaf(pl.t_fsel, dgam, gam0);
        const float ex = fmaf(pl.t_zsel, prm.wz2[kb], pl.t_qdc) - pl.t_symk * gq * (2.f - gq);
        const float ew = pl.t_esgn * fmaf(pl.t_efsel, dgam, gam0);
        float* dst = Mn + pl.t_dst;
        dst[0] = w0 * y0[0] + w1 * y1[0] + w2 * y2[0] + ew * E[0] + pl.t_m0 * ex;
        dst[1] = w0 * y0[16] + w1 * y1[16] + w2 * y2[16] + ew * E[3] + pl.t_m1 * ex;
        dst[2] = w0 * y0[32] + w1 * y1[32] + w2 * y2[32] + ew * E[6] + pl.t_m2 * ex;
    }
    if (ln == 0) lds_count(c.flag + 3);
    CPROF(1);
    if (k < N) {
        const int fixedmask = (~c.qmask[k]) & 63;
        const bool sk8 = ((fixedmask >> 0) & 7) == 7, sk9 = ((fixedmask >> 3) & 7) == 7;   // (a stance foot's block is skipped by the factorisation: nothing is published)
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        const int seq0 = 16 * ord;
        int avail = 0;                       // pivot blocks known to be published (uniform)
        auto need = [&](int nblk) {
            if (avail >= nblk) return;
            int spins = 0;
            for (;;) {
                avail = __builtin_amdgcn_readfirstlane(lds_peek(c.prog)) - seq0;
                if (avail >= nblk) break;
                if (++spins > SQ_SPIN_MAX) { gaveup = true; avail = 16; break; }
                __builtin_amdgcn_s_sleep(1);                // (polling the stage's last block without the sleep changes nothing: measured)
            }
        };
        const float* pa0 = Wb + pl.pa0; const float* pa1 = Wb + pl.pa1; const float* pa2 = Wb + pl.pa2; const float* pa3 = Wb + pl.pa3;
        const float* pb0 = Wb + pl.pb0; const float* pb1 = Wb + pl.pb1; const float* pb2 = Wb + pl.pb2; const float* pb3 = Wb + pl.pb3;
        auto operands = [&](int b, float& a, float& bv) {
            const int o = NPAN * 4 * b;
            const float x0 = pa0[o], x1 = pa1[o], x2 = pa2[o], x3 = pa3[o];
            const float y0 = pb0[o], y1 = pb1[o], y2 = pb2[o], y3 = pb3[o];   // (a diagonal tile loads the same four again: no branch in the load stream)
            // (the A operand negated: the accumulator then holds  assembled - Z^T Z  itself; both operands in one packed multiply-add chain)
            v2f ab = v2f{-wa0, wb0} * v2f{x0, y0};
            ab = __builtin_elementwise_fma(v2f{-wa1, wb1}, v2f{x1, y1}, ab);
            ab = __builtin_elementwise_fma(v2f{-wa2, wb2}, v2f{x2, y2}, ab);
            ab = __builtin_elementwise_fma(v2f{-pl.wa3, pl.wb3}, v2f{x3, y3}, ab);
            a = ab[0]; bv = ab[1];
        };
        // The gradient column (row 45 of M) has a wave of its own (no tile: with one it left the stage ~450 cycles after the others, every stage): v = W^T z_g a pair
        // of pivot blocks at a time -- see the plan (ri[5..12]) --, and at the end M[45][j] -= sum_a w_a(j) v[r_a(j)], the same sparse combination as a row of Z^T.
        // No LDS round trip: the six z of a pair are broadcast by v_readlane, the four v of the last step come from their lanes by ds_bpermute.
        const bool grole = wv == 5;
        if (grole) {
            float vacc = 0.f;
            const float gw0 = c.d[NS * kb + pl.ri[7]];
            const float gw1 = pl.rf[0] != 0.f ? 1.f : c.d[NS * kb + pl.ri[8]];
            const float gz0 = Bval[pl.ri[12]], gz1 = Bval[pl.ri[12] + 1], gz2 = Bval[pl.ri[12] + 2];
            double gbase = 0.0;
            auto gradient_pair = [&](int b, bool skip0, bool skip1) {
                const int o = NPAN * 4 * b;
                const float4 xa = *reinterpret_cast<const float4*>(Wb + o + pl.ri[10]);
                const float4 xb = *reinterpret_cast<const float4*>(Wb + o + NPAN * 4 + pl.ri[10]);
                const float z = oct_sum(gw0 * Wb[o + pl.ri[5]] + gw1 * Wb[o + pl.ri[6]]);
                if (!skip0) vacc += xa.x * readlane_f(z, 0) + xa.y * readlane_f(z, 8) + xa.z * readlane_f(z, 16);
                if (!skip1) vacc += xb.x * readlane_f(z, 24) + xb.y * readlane_f(z, 32) + xb.z * readlane_f(z, 40);
            };
            CPROF(2);
#pragma unroll
            for (int b = 0; b < 8; b += 2) {
                if (b == 6 && sk8 && sk9) {
                    // (the last pair one block at a time, like the tiles)
                    need(7);
                    gradient_pair(6, false, true);
                    need(8);
                    gradient_pair(6, true, false);
                    continue;
                }
                need(b + 2);
                gradient_pair(b, false, false);
                if (b == 0) {
                    // (q_u of wave 2 and q_s of wave 3 must be there: the assembly count covers them)
                    gaveup = !lds_wait_ge(c.flag + 3, SQ_TILE_WAVES * (ord + 1)) || gaveup;
                    gbase = pl.rf[3] != 0.f ? c.pv[ln < NU ? ln : 0] : qsn[ln >= NU && ln < NU + NS ? ln - NU : 0];
                }
            }
            CPROF(3);
            if (!sk8 && !sk9) {
                need(9);
                gradient_pair(8, false, true);
                need(10);
                gradient_pair(8, true, false);
            } else if (!(sk8 && sk9)) {
                need(sk9 ? 9 : 10);
                gradient_pair(8, sk8, sk9);
            }
            CPROF2(4);
            const unsigned rw = (unsigned)pl.ri[11];
            const int vi = __float_as_int(vacc);
            auto vof = [&](unsigned lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(lane << 2), vi)); };
            const float val = gz0 * vof(rw & 255u) + gz1 * vof((rw >> 8) & 255u) + gz2 * vof((rw >> 16) & 255u) + pl.rf[2] * vof(rw >> 24);
            const double res = gbase - (double)val;
            if (ln < NU) Pann[(NPAN - 1) * RLD + ln] = (float)res;
            else if (ln < NU + NS) qsn[ln - NU] = res;
        } else {
            const bool two = wv == 3;             // tiles (1, 0) and (1, 1): the second is a x a of the same operand rows
            v4f acc2 = {0.f, 0.f, 0.f, 0.f};
            CPROF(2);
            // blocks 0..7 (the forces: never skipped) in pairs: the loads of both in flight, two chained MFMAs
#pragma unroll
            for (int b = 0; b < 8; b += 2) {
                float a0, b0, a1, b1;
                if (b == 6 && sk8 && sk9) {
                    // the last pair of a stage without landing-offset blocks, one block at a time: only block 7's operands and one matrix-pipe instruction are
                    // left when the factorising wave publishes its last block
                    need(7);
                    operands(6, a0, b0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
                    if (two) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, -a0, acc2, 0, 0, 0);
                    need(8);
                    operands(7, a1, b1);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc, 0, 0, 0);
                    if (two) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, -a1, acc2, 0, 0, 0);
                    continue;
                }
                need(b + 2);
                operands(b, a0, b0);
                operands(b + 1, a1, b1);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
                if (two) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, -a0, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc, 0, 0, 0);
                if (two) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, -a1, acc2, 0, 0, 0);
                if (b == 0) {
                    // every consumer wave's part of the assembly must be in LDS before the tiles are subtracted from it (long there by now); then what the tile is
                    // subtracted from is fetched and ADDED to the accumulator (the products enter negated) -- far ahead of the last blocks: behind the last block
                    // only its matrix-pipe instruction and the stores are left
                    gaveup = !lds_wait_ge(c.flag + 3, SQ_TILE_WAVES * (ord + 1)) || gaveup;
                    acc += v4f{Mn[pl.d0], Mn[pl.d1], Mn[pl.d2], Mn[pl.d3]};
                    if (two) acc2 += v4f{Mn[pl.ri[5]], Mn[pl.ri[6]], Mn[pl.ri[7]], Mn[pl.ri[8]]};
                }
            }
            CPROF(3);
            // blocks 8, 9: the landing offsets of the two feet
            if (!sk8 && !sk9) {
                float a0, b0, a1, b1;
                need(9);
                operands(8, a0, b0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
                if (two) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, -a0, acc2, 0, 0, 0);
                need(10);
                operands(9, a1, b1);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc, 0, 0, 0);
                if (two) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, -a1, acc2, 0, 0, 0);
            } else if (!sk8 || !sk9) {
                const int b = sk8 ? 9 : 8;
                need(b + 1);
                float a0, b0;
                operands(b, a0, b0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
                if (two) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, -a0, acc2, 0, 0, 0);
            }
            CPROF2(4);
            {
                const float r0 = acc[0], r1 = acc[1], r2 = acc[2], r3 = acc[3];
                Mn[pl.d0] = r0; Mn[pl.d1] = r1; Mn[pl.d2] = r2; Mn[pl.d3] = r3;
                if (wv == 2 || wv == 4) { Mn[pl.ri[9]] = r0; Mn[pl.ri[10]] = r1; Mn[pl.ri[11]] = r2; Mn[pl.ri[12]] = r3; }   // (tiles with entries of Qss: both triangles)
            }
            if (two) {
                const float r0 = acc2[0], r1 = acc2[1], r2 = acc2[2], r3 = acc2[3];
                Mn[pl.ri[5]] = r0; Mn[pl.ri[6]] = r1; Mn[pl.ri[7]] = r2; Mn[pl.ri[8]] = r3;
                Mn[pl.ri[9]] = r0; Mn[pl.ri[10]] = r1; Mn[pl.ri[11]] = r2; Mn[pl.ri[12]] = r3;
            }
        }
        CPROF(4);
    }
    if (gaveup && ln == 0) sq_give_up(c);
}
// The consumers' side of a backward pass (waves 1..7), one call: the plan, the assembly of stage N-1 from the terminal cost, then a stage per barrier.
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) void sq_consume_loop(lds_t lds, int Nrt, float* fg_base, int k0_in, bool exact_in, float reg, float cmu)
{
    CMPC_PHASE_PROLOGUE;
    const int k0 = __builtin_amdgcn_readfirstlane(k0_in);
    const bool use_exact = __builtin_amdgcn_readfirstlane((int)exact_in) != 0;
    ConsPlan pl;
    cons_plan_build(pl, c, prm, tid >> 6, tid & 63, c.Brow + ((N - 1) & 1) * DSET_I);   // (the rows of set (N-1) & 1: built by sq_init whatever k0 is)
    // ONE copy of the stage body serves both sets (the set offsets in scalar registers): the consumers wait for the factorisation, so the few scalar instructions a runtime
    // set costs them are free, and a body per set was 12 KB more code in a 64 KB instruction cache that an iteration's ~80 KB cycle through: +0.6 % at B = 256.
    // (The terminal assembly -- stage N-1 from the terminal cost, no value function yet -- stays a call of its own: with `havep` a runtime value as well the stage
    //  body lost what the single copy had gained, 341.9 k -> 335.6 k solves/s.)
    sq_consume_stage(c, prm, pl, tid, N, N, N - 1, -1, 0, false, use_exact, reg, cmu);
    __syncthreads();
    int ord = 0;
#pragma unroll 1
    SQ_STAGE_LOOP(k, N, k0) {                          // (sq_pass_barriers: one barrier per trip, no early exit)
        ++ord;
        if (k > k0) sq_consume_stage(c, prm, pl, tid, N, k, k - 1, k - 2 >= k0 ? k - 2 : -1, ord, true, use_exact, reg, cmu);
        __syncthreads();
    }
}
// terminal "stage": Qss_N = diag(Q_N), qs_N = gradient of the terminal cost, in set N & 1; descriptors of the last two stages
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) void sq_init(lds_t lds, int Nrt, float* fg_base, int k0_in, bool exact_in, float cmu)
{
    CMPC_PHASE_PROLOGUE;
    const int k0 = __builtin_amdgcn_readfirstlane(k0_in);
    const bool use_exact = __builtin_amdgcn_readfirstlane((int)exact_in) != 0;
    {
        float* QN = c.Qb + (N & 1) * MSET;
        double* qN = (N & 1) ? c.qs1 : c.qs;
        if (tid < NS * 16) QN[tid] = (tid >> 4) == (tid & 15) ? qdiag(prm, N, tid & 15) : 0.f;
        else if (tid >= 256 && tid < 256 + NS) qN[tid - 256] = grad_track(c, prm, N, tid - 256);
        if (tid < 4) c.flag[tid] = 0;   // failure flag | sync give-up | Y count | assembly count (streaming stage)
        if (tid >= 128 && tid < 256) c.prog[tid - 128] = 0;   // progress of the factorising wave
        // the identity rows 18..29 take over lanes of finished L rows at block LATE_B: their entries of the blocks before it are zeros nobody ever writes
        if (tid >= 256 && tid < 256 + LATE_B * NLATE) {
            const int e = tid - 256, b = e / NLATE, r = e % NLATE;
            *reinterpret_cast<float4*>(c.ZT + (NPAN * b + NS + LATE_M0 + r) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (tid >= 256) {
        const int kd = tid >= 384 ? N - 1 : N - 2;
        if (kd >= k0) {
            use_desc_set(c, kd & 1);
            stage_desc_body(c, prm, desc_pack_t(tid & 127), kd, use_exact, cmu);
        }
    }
    __syncthreads();
}
template <int NT, int NC, bool FG>
__device__ inline int riccati_backward_sq(lds_t lds, const Ctx& c, float* fg_base, bool use_exact, float reg, float cmu, int k0)
{
    const int N = c.N;
    sq_init<NT, NC, FG>(lds, N, fg_base, k0, use_exact, cmu);
    if (threadIdx.x < 64) sq_factor_loop<NT, NC, FG>(lds, N, fg_base, k0);
    else sq_consume_loop<NT, NC, FG>(lds, N, fg_base, k0, use_exact, reg, cmu);
    // (a non-positive pivot raises the flag and the stages after it run on garbage, harmlessly -- every array they write is rebuilt by the retry.
    //  Bit 1: a wave gave up waiting at a hand-off word -- reported apart from a bad pivot, see the driver)
    return (c.flag[0] ? 1 : 0) | (c.flag[1] ? 2 : 0);
}

// ---- Riccati backward sweep (matrices + right-hand side of the affine step, or of a centring step with target
// cmu).  Returns (uniformly) 0 ok, 1 non-positive pivot. ----
// k0 > 0: only stages N-1 .. k0 (the tail polish: the state entering stage k0 is held, so nothing before it is needed).
template <int NT, int NC, bool FG>
__device__ int riccati_backward(lds_t lds, const Ctx& c, const CmpcConsts& prm, int tid, float* fg_base, bool use_exact, float reg, float cmu, int k0 = 0)
{
    if constexpr (!FG) return riccati_backward_sq<NT, NC, FG>(lds, c, fg_base, use_exact, reg, cmu, k0);   // (resident variants: the streaming square-root stage)
    const int N = c.N;
    for (int e = tid; e < NXA * PLD; e += NT) c.P0[e] = 0.f;
    if (tid == 0) *c.flag = 0;
    __syncthreads();
    if (tid < NS) {
        c.P0[tid * PLD + tid] = qdiag(prm, N, tid);
        c.pv[tid] = grad_track(c, prm, N, tid);
    } else if (tid < NXA) c.pv[tid] = 0.0;
    if (tid >= 128) stage_desc<NT, NC, FG>(lds, N, fg_base, N - 1, use_exact, cmu);
    __syncthreads();
    // Which entries of the triangle a thread owns does not change from stage to stage: the index-table look-ups (one dependent
    // LDS round trip at the head of phases 2, 3 and 4) are made once here and handed down packed:
    //   tpk = tri[id / 3] of the thread's (up to two) Quu triples;  tqp = tri[tid - 128] (Qss entry) | tri[tile] << 16 (phase 4)
    int tpk, tqp;
    {
        const int id0 = tid, id1 = tid + 128;                       // (ids >= 135 are panel triples: no look-up)
        tpk = (id0 < 135 ? c.tri[id0 / 3] : 0) | ((id1 < 135 ? c.tri[id1 / 3] : 0) << 16);
        const int tq = tid - 128, tile = tid;
        tqp = ((tq >= 0 && tq < 120) ? c.tri[tq] : 0) | ((tile < 210 ? c.tri[tile] : 0) << 16);
    }
    // P0 holds the value function of stage k+1 and is overwritten in place by phase 4 (its last reader, Qss, ran in phase 3)
    stage_pre<NT, NC, FG>(lds, N, fg_base, N - 1, false, use_exact, reg, cmu, tpk);
    for (int k = N - 1; k >= k0; --k) {
        // (a double-stance stage has its own copy of the factorisation: chol_block's FIX)
        if (((~__builtin_amdgcn_readfirstlane(c.qmask[k])) & 63) == 63) stage_mid<NT, NC, FG, true>(lds, N, fg_base, k, use_exact, cmu, tqp);
        else stage_mid<NT, NC, FG, false>(lds, N, fg_base, k, use_exact, cmu, tqp);
        if (k > k0 || k0 == 0) stage_post_pre<NT, NC, FG>(lds, N, fg_base, k, use_exact, reg, cmu, tpk, tqp);   // (the value function of stage k0 > 0 has no reader)
    }
    // (a non-positive pivot raises the flag and the stages after it run on garbage, harmlessly -- every array they write is
    // rebuilt by the retry; testing the flag once here instead of once per stage takes an LDS round trip out of every stage)
    return *c.flag ? 1 : 0;
}

// ---- forward sweep on wave 0: dS, dU.  All threads then compute dT, dZ (dZ holds the per-row
// complementarity target on entry unless affine).
// Per stage:  y = [Ws lq | -L^{-1}[:, :24] D] [ds; 1; du_prev]   (two lanes per row, 16-byte reads)
//             du = -L^{-T} y                                      (two lanes per column)
//             ds+ = A ds + B du + d                               (9 + 6 lanes, corner sums by DPP)
// Matrix operands do not depend on the recursion: they are read at the top of the stage. ----
// k0 > 0 (tail polish): the sweep starts at stage k0 with ds_k0 = 0 and the force step before it zero.
// PART 0: the whole of it; 1: the sweep alone (wave 0, no barrier); 2: the barrier and the element-wise part (all threads).
// The sweep itself (one wave).  A wave executes one instruction of any kind per four cycles, so the sweep is as long as its instruction count: everything that
// does not change from stage to stage -- per-lane operand addresses, role masks -- is set up once; a trip of UNR stages addresses its operands as pointer + immediate
// and bumps the pointers once; results of lanes without a task go to words of the staging buffer nobody reads instead of being masked off (round 4: 340 -> ~170
// instructions per stage).
template <int UNR, bool G>
__device__ __forceinline__ void forward_sweep(const Ctx& c, const CmpcConsts& prm, int tid, int k0)
{
    const int N = c.N;
    const int r = tid & 31, half = tid >> 5, blk = r >> 2;
    float* xb = c.ybuf;           // [ds (15), 1, -D du_prev (24)]
    float* yb = c.ybuf + 40;      // y (32)
    float* trash = c.ybuf + 72;   // 24 words nobody reads
    // per-lane offsets inside a stage record (loop invariant)
    // y-step: column r of [WT; U[0..23]] against x, 20 terms per half;  du-step: row r of U against y
    unsigned yoff[20], uoff[4];
#pragma unroll
    for (int t = 0; t < 20; ++t) {
        if (half == 0) yoff[t] = t < 16 ? wt_idx(t, r) : ub_row(t - 16) + r;
        else {
            const int m = 4 + t;
            yoff[t] = blk >= (m >> 2) ? ub_row(m) + r - 4 * (m >> 2) : REC_ZERO;
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int q4 = 4 * half + t;
        uoff[t] = q4 >= blk ? ub_row(r) + 4 * (q4 - blk) : REC_ZERO;
    }
    // roles in the dynamics step: lanes 0..23 form the corner terms (axis ga of corner cj), lanes 0..14 then
    // evaluate one row each of  ds+ = A ds + B du + d  with per-lane coefficients, branch-free:
    //   out = sj ds_j + ce ds_je + cD sumD_a + cH (sumH_a + dt sum_ct gam_ct (e_ct x Fc_ct)_a) + cp (R_ct dq_ct)_a + d_j
    const bool corner = tid < NF;
    const int ga = corner ? tid >> 3 : 0, cj = corner ? tid & 7 : 0;
    const int ga1 = (ga + 1) % 3, ga2 = (ga + 2) % 3;
    const int j = tid < NS ? tid : 0;
    const int ja = j % 3, ja1 = (ja + 1) % 3, ja2 = (ja + 2) % 3, jct = j >= 12 ? 1 : 0;
    const int je = j < 3 ? j + 3 : 0;
    const float ce = j < 3 ? prm.dt : 0.f;
    const float cD = (j >= 3 && j < 6) ? 1.f : 0.f, cH = (j >= 6 && j < 9) ? 1.f : 0.f;
    const float posm = j >= 9 ? 1.f : 0.f;
    const float cDm0 = ja == 0 ? cD : 0.f, cDm1 = ja == 1 ? cD : 0.f, cDm2 = ja == 2 ? cD : 0.f;
    const float cHm0 = ja == 0 ? cH : 0.f, cHm1 = ja == 1 ? cH : 0.f, cHm2 = ja == 2 ? cH : 0.f;
    const float gdtm = corner ? prm.dt : 0.f, gsel = cj >= 4 ? 1.f : 0.f, jsel = jct ? 1.f : 0.f;
    const float Dm = prm.D[r % 3];
    const float dtc = prm.dt;
    if (tid < 40) xb[tid] = tid == 15 ? 1.f : 0.f;
    if (tid < NS) c.dS[NS * k0 + tid] = 0.f;
    wave_lds_sync();
    // operand pointers of stage k0: LDS addresses, each made opaque -- left to itself the optimiser notices that they all advance together, keeps ONE base and
    // re-adds the lane's offset in front of every load (a v_add per operand and stage: 40 of the old 340 instructions)
    ldsf_t yp[20];
    ldsf_t upp[4];
    if (!G) {
#pragma unroll
        for (int t = 0; t < 20; ++t) yp[t] = lds_opaque(c.Lf + (size_t)REC_N * k0 + yoff[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) upp[t] = lds_opaque(c.Lf + (size_t)REC_N * k0 + uoff[t]);
    }
    ldsf_t gr1 = lds_opaque(c.geoA + GEO * k0 + 3 * cj + ga1);
    ldsf_t gr2 = lds_opaque(c.geoA + GEO * k0 + 3 * cj + ga2);
    ldsf_t gf1 = lds_opaque(c.geoA + GEO * k0 + 24 + ja1);
    ldsf_t gf2 = lds_opaque(c.geoA + GEO * k0 + 24 + ja2);
    ldsf_t gm0 = lds_opaque(c.sp + c.L.pGam(0) + k0);
    ldsf_t gm1 = lds_opaque(c.sp + c.L.pGam(1) + k0);
    ldsf_t rp = lds_opaque(c.sp + c.L.pR(jct) + 9 * k0 + ja);          // R(ja, 0..2) = rp[0], rp[3], rp[6] (vec(R) is column-major)
    ldsf_t dp = lds_opaque(c.d + NS * k0 + j);
    ldsf_t du0p = lds_opaque(c.dU + NU * k0 + 3 * cj + ga);
    ldsf_t du1p = lds_opaque(c.dU + NU * k0 + 3 * cj + ga1);
    ldsf_t du2p = lds_opaque(c.dU + NU * k0 + 3 * cj + ga2);
    ldsf_t dqp = lds_opaque(c.dU + NU * k0 + 24 + 3 * jct);
    // where results go: lanes without one keep writing to a fixed word of their own
    ldsw_t dUs = lds_opaque_w(tid < NU ? c.dU + NU * k0 + tid : trash + (tid & 7));
    const int dUst = tid < NU ? NU : 0;
    ldsw_t dSs = lds_opaque_w(tid < NS ? c.dS + NS * (k0 + 1) + tid : trash + 8 + (tid & 7));
    const int dSst = tid < NS ? NS : 0;
    ldsw_t xbs = lds_opaque_w(tid < NS ? xb + tid : trash + 16 + (tid & 3));
    ldsw_t xps = lds_opaque_w(tid < NF ? xb + 16 + tid : trash + 20 + (tid & 3));
    ldsw_t ybs = lds_opaque_w(yb + r);
    ldsf_t xsp = lds_opaque(xb + j), xep = lds_opaque(xb + je), x1p = lds_opaque(xb + ja1), x2p = lds_opaque(xb + ja2);
    ldsf_t xvp = lds_opaque(xb + 20 * half), yvp = lds_opaque(yb + 16 * half);
    PROF2_DECL;
    // records in HBM: the operands of a stage are fetched ONE STAGE AHEAD (they do not depend on the recursion): an L2 / fabric round trip of ~250-500 cycles per stage and
    // sweep was exposed at the top of every stage (4 % of a config-3 solve, profiles/r03_pmc_wait_config3.json)
    // (only the twenty scalars of the y-step: the four float4 of the du-step are issued at the top of their own stage and arrive under the y-step; a stage ahead
    //  as well they cost sixteen more registers, and at 168 the sweep then reloaded two spilled pointers at the top of every trip, each behind an s_waitcnt vmcnt(0))
    float ymn[20];
    auto fetch = [&](int k) {
        const RecRef<G> rec(c.Lf, N, k);
#pragma unroll
        for (int t = 0; t < 20; ++t) ymn[t] = rec.ld(yoff[t]);
    };
    if (G) fetch(k0);
    auto stage = [&](int i, int k) {
        float ym[20];
        float4 um[4];
        if (G) {
            {
                const RecRef<G> rec(c.Lf, N, k);
#pragma unroll
                for (int t = 0; t < 4; ++t) um[t] = rec.ld4(uoff[t]);
            }
#pragma unroll
            for (int t = 0; t < 20; ++t) ym[t] = ymn[t];
            fetch(k + 1 < N ? k + 1 : k);
        } else {
#pragma unroll
            for (int t = 0; t < 20; ++t) ym[t] = yp[t][REC_N * i];
#pragma unroll
            for (int t = 0; t < 4; ++t) um[t] = lds_ld4(upp[t] + REC_N * i);
        }
        // stage data of the dynamics step (independent of the recursion as well)
        const float g_r1 = gr1[GEO * i], g_r2 = gr2[GEO * i];
        const float F01 = gf1[GEO * i], F11 = gf1[GEO * i + 3], F02 = gf2[GEO * i], F12 = gf2[GEO * i + 3];
        const float gam0 = gm0[i], gam1 = gm1[i];
        const float R0 = rp[9 * i], R1 = rp[9 * i + 3], R2 = rp[9 * i + 6];
        const float dk = dp[NS * i];
        const float dgam = gam1 - gam0;
        const float g_dt = gdtm * fmaf(gsel, dgam, gam0);
        const float gamj = fmaf(jsel, dgam, gam0);
        PROF2(20);
        float4 xv[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) xv[t] = lds_ld4(xvp + 4 * t);
        // (packed multiply-add chains -- v_pk_fma_f32: two multiply-adds per instruction, and instructions are what a single-wave phase pays for; the operands
        //  arrive in register pairs: 16-byte loads, or two 4-byte loads the allocator places side by side.  20 scalar multiply-adds -> 10 packed + 3 adds here, 16 -> 8 in
        //  the du-step; the forward sweep 1 117 -> 1 019 instructions per four stages, the corrector sweep 1 078 -> 984 per two: +3.5 % at B = 256, +1.7 % on configs 3 / 5)
        float ya, yc;
        {
            v2f yp = v2f{ym[0], ym[1]} * v2f{xv[0].x, xv[0].y}, yq = v2f{ym[2], ym[3]} * v2f{xv[0].z, xv[0].w};
#pragma unroll
            for (int t = 1; t < 5; ++t) {
                yp = __builtin_elementwise_fma(v2f{ym[4 * t], ym[4 * t + 1]}, v2f{xv[t].x, xv[t].y}, yp);
                yq = __builtin_elementwise_fma(v2f{ym[4 * t + 2], ym[4 * t + 3]}, v2f{xv[t].z, xv[t].w}, yq);
            }
            yp += yq;
            ya = yp[0]; yc = yp[1];
        }
        const float y = half_sum(ya + yc);
        *ybs = y;                       // (both halves hold the same y: the same value to the same word)
        wave_lds_sync();
        PROF2(21);
        float4 yv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) yv[q] = lds_ld4(yvp + 4 * q);
        float wsum;
        {
            // two packed multiply-add chains (v_pk_fma_f32: two lanes' worth per instruction; both operands come in register pairs out of 16-byte loads)
            v2f wp = v2f{um[0].x, um[0].y} * v2f{yv[0].x, yv[0].y}, wq = v2f{um[0].z, um[0].w} * v2f{yv[0].z, yv[0].w};
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                wp = __builtin_elementwise_fma(v2f{um[q].x, um[q].y}, v2f{yv[q].x, yv[q].y}, wp);
                wq = __builtin_elementwise_fma(v2f{um[q].z, um[q].w}, v2f{yv[q].z, yv[q].w}, wq);
            }
            wp += wq;
            wsum = wp[0] + wp[1];
        }
        const float du = -half_sum(wsum);
        *dUs = du;
        dUs += dUst;
        wave_lds_sync();
        PROF2(22);
        // ---- ds+ = A ds + B du + d ----
        const float u0 = du0p[NU * i], u1 = du1p[NU * i], u2 = du2p[NU * i];
        const float xs = *xsp, xe = *xep;
        const float c1 = x1p[0], c2 = x2p[0];
        const float p01 = x1p[9], p02 = x2p[9], p11 = x1p[12], p12 = x2p[12];
        const float q0 = dqp[NU * i], q1 = dqp[NU * i + 1], q2 = dqp[NU * i + 2];
        const float tD = oct_sum(g_dt * u0);
        const float tH = oct_sum(g_dt * (g_r1 * u2 - g_r2 * u1));
        // the three axis sums of both kinds, broadcast; every lane takes its own by 0/1 coefficients (a select on the lane's axis made the compiler branch
        // around each v_readlane: 40 instructions of control flow per stage)
        const float sDH = cDm0 * readlane_f(tD, 0) + cDm1 * readlane_f(tD, 8) + cDm2 * readlane_f(tD, 16)
                          + cHm0 * readlane_f(tH, 0) + cHm1 * readlane_f(tH, 8) + cHm2 * readlane_f(tH, 16);
        const float cross = gam0 * ((p01 - c1) * F02 - (p02 - c2) * F01) + gam1 * ((p11 - c1) * F12 - (p12 - c2) * F11);
        const float land = R0 * q0 + R1 * q1 + R2 * q2;
        // (position rows: gam ds_j + (1 - gam) R dq; the others: ds_j + ...)
        const float out = fmaf(posm, gamj - 1.f, 1.f) * xs + ce * xe + sDH + cH * dtc * cross + dk + posm * (1.f - gamj) * land;
        wave_lds_sync();
        *dSs = out;
        dSs += dSst;
        *xbs = out;
        *xps = -Dm * du;
        wave_lds_sync();
        PROF2(23);
    };
    auto bump = [&](int n) {
        if (!G) {
#pragma unroll
            for (int t = 0; t < 20; ++t) yp[t] += REC_N * n;
#pragma unroll
            for (int t = 0; t < 4; ++t) upp[t] += REC_N * n;
        }
        gr1 += GEO * n; gr2 += GEO * n; gf1 += GEO * n; gf2 += GEO * n; gm0 += n; gm1 += n; rp += 9 * n; dp += NS * n;
        du0p += NU * n; du1p += NU * n; du2p += NU * n; dqp += NU * n;
    };
    int k = k0;
#pragma unroll 1
    for (; k + UNR <= N; k += UNR) {
#pragma unroll
        for (int i = 0; i < UNR; ++i) stage(i, k + i);
        bump(UNR);
    }
#pragma unroll 1
    for (; k < N; ++k) {
        stage(0, k);
        bump(1);
    }
}
template <int NT, int UNR, bool G, int PART = 0>
__device__ void riccati_forward(const Ctx& c, const CmpcConsts& prm, int tid, bool affine, int k0)
{
    const int N = c.N;
    if (PART != 2 && tid < 64) forward_sweep<UNR, G>(c, prm, tid, k0);
    if (PART == 1) return;
    __syncthreads();
    for (int e = tid + NI * k0; e < N * NI; e += NT) {
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

// ---- corrector right-hand side on wave 0: the row coefficients change by w = CMU / t.  On entry dT
// holds w (0 on inactive rows).  Updates lq (slot 15 of the Ws rows) in place through the stored factors.
// Per stage:  g = C^T w + fp_p + B^T fp_s ;  dl = L^{-1} g ;  lq += dl ;
//             fp_s <- A^T fp_s - Ws^T dl ;  fp_p <- D L^{-T} dl ----
// (lean form like forward_sweep: operand pointers set up once and addressed as pointer + immediate inside a trip of UNR stages, unmasked stores)
template <int UNR, bool G>
__device__ __forceinline__ void delta_sweep(const Ctx& c, const CmpcConsts& prm, int tid)
{
    const int N = c.N;
    const int r = tid & 31, half = tid >> 5, blk = r >> 2;
    float* gb = c.ybuf;        // g (32)
    float* lb = c.ybuf + 32;   // dl (32)
    float* trash = c.ybuf + 72;
    // dl-step: column r of U against g, 16 terms per half;  fp-step: row r of U (lanes 0..31) or row r of WT
    // (lanes 32..47) against dl
    unsigned loff[16], foff[8];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        // (row m = 16 half + t: both halves' constants at compile time and one select each, instead of ub_row's arithmetic on a runtime m)
        const int mq = half ? (16 + t) >> 2 : t >> 2;
        const int cm = half ? ub_row(16 + t) - 4 * ((16 + t) >> 2) : ub_row(t) - 4 * (t >> 2);
        loff[t] = blk >= mq ? cm + r : REC_ZERO;
    }
    {
        // (both halves' offsets and a bitwise blend: written as a conditional expression the eight selects became eight branchy regions of the set-up)
        const unsigned ubr = ub_row(r) - 4 * blk, wtr = REC_WT + 32 * (r & 15), r7 = r & 7;
        const unsigned hm = 0u - (unsigned)half, w16 = 0u - (unsigned)(r < 16);   // all ones: upper half / a Ws row
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned lo = q >= blk ? ubr + 4 * q : REC_ZERO;
            const unsigned hq = ((wtr + 4 * (q ^ r7)) & w16) | (REC_ZERO & ~w16);
            foff[q] = (lo & ~hm) | (hq & hm);
        }
    }
    // A^T row roles of lanes 32..46 (state index j = r): out = s v[j] + ce v[je] + cg (v[6+a1] F[a2] - v[6+a2] F[a1]) - W^T dl
    const bool hi = half == 1 && r < NS;
    const int j = hi ? r : 0;
    const int ja = j < 3 ? j : (j >= 9 ? (j - 9) % 3 : 0), ja1 = (ja + 1) % 3, ja2 = (ja + 2) % 3;
    const int jct = j >= 12 ? 1 : 0;
    const int gfo = j < 3 ? 30 : 24 + 3 * jct;    // Fsum or Fc of the foot (geometry record)
    const int je = (j >= 3 && j < 6) ? j - 3 : 0;
    const float him = hi ? 1.f : 0.f;
    const float hposm = (hi && j >= 9) ? 1.f : 0.f, hce = (hi && j >= 3 && j < 6) ? prm.dt : 0.f;
    const float hcg0 = (hi && j < 3) ? prm.dt : 0.f, hcg1 = (hi && j >= 9) ? -prm.dt : 0.f, hjsel = jct ? 1.f : 0.f;
    // B^T roles of lanes 0..29: force components (lanes 0..23) and landing offsets (24..29)
    const bool isF = tid < NF, isQ = tid >= NF && tid < NU;
    const int fi = isF ? tid : 0;
    const int fa = fi % 3, fa1 = (fa + 1) % 3, fa2 = (fa + 2) % 3, fcj = fi / 3;
    const int qq = isQ ? tid - 24 : 0, qct = qq >= 3 ? 1 : 0, qa = qq - 3 * qct;
    const float fsel = (fcj >> 2) ? 1.f : 0.f, qsel = qct ? 1.f : 0.f;
    const float lom = half == 0 ? 1.f : 0.f;
    const float Dm = prm.D[r % 3];
    const float dtc = prm.dt, mu = prm.mu_fr;
    if (tid < NXA) c.fpv[tid] = 0.f;
    if (tid < 32) { gb[tid] = 0.f; }
    wave_lds_sync();
    // operand pointers of stage N-1 (they move DOWN by a stage at a time: a trip addresses stage k - i as pointer + (UNR-1-i) strides from the trip's lowest stage)
    ldsf_t lp[16];
    ldsf_t fpp[8];
    const int kt = N - 1;
    if (!G) {
#pragma unroll
        for (int t = 0; t < 16; ++t) lp[t] = lds_opaque(c.Lf + (size_t)REC_N * kt + loff[t]);
#pragma unroll
        for (int q = 0; q < 8; ++q) fpp[q] = lds_opaque(c.Lf + (size_t)REC_N * kt + foff[q]);
    }
    ldsf_t wp = lds_opaque(c.dT + NI * kt + 4 * fcj);                       // the corner's four friction-row coefficients
    ldsf_t wqp = lds_opaque(c.dT + NI * kt + 32 + qq);                      // upper / lower (+6) row of the offset
    ldsf_t rfp = lds_opaque(c.sp + c.L.pR(fcj >> 2) + 9 * kt + fa);        // R(fa, 0..2) at +0, +3, +6
    ldsf_t rqp = lds_opaque(c.sp + c.L.pR(qct) + 9 * kt + 3 * qa);         // R(0..2, qa) at +0, +1, +2
    ldsf_t rr1 = lds_opaque(c.geoA + GEO * kt + 3 * fcj + fa1);
    ldsf_t rr2 = lds_opaque(c.geoA + GEO * kt + 3 * fcj + fa2);
    ldsf_t hf1 = lds_opaque(c.geoA + GEO * kt + gfo + ja1);
    ldsf_t hf2 = lds_opaque(c.geoA + GEO * kt + gfo + ja2);
    ldsf_t gm0 = lds_opaque(c.sp + c.L.pGam(0) + kt);
    ldsf_t gm1 = lds_opaque(c.sp + c.L.pGam(1) + kt);
    typedef __attribute__((address_space(3))) const int* ldsi_t;
    ldsi_t qmp = (ldsi_t)(c.qmask + kt);
    asm volatile("" : "+v"(qmp));
    // fixed addresses in the costate buffer
    ldsf_t vN = lds_opaque(c.fpv + NS + fi), v3 = lds_opaque(c.fpv + 3 + fa), v6a = lds_opaque(c.fpv + 6 + fa1), v6b = lds_opaque(c.fpv + 6 + fa2);
    ldsf_t vq = lds_opaque(c.fpv + 9 + 3 * qct);
    ldsf_t hvj = lds_opaque(c.fpv + j), hve = lds_opaque(c.fpv + je), hv1 = lds_opaque(c.fpv + 6 + ja1), hv2 = lds_opaque(c.fpv + 6 + ja2);
    ldsf_t gbp = lds_opaque(gb + 16 * half), lbp = lds_opaque(lb);
    ldsw_t gbs = lds_opaque_w(tid < NU ? gb + tid : trash + (tid & 7));
    ldsw_t lbs = lds_opaque_w(lb + r);
    ldsw_t fps = lds_opaque_w(half == 0 ? (tid < NF ? c.fpv + NS + tid : trash + 8 + (tid & 7)) : (hi ? c.fpv + j : trash + 16 + (tid & 7)));
    // the lq row of the record (slot 15 of the Ws rows) takes dl
    ldsw_t lqs = lds_opaque_w((!G && tid < NU) ? c.Lf + (size_t)REC_N * kt + wt_idx(15, tid) : trash + (tid & 7));
    const int lqst = (!G && tid < NU) ? REC_N : 0;
    PROF2_DECL;
    auto stage = [&](int o, int k) {      // o: strides above the trip's lowest stage
        float lm[16];
        float4 fm[8];
        if (G) {
            const RecRef<G> rec(c.Lf, N, k);
#pragma unroll
            for (int t = 0; t < 16; ++t) lm[t] = rec.ld(loff[t]);
#pragma unroll
            for (int q = 0; q < 8; ++q) fm[q] = rec.ld4(foff[q]);
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) lm[t] = lp[t][REC_N * o];
#pragma unroll
            for (int q = 0; q < 8; ++q) fm[q] = lds_ld4(fpp[q] + REC_N * o);
        }
        const float gam0 = gm0[o], gam1 = gm1[o];
        const float dgam = gam1 - gam0;
        PROF2(24);
        // ---- g = C^T w + fp_p + B^T fp_s: a force component (lanes 0..23) or a landing offset (24..29) ----
        // (both forms on every lane, on clamped indices, and a select: the wave walks through both anyway, and a branch on the lane's role costs the exec-mask
        //  bookkeeping of two regions per stage)
        float g;
        {
            const float w0 = wp[NI * o], w1 = wp[NI * o + 1], w2 = wp[NI * o + 2], w3 = wp[NI * o + 3];
            // sum_f w_f R (sx_f, sy_f, -mu)^T,  (sx, sy) = (+,+), (-,+), (-,-), (+,-)
            const float wx = w0 - w1 - w2 + w3, wy = w0 + w1 - w2 - w3, ws = w0 + w1 + w2 + w3;
            float gF = rfp[9 * o] * wx + rfp[9 * o + 3] * wy - mu * rfp[9 * o + 6] * ws;
            gF += vN[0] + dtc * fmaf(fsel, dgam, gam0) * (v3[0] + v6a[0] * rr2[GEO * o] - v6b[0] * rr1[GEO * o]);
            const float fr = (float)((qmp[o] >> qq) & 1);
            float gQ = wqp[NI * o] - wqp[NI * o + 6];
            gQ += fr * (1.f - fmaf(qsel, dgam, gam0)) * (rqp[9 * o] * vq[0] + rqp[9 * o + 1] * vq[1] + rqp[9 * o + 2] * vq[2]);
            g = isF ? gF : gQ;
        }
        *gbs = g;
        wave_lds_sync();
        PROF2(25);
        // ---- dl = L^{-1} g ----
        float4 gv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) gv[t] = lds_ld4(gbp + 4 * t);
        float la, lc;
        {
            // (packed multiply-add chains, as in forward_sweep)
            v2f lp = v2f{lm[0], lm[1]} * v2f{gv[0].x, gv[0].y}, lq = v2f{lm[2], lm[3]} * v2f{gv[0].z, gv[0].w};
#pragma unroll
            for (int t = 1; t < 4; ++t) {
                lp = __builtin_elementwise_fma(v2f{lm[4 * t], lm[4 * t + 1]}, v2f{gv[t].x, gv[t].y}, lp);
                lq = __builtin_elementwise_fma(v2f{lm[4 * t + 2], lm[4 * t + 3]}, v2f{gv[t].z, gv[t].w}, lq);
            }
            lp += lq;
            la = lp[0]; lc = lp[1];
        }
        const float dl = half_sum(la + lc);
        *lbs = dl;                      // (both halves hold the same dl)
        if (G) {
            if (tid < NU) { const RecRef<G> rec(c.Lf, N, k); const unsigned lo = wt_idx(15, tid); rec.st(lo, rec.ld(lo) + dl); }
        } else {
            *lqs = *lqs + dl;
            lqs -= lqst;
        }
        wave_lds_sync();
        PROF2(26);
        // ---- fp_s <- A^T fp_s - Ws^T dl (lanes 32..46);  fp_p <- D L^{-T} dl (lanes 0..23) ----
        float4 dv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) dv[q] = lds_ld4(lbp + 4 * q);
        float sm;
        {
            v2f sp = v2f{fm[0].x, fm[0].y} * v2f{dv[0].x, dv[0].y}, sq = v2f{fm[0].z, fm[0].w} * v2f{dv[0].z, dv[0].w};
#pragma unroll
            for (int q = 1; q < 8; ++q) {
                sp = __builtin_elementwise_fma(v2f{fm[q].x, fm[q].y}, v2f{dv[q].x, dv[q].y}, sp);
                sq = __builtin_elementwise_fma(v2f{fm[q].z, fm[q].w}, v2f{dv[q].z, dv[q].w}, sq);
            }
            sp += sq;
            sm = sp[0] + sp[1];
        }
        const float gamj = fmaf(hjsel, dgam, gam0);
        const float sj = fmaf(hposm, gamj - 1.f, 1.f);
        const float cg = fmaf(hcg1, gamj, hcg0);
        const float hiv = sj * hvj[0] + hce * hve[0] + cg * (hv1[0] * hf2[GEO * o] - hv2[0] * hf1[GEO * o]) - sm;
        const float out = lom * (k > 0 ? Dm : 0.f) * sm + him * hiv;
        wave_lds_sync();
        *fps = out;
        wave_lds_sync();
        PROF2(27);
    };
    auto bump = [&](int n) {          // n stages down
        if (!G) {
#pragma unroll
            for (int t = 0; t < 16; ++t) lp[t] -= REC_N * n;
#pragma unroll
            for (int q = 0; q < 8; ++q) fpp[q] -= REC_N * n;
        }
        wp -= NI * n; wqp -= NI * n; rfp -= 9 * n; rqp -= 9 * n; rr1 -= GEO * n; rr2 -= GEO * n; hf1 -= GEO * n; hf2 -= GEO * n;
        gm0 -= n; gm1 -= n; qmp -= n;
    };
    int k = N - 1;
    // (the stages that do not fill a trip first, one at a time)
#pragma unroll 1
    for (; (k + 1) % UNR != 0; --k) {
        stage(0, k);
        bump(1);
    }
#pragma unroll 1
    for (; k >= UNR - 1; k -= UNR) {
        bump(UNR - 1);               // to the trip's lowest stage
#pragma unroll
        for (int i = 0; i < UNR; ++i) stage(UNR - 1 - i, k - i);
        bump(1);                     // to the stage below the trip
    }
}
template <int UNR, bool G, int PART = 0>
__device__ void riccati_delta(const Ctx& c, const CmpcConsts& prm, int tid)
{
    if (tid < 64) delta_sweep<UNR, G>(c, prm, tid);
    if (PART == 0) __syncthreads();
}

// largest step lengths keeping t, z positive (fraction tau to the boundary)
template <int NT>
__device__ void step_lengths(const Ctx& c, int tid, float tau, float& ap, float& ad, int k0 = 0, int k1 = CMPC_NMAX)
{
    float a_p = 1.f, a_d = 1.f;
    const int ke = k1 < c.N ? k1 : c.N;   // rows of stages k0 .. ke-1
    for (int e = tid + NI * k0; e < ke * NI; e += NT) {
        const float dt_ = c.dT[e], dz_ = c.dZ[e];
        if (dt_ < 0.f) a_p = fminf(a_p, -tau * c.T[e] / dt_);
        if (dz_ < 0.f) a_d = fminf(a_d, -tau * c.Z[e] / dz_);
    }
    float m[2] = {-a_p, -a_d};
    block_maxn<NT, 2>(m, c.red, tid);
    ap = -m[0]; ad = -m[1];
}

// new costates (backward, wave 0), blended into LAM with step ap:
//   lam_k = gs_k + Q_k ds_k + S_k^T du_k + A_k^T lam_{k+1}
// gam-weighted sums of the force step per foot and in total (the defect array is free until the next
// iteration's residual pass): d[NS k + 3 ct + a], d[NS k + 6 + a]; nth threads share the 9 N entries
__device__ inline void costate_force_sums(const Ctx& c, int tid, int nth)
{
    const int N = c.N;
    for (int e = tid; e < 9 * N; e += nth) {
        const int k = e / 9, t = e - 9 * k;
        const float* du = c.dU + NU * k;
        const int a = t % 3;
        const float s0 = gam_of(c, 0, k) * (du[a] + du[3 + a] + du[6 + a] + du[9 + a]);
        const float s1 = gam_of(c, 1, k) * (du[12 + a] + du[15 + a] + du[18 + a] + du[21 + a]);
        c.d[NS * k + t] = t < 3 ? s0 : (t < 6 ? s1 : s0 + s1);
    }
}
// ---- The costates by scans over the stages (one wave, lane <-> stage k = 0 .. N; N <= CMPC_NMAX < 64).  The recursion
//   lam_k = base_k + sj lam_{k+1} + (terms in lam_{k+1} of OTHER components)
// is triangular in the components: the angular-momentum rows depend on nothing else (A_hh = I: a suffix sum of terms known in advance), the CoM rows and the foot rows
// take lam_h of the next stage through the cross products with the forces (a suffix sum, and one weighted by gam), the CoM-velocity rows take dt lam_com of the next stage.
// Four dependent levels of scans (DPP row shifts + one read of each row's leader) instead of N serial stages of ~45 instructions with two broadcasts each: ~500
// instructions against ~900 on a dependent chain.  Same terms as costate_recursion (which stays as the statement of the recursion and serves horizons beyond 63);
// sums are taken in scan order, so the float32 costates differ from the serial ones in the last bits. ----
template <int CTRL>
__device__ inline float dpp_keep(float old, float v) { return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(old), __float_as_uint(v), CTRL, 0xF, 0xF, false)); }
__device__ inline float lane_next(float v, int lane) { return __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (lane + 1), __float_as_int(v))); }
// v_k = sum_{m >= k} b_m over the lanes of the wave (lanes beyond the last stage hold zeros)
__device__ inline float suffix_sum(float v, int lane)
{
    v += dpp_f<0x101>(v);   // row_shl:1  (lane i takes lane i + 1 of its row of 16; zero beyond the row)
    v += dpp_f<0x102>(v);
    v += dpp_f<0x104>(v);
    v += dpp_f<0x108>(v);
    const float t1 = readlane_f(v, 16), t2 = readlane_f(v, 32), t3 = readlane_f(v, 48);
    const int row = lane >> 4;
    return v + (row == 0 ? t1 + (t2 + t3) : (row == 1 ? t2 + t3 : (row == 2 ? t3 : 0.f)));
}
// v_k = b_k + g_k v_{k+1}  (lanes beyond the last stage: b = 0, g = 1)
__device__ inline float suffix_lin(float g, float b, int lane)
{
#define CMPC_LIN_STEP(CT) { const float gn = dpp_keep<CT>(1.f, g), bn = dpp_f<CT>(b); b = fmaf(g, bn, b); g *= gn; }
    CMPC_LIN_STEP(0x101) CMPC_LIN_STEP(0x102) CMPC_LIN_STEP(0x104) CMPC_LIN_STEP(0x108)
#undef CMPC_LIN_STEP
    const int row = lane >> 4;
    const float v48 = readlane_f(b, 48);
    if (row == 2) b = fmaf(g, v48, b);
    const float v32 = readlane_f(b, 32);
    if (row == 1) b = fmaf(g, v32, b);
    const float v16 = readlane_f(b, 16);
    if (row == 0) b = fmaf(g, v16, b);
    return b;
}
template <bool DEFER>
__device__ inline void costate_scan(const Ctx& c, const CmpcConsts& prm, int lane, float ap, bool use_exact, float* vout)
{
    const int N = c.N;
    const bool valid = lane <= N, dyn = lane < N;   // dyn: a stage with dynamics (the terminal lane N carries the terminal cost only)
    const int k = valid ? lane : N, kd = dyn ? lane : N - 1;
    const float* S = c.S + NS * k;
    const float* dS = c.dS + NS * k;
    const float* geo = c.geoA + GEO * kd;
    const float* dF = c.d + NS * kd;
    const float dtm = dyn ? prm.dt : 0.f;           // (every coupling term carries dt: zero on the terminal lane and beyond)
    const float egm = use_exact ? dtm : 0.f;
    const float on = valid ? 1.f : 0.f;
    float lo[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) lo[a] = c.LAM[NS * (kd + 1) + 6 + a];   // lam_h of the next stage as the Hessian used it
    float lold[NS];
    if (!DEFER) {
#pragma unroll
        for (int j = 0; j < NS; ++j) lold[j] = c.LAM[NS * k + j];
    }
    float v[NS];
    // level 1: angular momentum
    float vh1[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float b = on * (2.f * prm.w_h) * ((S[6 + a] - c.sp[c.L.pHref() + a + 3 * k]) + dS[6 + a]);
        v[6 + a] = suffix_sum(b, lane);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) vh1[a] = lane_next(v[6 + a], lane);
    // level 2: CoM rows (total force) and foot rows (the foot's force, weighted by gam)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int a1 = (a + 1) % 3, a2 = (a + 2) % 3;
        const float w = a == 0 ? 2.f * prm.w_com0 : (a == 1 ? 2.f * prm.w_com1 : prm.wz2[k]);
        float b = w * ((S[a] - c.sp[c.L.pComref() + a + 3 * k]) + dS[a]);
        b += egm * (dF[6 + a2] * lo[a1] - dF[6 + a1] * lo[a2]);               // eg = -dt:  -dt (dF_a1 lamh_a2 - dF_a2 lamh_a1)
        b += dtm * (vh1[a1] * geo[30 + a2] - vh1[a2] * geo[30 + a1]);
        v[a] = suffix_sum(on * b, lane);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const float gam = c.sp[c.L.pGam(ct) + kd];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int a1 = (a + 1) % 3, a2 = (a + 2) % 3, j = 9 + 3 * ct + a;
            float b = (2.f * prm.w_pos) * ((S[j] - c.sp[c.L.pNom(ct) + a + 3 * k]) + dS[j]);
            b += egm * (dF[3 * ct + a1] * lo[a2] - dF[3 * ct + a2] * lo[a1]);   // eg = +dt
            b -= dtm * gam * (vh1[a1] * geo[24 + 3 * ct + a2] - vh1[a2] * geo[24 + 3 * ct + a1]);
            v[j] = suffix_lin(dyn ? gam : 1.f, on * b, lane);
        }
    }
    // level 3: CoM velocity (no cost of its own: dt lam_com of the next stage, summed)
#pragma unroll
    for (int a = 0; a < 3; ++a) v[3 + a] = suffix_sum(dtm * lane_next(v[a], lane), lane);
    if (valid && lane > 0) {   // (stage 0 has no costate anybody reads: the recursion never wrote it)
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            if (DEFER) vout[NS * k + j] = v[j];
            else c.LAM[NS * k + j] = lold[j] + ap * (v[j] - lold[j]);
        }
    }
}

#ifdef CMPC_COSTATE_SERIAL
// The serial recursion the scans replace -- the statement of what they compute (oracle/ipm_ref.c states the same), compiled only into -DCMPC_COSTATE_SERIAL builds
// (the A/B and the trace of profiles/r04_experiments_not_kept.txt, item 24).  One wave (tid < 64).  DEFER: the full-step costates go to vout (NS (N + 1) floats) and LAM is left alone -- the caller blends them in once
// the step length is known (phase_final_post: the recursion runs beside the slack steps the step length comes from)
template <bool DEFER>
__device__ inline void costate_recursion(const Ctx& c, const CmpcConsts& prm, int tid, float ap, bool use_exact, float* vout)
{
    const int N = c.N;
    {
        // lane j < 15 owns costate component j.  Everything below is one formula with per-lane coefficients:
        //   lam_j = w (s_j - ref_j + ds_j) + sj pv_j + ce pv_je + cg (pv_{6+a1} F_a2 - pv_{6+a2} F_a1)
        //           + eg (dF_a1 lamh_a2 - dF_a2 lamh_a1)                       (exact Hessian only)
        const bool own = tid < NS;
        const int j = own ? tid : 0;
        const int ja = j < 3 ? j : (j >= 9 ? (j - 9) % 3 : 0), ja1 = (ja + 1) % 3, ja2 = (ja + 2) % 3;
        const int jct = j >= 12 ? 1 : 0;
        const int roff = j < 3 ? c.L.pComref() + j : (j < 6 ? c.L.pComref() : (j < 9 ? c.L.pHref() + j - 6 : c.L.pNom(jct) + ja));
        const float wc = j == 0 ? 2.f * prm.w_com0 : (j == 1 ? 2.f * prm.w_com1 : (j < 6 ? 0.f : (j < 9 ? 2.f * prm.w_h : 2.f * prm.w_pos)));
        const int gfo = j < 3 ? 30 : 24 + 3 * jct;       // Fsum or Fc of the foot (geometry record)
        const int dfo = j < 3 ? 6 : 3 * jct;             // total or per-foot force-step sum
        const int gco = c.L.pGam(jct);
        // per-lane coefficients: sj = sA + sB gam, cg = cgA + cgB gam, ce, eg; 0/1 masks picking components a1, a2 of a 3-vector
        const float sA = j >= 9 ? 0.f : 1.f, sB = j >= 9 ? 1.f : 0.f;
        const float cgA = j < 3 ? prm.dt : 0.f, cgB = j >= 9 ? -prm.dt : 0.f;
        const float ce = (j >= 3 && j < 6) ? prm.dt : 0.f;
        const float eg = !use_exact ? 0.f : (j < 3 ? -prm.dt : (j >= 9 ? prm.dt : 0.f));
        const float m10 = ja1 == 0, m11 = ja1 == 1, m12 = ja1 == 2, m20 = ja2 == 0, m21 = ja2 == 1, m22 = ja2 == 2;
        // The recursion runs in registers.  Per stage: the linear-momentum costates of stage k+1 reach the lanes of the CoM-velocity
        // rows by a DPP row shift, its three angular-momentum costates (and, for the exact Hessian, the old ones the Hessian used)
        // are broadcast with v_readlane and meet per-lane coefficient vectors built from the cross-product operands.  Everything
        // that does not depend on the recursion (state, reference, step, forces, old costate) is fetched one stage ahead, two
        // operand sets alternating -- a stage costs ~45 VALU instructions and no LDS round trip (it was two, in float64).
        struct Ops { float s, ref, ds, F1, F2, dF1, dF2, gam, wz, lold; };
        auto fetch = [&](int k) {
            Ops o;
            const float* geo = c.geoA + GEO * k;
            o.s = c.S[NS * k + j]; o.ref = c.sp[roff + 3 * k]; o.ds = c.dS[NS * k + j];
            o.F1 = geo[gfo + ja1]; o.F2 = geo[gfo + ja2];
            o.dF1 = c.d[NS * k + dfo + ja1]; o.dF2 = c.d[NS * k + dfo + ja2];
            o.gam = c.sp[gco + k];
            o.wz = prm.wz2[k];
            o.lold = c.LAM[NS * k + j];
            return o;
        };
        float lold = c.LAM[NS * N + j];   // (lanes 6..8: lam_h,N as the Hessian used it)
        float pvj = (j == 2 ? prm.wz2[N] : wc) * ((c.S[NS * N + j] - c.sp[roff + 3 * N]) + c.dS[NS * N + j]);
        auto step = [&](int k, const Ops& o) {
            // off the recursion: base term and coefficient vectors
            const float base = (j == 2 ? o.wz : wc) * ((o.s - o.ref) + o.ds);
            const float sj = fmaf(sB, o.gam, sA), cg = fmaf(cgB, o.gam, cgA);
            const float cF1 = cg * o.F1, cF2 = cg * o.F2;
            const float c0 = m10 * cF2 - m20 * cF1, c1 = m11 * cF2 - m21 * cF1, c2 = m12 * cF2 - m22 * cF1;
            float v = base;
            if (use_exact) {
                const float eF1 = eg * o.dF1, eF2 = eg * o.dF2;
                const float g0 = m20 * eF1 - m10 * eF2, g1 = m21 * eF1 - m11 * eF2, g2 = m22 * eF1 - m12 * eF2;
                v += readlane_f(lold, 6) * g0 + readlane_f(lold, 7) * g1 + readlane_f(lold, 8) * g2;
            }
            // on it
            const float pe = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(pvj), 0x113, 0xF, 0xF, true));  // row_shr:3
            v += sj * pvj + ce * pe + readlane_f(pvj, 6) * c0 + readlane_f(pvj, 7) * c1 + readlane_f(pvj, 8) * c2;
            lold = o.lold;
            pvj = v;
            if (own) {
                if (DEFER) vout[NS * k + j] = v;
                else c.LAM[NS * k + j] = lold + ap * (v - lold);
            }
        };
        Ops oa = fetch(N > 1 ? N - 1 : 0), ob;
        if (own) {
            if (DEFER) vout[NS * N + j] = pvj;
            else c.LAM[NS * N + j] = lold + ap * (pvj - lold);
        }
        int k = N - 1;
#pragma unroll 1   // (left alone the compiler unrolls all N / 2 trips: 10 KB of straight-line code for a phase that runs once per iteration)
        for (; k >= 2; k -= 2) {
            ob = fetch(k - 1);
            step(k, oa);
            oa = fetch(k - 2);
            step(k - 1, ob);
        }
        if (k == 1) step(1, oa);
    }
}
#endif
__device__ void costate_update(const Ctx& c, const CmpcConsts& prm, int tid, float ap, bool use_exact)
{
    costate_force_sums(c, tid, blockDim.x);
    __syncthreads();
#ifdef CMPC_COSTATE_SERIAL
    if (tid < 64) costate_recursion<false>(c, prm, tid, ap, use_exact, nullptr);
#else
    if (tid < 64) costate_scan<false>(c, prm, tid, ap, use_exact, nullptr);
#endif
    __syncthreads();
}

// ---- out-of-line entry points of the sweeps ----
// The two sweeps run on wave 0.  In the HBM-factor variants (168 registers) the sweep functions use 28 / 16 callee-saved registers, which
// every wave that ENTERS the function saves to and restores from scratch memory: with all four waves calling, 576 scratch operations per
// workgroup and iteration -- 2.6 GB written and read back per 4096-problem launch, half of what the counters showed as "factor-record
// traffic" (found in round 3: one saved register per call stood out in SQ_INSTS_VMEM_WR of the resident variant).  So only wave 0 calls the
// sweep (CMPC_SWEEP_WAVE0_ONLY); the barrier and the element-wise part behind it are a function of their own that needs no callee-saved
// register.  (Inlining the sweeps into the kernel body instead was measured: 101 VGPRs of the body spilled, some inside the stage loops.)
#ifndef CMPC_SWEEP_WAVE0_ONLY
#define CMPC_SWEEP_WAVE0_ONLY(FG) (1)
#endif
#ifndef CMPC_SWEEP_WAVE
#define CMPC_SWEEP_WAVE 0   // which wave of an HBM-factor workgroup runs the sweeps (the sweep code sees tid - 64 * CMPC_SWEEP_WAVE)
#endif
template <int NT, int NC, bool FG, int PART>
__device__ __attribute__((noinline)) void phase_forward_part(lds_t lds, int Nrt, float* fg_base, bool affine_in)
{
    CMPC_PHASE_PROLOGUE;
    const bool affine = __builtin_amdgcn_readfirstlane((int)affine_in) != 0;
    riccati_forward<NT, NC == 0 ? 1 : (FG ? 2 : CMPC_SWEEP_UNROLL), FG, PART>(c, prm, PART == 1 ? tid - 64 * CMPC_SWEEP_WAVE : tid, affine, 0);
}
// the sweep alone, on its wave; what follows it -- a barrier and the slack steps -- is phase_forward_part<.., 2>, or fused with the passes behind it (phase_affine_post,
// phase_final_post)
#ifndef CMPC_FUSED_POST
#define CMPC_FUSED_POST(NC) ((NC) > 0)
#endif
template <int NT, int NC, bool FG>
__device__ __forceinline__ void phase_forward_sweep(lds_t lds, int Nrt, float* fg_base, bool affine_in)
{
    if ((threadIdx.x >> 6) == CMPC_SWEEP_WAVE) phase_forward_part<NT, NC, FG, 1>(lds, Nrt, fg_base, affine_in);
}
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) void phase_forward_tail(lds_t lds, int Nrt, float* fg_base, bool affine_in, int k0_in)
{
    CMPC_PHASE_PROLOGUE;
    const bool affine = __builtin_amdgcn_readfirstlane((int)affine_in) != 0;
    const int k0 = __builtin_amdgcn_readfirstlane(k0_in);
    riccati_forward<NT, 1, FG>(c, prm, tid, affine, k0);
}
template <int NT, int NC, bool FG, int PART>
__device__ __attribute__((noinline)) void phase_delta_part(lds_t lds, int Nrt, float* fg_base)
{
    CMPC_PHASE_PROLOGUE;
    riccati_delta<NC == 0 ? 1 : (FG ? 2 : CMPC_DELTA_UNROLL), FG, PART>(c, prm, PART == 1 ? tid - 64 * CMPC_SWEEP_WAVE : tid);
}
template <int NT, int NC, bool FG>
__device__ __forceinline__ void phase_delta(lds_t lds, int Nrt, float* fg_base)
{
    if constexpr (CMPC_SWEEP_WAVE0_ONLY(FG)) {
        if ((threadIdx.x >> 6) == CMPC_SWEEP_WAVE) phase_delta_part<NT, NC, FG, 1>(lds, Nrt, fg_base);
        __syncthreads();
    } else phase_delta_part<NT, NC, FG, 0>(lds, Nrt, fg_base);
}
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) void phase_costate(lds_t lds, int Nrt, float* fg_base, float ap, bool exact_in)
{
    CMPC_PHASE_PROLOGUE;
    const bool use_exact = __builtin_amdgcn_readfirstlane((int)exact_in) != 0;
    costate_update(c, prm, tid, ap, use_exact);
}
// ---- tail polish (out of line: rare, and the driver keeps its register allocation).  Stages k0 .. N-1 with the state entering stage
// k0 and the force before it held form a small problem of their own.  Why: the final extrapolation is exact to first order where the
// central path is smooth in mu; rows that are (nearly) degenerate -- slack and multiplier both -> 0: the friction rows of an unloaded
// corner at the apex of its pyramid -- follow sqrt(mu) and the step covers half of their distance.  That matters in the last stages
// only, whose forces the cost barely sees (no cost on the CoM velocity, nothing after them): bias sqrt(mu / curvature) = 3e-4 N/kg at
// mu = 5e-8 (N = 30: all-knot force error 8e-5, CoM velocity 2.3e-4 against the oracle; 2.6e-5 elsewhere).  So, when the
// extrapolation step of the tail is large (the symptom), the tail is re-solved: tail_iters Newton steps with per-row targets
// (row_target: conditioning as in the main loop), then its own affine-scaling step.  Stated in oracle/ipm_ref.c:tail_polish.
// Returns nothing: a factorisation that fails leaves the converged iterate of the pass before. ----
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) void tail_polish(lds_t lds, int Nrt, float* fg_base, int k0_in)
{
    CMPC_PHASE_PROLOGUE;
    const int k0 = __builtin_amdgcn_readfirstlane(k0_in);
    // tail_iters Newton steps at least; while a step was blocked (a row on its way to becoming active: the multipliers need their
    // iterations) up to six more, then the affine-scaling step
    bool last = false, blocked = true;   // (step lengths come out of block reductions: uniform)
    for (int pi = 0; pi <= prm.tail_iters + 6 && !last; ++pi) {
        last = pi >= prm.tail_iters + 6 || (pi >= prm.tail_iters && !blocked);
        all_geo<NT>(c, prm, tid);
        for (int e = tid + NS * k0; e < NS * N; e += NT) c.d[e] = (float)defect(c, prm, e / NS, e % NS);
        __syncthreads();
        const float cmu = last ? 0.f : -prm.mu_min;
        int fail = riccati_backward<NT, NC, FG>(lds, c, prm, tid, fg_base, prm.exact_hessian != 0, prm.reg, cmu, k0);
        if (fail) {
            __syncthreads();
            fail = riccati_backward<NT, NC, FG>(lds, c, prm, tid, fg_base, false, prm.reg, cmu, k0);
        }
        if (fail) break;
        if (!last) {
            for (int e = tid + NI * k0; e < NI * N; e += NT) c.dZ[e] = row_active(c, e / NI, e % NI) ? row_target(c.Z[e], prm.mu_min) : 0.f;
            __syncthreads();
        }
        phase_forward_tail<NT, NC, FG>(lds, N, fg_base, last, k0);
        float ap, ad;
        step_lengths<NT>(c, tid, last ? 0.999f : 0.99f, ap, ad, k0);
        // (a step that is not finite is not taken)
        float bad = 0.f;
        for (int e = tid + NU * k0; e < NU * N; e += NT) bad = amax_nan(bad, c.dU[e]);
        for (int e = tid + NS * (k0 + 1); e < NS * (N + 1); e += NT) bad = amax_nan(bad, c.dS[e]);
        float m1[1] = {bad};
        block_maxn<NT, 1>(m1, c.red, tid);
        if (!(m1[0] < INFINITY)) break;
        blocked = ap < 0.9f || ad < 0.9f;
        for (int e = tid + NS * (k0 + 1); e < NS * (N + 1); e += NT) c.S[e] += ap * c.dS[e];
        for (int e = tid + NU * k0; e < NU * N; e += NT) c.U[e] += ap * c.dU[e];
        if (!last)
            for (int e = tid + NI * k0; e < NI * N; e += NT) { c.T[e] += ap * c.dT[e]; c.Z[e] += ad * c.dZ[e]; }
        __syncthreads();
    }
}

// ---- the residual pass of an iteration, out of line (the driver is control flow and a handful of scalars; inlined there, this pass and the
// update pass below pushed the driver of the 168-register variants to scratch spills inside the iteration loop: -1.5 % on configs 3-5).
// Geometry, dynamics defects (float64 from the float32 iterate), inequality residuals, max t z, mean t z. ----
struct Resid { float ep, ec, mu; };
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) Resid phase_residuals(lds_t lds, int Nrt, float* fg_base, int nrow_in)
{
    CMPC_PHASE_PROLOGUE;
    const int nrow = __builtin_amdgcn_readfirstlane(nrow_in);
    float l_ep = 0.f, l_ec = 0.f, l_chk = 0.f;   // l_chk: a plain sum over everything the maxima see -- fmaxf drops a NaN operand, a sum keeps it
    double l_mu = 0.0;
    all_geo<NT>(c, prm, tid);
    // The angular-momentum defects are the heavy ones (eight corner torques in float64 each) and only 3 of a stage's 15: left inside defect() every wave walks through
    // that branch for a fifth of its lanes.  Here a corner per lane -- eight consecutive lanes form one (stage, axis) and meet through three DPP steps -- and the twelve
    // light components of a stage in a loop of their own.
    for (int e = tid; e < 24 * N; e += NT) {
        const int k = e / 24, r = e - 24 * k, a = r >> 3, cj = r & 7, ct = cj >> 2;
        const int a1 = (a + 1) % 3, a2 = (a + 2) % 3;
        const float* s = c.S + NS * k;
        const float* f = c.U + NU * k + 3 * cj;
        const float* R = c.sp + c.L.pR(ct) + 9 * k;
        const float* cn = prm.corners + 3 * cj;
        const double r1 = (double)Rm(R, a1, 0) * cn[0] + (double)Rm(R, a1, 1) * cn[1] + (double)Rm(R, a1, 2) * cn[2] + (double)s[9 + 3 * ct + a1] - (double)s[a1];
        const double r2 = (double)Rm(R, a2, 0) * cn[0] + (double)Rm(R, a2, 1) * cn[1] + (double)Rm(R, a2, 2) * cn[2] + (double)s[9 + 3 * ct + a2] - (double)s[a2];
        double t = (double)gam_of(c, ct, k) * (r1 * (double)f[a2] - r2 * (double)f[a1]);
        t += dpp_d<0xB1>(t);    // quad_perm [1,0,3,2]
        t += dpp_d<0x4E>(t);    // quad_perm [2,3,0,1]
        t += dpp_d<0x141>(t);   // row_half_mirror: the eight corners
        if (cj == 0) {
            const double dv = (double)s[6 + a] + (double)prm.dt * ((double)c.sp[c.L.pText() + 3 * k + a] + t) - (double)c.S[NS * (k + 1) + 6 + a];
            c.d[NS * k + 6 + a] = (float)dv;
            l_ep = fmaxf(l_ep, fabsf((float)dv));
            l_chk += (float)dv;
        }
    }
    for (int e = tid; e < 12 * N; e += NT) {
        const int k = e / 12, ii = e - 12 * k, i = ii < 6 ? ii : ii + 3;
        const double dv = defect(c, prm, k, i);
        c.d[NS * k + i] = (float)dv;
        l_ep = fmaxf(l_ep, fabsf((float)dv));
        l_chk += (float)dv;
    }
    if constexpr (NC > 0) {
        // (the rows a kind at a time: see row_slot)
        for_slots<0, RowSlots<NT, NC>::R>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            int e;
            float val, dot;
            const int st = row_slot<NT, NC, q, false>(c, prm, tid, e, val, dot);
            const float t = c.T[e], z = c.Z[e];
            if (st & 2) {
                const float rv = val + t;
                l_ep = fmaxf(l_ep, fabsf(rv));
                l_ec = fmaxf(l_ec, t * z);
                l_chk += rv + t * z;
                l_mu += (double)t * z;
            }
        });
    } else {
        for (int e = tid; e < NI * N; e += NT) {
            const int k = e / NI, i = e % NI;
            if (row_active(c, k, i)) {
                const float t = c.T[e], z = c.Z[e];
                const float rv = row_val(c, prm, k, i, c.U + NU * k) + t;
                l_ep = fmaxf(l_ep, fabsf(rv));
                l_ec = fmaxf(l_ec, t * z);
                l_chk += rv + t * z;
                l_mu += (double)t * z;
            }
        }
    }
    if (!(fabsf(l_chk) < INFINITY)) l_ep = INFINITY;   // (one test per thread, not per element: NaN, inf, inf - inf all end here)
    block_max2_sum<NT>(l_ep, l_ec, l_mu, c.red, c.redd, tid);
    Resid r;
    r.ep = l_ep; r.ec = l_ec; r.mu = (float)(l_mu / (double)nrow);
    return r;
}

// ---- the iterate takes its step (ap primal, ad dual) and the step is measured: out of line like the residual pass.  Returns the step
// norm of the termination test. ----
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) float phase_update(lds_t lds, int Nrt, float* fg_base, float ap_in, float ad_in)
{
    CMPC_PHASE_PROLOGUE;
    const float ap = uniform_f(ap_in), ad = uniform_f(ad_in);
    for (int e = tid; e < NS * (N + 1); e += NT) c.S[e] += ap * c.dS[e];
    for (int e = tid; e < NU * N; e += NT) c.U[e] += ap * c.dU[e];
    for (int e = tid; e < NI * N; e += NT) {
        const float tn = c.T[e] + ap * c.dT[e];
        float zn = c.Z[e] + ad * c.dZ[e];
        c.T[e] = tn; c.Z[e] = zn;
    }
    // ---- convergence: the Newton step itself is the error estimate.  Flat directions of the cost
    // (e.g. the internal force along the line joining the feet) are kept quiet by the Levenberg
    // shift `reg`.  The stationarity residual of a float32-stored iterate cannot go below ~1e-3
    // (one ulp of com_z moves its gradient by 2 w_z^2 ulp ~ 5e-3), so it is not the test. ----
    // The step is measured on what the cost and the dynamics see: the states, the deviation of
    // each corner force from its foot's mean, the force rate, the landing offsets.  A constant
    // internal force along the line joining two stance feet changes none of them (the NLP does
    // not determine it; it only drifts slowly towards the barrier's analytic centre).
    // Force steps count relative to the largest corner-force component of the iterate (the parity tolerance is
    // relative; forces are ~1-3 N/kg here), states and landing offsets absolutely (metres, m/s: order one or less).
    float l_st = 0.f, l_sf = 0.f, l_fm = 1.f, l_chk2 = 0.f;
    for (int e = tid; e < NS * (N + 1); e += NT) { const float ds = c.dS[e]; l_st = fmaxf(l_st, fabsf(ds)); l_chk2 += ds; }
    for (int e = tid; e < NU * N; e += NT) {
        const int k = e / NU, m = e % NU;
        const float du = c.dU[e];
        l_chk2 += du;
        if (m < NF) {
            const float* f = c.dU + NU * k + 12 * (m / 12) + m % 3;
            const float mean = 0.25f * (f[0] + f[3] + f[6] + f[9]);
            l_sf = fmaxf(l_sf, fabsf(du - gam_of(c, m / 12, k) * mean));
            if (k > 0) l_sf = fmaxf(l_sf, fabsf(du - c.dU[e - NU]));
            l_fm = fmaxf(l_fm, fabsf(c.U[e]));
        } else l_st = fmaxf(l_st, fabsf(du));
    }
    if (!(fabsf(l_chk2) < INFINITY)) l_st = INFINITY;   // (a step that is not finite is never "small": see the residual pass)
    float m3[3] = {l_st, l_sf, l_fm};
    block_maxn<NT, 3>(m3, c.red, tid);   // (the largest force is uniform: max(l_st, l_sf / fm) over threads = max(max l_st, max l_sf / fm))
    return ap * fmaxf(m3[0], m3[1] / m3[2]);
}

// ---- the last step of a converged solve (out of line: it runs once, and the driver keeps its register allocation): the affine-scaling
// step that phase_forward(affine) has just computed is applied -- primal only -- and, where the tail asks for it, the tail is polished.
// Returns true if the tail was polished. ----
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) bool phase_finish(lds_t lds, int Nrt, float* fg_base)
{
    CMPC_PHASE_PROLOGUE;
    float ap, ad;
    step_lengths<NT>(c, tid, 0.999f, ap, ad);
    // size of the step: over everything (NaN-aware: a step that is not finite is not taken -- the converged iterate stands) and over
    // the forces of the last tail_stages stages, relative to the largest force
    const int kt = N - prm.tail_stages;
    float l_all = 0.f, l_tail = 0.f, l_fm = 1.f;
    for (int e = tid; e < NS * (N + 1); e += NT) l_all = amax_nan(l_all, c.dS[e]);
    for (int e = tid; e < NU * N; e += NT) {
        const float du = c.dU[e];
        l_all = amax_nan(l_all, du);
        if (e % NU < NF) {
            l_fm = fmaxf(l_fm, fabsf(c.U[e]));
            if (e >= NU * kt) l_tail = fmaxf(l_tail, fabsf(du));
        }
    }
    float m3[3] = {l_all, l_tail, l_fm};
    block_maxn<NT, 3>(m3, c.red, tid);
    if (!(m3[0] < INFINITY)) return false;
    // A large extrapolation step in the tail: nearly degenerate rows there (see tail_polish).  The FULL step is measured, not ap x step:
    // such a row is exactly what blocks the step length, and a blocked extrapolation leaves the whole bias in place (measured: the
    // worst problem of 512 had ap = 0.002 and was missed by the scaled test).
    const int k0 = (prm.tail_stages > 0 && m3[1] > prm.tail_trigger * m3[2]) ? kt : N;
    // the tail then goes its own way, and the step of the stages before it is limited by their own rows only (with the common step
    // length the 27 stages that had nothing wrong got no extrapolation at all on such a problem)
    if (k0 < N) step_lengths<NT>(c, tid, 0.999f, ap, ad, 0, k0);
    for (int e = tid; e < NS * (k0 < N ? k0 + 1 : N + 1); e += NT) c.S[e] += ap * c.dS[e];
    for (int e = tid; e < NU * k0; e += NT) c.U[e] += ap * c.dU[e];
    __syncthreads();
    if (k0 == N) return false;
    tail_polish<NT, NC, FG>(lds, N, fg_base, k0);
    return true;
}

// ---- the remaining pieces of the driver, out of line for the same reason as the passes above: the kernel body below is control flow
// over a dozen scalars and holds no LDS map of its own (with the ~45 pointers of Ctx live across every phase call the 168-register
// variants spilled SGPRs into VGPR lanes and those VGPRs into scratch memory, reloaded in front of every use: -1.7 % on configs 3-5) ----
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) void phase_setup(lds_t lds, int Nrt, float* fg_base, const float* gp)
{
    CMPC_PHASE_PROLOGUE;
    float* spw = const_cast<float*>(c.sp);
    // one-off tables, parameter vector (coalesced)
    for (int e = tid; e < NTRI; e += NT) {
        int i = (int)((sqrtf(8.f * (float)e + 1.f) - 1.f) * 0.5f);
        while ((i + 1) * (i + 2) / 2 <= e) ++i;
        while (i * (i + 1) / 2 > e) --i;
        c.tri[e] = (unsigned short)((i << 8) | (e - i * (i + 1) / 2));
    }
    for (int e = tid; e < ID_STRIPS; e += NT) c.idstrip[e] = (e % ID_STRIP_LEN) == 32 + e / ID_STRIP_LEN ? 1.f : 0.f;
    // (zero blocks of the factor records: LDS was zeroed by the kernel; HBM scratch is zeroed once, at cmpc_create)
    for (int e = tid; e < c.L.np(); e += NT) spw[e] = gp[e];
    __syncthreads();
    if (tid < N) {
        int m = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) m |= qfree_compute(c, tid, q) ? (1 << q) : 0;
        c.qmask[tid] = m;
    }
    __syncthreads();
}

// initial iterate from x0 (or, cold: the cold start of SURVEY 8d built in place: CoM at com0, feet at nominal, f_z = g/8); returns the
// number of inequality rows in use
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) int phase_init(lds_t lds, int Nrt, float* fg_base, const float* x0, const float* dprev, int flags, float mu_init,
                                                    float t_floor)
{
    CMPC_PHASE_PROLOGUE;
    const bool cold = (flags & 1) != 0, use_duals = (flags & 2) != 0, use_mult = (flags & 4) != 0;
    for (int e = tid; e < NS * (N + 1); e += NT) {
        const int k = e / NS, i = e % NS;
        float v;
        if (k == 0) {  // initial-condition rows of g hold exactly
            if (i < 9) v = c.sp[c.L.pCom0() + i];
            else v = c.sp[c.L.pCur((i - 9) / 3) + (i - 9) % 3];
        } else if (cold) v = i < 3 ? c.sp[c.L.pCom0() + i] : (i < 9 ? 0.f : c.sp[c.L.pNom((i - 9) / 3) + 3 * k + (i - 9) % 3]);
        else if (i < 9) v = x0[c.L.oCom() + 3 * (N + 1) * (i / 3) + 3 * k + i % 3];
        else v = x0[c.L.oPos((i - 9) / 3) + 3 * k + (i - 9) % 3];
        c.S[e] = v;
        // costates: zero, or -- warm start with duals -- the previous solve's shifted by one knot (they only enter the
        // first iteration's exact-Hessian term; every iteration recomputes them)
        c.LAM[e] = (use_duals && !cold) ? dprev[NS * (k < N ? k + 1 : N) + i] : 0.f;
    }
    for (int e = tid; e < NU * N; e += NT) {
        const int k = e / NU, m = e % NU;
        float v = 0.f;
        if (m < NF) v = cold ? (m % 3 == 2 ? 0.125f * prm.grav : 0.f) : x0[c.L.oF(m / 12, (m % 12) / 3) + 3 * k + m % 3];
        else {
            const int q = m - 24, ct = q / 3, i = q % 3;
            if (qfree(c, k, q)) {
                const float* R = c.sp + c.L.pR(ct) + 9 * k;
                const float lo = qlo(c, k, q), hi = qhi(c, k, q), push = 0.01f * (hi - lo);
                for (int a = 0; a < 3 && !cold; ++a)
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
            if (i < 32) t = fmaxf(t, t_floor);
            z = mu_init / t;
            if (use_mult && !cold) {
                // multiplier of the same row one knot later in the previous solve, where it is the larger (an active
                // row keeps its multiplier; inactive ones keep the centred value mu / t)
                const int kk = k + 1 < N ? k + 1 : N - 1;
                z = fmaxf(z, dprev[NS * (N + 1) + NI * N + NI * kk + i]);
            }
        }
        c.T[e] = t; c.Z[e] = z;
    }
    __syncthreads();
    int nrow = 0;
    for (int k = 0; k < N; ++k)
        for (int i = 32; i < NI; ++i) nrow += row_active(c, k, i) ? 1 : 0;
    return __builtin_amdgcn_readfirstlane(nrow + 32 * N);
}

// mode 0: every multiplier scaled by `a` (cold start: the initial barrier parameter follows the initial infeasibility).
// mode 1: emergency re-centring -- multipliers with t z > 10 a pulled back to 10 a / t; returns the new mean t z.
// mode 2: every row's complementarity target set to `a` (plain centring step).
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) float phase_multipliers(lds_t lds, int Nrt, float* fg_base, int mode_in, float a_in, int nrow_in)
{
    CMPC_PHASE_PROLOGUE;
    const int mode = __builtin_amdgcn_readfirstlane(mode_in), nrow = __builtin_amdgcn_readfirstlane(nrow_in);
    const float a = uniform_f(a_in);
    float out = a;
    if (mode == 0) {
        for (int e = tid; e < NI * N; e += NT) c.Z[e] *= a;
    } else if (mode == 1) {
        double l2 = 0.0;
        for (int e = tid; e < NI * N; e += NT)
            if (row_active(c, e / NI, e % NI)) {
                const float t = c.T[e];
                float z = c.Z[e];
                if (t * z > 10.f * a) { z = 10.f * a / t; c.Z[e] = z; }
                l2 += (double)t * z;
            }
        out = (float)(block_sum<NT>(l2, c.redd, tid) / (double)nrow);
    } else {
        for (int e = tid; e < NI * N; e += NT) c.dZ[e] = row_active(c, e / NI, e % NI) ? a : 0.f;
    }
    __syncthreads();
    return out;
}

template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) int phase_backward(lds_t lds, int Nrt, float* fg_base, bool exact_in, float reg, float cmu)
{
    CMPC_PHASE_PROLOGUE;
    const bool exact = __builtin_amdgcn_readfirstlane((int)exact_in) != 0;
#ifdef CMPC_PROFILE
    const long long pb_ = __builtin_amdgcn_s_memtime();
    const int rb_ = riccati_backward<NT, NC, FG>(lds, c, prm, tid, fg_base, exact, reg, cmu);
    if (tid == 0 && blockIdx.x == 0) { g_prof[31] += 1; g_prof[63] += __builtin_amdgcn_s_memtime() - pb_; }   // backward passes of workgroup 0 and their cycles
    return rb_;
#endif
    return riccati_backward<NT, NC, FG>(lds, c, prm, tid, fg_base, exact, reg, cmu);
}

struct StepLen { float ap, ad; };
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) StepLen phase_step_lengths(lds_t lds, int Nrt, float* fg_base, float tau)
{
    CMPC_PHASE_PROLOGUE;
    StepLen r;
    step_lengths<NT>(c, tid, uniform_f(tau), r.ap, r.ad);
    return r;
}

// Mehrotra's centring parameter from the affine step (step lengths ap, ad), then the corrector's per-row complementarity targets
struct Centre { float sigma, mu_t; };
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) Centre phase_corrector_targets(lds_t lds, int Nrt, float* fg_base, float ap_in, float ad_in, float mu_in, int nrow_in)
{
    CMPC_PHASE_PROLOGUE;
    const float ap = uniform_f(ap_in), ad = uniform_f(ad_in), mu_cur = uniform_f(mu_in);
    const int nrow = __builtin_amdgcn_readfirstlane(nrow_in);
    double l_aff = 0.0;
    for (int e = tid; e < NI * N; e += NT)
        if (row_active(c, e / NI, e % NI)) l_aff += (double)(c.T[e] + ap * c.dT[e]) * (double)(c.Z[e] + ad * c.dZ[e]);
    const float mu_aff = (float)(block_sum<NT>(l_aff, c.redd, tid) / (double)nrow);
    Centre r;
    r.sigma = mu_aff / mu_cur;
    r.sigma = r.sigma * r.sigma * r.sigma;
    r.mu_t = fmaxf(fmaxf(r.sigma, prm.sigma_min) * mu_cur, prm.mu_min);
    for (int e = tid; e < NI * N; e += NT) {
        const float cmu = row_active(c, e / NI, e % NI) ? r.mu_t - c.dT[e] * c.dZ[e] : 0.f;  // complementarity target
        c.dZ[e] = cmu;
        c.dT[e] = cmu / c.T[e];   // row coefficient change, read by the corrector sweep (dT is rebuilt by the forward sweep)
    }
    __syncthreads();
    return r;
}

// ---- The element-wise passes between the sweeps, fused (compile-time horizons; the runtime-N variants keep the separate passes above).  Each of the small passes was a
// call, a rebuilt LDS map, one or two block reductions (two barriers each) and a round trip of its operands through LDS: 4.6 k cycles for the step lengths and the
// centring parameter of a 20-stage problem, 15 k for slack steps + step lengths + costates + update, of a 244 k-cycle iteration.  Here a thread keeps the rows it owns
// (t, z, dt, dz) in registers from the slack step to the last use.
// phase_affine_post: behind the affine forward sweep -- slack and multiplier steps, step lengths to the boundary, Mehrotra's centring parameter, the corrector's per-row
// targets (dZ) and row coefficient changes (dT): what riccati_forward's element-wise part, phase_step_lengths(1) and phase_corrector_targets did, same arithmetic. ----
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) Centre phase_affine_post(lds_t lds, int Nrt, float* fg_base, float mu_in, int nrow_in)
{
    CMPC_PHASE_PROLOGUE;
    typedef RowSlots<NT, NC> RS;
    constexpr int R = RS::R;
    const float mu_cur = uniform_f(mu_in);
    const int nrow = __builtin_amdgcn_readfirstlane(nrow_in);
    __syncthreads();   // (the sweep, one wave, is complete)
    float t[R], z[R], dt[R], dz[R];
    int ei[R], st[R];
    float a_p = 1.f, a_d = 1.f;
    for_slots<0, R>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        float val, dot;
        st[q] = row_slot<NT, NC, q, true>(c, prm, tid, ei[q], val, dot);
        t[q] = c.T[ei[q]]; z[q] = c.Z[ei[q]];
        const float r = val + t[q];
        dt[q] = -r - dot;
        dz[q] = (0.f - z[q] * t[q]) / t[q] - (z[q] / t[q]) * dt[q];
        if (st[q] & 2) {
            if (dt[q] < 0.f) a_p = fminf(a_p, -t[q] / dt[q]);
            if (dz[q] < 0.f) a_d = fminf(a_d, -z[q] / dz[q]);
        }
    });
    float m[2] = {-a_p, -a_d};
    block_maxn<NT, 2>(m, c.red, tid);
    const float ap = -m[0], ad = -m[1];
    double l_aff = 0.0;
#pragma unroll
    for (int q = 0; q < R; ++q)
        if (st[q] & 2) l_aff += (double)(t[q] + ap * dt[q]) * (double)(z[q] + ad * dz[q]);
    const float mu_aff = (float)(block_sum<NT>(l_aff, c.redd, tid) / (double)nrow);
    Centre r;
    r.sigma = mu_aff / mu_cur;
    r.sigma = r.sigma * r.sigma * r.sigma;
    r.mu_t = fmaxf(fmaxf(r.sigma, prm.sigma_min) * mu_cur, prm.mu_min);
#pragma unroll
    for (int q = 0; q < R; ++q) {
        if (st[q] & 1) {
            const float cmu = (st[q] & 2) ? r.mu_t - dt[q] * dz[q] : 0.f;   // complementarity target
            c.dZ[ei[q]] = cmu;
            c.dT[ei[q]] = (st[q] & 2) ? cmu / t[q] : 0.f;   // row coefficient change, read by the corrector sweep
        }
    }
    __syncthreads();
    return r;
}

// phase_final_post: behind the last forward sweep of an iteration.  The costate recursion (one wave, serial over the stages: the longest pole of these passes) runs
// BESIDE the slack steps and step lengths of the other waves; it leaves the full-step costates in the dT array -- dead by now: the rows' steps stay in registers -- and
// every thread blends them in once the step length is known.  Then the iterate takes its step and the step is measured (phase_update's arithmetic and thread mapping).
struct FinalStep { float ap, ad, step; };
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) FinalStep phase_final_post(lds_t lds, int Nrt, float* fg_base, float tau_in, bool exact_in)
{
    CMPC_PHASE_PROLOGUE;
    constexpr int NTR = NT - 64;   // threads on the rows
    typedef RowSlots<NTR, NC> RS;
    constexpr int R = RS::R;
    const float tau = uniform_f(tau_in);
    const bool use_exact = __builtin_amdgcn_readfirstlane((int)exact_in) != 0;
    float* vbuf = c.dT;
    __syncthreads();   // (the sweep, one wave, is complete)
    float t[R], z[R], dt[R], dz[R];
    int ei[R], st[R];
    float a_p = 1.f, a_d = 1.f;
    if (tid < 64) {
        costate_force_sums(c, tid, 64);
        wave_lds_sync();
#ifdef CMPC_COSTATE_SERIAL
        costate_recursion<true>(c, prm, tid, 0.f, use_exact, vbuf);
#else
        costate_scan<true>(c, prm, tid, 0.f, use_exact, vbuf);
#endif
#pragma unroll
        for (int q = 0; q < R; ++q) { t[q] = 1.f; z[q] = 0.f; dt[q] = 0.f; dz[q] = 0.f; ei[q] = 0; st[q] = 0; }
    } else {
        for_slots<0, R>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            float val, dot;
            st[q] = row_slot<NTR, NC, q, true>(c, prm, tid - 64, ei[q], val, dot);
            t[q] = c.T[ei[q]]; z[q] = c.Z[ei[q]];
            const float cmu = c.dZ[ei[q]];
            const float r = val + t[q];
            dt[q] = -r - dot;
            dz[q] = (cmu - z[q] * t[q]) / t[q] - (z[q] / t[q]) * dt[q];
            if (st[q] & 2) {
                if (dt[q] < 0.f) a_p = fminf(a_p, -tau * t[q] / dt[q]);
                if (dz[q] < 0.f) a_d = fminf(a_d, -tau * z[q] / dz[q]);
            }
        });
    }
    float m[2] = {-a_p, -a_d};
    block_maxn<NT, 2>(m, c.red, tid);   // (its barriers are also what makes the costates in vbuf visible)
    const float ap = -m[0], ad = -m[1];
#pragma unroll
    for (int q = 0; q < R; ++q)
        if (st[q] & 2) {
            c.T[ei[q]] = t[q] + ap * dt[q];
            c.Z[ei[q]] = z[q] + ad * dz[q];
        }
    for (int e = tid; e < NS * (N + 1); e += NT) {
        const float lold = c.LAM[e];
        c.LAM[e] = lold + ap * (vbuf[e] - lold);
        c.S[e] += ap * c.dS[e];
    }
    for (int e = tid; e < NU * N; e += NT) c.U[e] += ap * c.dU[e];
    // the step norm of the termination test: see phase_update
    float l_st = 0.f, l_sf = 0.f, l_fm = 1.f, l_chk2 = 0.f;
    for (int e = tid; e < NS * (N + 1); e += NT) { const float ds = c.dS[e]; l_st = fmaxf(l_st, fabsf(ds)); l_chk2 += ds; }
    for (int e = tid; e < NU * N; e += NT) {
        const int k = e / NU, mm = e % NU;
        const float du = c.dU[e];
        l_chk2 += du;
        if (mm < NF) {
            const float* f = c.dU + NU * k + 12 * (mm / 12) + mm % 3;
            const float mean = 0.25f * (f[0] + f[3] + f[6] + f[9]);
            l_sf = fmaxf(l_sf, fabsf(du - gam_of(c, mm / 12, k) * mean));
            if (k > 0) l_sf = fmaxf(l_sf, fabsf(du - c.dU[e - NU]));
            l_fm = fmaxf(l_fm, fabsf(c.U[e]));
        } else l_st = fmaxf(l_st, fabsf(du));
    }
    if (!(fabsf(l_chk2) < INFINITY)) l_st = INFINITY;
    float m3[3] = {l_st, l_sf, l_fm};
    block_maxn<NT, 3>(m3, c.red, tid);
    FinalStep r;
    r.ap = ap; r.ad = ad; r.step = ap * fmaxf(m3[0], m3[1] / m3[2]);
    return r;
}

// x in the reference layout (and, warm starts with duals, the costates, slacks and multipliers)
template <int NT, int NC, bool FG>
__device__ __attribute__((noinline)) void phase_export(lds_t lds, int Nrt, float* fg_base, float* x, float* dq)
{
    CMPC_PHASE_PROLOGUE;
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
    if (dq) {
        __syncthreads();   // (the warm-start reads of this block's own record are long done; other blocks own other rows)
        for (int e = tid; e < NS * (N + 1); e += NT) dq[e] = c.LAM[e];
        for (int e = tid; e < NI * N; e += NT) { dq[NS * (N + 1) + e] = c.T[e]; dq[NS * (N + 1) + NI * N + e] = c.Z[e]; }
    }
#ifdef CMPC_PROFILE
    if (prm.hwid_probe) {     // developer probe: where the hardware put each wave (overwrites x[0..7]; HW_ID: wave slot
        __syncthreads();      // [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13])
        if ((threadIdx.x & 63) == 0) x[threadIdx.x >> 6] = (float)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffff);
    }
#endif
}

// NC > 0: horizon known at compile time (every LDS offset becomes an immediate); NC == 0: runtime N
// FG: the per-stage factors (Linv, Ws: 915 floats per stage) live in global scratch instead of LDS
// (horizons whose LDS image would exceed 160 KiB)
template <int NT, int NC, bool FG>
__global__ __launch_bounds__(NT, FG ? 3 : 1) void cmpc_solve_kernel(CmpcParams kp)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int N = NC > 0 ? NC : kp.N;
    // The whole LDS image starts at zero.  LDS arrives with whatever the previous workgroup -- or the previous
    // kernel -- left in it, and several formulas read entries under a zero weight (the E^T P rows of G before the
    // first value function exists, zero blocks of the factor records): 0 x NaN is NaN.
    {
        int* z = reinterpret_cast<int*>(smem);
        for (int e = tid; e < kp.lds_words; e += NT) z[e] = 0;
    }
    __syncthreads();
    // constants first in LDS
    {
        const int* src = reinterpret_cast<const int*>(kp.kc);
        int* dst = reinterpret_cast<int*>(smem);
        for (int e = tid; e < (int)(sizeof(CmpcConsts) / 4); e += NT) dst[e] = src[e];
    }
    const CmpcConsts& prm = *reinterpret_cast<const CmpcConsts*>(smem);
    const long long t_start = __builtin_amdgcn_s_memtime();
    // (made opaque: interprocedural constant propagation would otherwise hand every phase the dynamic-LDS symbol
    // back and with it a table lookup per call)
    unsigned ldsv = (unsigned)(unsigned long long)(lds_t)smem;
    asm volatile("" : "+s"(ldsv));
    const lds_t lds = (lds_t)(unsigned long long)ldsv;
    float* const fg_base = FG ? kp.scratch + (size_t)b * kp.scratch_stride : nullptr;
    const CmpcIdx L{N};
    __syncthreads();
    phase_setup<NT, NC, FG>(lds, N, fg_base, kp.P + (size_t)b * L.np());
    // Two passes at most: a warm-started solve (shifted previous solution) that exhausts its iteration budget is started
    // again from the cold start -- rare (a landing or lift-off tick, 1 in ~60000 solves of a walking roll-out) and cheaper than
    // failing the tick, which is all the caller could do (CentroidalMPCBlock.cpp:615-619 aborts).
    int status = 1, sg = 0, it_total = 0;   // sg: the safeguard word of info[3] (include/cmpc.h)
    float err = 0.f, ep = 0.f, mu_cur = 0.f, step_out = 0.f, step_prev = 0.f;
    float* const dq = kp.duals ? kp.duals + (size_t)b * (NS * (N + 1) + 2 * NI * N) : nullptr;
    for (int pass = 0; pass < 2; ++pass) {
        const float mu_init = pass ? 0.1f : kp.mu_init, t_floor = pass ? 1e-2f : kp.t_floor, mu_adapt = pass ? 3.5f : kp.mu_adapt;
        step_out = step_prev = 0.f;
        const bool use_duals = kp.warm && kp.warm_duals && kp.duals;
        const int nrow = phase_init<NT, NC, FG>(lds, N, fg_base, kp.X0 + (size_t)b * L.nx(), dq,
                                                (pass > 0 ? 1 : 0) | (use_duals ? 2 : 0) | (use_duals && kp.warm_duals > 1 ? 4 : 0), mu_init, t_floor);
        int it = 0;
        status = 1;
        bool finishing = false;
        // (a warm-started pass may be given a smaller budget: kp.warm_budget -- its stragglers are then re-solved from the cold
        // start, inside this kernel or by the caller, see CmpcParams)
        const int budget = (kp.warm && pass == 0 && kp.warm_budget > 0 && kp.warm_budget < prm.max_iter) ? kp.warm_budget : prm.max_iter;
        for (it = 0; it < budget + 1; ++it) {
            if (it == budget && !finishing) break;
            // ---- residuals of the current iterate ----
            PROF_DECL;
            const Resid rs = phase_residuals<NT, NC, FG>(lds, N, fg_base, nrow);
            ep = rs.ep;
            const float ec = rs.ec;
            mu_cur = rs.mu;
            // a residual that is not finite (NaN or inf in P or X0, or a step that went wrong): "invalid number" -- the reference's
            // IPOPT stops there and advance() returns false.  (phase_residuals turns NaN into +inf; plain fmaxf would drop it, the
            // iterate would read as converged and NaN forces would go downstream with status 0.)
            if (!(ep < INFINITY)) { status = 2; err = INFINITY; break; }
            if (it == 0 && mu_adapt > 0.f) {
                // cold start: the initial barrier parameter scales with the squared initial infeasibility (z = mu / t)
                const float mu0 = fminf(fmaxf(mu_adapt * ep * ep, 0.03f), 0.5f);
                phase_multipliers<NT, NC, FG>(lds, N, fg_base, 0, mu0 / mu_cur, nrow);
                mu_cur = mu0;
            }
            if (kp.warm && pass == 0 && it > 0 && ec > 100.f * mu_cur && ec > 0.1f) {
                // Emergency re-centring.  Primal and dual step lengths differ; when a blocked primal step (ap ~ 0.2) meets a
                // full dual step, the multipliers of rows that were about to become active grow while their slacks stay:
                // products t z hundreds of times the average, and the predictor-corrector can fall into a two-cycle (seen on
                // warm starts when a new swing phase enters the last stage of the horizon: max t z / mu ~ 600, 40 iterations
                // without progress).  Pull those multipliers back to 10 mu / t.  Only products of order one count (ec > 0.1):
                // near the barrier floor a few lagging rows are legitimately 100 x the mean and must be left alone
                // (re-centring them there cost config 2 five of 4096 problems).  Warm starts only: cold starts pass through such
                // iterates on their own (5 of 40960 config-2 problems trip the test and then need 20 iterations instead of 12).
                mu_cur = phase_multipliers<NT, NC, FG>(lds, N, fg_base, 1, mu_cur, nrow);
                sg += 100;
            }
            PROF(10);
            // The step that produced this iterate was already below the step tolerance and its residuals are converged:
            // stop here, before paying for a factorisation whose step would only confirm it (the error of the iterate is
            // the size of that unneeded step, an order of magnitude or more below the last one taken).
            // Its error is what the steps still to come would add: at most the last step, and -- once two steps are known --
            // their geometric tail s rho / (1 - rho) with the observed contraction rho = s_k / s_{k-1}, taken no smaller
            // than 0.2 (the convergence is superlinear only at the very end; floors of 0.1 ... 0.3 give the same worst parity error) and no larger than 0.9.
            float est = step_out;
            if (it > 1 && step_prev > 0.f) {
                const float rho = fminf(fmaxf(step_out / step_prev, 0.2f), 0.9f);
                est = fminf(est, step_out * rho / (1.f - rho));
            }
            if (it > 0 && !finishing && fmaxf(ep, ec) <= prm.tol && est <= prm.step_tol) {
                err = fmaxf(ep, ec);
                status = 0;
                if (!prm.final_extrap) break;
                finishing = true;
            }
            // ---- predictor (affine scaling): factorise; on a non-positive pivot fall back to the
            // Gauss-Newton Hessian, then to a larger Levenberg shift ----
            // once the complementarity products sit at the barrier floor the predictor has nothing to predict
            // (sigma ~ 0, second-order term ~ 0): take plain centring Newton steps, one sweep pair instead of three
            const bool centring = !finishing && mu_cur <= fmaxf(1.5f * prm.mu_min, 0.15f * prm.tol) && ec <= fmaxf(4.f * prm.mu_min, 0.4f * prm.tol);
            bool exact = prm.exact_hessian != 0;
            float reg = prm.reg;
            int fail = 1;
            bool resync = false;
            for (int attempt = 0; attempt < 4; ++attempt) {
                fail = phase_backward<NT, NC, FG>(lds, N, fg_base, exact, reg, centring ? prm.mu_min : 0.f);
                if (!fail) break;
                __syncthreads();
                if (fail & 2) {
                    // a wave of the streaming stage gave up at a hand-off word (never seen; a protocol bug would look like this): counted in its own digit of
                    // info[3], and the same pass is tried once more as it was -- it is not a bad pivot and must not change the algorithm
                    sg += 1000000;
                    if (!resync) { resync = true; --attempt; continue; }
                }
                ++sg; exact = false;
                if (attempt > 0) reg *= 1e3f;
            }
            if (fail) { if (!finishing) status = 2; break; }   // (a failed extrapolation step leaves the converged iterate)
            PROF(11);
            float sigma = 0.f, mu_t = prm.mu_min;
            if (centring) {
                phase_multipliers<NT, NC, FG>(lds, N, fg_base, 2, prm.mu_min, nrow);
            } else {
                phase_forward_sweep<NT, NC, FG>(lds, N, fg_base, true);
                PROF(12);
                if (finishing) {
                    // last step: affine-scaling extrapolation of the central path to mu = 0 (primal only), and the tail polish behind it
                    phase_forward_part<NT, NC, FG, 2>(lds, N, fg_base, true);
                    if (phase_finish<NT, NC, FG>(lds, N, fg_base)) sg += 100000;
                    ++it;
                    break;
                }
                Centre ce;
                if constexpr (CMPC_FUSED_POST(NC)) ce = phase_affine_post<NT, NC, FG>(lds, N, fg_base, mu_cur, nrow);
                else {
                    phase_forward_part<NT, NC, FG, 2>(lds, N, fg_base, true);
                    const StepLen sa = phase_step_lengths<NT, NC, FG>(lds, N, fg_base, 1.f);
                    ce = phase_corrector_targets<NT, NC, FG>(lds, N, fg_base, sa.ap, sa.ad, mu_cur, nrow);
                }
                sigma = ce.sigma; mu_t = ce.mu_t;
                PROF(13);
                phase_delta<NT, NC, FG>(lds, N, fg_base);
                PROF(14);
            }
            phase_forward_sweep<NT, NC, FG>(lds, N, fg_base, false);   // (one call site for both branches: the sweep may be inlined here)
            PROF(15);
            // ---- slack steps, step lengths, costates, then the iterate ----
            float ap, ad, step;
            if constexpr (CMPC_FUSED_POST(NC)) {
                const FinalStep fs = phase_final_post<NT, NC, FG>(lds, N, fg_base, fmaxf(0.99f, 1.f - mu_t), exact);
                ap = fs.ap; ad = fs.ad; step = fs.step;
                PROF(16);
            } else {
                phase_forward_part<NT, NC, FG, 2>(lds, N, fg_base, false);
                const StepLen sl = phase_step_lengths<NT, NC, FG>(lds, N, fg_base, fmaxf(0.99f, 1.f - mu_t));
                ap = sl.ap; ad = sl.ad;
                phase_costate<NT, NC, FG>(lds, N, fg_base, ap, exact);
                PROF(16);
                step = phase_update<NT, NC, FG>(lds, N, fg_base, ap, ad);
            }
            step_prev = step_out;
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
        it_total += it;
        if (status == 0 || !kp.warm || pass > 0 || kp.warm_no_restart) break;
        sg += 10000;
        __syncthreads();
    }
    phase_export<NT, NC, FG>(lds, N, fg_base, kp.X + (size_t)b * L.nx(), dq);
    if (kp.info && tid == 0) {
        float* inf = kp.info + (size_t)b * CMPC_INFO_N;
        inf[0] = (float)it_total; inf[1] = err; inf[2] = mu_cur; inf[3] = (float)sg; inf[4] = ep; inf[5] = (float)status;
        inf[6] = (float)(__builtin_amdgcn_s_memtime() - t_start); inf[7] = step_out;
    }
}

}  // namespace

// Workgroup barriers one role of the streaming stage executes in a backward pass over stages N-1 .. k0 (role 0: the factorising wave, sq_factor_loop; 1: the
// consumers, sq_consume_loop), counted on the loop skeleton both device loops are written with.  A mismatch would be a hang of the whole workgroup on the GPU:
// tests/test_host_logic.py holds the two together.
extern "C" int cmpc_sq_pass_barriers(int N, int k0, int role)
{
    int n = SQ_PRE_BARRIERS;
    SQ_STAGE_LOOP(k, N, k0) { (void)role; ++n; }
    return n == sq_pass_barriers(N, k0) ? n : -1;
}

// LDS bytes the kernel needs for horizon N
extern "C" size_t cmpc_solver_lds_bytes(int N, int factors_global)
{
    CmpcLayout L;
    cmpc_layout_init(L, N);
    const size_t dbl = 90 + 40 + 40 + 16 + 40 + 4 * NI + 8 + (factors_global ? 0 : 90 + 16);
    const size_t work = factors_global ? ((NXA * PLD + 3) & ~3) + NS * 16 + ((NXA * GLD + 3) & ~3) : (size_t)MSET + NS * 16 + SQ_PUB_FLOATS + 128;   // (see make_ctx)
    const size_t flt = (size_t)NU * RLD + (size_t)NPAN * RLD + 96 + ID_STRIPS + ((L.np + 3) & ~3)
                       + ((size_t)NS * (N + 1) + (size_t)NU * N + ((factors_global && N > CMPC_TZ_LDS_NMAX) ? 0 : 2 * (size_t)NI * N)) + (size_t)NS * N + (size_t)NS * (N + 1)
                       + (size_t)GEO * N + work + 2 * DSET_F + 40 + 40 + 24
                       + 2 * DSET_I + 4 + CMPC_NMAX + NTRI / 2 + (factors_global ? 0 : (size_t)REC_N * N);
    return ((sizeof(CmpcConsts) + 15) & ~(size_t)15) + dbl * 8 + flt * 4;
}

#ifdef CMPC_PROFILE
extern "C" int cmpc_profile_read(long long* out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(long long) * 128);
    if (reset) { long long z[128] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)); }
    return (int)e;
}
extern "C" int cmpc_trace_read(float* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(float) * 64 * 8); }
#endif

// the kernel variant for a horizon and a factor storage: resident (512 threads, records in LDS, one workgroup per CU) or HBM-factor (256 threads, three per CU)
typedef void (*cmpc_kernel_t)(CmpcParams);
static cmpc_kernel_t solver_variant(int N, bool factors_global, int& nthreads)
{
    constexpr int NT = 256, NR = 512;
    if (factors_global) {
        nthreads = NT;
        return N == 20 ? cmpc_solve_kernel<NT, 20, true> : (N == 30 ? cmpc_solve_kernel<NT, 30, true> : cmpc_solve_kernel<NT, 0, true>);
    }
    nthreads = NR;
    switch (N) {  // horizons of the shipped configurations get compile-time layouts
        case 10: return cmpc_solve_kernel<NR, 10, false>;
        case 12: return cmpc_solve_kernel<NR, 12, false>;
        case 13: return cmpc_solve_kernel<NR, 13, false>;  // ergoCubSN000
        case 15: return cmpc_solve_kernel<NR, 15, false>;  // iCubGazeboV3
        case 20: return cmpc_solve_kernel<NR, 20, false>;  // ergoCubGazeboV1
        case 22: return cmpc_solve_kernel<NR, 22, false>;  // ergoCubSN001
        default: return cmpc_solve_kernel<NR, 0, false>;
    }
}
// once per handle (cmpc_create): raise the variant's dynamic-LDS limit to what this horizon needs (the runtime-N variants serve several horizons: never lowered)
extern "C" int cmpc_prepare_solver(int N, int factors_global, size_t lds_bytes)
{
    int nthreads = 0;
    const cmpc_kernel_t kern = solver_variant(N, factors_global != 0, nthreads);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> limit;   // (the attribute belongs to the function on ONE device)
    std::lock_guard<std::mutex> lock(mu);
    size_t& cur = limit[std::make_pair(dev, reinterpret_cast<const void*>(kern))];
    if (lds_bytes <= cur) return 0;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) cur = lds_bytes;
    return (int)e;
}
extern "C" int cmpc_launch_solver(const CmpcParams* prm, size_t lds_bytes, hipStream_t stream)
{
    int nthreads = 0;
    const cmpc_kernel_t kern = solver_variant(prm->N, prm->scratch != nullptr, nthreads);
    CmpcParams kp = *prm;
    kp.lds_words = (int)(lds_bytes / 4);
    (void)hipGetLastError();   // (a stale error of an earlier, unrelated runtime call must not be read as this launch's)
    hipLaunchKernelGGL(kern, dim3(prm->B), dim3(nthreads), lds_bytes, stream, kp);
    return (int)hipGetLastError();
}
