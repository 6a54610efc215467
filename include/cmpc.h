/* cmpc.h -- C ABI of the MI355X batched centroidal-MPC solver (libcmpc_hip.so).
 *
 * Drop-in boundary for ONE path of GiulioRomualdi/paper_romualdi_2022_icra_centroidal-mpc-walking:
 * the solve inside BipedalLocomotion::ReducedModelControllers::CentroidalMPC, as the reference
 * drives it from src/centroidal-mpc-walking/src/CentroidalMPCBlock.cpp:
 *     initialize              :144   -> cmpc_create            (keys of config/robots/<robot>/centroidal_mpc.ini)
 *     setState                :407   -> cmpc_set_state
 *     setReferenceTrajectory  :579   -> cmpc_set_reference
 *     setContactPhaseList     :609   -> cmpc_set_contacts
 *     advance                 :615   -> cmpc_solve / cmpc_solve_device   (replaces CasADi Opti -> IPOPT)
 *     getOutput               :622   -> cmpc_get_solution / cmpc_get_output
 * and the NLP callbacks IPOPT would call (generated code config/robots/ergoCubGazeboV1/tmp.c:
 * nlp_fg :12430, nlp_jac_fg :71962, nlp_hess_l :58926) -> cmpc_eval_nlp_device; nlp_grad :24791 -> cmpc_eval_nlp_grad_device.
 *
 * Conventions: plain C, no exceptions; every function returns 0 on success or a negative
 * cmpc_status; cmpc_last_error() gives the text.  The caller owns every buffer it passes; the
 * handle owns its device buffers; one handle = one device + one HIP stream, single caller
 * (the reference calls the class from one thread, Main.cpp:98-110).
 *
 * Data layout: decision vector x[n_x] and parameter vector p[n_p] of every problem are laid out
 * exactly as in the reference's generated NLP (tmp.c:62-67): n_x = 45N+15, n_p = 50N+27,
 *   x = com[3(N+1)] dcom[3(N+1)] h[3(N+1)] then per contact (left_foot, right_foot):
 *       pos[3(N+1)] vel[3N] f_corner0..3[3N each]           (3 x knots, column-major)
 *   p = per contact: R[9N] (vec of 3x3 col-major per knot) upper[3N] lower[3N] enabled[N]
 *       nominalPos[3(N+1)] currentPos[3]; then com0 dcom0 h0 comRef[3(N+1)] hRef[3(N+1)]
 *       fExt[3N] tauExt[3N]
 * Batches are row-major: P[B][n_p], X[B][n_x], float32.
 */
#ifndef CMPC_H
#define CMPC_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cmpc_handle_s* cmpc_handle;

typedef enum {
    CMPC_OK = 0,
    CMPC_ERR_ARG = -1,       /* bad argument / unsupported configuration */
    CMPC_ERR_HIP = -2,       /* HIP runtime error (no device, allocation, launch) */
    CMPC_ERR_NOT_CONVERGED = -3 /* at least one problem of the batch did not converge (see info) */
} cmpc_status;

/* keys of centroidal_mpc.ini (ergoCubGazeboV1/centroidal_mpc.ini:3-42) + solver options */
typedef struct {
    int horizon;              /* N = time_horizon / sampling_time (or controller_horizon)        */
    double sampling_time;     /* dt                                                               */
    double friction_coefficient; /* static_friction_coefficient (number_of_slices must be 1)      */
    double gravity;           /* 9.80665                                                          */
    double com_weight[3];
    double angular_momentum_weight;
    double contact_position_weight;
    double force_rate_of_change_weight[3];
    double contact_force_symmetry_weight;
    double corners[2][4][3];  /* [CONTACT_i] corner_j, contacts in alphabetical name order        */
    /* solver options (ipopt_tolerance / ipopt_max_iteration take the place of IPOPT's) */
    int max_iterations;       /* Newton iteration budget per solve (default 40)                   */
    double tolerance;         /* on the primal residuals and on max t*z (<= 0: default 1e-6 up to N = 20, 3e-7 beyond) */
    double step_tolerance;    /* on the last Newton step, max-norm over states and forces (default 1e-4) */
    double mu_init;           /* initial barrier parameter; <= 0 (default): per problem, from its
                               * initial infeasibility ep0: clamp(3.5 ep0^2, 0.03, 0.5)           */
    double mu_min;            /* final barrier parameter (<= 0: default max(0.05 x tolerance, 5e-8): the float32 factorisations start to fail below) */
    int exact_hessian;        /* 1 (default): Lagrangian Hessian; 0: Gauss-Newton                 */
    int final_extrapolation;  /* 1 (default): a converged solve finishes with one affine-scaling Newton step towards
                               * mu = 0 (one more factorisation): removes the O(mu) bias of the barrier floor and
                               * the remaining termination error -- worst parity error of 1024 problems 8.7e-5 ->
                               * 1.7e-5 (config 2) for +0.8 iteration; 0: stop at the barrier floor */
    /* tail polish (needs final_extrapolation): when the extrapolation step of the forces of the last tail_stages stages
     * exceeds tail_trigger x the largest force component -- the symptom of nearly degenerate friction rows there, e.g. an
     * unloaded corner at the apex of its pyramid, which the barrier keeps sqrt(mu / curvature) away from the optimum --
     * those stages are re-solved on their own (state entering them held): tail_iterations Newton steps with per-row
     * barrier targets, then their own extrapolation step.  Defaults 3 / 2 / 2e-5; tail_stages 0 switches it off. */
    int tail_stages;
    int tail_iterations;
    double tail_trigger;
    /* where the per-stage factor records of the Riccati recursion live: CMPC_FACTORS_AUTO (default) = in LDS, one problem per compute unit, eight waves
     * (the latency variant) when the batch does not exceed the number of compute units and the horizon's image fits 160 KiB, else in HBM scratch with three
     * problems per compute unit (the throughput variant); CMPC_FACTORS_LDS / _HBM force one (LDS only where it fits).  Both give the same solutions to
     * the tolerance; tests use the switch to hold them together. */
    int factor_storage;
} cmpc_config;
#define CMPC_FACTORS_AUTO 0
#define CMPC_FACTORS_LDS 1
#define CMPC_FACTORS_HBM 2

/* number of floats per solve in the info array */
#define CMPC_INFO 8
/* info[b] = { iterations, kkt_error = max(primal_inf, max t*z), mu, safeguards, primal_inf,
 * status (0 ok, 1 iteration budget exhausted, 2 factorisation failed or a residual that is not finite -- NaN/inf in P or
 * X0: IPOPT's "invalid number"), solve_cycles (shader clock),
 * last_step (max-norm of the last Newton step taken inside the loop, forces relative to the largest force) }.
 * kkt_error, mu and primal_inf are those of the last iterate whose residuals were evaluated: the iterate the
 * termination test accepted.  With final_extrapolation the returned x is one affine-scaling step beyond it.
 * safeguards = Gauss-Newton fallbacks + 100 x emergency re-centrings (warm starts) + 10000 x (1 if the warm-started pass
 * was abandoned and the problem solved again from the cold start) + 100000 x (1 if the tail was polished)
 * + 1000000 x (times a wave of the streaming backward stage gave up waiting at a hand-off word: never observed; a protocol bug would show HERE and not
 * as a failed factorisation -- the pass is repeated once unchanged.  Tests, the soak tool and bench.py assert / report that this digit is zero).
 * iterations counts both passes of a restarted warm start: it can reach 2 x max_iterations. */

void cmpc_default_config(cmpc_config* cfg);                      /* ergoCubGazeboV1 values, N=20 */
int cmpc_dims(int horizon, int* n_x, int* n_p, int* n_g, int* nnz_jac, int* nnz_hess);

int cmpc_create(const cmpc_config* cfg, int batch, int device, cmpc_handle* out);
int cmpc_destroy(cmpc_handle h);
const char* cmpc_last_error(cmpc_handle h);                       /* h may be NULL */
int cmpc_batch(cmpc_handle h);
void* cmpc_stream(cmpc_handle h);                                 /* hipStream_t of the handle */

/* ---- the hot path: solve a batch, operands resident in device memory ----
 * dP[B][n_p], dX0[B][n_x] (initial guess), dX[B][n_x] (solution), dInfo[B][CMPC_INFO] or NULL.
 * Asynchronous on the handle's stream (or on `stream` if non-NULL); nothing is copied. */
int cmpc_solve_device(cmpc_handle h, const float* dP, const float* dX0, float* dX, float* dInfo,
                      void* stream);
/* the same when dX0 is the previous solution shifted by one knot (cmpc_shift_solution_device; is_warm_start_enabled,
 * ergoCubGazeboV1/centroidal_mpc.ini:9): the barrier starts near the central path (mu = 1e-2) and a problem whose warm
 * start does not converge is solved again inside the kernel from the cold start.  The warm property is this argument
 * list's, not the handle's: any buffer, any stream. */
int cmpc_solve_device_warm(cmpc_handle h, const float* dP, const float* dX0, float* dX, float* dInfo,
                           void* stream);
/* What a warm-started problem that does not converge costs (both cmpc_solve_device_warm and the class path): warm_budget =
 * iterations the warm-started pass may take (0: max_iterations; default 14: healthy warm ticks of the walking roll-out need at
 * most 13, and a tick is as slow as its slowest problem); restart_in_kernel != 0 (default): such a problem is then
 * started again from the cold start inside the same launch (info: safeguards += 10000); 0: it comes back with status 1 and
 * the caller re-solves it -- in a batch, the few stragglers of a tick in a small launch of their own (one CU each) instead of
 * one workgroup holding its CU for two budgets.  The reference can only abort the tick (CentroidalMPCBlock.cpp:615-619). */
int cmpc_set_warm_policy(cmpc_handle h, int warm_budget, int restart_in_kernel);
/* host buffers (includes the PCIe copies; synchronous). info may be NULL. Returns
 * CMPC_ERR_NOT_CONVERGED if any problem's status != 0 (the solutions are still written). */
int cmpc_solve(cmpc_handle h, const float* P, const float* X0, float* X, float* info);
/* duration of the last solve kernel in ms (HIP events on the launch stream); < 0 if none, or if timing is off */
float cmpc_last_solve_ms(cmpc_handle h);
/* The event pair every solve launch is bracketed with (what cmpc_last_solve_ms reads) costs the stream two barrier packets per solve: measured ~9 us each
 * between back-to-back launches on MI355X.  enabled = 0 stops recording them (a caller that queues solves back to back and times them itself);
 * default: enabled.  No reference counterpart (the reference times its tick on the host, CentroidalMPCBlock.cpp:615-634). */
int cmpc_set_timing(cmpc_handle h, int enabled);
/* test hook: fills the LDS of every compute unit with NaN bit patterns (a kernel on the handle's stream), so that a
 * test can show that a solve does not depend on what an earlier workgroup or kernel left there.  No reference
 * counterpart. */
int cmpc_test_poison_lds(cmpc_handle h);
/* test hook (host only, no GPU): workgroup barriers one role of the streaming backward stage executes in a pass over stages N-1 .. k0 -- role 0 the
 * factorising wave, 1 the consumers -- counted on the loop skeleton both device loops are written with.  Unequal counts would hang a workgroup. */
int cmpc_sq_pass_barriers(int horizon, int k0, int role);

/* ---- NLP callbacks (what IPOPT evaluated through the generated code), batched on the device ----
 * any output pointer may be NULL.  dLamG[B][n_g], lam_f scalar (hess of lam_f f + lam_g^T g).
 * dJac[B][nnz_jac] / dHess[B][nnz_hess] are in the reference's CCS nonzero order
 * (cmpc_nlp_sparsity gives row/col per nonzero; tmp.c:66-67 for N=12). */
int cmpc_eval_nlp_device(cmpc_handle h, const float* dX, const float* dP, const float* dLamG,
                         float lam_f, float* dF, float* dG, float* dGradF, float* dJac,
                         float* dHess, void* stream);
int cmpc_nlp_sparsity(int horizon, int* jac_row, int* jac_col, int* hess_row, int* hess_col);
/* nlp_grad (tmp.c:24791): gradient of gamma = lam_f f + lam_g^T g with respect to x (dGradX[B][n_x]) and to the
 * parameters (dGradP[B][n_p]; zero for limA/limB, currentPos, com0/dcom0/h0, which only enter the bounds).  Either
 * output may be NULL. */
int cmpc_eval_nlp_grad_device(cmpc_handle h, const float* dX, const float* dP, const float* dLamG, float lam_f,
                              float* dGradX, float* dGradP, void* stream);

/* ---- class-shaped setters (host buffers -> the handle's own device P, X0) ----
 * batch-major float32; NULL keeps the previous value (zeros initially).
 *   state    [B][9]            com0, dcom0, h0 (h and wrench already mass-normalised,
 *                              CentroidalMPCBlock.cpp:403-410)
 *   wrench   [B][N][6]         external force (3) and torque (3) per knot, or NULL = 0
 *   com_ref, h_ref [B][N+1][3]
 *   R [B][2][N][9] row-major 3x3, upper/lower [B][2][N][3], enabled [B][2][N],
 *   nominal [B][2][N+1][3], current [B][2][3] */
int cmpc_set_state(cmpc_handle h, const float* state, const float* wrench);
int cmpc_set_reference(cmpc_handle h, const float* com_ref, const float* h_ref);
int cmpc_set_contacts(cmpc_handle h, const float* R, const float* upper, const float* lower,
                      const float* enabled, const float* nominal, const float* current);
/* x0: [B][n_x] or NULL = cold start (CoM at com0, feet at nominal, f_z = g/8 per corner);
 * shift_previous != 0: warm start from the previous solution shifted by one knot
 * (is_warm_start_enabled, ergoCubGazeboV1/centroidal_mpc.ini:9) */
int cmpc_set_initial_guess(cmpc_handle h, const float* x0, int shift_previous);
/* solve the handle's own problem set (set_* above); synchronous */
int cmpc_advance(cmpc_handle h);
int cmpc_get_solution(cmpc_handle h, float* X, float* info);
/* read-back of what the setters above have written: the handle's parameter set P[B][n_p] exactly as the next cmpc_advance will solve it
 * (the reference's setState / setReferenceTrajectory / setContactPhaseList fill CasADi's parameter vector p the same way, CentroidalMPCBlock.cpp:407, :579,
 * :609) -- host copy, or the handle's device buffer after uploading the staged values (valid until the next setter call / cmpc_advance). */
int cmpc_get_parameters(cmpc_handle h, float* P);
int cmpc_get_parameters_device(cmpc_handle h, const float** dP);
/* compact output of getOutput(): per problem first-knot corner forces [2][4][3], contact
 * positions at knot 0 [2][3], next (adjusted) landing position per contact [2][3] and its knot
 * index [2] (-1 if the contact does not land inside the horizon) */
int cmpc_get_output(cmpc_handle h, float* forces0, float* pos0, float* next_pos, int* next_knot);

/* ---- rows next to the solve ----
 * 8f-3, CentroidalMPCBlock.cpp:525-577: planner trajectories (n_in knots every in_dt seconds, the first one t_offset
 * seconds before "now"; com_in/h_in [B][n_in][3], h_in NOT yet divided by the mass) -> comRef/hRef at the N+1 MPC knots
 * by linear interpolation; the CoM height is replaced by com_height unless it is NaN (the reference forces 0.7, :534). */
int cmpc_set_reference_from_planner(cmpc_handle h, const float* com_in, const float* h_in, int n_in, double in_dt,
                                    double t_offset, double robot_mass, double com_height);
/* the same on the device (one thread per problem and knot), into the reference rows of the caller's dP[B][n_p]; dComIn / dHIn [B][n_in][3] device pointers;
 * asynchronous on `stream` (NULL: the handle's) */
int cmpc_write_reference_from_planner_device(cmpc_handle h, const float* dComIn, const float* dHIn, int n_in, double in_dt, double t_offset,
                                             double robot_mass, double com_height, float* dP, void* stream);
/* 8f-4, WholeBodyQPBlock.cpp:805-873, 1083-1084, 1150, 1259-1262: between two MPC ticks the plant integrates the
 * centroidal dynamics under the first-knot corner forces of the active contacts + the external wrench of knot 0 (RK4,
 * `substeps` steps of `step` seconds, forces held) and reports the desired ZMP (local ZMP clamped to +-zmp_half_x/y:
 * 0.08 / 0.03 in the reference).  dStateIn/dStateOut [B][9] (com, dcom, h; may alias), dZmp [B][2] or NULL. Device
 * pointers; asynchronous on `stream` (NULL: the handle's). */
int cmpc_plant_step_device(cmpc_handle h, const float* dX, const float* dP, const float* dStateIn, float* dStateOut,
                           float* dZmp, double step, int substeps, double zmp_half_x, double zmp_half_y, void* stream);

/* 8e, the record a Monte-Carlo driver gathers across GPUs (no reference counterpart: the reference runs one problem):
 * dOut[B][3(N+1) + 38] = CoM trajectory 3(N+1) | first-knot corner forces 24 | knot-0 and knot-1 foot positions 12 |
 * iterations | status, from dX[B][n_x] and dInfo[B][8].  Device pointers; asynchronous on `stream` (NULL: the handle's). */
int cmpc_compact_output_device(cmpc_handle h, const float* dX, const float* dInfo, float* dOut, void* stream);
/* ... and the gather itself: ncclAllGather (RCCL, over xGMI) of every rank's B compact records, dLocal[B][3(N+1) + 38] -> dAll[world_size][B][...], on
 * `stream` (NULL: the handle's).  nccl_comm is the caller's ncclComm_t (one rank per GPU and process, ncclCommInitRank; every rank's handle must have the
 * same batch -- pad ragged shards).  librccl.so is opened on first use.  The Python harness does the same with torch.distributed (distributed.py); this entry
 * point is what a C++ Monte-Carlo driver shaped after the reference's Main.cpp:98-134 calls (examples/montecarlo_allgather.cpp). */
int cmpc_allgather_compact_device(cmpc_handle h, void* nccl_comm, int world_size, const float* dLocal, float* dAll, void* stream);

/* ---- 8f-1: contact schedules, batched ----
 * What the reference does with BipedalLocomotion::Contacts::ContactPhaseList objects around the solve, for a batch.
 * One foot of one problem = a list of at most M = max_contacts contacts sorted by activation time (contacts in
 * alphabetical name order: left_foot, right_foot):
 *     t[B][2][M][2]     activation, deactivation time [s] (double)
 *     pose[B][2][M][7]  position x y z, orientation quaternion w x y z (float)
 *     n[B][2]           contacts in use
 * ContactList::getActiveContact(t) = the contact with activation <= t < deactivation; getNextContact(t) = the first
 * one that activates after t (the two queries the reference makes, CentroidalMPCBlock.cpp:44, :61, :69). */

/* updateContactPhaseList, CentroidalMPCBlock.cpp:32-110 (call site :594-607): out = the planner's future contacts
 * (activation > now), preceded -- when the MPC's previous list has an active contact -- by that contact with the pose
 * the MPC gave it and the timing of the planner's active contact (:79-82).  Where the planner has no active contact
 * for such a foot the reference returns false (:69-77): ok[b] = 0 and the function returns CMPC_ERR_ARG after
 * processing the whole batch.  Host buffers; no handle, no GPU. */
int cmpc_contacts_merge(int batch, int max_contacts, double now, const double* plan_t, const float* plan_pose, const int* plan_n,
                        const double* mpc_t, const float* mpc_pose, const int* mpc_n, double* out_t, float* out_pose, int* out_n,
                        int* ok /* [B] or NULL */);
/* same on the device (one thread per problem and foot); dOk[B] or NULL; asynchronous on `stream` (NULL: the handle's).
 * A list length outside 0..max_contacts gives an empty merged list and dOk = 0 (nothing is read through it). */
int cmpc_contacts_merge_device(cmpc_handle h, int max_contacts, double now, const double* dPlanT, const float* dPlanPose,
                               const int* dPlanN, const double* dMpcT, const float* dMpcPose, const int* dMpcN, double* dOutT,
                               float* dOutPose, int* dOutN, int* dOk, void* stream);

/* setContactPhaseList, CentroidalMPCBlock.cpp:609: samples the lists at the knots now + k dt into the contact blocks of
 * P[B][n_p] (R, upper, lower, enabled, nominalPos, currentPos; the other entries of P are left alone).  Rule: stage k
 * is in contact iff a contact is active at its start; its orientation, box limits and the nominal position of knot
 * k+1 come from the stage's owner = the active contact, else the next one to activate, else the last.  box_upper /
 * box_lower [2][3]: bounding_box_{upper,lower}_limit of [CONTACT_i].  land[B][2] (or NULL) receives the landing knot of
 * each foot: first knot in contact after a swing stage, N if still in the air at the end, -1 if it never lifts.
 * A foot whose list is empty or longer than max_contacts: the host entry points return CMPC_ERR_ARG; the device kernel
 * leaves that foot's blocks of dP untouched and writes land = -2 (the caller must not solve that problem: the reference
 * aborts the tick when its merge fails, CentroidalMPCBlock.cpp:603-607). */
int cmpc_contacts_sample(int horizon, double dt, int batch, int max_contacts, double now, const double* t, const float* pose,
                         const int* n, const float* box_upper, const float* box_lower, float* P, int* land);
int cmpc_contacts_sample_device(cmpc_handle h, int max_contacts, double now, const double* dT, const float* dPose, const int* dN,
                                const float* box_upper /* host */, const float* box_lower /* host */, float* dP, int* dLand,
                                void* stream);
/* the same into the handle's own parameter set (what the class facade's setContactPhaseList calls) */
int cmpc_set_contact_lists(cmpc_handle h, int max_contacts, double now, const double* t, const float* pose, const int* n,
                           const float* box_upper, const float* box_lower, int* land);

/* getOutput().contactPhaseList, CentroidalMPCBlock.cpp:598, :626: the next contact (getNextContact(now)) of every foot
 * that lands inside the horizon takes the optimised landing position x.pos[land]. */
int cmpc_contacts_adjust(int horizon, int batch, int max_contacts, double now, const float* X, const int* land, const double* t,
                         float* pose, const int* n);
int cmpc_contacts_adjust_device(cmpc_handle h, int max_contacts, double now, const float* dX, const int* dLand, const double* dT,
                                float* dPose, const int* dN, void* stream);

/* setState on the device: dState[B][9] (com, dcom, h) and dWrench[B][N][6] (or NULL: left alone) into the rows of dP */
int cmpc_write_state_device(cmpc_handle h, const float* dState, const float* dWrench, float* dP, void* stream);

/* ONE receding-horizon tick of the whole batch as ONE call: what CentroidalMPCBlock::advance does between two solves
 * (CentroidalMPCBlock.cpp:594-626) and WholeBodyQPBlock::advance after it (WholeBodyQPBlock.cpp:1083-1150), chained on
 * `stream` (NULL: the handle's) without returning to the host in between:
 *   cmpc_contacts_merge_device (updateContactPhaseList :594-607) -> cmpc_contacts_sample_device (setContactPhaseList :609)
 *   -> cmpc_write_state_device (setState :407) -> cmpc_shift_solution_device (is_warm_start_enabled; warm != 0)
 *   -> cmpc_solve_device[_warm] (advance :615) -> cmpc_contacts_adjust_device (getOutput :626) -> cmpc_plant_step_device.
 * Every step is what the entry point of that name computes, with the same argument checks; the steps in front of the solve are ONE launch and so are the
 * two behind it (they touch disjoint entries; same per-problem device functions: results identical to the last bit), so that a tick is three dispatches instead of ten.  It saves the
 * caller six trips through its FFI and the idle GPU time between them (a seventh of a tick at B <= 256,
 * tools/gpu_rollout_tick_overhead.py).  The reference rows of dP (comRef, hRef) are the caller's -- write them before the call -- unless dPlanCom / dPlanH are given.
 * All pointers are device pointers except box_upper / box_lower (host, [2][3]). */
typedef struct cmpc_tick_io {
    const double* dPlanT; const float* dPlanPose; const int* dPlanN; /* the planner's lists (layout above) */
    const double* dPrevT; const float* dPrevPose; const int* dPrevN; /* the MPC's lists of the previous tick.  All NULL (first tick):
                                                                        no merge, dList* is taken as the caller filled it, dOk is left alone */
    double* dListT; float* dListPose; int* dListN;                   /* this tick's lists: merged here, sampled, and step-adjusted after
                                                                        the solve (the next tick's dPrev*); must not alias dPrev* */
    int* dOk;               /* [B] merge status (0: the tick of that problem must be discarded, :603-607), or NULL */
    int* dLand;             /* [B][2] landing knots */
    const float* box_upper; const float* box_lower;
    const float* dState;    /* [B][9] measured com, dcom, h */
    const float* dWrench;   /* [B][N][6] or NULL (rows of dP left alone) */
    float* dP;              /* [B][n_p] */
    float* dX0;             /* [B][n_x] starting point: written (dX shifted by one knot) when warm != 0, read when warm == 0 */
    float* dX;              /* [B][n_x] previous solution in (warm != 0), this tick's solution out */
    float* dInfo;           /* [B][CMPC_INFO] */
    float* dStateOut;       /* [B][9] state at the next tick (may alias dState) */
    float* dZmp;            /* [B][2] or NULL */
    double plant_step; int plant_substeps; double zmp_half_x, zmp_half_y; /* as cmpc_plant_step_device */
    /* optional: the planner's CoM / angular-momentum trajectories (cmpc_write_reference_from_planner_device: [B][plan_knots][3] each, a knot every plan_dt seconds,
     * the first one plan_t_offset seconds before `now`) -- the tick then writes comRef / hRef of dP itself (setReferenceTrajectory, CentroidalMPCBlock.cpp:525-579).
     * Both NULL: the reference rows of dP are the caller's. */
    const float* dPlanCom; const float* dPlanH; int plan_knots; double plan_dt, plan_t_offset, robot_mass, com_height;
} cmpc_tick_io;
int cmpc_rollout_tick_device(cmpc_handle h, int max_contacts, double now, int warm, const cmpc_tick_io* io, void* stream);
/* is_warm_start_enabled on the device: dX0 = dXprev shifted by one knot; solve from it with cmpc_solve_device_warm
 * (cmpc_set_initial_guess(NULL, 1) + cmpc_advance do the same for the handle's own buffers) */
int cmpc_shift_solution_device(cmpc_handle h, const float* dXprev, float* dX0, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CMPC_H */
