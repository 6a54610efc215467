// Contact-schedule logic shared by the host entry points and the device kernels:
//   * the merge of the planner's footsteps with the MPC-adjusted current contact -- updateContactPhaseList,
//     src/centroidal-mpc-walking/src/CentroidalMPCBlock.cpp:32-110 (call site :594-607);
//   * the sampling of a contact list at the MPC knots into the parameter tensors -- the job of
//     CentroidalMPC::setContactPhaseList (call site :609; the rule itself lives in BipedalLocomotionFramework, whose
//     source is not in the reference tree: ours is stated in contacts.py and restated here);
//   * the step adjustment getOutput() reports (:598, :626): the next contact takes the optimised landing position.
// One foot of one problem = a list of at most M contacts sorted by activation time:
//   t[m][2]    activation, deactivation time in seconds (double)
//   pose[m][7] position x y z, orientation quaternion w x y z (float)
#pragma once
#include "cmpc_device.h"

#define CMPC_TIME_EPS 1e-9

// ContactList::getActiveContact(t): activation <= t < deactivation (CentroidalMPCBlock.cpp:61, :69), or -1
__host__ __device__ inline int cmpc_active_contact(const double* t, int n, double now)
{
    for (int m = 0; m < n; ++m)
        if (t[2 * m] <= now + CMPC_TIME_EPS && now + CMPC_TIME_EPS < t[2 * m + 1]) return m;
    return -1;
}
// ContactList::getNextContact(t): the contact with the lowest activation time after t (:44), or -1
__host__ __device__ inline int cmpc_next_contact(const double* t, int n, double now)
{
    for (int m = 0; m < n; ++m)
        if (t[2 * m] > now + CMPC_TIME_EPS) return m;
    return -1;
}

// updateContactPhaseList for one foot (CentroidalMPCBlock.cpp:41-103).  Returns false when the MPC list has an
// active contact but the planner's has none (:69-77).  out must hold M contacts.
__host__ __device__ inline bool cmpc_merge_foot(double now, const double* plan_t, const float* plan_pose, int plan_n,
                                                const double* mpc_t, const float* mpc_pose, int mpc_n, int M,
                                                double* out_t, float* out_pose, int* out_n)
{
    int n = 0;
    // the current contact keeps the pose the MPC gave it and takes the planner's timing (:79-82); it starts before
    // every future contact, so it goes first (ContactList orders by time)
    const int ma = cmpc_active_contact(mpc_t, mpc_n, now);          // :61
    if (ma >= 0) {
        const int pa = cmpc_active_contact(plan_t, plan_n, now);    // :69
        if (pa < 0) { *out_n = 0; return false; }                   // :70-77
        out_t[0] = plan_t[2 * pa]; out_t[1] = plan_t[2 * pa + 1];
        for (int i = 0; i < 7; ++i) out_pose[i] = mpc_pose[7 * ma + i];
        n = 1;
    }
    // every future contact of the planner (:44-58)
    const int first = cmpc_next_contact(plan_t, plan_n, now);
    if (first >= 0)
        for (int m = first; m < plan_n && n < M; ++m, ++n) {
            out_t[2 * n] = plan_t[2 * m]; out_t[2 * n + 1] = plan_t[2 * m + 1];
            for (int i = 0; i < 7; ++i) out_pose[7 * n + i] = plan_pose[7 * m + i];
        }
    *out_n = n;
    return true;
}

// owner of the stage that starts at time t: the active contact, else the next one to activate, else the last
__host__ __device__ inline int cmpc_stage_owner(const double* t, int n, double now, bool* active)
{
    int m = cmpc_active_contact(t, n, now);
    *active = m >= 0;
    if (m >= 0) return m;
    m = cmpc_next_contact(t, n, now);
    return m >= 0 ? m : n - 1;
}

__host__ __device__ inline void cmpc_quat_to_R(const float* q /* w x y z */, float* R /* row-major 3x3 */)
{
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - w * z);       R[2] = 2.f * (x * z + w * y);
    R[3] = 2.f * (x * y + w * z);       R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - w * x);
    R[6] = 2.f * (x * z - w * y);       R[7] = 2.f * (y * z + w * x);       R[8] = 1.f - 2.f * (x * x + y * y);
}

// setContactPhaseList for one foot `c` of one problem: fills the foot's blocks of the parameter vector p
// (reference layout) and returns the landing knot (first knot in contact after a swing stage, N if the foot is still
// in the air at the end of the horizon, -1 if it never leaves the ground).  Rule (contacts.py): stage k starts at
// now + k dt; Gamma_k = 1 iff a contact is active then; R_k, the limits of row k and nominal_{k+1} come from the
// stage's owner; nominal_0 and currentPos from the owner of stage 0.
// stage k of foot c (the body of cmpc_sample_foot's loop): writes the stage's entries of p, returns whether the foot is in contact at the stage's start
__host__ __device__ inline bool cmpc_sample_stage(int N, double dt, double now, int c, int k, const double* t, const float* pose, int n,
                                                  const float* box_upper, const float* box_lower, float* p)
{
    const CmpcIdx L{N};
    bool act;
    const int o = cmpc_stage_owner(t, n, now + k * dt, &act);
    float R[9];
    cmpc_quat_to_R(pose + 7 * o + 3, R);
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) p[L.pR(c) + 9 * k + 3 * cc + r] = R[3 * r + cc];   // vec(R) column-major
    p[L.pGam(c) + k] = act ? 1.f : 0.f;
    for (int i = 0; i < 3; ++i) {
        p[L.pUp(c) + 3 * k + i] = box_upper[3 * c + i];
        p[L.pLo(c) + 3 * k + i] = box_lower[3 * c + i];
        p[L.pNom(c) + 3 * (k + 1) + i] = pose[7 * o + i];
        if (k == 0) { p[L.pNom(c) + i] = pose[7 * o + i]; p[L.pCur(c) + i] = pose[7 * o + i]; }
    }
    return act;
}
// the landing knot from the stages' contact flags (first knot in contact after a swing stage, N if still in the air at the end, -1 if the foot never lifts);
// act(k) is called once per stage, in order
template <class ActOf>
__host__ __device__ inline int cmpc_landing_knot(int N, ActOf act)
{
    int land = -1;
    bool prev_act = true;
    for (int k = 0; k < N; ++k) {
        const bool a = act(k);
        if (a && !prev_act && land < 0) land = k;
        prev_act = a;
    }
    if (!prev_act && land < 0) land = N;
    return land;
}
__host__ __device__ inline int cmpc_sample_foot(int N, double dt, double now, int c, const double* t, const float* pose, int n,
                                                const float* box_upper, const float* box_lower, float* p)
{
    return cmpc_landing_knot(N, [&](int k) { return cmpc_sample_stage(N, dt, now, c, k, t, pose, n, box_upper, box_lower, p); });
}


// 8f-3, CentroidalMPCBlock.cpp:525-577: knot k of the references from the planner's trajectories (n_in knots every in_dt seconds, the first one t_offset seconds
// before "now"; h_in not yet divided by the mass) by linear interpolation, clamped to the trajectory's ends; the CoM height is replaced by com_height unless it is
// NaN (the reference forces 0.7, :534).  ci / hi: the problem's [n_in][3]; com_out / h_out: the knot's three entries of comRef / hRef.
__host__ __device__ inline void cmpc_resample_reference_knot(const float* ci, const float* hi, int n_in, double in_dt, double t_offset, double dt, int k,
                                                             double robot_mass, double com_height, float* com_out, float* h_out)
{
    double s = (t_offset + k * dt) / in_dt;
    if (s < 0) s = 0;
    if (s > n_in - 1) s = n_in - 1;
    int i0 = (int)s;
    if (i0 > n_in - 2) i0 = n_in - 2;
    const double w = s - i0;
    for (int a = 0; a < 3; ++a) {
        double cv = (1 - w) * ci[3 * i0 + a] + w * ci[3 * (i0 + 1) + a];
        if (a == 2 && com_height == com_height) cv = com_height;
        com_out[a] = (float)cv;
        h_out[a] = (float)(((1 - w) * hi[3 * i0 + a] + w * hi[3 * (i0 + 1) + a]) / robot_mass);
    }
}
