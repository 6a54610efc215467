"""SURVEY 8f-1 on the CPU: the contact-schedule logic around the solve -- updateContactPhaseList
(CentroidalMPCBlock.cpp:32-110) and the sampling of a phase list into the MPC's parameter tensors
(setContactPhaseList, :609) -- in its three host forms: the float64 restatement of the reference's function
(oracle/contacts_ref.py), the host entry points of the C ABI (cmpc_contacts_merge / _sample / _adjust: no GPU needed)
and the vectorised numpy mirror (contacts.sample_schedule_batch)."""
import ctypes as C

import numpy as np
import pytest

import cmpc_amd as cm
from cmpc_amd.contacts import PlannedContact, pack_lists, sample_schedule, sample_schedule_batch, update_contact_phase_list
from oracle import contacts_ref


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _random_walks(cfg, B, seed, t_end=4.0):
    """B random alternating-foot plans with absolute times from zero (yawed footsteps)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(B):
        step_T, ds = rng.uniform(0.5, 0.9), rng.uniform(0.1, 0.25)
        swing_left = bool(rng.integers(2))
        feet = {True: [PlannedContact(0.0, 0.0, (0.0, 0.08, 0.0))], False: [PlannedContact(0.0, 0.0, (0.0, -0.08, 0.0))]}
        t = ds
        while t < t_end:
            land = t + step_T - ds
            st = feet[not swing_left][-1]
            feet[swing_left][-1].deactivation_time = t
            feet[swing_left].append(PlannedContact(land, land, (st.position[0] + rng.uniform(0, 0.15), (0.08 if swing_left else -0.08) + rng.uniform(-0.02, 0.02), 0.0),
                                                   rng.uniform(-0.3, 0.3)))
            t = land + ds
            swing_left = not swing_left
        for lst in feet.values():
            lst[-1].deactivation_time = 1e9
        out.append({cfg.contacts[0].contact_name: feet[True], cfg.contacts[1].contact_name: feet[False]})
    return out


@pytest.mark.parametrize("t0", [0.0, 0.06 * 7, 1.37])
def test_batched_sampling_matches_the_per_problem_rule(t0):
    cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
    lists = _random_walks(cfg, 48, 5)
    t, pose, n = pack_lists(cfg, lists)
    got, land = sample_schedule_batch(cfg, t, pose, n, t0)
    for b, l in enumerate(lists):
        ref = sample_schedule(cfg, l, t0)
        np.testing.assert_array_equal(got["enabled"][b], ref["enabled"])
        for k in ("R", "upper", "lower", "nominal", "current"):
            np.testing.assert_allclose(got[k][b], ref[k], atol=3e-7)
        for c in range(2):
            en = ref["enabled"][c]
            lk = -1
            for k in range(cfg.N):
                if en[k] < 0.5 and (k + 1 == cfg.N or en[k + 1] > 0.5):
                    lk = k + 1
                    break
            assert land[b, c] == lk


def test_c_abi_sampling_matches_numpy_and_fills_only_the_contact_blocks():
    cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
    lib = cm._capi.lib()
    L = cm.Layout(cfg.N)
    lists = _random_walks(cfg, 32, 9)
    t, pose, n = pack_lists(cfg, lists)
    B, M = t.shape[0], t.shape[2]
    up = np.array([c.bounding_box_upper_limit for c in cfg.contacts], np.float32)
    lo = np.array([c.bounding_box_lower_limit for c in cfg.contacts], np.float32)
    P = np.full((B, L.np), 7.0, np.float32)
    land = np.zeros((B, 2), np.int32)
    now = 0.06 * 11
    assert lib.cmpc_contacts_sample(cfg.N, cfg.sampling_time, B, M, now, _ptr(t), _ptr(pose), _ptr(n), _ptr(up), _ptr(lo), _ptr(P), _ptr(land)) == 0
    ref, rland = sample_schedule_batch(cfg, t, pose, n, now)
    z = np.zeros((B, 3))
    Pref = cm.pack_parameters(cfg.N, ref["R"], ref["upper"], ref["lower"], ref["enabled"], ref["nominal"], ref["current"], z, z, z,
                              np.zeros((B, cfg.N + 1, 3)), np.zeros((B, cfg.N + 1, 3)), dtype=np.float32)
    np.testing.assert_array_equal(land, rland)
    np.testing.assert_allclose(P[:, :L.p_com0], Pref[:, :L.p_com0], atol=1e-7)
    assert (P[:, L.p_com0:] == 7.0).all()     # state, references and wrench rows are left alone
    # argument checking: a foot without contacts is refused
    n0 = n.copy(); n0[3, 1] = 0
    assert lib.cmpc_contacts_sample(cfg.N, cfg.sampling_time, B, M, now, _ptr(t), _ptr(pose), _ptr(n0), _ptr(up), _ptr(lo), _ptr(P), None) != 0


def _to_ref(t, pose, n, b, names):
    return {nm: [dict(activation=float(t[b, c, m, 0]), deactivation=float(t[b, c, m, 1]), position=pose[b, c, m, :3].copy(),
                      quaternion=pose[b, c, m, 3:].copy()) for m in range(int(n[b, c]))] for c, nm in enumerate(names)}


def test_merge_matches_the_restatement_of_the_reference_function():
    cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
    names = [c.contact_name for c in cfg.contacts]
    lists = _random_walks(cfg, 64, 21)
    plan = pack_lists(cfg, lists, max_contacts=12)
    # the MPC's previous output: the same plan with every position moved (as the step adjustment does)
    mpc = (plan[0].copy(), plan[1].copy(), plan[2].copy())
    mpc[1][..., :3] += np.random.default_rng(1).uniform(-0.01, 0.01, mpc[1][..., :3].shape).astype(np.float32)
    for now in (0.0, 0.06 * 5, 0.06 * 13, 2.2):
        (ot, op, on), ok = update_contact_phase_list(now, plan, mpc)
        assert ok.all()
        for b in range(64):
            good, ref = contacts_ref.update_contact_phase_list(now + 1e-9, _to_ref(*plan, b, names), _to_ref(*mpc, b, names))
            assert good
            for c, nm in enumerate(names):
                assert on[b, c] == len(ref[nm])
                for m, rc in enumerate(ref[nm]):
                    assert ot[b, c, m, 0] == rc["activation"] and ot[b, c, m, 1] == rc["deactivation"]
                    np.testing.assert_array_equal(op[b, c, m, :3], rc["position"])
                    np.testing.assert_array_equal(op[b, c, m, 3:], rc["quaternion"])
    # the current contact carries the MPC's pose with the planner's timing (CentroidalMPCBlock.cpp:79-82)
    (ot, op, on), ok = update_contact_phase_list(0.0, plan, mpc)
    np.testing.assert_array_equal(op[:, :, 0, :3], mpc[1][:, :, 0, :3])
    np.testing.assert_array_equal(ot[:, :, 0], plan[0][:, :, 0])


def test_merge_reports_false_like_the_reference_when_the_planner_has_no_active_contact():
    cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
    names = [c.contact_name for c in cfg.contacts]
    lists = _random_walks(cfg, 4, 3)
    plan = pack_lists(cfg, lists, max_contacts=12)
    mpc = (plan[0].copy(), plan[1].copy(), plan[2].copy())
    # problem 2: the planner's left foot is in the air at `now` while the MPC's previous list still has it on the ground
    now = 1.0
    plan[0][2, 0, :, :] += 100.0          # every planner contact of that foot starts later
    (ot, op, on), ok = update_contact_phase_list(now, plan, mpc)
    good, _ = contacts_ref.update_contact_phase_list(now, _to_ref(*plan, 2, names), _to_ref(*mpc, 2, names))
    mpc_active = contacts_ref.get_active_contact(_to_ref(*mpc, 2, names)[names[0]], now) is not None
    assert good == (not mpc_active)
    assert ok.tolist() == [True, True, good, True]


def test_step_adjustment_writes_the_landing_position_into_the_next_contact():
    cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
    lib = cm._capi.lib()
    L = cm.Layout(cfg.N)
    lists = _random_walks(cfg, 16, 2)
    t, pose, n = pack_lists(cfg, lists)
    B, M = t.shape[0], t.shape[2]
    now = 0.3
    _, land = sample_schedule_batch(cfg, t, pose, n, now)
    X = np.random.default_rng(0).normal(size=(B, L.nx)).astype(np.float32)
    before = pose.copy()
    assert lib.cmpc_contacts_adjust(cfg.N, B, M, now, _ptr(X), _ptr(land), _ptr(t), _ptr(pose), _ptr(n)) == 0
    for b in range(B):
        for c in range(2):
            nxt = [m for m in range(n[b, c]) if t[b, c, m, 0] > now + 1e-9]
            for m in range(M):
                if land[b, c] >= 0 and nxt and m == nxt[0]:
                    np.testing.assert_array_equal(pose[b, c, m, :3], L.x_pos(X[b], c)[land[b, c]])
                    np.testing.assert_array_equal(pose[b, c, m, 3:], before[b, c, m, 3:])
                else:
                    np.testing.assert_array_equal(pose[b, c, m], before[b, c, m])


# ---- against oracle/schedule_ref.py: the independent (integer-nanosecond, one problem, plain Python) restatement of the sampling
# and step-adjustment rules.  The helpers are shared with tests/test_gpu_rollout.py, which holds the device kernels to the same oracle.
def assert_sample_matches_schedule_oracle(cfg, t, pose, n, now, P, land, rows=None):
    """P[B, np] (contact blocks as a sampler left them) and land[B, 2] against schedule_ref.sample_contact_phase_list:
    Gamma and the landing knots bit-exact, positions and limits exact (they are copies of float32 inputs), R to 3e-7 (the
    kernels evaluate the quaternion formula in float32: two or three ulp of 1)."""
    from oracle import schedule_ref
    names = [c.contact_name for c in cfg.contacts]
    boxes = {c.contact_name: (np.asarray(c.bounding_box_upper_limit, np.float32), np.asarray(c.bounding_box_lower_limit, np.float32)) for c in cfg.contacts}
    L, N = cm.Layout(cfg.N), cfg.N
    for b in (range(t.shape[0]) if rows is None else rows):
        ref = schedule_ref.sample_contact_phase_list(N, cfg.sampling_time, now, _to_ref(t, pose, n, b, names), boxes)
        for c, nm in enumerate(names):
            r = ref[nm]
            assert int(land[b, c]) == r["land"], (b, c, land[b, c], r["land"])
            np.testing.assert_array_equal(P[b, L.p_gam[c]:L.p_gam[c] + N], np.asarray(r["gamma"], np.float32))
            Rk = P[b, L.p_R[c]:L.p_R[c] + 9 * N].reshape(N, 3, 3).transpose(0, 2, 1)      # vec(R) is column-major
            np.testing.assert_allclose(Rk, np.asarray(r["R"]), rtol=0, atol=3e-7)
            np.testing.assert_array_equal(P[b, L.p_up[c]:L.p_up[c] + 3 * N].reshape(N, 3), np.asarray(r["upper"], np.float32))
            np.testing.assert_array_equal(P[b, L.p_lo[c]:L.p_lo[c] + 3 * N].reshape(N, 3), np.asarray(r["lower"], np.float32))
            np.testing.assert_array_equal(P[b, L.p_nom[c]:L.p_nom[c] + 3 * (N + 1)].reshape(N + 1, 3), np.asarray(r["nominal"], np.float32))
            np.testing.assert_array_equal(P[b, L.p_cur[c]:L.p_cur[c] + 3], np.asarray(r["current"], np.float32))


def assert_adjust_matches_schedule_oracle(cfg, t, pose_before, pose_after, n, now, X, land, rows=None):
    from oracle import schedule_ref
    names = [c.contact_name for c in cfg.contacts]
    L = cm.Layout(cfg.N)
    for b in (range(t.shape[0]) if rows is None else rows):
        lists = _to_ref(t, pose_before, n, b, names)
        for c, nm in enumerate(names):
            lk = int(land[b, c])
            ref = schedule_ref.adjust_contact_list(now, lists[nm], lk, L.x_pos(X[b], c)[lk] if 0 <= lk <= cfg.N else None)
            assert len(ref) == n[b, c]
            for m, rc in enumerate(ref):
                np.testing.assert_array_equal(pose_after[b, c, m, :3], np.asarray(rc["position"], np.float32))
                np.testing.assert_array_equal(pose_after[b, c, m, 3:], rc["quaternion"])


def assert_merge_matches_contacts_oracle(cfg, now, plan, mpc, out, ok, rows=None):
    """(t, pose, n) of a merge and its ok flags against oracle/contacts_ref.update_contact_phase_list (the restatement of the
    reference's function, CentroidalMPCBlock.cpp:32-110), including the problems where it returns false."""
    names = [c.contact_name for c in cfg.contacts]
    ot, op, on = out
    for b in (range(plan[0].shape[0]) if rows is None else rows):
        good, ref = contacts_ref.update_contact_phase_list(now + 1e-9, _to_ref(*plan, b, names), _to_ref(*mpc, b, names))
        assert bool(ok[b]) == good, (b, ok[b], good)
        if not good:
            continue
        for c, nm in enumerate(names):
            assert on[b, c] == len(ref[nm])
            for m, rc in enumerate(ref[nm]):
                assert ot[b, c, m, 0] == rc["activation"] and ot[b, c, m, 1] == rc["deactivation"]
                np.testing.assert_array_equal(op[b, c, m, :3], rc["position"])
                np.testing.assert_array_equal(op[b, c, m, 3:], rc["quaternion"])


@pytest.mark.parametrize("now", [0.0, 0.06 * 6, 0.06 * 14, 1.2345])
def test_host_sampling_and_adjustment_match_the_schedule_oracle(now):
    cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
    lib = cm._capi.lib()
    L = cm.Layout(cfg.N)
    lists = _random_walks(cfg, 24, 31)
    # one problem on the reference's own kind of clock: every time a multiple of dT, so that knots fall exactly on activations
    lists[0] = cm.rollout.walking_plan(cfg)
    t, pose, n = pack_lists(cfg, lists, max_contacts=12)
    B, M = t.shape[0], t.shape[2]
    up = np.array([c.bounding_box_upper_limit for c in cfg.contacts], np.float32)
    lo = np.array([c.bounding_box_lower_limit for c in cfg.contacts], np.float32)
    P = np.zeros((B, L.np), np.float32)
    land = np.zeros((B, 2), np.int32)
    assert lib.cmpc_contacts_sample(cfg.N, cfg.sampling_time, B, M, now, _ptr(t), _ptr(pose), _ptr(n), _ptr(up), _ptr(lo), _ptr(P), _ptr(land)) == 0
    assert_sample_matches_schedule_oracle(cfg, t, pose, n, now, P, land)
    X = np.random.default_rng(4).normal(size=(B, L.nx)).astype(np.float32)
    after = pose.copy()
    assert lib.cmpc_contacts_adjust(cfg.N, B, M, now, _ptr(X), _ptr(land), _ptr(t), _ptr(after), _ptr(n)) == 0
    assert_adjust_matches_schedule_oracle(cfg, t, pose, after, n, now, X, land)
    assert np.abs(after - pose).max() > 0


def test_schedule_oracle_on_a_hand_checked_walk():
    """The oracle itself on a case worked out by hand: dT = 0.06, the left foot lifts at 0.36 s and lands 0.1 m ahead at 0.84 s."""
    from oracle import schedule_ref
    q = [1.0, 0.0, 0.0, 0.0]
    left = [dict(activation=0.0, deactivation=0.36, position=[0, 0.08, 0], quaternion=q), dict(activation=0.84, deactivation=100.0, position=[0.1, 0.08, 0], quaternion=q)]
    s = schedule_ref.sample_contact_list(20, 0.06, 0.0, left, [0.01, 0.05, 0], [-0.01, 0, 0])
    assert s["gamma"] == [1.0] * 6 + [0.0] * 8 + [1.0] * 6 and s["land"] == 14            # stage 6 starts AT the lift-off: in the air
    assert s["nominal"][:7] == [[0, 0.08, 0]] * 7 and s["nominal"][7:] == [[0.1, 0.08, 0]] * 14 and s["current"] == [0, 0.08, 0]
    s = schedule_ref.sample_contact_list(20, 0.06, 0.06 * 10, left, [0.01, 0.05, 0], [-0.01, 0, 0])       # ten ticks later: in the air, lands at knot 4
    assert s["gamma"] == [0.0] * 4 + [1.0] * 16 and s["land"] == 4 and s["current"] == [0.1, 0.08, 0]
    assert schedule_ref.sample_contact_list(20, 0.06, 0.0, [], [0, 0, 0], [0, 0, 0]) is None
    adj = schedule_ref.adjust_contact_list(0.06 * 10, left, 4, [0.105, 0.09, 0.0])
    assert adj[1]["position"] == [0.105, 0.09, 0.0] and adj[0]["position"] == [0, 0.08, 0]
    assert schedule_ref.adjust_contact_list(0.06 * 15, left, -1, None) == left
