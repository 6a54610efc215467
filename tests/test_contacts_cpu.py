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
