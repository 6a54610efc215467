"""SURVEY 8f-1 on the GPU: the batched contact-schedule kernels against the host entry points, and a receding-horizon
WALKING roll-out kept in HBM (merge -> sample -> setState -> warm shift -> solve -> step adjustment -> plant), 24 ticks
across a lift-off, a landing and the next lift-off."""
import ctypes as C

import numpy as np
import pytest

import cmpc_amd as cm
from cmpc_amd.contacts import pack_lists, sample_schedule_batch, update_contact_phase_list
from tests.test_contacts_cpu import _random_walks

pytestmark = pytest.mark.gpu


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def test_device_contact_kernels_match_the_oracles():
    """All three device kernels of row 8f-1 / 8a-4 against NON-product code: the merge against oracle/contacts_ref.py (the
    restatement of the reference's updateContactPhaseList, incl. its `return false`), the sampling and the step adjustment
    against oracle/schedule_ref.py (integer-nanosecond restatement of the rule) -- Gamma, landing knots, list lengths and
    times bit-exact, positions exact, R to 3e-7.  The host entry points of the C ABI are then held to the device results
    (bit-exact: both run the same __host__ __device__ functions)."""
    import torch
    from tests.test_contacts_cpu import (assert_adjust_matches_schedule_oracle, assert_merge_matches_contacts_oracle,
                                         assert_sample_matches_schedule_oracle)
    cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
    L = cm.Layout(cfg.N)
    B = 96
    lists = _random_walks(cfg, B, 17)
    lists[1] = cm.rollout.walking_plan(cfg)       # every time a multiple of dT: knots fall exactly on activations
    plan = pack_lists(cfg, lists, max_contacts=12)
    mpc = (plan[0].copy(), plan[1].copy(), plan[2].copy())
    mpc[1][..., :3] += np.random.default_rng(2).uniform(-0.01, 0.01, mpc[1][..., :3].shape).astype(np.float32)
    plan[0][5] += 50.0       # one problem where the reference's function returns false (a stance foot the planner does not know)
    s = cm.BatchSolver(cfg, B)
    dev = lambda t: tuple(torch.from_numpy(a).cuda() for a in t)
    host = lambda t: tuple(a.cpu().numpy() for a in t)
    now = 0.06 * 9
    out_d, dok = s.contacts_merge_device(now, dev(plan), dev(mpc))
    torch.cuda.synchronize()
    (dt_, dp_, dn_), ok_d = host(out_d), dok.cpu().numpy()
    assert_merge_matches_contacts_oracle(cfg, now, plan, mpc, (dt_, dp_, dn_), ok_d)
    assert not ok_d[5] and ok_d.sum() == B - 1 and dn_[5].min() == 0
    (ot, op, on), ok = update_contact_phase_list(now, plan, mpc)            # host C entry point == device kernel
    np.testing.assert_array_equal(ok_d.astype(bool), ok)
    good = np.where(ok)[0]
    np.testing.assert_array_equal(dn_[good], on[good])
    for b in good:
        for c in range(2):
            m = on[b, c]
            np.testing.assert_array_equal(dt_[b, c, :m], ot[b, c, :m])
            np.testing.assert_array_equal(dp_[b, c, :m], op[b, c, :m])
    # sampling of the merged lists, the failed problem (an empty list) included: it must be left alone and flagged
    dP = torch.full((B, L.np), 3.0, dtype=torch.float32, device="cuda")
    land = s.contacts_sample_device(now, out_d, dP)
    torch.cuda.synchronize()
    Ph, lh = dP.cpu().numpy(), land.cpu().numpy()
    assert_sample_matches_schedule_oracle(cfg, dt_, dp_, dn_, now, Ph, lh, rows=good)
    assert (Ph[good][:, L.p_com0:] == 3.0).all()                         # state, reference and wrench rows are not the sampler's
    empty = [c for c in range(2) if dn_[5, c] == 0]
    assert empty and all(lh[5, c] == -2 for c in empty)
    for c in empty:
        assert (Ph[5, L.p_R[c]:L.p_cur[c] + 3] == 3.0).all()
    ref, rland = sample_schedule_batch(cfg, dt_[good], dp_[good], dn_[good], now)    # (and the numpy mirror agrees)
    np.testing.assert_array_equal(lh[good], rland)
    # step adjustment
    X = np.random.default_rng(0).normal(size=(B, L.nx)).astype(np.float32)
    before = dp_.copy()
    s.contacts_adjust_device(now, torch.from_numpy(X).cuda(), land, out_d)
    torch.cuda.synchronize()
    after = out_d[1].cpu().numpy()
    assert_adjust_matches_schedule_oracle(cfg, dt_, before, after, dn_, now, X, lh, rows=good)
    for c in empty:
        np.testing.assert_array_equal(after[5, c], before[5, c])          # a foot without a list is not touched
    hp = before[good].copy()
    assert s._lib.cmpc_contacts_adjust(cfg.N, len(good), 12, now, _ptr(np.ascontiguousarray(X[good])), _ptr(np.ascontiguousarray(lh[good])),
                                       _ptr(np.ascontiguousarray(dt_[good])), _ptr(hp), _ptr(np.ascontiguousarray(dn_[good]))) == 0
    np.testing.assert_array_equal(after[good], hp)
    assert np.abs(after - before).max() > 0
    # list lengths outside 0..M never reach the lists: empty result, ok = 0
    bad_n = plan[2].copy(); bad_n[7, 0] = 13; bad_n[8, 1] = -1
    out_b, okb = s.contacts_merge_device(now, (dev(plan)[0], dev(plan)[1], torch.from_numpy(bad_n).cuda()), dev(mpc))
    torch.cuda.synchronize()
    okb, nb = okb.cpu().numpy(), out_b[2].cpu().numpy()
    assert okb[7] == 0 and okb[8] == 0 and nb[7, 0] == 0 and nb[8, 1] == 0 and okb.sum() == B - 3


def test_rollout_aborts_the_tick_like_the_reference_when_the_merge_fails():
    """updateContactPhaseList returning false makes the reference abort the tick (CentroidalMPCBlock.cpp:603-607); the
    roll-out stops there too instead of sampling an empty list."""
    import torch
    cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
    B = 8
    ro = cm.rollout.WalkingRollout(cfg, B)
    t = ro.plan[0].clone()
    t[3, 0] += 100.0          # from tick 2 on the planner of problem 3 no longer knows the left foot's current contact
    com0 = np.tile([0.0, 0.0, 0.7], (B, 1)); z = np.zeros((B, 3))
    rec = ro.run(5, com0, z, z, replan={2: (t, ro.plan[1], ro.plan[2])})
    assert rec.get("aborted_tick") == 2 and rec["merge_ok"] == [True, True, False]
    # the per-tick records stay aligned: the aborted tick has its (failed) entry in every list
    assert rec["converged"] == [True, True, False] and len(rec["tick_ms"]) == len(rec["iterations_max"]) == len(rec["unconverged"]) == 3


def test_walking_rollout_stays_in_hbm_and_on_its_feet():
    cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
    B, ticks = 64, 24
    rng = np.random.default_rng(5)
    com0 = np.array([0.0, 0.0, 0.7]) + rng.uniform(-0.01, 0.01, (B, 3))
    dcom0 = rng.uniform(-0.05, 0.05, (B, 3))
    h0 = rng.uniform(-0.02, 0.02, (B, 3))
    push = np.zeros((B, 3))
    push[:, :2] = rng.uniform(-20.0, 20.0, (B, 2)) / cm.synthetic.ROBOT_MASS
    ro = cm.rollout.WalkingRollout(cfg, B)
    rec = ro.run(ticks, com0, dcom0, h0, push=push, push_ticks=3)
    assert all(rec["converged"]) and all(rec["merge_ok"])
    # the schedule seen by the MPC follows the class-free clock: left foot lands at 0.84 s = tick 14
    lands = np.array([l[0] for l in rec["land"]])          # problem 0: [ticks, 2]
    assert lands[0, 0] == 14 and lands[5, 0] == 9 and lands[13, 0] == 1     # left foot: landing knot counts down
    assert lands[14, 0] == -1 or lands[14, 0] > 14                         # landed: the next landing is far (or beyond the horizon)
    assert lands[16, 1] > 0                                                # the right foot lifts at 0.96 s = tick 16
    # the CoM follows the reference: height within 3 cm, lateral sway within 8 cm, forward progress bounded by the plan
    com = np.stack(rec["com"])                              # [ticks, B, 3]
    assert np.abs(com[:, :, 2] - 0.7).max() < 0.03
    assert np.abs(com[:, :, 1]).max() < 0.08
    assert com[-1, :, 0].min() > 0.0 and com[-1, :, 0].max() < 0.35
    # every landing position the MPC chose stays inside the bounding box around the nominal footstep
    off = np.stack(rec["landing_offset"])                   # [ticks, B, 2, 3]
    for c in range(2):
        assert (off[:, :, c] <= rec["box_upper"][c] + 2e-5).all() and (off[:, :, c] >= rec["box_lower"][c] - 2e-5).all()
    assert np.abs(off).max() > 1e-4                         # and the pushes do move it
    # warm starts: the ticks after the first need fewer iterations than the cold one
    it = np.array(rec["iterations_mean"])
    assert it[1:].mean() < it[0], it
    print("iterations per tick (mean):", np.round(it, 2).tolist(), "max:", rec["iterations_max"])


def test_warm_policy_budget_restart_and_launch_level_retry():
    """cmpc_set_warm_policy: a warm-started pass that exhausts its budget is (a) returned with status 1, (b) started again from the
    cold start inside the launch (safeguard word += 10000), or (c) re-solved by the roll-out in a launch of its own -- with the same
    answer as a plain cold solve in (b) and (c).  A budget of 3 iterations makes every problem of a landing tick a "straggler"."""
    import torch
    cfg, P, X0 = cm.synthetic.config3_external_push(64, seed=5)
    L = cm.Layout(cfg.N)
    P32, X032 = P.astype(np.float32), X0.astype(np.float32)
    s = cm.BatchSolver(cfg, 64)
    dP, dX0 = torch.from_numpy(P32).cuda(), torch.from_numpy(X032).cuda()
    Xc, Ic = s.solve_device(dP, dX0)                      # cold reference
    torch.cuda.synchronize()
    Xc, Ic = Xc.cpu().numpy(), Ic.cpu().numpy()
    assert (Ic[:, 5] == 0).all()
    # the cold-start guess handed over as if it were a shifted solution: far from the central path at mu = 1e-2, 3 iterations are not enough
    s.set_warm_policy(3, restart_in_kernel=False)
    Xa, Ia = s.solve_device(dP, dX0, warm=True)
    torch.cuda.synchronize()
    Ia = Ia.cpu().numpy()
    assert (Ia[:, 5] == 1).all() and (Ia[:, 0] == 3).all()
    s.set_warm_policy(3, restart_in_kernel=True)
    Xb, Ib = s.solve_device(dP, dX0, warm=True)
    torch.cuda.synchronize()
    Xb, Ib = Xb.cpu().numpy(), Ib.cpu().numpy()
    assert (Ib[:, 5] == 0).all() and ((Ib[:, 3].astype(np.int64) // 10000) % 10 == 1).all()
    np.testing.assert_array_equal(Ib[:, 0], Ic[:, 0] + 3)            # three warm iterations, then exactly the cold solve
    np.testing.assert_array_equal(Xb, Xc)
    # (c) through the roll-out: ticks after the first are warm; with a budget of 3 and retry="launch" every tick re-solves its stragglers
    B = 32
    rng = np.random.default_rng(8)
    com0 = np.array([0.0, 0.0, 0.7]) + rng.uniform(-0.01, 0.01, (B, 3))
    z = np.zeros((B, 3))
    ro = cm.rollout.WalkingRollout(cfg, B, warm_budget=3, retry="launch", retry_batch=16)
    rec = ro.run(4, com0, z, z, record="light")
    assert all(rec["converged"]) and sum(rec["unconverged"]) == 0 and sum(rec["retried"]) > 0
    ro2 = cm.rollout.WalkingRollout(cfg, B, warm_budget=3, retry=None)
    rec2 = ro2.run(3, com0, z, z, record="light")
    assert sum(rec2["unconverged"]) > 0 and not all(rec2["converged"])


def test_native_tick_is_the_seven_entry_points_chained_bit_for_bit():
    """cmpc_rollout_tick_device (one call per tick) against the same tick as seven calls of the C ABI: the same kernels in the same order on the same
    buffers, so every record of the roll-out -- CoM, ZMP, landing knots, landing offsets, iteration counts -- is identical to the last bit, across the
    push, a lift-off and a landing."""
    cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
    B, ticks = 24, 18
    rng = np.random.default_rng(11)
    com0 = np.array([0.0, 0.0, 0.7]) + rng.uniform(-0.01, 0.01, (B, 3))
    dcom0 = rng.uniform(-0.05, 0.05, (B, 3))
    h0 = rng.uniform(-0.02, 0.02, (B, 3))
    push = np.zeros((B, 3))
    push[:, :2] = rng.uniform(-20.0, 20.0, (B, 2)) / cm.synthetic.ROBOT_MASS
    recs = []
    for native in (True, False):
        ro = cm.rollout.WalkingRollout(cfg, B, native_tick=native)
        assert ro.native_tick == native
        recs.append(ro.run(ticks, com0, dcom0, h0, push=push, push_ticks=3))
    a, b = recs
    assert all(a["converged"]) and all(a["merge_ok"]) and len(a["com"]) == ticks
    for key in ("com", "zmp", "land", "landing_offset"):
        assert np.array_equal(np.stack(a[key]), np.stack(b[key])), key
    assert a["iterations_max"] == b["iterations_max"] and a["iterations_mean"] == b["iterations_mean"]
    lands = np.stack(a["land"])
    assert len(np.unique(lands)) > 3                        # the window crossed swing and double-support phases: the landing knots moved through the horizon


def test_native_tick_rejects_aliased_lists_and_null_buffers():
    import ctypes
    import torch
    CmpcTickIO = cm._capi.CmpcTickIO
    cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
    s = cm.BatchSolver(cfg, 4)
    io = CmpcTickIO()
    assert s._lib.cmpc_rollout_tick_device(s._h, 3, 0.0, 1, ctypes.byref(io), None) != 0 and "dLand" in s.last_error
    t = torch.zeros((4, 2, 3, 2), dtype=torch.float64, device="cuda")
    scratch = torch.zeros((4, 8), dtype=torch.float32, device="cuda")
    io.dLand = io.dInfo = scratch.data_ptr()
    io.dPrevT = io.dListT = t.data_ptr()
    assert s._lib.cmpc_rollout_tick_device(s._h, 3, 0.0, 1, ctypes.byref(io), None) != 0 and "alias" in s.last_error
    assert s._lib.cmpc_rollout_tick_device(s._h, 3, 0.0, 1, None, None) != 0
