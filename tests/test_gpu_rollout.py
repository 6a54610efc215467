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


def test_device_contact_kernels_match_the_host_entry_points():
    import torch
    cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
    L = cm.Layout(cfg.N)
    B = 96
    lists = _random_walks(cfg, B, 17)
    plan = pack_lists(cfg, lists, max_contacts=12)
    mpc = (plan[0].copy(), plan[1].copy(), plan[2].copy())
    mpc[1][..., :3] += np.random.default_rng(2).uniform(-0.01, 0.01, mpc[1][..., :3].shape).astype(np.float32)
    plan[0][5] += 50.0       # one problem where the reference's function returns false (a stance foot the planner does not know)
    s = cm.BatchSolver(cfg, B)
    dev = lambda t: tuple(torch.from_numpy(a).cuda() for a in t)
    now = 0.06 * 9
    (ot, op, on), ok = update_contact_phase_list(now, plan, mpc)
    (dt_, dp_, dn_), dok = s.contacts_merge_device(now, dev(plan), dev(mpc))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dok.cpu().numpy().astype(bool), ok)
    assert not ok[5] and ok.sum() == B - 1
    good = np.where(ok)[0]
    np.testing.assert_array_equal(dn_.cpu().numpy()[good], on[good])
    for b in good:
        for c in range(2):
            m = on[b, c]
            np.testing.assert_array_equal(dt_.cpu().numpy()[b, c, :m], ot[b, c, :m])
            np.testing.assert_array_equal(dp_.cpu().numpy()[b, c, :m], op[b, c, :m])
    # sampling: device kernel == numpy mirror == C host function
    lists_ok = (ot[good[:64]], op[good[:64]], on[good[:64]])
    s2 = cm.BatchSolver(cfg, 64)
    dP = torch.full((64, L.np), 3.0, dtype=torch.float32, device="cuda")
    land = s2.contacts_sample_device(now, dev(lists_ok), dP)
    torch.cuda.synchronize()
    ref, rland = sample_schedule_batch(cfg, *lists_ok, now)
    z = np.zeros((64, 3))
    Pref = cm.pack_parameters(cfg.N, ref["R"], ref["upper"], ref["lower"], ref["enabled"], ref["nominal"], ref["current"], z, z, z,
                              np.zeros((64, cfg.N + 1, 3)), np.zeros((64, cfg.N + 1, 3)), dtype=np.float32)
    np.testing.assert_array_equal(land.cpu().numpy(), rland)
    np.testing.assert_allclose(dP.cpu().numpy()[:, :L.p_com0], Pref[:, :L.p_com0], atol=1e-7)
    assert (dP.cpu().numpy()[:, L.p_com0:] == 3.0).all()
    # step adjustment
    X = np.random.default_rng(0).normal(size=(64, L.nx)).astype(np.float32)
    dl = dev(lists_ok)
    s2.contacts_adjust_device(now, torch.from_numpy(X).cuda(), land, dl)
    torch.cuda.synchronize()
    hp = lists_ok[1].copy()
    assert s2._lib.cmpc_contacts_adjust(cfg.N, 64, 12, now, _ptr(X), _ptr(rland), _ptr(lists_ok[0]), _ptr(hp), _ptr(lists_ok[2])) == 0
    np.testing.assert_array_equal(dl[1].cpu().numpy(), hp)
    assert np.abs(hp - lists_ok[1]).max() > 0


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
