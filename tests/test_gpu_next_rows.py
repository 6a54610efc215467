"""GPU parity of the rows next to the solve (SURVEY 8f-3 reference resampling, 8f-4 closed-loop plant step) against
their float64 restatements in oracle/plant_ref.py."""
import numpy as np
import pytest

import cmpc_amd as cm

pytestmark = pytest.mark.gpu


def test_reference_resampling_matches_restatement():
    from oracle import plant_ref
    cfg = cm.config.ergocub_gazebo_v1()
    B, M, N = 4, 75, cfg.N
    rng = np.random.default_rng(3)
    com_in = np.cumsum(rng.normal(scale=0.005, size=(B, M, 3)), axis=1) + [0, 0, 0.72]
    h_in = rng.normal(scale=2.0, size=(B, M, 3))
    mpc = cm.CentroidalMPC(batch=B)
    assert mpc.initialize(cfg)
    assert mpc.set_reference_from_planner(com_in, h_in, in_dt=0.02, t_offset=0.04, robot_mass=56.0, com_height=0.7)
    # read back through a solve-free path: set the rest of the problem and look at what the solver was given
    _, P, X0 = cm.synthetic.config2_perturbed_com(B)
    L = cm.Layout(N)
    assert mpc.set_state(P[:, L.p_com0:L.p_com0 + 3], P[:, L.p_dcom0:L.p_dcom0 + 3], P[:, L.p_h0:L.p_h0 + 3])
    from cmpc_amd.synthetic import _standing_lists
    assert mpc.set_contact_phase_list(_standing_lists(cfg, 2.0))
    assert mpc.advance(), mpc.last_error
    X, _ = mpc.get_solution()
    for b in range(B):
        cr, hr = plant_ref.resample_references(com_in[b], h_in[b], 0.02, 0.04, N, cfg.sampling_time, 56.0, 0.7)
        # the optimum tracks the resampled references: CoM z reference 0.7 everywhere, and a second solve fed the
        # restated references directly must give the same answer
        P2 = P[b:b + 1].copy()
        P2[0, L.p_comref:L.p_comref + 3 * (N + 1)] = cr.reshape(-1)
        P2[0, L.p_href:L.p_href + 3 * (N + 1)] = hr.reshape(-1)
        s = cm.BatchSolver(cfg, 1)
        X2, info, rc = s.solve_host(P2, X0[b:b + 1])
        assert rc == 0
        np.testing.assert_allclose(X[b], X2[0], rtol=0, atol=5e-5)


def test_plant_step_matches_restatement():
    import torch
    from oracle import plant_ref
    cfg, P, X0 = cm.synthetic.config3_external_push(32)
    P32, X032 = P.astype(np.float32), X0.astype(np.float32)
    L = cm.Layout(cfg.N)
    s = cm.BatchSolver(cfg, 32)
    dP, dX0 = torch.from_numpy(P32).cuda(), torch.from_numpy(X032).cuda()
    dX, dInfo = s.solve_device(dP, dX0)
    state = torch.from_numpy(P32[:, L.p_com0:L.p_com0 + 9].copy()).cuda()
    new_state, zmp = s.plant_step_device(dX, dP, state, step=0.01, substeps=6)
    torch.cuda.synchronize()
    X = dX.cpu().numpy().astype(np.float64)
    corners = np.asarray([c.corners for c in cfg.contacts])
    for b in range(32):
        ref_state, ref_zmp = plant_ref.plant_step(L, corners, X[b], P32[b].astype(np.float64), P32[b, L.p_com0:L.p_com0 + 9], 0.01, 6)
        np.testing.assert_allclose(new_state[b].cpu().numpy(), ref_state, rtol=0, atol=2e-6)
        np.testing.assert_allclose(zmp[b].cpu().numpy(), ref_zmp, rtol=0, atol=2e-6)
    # one MPC period of the plant reproduces the MPC's own prediction of knot 1 up to the integration scheme
    pred = np.concatenate([L.x_com(X)[:, 1], L.x_dcom(X)[:, 1]], axis=1)
    got = new_state.cpu().numpy()[:, :6]
    assert np.abs(got[:, 3:6] - pred[:, 3:6]).max() < 1e-4      # velocities: forces are constant over the knot -> exact
    assert np.abs(got[:, 0:3] - pred[:, 0:3]).max() < 5e-3      # positions: explicit Euler in the MPC vs RK4 here


def test_closed_loop_rollout_stays_upright():
    """Ten MPC ticks of solve -> plant -> feedback on the device, pushes included: the CoM height stays near its
    reference and the ZMP stays inside the support polygon's bounding box."""
    import torch
    cfg, P, X0 = cm.synthetic.config2_perturbed_com(64)
    L = cm.Layout(cfg.N)
    s = cm.BatchSolver(cfg, 64)
    dP = torch.from_numpy(P.astype(np.float32)).cuda()
    dX0 = torch.from_numpy(X0.astype(np.float32)).cuda()
    state = dP[:, L.p_com0:L.p_com0 + 9].clone()
    for tick in range(10):
        dP[:, L.p_com0:L.p_com0 + 9] = state
        dX, dInfo = s.solve_device(dP, dX0)
        assert (dInfo[:, 5] == 0).all()
        state, zmp = s.plant_step_device(dX, dP, state, step=0.01, substeps=6)
        dX0 = dX.clone()
    torch.cuda.synchronize()
    st = state.cpu().numpy()
    assert np.abs(st[:, 2] - 0.7).max() < 0.02 and np.abs(st[:, :2]).max() < 0.05
    z = zmp.cpu().numpy()
    assert (np.abs(z[:, 0]) <= 0.08 + 1e-6).all() and (np.abs(z[:, 1]) <= 0.08 + 0.01 + 1e-6).all()


def test_warm_start_saves_iterations_in_receding_horizon():
    """is_warm_start_enabled (ergoCubGazeboV1/centroidal_mpc.ini:9): ticks started from the previous solution shifted
    by one knot, with the barrier started at 1e-2 instead of 0.1, need fewer iterations than cold ticks and reach the
    same optimum."""
    import torch
    from cmpc_amd.synthetic import _standing_lists
    from tests import parity
    B = 32
    cfg, P, X0 = cm.synthetic.config2_perturbed_com(B)
    L, N = cm.Layout(cfg.N), cfg.N
    its = {}
    sols = {}
    for mode in ("cold", "warm"):
        mpc = cm.CentroidalMPC(batch=B)
        assert mpc.initialize(cfg)
        state = P[:, L.p_com0:L.p_com0 + 9].astype(np.float32).copy()
        dP = torch.from_numpy(P.astype(np.float32)).cuda()
        it = []
        for tick in range(4):
            assert mpc.set_state(state[:, 0:3], state[:, 3:6], state[:, 6:9])
            assert mpc.set_reference_trajectory(P[:, L.p_comref:L.p_comref + 3 * (N + 1)], P[:, L.p_href:L.p_href + 3 * (N + 1)])
            assert mpc.set_contact_phase_list(_standing_lists(cfg, 10.0))
            if tick > 0 and mode == "warm":
                assert mpc.set_initial_guess(None, shift_previous=True)
            assert mpc.advance(), mpc.last_error
            X, info = mpc.get_solution()
            it.append(float(info[:, 0].mean()))
            st, _ = mpc._solver.plant_step_device(torch.from_numpy(X).cuda(), dP, torch.from_numpy(state).cuda(), step=0.01, substeps=6)
            torch.cuda.synchronize()
            state = st.cpu().numpy()
        its[mode], sols[mode] = it, X
    assert np.mean(its["warm"][1:]) < np.mean(its["cold"][1:]) - 0.5, its
    for b in range(B):  # same closed-loop trajectory either way
        e = parity.errors(cfg.N, P[b], sols["warm"][b], sols["cold"][b])
        assert e["com"] < 1e-4 and e["force0"] < 1e-4, e


@pytest.mark.parametrize("com_height", [0.7, None])
def test_device_reference_resampling_matches_the_oracle_tensor_for_tensor(com_height):
    """cmpc_write_reference_from_planner_device (8f-3 on the device, CentroidalMPCBlock.cpp:525-577) against oracle/plant_ref.resample_references: the comRef / hRef rows of
    dP themselves, 1e-6, including offsets before the trajectory's start and beyond its end (clamped) and the planner's own CoM height (NaN = None)."""
    import torch
    from oracle import plant_ref
    cfg = cm.config.ergocub_gazebo_v1()
    B, M, N = 6, 40, cfg.N
    L = cm.Layout(N)
    rng = np.random.default_rng(8)
    com_in = (np.cumsum(rng.normal(scale=0.005, size=(B, M, 3)), axis=1) + [0, 0, 0.72]).astype(np.float32)
    h_in = rng.normal(scale=2.0, size=(B, M, 3)).astype(np.float32)
    s = cm.BatchSolver(cfg, B)
    for t_offset in (-0.05, 0.04, 0.31, 0.9):       # 0.9 + 20 * 0.06 s runs past the 0.78 s the trajectory covers
        dP = torch.full((B, L.np), 7.0, dtype=torch.float32, device="cuda")
        s.write_reference_from_planner_device(torch.from_numpy(com_in).cuda(), torch.from_numpy(h_in).cuda(), 0.02, t_offset, 56.0, com_height, dP)
        torch.cuda.synchronize()
        P = dP.cpu().numpy()
        for b in range(B):
            cr, hr = plant_ref.resample_references(com_in[b].astype(np.float64), h_in[b].astype(np.float64), 0.02, t_offset, N, cfg.sampling_time, 56.0,
                                                   com_height if com_height is not None else float("nan"))
            np.testing.assert_allclose(P[b, L.p_comref:L.p_comref + 3 * (N + 1)], cr.reshape(-1), rtol=0, atol=1e-6)
            np.testing.assert_allclose(P[b, L.p_href:L.p_href + 3 * (N + 1)], hr.reshape(-1), rtol=0, atol=1e-6)
        rest = np.ones(L.np, bool); rest[L.p_comref:L.p_href + 3 * (N + 1)] = False
        assert (P[:, rest] == 7.0).all()             # nothing else of dP is touched
