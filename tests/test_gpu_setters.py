"""The class-shaped setters of the C ABI, tested on what they WRITE -- the handle's parameter
vector p, read back with cmpc_get_parameters / cmpc_get_parameters_device -- against the oracles, tensor against tensor:
the reference's setState / setReferenceTrajectory / setContactPhaseList fill CasADi's p the same way
(CentroidalMPCBlock.cpp:407, :579, :609; resampling :201-263, :525-577).  Rounds 1-3 tested them only through the solution
they led to, which an optimum can absorb."""
import ctypes as C

import numpy as np
import pytest

import cmpc_amd as cm
from cmpc_amd.contacts import pack_lists

pytestmark = pytest.mark.gpu


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _read_back(s, L, B):
    """(host copy, device copy) of the handle's parameter set"""
    import torch
    Ph = np.full((B, L.np), np.nan, np.float32)
    assert s._lib.cmpc_get_parameters(s._h, _ptr(Ph)) == 0, s.last_error
    dptr = C.c_void_p()
    assert s._lib.cmpc_get_parameters_device(s._h, C.byref(dptr)) == 0, s.last_error
    Pd = torch.empty((B, L.np), dtype=torch.float32, device="cuda")
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(C.c_void_p(Pd.data_ptr()), dptr, Ph.nbytes, 3) == 0          # hipMemcpyDeviceToDevice
    torch.cuda.synchronize()
    return Ph, Pd.cpu().numpy()


def test_setters_write_the_parameter_vector_the_oracles_describe():
    from oracle import plant_ref, problem_nlp
    from tests.test_contacts_cpu import _random_walks, assert_sample_matches_schedule_oracle
    cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
    B, N = 12, cfg.N
    L = cm.Layout(N)
    rng = np.random.default_rng(5)
    s = cm.BatchSolver(cfg, B)
    lib = s._lib
    # 8a-2 setState: com0 | dcom0 | h0 and the per-knot wrench
    state = rng.normal(size=(B, 9)).astype(np.float32)
    wrench = rng.normal(size=(B, N, 6)).astype(np.float32)
    assert lib.cmpc_set_state(s._h, _ptr(state), _ptr(wrench)) == 0
    # 8a-3 setReferenceTrajectory
    com_ref = rng.normal(size=(B, N + 1, 3)).astype(np.float32)
    h_ref = rng.normal(size=(B, N + 1, 3)).astype(np.float32)
    assert lib.cmpc_set_reference(s._h, _ptr(com_ref), _ptr(h_ref)) == 0
    # 8a-4 setContactPhaseList from lists (the C entry point samples them itself)
    lists = _random_walks(cfg, B, 23)
    t, pose, n = pack_lists(cfg, lists, max_contacts=12)
    up = np.asarray([c.bounding_box_upper_limit for c in cfg.contacts], np.float32)
    lo = np.asarray([c.bounding_box_lower_limit for c in cfg.contacts], np.float32)
    land = np.zeros((B, 2), np.int32)
    now = 0.06 * 4
    assert lib.cmpc_set_contact_lists(s._h, 12, now, _ptr(t), _ptr(pose), _ptr(n), _ptr(up), _ptr(lo), _ptr(land)) == 0, s.last_error
    Ph, Pd = _read_back(s, L, B)
    np.testing.assert_array_equal(Ph, Pd)                      # what cmpc_advance will solve is what the setters staged
    # the index layout is the reference's (oracle/problem_nlp.py restates tmp.c:62-67): every block where the oracle's layout puts it
    np.testing.assert_array_equal(Ph[:, L.p_com0:L.p_com0 + 9], state)
    np.testing.assert_array_equal(Ph[:, L.p_fext:L.p_fext + 3 * N].reshape(B, N, 3), wrench[..., :3])
    np.testing.assert_array_equal(Ph[:, L.p_text:L.p_text + 3 * N].reshape(B, N, 3), wrench[..., 3:])
    np.testing.assert_array_equal(Ph[:, L.p_comref:L.p_comref + 3 * (N + 1)].reshape(B, N + 1, 3), com_ref)
    np.testing.assert_array_equal(Ph[:, L.p_href:L.p_href + 3 * (N + 1)].reshape(B, N + 1, 3), h_ref)
    assert L.np == 50 * N + 27 and L.p_text + 3 * N == L.np
    assert_sample_matches_schedule_oracle(cfg, t, pose, n, now, Ph, land)
    # the oracle's own reading of p (oracle/problem_nlp.bounds: how CasADi's Opti turns parameters into lbg / ubg) finds the state and the
    # current foot positions in the initial-condition rows and the sampled box limits in the bounding-box rows
    for b in range(B):
        lb, ub = problem_nlp.bounds(cfg, Ph[b].astype(np.float64))
        np.testing.assert_array_equal(lb[:9].astype(np.float32), state[b])
        np.testing.assert_array_equal(lb[9:15].astype(np.float32), np.concatenate([Ph[b, L.p_cur[0]:L.p_cur[0] + 3], Ph[b, L.p_cur[1]:L.p_cur[1] + 3]]))
        o = 15 + 15 * N
        for c in range(2):
            np.testing.assert_array_equal(ub[o:o + 3 * N].astype(np.float32), np.tile(up[c], N))
            np.testing.assert_array_equal(lb[o:o + 3 * N].astype(np.float32), np.tile(lo[c], N))
            o += 19 * N
    # 8f-3 the planner path: linear interpolation at the MPC knots, h / mass, CoM height forced -- against oracle/plant_ref.py, tensors, 1e-6
    M = 75
    com_in = (np.cumsum(rng.normal(scale=0.005, size=(B, M, 3)), axis=1) + [0, 0, 0.72]).astype(np.float32)
    h_in = rng.normal(scale=2.0, size=(B, M, 3)).astype(np.float32)
    for com_height in (0.7, float("nan")):
        assert lib.cmpc_set_reference_from_planner(s._h, _ptr(com_in), _ptr(h_in), M, 0.02, 0.04, 56.0, com_height) == 0
        Ph, Pd = _read_back(s, L, B)
        np.testing.assert_array_equal(Ph, Pd)
        for b in range(B):
            cr, hr = plant_ref.resample_references(com_in[b].astype(np.float64), h_in[b].astype(np.float64), 0.02, 0.04, N, cfg.sampling_time, 56.0, com_height)
            np.testing.assert_allclose(Ph[b, L.p_comref:L.p_comref + 3 * (N + 1)].reshape(N + 1, 3), cr, rtol=0, atol=1e-6)
            np.testing.assert_allclose(Ph[b, L.p_href:L.p_href + 3 * (N + 1)].reshape(N + 1, 3), hr, rtol=0, atol=1e-6)
        # ... and it touched nothing else
        np.testing.assert_array_equal(Ph[:, L.p_com0:L.p_com0 + 9], state)
        np.testing.assert_array_equal(Ph[:, :L.p_com0], Pd[:, :L.p_com0])
    s.close()


def test_allgather_through_the_c_abi_with_a_one_rank_communicator():
    """cmpc_allgather_compact_device (ncclAllGather of RCCL behind the C ABI) on a communicator of one rank: the gathered block is the local
    compact record.  More than one rank needs more than one GPU: the N > 1 path stays unmeasured on this one-GPU box (bench.py --gpus N)."""
    import torch
    try:
        rccl = C.CDLL("librccl.so")
    except OSError:
        pytest.skip("librccl.so not present")
    cfg, P, X0 = cm.synthetic.config2_perturbed_com(16)
    s = cm.BatchSolver(cfg, 16)
    dP, dX0 = torch.from_numpy(P.astype(np.float32)).cuda(), torch.from_numpy(X0.astype(np.float32)).cuda()
    dX, dInfo = s.solve_device(dP, dX0)
    torch.cuda.synchronize()
    rec = 3 * (cfg.N + 1) + 38
    local = torch.zeros((16, rec), dtype=torch.float32, device="cuda")
    assert s._lib.cmpc_compact_output_device(s._h, dX.data_ptr(), dInfo.data_ptr(), local.data_ptr(), None) == 0
    comm = C.c_void_p()
    devs = (C.c_int * 1)(0)
    rccl.ncclCommInitAll.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    assert rccl.ncclCommInitAll(C.byref(comm), 1, devs) == 0
    gathered = torch.full((1, 16, rec), -7.0, dtype=torch.float32, device="cuda")
    assert s._lib.cmpc_allgather_compact_device(s._h, comm, 1, local.data_ptr(), gathered.data_ptr(), None) == 0, s.last_error
    stream = s._lib.cmpc_stream(s._h)
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    assert hip.hipStreamSynchronize(stream) == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(gathered[0].cpu().numpy(), local.cpu().numpy())
    assert (local[:, -1].cpu().numpy() == 0).all()                    # status column: all converged
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)
    s.close()
