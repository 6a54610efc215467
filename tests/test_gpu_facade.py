"""The C++ drop-in class (include/BipedalLocomotion/ReducedModelControllers/CentroidalMPC.h) linked against
libcmpc_hip.so and driven for 12 ticks across a lift-off exactly the way the reference's block drives it
(examples/facade_demo.cpp = the Block of csrc/facade_check.cpp, shaped after CentroidalMPCBlock.cpp:396-411, 579-631):
no call the reference does not make -- in particular nobody tells the controller the time."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import cmpc_amd as cm

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _replay_through_the_c_abi(ticks):
    """The demo's twelve ticks again, without the class: the same inputs through the C ABI (the entry points the class
    calls, in its order) from Python.  Returns {tick: (forces0[2,4,3], com[N+1,3])}."""
    cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
    N, L = cfg.N, cm.Layout(cfg.N)
    lib = cm._capi.lib()
    h = C.c_void_p()
    ccfg = cm.solver._c_config(cfg)
    assert lib.cmpc_create(C.byref(ccfg), 1, 0, C.byref(h)) == 0
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    sec = lambda ns: float(ns) * 1e-9
    # the planner's list of the demo (nanoseconds, as the class receives them)
    plan_t = np.zeros((1, 2, 2, 2)); plan_p = np.zeros((1, 2, 2, 7), np.float32); plan_p[..., 3] = 1.0
    plan_t[0, 0, 0] = (sec(0), sec(360_000_000)); plan_p[0, 0, 0, :3] = (0.0, 0.08, 0.0)
    plan_t[0, 0, 1] = (sec(840_000_000), sec(100_000_000_000)); plan_p[0, 0, 1, :3] = (0.1, 0.08, 0.0)
    plan_t[0, 1, 0] = (sec(0), sec(100_000_000_000)); plan_p[0, 1, 0, :3] = (0.0, -0.08, 0.0)
    plan_n = np.array([[2, 1]], np.int32)
    up = np.array([c.bounding_box_upper_limit for c in cfg.contacts], np.float32)
    lo = np.array([c.bounding_box_lower_limit for c in cfg.contacts], np.float32)
    com, dcom, ang = np.array([0.01, -0.005, 0.69]), np.array([0.02, 0.0, 0.0]), np.zeros(3)
    com_ref = np.tile(np.array([0.03, -0.02, 0.7], np.float32), (N + 1, 1)); h_ref = np.zeros((N + 1, 3), np.float32)
    prev, have, out = None, False, {}
    for tick in range(ticks):
        now = sec(tick * 60_000_000)
        st = np.concatenate([com, dcom, ang]).astype(np.float32)
        w = np.zeros((N, 6), np.float32)
        assert lib.cmpc_set_state(h, ptr(st), ptr(w)) == 0
        assert lib.cmpc_set_reference(h, ptr(com_ref), ptr(h_ref)) == 0
        if prev is None:
            t, pose, n = plan_t.copy(), plan_p.copy(), plan_n.copy()
        else:       # the block's updateContactPhaseList (the demo checks every tick that cmpc_contacts_merge rebuilds its result)
            t, pose, n = np.zeros_like(plan_t), np.zeros_like(plan_p), np.zeros_like(plan_n)
            ok = np.zeros(1, np.int32)
            assert lib.cmpc_contacts_merge(1, 2, now, ptr(plan_t), ptr(plan_p), ptr(plan_n), ptr(prev[0]), ptr(prev[1]), ptr(prev[2]),
                                           ptr(t), ptr(pose), ptr(n), ptr(ok)) == 0 and ok[0] == 1
        land = np.zeros((1, 2), np.int32)
        assert lib.cmpc_set_contact_lists(h, 2, now, ptr(t), ptr(pose), ptr(n), ptr(up), ptr(lo), ptr(land)) == 0
        assert lib.cmpc_set_initial_guess(h, None, 1 if have else 0) == 0       # is_warm_start_enabled
        assert lib.cmpc_advance(h) == 0, lib.cmpc_last_error(h)
        have = True
        f0 = np.zeros((2, 4, 3), np.float32)
        assert lib.cmpc_get_output(h, ptr(f0), None, None, None) == 0
        x = np.zeros(L.nx, np.float32)
        assert lib.cmpc_get_solution(h, ptr(x), None) == 0
        assert lib.cmpc_contacts_adjust(N, 1, 2, now, ptr(x), ptr(land), ptr(t), ptr(pose), ptr(n)) == 0   # getOutput().contactPhaseList
        prev = (t, pose, n)
        traj = L.x_com(x.astype(np.float64))
        out[tick] = (f0.astype(np.float64), traj, n.copy(), t.copy())
        com = traj[1].copy()
    lib.cmpc_destroy(h)
    return out


def test_facade_links_and_walks(tmp_path):
    pkg = os.path.dirname(cm._capi.LIB_PATH)
    exe = str(tmp_path / "facade_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(pkg, "csrc", "shim"),
                           os.path.join(ROOT, "examples", "facade_demo.cpp"), "-L", pkg, "-lcmpc_hip", f"-Wl,-rpath,{pkg}",
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    ticks = [l.split() for l in out.stdout.splitlines() if l.startswith("tick ")]
    assert len(ticks) == 12
    ncontacts = [int(t[3]) for t in ticks]
    # the left foot lifts at 0.36 s = tick 6: the class's own clock (never set by the caller) must see it
    assert ncontacts[:6] == [2] * 6 and ncontacts[6:] == [1] * 6, ncontacts
    for t in ticks:
        assert abs(float(t[5]) - 9.8) < 1.5          # first-knot forces carry the (unit-mass) weight
        assert abs(float(t[9]) - 0.7) < 0.02 and abs(float(t[7])) < 0.1 and abs(float(t[8])) < 0.1   # the CoM stays put
    vals = {l.split()[0]: [float(x) for x in l.split()[1:]] for l in out.stdout.splitlines() if l.split()[0] in ("total_fz", "next_left", "ticks")}
    nx, ny, nz = vals["next_left"]
    assert abs(nx - 0.1) <= 0.01 + 1e-5 and -1e-5 <= ny - 0.08 <= 0.05 + 1e-5 and abs(nz) < 1e-6

    # ---- the class's numbers ARE the C ABI's numbers: every tick replayed through the entry points the class calls, the 12 (or 24)
    # first-knot force components of the contacts it reports and the whole CoM trajectory compared at 2e-6 (the kernel and the
    # inputs are the same: bit-identical is what is expected, and printed) ----
    replay = _replay_through_the_c_abi(12)
    names = ["left_foot", "right_foot"]
    worst = 0.0
    forces = {}
    for l in out.stdout.splitlines():
        w = l.split()
        if w[0] == "forces":
            forces.setdefault(int(w[1]), {})[w[2]] = np.array([float(x) for x in w[3:]]).reshape(4, 3)
        elif w[0] == "comtraj":
            tick = int(w[1])
            traj = np.array([float(x) for x in w[2:]]).reshape(21, 3)
            worst = max(worst, np.abs(traj - replay[tick][1]).max())
            np.testing.assert_allclose(traj, replay[tick][1], rtol=0, atol=2e-6)
    assert sorted(forces) == list(range(12))
    for tick, fc in forces.items():
        assert sorted(fc) == (names if tick < 6 else ["right_foot"])      # only active contacts are reported (WholeBodyQPBlock.cpp:824)
        for nm, f in fc.items():
            worst = max(worst, np.abs(f - replay[tick][0][names.index(nm)]).max())
            np.testing.assert_allclose(f, replay[tick][0][names.index(nm)], rtol=0, atol=2e-6)
    print("facade vs C ABI replay, 12 ticks, forces + CoM trajectories: max |difference| =", worst)


def test_cpp_monte_carlo_driver_gathers_through_rccl(tmp_path):
    """examples/montecarlo_allgather.cpp -- a host C++ driver over the C ABI, one process per GPU: setters -> cmpc_get_parameters -> cmpc_solve_device ->
    cmpc_compact_output_device -> cmpc_allgather_compact_device (ncclAllGather of RCCL) -- built with hipcc and run as the single rank of a one-rank
    communicator (4096 problems).  W > 1 ranks need W GPUs: unmeasured on this pool's one-GPU lease."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc) or not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("hipcc / rccl.h not present")
    pkg = os.path.dirname(cm._capi.LIB_PATH)
    exe = str(tmp_path / "montecarlo_allgather")
    subprocess.check_call([hipcc, "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "montecarlo_allgather.cpp"),
                           "-L", pkg, "-lcmpc_hip", "-lrccl", f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe, "0", "1", str(tmp_path / "nccl_id"), "4096"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    import re
    assert "4096 gathered," in out.stdout and "0 not converged" in out.stdout, out.stdout
    fz = float(re.search(r"force per problem ([0-9.]+)", out.stdout).group(1))
    assert abs(fz - 9.80665) < 0.2, out.stdout                # (summed over the eight corners) the first-knot forces carry the unit-mass weight
