"""CPU tests of the host side: configuration reader, layout, contact-schedule sampling, synthetic
generators, and that the C-ABI library loads and exports every symbol include/cmpc.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

import cmpc_amd as cm
from cmpc_amd.contacts import PlannedContact, sample_schedule

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INI = "/root/reference/src/centroidal-mpc-walking/config/robots"

ERGOCUB_INI = """
linear_solver                   "ma97"
sampling_time                   0.06
time_horizon                    1.2
number_of_maximum_contacts      2
number_of_slices                1
static_friction_coefficient     0.33
is_warm_start_enabled           true
ipopt_tolerance                 1e-4
#ipopt_max_iteration              14
com_weight                     (10.0, 10.0, 200.0)
contact_position_weight         2e3
force_rate_of_change_weight    (10.0, 10.0, 10.0)
angular_momentum_weight         1e2
contact_force_symmetry_weight   100.0
[CONTACT_0]
number_of_corners         4
contact_name              "left_foot"
corner_0                   (0.08, 0.01, 0.0)
corner_1                   (0.08, -0.01, 0.0)
corner_2                   (-0.08, -0.01, 0.0)
corner_3                   (-0.08 0.01, 0.0)
bounding_box_upper_limit   (0.01, 0.05, 0.0)
bounding_box_lower_limit   (-0.01, -0.00, 0.0)
[CONTACT_1]
number_of_corners         4
contact_name              "right_foot"
corner_0                   (0.08, 0.01, 0.0)
corner_1                   (0.08, -0.01, 0.0)
corner_2                   (-0.08, -0.01, 0.0)
corner_3                   (-0.08 0.01, 0.0)
bounding_box_upper_limit   (0.01, 0.00, 0.0)
bounding_box_lower_limit   (-0.01, -0.05, 0.0)
"""


def test_ini_reader_matches_preset():
    cfg = cm.config.from_ini(ERGOCUB_INI)  # same keys as ergoCubGazeboV1/centroidal_mpc.ini:3-42
    ref = cm.config.ergocub_gazebo_v1()
    assert cfg.N == 20 and cfg.sampling_time == 0.06
    for k in ("com_weight", "contact_position_weight", "force_rate_of_change_weight", "angular_momentum_weight",
              "contact_force_symmetry_weight", "static_friction_coefficient"):
        assert getattr(cfg, k) == getattr(ref, k), k
    assert [c.contact_name for c in cfg.contacts] == ["left_foot", "right_foot"]
    assert cfg.contacts[0].corners == ref.contacts[0].corners
    assert cfg.contacts[1].bounding_box_lower_limit == (-0.01, -0.05, 0.0)


@pytest.mark.skipif(not os.path.isdir(REF_INI), reason="reference tree not present")
@pytest.mark.parametrize("robot,N", [("ergoCubGazeboV1", 20), ("ergoCubGazeboV1_1", 12), ("ergoCubSN000", 13),
                                     ("ergoCubSN001", 22), ("iCubGazeboV3", 15)])
def test_reads_the_reference_ini_files(robot, N):
    cfg = cm.config.from_ini(open(os.path.join(REF_INI, robot, "centroidal_mpc.ini")).read())
    assert cfg.N == N
    assert len(cfg.contacts) == 2 and all(len(c.corners) == 4 for c in cfg.contacts)


def test_layout_sizes():
    for N, nx, npar in ((10, 465, 527), (12, 555, 627), (20, 915, 1027), (30, 1365, 1527)):
        L = cm.Layout(N)
        assert (L.nx, L.np, L.ng) == (nx, npar, 53 * N + 15)
    L = cm.Layout(12)  # offsets quoted in SURVEY 8a-NLP
    assert (L.com, L.dcom, L.h, L.pos[0], L.vel[0], L.f[0], L.pos[1], L.vel[1], L.f[1]) == \
        (0, 39, 78, 117, 156, [192, 228, 264, 300], 336, 375, [411, 447, 483, 519])


def test_schedule_sampling_rule():
    cfg = cm.config.ergocub_gazebo_v1()
    dt = cfg.sampling_time
    lists = {"left_foot": [PlannedContact(-1, 6 * dt, (0, 0.08, 0)), PlannedContact(14 * dt, 99, (0.1, 0.08, 0), yaw=0.3)],
             "right_foot": [PlannedContact(-1, 99, (0, -0.08, 0))]}
    s = sample_schedule(cfg, lists)
    np.testing.assert_array_equal(s["enabled"][0], [1] * 6 + [0] * 8 + [1] * 6)
    np.testing.assert_array_equal(s["enabled"][1], [1] * 20)
    # rows of swing stages and of the new stance are owned by the next contact; z limits form an equality row
    np.testing.assert_allclose(s["nominal"][0, :7], np.tile([0, 0.08, 0], (7, 1)))
    np.testing.assert_allclose(s["nominal"][0, 7:], np.tile([0.1, 0.08, 0], (14, 1)))
    assert abs(s["R"][0, 10, 0, 1] + np.sin(0.3)) < 1e-12 and s["R"][0, 3, 0, 1] == 0
    assert (s["upper"][..., 2] == 0).all() and (s["lower"][..., 2] == 0).all()
    np.testing.assert_allclose(s["current"][0], [0, 0.08, 0])


def test_pack_parameters_uses_reference_layout():
    cfg, P, X0 = cm.synthetic.config3_external_push(4)
    L = cm.Layout(cfg.N)
    assert P.shape == (4, L.np) and X0.shape == (4, L.nx)
    R0 = P[0, L.p_R[0]:L.p_R[0] + 9].reshape(3, 3, order="F")   # vec(R) is column-major (tmp.c layout)
    np.testing.assert_allclose(R0, np.eye(3))
    assert (P[:, L.p_gam[0]:L.p_gam[0] + cfg.N] == np.array([1] * 6 + [0] * 8 + [1] * 6)).all()
    np.testing.assert_allclose(L.x_force(X0, 1, 2)[..., 2], 9.80665 / 8)
    push = P[:, L.p_fext:L.p_fext + 3 * cfg.N].reshape(4, cfg.N, 3)
    assert (np.abs(push[:, :4, :2]) <= 50 / 56.0).all() and (push[:, 4:] == 0).all() and (push[..., 2] == 0).all()


def test_synthetic_configs_are_seeded():
    a = cm.synthetic.config2_perturbed_com(8)[1]
    b = cm.synthetic.config2_perturbed_com(8)[1]
    np.testing.assert_array_equal(a, b)
    cfg5, P5, _ = cm.synthetic.config5_footstep_candidates(3)
    assert cfg5.N == 30 and P5.shape == (3, 1527)


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "cmpc.h")).read()
    declared = set(re.findall(r"\b(cmpc_[a-z_]+)\s*\(", hdr))
    assert declared == set(cm._capi.EXPORTS), declared ^ set(cm._capi.EXPORTS)
    assert os.path.exists(cm._capi.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(cm._capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_streaming_stage_roles_hold_the_same_number_of_barriers():
    """The factorising wave and the consumer waves of the streaming backward stage run their stage loops in two different out-of-line functions
    and meet at workgroup barriers: unequal counts would hang the workgroup (a GPU reset for everyone on the node), not fail a test.  Both loops are
    written with one loop-header macro and hold one barrier per trip plus one before the loop; the library counts them on that skeleton (host code)."""
    lib = cm._capi.lib()
    for N in (2, 10, 12, 13, 15, 20, 22, 30, 40):
        for k0 in sorted({0, 1, max(N - 3, 0), N - 1}):
            a, b = lib.cmpc_sq_pass_barriers(N, k0, 0), lib.cmpc_sq_pass_barriers(N, k0, 1)
            assert a == b == 1 + N - k0, (N, k0, a, b)


def test_lds_images_fit_the_cu():
    """The LDS image the launcher asks for (host function, no GPU): the resident variants -- whole problem and the working sets of the streaming stage in LDS,
    one workgroup per CU -- fit the 160 KiB of a CU for every horizon the reference ships a configuration for (10..22); the HBM-factor image fits THREE times
    at the bench horizons (three workgroups per CU is what cmpc_create assumes above the CU count)."""
    lib = ctypes.CDLL(cm._capi.LIB_PATH)
    lib.cmpc_solver_lds_bytes.restype = ctypes.c_size_t
    lib.cmpc_solver_lds_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    for N in (10, 12, 13, 15, 20, 22):
        assert lib.cmpc_solver_lds_bytes(N, 0) <= 160 * 1024, (N, lib.cmpc_solver_lds_bytes(N, 0))
    for N in (20, 30):
        assert 3 * lib.cmpc_solver_lds_bytes(N, 1) <= 160 * 1024, (N, lib.cmpc_solver_lds_bytes(N, 1))
    assert lib.cmpc_solver_lds_bytes(30, 0) > 160 * 1024   # (N = 30 never runs resident: cmpc_create falls back to the HBM-factor variant)


def test_dims_and_sparsity_need_no_gpu():
    lib = cm._capi.lib()
    v = [ctypes.c_int() for _ in range(5)]
    assert lib.cmpc_dims(20, *[ctypes.byref(a) for a in v]) == 0
    assert [a.value for a in v] == [915, 1027, 1075, 4875, 6924]
    jr = np.empty(2931, np.int32); jc = np.empty(2931, np.int32); hr = np.empty(4140, np.int32); hc = np.empty(4140, np.int32)
    assert lib.cmpc_nlp_sparsity(12, jr.ctypes.data, jc.ctypes.data, hr.ctypes.data, hc.ctypes.data) == 0
    d = np.load(os.path.join(ROOT, "tests", "golden", "nlp_tmp.npz"))
    np.testing.assert_array_equal(jr, d["jac_row"])      # casadi_s5 (tmp.c:67)
    np.testing.assert_array_equal(hr, d["hess_row"])     # casadi_s4 (tmp.c:66)
    assert (np.diff(jc) >= 0).all() and (np.diff(hc) >= 0).all()


def test_product_fails_loudly_without_gpu_or_library():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        cm.BatchSolver(cm.config.ergocub_gazebo_v1(), 4)
    mpc = cm.CentroidalMPC(batch=1)
    assert not mpc.initialize(cm.config.ergocub_gazebo_v1()) and "no HIP device" in mpc.last_error


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "paper_romualdi_2022_icra_centroidal-mpc-walking_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libcmpc_oracle" not in txt, f
