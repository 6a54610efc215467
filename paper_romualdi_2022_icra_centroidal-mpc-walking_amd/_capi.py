"""ctypes binding of libcmpc_hip.so (include/cmpc.h).  There is no CPU path: if the HIP library is
missing or fails to load, every entry point raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("CMPC_LIB", "libcmpc_hip.so"))   # (CMPC_LIB: developer knob, another build of the same library in the package directory)

INFO = 8


class CmpcConfig(C.Structure):
    _fields_ = [
        ("horizon", C.c_int),
        ("sampling_time", C.c_double),
        ("friction_coefficient", C.c_double),
        ("gravity", C.c_double),
        ("com_weight", C.c_double * 3),
        ("angular_momentum_weight", C.c_double),
        ("contact_position_weight", C.c_double),
        ("force_rate_of_change_weight", C.c_double * 3),
        ("contact_force_symmetry_weight", C.c_double),
        ("corners", C.c_double * 24),
        ("max_iterations", C.c_int),
        ("tolerance", C.c_double),
        ("step_tolerance", C.c_double),
        ("mu_init", C.c_double),
        ("mu_min", C.c_double),
        ("exact_hessian", C.c_int),
        ("final_extrapolation", C.c_int),
        ("tail_stages", C.c_int),
        ("tail_iterations", C.c_int),
        ("tail_trigger", C.c_double),
        ("factor_storage", C.c_int),
    ]


class CmpcTickIO(C.Structure):
    """mirror of cmpc_tick_io (include/cmpc.h): the buffers of one receding-horizon tick, cmpc_rollout_tick_device"""
    _fields_ = [(k, C.c_void_p) for k in (
        "dPlanT", "dPlanPose", "dPlanN", "dPrevT", "dPrevPose", "dPrevN", "dListT", "dListPose", "dListN", "dOk", "dLand",
        "box_upper", "box_lower", "dState", "dWrench", "dP", "dX0", "dX", "dInfo", "dStateOut", "dZmp")] + [
        ("plant_step", C.c_double), ("plant_substeps", C.c_int), ("zmp_half_x", C.c_double), ("zmp_half_y", C.c_double),
        ("dPlanCom", C.c_void_p), ("dPlanH", C.c_void_p), ("plan_knots", C.c_int), ("plan_dt", C.c_double), ("plan_t_offset", C.c_double),
        ("robot_mass", C.c_double), ("com_height", C.c_double)]


FACTORS = {None: 0, "auto": 0, "lds": 1, "hbm": 2}   # cmpc_config.factor_storage


EXPORTS = [
    "cmpc_default_config", "cmpc_dims", "cmpc_create", "cmpc_destroy", "cmpc_last_error",
    "cmpc_batch", "cmpc_stream", "cmpc_solve_device", "cmpc_solve", "cmpc_last_solve_ms", "cmpc_set_timing",
    "cmpc_eval_nlp_device", "cmpc_nlp_sparsity", "cmpc_set_state", "cmpc_set_reference",
    "cmpc_set_contacts", "cmpc_set_initial_guess", "cmpc_advance", "cmpc_get_solution",
    "cmpc_get_output", "cmpc_set_reference_from_planner", "cmpc_plant_step_device", "cmpc_test_poison_lds",
    "cmpc_compact_output_device", "cmpc_contacts_merge", "cmpc_contacts_merge_device", "cmpc_contacts_sample",
    "cmpc_contacts_sample_device", "cmpc_set_contact_lists", "cmpc_contacts_adjust", "cmpc_contacts_adjust_device",
    "cmpc_write_state_device", "cmpc_shift_solution_device", "cmpc_eval_nlp_grad_device", "cmpc_solve_device_warm", "cmpc_set_warm_policy",
    "cmpc_get_parameters", "cmpc_get_parameters_device", "cmpc_allgather_compact_device", "cmpc_sq_pass_barriers",
    "cmpc_rollout_tick_device", "cmpc_write_reference_from_planner_device",
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). This package has no CPU fallback.")
        try:
            # PyTorch-ROCm wheels bundle their own libamdhip64.so.7; import it first so that this
            # library binds to the same HIP runtime (two runtimes in one process cannot share the GPU)
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.c_void_p
        L.cmpc_default_config.argtypes = [C.POINTER(CmpcConfig)]
        L.cmpc_default_config.restype = None
        L.cmpc_dims.argtypes = [C.c_int, ip, ip, ip, ip, ip]
        L.cmpc_create.argtypes = [C.POINTER(CmpcConfig), C.c_int, C.c_int, C.POINTER(vp)]
        L.cmpc_destroy.argtypes = [vp]
        L.cmpc_last_error.argtypes = [vp]
        L.cmpc_last_error.restype = C.c_char_p
        L.cmpc_batch.argtypes = [vp]
        L.cmpc_stream.argtypes = [vp]
        L.cmpc_stream.restype = vp
        L.cmpc_solve_device.argtypes = [vp, fp, fp, fp, fp, vp]
        if hasattr(L, "cmpc_solve_device_warm"):   # (absent from round-2 builds of the library, which tools/ab_bench.sh may load as the baseline)
            L.cmpc_solve_device_warm.argtypes = [vp, fp, fp, fp, fp, vp]
        L.cmpc_solve.argtypes = [vp, fp, fp, fp, fp]
        L.cmpc_last_solve_ms.argtypes = [vp]
        if hasattr(L, "cmpc_set_timing"):
            L.cmpc_set_timing.argtypes = [vp, C.c_int]
        if hasattr(L, "cmpc_set_warm_policy"):
            L.cmpc_set_warm_policy.argtypes = [vp, C.c_int, C.c_int]
        L.cmpc_test_poison_lds.argtypes = [vp]
        L.cmpc_compact_output_device.argtypes = [vp, fp, fp, fp, vp]
        L.cmpc_last_solve_ms.restype = C.c_float
        L.cmpc_eval_nlp_device.argtypes = [vp, fp, fp, fp, C.c_float, fp, fp, fp, fp, fp, vp]
        L.cmpc_nlp_sparsity.argtypes = [C.c_int, vp, vp, vp, vp]
        L.cmpc_set_state.argtypes = [vp, fp, fp]
        L.cmpc_set_reference.argtypes = [vp, fp, fp]
        L.cmpc_set_contacts.argtypes = [vp, fp, fp, fp, fp, fp, fp]
        L.cmpc_set_initial_guess.argtypes = [vp, fp, C.c_int]
        L.cmpc_advance.argtypes = [vp]
        L.cmpc_get_solution.argtypes = [vp, fp, fp]
        L.cmpc_get_output.argtypes = [vp, fp, fp, fp, vp]
        L.cmpc_set_reference_from_planner.argtypes = [vp, fp, fp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
        L.cmpc_plant_step_device.argtypes = [vp, fp, fp, fp, fp, fp, C.c_double, C.c_int, C.c_double, C.c_double, vp]
        d, i = C.c_double, C.c_int
        L.cmpc_contacts_merge.argtypes = [i, i, d, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.cmpc_contacts_merge_device.argtypes = [vp, i, d, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.cmpc_contacts_sample.argtypes = [i, d, i, i, d, vp, vp, vp, vp, vp, vp, vp]
        L.cmpc_contacts_sample_device.argtypes = [vp, i, d, vp, vp, vp, vp, vp, vp, vp, vp]
        L.cmpc_set_contact_lists.argtypes = [vp, i, d, vp, vp, vp, vp, vp, vp]
        L.cmpc_contacts_adjust.argtypes = [i, i, i, d, vp, vp, vp, vp, vp]
        L.cmpc_contacts_adjust_device.argtypes = [vp, i, d, vp, vp, vp, vp, vp, vp]
        L.cmpc_write_state_device.argtypes = [vp, fp, fp, fp, vp]
        L.cmpc_shift_solution_device.argtypes = [vp, fp, fp, vp]
        L.cmpc_eval_nlp_grad_device.argtypes = [vp, fp, fp, fp, C.c_float, fp, fp, vp]
        if hasattr(L, "cmpc_get_parameters"):   # (absent from earlier rounds' builds of the library, which tools/ab_multi.sh may load as a baseline)
            L.cmpc_get_parameters.argtypes = [vp, fp]
            L.cmpc_get_parameters_device.argtypes = [vp, C.POINTER(vp)]
            L.cmpc_allgather_compact_device.argtypes = [vp, vp, C.c_int, fp, fp, vp]
            L.cmpc_sq_pass_barriers.argtypes = [C.c_int, C.c_int, C.c_int]
        if hasattr(L, "cmpc_rollout_tick_device"):
            L.cmpc_rollout_tick_device.argtypes = [vp, i, d, i, C.POINTER(CmpcTickIO), vp]
        if hasattr(L, "cmpc_write_reference_from_planner_device"):
            L.cmpc_write_reference_from_planner_device.argtypes = [vp, fp, fp, i, d, d, d, d, fp, vp]
        _lib = L
    return _lib
