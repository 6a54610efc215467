"""Host-side mirror of the reference's CentroidalMPC surface for a batch of problems.

Reference interface (BipedalLocomotion::ReducedModelControllers::CentroidalMPC, used at
src/centroidal-mpc-walking/src/CentroidalMPCBlock.cpp:144,407,579,609,615,622):
    initialize / setState / setReferenceTrajectory / setContactPhaseList / advance / getOutput
Every mutator returns bool like the reference's (False => the caller logs and aborts the tick,
CentroidalMPCBlock.cpp:609-619); `last_error` holds the text.  All compute happens in
libcmpc_hip.so (hand-written gfx950 kernels) through the C ABI of include/cmpc.h.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _capi
from .config import GRAVITY, CentroidalMPCConfig, from_ini
from .contacts import PlannedContact, pack_lists, sample_schedule, sample_schedule_batch
from .layout import Layout


def _c_config(cfg: CentroidalMPCConfig, tolerance=None, mu_min=None, max_iterations=None,
              exact_hessian=True, final_extrapolation=True, step_tolerance=None, mu_init=None,
              tail_stages=3, tail_iterations=2, tail_trigger=2e-5, factors=None) -> _capi.CmpcConfig:
    c = _capi.CmpcConfig()
    c.horizon = cfg.N
    c.sampling_time = cfg.sampling_time
    c.friction_coefficient = cfg.static_friction_coefficient
    c.gravity = GRAVITY
    c.com_weight[:] = cfg.com_weight
    c.angular_momentum_weight = cfg.angular_momentum_weight
    c.contact_position_weight = cfg.contact_position_weight
    c.force_rate_of_change_weight[:] = cfg.force_rate_of_change_weight
    c.contact_force_symmetry_weight = cfg.contact_force_symmetry_weight
    c.corners[:] = np.asarray([cc.corners for cc in cfg.contacts], np.float64).reshape(-1)
    c.max_iterations = max_iterations or cfg.ipopt_max_iteration
    # the reference's ipopt_tolerance (1e-4 / 1e-2) is looser than the parity target; the GPU
    # solver always converges at least to 1e-6 so that its answer is reproducible to 1e-4
    # (0: the library's default -- 1e-6 up to N = 20, 3e-7 beyond; step tolerance and barrier floor follow it)
    c.tolerance = tolerance if tolerance is not None else (cfg.ipopt_tolerance if cfg.ipopt_tolerance < 5e-7 else 0.0)
    c.step_tolerance = step_tolerance if step_tolerance is not None else 0.0
    c.mu_init = mu_init if mu_init is not None else 0.0   # <= 0: per problem, from its initial infeasibility
    c.mu_min = mu_min if mu_min is not None else 0.0
    c.exact_hessian = int(exact_hessian)
    c.final_extrapolation = int(final_extrapolation)
    # tail polish (include/cmpc.h): the last stages re-solved when their extrapolation step is large
    c.tail_stages, c.tail_iterations, c.tail_trigger = int(tail_stages), int(tail_iterations), float(tail_trigger)
    c.factor_storage = _capi.FACTORS[factors]   # None / "auto": by batch size and horizon; "lds" / "hbm": the resident / the HBM-factor variant
    return c


class BatchSolver:
    """Thin RAII wrapper of a cmpc_handle: device-resident batched solve."""

    def __init__(self, cfg: CentroidalMPCConfig, batch: int, device: int = 0, **opts):
        self.cfg = cfg
        self.layout = Layout(cfg.N)
        self.batch = batch
        self._device_index = device
        self._lib = _capi.lib()
        self._ccfg = _c_config(cfg, **opts)
        h = C.c_void_p()
        rc = self._lib.cmpc_create(C.byref(self._ccfg), batch, device, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"cmpc_create failed ({rc}): {self._lib.cmpc_last_error(None).decode()}")
        self._h = h
        self._stream = None  # torch.cuda.Stream the kernels are launched on (created lazily)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cmpc_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def last_error(self) -> str:
        return self._lib.cmpc_last_error(self._h).decode()

    def solve_device(self, dP, dX0, dX=None, dInfo=None, stream=None, warm=False):
        """torch CUDA tensors float32: P[B,np], X0[B,nx] -> X[B,nx], info[B,8].  Launches on
        torch's current stream unless `stream` (a raw hipStream_t) is given; asynchronous.
        warm: dX0 is the previous solution shifted by one knot (shift_solution_device) -> cmpc_solve_device_warm."""
        import torch
        L = self.layout
        assert dP.is_cuda and dP.dtype == torch.float32 and dP.is_contiguous() and tuple(dP.shape) == (self.batch, L.np)
        assert dX0.is_cuda and dX0.dtype == torch.float32 and dX0.is_contiguous() and tuple(dX0.shape) == (self.batch, L.nx)
        if dX is None:
            dX = torch.empty_like(dX0)
        if dInfo is None:
            dInfo = torch.empty((self.batch, _capi.INFO), dtype=torch.float32, device=dP.device)
        cur = None
        if stream is None:
            stream, cur = self._stream_pair(dP.device)
        fn = self._lib.cmpc_solve_device_warm if warm else self._lib.cmpc_solve_device
        rc = fn(self._h, dP.data_ptr(), dX0.data_ptr(), dX.data_ptr(), dInfo.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError(f"cmpc_solve_device failed ({rc}): {self.last_error}")
        if cur is not None:
            cur.wait_stream(self._stream)
        return dX, dInfo

    def _stream_pair(self, dev):
        """(raw hipStream_t to launch on, torch stream to join back into or None).  Torch's current stream itself when it is not the default stream -- no
        cross-stream dependency at all; run a loop of device calls under `with torch.cuda.stream(solver.launch_stream):` to get that.  The default stream's
        handle is 0, which the C ABI reads as "the handle's own stream": there the call goes to a side stream ordered after the default stream and joined
        back into it -- two event dependencies per call, measured at 24 us of idle GPU per solve between back-to-back launches (tools/gpu_enqueue_probe.py)."""
        import torch
        cur = torch.cuda.current_stream(dev)
        if cur.cuda_stream != 0:
            return cur.cuda_stream, None
        if self._stream is None:
            self._stream = torch.cuda.Stream(dev)
        self._stream.wait_stream(cur)
        return self._stream.cuda_stream, cur

    @property
    def launch_stream(self):
        """torch.cuda.Stream the solve kernel runs on (record timing events here)."""
        import torch
        if self._stream is None:
            self._stream = torch.cuda.Stream(torch.device("cuda", self._device_index))
        return self._stream

    def solve_host(self, P: np.ndarray, X0: np.ndarray):
        """numpy float32 in/out through the PCIe-inclusive entry point.  Returns (X, info, rc)."""
        L = self.layout
        P = np.ascontiguousarray(P, np.float32)
        X0 = np.ascontiguousarray(X0, np.float32)
        assert P.shape == (self.batch, L.np) and X0.shape == (self.batch, L.nx)
        X = np.empty_like(X0)
        info = np.empty((self.batch, _capi.INFO), np.float32)
        rc = self._lib.cmpc_solve(self._h, P.ctypes.data, X0.ctypes.data, X.ctypes.data, info.ctypes.data)
        if rc not in (0, -3):
            raise RuntimeError(f"cmpc_solve failed ({rc}): {self.last_error}")
        return X, info, rc

    def set_warm_policy(self, warm_budget: int = 0, restart_in_kernel: bool = True):
        """cmpc_set_warm_policy: iteration budget of a warm-started pass (0: max_iterations) and whether a warm start that does not
        converge is started again from the cold start inside the launch (True) or returned with status 1 (False)."""
        rc = self._lib.cmpc_set_warm_policy(self._h, int(warm_budget), 1 if restart_in_kernel else 0)
        if rc != 0:
            raise RuntimeError(f"cmpc_set_warm_policy failed ({rc}): {self.last_error}")

    def last_solve_ms(self) -> float:
        return float(self._lib.cmpc_last_solve_ms(self._h))

    def set_timing(self, enabled: bool = True):
        """cmpc_set_timing: record (default) or not the event pair around every solve launch that last_solve_ms() reads."""
        if hasattr(self._lib, "cmpc_set_timing"):
            self._lib.cmpc_set_timing(self._h, 1 if enabled else 0)

    def compact_output_device(self, dX, dInfo, out=None):
        """[B, 3(N+1) + 38] compact record of every problem (what distributed.compact_output builds with torch ops), by
        one kernel on the solver's stream; torch CUDA tensors."""
        import torch
        W = 3 * (self.cfg.N + 1) + 38
        if out is None:
            out = torch.empty((self.batch, W), dtype=torch.float32, device=dX.device)
        assert out.is_contiguous() and tuple(out.shape) == (self.batch, W)
        st, cur = self._stream_pair(dX.device)
        rc = self._lib.cmpc_compact_output_device(self._h, dX.data_ptr(), dInfo.data_ptr(), out.data_ptr(), st)
        if rc != 0:
            raise RuntimeError(f"cmpc_compact_output_device failed ({rc}): {self.last_error}")
        if cur is not None:
            cur.wait_stream(self._stream)
        return out

    def plant_step_device(self, dX, dP, dState, dStateOut=None, dZmp=None, step=0.01, substeps=6,
                          zmp_half_x=0.08, zmp_half_y=0.03):
        """Closed-loop plant between two MPC ticks (WholeBodyQPBlock.cpp:805-873, 1083-1084, 1150): RK4 of the
        centroidal dynamics under the first-knot forces; torch CUDA tensors; returns (state[B,9], zmp[B,2])."""
        import torch
        if dStateOut is None:
            dStateOut = torch.empty_like(dState)
        if dZmp is None:
            dZmp = torch.empty((self.batch, 2), dtype=torch.float32, device=dState.device)
        st, cur = self._stream_pair(dState.device)
        rc = self._lib.cmpc_plant_step_device(self._h, dX.data_ptr(), dP.data_ptr(), dState.data_ptr(), dStateOut.data_ptr(),
                                              dZmp.data_ptr(), float(step), int(substeps), float(zmp_half_x), float(zmp_half_y), st)
        if rc != 0:
            raise RuntimeError(f"cmpc_plant_step_device failed ({rc}): {self.last_error}")
        if cur is not None:
            cur.wait_stream(self._stream)
        return dStateOut, dZmp


    # ---- SURVEY 8f-1 / 8f-2 on the device (torch CUDA tensors; everything stays in HBM) ----
    def _launch(self, dev, fn):
        """runs fn(raw_stream) on torch's current stream (the default stream: on the solver's side stream, ordered after it and joined back; _stream_pair)"""
        st, cur = self._stream_pair(dev)
        rc = fn(st)
        if rc != 0:
            raise RuntimeError(f"libcmpc_hip call failed ({rc}): {self.last_error}")
        if cur is not None:
            cur.wait_stream(self._stream)

    def contacts_merge_device(self, now, plan, mpc, out=None):
        """updateContactPhaseList (CentroidalMPCBlock.cpp:32-110) for the batch: plan / mpc / out = (t[B,2,M,2] float64,
        pose[B,2,M,7] float32, n[B,2] int32) CUDA tensors.  Returns (out, ok[B] int32)."""
        import torch
        pt, pp, pn = plan
        mt, mp, mn = mpc
        M = pt.shape[2]
        if out is None:
            out = (torch.zeros_like(pt), torch.zeros_like(pp), torch.zeros_like(pn))
        ok = torch.empty((self.batch,), dtype=torch.int32, device=pt.device)
        self._launch(pt.device, lambda st: self._lib.cmpc_contacts_merge_device(
            self._h, M, float(now), pt.data_ptr(), pp.data_ptr(), pn.data_ptr(), mt.data_ptr(), mp.data_ptr(), mn.data_ptr(),
            out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), ok.data_ptr(), st))
        return out, ok

    def contacts_sample_device(self, now, lists, dP, land=None):
        """setContactPhaseList for the batch: samples `lists` at now + k dt into the contact blocks of dP[B,np]; returns
        land[B,2] (landing knots)."""
        import torch
        t, pose, n = lists
        if land is None:
            land = torch.empty((self.batch, 2), dtype=torch.int32, device=dP.device)
        up = np.ascontiguousarray([c.bounding_box_upper_limit for c in self.cfg.contacts], np.float32)
        lo = np.ascontiguousarray([c.bounding_box_lower_limit for c in self.cfg.contacts], np.float32)
        self._launch(dP.device, lambda st: self._lib.cmpc_contacts_sample_device(
            self._h, t.shape[2], float(now), t.data_ptr(), pose.data_ptr(), n.data_ptr(), up.ctypes.data, lo.ctypes.data, dP.data_ptr(),
            land.data_ptr(), st))
        return land

    def contacts_adjust_device(self, now, dX, land, lists):
        """getOutput().contactPhaseList: the next contact of every foot landing inside the horizon takes x.pos[land] (in place)."""
        t, pose, n = lists
        self._launch(dX.device, lambda st: self._lib.cmpc_contacts_adjust_device(
            self._h, t.shape[2], float(now), dX.data_ptr(), land.data_ptr(), t.data_ptr(), pose.data_ptr(), n.data_ptr(), st))

    def write_state_device(self, dState, dP, dWrench=None):
        """setState for the batch: dState[B,9] (+ dWrench[B,N,6]) into the rows of dP."""
        self._launch(dP.device, lambda st: self._lib.cmpc_write_state_device(
            self._h, dState.data_ptr(), dWrench.data_ptr() if dWrench is not None else None, dP.data_ptr(), st))

    def write_reference_from_planner_device(self, dComIn, dHIn, in_dt, t_offset, robot_mass, com_height, dP):
        """setReferenceTrajectory from the planner's trajectories (8f-3) on the device: dComIn / dHIn [B, n_in, 3] float32 CUDA tensors -> comRef / hRef rows of dP."""
        assert dComIn.is_contiguous() and dHIn.is_contiguous() and dComIn.shape == dHIn.shape and dComIn.shape[0] == self.batch
        self._launch(dP.device, lambda st: self._lib.cmpc_write_reference_from_planner_device(
            self._h, dComIn.data_ptr(), dHIn.data_ptr(), int(dComIn.shape[1]), float(in_dt), float(t_offset), float(robot_mass),
            float("nan") if com_height is None else float(com_height), dP.data_ptr(), st))

    def rollout_tick_device(self, now, plan, prev, lists, ok, land, dState, dWrench, dP, dX0, dX, dInfo, dStateOut, dZmp, warm,
                            step=0.01, substeps=6, zmp_half_x=0.08, zmp_half_y=0.03, planner=None):
        """cmpc_rollout_tick_device: merge -> sample -> setState -> shift -> solve -> step adjustment -> plant as ONE call (include/cmpc.h); plan / prev /
        lists = (t, pose, n) CUDA tensors, prev None on the first tick (lists is then taken as filled by the caller); dWrench may be None."""
        from ._capi import CmpcTickIO
        if getattr(self, "_box", None) is None:
            self._box = (np.ascontiguousarray([c.bounding_box_upper_limit for c in self.cfg.contacts], np.float32),
                         np.ascontiguousarray([c.bounding_box_lower_limit for c in self.cfg.contacts], np.float32))
        ptr = lambda a: a.data_ptr() if a is not None else None
        io = CmpcTickIO(ptr(plan[0]), ptr(plan[1]), ptr(plan[2]),
                        *(tuple(ptr(a) for a in prev) if prev is not None else (None, None, None)),
                        ptr(lists[0]), ptr(lists[1]), ptr(lists[2]), ptr(ok), ptr(land), self._box[0].ctypes.data, self._box[1].ctypes.data,
                        ptr(dState), ptr(dWrench), ptr(dP), ptr(dX0), ptr(dX), ptr(dInfo), ptr(dStateOut), ptr(dZmp),
                        float(step), int(substeps), float(zmp_half_x), float(zmp_half_y))
        if planner is not None:   # (dComIn, dHIn, in_dt, t_offset, robot_mass, com_height): the tick writes the reference rows itself
            pc, ph, pdt, poff, mass, height = planner
            io.dPlanCom, io.dPlanH, io.plan_knots, io.plan_dt, io.plan_t_offset = pc.data_ptr(), ph.data_ptr(), int(pc.shape[1]), float(pdt), float(poff)
            io.robot_mass, io.com_height = float(mass), float("nan") if height is None else float(height)
        self._launch(dP.device, lambda st: self._lib.cmpc_rollout_tick_device(self._h, lists[0].shape[2], float(now), 1 if warm else 0, io, st))

    def shift_solution_device(self, dXprev, dX0):
        """is_warm_start_enabled: dX0 = dXprev shifted by one knot; solve from it with solve_device(..., warm=True)."""
        self._launch(dX0.device, lambda st: self._lib.cmpc_shift_solution_device(self._h, dXprev.data_ptr(), dX0.data_ptr(), st))


class CentroidalMPCOutput:
    """What getOutput() exposes downstream (WholeBodyQPBlock.cpp:824-829, 1319-1335): per contact
    the first-knot corner forces (world frame, mass-normalised) and pose, plus the step-adjusted
    next landing position (CentroidalMPCBlock.cpp:598, 626)."""

    def __init__(self, names, forces0, pos0, next_pos, next_knot):
        self.contact_names = names
        self.forces = forces0        # [B,2,4,3]
        self.positions = pos0        # [B,2,3]
        self.next_positions = next_pos  # [B,2,3]
        self.next_knots = next_knot  # [B,2]


class CentroidalMPC:
    """Batch counterpart of BipedalLocomotion::ReducedModelControllers::CentroidalMPC."""

    def __init__(self, batch: int = 1, device: int = 0):
        self._batch = batch
        self._device = device
        self._solver: Optional[BatchSolver] = None
        self._out: Optional[CentroidalMPCOutput] = None
        self.last_error = ""
        self._valid = False
        self._now = 0.0

    # -- initialize(handler): handler = CentroidalMPCConfig, ini text, or dict of options
    def initialize(self, handler, **solver_opts) -> bool:
        try:
            cfg = from_ini(handler) if isinstance(handler, str) else handler
            if not isinstance(cfg, CentroidalMPCConfig):
                raise TypeError("initialize() needs a CentroidalMPCConfig or the text of a centroidal_mpc.ini")
            self.cfg = cfg
            self._now = 0.0
            self._solver = BatchSolver(cfg, self._batch, self._device, **solver_opts)
            self._lib = self._solver._lib
            self._h = self._solver._h
            return True
        except Exception as e:  # mirrors the reference: log + return false
            self.last_error = str(e)
            return False

    def _ok(self, rc) -> bool:
        if rc != 0:
            self.last_error = self._solver.last_error
            return False
        return True

    def _need_init(self) -> bool:
        if self._solver is None:
            self.last_error = "initialize() has not been called"
            return False
        return True

    def set_state(self, com, dcom, angular_momentum, external_wrench=None) -> bool:
        """com, dcom, angular_momentum: [B,3]; external_wrench [B,6] (the measured wrench: enters the first knot only, like
        the C++ facade -- BLF's own rule is not visible from the reference tree, parity unpinned) or [B,N,6] (per knot) or
        None.  h and the wrench are mass-normalised (CentroidalMPCBlock.cpp:403-410)."""
        if not self._need_init():
            return False
        B, N = self._batch, self.cfg.N
        st = np.concatenate([np.reshape(com, (B, 3)), np.reshape(dcom, (B, 3)), np.reshape(angular_momentum, (B, 3))], 1)
        st = np.ascontiguousarray(st, np.float32)
        w = None
        if external_wrench is not None:
            w = np.asarray(external_wrench, np.float32)
            if w.ndim == 2:
                w0 = w
                w = np.zeros((B, N, 6), np.float32)
                w[:, 0, :] = w0
            w = np.ascontiguousarray(w.reshape(B, N, 6))
        return self._ok(self._lib.cmpc_set_state(self._h, st.ctypes.data, w.ctypes.data if w is not None else None))

    def set_reference_trajectory(self, com, angular_momentum) -> bool:
        """com, angular_momentum: [B,N+1,3] (the reference passes N+1 knots, CentroidalMPCBlock.cpp:230-235)."""
        if not self._need_init():
            return False
        B, N = self._batch, self.cfg.N
        c = np.ascontiguousarray(np.reshape(com, (B, N + 1, 3)), np.float32)
        h = np.ascontiguousarray(np.reshape(angular_momentum, (B, N + 1, 3)), np.float32)
        return self._ok(self._lib.cmpc_set_reference(self._h, c.ctypes.data, h.ctypes.data))

    def set_reference_from_planner(self, com_in, h_in, in_dt: float, t_offset: float, robot_mass: float,
                                   com_height: float = 0.7) -> bool:
        """Planner trajectories [B,M,3] every in_dt seconds -> MPC knots (CentroidalMPCBlock.cpp:525-577): angular
        momentum divided by the mass, CoM height forced to com_height (NaN keeps the planner's)."""
        if not self._need_init():
            return False
        B = self._batch
        c = np.ascontiguousarray(np.reshape(com_in, (B, -1, 3)), np.float32)
        h = np.ascontiguousarray(np.reshape(h_in, (B, -1, 3)), np.float32)
        return self._ok(self._lib.cmpc_set_reference_from_planner(self._h, c.ctypes.data, h.ctypes.data, c.shape[1], float(in_dt),
                                                                  float(t_offset), float(robot_mass), float(com_height)))

    def set_contact_phase_list(self, lists, t0: Optional[float] = None) -> bool:
        """lists: one dict {contact_name: [PlannedContact,...]} for every problem of the batch (or a single dict shared
        by all), with absolute times; or the packed arrays (t, pose, n) of contacts.pack_lists; or a dict of ready
        tensors with keys R, upper, lower, enabled, nominal, current.  Lists are sampled at the knots now + k dt, where
        `now` is the class's own clock: zero at initialize(), + dt per successful advance() -- the reference's caller
        never passes the time, it advances its own clock the same way (CentroidalMPCBlock.cpp:631).  t0 overrides it."""
        if not self._need_init():
            return False
        B = self._batch
        now = self._now if t0 is None else float(t0)
        try:
            if isinstance(lists, dict) and "R" in lists:
                t = lists
            else:
                if isinstance(lists, tuple):
                    packed = lists
                elif isinstance(lists, dict):
                    packed = pack_lists(self.cfg, [lists])
                else:
                    packed = pack_lists(self.cfg, list(lists))
                t, land = sample_schedule_batch(self.cfg, *packed, now)     # vectorised over the batch
                if packed[0].shape[0] == 1 and B > 1:
                    t = {k: np.broadcast_to(v, (B,) + v.shape[1:]) for k, v in t.items()}
                self._lists = packed
            self._sched = t
            a = {k: np.ascontiguousarray(t[k], np.float32) for k in ("R", "upper", "lower", "enabled", "nominal", "current")}
        except Exception as e:
            self.last_error = str(e)
            return False
        return self._ok(self._lib.cmpc_set_contacts(self._h, a["R"].ctypes.data, a["upper"].ctypes.data, a["lower"].ctypes.data,
                                                    a["enabled"].ctypes.data, a["nominal"].ctypes.data, a["current"].ctypes.data))

    def set_initial_guess(self, x0=None, shift_previous=False) -> bool:
        if not self._need_init():
            return False
        if x0 is not None:
            x0 = np.ascontiguousarray(x0, np.float32)
        return self._ok(self._lib.cmpc_set_initial_guess(self._h, x0.ctypes.data if x0 is not None else None, int(shift_previous)))

    def advance(self) -> bool:
        if not self._need_init():
            return False
        self._valid = False
        rc = self._lib.cmpc_advance(self._h)
        if rc != 0:
            self.last_error = self._solver.last_error
            return False
        B = self._batch
        f0 = np.empty((B, 2, 4, 3), np.float32)
        p0 = np.empty((B, 2, 3), np.float32)
        pn = np.empty((B, 2, 3), np.float32)
        kn = np.empty((B, 2), np.int32)
        if not self._ok(self._lib.cmpc_get_output(self._h, f0.ctypes.data, p0.ctypes.data, pn.ctypes.data, kn.ctypes.data)):
            return False
        self._out = CentroidalMPCOutput([c.contact_name for c in self.cfg.contacts], f0, p0, pn, kn)
        self._now += self.cfg.sampling_time
        self._valid = True
        return True

    def get_output(self) -> CentroidalMPCOutput:
        return self._out

    def is_output_valid(self) -> bool:
        return self._valid

    def get_solution(self):
        L = Layout(self.cfg.N)
        X = np.empty((self._batch, L.nx), np.float32)
        info = np.empty((self._batch, _capi.INFO), np.float32)
        if not self._ok(self._lib.cmpc_get_solution(self._h, X.ctypes.data, info.ctypes.data)):
            return None, None
        return X, info
