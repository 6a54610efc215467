"""Receding-horizon walking roll-out of a whole batch, resident in HBM: every tick is what the reference's two blocks do
between them -- merge the planner's footsteps with the MPC-adjusted current contact (updateContactPhaseList,
CentroidalMPCBlock.cpp:594-607), sample the list into the MPC's parameters (setContactPhaseList :609), feed the measured
state (setState :407), warm-start from the shifted previous solution (is_warm_start_enabled), solve (advance :615), write
the optimised landing position back into the list (getOutput :626) and integrate the centroidal dynamics under the
first-knot forces until the next tick (WholeBodyQPBlock.cpp:1083-1150) -- as seven launches on one stream through the
C ABI (include/cmpc.h); torch only owns the buffers.  SURVEY 8f-1 .. 8f-4 chained."""
from __future__ import annotations

import numpy as np

from .config import GRAVITY

from .contacts import PlannedContact, pack_lists
from .layout import Layout
from .solver import BatchSolver

FOOT_Y = 0.08


def walking_plan(cfg, steps=6, step_length=0.1, swing=0.48, double_support=0.12, first_lift=0.36):
    """A periodic straight walk (absolute times from zero): the left foot lifts first at `first_lift`."""
    names = [c.contact_name for c in cfg.contacts]
    feet = {0: [PlannedContact(0.0, 0.0, (0.0, FOOT_Y, 0.0))], 1: [PlannedContact(0.0, 0.0, (0.0, -FOOT_Y, 0.0))]}
    t, side = first_lift, 0
    for s in range(steps):
        other = feet[1 - side][-1]
        feet[side][-1].deactivation_time = t
        x = other.position[0] + step_length * (0.5 if s == 0 else 1.0)
        feet[side].append(PlannedContact(t + swing, 0.0, (x, FOOT_Y if side == 0 else -FOOT_Y, 0.0)))
        t += swing + double_support
        side = 1 - side
    for lst in feet.values():
        lst[-1].deactivation_time = 1e9
    return {names[0]: feet[0], names[1]: feet[1]}


class WalkingRollout:
    """warm_budget / retry: what a warm-started problem that does not converge costs its tick (include/cmpc.h, cmpc_set_warm_policy).
    warm_budget = iterations of the warm-started pass (0: the full budget; 14 = the library's default); retry = "kernel": such a problem starts again from the cold
    start inside the same launch (one workgroup holds its CU for two budgets); "launch": it comes back unconverged and the tick's few
    stragglers are solved again from the cold start in a small launch of their own (a CU each); None: they stay unconverged -- which is all
    the reference can do: its advance() returns false and the tick is aborted (CentroidalMPCBlock.cpp:615-619)."""

    def __init__(self, cfg, batch, plan=None, device=0, substeps=6, com_speed=None, warm_budget=14, retry="kernel", retry_batch=256, native_tick=True, **solver_opts):
        import torch
        self.torch = torch
        self.cfg, self.B = cfg, batch
        self.L = Layout(cfg.N)
        self.dev = torch.device("cuda", device)
        self.solver = BatchSolver(cfg, batch, device=device, **solver_opts)
        assert retry in ("kernel", "launch", None)
        self.retry, self.retry_batch = retry, min(retry_batch, batch)
        # native_tick: a warm-started tick is ONE call of the C ABI (cmpc_rollout_tick_device: the same seven entry points chained inside the library, bit-identical
        # results) instead of seven; ticks that need the host between the steps (cold starts, retry="launch", the dump hook) take the step-by-step path
        self.native_tick = native_tick and retry != "launch"
        self.solver.set_warm_policy(warm_budget, restart_in_kernel=(retry == "kernel"))
        self.solver2 = BatchSolver(cfg, self.retry_batch, device=device, **solver_opts) if retry == "launch" else None
        plan = plan or walking_plan(cfg)
        t, pose, n = pack_lists(cfg, [plan])
        M = t.shape[2] + 1     # the merged list holds at most the current contact + the planner's future contacts
        tt = np.zeros((1, 2, M, 2)); pp = np.zeros((1, 2, M, 7), np.float32); pp[..., 3] = 1.0
        tt[:, :, :M - 1] = t; pp[:, :, :M - 1] = pose
        rep = lambda a: torch.from_numpy(np.ascontiguousarray(np.broadcast_to(a, (batch,) + a.shape[1:]))).to(self.dev)
        self.plan = (rep(tt), rep(pp), rep(n))
        self.M = M
        self.substeps = substeps
        # mean walking speed of the plan, for the CoM reference
        if com_speed is None:
            last = max(c.position[0] for lst in plan.values() for c in lst)
            t_last = max(c.activation_time for lst in plan.values() for c in lst)
            com_speed = last / t_last if t_last > 0 else 0.0
        self.com_speed = com_speed

    def run(self, *args, **kwargs):
        """The roll-out (see _run for the arguments), with the solver's launch stream as torch's current stream: every device call of a tick -- the library's
        kernels and the torch ops between them -- is then queued on ONE non-default stream, in order, without the event dependencies a call from the default
        stream needs (BatchSolver._stream_pair: two per call, ~24 us of idle GPU each time)."""
        torch = self.torch
        ls = self.solver.launch_stream
        cur = torch.cuda.current_stream(self.dev)
        ls.wait_stream(cur)
        try:
            with torch.cuda.stream(ls):
                return self._run(*args, **kwargs)
        finally:
            cur.wait_stream(ls)

    def _tick_by_steps(self, i, now, mpc_prev, warm, dump, dP, dX0, dX, dInfo, state, wrench, dpush, push_ticks, planner):
        """one tick as seven calls of the C ABI with the host between them (cold starts, retry="launch", the dump hook; native_tick=False)"""
        torch, L, cfg, B, N = self.torch, self.L, self.cfg, self.B, self.cfg.N
        dt, dev, s = cfg.sampling_time, self.dev, self.solver
        if mpc_prev is None:
            lists = tuple(a.clone() for a in self.plan)
            ok = torch.ones((B,), dtype=torch.int32, device=dev)
        else:
            # (whether every merge succeeded is read by the host at the END of the tick, with the status words: a read here would drain the stream in the
            #  middle of the tick and leave the GPU idle while the host queues the six launches in front of the solve -- 0.17 ms of a 0.83 ms tick at
            #  B <= 256, tools/gpu_rollout_tick_overhead.py.  Until then a failed problem is harmless: its merged list is empty, the sampling kernel leaves
            #  its blocks of dP as they were and writes land = -2, the adjustment kernel skips it -- include/cmpc.h)
            lists, ok = s.contacts_merge_device(now, self.plan, mpc_prev)
        land = s.contacts_sample_device(now, lists, dP)
        s.write_reference_from_planner_device(planner[0], planner[1], planner[2], planner[3], planner[4], planner[5], dP)
        if dpush is not None and i <= push_ticks:      # (the wrench rows of dP change while the push lasts and once more when it ends; zero from the start otherwise)
            wrench.zero_()
            if i < push_ticks:
                wrench[:, :max(push_ticks - i, 1), :3] = dpush[:, None, :]
            s.write_state_device(state, dP, wrench)
        else:
            s.write_state_device(state, dP, None)
        shifted = not (mpc_prev is None or not warm)
        if not shifted:
            # cold start: CoM at com0, feet at nominal, f_z = g/8 per corner
            dX0.zero_()
            dX0[:, L.com:L.com + 3 * (N + 1)] = state[:, 0:3].repeat(1, N + 1)
            for c in range(2):
                dX0[:, L.pos[c]:L.pos[c] + 3 * (N + 1)] = dP[:, L.p_nom[c]:L.p_nom[c] + 3 * (N + 1)]
                for j in range(4):
                    dX0[:, L.f[c][j] + 2:L.f[c][j] + 3 * N:3] = GRAVITY / 8.0   # (the library's cold start: cmpc_config.gravity / 8, which this package always sets to GRAVITY)
        else:
            s.shift_solution_device(dX, dX0)
        if dump is not None and i == dump[0]:   # developer hook: (tick, path) -> the tick's P and X0
            np.savez(dump[1], P=dP.cpu().numpy(), X0=dX0.cpu().numpy())
        s.solve_device(dP, dX0, dX, dInfo, warm=shifted)
        nretry = 0
        if self.retry == "launch" and shifted:
            bad = (dInfo[:, 5] != 0).nonzero().flatten()        # (one scalar comes to the host: the count)
            nretry = int(bad.numel())
            for lo in range(0, nretry, self.retry_batch):
                chunk = bad[lo:lo + self.retry_batch]
                idx = torch.cat([chunk, chunk[:1].expand(self.retry_batch - chunk.numel())]) if chunk.numel() < self.retry_batch else chunk
                P2 = dP.index_select(0, idx)
                X02 = torch.zeros((self.retry_batch, L.nx), dtype=torch.float32, device=dev)       # the cold start of SURVEY 8d
                X02[:, L.com:L.com + 3 * (N + 1)] = P2[:, L.p_com0:L.p_com0 + 3].repeat(1, N + 1)
                for c in range(2):
                    X02[:, L.pos[c]:L.pos[c] + 3 * (N + 1)] = P2[:, L.p_nom[c]:L.p_nom[c] + 3 * (N + 1)]
                    for j in range(4):
                        X02[:, L.f[c][j] + 2:L.f[c][j] + 3 * N:3] = GRAVITY / 8.0
                X2, I2 = self.solver2.solve_device(P2, X02)
                I2[:, 0] += dInfo.index_select(0, idx)[:, 0]     # iterations of both attempts
                I2[:, 3] += 10000.0                              # safeguard word: solved again from the cold start
                dX.index_copy_(0, chunk, X2[:chunk.numel()])
                dInfo.index_copy_(0, chunk, I2[:chunk.numel()])
        s.contacts_adjust_device(now, dX, land, lists)
        state, zmp = s.plant_step_device(dX, dP, state, step=dt / self.substeps, substeps=self.substeps)
        return ok, lists, land, nretry, state, zmp

    def _run(self, ticks, com0, dcom0, h0, push=None, push_ticks=0, warm=True, dump=None, replan=None, slow=None, record="full", timing=True):
        """com0/dcom0/h0 [B,3] numpy; push [B,3] (mass-normalised force held for the first `push_ticks` ticks);
        replan {tick: (t, pose, n)}: the planner's lists from that tick on (the reference's generator re-plans while walking);
        slow (threshold, list): developer hook -- (tick, problem, P row, X0 row, info row) of every solve with more iterations;
        record "full": per-tick CoM, ZMP, landing offsets (host loops over the batch); "light": iteration statistics and the tick's
        wall-clock latency only (what bench.py times).  timing=False: without the library's event pair around every solve (cmpc_set_timing: an event record
        is a barrier packet on the stream, ~10 us of a tick each); rec["solve_ms"] is then NaN.
        Returns a dict of per-tick numpy records."""
        torch, L, cfg, B, N = self.torch, self.L, self.cfg, self.B, self.cfg.N
        dt = cfg.sampling_time
        dev = self.dev
        s = self.solver
        dP = torch.zeros((B, L.np), dtype=torch.float32, device=dev)
        dX0 = torch.zeros((B, L.nx), dtype=torch.float32, device=dev)
        dX = torch.zeros_like(dX0)
        dInfo = torch.zeros((B, 8), dtype=torch.float32, device=dev)
        state = torch.from_numpy(np.concatenate([com0, dcom0, h0], 1).astype(np.float32)).to(dev)
        wrench = torch.zeros((B, N, 6), dtype=torch.float32, device=dev)
        dpush = torch.from_numpy(np.asarray(push, np.float32)).to(dev) if push is not None else None
        # references (CentroidalMPCBlock.cpp:525-577 resamples the planner's trajectories at the MPC knots): the planner here is a straight line at the plan's mean
        # speed and zero angular momentum, a knot every dt from time zero to the end of the last tick's horizon -- resampled on the device every tick
        # (cmpc_write_reference_from_planner_device; inside cmpc_rollout_tick_device for the warm ticks), CoM height forced to 0.7 as the reference does (:534)
        n_plan = ticks + N + 2
        plan_com = torch.zeros((B, n_plan, 3), dtype=torch.float32, device=dev)
        plan_com[:, :, 0] = (self.com_speed * dt * torch.arange(n_plan, dtype=torch.float64, device=dev)).to(torch.float32)[None, :]
        plan_h = torch.zeros_like(plan_com)
        planner = lambda now: (plan_com, plan_h, dt, now, 1.0, 0.7)
        import time
        rec = dict(iterations_mean=[], iterations_max=[], converged=[], merge_ok=[], com=[], land=[], landing_offset=[], solve_ms=[], zmp=[],
                   tick_ms=[], retried=[], unconverged=[])
        mpc_prev, tick_bufs = None, None
        s.set_timing(timing)
        box_up = np.array([c.bounding_box_upper_limit for c in cfg.contacts])
        box_lo = np.array([c.bounding_box_lower_limit for c in cfg.contacts])
        for i in range(ticks):
            now = i * dt
            torch.cuda.synchronize()
            t_tick = time.perf_counter()
            if replan and i in replan:
                self.plan = replan[i]
            if self.native_tick and warm and mpc_prev is not None and not (dump is not None and i == dump[0]):
                if tick_bufs is None:
                    tick_bufs = ([tuple(torch.zeros_like(a) for a in self.plan) for _ in range(2)], torch.empty((B, 2), dtype=torch.int32, device=dev),
                                 torch.empty((B, 2), dtype=torch.float32, device=dev))
                lists = tick_bufs[0][i & 1] if mpc_prev[0] is not tick_bufs[0][i & 1][0] else tick_bufs[0][1 - (i & 1)]
                ok = torch.empty((B,), dtype=torch.int32, device=dev)
                land, zmp = tick_bufs[1], tick_bufs[2]
                wr = None
                if dpush is not None and i <= push_ticks:
                    wrench.zero_()
                    if i < push_ticks:
                        wrench[:, :max(push_ticks - i, 1), :3] = dpush[:, None, :]
                    wr = wrench
                s.rollout_tick_device(now, self.plan, mpc_prev, lists, ok, land, state, wr, dP, dX0, dX, dInfo, state, zmp, True,
                                      step=dt / self.substeps, substeps=self.substeps, planner=planner(now))
                mpc_prev = lists
                nretry = 0
            else:
                ok, lists, land, nretry, state, zmp = self._tick_by_steps(i, now, mpc_prev, warm, dump, dP, dX0, dX, dInfo, state, wrench, dpush, push_ticks, planner(now))
                mpc_prev = lists
            torch.cuda.synchronize()
            tick_ms = (time.perf_counter() - t_tick) * 1e3
            if not bool(ok.cpu().numpy().all()):
                # the reference aborts the tick when updateContactPhaseList returns false (CentroidalMPCBlock.cpp:603-607): so does the roll-out -- what
                # the tick computed is discarded (every per-tick list gets its entry, so that the records stay aligned; bench.py fails the roll-out
                # when it sees 'aborted_tick')
                rec["merge_ok"].append(False)
                rec["aborted_tick"] = i
                rec["tick_ms"].append(float("nan")); rec["retried"].append(0); rec["unconverged"].append(B)
                rec["iterations_mean"].append(float("nan")); rec["iterations_max"].append(0); rec["converged"].append(False)
                rec["solve_ms"].append(float("nan"))
                break
            rec["tick_ms"].append(tick_ms)
            rec["retried"].append(nretry)
            info = dInfo.cpu().numpy()
            rec["unconverged"].append(int((info[:, 5] != 0).sum()))
            if slow is not None:
                for b in np.where(info[:, 0] > slow[0])[0]:
                    slow[1].append((i, int(b), dP[b].cpu().numpy(), dX0[b].cpu().numpy(), info[b].copy()))
            rec["iterations_mean"].append(float(info[:, 0].mean()))
            rec["iterations_max"].append(int(info[:, 0].max()))
            rec["converged"].append(bool((info[:, 5] == 0).all()))
            rec.setdefault("failed_info", []).append(info[info[:, 5] != 0])
            rec["merge_ok"].append(True)
            rec["solve_ms"].append(s.last_solve_ms() if timing else float("nan"))
            if record != "full":
                continue
            rec["com"].append(state[:, 0:3].cpu().numpy())
            rec["zmp"].append(zmp.cpu().numpy())
            ln = land.cpu().numpy()
            rec["land"].append(ln)
            # landing position against the nominal one of the same knot, in the foot frame (the bounding box the NLP imposes)
            Xh, Ph = dX.cpu().numpy(), dP.cpu().numpy()
            off = np.zeros((B, 2, 3))
            for c in range(2):
                for b in range(B):
                    k = ln[b, c]
                    if 0 < k <= N:
                        R = Ph[b, L.p_R[c] + 9 * (k - 1):L.p_R[c] + 9 * k].reshape(3, 3).T   # vec(R) column-major
                        d = Xh[b, L.pos[c] + 3 * k:L.pos[c] + 3 * k + 3] - Ph[b, L.p_nom[c] + 3 * k:L.p_nom[c] + 3 * k + 3]
                        off[b, c] = R.T @ d
            rec["landing_offset"].append(off)
        s.set_timing(True)
        rec["box_upper"], rec["box_lower"] = box_up, box_lo
        return rec
