"""Developer probe (GPU box): tick latency of the warm-started walking roll-out under the warm-start policies (cmpc_set_warm_policy /
WalkingRollout(warm_budget, retry)): wall-clock per tick (seven launches + the host's look at the status), p50 / p99 / max."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmpc_amd as cm
cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
B, ticks = int(os.environ.get("PROBE_B", "2048")), int(os.environ.get("PROBE_TICKS", "120"))
rng = np.random.default_rng(9)
com0 = np.array([0.0, 0.0, 0.7]) + rng.uniform(-0.01, 0.01, (B, 3))
dcom0 = rng.uniform(-0.05, 0.05, (B, 3)); h0 = rng.uniform(-0.02, 0.02, (B, 3))
push = np.zeros((B, 3)); push[:, :2] = rng.uniform(-30.0, 30.0, (B, 2)) / cm.synthetic.ROBOT_MASS
plan = cm.rollout.walking_plan(cfg, steps=11)
pols = [(0, "kernel"), (14, "kernel"), (14, "launch"), (12, "launch"), (16, "launch"), (14, None)]
for budget, retry in pols:
    ro = cm.rollout.WalkingRollout(cfg, B, plan=plan, warm_budget=budget, retry=retry)
    ro.run(3, com0, dcom0, h0, push=push, push_ticks=3, record="light")        # warm-up (allocations, first launches)
    rec = ro.run(ticks, com0, dcom0, h0, push=push, push_ticks=3, record="light")
    ms = np.array(rec["tick_ms"][1:]); it = np.array(rec["iterations_max"]); itm = np.array(rec["iterations_mean"])
    print(f"B={B} ticks={ticks} warm_budget={budget} retry={retry}: tick ms p50 {np.median(ms):.3f} p99 {np.percentile(ms, 99):.3f} max {ms.max():.3f} "
          f"(max/p50 {ms.max() / np.median(ms):.2f}) | solves/s {B * len(ms) / ms.sum() * 1e3:.0f} | iterations mean {itm.mean():.2f} max {it.max()} | "
          f"retried {sum(rec['retried'])} unconverged {sum(rec['unconverged'])} merge ok {all(rec['merge_ok'])}", flush=True)
    ro.solver.close()
    if ro.solver2: ro.solver2.close()
