"""Developer probe (GPU box): what a roll-out tick costs beside its solve -- tick wall clock (seven launches through the C ABI + the torch ops between
them + the host's read of the status word) against the solve kernel's own duration (cmpc_last_solve_ms), per batch size.  The gap is what a native
tick entry point / a captured hipGraph could remove; it is what a single robot (B = 1, the reference's real operating mode) pays per tick."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmpc_amd as cm
cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
ticks = int(os.environ.get("PROBE_TICKS", "80"))
for B in (1, 16, 256, 2048):
    rng = np.random.default_rng(5)
    com0 = np.array([0.0, 0.0, 0.7]) + rng.uniform(-0.01, 0.01, (B, 3))
    dcom0 = rng.uniform(-0.05, 0.05, (B, 3))
    h0 = rng.uniform(-0.02, 0.02, (B, 3))
    push = np.zeros((B, 3)); push[:, :2] = rng.uniform(-20.0, 20.0, (B, 2)) / cm.synthetic.ROBOT_MASS
    ro = cm.rollout.WalkingRollout(cfg, B)
    ro.run(8, com0, dcom0, h0, push=push, push_ticks=3, record="light")          # warm-up (module load, allocator)
    rec = ro.run(ticks, com0, dcom0, h0, push=push, push_ticks=3, record="light")
    tick, solve = np.array(rec["tick_ms"][1:]), np.array(rec["solve_ms"][1:])      # (tick 0 is the cold start)
    gap = tick - solve
    print(f"B={B:5d} ticks={ticks} converged {all(rec['converged'])} | tick ms p50 {np.median(tick):.3f} p99 {np.percentile(tick, 99):.3f} | solve kernel ms p50 "
          f"{np.median(solve):.3f} | tick - solve p50 {np.median(gap):.3f} ms ({100 * np.median(gap) / np.median(tick):.1f} % of the tick) | iterations mean "
          f"{np.mean(rec['iterations_mean']):.2f}", flush=True)
    rec = ro.run(ticks, com0, dcom0, h0, push=push, push_ticks=3, record="light", timing=False)
    tick = np.array(rec["tick_ms"][1:])
    print(f"        without the library's event pair around the solve: tick ms p50 {np.median(tick):.3f} p99 {np.percentile(tick, 99):.3f}", flush=True)
