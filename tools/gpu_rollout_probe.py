"""Developer probe (GPU box): walking roll-out, per-tick iteration counts, warm (primal shift) against cold starts."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmpc_amd as cm
cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
B, ticks = int(os.environ.get("PROBE_B", "256")), int(os.environ.get("PROBE_TICKS", "24"))
rng = np.random.default_rng(5)
com0 = np.array([0.0, 0.0, 0.7]) + rng.uniform(-0.01, 0.01, (B, 3))
dcom0 = rng.uniform(-0.05, 0.05, (B, 3))
h0 = rng.uniform(-0.02, 0.02, (B, 3))
push = np.zeros((B, 3)); push[:, :2] = rng.uniform(-20.0, 20.0, (B, 2)) / cm.synthetic.ROBOT_MASS
for warm in (True, False):
    ro = cm.rollout.WalkingRollout(cfg, B)
    rec = ro.run(ticks, com0, dcom0, h0, push=push, push_ticks=3, warm=warm)
    print("warm" if warm else "cold", "converged", rec["converged"], "merge", all(rec["merge_ok"]))
    for i, fi in enumerate(rec["failed_info"]):
        if len(fi):
            print("  tick", i, "failed", len(fi), "info rows (it, kkt, mu, gn, ep, status, cycles, step):\n", np.array2string(fi[:4], precision=3, suppress_small=False))
    print("  iterations mean", np.round(rec["iterations_mean"], 2).tolist())
    print("  iterations max ", rec["iterations_max"])
    print("  solve ms       ", np.round(rec["solve_ms"], 3).tolist())
    com = np.stack(rec["com"])
    print("  com x range at end", com[-1, :, 0].min(), com[-1, :, 0].max(), "max |y|", np.abs(com[:, :, 1]).max(), "max |z-0.7|", np.abs(com[:, :, 2] - 0.7).max())
    off = np.stack(rec["landing_offset"])
    print("  landing offsets min/max", off.min((0, 1)), off.max((0, 1)))
    print("  total iterations per tick (mean over ticks)", np.mean(rec["iterations_mean"]), "slowest", np.mean(rec["iterations_max"]))
