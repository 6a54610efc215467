"""Developer probe (GPU box): a long walking roll-out (many contact transitions) to exercise the warm-start safeguards."""
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
ro = cm.rollout.WalkingRollout(cfg, B, plan=plan)
rec = ro.run(ticks, com0, dcom0, h0, push=push, push_ticks=3)
itmax = np.array(rec["iterations_max"]); itmean = np.array(rec["iterations_mean"])
nfail = sum(len(f) for f in rec["failed_info"])
print(f"B={B} ticks={ticks} solves={B * ticks} not converged {nfail} merge ok {all(rec['merge_ok'])}")
print("iterations per tick: mean %.2f, mean of slowest %.2f, overall max %d, ticks with a cold restart (max > 40): %d" % (itmean.mean(), itmax.mean(), itmax.max(), int((itmax > 40).sum())))
com = np.stack(rec["com"])
print("final CoM x range %.3f..%.3f, max |y| %.3f, max |z-0.7| %.3f" % (com[-1, :, 0].min(), com[-1, :, 0].max(), np.abs(com[:, :, 1]).max(), np.abs(com[:, :, 2] - 0.7).max()))
off = np.stack(rec["landing_offset"])
print("landing offsets (foot frame) min", off.min((0, 1)).round(4).tolist(), "max", off.max((0, 1)).round(4).tolist())
