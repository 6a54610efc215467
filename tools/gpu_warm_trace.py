"""Developer probe (GPU box): dump the tick of a walking roll-out whose warm-started solve stalls, then re-solve the
first failing problem with the -DCMPC_PROFILE build and print its iteration trace."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import cmpc_amd as cm
cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
B, tick = 256, int(os.environ.get("PROBE_TICK", "7"))
path = os.path.join(ROOT, "gpurun_out", "warm_dump.npz")
if not os.path.exists(path):
    rng = np.random.default_rng(5)
    com0 = np.array([0.0, 0.0, 0.7]) + rng.uniform(-0.01, 0.01, (B, 3))
    dcom0 = rng.uniform(-0.05, 0.05, (B, 3)); h0 = rng.uniform(-0.02, 0.02, (B, 3))
    push = np.zeros((B, 3)); push[:, :2] = rng.uniform(-20.0, 20.0, (B, 2)) / cm.synthetic.ROBOT_MASS
    ro = cm.rollout.WalkingRollout(cfg, B)
    rec = ro.run(tick + 1, com0, dcom0, h0, push=push, push_ticks=3, dump=(tick, path))
    print("dumped; failed at that tick:", len(rec["failed_info"][tick]))
    sys.exit(0)
d = np.load(path)
cm._capi.LIB_PATH = os.path.join(os.path.dirname(cm._capi.LIB_PATH), "libcmpc_hip_prof.so")
os.environ["CMPC_FORCE_WARM"] = "1"
s = cm.BatchSolver(cfg, B)
X, info, rc = s.solve_host(d["P"], d["X0"])
bad = np.where(info[:, 5] != 0)[0]
print("failing problems", bad[:10], "of", len(bad))
b = int(bad[0]) if len(bad) else 0
s1 = cm.BatchSolver(cfg, 1)
X1, info1, rc = s1.solve_host(d["P"][b:b + 1], d["X0"][b:b + 1])
tr = (C.c_float * 512)()
cm._capi.lib().cmpc_trace_read(tr)
tr = np.array(tr[:]).reshape(64, 8)
print("problem", b, "info", info1[0])
print("iteration trace:  mu_cur      ep       ec(max tz)  step     ap    ad    sigma    mu_t")
for i in range(min(int(info1[0, 0]), 40)):
    print("  it %2d  %.2e %.2e %.2e %.2e %.3f %.3f %.2e %.2e" % ((i,) + tuple(tr[i])))
L = cm.Layout(cfg.N)
p, x0 = d["P"][b], d["X0"][b]
print("gamma left ", p[L.p_gam[0]:L.p_gam[0] + 20]); print("gamma right", p[L.p_gam[1]:L.p_gam[1] + 20])
for c in range(2):
    print("x0 fz corner0 contact", c, np.round(L.x_force(x0, c, 0)[:, 2], 3))
    print("x0 pos contact", c, np.round(L.x_pos(x0, c)[:, :2], 4).tolist())
    print("nominal", c, np.round(p[L.p_nom[c]:L.p_nom[c] + 63].reshape(21, 3)[:, :2], 4).tolist())
    print("current", p[L.p_cur[c]:L.p_cur[c] + 3])
