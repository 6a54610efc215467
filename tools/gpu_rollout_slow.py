"""Developer probe (GPU box): the stragglers of a long walking roll-out.  Runs the roll-out (default 2048 problems x 120 ticks),
keeps every warm-started solve that needed more than PROBE_THR iterations, then solves each of them again alone with the
-DCMPC_PROFILE build and prints its per-iteration trace (mu, residuals, step, step lengths)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import cmpc_amd as cm
cfg = cm.config.ergocub_gazebo_v1(20, 0.06)
B, ticks, thr = int(os.environ.get("PROBE_B", "2048")), int(os.environ.get("PROBE_TICKS", "120")), int(os.environ.get("PROBE_THR", "13"))
path = os.path.join(ROOT, "gpurun_out", "slow_dump.npz")
if not os.path.exists(path) or os.environ.get("PROBE_REDO"):
    rng = np.random.default_rng(9)
    com0 = np.array([0.0, 0.0, 0.7]) + rng.uniform(-0.01, 0.01, (B, 3))
    dcom0 = rng.uniform(-0.05, 0.05, (B, 3)); h0 = rng.uniform(-0.02, 0.02, (B, 3))
    push = np.zeros((B, 3)); push[:, :2] = rng.uniform(-30.0, 30.0, (B, 2)) / cm.synthetic.ROBOT_MASS
    ro = cm.rollout.WalkingRollout(cfg, B, plan=cm.rollout.walking_plan(cfg, steps=11))
    slow = []
    rec = ro.run(ticks, com0, dcom0, h0, push=push, push_ticks=3, slow=(thr, slow))
    it = np.array(rec["iterations_max"]); ms = np.array(rec["solve_ms"])
    print("ticks", ticks, "iterations max per tick:", it.tolist())
    print("solve ms per tick: p50 %.3f p99 %.3f max %.3f" % (np.median(ms), np.percentile(ms, 99), ms.max()))
    print("slow solves (>%d iterations):" % thr, [(t, b, int(i[0]), int(i[3])) for t, b, _, _, i in slow])
    np.savez(path, P=np.array([s[2] for s in slow]), X0=np.array([s[3] for s in slow]), info=np.array([s[4] for s in slow]),
             tick=np.array([s[0] for s in slow]), prob=np.array([s[1] for s in slow]))
d = np.load(path)
cm._capi._lib = None
cm._capi.LIB_PATH = os.path.join(os.path.dirname(cm._capi.LIB_PATH), "libcmpc_hip_prof.so")
n = len(d["P"])
print("traces of", min(n, int(os.environ.get("PROBE_NTRACE", "6"))), "of", n, "slow solves")
for q in np.argsort(-d["info"][:, 0])[:int(os.environ.get("PROBE_NTRACE", "6"))]:
    s1 = cm.BatchSolver(cfg, 1)
    dP, dX0 = torch.from_numpy(d["P"][q:q + 1]).cuda(), torch.from_numpy(d["X0"][q:q + 1]).cuda()
    dX, dI = s1.solve_device(dP, dX0, warm=True)
    torch.cuda.synchronize()
    info1 = dI.cpu().numpy()
    tr = (C.c_float * 512)()
    cm._capi.lib().cmpc_trace_read(tr)
    tr = np.array(tr[:]).reshape(64, 8)
    print("tick", int(d["tick"][q]), "problem", int(d["prob"][q]), "in the batch:", d["info"][q, [0, 3, 5]], "alone:", info1[0, [0, 3, 5]])
    print("  it    mu_cur      ep       ec(max tz)  step     ap    ad    sigma    mu_t")
    for i in range(min(int(info1[0, 0]), 60)):
        print("  %2d  %.2e %.2e %.2e %.2e %.3f %.3f %.2e %.2e" % ((i,) + tuple(tr[i])))
    s1.close()
