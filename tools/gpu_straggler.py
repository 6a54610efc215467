"""Developer probe: iteration histogram of the bench batch and the per-iteration trace of its slowest problem."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import cmpc_amd as cm
cm._capi.LIB_PATH = os.path.join(os.path.dirname(cm._capi.LIB_PATH), "libcmpc_hip_prof.so")
gen = cm.synthetic.config2_perturbed_com if len(sys.argv) < 2 or sys.argv[1] == "config2" else cm.synthetic.config3_external_push
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg, P, X0 = gen(B, seed=int(os.environ.get('SEED', '7')))
s = cm.BatchSolver(cfg, B)
X, info, rc = s.solve_host(P, X0)
print("histogram of iterations:", np.bincount(info[:, 0].astype(int)))
for b in list(np.argsort(-info[:, 0])[:3]) + [int(np.argsort(info[:, 0])[B // 2])]:
    s1 = cm.BatchSolver(cfg, 1)
    X1, info1, rc = s1.solve_host(P[b:b + 1], X0[b:b + 1])
    tr = (C.c_float * 512)()
    cm._capi.lib().cmpc_trace_read(tr)
    tr = np.array(tr[:]).reshape(64, 8)
    print("problem", b, "iters", info1[0, 0], "gn", info1[0, 3], ":  mu_cur      ep       ec(max tz)  step     ap    ad    sigma    mu_t")
    for i in range(int(info1[0, 0])):
        print("  it %2d  %.2e %.2e %.2e %.2e %.3f %.3f %.2e %.2e" % ((i,) + tuple(tr[i])))
