"""Developer probe (GPU box): iteration traces (mu, residuals, step lengths) of the slowest problems of the bench batch,
from the -DCMPC_PROFILE build (problem re-solved alone so that it is workgroup 0)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import cmpc_amd as cm
cm._capi.LIB_PATH = os.path.join(os.path.dirname(cm._capi.LIB_PATH), "libcmpc_hip_prof.so")
gen = {"cfg2": cm.synthetic.config2_perturbed_com, "cfg3": cm.synthetic.config3_external_push}[os.environ.get("TRACE_CFG", "cfg2")]
cfg, P, X0 = gen(256)
s = cm.BatchSolver(cfg, 256)
X, info, rc = s.solve_host(P, X0)
print("iterations histogram", np.bincount(info[:, 0].astype(int)))
order = np.argsort(-info[:, 0])
s1 = cm.BatchSolver(cfg, 1)
for b in list(order[:3]) + [int(order[128])]:
    X1, info1, rc = s1.solve_host(P[b:b + 1], X0[b:b + 1])
    tr = (C.c_float * 512)()
    cm._capi.lib().cmpc_trace_read(tr)
    tr = np.array(tr[:]).reshape(64, 8)
    print(f"problem {b}: iterations {int(info1[0, 0])}")
    print("      mu_cur      ep       ec(max tz)  step     ap    ad    sigma    mu_t")
    for i in range(int(info1[0, 0])):
        print("  it %2d  %.2e %.2e %.2e %.2e %.3f %.3f %.2e %.2e" % ((i,) + tuple(tr[i])))
