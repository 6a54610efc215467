import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmpc_amd as cm
from tests import parity
GOLD = os.path.join(ROOT, "tests", "golden")
for name, gen in (("cfg2", lambda: cm.synthetic.config2_perturbed_com(8)), ("cfg3", lambda: cm.synthetic.config3_external_push(8))):
    cfg = gen()[0]
    d = np.load(os.path.join(GOLD, f"argmin_{name}.npz"))
    for mi in (40,):
        s = cm.BatchSolver(cfg, 8, max_iterations=mi)
        X, info, rc = s.solve_host(d["P"], d["X0"])
        e = [parity.errors(cfg.N, d["P"][b], X[b], d["x_star"][b]) for b in range(8)]
        print(name, "iters", info[:,0], "maxit", mi, "err %.1e ep %.1e es %.1e mu %.1e" % (info[:,1].max(), info[:,4].max(), info[:,7].max(), info[:,2].max()),
              "com %.1e f0 %.1e pos %.1e" % (max(x["com"] for x in e), max(x["force0"] for x in e), max(x["pos"] for x in e)))
        s.close()
