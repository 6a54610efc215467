"""Developer probe (GPU box): worst parity errors of the 10 reference-code argmin goldens (what test_matches_argmin_computed_on_the_reference_code asserts), per quantity,
for the library named by CMPC_LIB -- how much margin a change of the arithmetic leaves."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmpc_amd as cm
from tests import parity
g = os.path.join(ROOT, "tests", "golden")
for name in ("walk", "yaw", "push", "ssend", "stand"):
    for which in ("tmp", "jit"):
        d = np.load(os.path.join(g, f"argmin_ref_{name}_{which}.npz"))
        cfg = cm.config.generated_code_weights(which, 12, 0.1)
        s = cm.BatchSolver(cfg, d["P"].shape[0])
        X, info, rc = s.solve_host(d["P"], d["X0"])
        w = parity.worst_errors(cfg.N, d["P"], X, d["x_star"])
        print(name, which, "rc", rc, "iterations mean %.2f max %d" % (info[:, 0].mean(), info[:, 0].max()), " ".join("%s %.2e" % (k, w[k]) for k in ("com", "dcom", "h", "force0", "forces")), flush=True)
        s.close()
