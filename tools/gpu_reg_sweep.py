import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmpc_amd as cm
from tests import parity
from oracle import oracle_lib as ol, problem_nlp
B = 128
for name, gen in (("cfg2", cm.synthetic.config2_perturbed_com), ("cfg3", cm.synthetic.config3_external_push)):
    cfg, P, X0 = gen(B)
    P32, X032 = P.astype(np.float32), X0.astype(np.float32)
    oc = problem_nlp.oracle_cfg(cfg)
    Xr, infr = ol.ref_solve_batch(oc, P32.astype(np.float64), X032.astype(np.float64), ol.ipm_opts(tol=1e-9, mu_min=1e-10), nthreads=16)
    for reg in ("1e-5", "1e-4", "1e-3", "1e-2", "1e-1"):
        for steptol in (1e-5, 1e-4):
            os.environ["CMPC_REG"] = reg
            s = cm.BatchSolver(cfg, B, tolerance=1e-6, mu_min=1e-7, step_tolerance=steptol)
            X, info, rc = s.solve_host(P32, X032)
            e = [parity.errors(cfg.N, P32[b], X[b], Xr[b]) for b in range(B)]
            print(name, "reg", reg, "step_tol %.0e" % steptol, "iters mean %.1f max %d bad %d gn %d" % (info[:, 0].mean(), info[:, 0].max(), (info[:, 5] != 0).sum(), info[:, 3].sum()),
                  "| max err com %.1e force0 %.1e pos %.1e" % (max(x["com"] for x in e), max(x["force0"] for x in e), max(x["pos"] for x in e)))
            s.close()
