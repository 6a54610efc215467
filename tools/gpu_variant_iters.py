"""Developer probe (GPU box): iteration counts of the resident and the HBM-factor variant on the problems of test_hbm_factor_and_resident_variants_agree
(config-3 generator, 32 problems, seed 44) and on 512 problems of config 3, for the library named by CMPC_LIB."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmpc_amd as cm  # noqa: E402

for N in (13, 20):
    cfg, P, X0 = cm.synthetic.config3_external_push(32, N=N, seed=44)
    P32, X032 = P.astype(np.float32), X0.astype(np.float32)
    for factors in ("lds", "hbm"):
        s = cm.BatchSolver(cfg, 32, factors=factors)
        X, info, rc = s.solve_host(P32, X032)
        print(os.environ.get("CMPC_LIB", "default"), "N", N, factors, "iters", info[:, 0].astype(int).tolist(), "gn", int(info[:, 3].sum()))
        s.close()
cfg, P, X0 = cm.synthetic.config3_external_push(512, seed=7)
P32, X032 = P.astype(np.float32), X0.astype(np.float32)
for factors in ("lds", "hbm"):
    s = cm.BatchSolver(cfg, 512, factors=factors)
    X, info, rc = s.solve_host(P32, X032)
    it = info[:, 0]
    print(os.environ.get("CMPC_LIB", "default"), "cfg3 x512", factors, "mean %.3f max %d" % (it.mean(), it.max()), "hist", np.bincount(it.astype(int)).tolist())
    s.close()
