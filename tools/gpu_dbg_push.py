"""Developer probe (GPU box): push-recovery problems through the library named by CMPC_LIB; iterations, status, safeguard word."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import cmpc_amd as cm
for gen in (cm.synthetic.push_recovery_n12, cm.synthetic.walking_push_n12):
    cfg, P, X0 = gen("tmp")
    s = cm.BatchSolver(cfg, P.shape[0])
    X, info, rc = s.solve_host(P.astype(np.float32), X0.astype(np.float32))
    L = cm.Layout(cfg.N)
    print(gen.__name__, "rc", rc, "iters", info[:, 0].astype(int).tolist(), "status", info[:, 5].astype(int).tolist(), "safeguards", info[:, 3].astype(int).tolist())
