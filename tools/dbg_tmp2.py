import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmpc_amd as cm
if os.environ.get("DBG_LIB"): cm._capi.LIB_PATH = os.path.join(os.path.dirname(cm._capi.LIB_PATH), os.environ["DBG_LIB"])
for N in (13, 20):
    cfg, P, X0 = cm.synthetic.config3_external_push(400, N=N, seed=44)
    for B in (1, 8, 400):
        s = cm.BatchSolver(cfg, B)
        X, info, rc = s.solve_host(P[:B].astype(np.float32), X0[:B].astype(np.float32))
        print(os.environ.get("DBG_LIB", "shipped"), "factors", os.environ.get("CMPC_FACTORS", "auto"), "N", N, "B", B, "not converged", int((info[:, 5] != 0).sum()), "iters", info[:4, 0], flush=True)
        s.close()
