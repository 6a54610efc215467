"""Developer probe: convergence count by batch size and factor placement."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import cmpc_amd as cm
for B in ([int(a) for a in sys.argv[1:]] or [256, 257, 300, 512, 1024, 4096]):
    cfg, P, X0 = cm.synthetic.config2_perturbed_com(B, seed=7)
    s = cm.BatchSolver(cfg, B)
    X, info, rc = s.solve_host(P, X0)
    it = info[:, 0].astype(int)
    bad = np.nonzero(info[:, 5] != 0)[0]
    print("B", B, "rc", rc, "mean it %.2f" % it.mean(), "bad", len(bad), "first bad", bad[:8], "status", info[bad[:4], 5], "nan", int(np.isnan(X).any(axis=1).sum()), "good idx", np.nonzero(info[:, 5] == 0)[0][:12], "bad/256-chunk", [int((info[i:i+256, 5] != 0).sum()) for i in range(0, B, 256)], flush=True)
    s.close()
