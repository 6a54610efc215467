"""Developer probe: iteration-count distribution (mean, tail) over many seeded problems of configs 2 and 3."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import cmpc_amd as cm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name, gen in (("cfg2", cm.synthetic.config2_perturbed_com), ("cfg3", cm.synthetic.config3_external_push)):
    cfg, P, X0 = gen(B, seed=7)
    s = cm.BatchSolver(cfg, B, **({'mu_init': float(os.environ['MU_INIT'])} if 'MU_INIT' in os.environ else {}))
    X, info, rc = s.solve_host(P, X0)
    it = info[:, 0].astype(int)
    # the B = 256 bench is bound by the slowest problem of each batch of 256
    mx = it[: (B // 256) * 256].reshape(-1, 256).max(axis=1)
    if rc != 0:
        print("solve_host rc", rc, s.last_error)
    print(name, "env", {k: v for k, v in os.environ.items() if k.startswith("CMPC_") or k == "MU_INIT"}, "mean %.2f" % it.mean(), "hist", np.bincount(it),
          "bad", int((info[:, 5] != 0).sum()), "mean of per-256 max %.2f" % mx.mean(), flush=True)
    s.close()
