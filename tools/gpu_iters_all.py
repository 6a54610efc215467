"""Developer probe (GPU box): iteration statistics and safeguard counts (info[3]) of the standard workloads."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmpc_amd as cm
for name, gen, B in (("cfg2", cm.synthetic.config2_perturbed_com, 4096), ("cfg3", cm.synthetic.config3_external_push, 4096),
                     ("cfg4 shard", lambda B: cm.synthetic.config4_monte_carlo(65536, shard=(0, B)), 8192), ("cfg5", cm.synthetic.config5_footstep_candidates, 4096)):
    cfg, P, X0 = gen(B)
    s = cm.BatchSolver(cfg, B)
    X, info, rc = s.solve_host(P.astype(np.float32), X0.astype(np.float32))
    print(name, "B", B, "iters mean %.3f max %d" % (info[:, 0].mean(), info[:, 0].max()), "not converged", int((info[:, 5] != 0).sum()),
          "safeguards/fallbacks: problems", int((info[:, 3] > 0).sum()), "total", int(info[:, 3].sum()), "kernel ms %.3f" % s.last_solve_ms(), flush=True)
    s.close()
