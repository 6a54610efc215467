"""Developer probe (GPU box): what a B = 256 batch costs = its slowest problem.  Solves 40960 problems per config, cuts them
into 160 batches of 256 and prints the mean over batches of the per-batch maximum iteration count next to the overall mean.
Knobs through the environment (STEPTOL, CMPC_SIGMA_MIN, ...)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmpc_amd as cm
for name, gen in (("cfg2", cm.synthetic.config2_perturbed_com), ("cfg3", cm.synthetic.config3_external_push)):
    its = []
    bad = 0
    for sd in range(200, 210):
        cfg, P, X0 = gen(4096, seed=sd)
        kw = {"step_tolerance": float(os.environ["STEPTOL"])} if os.environ.get("STEPTOL") else {}
        s = cm.BatchSolver(cfg, 4096, **kw)
        X, info, rc = s.solve_host(P.astype(np.float32), X0.astype(np.float32))
        s.close()
        its.append(info[:, 0]); bad += int((info[:, 5] != 0).sum())
    its = np.concatenate(its).reshape(-1, 256)
    print(name, os.environ.get("TAG", ""), "mean %.3f  mean of batch max %.3f  overall max %d  not converged %d" % (its.mean(), its.max(1).mean(), its.max(), bad), flush=True)
