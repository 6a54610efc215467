"""Developer probe (GPU box): histogram of the iteration counts of the bench batch (config 2, B = 256, seed 0) and of a few more seeds, with the
final residual / last step of the slowest problems: the batch is as slow as its slowest problem."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmpc_amd as cm
for seed in (0, 1, 2, 3):
    cfg, P, X0 = cm.synthetic.config2_perturbed_com(256, seed=seed)
    s = cm.BatchSolver(cfg, 256)
    X, info, rc = s.solve_host(P.astype(np.float32), X0.astype(np.float32))
    it = info[:, 0].astype(int)
    print("seed", seed, "rc", rc, "histogram", {int(k): int((it == k).sum()) for k in np.unique(it)}, "mean %.2f" % it.mean())
    for b in np.argsort(-it)[:4]:
        print("   problem %3d: %d iterations, kkt %.2e, mu %.2e, last step %.2e, safeguards %d" % (b, it[b], info[b, 1], info[b, 2], info[b, 7], int(info[b, 3])))
    s.close()
