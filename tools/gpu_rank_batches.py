"""Developer probe: the batches the 8 ranks of `bench.py --gpus 8` would solve (seed 1000 * rank), one after the other on this GPU."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cmpc_amd as cm
for rank in range(8):
    cfg, P, X0 = cm.synthetic.config2_perturbed_com(256, seed=1000 * rank)
    s = cm.BatchSolver(cfg, 256)
    dP = torch.from_numpy(P.astype(np.float32)).cuda(); dX0 = torch.from_numpy(X0.astype(np.float32)).cuda()
    dX, dI = s.solve_device(dP, dX0); torch.cuda.synchronize()
    ms = []
    for _ in range(10):
        s.solve_device(dP, dX0, dX, dI); torch.cuda.synchronize(); ms.append(s.last_solve_ms())
    it = dI.cpu().numpy()[:, 0]
    print("rank", rank, "kernel ms %.3f" % np.median(ms), "iterations mean %.2f max %d" % (it.mean(), it.max()), np.bincount(it.astype(int)), flush=True)
    s.close()
