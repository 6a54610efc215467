"""Developer probe (GPU box): HBM-factor variant, N = 20, exactly 1 / 2 / 3 workgroups per CU (B = 256 / 512 / 768) of the SAME
256 problems repeated: the time of a batch against the number of workgroups sharing a CU."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cmpc_amd as cm
cfg, P, X0 = cm.synthetic.config3_external_push(256)
for rep in (1, 2, 3, 6, 12):
    B = 256 * rep
    P32 = np.tile(P.astype(np.float32), (rep, 1)); X032 = np.tile(X0.astype(np.float32), (rep, 1))
    s = cm.BatchSolver(cfg, B, factors="hbm")
    dP, dX0 = torch.from_numpy(P32).cuda(), torch.from_numpy(X032).cuda()
    dX, dI = s.solve_device(dP, dX0); torch.cuda.synchronize()
    ms = []
    for _ in range(7):
        s.solve_device(dP, dX0, dX, dI); torch.cuda.synchronize(); ms.append(s.last_solve_ms())
    info = dI.cpu().numpy()
    print(f"B={B} ({rep} x the same 256 problems): {np.median(ms):.3f} ms  -> {B / np.median(ms) * 1e3:.0f} solves/s, max iterations {int(info[:, 0].max())}, mean {info[:, 0].mean():.2f}", flush=True)
    s.close()
