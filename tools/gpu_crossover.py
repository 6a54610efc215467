"""Developer probe (GPU box): resident (factors="lds") against HBM-factor ("hbm") variants at batch sizes around the CU count; solves/s of config-2 and config-3 problems."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cmpc_amd as cm
for name, gen in (("cfg2", cm.synthetic.config2_perturbed_com), ("cfg3", cm.synthetic.config3_external_push)):
    for B in (256, 384, 512, 768, 1024):
        row = []
        for fac in ("lds", "hbm"):
            cfg, P, X0 = gen(B)
            s = cm.BatchSolver(cfg, B, factors=fac)
            dP, dX0 = torch.from_numpy(P.astype(np.float32)).cuda(), torch.from_numpy(X0.astype(np.float32)).cuda()
            for _ in range(3):
                s.solve_device(dP, dX0)
            torch.cuda.synchronize()
            ms = []
            for _ in range(10):
                s.solve_device(dP, dX0); torch.cuda.synchronize(); ms.append(s.last_solve_ms())
            row.append(B / np.median(ms))
            s.close()
        print(name, "B", B, "resident %.1f k solves/s | HBM-factor %.1f k" % (row[0], row[1]), flush=True)
