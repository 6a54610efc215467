"""Full-size runs of BASELINE.json configs 4 and 5 on one GPU: convergence and throughput."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cmpc_amd as cm
for name, gen, B in (("config4 (65536 Monte-Carlo disturbances, one GPU)", cm.synthetic.config4_monte_carlo, 65536),
                     ("config5 (8192 footstep candidates, N=30)", cm.synthetic.config5_footstep_candidates, 8192)):
    t = time.time(); cfg, P, X0 = gen(B); tg = time.time() - t
    s = cm.BatchSolver(cfg, B)
    dP = torch.from_numpy(P.astype(np.float32)).cuda(); dX0 = torch.from_numpy(X0.astype(np.float32)).cuda()
    dX, dI = s.solve_device(dP, dX0); torch.cuda.synchronize()
    t = time.time(); s.solve_device(dP, dX0, dX, dI); torch.cuda.synchronize(); dt = time.time() - t
    info = dI.cpu().numpy()
    print(name, "gen %.1fs" % tg, "solve %.1f ms -> %.0f solves/s" % (dt * 1e3, B / dt), "iters mean %.2f max %d" % (info[:, 0].mean(), info[:, 0].max()),
          "converged %d/%d" % ((info[:, 5] == 0).sum(), B), "gn fallbacks %d" % info[:, 3].sum(), flush=True)
    s.close(); del dP, dX0, dX, dI
