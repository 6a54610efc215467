"""Developer probe: two independent B=256 batches in flight (two handles, two streams) against one at a time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cmpc_amd as cm
B, K = 256, 40
cfg, P, X0 = cm.synthetic.config2_perturbed_com(B, seed=0)
dP = torch.from_numpy(P.astype(np.float32)).cuda(); dX0 = torch.from_numpy(X0.astype(np.float32)).cuda()
for nfl in (1, 2, 3):
    solvers = [cm.BatchSolver(cfg, B) for _ in range(nfl)]
    outs = [(torch.empty_like(dX0), torch.empty((B, 8), dtype=torch.float32, device="cuda")) for _ in range(nfl)]
    streams = [s.launch_stream for s in solvers]
    lib = cm._capi.lib()
    def run(k):
        for i in range(k):
            j = i % nfl
            rc = lib.cmpc_solve_device(solvers[j]._h, dP.data_ptr(), dX0.data_ptr(), outs[j][0].data_ptr(), outs[j][1].data_ptr(), streams[j].cuda_stream)
            assert rc == 0
    run(6); torch.cuda.synchronize()
    t = time.perf_counter(); run(K); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("batches in flight", nfl, "-> %.0f solves/s (%.3f ms per batch)" % (B * K / dt, dt / K * 1e3), flush=True)
    for s in solvers: s.close()
