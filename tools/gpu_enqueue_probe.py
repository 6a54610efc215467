"""Developer probe (GPU box): is the bench loop asynchronous?  Time to ENQUEUE K solves against the time until they are done."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cmpc_amd as cm
B, K = 256, 30
cfg, P, X0 = cm.synthetic.config2_perturbed_com(B, seed=0)
dP = torch.from_numpy(P.astype(np.float32)).cuda(); dX0 = torch.from_numpy(X0.astype(np.float32)).cuda()
dX = torch.empty_like(dX0); dInfo = torch.empty((B, 8), dtype=torch.float32, device="cuda")
s = cm.BatchSolver(cfg, B)
for timing in (True, False):
    s.set_timing(timing)
    for _ in range(3):
        s.solve_device(dP, dX0, dX, dInfo)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        s.solve_device(dP, dX0, dX, dInfo)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("library events", timing, ": enqueue of %d solves %.3f ms (%.1f us each), all done after %.3f ms (%.4f ms per solve)" % (K, (t1 - t0) * 1e3, (t1 - t0) / K * 1e6, (t2 - t0) * 1e3, (t2 - t0) / K * 1e3))
# straight through the C ABI, no python wrapper work per call
lib = cm._capi.lib()
args = (s._h, dP.data_ptr(), dX0.data_ptr(), dX.data_ptr(), dInfo.data_ptr(), s.launch_stream.cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    lib.cmpc_solve_device(*args)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("C ABI direct: enqueue %.1f us each, %.4f ms per solve" % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e3))
