"""Developer probe (GPU box): solve golden + synthetic batches, print errors / iterations / timing."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmpc_amd as cm  # noqa: E402
from tests import parity  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def golden(name, gen):
    cfg = gen()[0]
    d = np.load(os.path.join(GOLD, f"argmin_{name}.npz"))
    B = d["P"].shape[0]
    s = cm.BatchSolver(cfg, B)
    X, info, rc = s.solve_host(d["P"], d["X0"])
    print(name, "rc", rc, "iters", info[:, 0], "status", info[:, 5], "gn", info[:, 3], "kkt", info[:, 1])
    for b in range(B):
        e = parity.errors(cfg.N, d["P"][b], X[b], d["x_star"][b])
        print("   ", b, {k: "%.2e" % v for k, v in e.items()})
    s.close()


def timing(name, gen, B):
    import torch
    cfg, P, X0 = gen(B)
    s = cm.BatchSolver(cfg, B)
    dP = torch.tensor(P, dtype=torch.float32, device="cuda")
    dX0 = torch.tensor(X0, dtype=torch.float32, device="cuda")
    dX, dI = s.solve_device(dP, dX0)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(5):
        s.solve_device(dP, dX0, dX, dI)
    torch.cuda.synchronize()
    dt = (time.time() - t) / 5
    info = dI.cpu().numpy()
    print(name, "B", B, "ms/batch %.3f" % (dt * 1e3), "solves/s %.0f" % (B / dt), "kernel ms %.3f" % s.last_solve_ms(),
          "iters mean %.1f max %d" % (info[:, 0].mean(), info[:, 0].max()), "bad", int((info[:, 5] != 0).sum()),
          "cycles/solve %.3g" % info[:, 6].mean())
    s.close()


if __name__ == "__main__":
    golden("cfg1", cm.synthetic.config1_plumbing)
    golden("cfg2", lambda: cm.synthetic.config2_perturbed_com(8))
    golden("cfg3", lambda: cm.synthetic.config3_external_push(8))
    timing("cfg2", cm.synthetic.config2_perturbed_com, 256)
    timing("cfg3", cm.synthetic.config3_external_push, 256)
    timing("cfg3", cm.synthetic.config3_external_push, 4096)
