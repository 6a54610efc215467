"""Developer probe (GPU box): the same problems through the resident variant (factors="lds") and the HBM-factor variant
(factors="hbm"), several horizons, generators and seeds; prints the worst disagreement and any unconverged problem."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmpc_amd as cm
from tests import parity
B = 48
worst = {}
for N in (10, 12, 13, 15, 17, 20, 22):
    for seed in (7, 44, 91):
        cfg, P, X0 = cm.synthetic.config3_external_push(B, N=N, seed=seed)
        P32, X032 = P.astype(np.float32), X0.astype(np.float32)
        out = {}
        for f in ("lds", "hbm"):
            s = cm.BatchSolver(cfg, B, factors=f)
            X, info, rc = s.solve_host(P32, X032)
            out[f] = (X, info)
            s.close()
        bad = int((out["lds"][1][:, 5] != 0).sum()), int((out["hbm"][1][:, 5] != 0).sum())
        dit = int(np.abs(out["lds"][1][:, 0] - out["hbm"][1][:, 0]).max())
        w = {}
        for b in range(B):
            e = parity.errors(cfg.N, P32[b], out["hbm"][0][b], out["lds"][0][b])
            for k, v in e.items():
                w[k] = max(w.get(k, 0.0), v)
        print(f"N {N} seed {seed}: unconverged lds/hbm {bad}, max iteration difference {dit}, worst disagreement " + " ".join(f"{k} {v:.1e}" for k, v in w.items()), flush=True)
for gen, name in ((cm.synthetic.config5_footstep_candidates, "cfg5"), (cm.synthetic.config2_perturbed_com, "cfg2")):
    cfg, P, X0 = gen(B, seed=17)
    s = cm.BatchSolver(cfg, B)
    X, info, rc = s.solve_host(P.astype(np.float32), X0.astype(np.float32))
    print(name, "N", cfg.N, "unconverged", int((info[:, 5] != 0).sum()))
