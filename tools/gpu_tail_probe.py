"""Developer probe (GPU box): what the tail polish does on config 5 -- how many problems it runs on, the worst errors against the tight
float64 oracle with and without it, the per-knot profile of the force error, and whether the worst problems were polished."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmpc_amd as cm
from tests import parity
from oracle import oracle_lib as ol, problem_nlp
B = int(os.environ.get("SWEEP_B", "512")); seed = int(os.environ.get("SWEEP_SEED", "101"))
cfg, P, X0 = cm.synthetic.config5_footstep_candidates(B, seed=seed)
N, L = cfg.N, cm.Layout(cfg.N)
P32, X032 = P.astype(np.float32), X0.astype(np.float32)
Xr, infr = ol.ref_solve_batch(problem_nlp.oracle_cfg(cfg), P32.astype(np.float64), X032.astype(np.float64), ol.ipm_opts(tol=1e-9, mu_min=1e-10), nthreads=16)
for kw in (dict(tail_stages=0), dict(), dict(tail_trigger=1e-6), dict(tail_stages=3, tail_iterations=4, tail_trigger=1e-6)):
    s = cm.BatchSolver(cfg, B, **kw)
    X, info, rc = s.solve_host(P32, X032)
    s.close()
    errs = [parity.errors(N, P32[b], X[b], Xr[b]) for b in range(B)]
    pol = info[:, 3] >= 100000
    print(kw, "rc", rc, "polished", int(pol.sum()), "iters mean %.2f" % info[:, 0].mean(), "info3 values", np.unique(info[:, 3]).tolist()[:8],
          "| worst forces %.2e dcom %.2e force0 %.2e" % (max(e["forces"] for e in errs), max(e["dcom"] for e in errs), max(e["force0"] for e in errs)))
    frel = np.zeros(N)
    for b in range(B):
        d = X[b].astype(np.float64) - Xr[b]
        fr = max(np.abs(L.x_force(Xr[b], c, j)).max() for c in range(2) for j in range(4))
        for c in range(2):
            for j in range(4):
                frel = np.maximum(frel, np.abs(L.x_force(d, c, j)).max(1) / fr)
    print("   force err per knot x1e-5:", np.round(frel * 1e5, 1).tolist())
    for b in np.argsort([-e["forces"] for e in errs])[:4]:
        print("   worst forces: problem", b, "err %.2e" % errs[b]["forces"], "polished", bool(pol[b]), "iters", int(info[b, 0]))
