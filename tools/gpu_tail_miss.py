"""Developer probe (GPU box): config-5 problems the tail polish does not fix -- per problem the safeguard word info[3] and the per-knot
force / CoM-velocity error against the tight oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmpc_amd as cm
from tests import parity
from oracle import oracle_lib as ol, problem_nlp
B = 512
for seed in [int(v) for v in os.environ.get("SWEEP_SEEDS", "101,202,303,404,505").split(",")]:
    cfg, P, X0 = cm.synthetic.config5_footstep_candidates(B, seed=seed)
    N, L = cfg.N, cm.Layout(cfg.N)
    P32, X032 = P.astype(np.float32), X0.astype(np.float32)
    Xr, infr = ol.ref_solve_batch(problem_nlp.oracle_cfg(cfg), P32.astype(np.float64), X032.astype(np.float64), ol.ipm_opts(tol=1e-9, mu_min=1e-10), nthreads=16)
    s = cm.BatchSolver(cfg, B)
    X, info, rc = s.solve_host(P32, X032)
    s.close()
    errs = [parity.errors(N, P32[b], X[b], Xr[b]) for b in range(B)]
    bad = [b for b in range(B) if errs[b]["forces"] > 5e-5 or errs[b]["dcom"] > 9e-5]
    print("seed", seed, "worst forces %.2e dcom %.2e" % (max(e["forces"] for e in errs), max(e["dcom"] for e in errs)), "bad problems", bad)
    for b in bad[:4]:
        d = X[b].astype(np.float64) - Xr[b]
        fr = max(np.abs(L.x_force(Xr[b], c, j)).max() for c in range(2) for j in range(4))
        fk = np.zeros(N)
        for c in range(2):
            for j in range(4):
                fk = np.maximum(fk, np.abs(L.x_force(d, c, j)).max(1) / fr)
        print("  problem", b, "forces %.2e dcom %.2e" % (errs[b]["forces"], errs[b]["dcom"]), "info: its %d kkt %.1e mu %.1e safeguards %d status %d" %
              (info[b, 0], info[b, 1], info[b, 2], info[b, 3], info[b, 5]))
        print("    force err per knot x1e-5 (last 8):", np.round(fk[-8:] * 1e5, 1).tolist(), " gamma L/R last 4:", P32[b, L.p_gam[0] + N - 4:L.p_gam[0] + N].tolist(), P32[b, L.p_gam[1] + N - 4:L.p_gam[1] + N].tolist())
        dd = np.abs(L.x_dcom(d)).max(1) / max(np.abs(L.x_dcom(Xr[b])).max(), 1e-2)
        print("    dcom err per knot x1e-5 (last 8):", np.round(dd[-8:] * 1e5, 1).tolist())
        print("    dcom err (abs x1e-5) last 3 knots xyz:", np.round(L.x_dcom(d)[-3:] * 1e5, 2).tolist(), "net force err last 6 stages xyz x1e-5:", np.round(sum(P32[b, L.p_gam[c]:L.p_gam[c] + N][-6:, None] * L.x_force(d, c, j)[-6:] for c in range(2) for j in range(4)) * 1e5, 1).tolist())
