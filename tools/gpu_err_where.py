"""Developer probe (GPU box): where the largest all-knot force / CoM-velocity errors against the float64 oracle sit."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmpc_amd as cm
from tests import parity
from oracle import oracle_lib as ol, problem_nlp
B = int(os.environ.get("SWEEP_B", "512"))
seed = int(os.environ.get("SWEEP_SEED", "101"))
for name, gen in (("cfg5", cm.synthetic.config5_footstep_candidates), ("cfg3", cm.synthetic.config3_external_push)):
    cfg, P, X0 = gen(B, seed=seed)
    N = cfg.N
    L = cm.Layout(N)
    P32, X032 = P.astype(np.float32), X0.astype(np.float32)
    Xr, infr = ol.ref_solve_batch(problem_nlp.oracle_cfg(cfg), P32.astype(np.float64), X032.astype(np.float64), ol.ipm_opts(tol=1e-9, mu_min=1e-10), nthreads=16)
    Xr2, _ = ol.ref_solve_batch(problem_nlp.oracle_cfg(cfg), P32.astype(np.float64), X032.astype(np.float64), ol.ipm_opts(tol=1e-6, mu_min=5e-8), nthreads=16)
    s = cm.BatchSolver(cfg, B)
    X, info, rc = s.solve_host(P32, X032)
    errs = [parity.errors(N, P32[b], X[b], Xr[b]) for b in range(B)]
    errs2 = [parity.errors(N, P32[b], Xr2[b], Xr[b]) for b in range(B)]
    print(name, "GPU vs tight oracle: forces", max(e["forces"] for e in errs), "dcom", max(e["dcom"] for e in errs))
    print(name, "f64 oracle at the GPU's tolerances (1e-6, mu_min 5e-8) vs tight oracle: forces", max(e["forces"] for e in errs2), "dcom", max(e["dcom"] for e in errs2),
          "force0", max(e["force0"] for e in errs2))
    for key in ("forces", "dcom"):
        order = np.argsort([-e[key] for e in errs])[:4]
        for b in order:
            d = X[b].astype(np.float64) - Xr[b]
            if key == "forces":
                best = (0, None)
                for c in range(2):
                    for j in range(4):
                        dd = L.x_force(d, c, j)
                        k, a = np.unravel_index(np.abs(dd).argmax(), dd.shape)
                        if abs(dd[k, a]) > best[0]:
                            best = (abs(dd[k, a]), (c, j, int(k), int(a)))
                c, j, k, a = best[1]
                gam = P32[b, L.p_gam[c] + k]
                print(f"  {key} prob {b} err {errs[b][key]:.2e} iters {int(info[b,0])}: contact {c} corner {j} knot {k} axis {a} gamma {gam} f_ref {L.x_force(Xr[b], c, j)[k]} d {L.x_force(d, c, j)[k]}")
            else:
                dd = L.x_dcom(d)
                k, a = np.unravel_index(np.abs(dd).argmax(), dd.shape)
                print(f"  {key} prob {b} err {errs[b][key]:.2e} iters {int(info[b,0])}: knot {k} axis {a} ref {L.x_dcom(Xr[b])[k]} d {dd[k]}  max|dcom| {np.abs(L.x_dcom(Xr[b])).max():.3f}")
