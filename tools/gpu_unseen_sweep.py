"""Developer probe (GPU box): accuracy of the shipped defaults against the float64 oracle (converged to 1e-9) on seeds
that were never used for tuning; 5 seeds x 512 problems per config by default.  SWEEP_STEPTOL=a,b,.. adds
step-tolerance variants; SWEEP_B / SWEEP_SEEDS override the sizes."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmpc_amd as cm
from tests import parity
from oracle import oracle_lib as ol, problem_nlp

B = int(os.environ.get("SWEEP_B", "512"))
SEEDS = [int(s) for s in os.environ.get("SWEEP_SEEDS", "101,202,303,404,505").split(",")]
STEPTOL = [None] + [float(v) for v in os.environ.get("SWEEP_STEPTOL", "").split(",") if v]
KEYS = ("com", "force0", "forces", "dcom", "h", "pos")
ONLY = os.environ.get("SWEEP_ONLY", "cfg2,cfg3,cfg5").split(",")
for name, gen in (("cfg2", cm.synthetic.config2_perturbed_com), ("cfg3", cm.synthetic.config3_external_push),
                  ("cfg5", cm.synthetic.config5_footstep_candidates)):
    if name not in ONLY:
        continue
    for st in STEPTOL:
        worst = {k: 0.0 for k in KEYS}
        its, mx, bad, pol_all = [], 0, 0, []
        for seed in SEEDS:
            cfg, P, X0 = gen(B, seed=seed)
            P32, X032 = P.astype(np.float32), X0.astype(np.float32)
            Xr, infr = ol.ref_solve_batch(problem_nlp.oracle_cfg(cfg), P32.astype(np.float64), X032.astype(np.float64),
                                          ol.ipm_opts(tol=1e-9, mu_min=1e-10), nthreads=16)
            kw = {} if st is None else {"step_tolerance": st}
            if os.environ.get("SWEEP_EXTRAP"):
                kw["final_extrapolation"] = int(os.environ["SWEEP_EXTRAP"])
            for kk in ("tail_stages", "tail_iterations"):
                if os.environ.get("SWEEP_" + kk.upper()):
                    kw[kk] = int(os.environ["SWEEP_" + kk.upper()])
            for kk in ("tolerance", "mu_min", "step_tolerance", "tail_trigger"):
                if os.environ.get("SWEEP_" + kk.upper()):
                    kw[kk] = float(os.environ["SWEEP_" + kk.upper()])
            s = cm.BatchSolver(cfg, B, **kw)
            X, info, rc = s.solve_host(P32, X032)
            s.close()
            bad += int((info[:, 5] != 0).sum()) + int((infr[:, 5] != 0).sum())
            its.append(info[:, 0].mean()); mx = max(mx, int(info[:, 0].max())); pol_all += info[:, 3].tolist()
            for b in range(B):
                if infr[b, 5] == 0:
                    e = parity.errors(cfg.N, P32[b], X[b], Xr[b])
                    for k in KEYS:
                        worst[k] = max(worst[k], e[k])
        pol = int(np.sum(polished)) if (polished := [int((i >= 100000)) for i in pol_all]) else 0
        print(name, f"N={cfg.N} seeds {SEEDS} x {B} step_tol {'default' if st is None else '%.0e' % st} iters mean {np.mean(its):.2f} max {mx} polished {pol} "
              f"not-converged {bad} | max err " + " ".join(f"{k} {worst[k]:.1e}" for k in KEYS), flush=True)
