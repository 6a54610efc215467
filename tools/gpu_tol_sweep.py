"""Developer probe (GPU box): iterations and accuracy against the float64 oracle for several
termination settings."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cmpc_amd as cm
from tests import parity
from oracle import oracle_lib as ol, problem_nlp
B = int(os.environ.get('SWEEP_B', '128'))
SETS = ((1e-6, 1e-7, 1e-5), (1e-6, 1e-7, 1e-4), (3e-6, 3e-7, 1e-4), (1e-5, 1e-6, 1e-4), (1e-5, 1e-6, 1e-3), (1e-6, 1e-7, 1e-3))
if 'SWEEP_SETS' in os.environ:
    SETS = tuple(tuple(float(x) for x in v.split(':')) for v in os.environ['SWEEP_SETS'].split(','))
elif 'SWEEP_STEPTOL' in os.environ:
    SETS = tuple((1e-6, 1e-7, float(v)) for v in os.environ['SWEEP_STEPTOL'].split(','))
CONFIGS = [("cfg2", cm.synthetic.config2_perturbed_com), ("cfg3", cm.synthetic.config3_external_push)]
if os.environ.get("SWEEP_CFG5") == "only":
    CONFIGS = []
if os.environ.get("SWEEP_CFG5"):
    CONFIGS.append(("cfg5", cm.synthetic.config5_footstep_candidates))
for name, gen in CONFIGS:
    cfg, P, X0 = gen(B)
    P32, X032 = P.astype(np.float32), X0.astype(np.float32)
    oc = problem_nlp.oracle_cfg(cfg)
    Xr, infr = ol.ref_solve_batch(oc, P32.astype(np.float64), X032.astype(np.float64), ol.ipm_opts(tol=1e-9, mu_min=1e-10), nthreads=16)
    assert (infr[:, 5] == 0).all()
    for tol, mumin, steptol in SETS:
        s = cm.BatchSolver(cfg, B, tolerance=tol, mu_min=mumin, step_tolerance=steptol, final_extrapolation=bool(int(os.environ.get('SWEEP_EXTRAP', '0'))))
        X, info, rc = s.solve_host(P32, X032)
        e = [parity.errors(cfg.N, P32[b], X[b], Xr[b]) for b in range(B)]
        print(name, "tol %.0e mu_min %.0e step_tol %.0e" % (tol, mumin, steptol), "iters mean %.1f max %d bad %d" % (info[:, 0].mean(), info[:, 0].max(), (info[:, 5] != 0).sum()),
              "| max err com %.1e force0 %.1e forces %.1e dcom %.1e h %.1e pos %.1e" % tuple(max(x[q] for x in e) for q in ("com", "force0", "forces", "dcom", "h", "pos")), flush=True)
        s.close()
