"""Developer probe (GPU box, -DCMPC_PROFILE build: it reads CMPC_SIGMA_MIN): iteration histograms of config-2 batches against the lower bound of the centring parameter."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import cmpc_amd as cm
    cm._capi.LIB_PATH = os.path.join(os.path.dirname(cm._capi.LIB_PATH), "libcmpc_hip_proflight.so")
    its, mx, bad = [], [], 0
    which = os.environ.get("SWEEP_CFG", "config2")
    gen, B, ns = {"config2": (cm.synthetic.config2_perturbed_com, 256, 8), "config3": (cm.synthetic.config3_external_push, 4096, 2),
                  "config5": (cm.synthetic.config5_footstep_candidates, 4096, 2)}[which]
    for seed in range(ns):
        cfg, P, X0 = gen(B, seed=seed + 10)
        s = cm.BatchSolver(cfg, B)
        X, info, rc = s.solve_host(P.astype(np.float32), X0.astype(np.float32))
        it = info[:, 0].astype(int)
        its.append(it); mx.append(int(it.max())); bad += int((info[:, 5] != 0).sum())
        s.close()
    its = np.concatenate(its)
    print(os.environ.get("SWEEP_CFG", "config2"), "sigma_min", os.environ.get("CMPC_SIGMA_MIN"), "mean %.2f" % its.mean(), "batch maxima", mx, "not converged", bad, "histogram", np.bincount(its)[3:].tolist(), flush=True)
else:
    for v in ("0.03", "0.02", "0.01", "0.005", "0.06"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, CMPC_SIGMA_MIN=v), check=True)
