"""Generates tests/golden/argmin_*.npz: converged float64 solutions ("argmin golden") of sample
problems of the BASELINE.json configurations, from oracle/ipm_generic.py (IPOPT-style solver on the
reference NLP as the reference states it; functions pinned to the reference's generated code).

    python tests/golden/make_argmin_golden.py

Inputs are stored in float32 (what the GPU solver is fed); the golden solve uses exactly those
rounded inputs promoted to float64, so input rounding is not counted as solver error.
The reference holds no solver outputs (no tests), so the argmin is pinned at the KKT
level: tests/test_oracle_ipm.py re-checks stationarity/feasibility of every stored solution with
oracle/nlp_ref.c and, for the N=12 case, with the reference's own compiled code (oracle/_ref).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cmpc_amd as cm  # noqa: E402
from oracle import ipm_generic, problem_nlp  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def make(name, gen, nb):
    cfg, P, X0 = gen
    P32 = P[:nb].astype(np.float32)
    X032 = X0[:nb].astype(np.float32)
    oc = problem_nlp.oracle_cfg(cfg)
    xs, lams, fs, its = [], [], [], []
    for b in range(nb):
        p = P32[b].astype(np.float64)
        lb, ub = problem_nlp.bounds(cfg, p)
        r = ipm_generic.solve(oc, p, lb, ub, X032[b].astype(np.float64), tol=1e-9, max_iter=400)
        assert r["status"] == 0, (name, b, r["kkt"])
        xs.append(r["x"]); lams.append(r["lam_g"]); fs.append(r["f"]); its.append(r["iters"])
        print(name, b, "iters", r["iters"], "f*", r["f"])
    np.savez_compressed(os.path.join(OUT, f"argmin_{name}.npz"), P=P32, X0=X032,
                        x_star=np.array(xs), lam_g=np.array(lams), f_star=np.array(fs),
                        iters=np.array(its), N=cfg.N, dt=cfg.sampling_time)


if __name__ == "__main__":
    make("known_answer_n12", cm.synthetic.standing_known_answer(), 1)
    make("cfg1", cm.synthetic.config1_plumbing(), 1)
    make("cfg2", cm.synthetic.config2_perturbed_com(8), 8)
    make("cfg3", cm.synthetic.config3_external_push(8), 8)
    make("cfg5", cm.synthetic.config5_footstep_candidates(4), 4)
