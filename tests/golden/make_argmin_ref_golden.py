"""Generates tests/golden/argmin_ref_{walk,yaw}_{tmp,jit}.npz: argmin vectors of N = 12 problems computed ON THE
REFERENCE'S OWN COMPILED NLP FUNCTIONS (oracle/_ref = tmp.c / jit_tmpComMiH.c compiled where they lie): the
IPOPT-style solver oracle/ipm_generic.py is run with f, g, grad f, jac g and hess L taken from that code, i.e. it
minimises exactly what IPOPT minimises in the reference (N = 12, dt = 0.1 and the weights are baked into that code).

    python tests/golden/make_argmin_ref_golden.py          (build container only: needs /root/reference)

Problems, 16 of each with both baked weight sets (160 in all): a swing phase with a push (step adjustment active), two yawed
footsteps (R != I), push recovery with active friction rows, single support at the horizon end, the standing problem.  Stored: float32 inputs (P, X0), the float64 argmin, the multipliers of every constraint row, the
objective, and the KKT residuals of the stored point evaluated with the reference's code (stationarity, feasibility,
complementarity, smallest eigenvalue of the Hessian reduced to the null space of the active constraints).
These are data (vectors), not reference source."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cmpc_amd as cm  # noqa: E402
from oracle import ipm_generic, problem_nlp, ref_nlp  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def kkt_report(ref, x, lam, p, lb, ub):
    """KKT residuals of (x, lam) on the reference's compiled functions."""
    f, gf, g, J = ref.jac_fg(x, p)
    scale = max(1.0, np.abs(lam).max())
    stat = np.abs(gf + J.T @ lam).max() / scale
    feas = max(np.maximum(lb - g, 0).max(), np.maximum(g - ub, 0).max())
    ineq = ub - lb > 1e-12
    dist = np.minimum(g - lb, ub - g)
    compl = np.abs(lam[ineq] * dist[ineq]).max() / scale
    # multiplier signs (lam^T g enters the Lagrangian with +): a row bounded above only carries lam >= 0, below only lam <= 0, and a
    # two-sided row a multiplier of the sign of its NEARER bound (a weakly active row of an interior-point solution sits t = mu / lam
    # off its bound: "at the bound to 1e-7" is the wrong test for it, the complementarity product above is the right one)
    tol = 1e-7 * scale
    up_only, lo_only = ineq & (lb < -1e19), ineq & (ub > 1e19)
    two = ineq & ~up_only & ~lo_only
    nearer_ub = (ub - g) < (g - lb)
    sign_ok = bool((lam[up_only] >= -tol).all() and (lam[lo_only] <= tol).all()
                   and ((np.abs(lam[two]) <= tol) | ((lam[two] > 0) == nearer_ub[two])).all())
    # second order: Hessian of the Lagrangian on the null space of the active constraint gradients
    H = ref.hess_l(x, p, 1.0, lam)
    active = (~ineq) | (np.abs(lam) > 1e-7 * scale)
    A = J[active]
    _, sv, Vt = np.linalg.svd(A, full_matrices=True)
    rank = int((sv > 1e-9 * sv[0]).sum())
    Z = Vt[rank:].T
    red = Z.T @ H @ Z
    min_eig = float(np.linalg.eigvalsh(0.5 * (red + red.T)).min()) if Z.shape[1] else 0.0
    return dict(f=f, stationarity=stat, feasibility=feas, complementarity=compl, sign_ok=sign_ok, min_reduced_eig=min_eig,
                n_active=int(active.sum()), null_dim=int(Z.shape[1]))


def make(name, which, gen):
    cfg, P, X0 = gen(which)
    ref = ref_nlp.RefNLP(which)
    fun = ipm_generic.ReferenceFunctions(which)
    oc = problem_nlp.oracle_cfg(cfg)
    P32, X032 = P.astype(np.float32), X0.astype(np.float32)
    xs, lams, fs, reps = [], [], [], []
    for b in range(P32.shape[0]):
        p = P32[b].astype(np.float64)
        lb, ub = problem_nlp.bounds(cfg, p)
        r = ipm_generic.solve(oc, p, lb, ub, X032[b].astype(np.float64), tol=1e-9, max_iter=400, fun=fun)
        assert r["status"] == 0, (name, which, b, r["kkt"])
        rep = kkt_report(ref, r["x"], r["lam_g"], p, lb, ub)
        print(name, which, b, "iters", r["iters"], rep)
        assert rep["stationarity"] < 1e-7 and rep["feasibility"] < 1e-8 and rep["complementarity"] < 1e-6 and rep["sign_ok"]
        assert rep["min_reduced_eig"] > -1e-8
        xs.append(r["x"]); lams.append(r["lam_g"]); fs.append(rep["f"])
        reps.append([rep["stationarity"], rep["feasibility"], rep["complementarity"], rep["min_reduced_eig"], rep["n_active"], rep["null_dim"]])
    np.savez_compressed(os.path.join(OUT, f"argmin_ref_{name}_{which}.npz"), P=P32, X0=X032, x_star=np.array(xs), lam_g=np.array(lams),
                        f_star=np.array(fs), kkt=np.array(reps), N=cfg.N, dt=cfg.sampling_time)


if __name__ == "__main__":
    for which in ("tmp", "jit"):
        make("walk", which, cm.synthetic.walking_push_n12)
        make("yaw", which, cm.synthetic.yawed_steps_n12)
        make("push", which, cm.synthetic.push_recovery_n12)
        make("ssend", which, cm.synthetic.single_support_end_n12)
        make("stand", which, cm.synthetic.standing_n12)
