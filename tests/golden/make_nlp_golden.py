"""Generates tests/golden/nlp_{tmp,jit}.npz from the reference's own generated NLP code.

Run in the build container (needs oracle/_ref, i.e. /root/reference):
    python tests/golden/make_nlp_golden.py
Inputs: seeded random (x, p, lam_f, lam_g) -- random Gamma and R exercise every term -- plus one
physically meaningful standing state.  Outputs: f, g, grad f, jac g (CCS nnz), hess L (CCS nnz) and nlp_grad's
grad_gamma_x / grad_gamma_p (tmp.c:24791), evaluated by oracle/_ref/libnlp_{tmp,jit}.so (= tmp.c / jit_tmpComMiH.c compiled as is), and the
CCS sparsity tables (tmp.c:66-67).  These are data (vectors), not reference source.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_nlp  # noqa: E402

N = 12


def standing_xp(rng):
    """Physically meaningful sample: double support, feet at (0,+-0.08,0), small perturbations."""
    nx, npar = ref_nlp.NX, ref_nlp.NP
    x = np.zeros(nx)
    p = np.zeros(npar)
    o = 0
    x[0:3 * (N + 1)] = np.tile([0.0, 0.0, 0.7], N + 1) + 0.01 * rng.normal(size=3 * (N + 1))
    x[3 * (N + 1):9 * (N + 1)] = 0.02 * rng.normal(size=6 * (N + 1))
    for c, y in enumerate((0.08, -0.08)):
        base = 9 * (N + 1) + c * (18 * N + 3)
        x[base:base + 3 * (N + 1)] = np.tile([0.0, y, 0.0], N + 1)
        x[base + 3 * (N + 1):base + 3 * (N + 1) + 3 * N] = 0.0
        fo = base + 3 * (N + 1) + 3 * N
        x[fo:fo + 12 * N] = np.tile([0.0, 0.0, 9.80665 / 8], 4 * N) + 0.1 * rng.normal(size=12 * N)
        pb = c * (19 * N + 6)
        p[pb:pb + 9 * N] = np.tile(np.eye(3).reshape(-1, order="F"), N)
        p[pb + 9 * N:pb + 12 * N] = np.tile([-0.01, -0.05 if c else 0.0, 0.0], N)
        p[pb + 12 * N:pb + 15 * N] = np.tile([0.01, 0.0 if c else 0.05, 0.0], N)
        p[pb + 15 * N:pb + 16 * N] = 1.0
        p[pb + 16 * N:pb + 16 * N + 3 * (N + 1)] = np.tile([0.0, y, 0.0], N + 1)
        p[pb + 19 * N + 3:pb + 19 * N + 6] = [0.0, y, 0.0]
    o = 2 * (19 * N + 6)
    p[o:o + 9] = [0.01, -0.005, 0.69, 0.02, 0, 0, 0, 0, 0]
    p[o + 9:o + 9 + 3 * (N + 1)] = np.tile([0.0, 0.0, 0.7], N + 1)
    return x, p


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for which in ("tmp", "jit"):
        ref = ref_nlp.RefNLP(which)
        rng = np.random.default_rng(20221 if which == "tmp" else 20222)
        X, P, LF, LG, F, G, GF, JN, HN, GX, GP = [], [], [], [], [], [], [], [], [], [], []
        _, _, jc, jr = ref.sparsity("nlp_jac_fg", "out", 3)
        _, _, hc, hr = ref.sparsity("nlp_hess_l", "out", 0)
        jcol = np.repeat(np.arange(ref_nlp.NX), np.diff(jc))
        hcol = np.repeat(np.arange(ref_nlp.NX), np.diff(hc))
        # samples 0-2 random, 3 standing (as in round 1: the tests index them), 4-5 random, 6-9 stepping states from the package's
        # N = 12 generators (swing phase + push; yawed feet, R != I) around their cold starts -- physically meaningful x and p
        import cmpc_amd as cm
        _, Pw, Xw = cm.synthetic.walking_push_n12(which, B=2, seed=141)
        _, Py, Xy = cm.synthetic.yawed_steps_n12(which, B=2, seed=142)
        stepping = [(Xw[0], Pw[0]), (Xw[1], Pw[1]), (Xy[0], Py[0]), (Xy[1], Py[1])]
        for t in range(10):
            if t < 3 or t in (4, 5):
                x = rng.normal(size=ref_nlp.NX)
                p = rng.normal(size=ref_nlp.NP)
            elif t == 3:
                x, p = standing_xp(rng)
            else:
                x0, p = stepping[t - 6]
                x = x0 + 0.02 * rng.normal(size=ref_nlp.NX)
                p = np.array(p, dtype=np.float64)
            lf = float(rng.normal()) if (t < 3 or t in (4, 5)) else 1.0
            lg = rng.normal(size=ref_nlp.NG)
            f, gf, g, J = ref.jac_fg(x, p)
            H = ref.hess_l(x, p, lf, lg)
            f2, g2, gx, gp = ref.grad(x, p, lf, lg)
            assert f2 == f and (g2 == g).all()
            GX.append(gx); GP.append(gp)
            X.append(x); P.append(p); LF.append(lf); LG.append(lg)
            F.append(f); G.append(g); GF.append(gf)
            JN.append(J[jr, jcol]); HN.append(H[hr, hcol])
        np.savez_compressed(
            os.path.join(out_dir, f"nlp_{which}.npz"),
            N=N, dt=0.1, x=np.array(X), p=np.array(P), lam_f=np.array(LF), lam_g=np.array(LG),
            f=np.array(F), g=np.array(G), grad_f=np.array(GF), jac_nnz=np.array(JN),
            hess_nnz=np.array(HN), grad_gamma_x=np.array(GX), grad_gamma_p=np.array(GP), jac_colind=jc, jac_row=jr, hess_colind=hc, hess_row=hr)
        print(which, "written")


if __name__ == "__main__":
    main()
