"""Pins oracle/nlp_ref.c (the float64 restatement) against the reference's generated NLP code:
(a) the committed golden vectors made from it (tests/golden/make_nlp_golden.py), always;
(b) oracle/_ref itself (tmp.c / jit_tmpComMiH.c compiled as is) on fresh seeds, when present.
Reference: /root/reference/src/centroidal-mpc-walking/config/robots/ergoCubGazeboV1/tmp.c:62-67,
12430, 58926, 71962.
"""
import os

import numpy as np
import pytest

from oracle import oracle_lib as ol

WEIGHTS = {"tmp": {}, "jit": dict(w_com=(10, 100, 200), w_sym=100)}


def _dense(shape, r, c, v):
    M = np.zeros(shape)
    np.add.at(M, (r, c), v)
    return M


@pytest.mark.parametrize("which", ["tmp", "jit"])
def test_restatement_matches_golden(which, golden_dir):
    d = np.load(os.path.join(golden_dir, f"nlp_{which}.npz"))
    cfg = ol.make_cfg(int(d["N"]), float(d["dt"]), **WEIGHTS[which])
    nx, npar, ng, nnzj, nnzh = ol.dims(cfg)
    assert (nx, npar, ng, nnzj, nnzh) == (555, 627, 651, 2931, 4140)  # tmp.c:62-67
    jcol = np.repeat(np.arange(nx), np.diff(d["jac_colind"]))
    hcol = np.repeat(np.arange(nx), np.diff(d["hess_colind"]))
    for t in range(d["x"].shape[0]):
        x, p = d["x"][t], d["p"][t]
        f, g = ol.nlp_fg(cfg, x, p)
        assert abs(f - d["f"][t]) <= 1e-12 * max(1.0, abs(d["f"][t]))
        np.testing.assert_allclose(g, d["g"][t], rtol=0, atol=1e-12)
        gf = ol.nlp_grad_f(cfg, x, p)
        np.testing.assert_allclose(gf, d["grad_f"][t], rtol=1e-12, atol=1e-12 * np.abs(d["grad_f"][t]).max())
        r, c, v = ol.nlp_jac(cfg, x, p)
        J = _dense((ng, nx), r, c, v)
        np.testing.assert_allclose(J[d["jac_row"], jcol], d["jac_nnz"][t], rtol=0, atol=1e-12)
        # structural pattern identical to the CCS table (casadi_s5, tmp.c:67)
        pat = np.zeros((ng, nx), bool); pat[r, c] = True
        ref = np.zeros((ng, nx), bool); ref[d["jac_row"], jcol] = True
        assert (pat == ref).all()
        r, c, v = ol.nlp_hess(cfg, x, p, d["lam_f"][t], d["lam_g"][t])
        H = _dense((nx, nx), r, c, v)
        scale = np.abs(d["hess_nnz"][t]).max()
        np.testing.assert_allclose(H[d["hess_row"], hcol], d["hess_nnz"][t], rtol=0, atol=1e-12 * scale)
        pat = np.zeros((nx, nx), bool); pat[r, c] = True
        ref = np.zeros((nx, nx), bool); ref[d["hess_row"], hcol] = True
        assert (pat == ref).all()  # casadi_s4, tmp.c:66
        # nlp_grad (tmp.c:24791): gradient of lam_f f + lam_g^T g w.r.t. x and w.r.t. the parameters
        gx, gp = ol.nlp_grad(cfg, x, p, d["lam_f"][t], d["lam_g"][t])
        np.testing.assert_allclose(gx, d["grad_gamma_x"][t], rtol=0, atol=1e-12 * np.abs(d["grad_gamma_x"][t]).max())
        np.testing.assert_allclose(gp, d["grad_gamma_p"][t], rtol=0, atol=1e-12 * np.abs(d["grad_gamma_p"][t]).max())
        assert ((gp != 0) == (d["grad_gamma_p"][t] != 0)).all()   # limA/limB, currentPos, com0/dcom0/h0 never enter f, g


@pytest.mark.parametrize("which", ["tmp", "jit"])
def test_restatement_matches_compiled_reference(which):
    ref_nlp = pytest.importorskip("oracle.ref_nlp")
    try:
        ref = ref_nlp.RefNLP(which)
    except FileNotFoundError:
        pytest.skip("oracle/_ref not built (reference sources absent)")
    cfg = ol.make_cfg(12, 0.1, **WEIGHTS[which])
    nx, npar, ng, _, _ = ol.dims(cfg)
    rng = np.random.default_rng(7)
    for _ in range(3):
        x, p = rng.normal(size=nx), rng.normal(size=npar)
        lf, lg = rng.normal(), rng.normal(size=ng)
        f0, gf0, g0, J0 = ref.jac_fg(x, p)
        H0 = ref.hess_l(x, p, lf, lg)
        f1, g1 = ol.nlp_fg(cfg, x, p)
        assert abs(f0 - f1) <= 1e-12 * abs(f0)
        np.testing.assert_allclose(g1, g0, atol=1e-12)
        np.testing.assert_allclose(ol.nlp_grad_f(cfg, x, p), gf0, atol=1e-12 * np.abs(gf0).max())
        np.testing.assert_allclose(_dense((ng, nx), *ol.nlp_jac(cfg, x, p)), J0, atol=1e-12)
        np.testing.assert_allclose(_dense((nx, nx), *ol.nlp_hess(cfg, x, p, lf, lg)), H0,
                                   atol=1e-12 * np.abs(H0).max())
        _, _, gx0, gp0 = ref.grad(x, p, lf, lg)
        gx1, gp1 = ol.nlp_grad(cfg, x, p, lf, lg)
        np.testing.assert_allclose(gx1, gx0, atol=1e-12 * np.abs(gx0).max())
        np.testing.assert_allclose(gp1, gp0, atol=1e-12 * np.abs(gp0).max())


def test_dims_generalise_in_N():
    # SURVEY 8a-NLP size table
    for N, exp in ((10, (465, 527, 545, 2445, 3444)), (20, (915, 1027, 1075, 4875, 6924)),
                   (30, (1365, 1527, 1605, 7305, 10404))):
        assert ol.dims(ol.make_cfg(N, 0.06)) == exp
