"""GPU parity of the batched NLP callbacks (f, g, grad f, jac g, hess L) against the golden
vectors made from the reference's own generated code (tmp.c / jit_tmpComMiH.c, N=12) and against
oracle/nlp_ref.c at N=20.  float32 kernel vs float64 reference: 1e-5 relative."""
import os

import numpy as np
import pytest

import cmpc_amd as cm

pytestmark = pytest.mark.gpu


def _eval(cfg, X, P, lam_f, LamG):
    import ctypes as C

    import torch
    B = X.shape[0]
    L = cm.Layout(cfg.N)
    s = cm.BatchSolver(cfg, B)
    lib = cm._capi.lib()
    nnzj, nnzh = 243 * cfg.N + 15, 348 * cfg.N - 36
    dX, dP, dL = (torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda() for a in (X, P, LamG))
    F = torch.empty(B, device="cuda"); G = torch.empty(B, L.ng, device="cuda"); GF = torch.empty(B, L.nx, device="cuda")
    J = torch.empty(B, nnzj, device="cuda"); H = torch.empty(B, nnzh, device="cuda")
    rc = lib.cmpc_eval_nlp_device(s._h, dX.data_ptr(), dP.data_ptr(), dL.data_ptr(), C.c_float(lam_f), F.data_ptr(), G.data_ptr(),
                                  GF.data_ptr(), J.data_ptr(), H.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, s.last_error
    torch.cuda.synchronize()
    return [t.cpu().numpy().astype(np.float64) for t in (F, G, GF, J, H)]


def _close(a, b, rel=1e-5):
    scale = max(np.abs(b).max(), 1e-30)
    assert np.abs(a - b).max() <= rel * scale, (np.abs(a - b).max(), scale)


@pytest.mark.parametrize("which", ["tmp", "jit"])
def test_callbacks_match_reference_generated_code(which, golden_dir):
    d = np.load(os.path.join(golden_dir, f"nlp_{which}.npz"))
    cfg = cm.config.generated_code_weights(which)
    n = d["x"].shape[0]
    lam_f = float(d["lam_f"][3])  # one lam_f per launch: use the physical sample's (1.0) for all
    F, G, GF, J, H = _eval(cfg, d["x"], d["p"], 1.0, d["lam_g"])
    jr = np.empty(2931, np.int32); jc = np.empty(2931, np.int32); hr = np.empty(4140, np.int32); hc = np.empty(4140, np.int32)
    assert cm._capi.lib().cmpc_nlp_sparsity(12, jr.ctypes.data, jc.ctypes.data, hr.ctypes.data, hc.ctypes.data) == 0
    # CCS structure identical to the reference's tables (casadi_s5 / casadi_s4, tmp.c:66-67)
    np.testing.assert_array_equal(jr, d["jac_row"]); np.testing.assert_array_equal(hr, d["hess_row"])
    np.testing.assert_array_equal(np.bincount(jc, minlength=555).cumsum(), d["jac_colind"][1:])
    np.testing.assert_array_equal(np.bincount(hc, minlength=555).cumsum(), d["hess_colind"][1:])
    assert lam_f == 1.0 and n >= 10
    for t in range(n):
        _close(F[t], d["f"][t]); _close(G[t], d["g"][t]); _close(GF[t], d["grad_f"][t]); _close(J[t], d["jac_nnz"][t])
    # hess goldens were taken with per-sample lam_f and one launch has one lam_f: compare the samples taken at lam_f = 1 (the standing
    # state and the four stepping states: swing phase + push, yawed feet)
    ones = [t for t in range(n) if d["lam_f"][t] == 1.0]
    assert len(ones) >= 5
    for t in ones:
        _close(H[t], d["hess_nnz"][t])


def test_callbacks_match_oracle_at_n20():
    from oracle import oracle_lib as ol, problem_nlp
    cfg, P, X0 = cm.synthetic.config3_external_push(4)
    rng = np.random.default_rng(5)
    X = (X0 + 0.05 * rng.normal(size=X0.shape)).astype(np.float32)
    P32 = P.astype(np.float32)
    L = cm.Layout(cfg.N)
    LamG = rng.normal(size=(4, L.ng)).astype(np.float32)
    F, G, GF, J, H = _eval(cfg, X, P32, 0.7, LamG)
    oc = problem_nlp.oracle_cfg(cfg)
    nnzj, nnzh = 243 * cfg.N + 15, 348 * cfg.N - 36
    jr = np.empty(nnzj, np.int32); jc = np.empty(nnzj, np.int32); hr = np.empty(nnzh, np.int32); hc = np.empty(nnzh, np.int32)
    cm._capi.lib().cmpc_nlp_sparsity(cfg.N, jr.ctypes.data, jc.ctypes.data, hr.ctypes.data, hc.ctypes.data)
    for b in range(4):
        x, p = X[b].astype(np.float64), P32[b].astype(np.float64)
        f, g = ol.nlp_fg(oc, x, p)
        _close(F[b], f); _close(G[b], g); _close(GF[b], ol.nlp_grad_f(oc, x, p))
        r, c, v = ol.nlp_jac(oc, x, p)
        Jd = np.zeros((L.ng, L.nx)); np.add.at(Jd, (r, c), v)
        _close(J[b], Jd[jr, jc])
        r, c, v = ol.nlp_hess(oc, x, p, 0.7, LamG[b].astype(np.float64))
        Hd = np.zeros((L.nx, L.nx)); np.add.at(Hd, (r, c), v)
        _close(H[b], Hd[hr, hc])


def _grad(cfg, X, P, lam_f, LamG):
    import ctypes as C

    import torch
    B = X.shape[0]
    L = cm.Layout(cfg.N)
    s = cm.BatchSolver(cfg, B)
    dX, dP, dL = (torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda() for a in (X, P, LamG))
    GX = torch.empty(B, L.nx, device="cuda"); GP = torch.empty(B, L.np, device="cuda")
    rc = cm._capi.lib().cmpc_eval_nlp_grad_device(s._h, dX.data_ptr(), dP.data_ptr(), dL.data_ptr(), C.c_float(lam_f), GX.data_ptr(),
                                                  GP.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, s.last_error
    torch.cuda.synchronize()
    return GX.cpu().numpy().astype(np.float64), GP.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("which", ["tmp", "jit"])
def test_nlp_grad_matches_reference_generated_code(which, golden_dir):
    """nlp_grad (tmp.c:24791): grad_gamma_x and grad_gamma_p against the vectors made from the reference's compiled code."""
    d = np.load(os.path.join(golden_dir, f"nlp_{which}.npz"))
    cfg = cm.config.generated_code_weights(which)
    for t in range(d["x"].shape[0]):   # lam_f differs per sample: one launch each
        GX, GP = _grad(cfg, d["x"][t:t + 1], d["p"][t:t + 1], float(d["lam_f"][t]), d["lam_g"][t:t + 1])
        _close(GX[0], d["grad_gamma_x"][t]); _close(GP[0], d["grad_gamma_p"][t])
        assert (GP[0][d["grad_gamma_p"][t] == 0] == 0).all()


def test_nlp_grad_matches_oracle_at_n20_and_n30():
    from oracle import oracle_lib as ol, problem_nlp
    for gen in (cm.synthetic.config3_external_push, cm.synthetic.config5_footstep_candidates):
        cfg, P, X0 = gen(4)
        rng = np.random.default_rng(6)
        X = (X0 + 0.05 * rng.normal(size=X0.shape)).astype(np.float32)
        P32 = P.astype(np.float32)
        LamG = rng.normal(size=(4, cm.Layout(cfg.N).ng)).astype(np.float32)
        GX, GP = _grad(cfg, X, P32, -0.6, LamG)
        oc = problem_nlp.oracle_cfg(cfg)
        for b in range(4):
            gx, gp = ol.nlp_grad(oc, X[b].astype(np.float64), P32[b].astype(np.float64), -0.6, LamG[b].astype(np.float64))
            _close(GX[b], gx); _close(GP[b], gp)
