"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/libcmpc_oracle.so (nlp_ref.c + ipm_ref.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libcmpc_oracle.so")


class NlpCfg(C.Structure):
    _fields_ = [
        ("N", C.c_int),
        ("dt", C.c_double),
        ("mu", C.c_double),
        ("gravity", C.c_double),
        ("w_com", C.c_double * 3),
        ("w_h", C.c_double),
        ("w_pos", C.c_double),
        ("w_rate", C.c_double * 3),
        ("w_sym", C.c_double),
        ("corners", C.c_double * 24),
    ]


class IpmOpts(C.Structure):
    _fields_ = [
        ("max_iter", C.c_int),
        ("tol", C.c_double),
        ("mu_init", C.c_double),
        ("mu_min", C.c_double),
        ("exact_hessian", C.c_int),
        ("verbose", C.c_int),
        ("tail_stages", C.c_int),
        ("tail_iters", C.c_int),
        ("tail_trigger", C.c_double),
    ]


def build(force=False):
    srcs = [os.path.join(_HERE, s) for s in ("nlp_ref.c", "ipm_ref.c", "ipm_ref_f32.c", "ipm_ref_mix.c", "cmpc_oracle.h")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "libcmpc_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
    return _lib


def make_cfg(N, dt, mu=0.33, w_com=(10, 10, 200), w_h=100, w_pos=200, w_rate=(10, 10, 10),
             w_sym=10, corners=None, gravity=9.80665):
    """Defaults = the weight set baked into the reference's tmp.c."""
    cfg = NlpCfg()
    cfg.N, cfg.dt, cfg.mu, cfg.gravity = N, dt, mu, gravity
    cfg.w_com[:] = w_com
    cfg.w_h, cfg.w_pos, cfg.w_sym = w_h, w_pos, w_sym
    cfg.w_rate[:] = w_rate
    if corners is None:
        one = [(0.08, 0.01, 0.0), (0.08, -0.01, 0.0), (-0.08, -0.01, 0.0), (-0.08, 0.01, 0.0)]
        corners = [one, one]
    cfg.corners[:] = np.asarray(corners, np.float64).reshape(-1)
    return cfg


def dims(cfg):
    v = [C.c_int() for _ in range(5)]
    lib().cmpc_nlp_dims(C.byref(cfg), *[C.byref(a) for a in v])
    return tuple(a.value for a in v)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def nlp_fg(cfg, x, p):
    nx, npar, ng, _, _ = dims(cfg)
    x = np.ascontiguousarray(x, np.float64)
    p = np.ascontiguousarray(p, np.float64)
    assert x.size == nx and p.size == npar
    f = C.c_double()
    g = np.zeros(ng)
    lib().cmpc_nlp_fg(C.byref(cfg), _dp(x), _dp(p), C.byref(f), _dp(g))
    return f.value, g


def nlp_grad_f(cfg, x, p):
    nx = dims(cfg)[0]
    x = np.ascontiguousarray(x, np.float64)
    p = np.ascontiguousarray(p, np.float64)
    gf = np.zeros(nx)
    lib().cmpc_nlp_grad_f(C.byref(cfg), _dp(x), _dp(p), _dp(gf))
    return gf


def nlp_jac(cfg, x, p):
    """-> (row, col, val) COO."""
    nnzj = dims(cfg)[3]
    x = np.ascontiguousarray(x, np.float64)
    p = np.ascontiguousarray(p, np.float64)
    row = np.zeros(nnzj, np.int32)
    col = np.zeros(nnzj, np.int32)
    val = np.zeros(nnzj)
    n = lib().cmpc_nlp_jac(C.byref(cfg), _dp(x), _dp(p), _ip(row), _ip(col), _dp(val))
    assert n == nnzj, (n, nnzj)
    return row, col, val


def nlp_hess(cfg, x, p, lam_f, lam_g):
    nnzh = dims(cfg)[4]
    x = np.ascontiguousarray(x, np.float64)
    p = np.ascontiguousarray(p, np.float64)
    lam_g = np.ascontiguousarray(lam_g, np.float64)
    row = np.zeros(nnzh, np.int32)
    col = np.zeros(nnzh, np.int32)
    val = np.zeros(nnzh)
    lib().cmpc_nlp_hess.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
    n = lib().cmpc_nlp_hess(C.byref(cfg), x.ctypes.data, p.ctypes.data, float(lam_f),
                            lam_g.ctypes.data, row.ctypes.data, col.ctypes.data, val.ctypes.data)
    assert n == nnzh, (n, nnzh)
    return row, col, val


def nlp_grad(cfg, x, p, lam_f, lam_g):
    """nlp_grad (tmp.c:24791): -> (grad_gamma_x[nx], grad_gamma_p[np]) of gamma = lam_f f + lam_g^T g."""
    nx, npar = dims(cfg)[:2]
    x = np.ascontiguousarray(x, np.float64)
    p = np.ascontiguousarray(p, np.float64)
    lam_g = np.ascontiguousarray(lam_g, np.float64)
    gx, gp = np.zeros(nx), np.zeros(npar)
    fn = lib().cmpc_nlp_grad
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    fn(C.addressof(cfg), x.ctypes.data, p.ctypes.data, float(lam_f), lam_g.ctypes.data, gx.ctypes.data, gp.ctypes.data)
    return gx, gp


def ipm_opts(max_iter=60, tol=1e-9, mu_init=0.1, mu_min=1e-10, exact_hessian=1, verbose=0, tail_stages=0, tail_iters=2, tail_trigger=2e-5):
    o = IpmOpts()
    o.max_iter, o.tol, o.mu_init, o.mu_min, o.exact_hessian, o.verbose = max_iter, tol, mu_init, mu_min, exact_hessian, verbose
    o.tail_stages, o.tail_iters, o.tail_trigger = tail_stages, tail_iters, tail_trigger
    return o


def ref_solve_batch(cfg, P, X0, opts=None, f32=False, nthreads=1, mix=False):
    """Structured reference solver (ipm_ref.c).  P[B,np], X0[B,nx] -> X[B,nx], info[B,6]
    (iterations, kkt error, final mu, #GN fallbacks, primal infeasibility, status)."""
    opts = opts or ipm_opts()
    dt = np.float32 if (f32 or mix) else np.float64
    P = np.ascontiguousarray(P, dt)
    X0 = np.ascontiguousarray(X0, dt)
    B = P.shape[0]
    nx, npar = dims(cfg)[:2]
    assert P.shape == (B, npar) and X0.shape == (B, nx)
    X = np.zeros((B, nx), dt)
    info = np.zeros((B, 6))
    fn = lib().cmpc_ref_solve_batch_mix if mix else (lib().cmpc_ref_solve_batch_f32 if f32 else lib().cmpc_ref_solve_batch)
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    fn(C.addressof(cfg), C.addressof(opts), B, P.ctypes.data, X0.ctypes.data, X.ctypes.data,
       info.ctypes.data, nthreads)
    return X, info
