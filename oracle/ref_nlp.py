"""TEST INFRASTRUCTURE ONLY -- ctypes loader for oracle/_ref/libnlp_{tmp,jit}.so.

Those two libraries are the reference's own CasADi-generated NLP code
(/root/reference/src/centroidal-mpc-walking/config/robots/ergoCubGazeboV1/tmp.c and
.../jit_tmpComMiH.c) compiled by oracle/Makefile from the sources where they lie.  They are the
exact float64 oracle for f, g, grad f, jac g, hess L at N=12, dt=0.1 (tmp.c:69, 12430, 24791,
58926, 71962).  Nothing under the product package may import this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NX, NP, NG = 555, 627, 651  # tmp.c:62-65


class RefNLP:
    def __init__(self, which="tmp"):
        path = os.path.join(_HERE, "_ref", f"libnlp_{which}.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self._iw = (C.c_longlong * 8)()
        self._w = (C.c_double * 8)()
        for fn in ("nlp", "nlp_fg", "nlp_grad", "nlp_hess_l", "nlp_jac_fg"):
            getattr(self.lib, fn).restype = C.c_int
            for io in ("in", "out"):
                f = getattr(self.lib, f"{fn}_sparsity_{io}")
                f.restype = C.POINTER(C.c_longlong)
                f.argtypes = [C.c_longlong]

    def _call(self, name, ins, outs):
        arg = (C.POINTER(C.c_double) * len(ins))()
        res = (C.POINTER(C.c_double) * len(outs))()
        for i, a in enumerate(ins):
            arg[i] = a.ctypes.data_as(C.POINTER(C.c_double))
        for i, a in enumerate(outs):
            res[i] = a.ctypes.data_as(C.POINTER(C.c_double))
        rc = getattr(self.lib, name)(arg, res, self._iw, self._w, 0)
        assert rc == 0

    def sparsity(self, fn, io, idx):
        """CCS sparsity -> (nrow, ncol, colind, row)."""
        sp = getattr(self.lib, f"{fn}_sparsity_{io}")(idx)
        nrow, ncol = sp[0], sp[1]
        colind = np.array([sp[2 + i] for i in range(ncol + 1)], dtype=np.int64)
        nnz = int(colind[-1])
        if nnz == nrow * ncol and False:
            return nrow, ncol, colind, None
        row = np.array([sp[2 + ncol + 1 + i] for i in range(nnz)], dtype=np.int64)
        return int(nrow), int(ncol), colind, row

    def fg(self, x, p):
        x = np.ascontiguousarray(x, np.float64)
        p = np.ascontiguousarray(p, np.float64)
        f = np.zeros(1)
        g = np.zeros(NG)
        self._call("nlp_fg", [x, p], [f, g])
        return f[0], g

    def jac_fg(self, x, p):
        x = np.ascontiguousarray(x, np.float64)
        p = np.ascontiguousarray(p, np.float64)
        f = np.zeros(1)
        gf = np.zeros(NX)
        g = np.zeros(NG)
        _, _, colind, row = self.sparsity("nlp_jac_fg", "out", 3)
        jn = np.zeros(len(row))
        self._call("nlp_jac_fg", [x, p], [f, gf, g, jn])
        J = np.zeros((NG, NX))
        col = np.repeat(np.arange(NX), np.diff(colind))
        J[row, col] = jn
        return f[0], gf, g, J

    def hess_l(self, x, p, lam_f, lam_g):
        x = np.ascontiguousarray(x, np.float64)
        p = np.ascontiguousarray(p, np.float64)
        lf = np.array([lam_f], np.float64)
        lg = np.ascontiguousarray(lam_g, np.float64)
        _, _, colind, row = self.sparsity("nlp_hess_l", "out", 0)
        hn = np.zeros(len(row))
        self._call("nlp_hess_l", [x, p, lf, lg], [hn])
        H = np.zeros((NX, NX))
        col = np.repeat(np.arange(NX), np.diff(colind))
        H[row, col] = hn
        return H

    def grad(self, x, p, lam_f, lam_g):
        x = np.ascontiguousarray(x, np.float64)
        p = np.ascontiguousarray(p, np.float64)
        lf = np.array([lam_f], np.float64)
        lg = np.ascontiguousarray(lam_g, np.float64)
        f = np.zeros(1)
        g = np.zeros(NG)
        gx = np.zeros(NX)
        gp = np.zeros(NP)
        self._call("nlp_grad", [x, p, lf, lg], [f, g, gx, gp])
        return f[0], g, gx, gp
