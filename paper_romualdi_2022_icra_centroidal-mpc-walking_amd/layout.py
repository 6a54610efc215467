"""Index layout of the decision vector x and the parameter vector p.

This is the layout of the reference's CasADi-generated NLP
(src/centroidal-mpc-walking/config/robots/ergoCubGazeboV1/tmp.c:62-67, decoded in SURVEY 8a-NLP):
every matrix is column-major with the time knot as column, contacts in std::map (alphabetical)
order.  The batched solver takes P[B, n_p] and X[B, n_x] in exactly this layout, so a buffer that
CasADi's Opti would hand to IPOPT can be handed to the GPU solver unchanged.
"""
from __future__ import annotations

import numpy as np


class Layout:
    def __init__(self, N: int):
        self.N = N
        o = 0
        self.com = o; o += 3 * (N + 1)
        self.dcom = o; o += 3 * (N + 1)
        self.h = o; o += 3 * (N + 1)
        self.pos, self.vel, self.f = [], [], []
        for _ in range(2):
            self.pos.append(o); o += 3 * (N + 1)
            self.vel.append(o); o += 3 * N
            fc = []
            for _ in range(4):
                fc.append(o); o += 3 * N
            self.f.append(fc)
        self.nx = o
        o = 0
        self.p_R, self.p_up, self.p_lo, self.p_gam, self.p_nom, self.p_cur = [], [], [], [], [], []
        for _ in range(2):
            self.p_R.append(o); o += 9 * N
            self.p_up.append(o); o += 3 * N   # "limA"; upper/lower order is not observable in f,g
            self.p_lo.append(o); o += 3 * N   # "limB"
            self.p_gam.append(o); o += N
            self.p_nom.append(o); o += 3 * (N + 1)
            self.p_cur.append(o); o += 3
        self.p_com0 = o; o += 3
        self.p_dcom0 = o; o += 3
        self.p_h0 = o; o += 3
        self.p_comref = o; o += 3 * (N + 1)
        self.p_href = o; o += 3 * (N + 1)
        self.p_fext = o; o += 3 * N
        self.p_text = o; o += 3 * N
        self.np = o
        self.ng = 53 * N + 15

    # ---- views into x (batch leading dims allowed) ----
    def x_com(self, x):
        return x[..., self.com:self.com + 3 * (self.N + 1)].reshape(*x.shape[:-1], self.N + 1, 3)

    def x_dcom(self, x):
        return x[..., self.dcom:self.dcom + 3 * (self.N + 1)].reshape(*x.shape[:-1], self.N + 1, 3)

    def x_h(self, x):
        return x[..., self.h:self.h + 3 * (self.N + 1)].reshape(*x.shape[:-1], self.N + 1, 3)

    def x_pos(self, x, c):
        return x[..., self.pos[c]:self.pos[c] + 3 * (self.N + 1)].reshape(*x.shape[:-1], self.N + 1, 3)

    def x_force(self, x, c, j):
        o = self.f[c][j]
        return x[..., o:o + 3 * self.N].reshape(*x.shape[:-1], self.N, 3)

    def first_forces(self, x):
        """First-knot corner forces [..., 2, 4, 3] (what getOutput() exposes)."""
        out = np.empty(x.shape[:-1] + (2, 4, 3), x.dtype)
        for c in range(2):
            for j in range(4):
                out[..., c, j, :] = x[..., self.f[c][j]:self.f[c][j] + 3]
        return out


def pack_parameters(N, R, upper, lower, enabled, nominal, current, com0, dcom0, h0, com_ref, h_ref,
                    f_ext=None, tau_ext=None, dtype=np.float64):
    """Builds P[B, n_p].  Shapes (B = batch, broadcast over a missing batch dim is not done):
      R[B,2,N,3,3] (row-major rotation matrices), upper/lower[B,2,N,3], enabled[B,2,N],
      nominal[B,2,N+1,3], current[B,2,3], com0/dcom0/h0[B,3], com_ref/h_ref[B,N+1,3],
      f_ext/tau_ext[B,N,3] or None (zeros)."""
    L = Layout(N)
    B = com0.shape[0]
    P = np.zeros((B, L.np), dtype)
    for c in range(2):
        # reference stores vec(R) column-major per knot: index 9k + 3*col + row
        P[:, L.p_R[c]:L.p_R[c] + 9 * N] = np.transpose(R[:, c], (0, 1, 3, 2)).reshape(B, 9 * N)
        P[:, L.p_up[c]:L.p_up[c] + 3 * N] = upper[:, c].reshape(B, 3 * N)
        P[:, L.p_lo[c]:L.p_lo[c] + 3 * N] = lower[:, c].reshape(B, 3 * N)
        P[:, L.p_gam[c]:L.p_gam[c] + N] = enabled[:, c]
        P[:, L.p_nom[c]:L.p_nom[c] + 3 * (N + 1)] = nominal[:, c].reshape(B, 3 * (N + 1))
        P[:, L.p_cur[c]:L.p_cur[c] + 3] = current[:, c]
    P[:, L.p_com0:L.p_com0 + 3] = com0
    P[:, L.p_dcom0:L.p_dcom0 + 3] = dcom0
    P[:, L.p_h0:L.p_h0 + 3] = h0
    P[:, L.p_comref:L.p_comref + 3 * (N + 1)] = com_ref.reshape(B, -1)
    P[:, L.p_href:L.p_href + 3 * (N + 1)] = h_ref.reshape(B, -1)
    if f_ext is not None:
        P[:, L.p_fext:L.p_fext + 3 * N] = f_ext.reshape(B, -1)
    if tau_ext is not None:
        P[:, L.p_text:L.p_text + 3 * N] = tau_ext.reshape(B, -1)
    return P


def cold_start(N, P, gravity=9.80665, dtype=np.float64):
    """x0 of SURVEY 8d config 2: CoM constant at com0, feet at nominal, f_z = g/8 per corner,
    everything else zero."""
    L = Layout(N)
    B = P.shape[0]
    X = np.zeros((B, L.nx), dtype)
    L.x_com(X)[:] = P[:, None, L.p_com0:L.p_com0 + 3]
    for c in range(2):
        X[:, L.pos[c]:L.pos[c] + 3 * (N + 1)] = P[:, L.p_nom[c]:L.p_nom[c] + 3 * (N + 1)]
        for j in range(4):
            L.x_force(X, c, j)[..., 2] = gravity / 8.0
    return X
