"""TEST INFRASTRUCTURE ONLY -- float64 numpy restatement of the rows next to the solve.

resample_references: src/centroidal-mpc-walking/src/CentroidalMPCBlock.cpp:525-577 (angular momentum / mass, CoM height
  forced to 0.7, LinearSpline from the planner's knots to the N+1 MPC knots).  BLF's LinearSpline is not in the
  reference tree; plain piecewise-linear interpolation clamped at the ends is assumed (parity unpinned).
plant_step: src/centroidal-mpc-walking/src/WholeBodyQPBlock.cpp:1083-1084 (control input = MPC contacts + external
  wrench), :1150 (integrate one WBC period, RK4), :1259-1262 (feedback = integrated com, dcom, h), :805-873 (desired
  ZMP from the corner forces).  Mass-normalised like the MPC; BLF's CentroidalDynamics is not in the tree (unpinned).
"""
import numpy as np


def resample_references(com_in, h_in, in_dt, t_offset, N, dt, mass, com_height=0.7):
    n_in = com_in.shape[0]
    com_ref = np.zeros((N + 1, 3)); h_ref = np.zeros((N + 1, 3))
    for k in range(N + 1):
        s = min(max((t_offset + k * dt) / in_dt, 0.0), n_in - 1.0)
        i0 = min(int(s), n_in - 2)
        w = s - i0
        com_ref[k] = (1 - w) * com_in[i0] + w * com_in[i0 + 1]
        h_ref[k] = ((1 - w) * h_in[i0] + w * h_in[i0 + 1]) / mass
        if com_height == com_height:
            com_ref[k, 2] = com_height
    return com_ref, h_ref


def plant_step(L, corners, x, p, state, step, substeps, gravity=9.80665, zx=0.08, zy=0.03):
    """-> (new_state[9], zmp[2]); L = package Layout, corners[2][4][3]."""
    com, v, h = state[0:3].astype(float).copy(), state[3:6].astype(float).copy(), state[6:9].astype(float).copy()
    cp, cf = [], []
    ztot, zw = 0.0, np.zeros(2)
    for c in range(2):
        R = p[L.p_R[c]:L.p_R[c] + 9].reshape(3, 3, order="F")
        pos = x[L.pos[c]:L.pos[c] + 3]
        on = p[L.p_gam[c]] > 0.5
        F = np.zeros(3); T = np.zeros(3)
        for j in range(4):
            f = x[L.f[c][j]:L.f[c][j] + 3] if on else np.zeros(3)
            cp.append(pos + R @ corners[c][j]); cf.append(np.asarray(f, float))
            F += f
            T += np.cross(corners[c][j], R.T @ f)
        if F[2] > 0.001:
            lz = np.array([np.clip(-T[1] / F[2], -zx, zx), np.clip(T[0] / F[2], -zy, zy), 0.0])
            ztot += F[2]
            zw += F[2] * (pos + R @ lz)[:2]
    fsum = np.sum(cf, axis=0)
    fext, text = p[L.p_fext:L.p_fext + 3], p[L.p_text:L.p_text + 3]

    def deriv(cm, vv):
        dv = fsum + fext - np.array([0, 0, gravity])
        dh = text + sum(np.cross(cp[q] - cm, cf[q]) for q in range(8))
        return vv, dv, dh

    for _ in range(substeps):
        k1 = deriv(com, v)
        k2 = deriv(com + 0.5 * step * k1[0], v + 0.5 * step * k1[1])
        k3 = deriv(com + 0.5 * step * k2[0], v + 0.5 * step * k2[1])
        k4 = deriv(com + step * k3[0], v + step * k3[1])
        com = com + step / 6 * (k1[0] + 2 * k2[0] + 2 * k3[0] + k4[0])
        v = v + step / 6 * (k1[1] + 2 * k2[1] + 2 * k3[1] + k4[1])
        h = h + step / 6 * (k1[2] + 2 * k2[2] + 2 * k3[2] + k4[2])
    zmp = zw / ztot if ztot > 0.001 else np.full(2, np.nan)
    return np.concatenate([com, v, h]), zmp
