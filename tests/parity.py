"""Parity metrics shared by the oracle and GPU tests.

north_star tolerance: <= 1e-4 relative error on the CoM trajectory and the contact forces, against
the converged float64 solve of the same NLP (SURVEY 7 hard part 5 / 8d).

Contact forces are compared modulo the one direction the NLP does not determine: when both feet
are in stance over the whole horizon, a constant internal force along the line joining the two
feet (left corners +e, right corners -e, every knot) changes neither the dynamics (zero net force,
zero net moment) nor the cost (the symmetry cost sees differences between corners of one foot, the
rate cost differences in time), so the optimal set is a segment, not a point, and two exact solvers
(or two IPOPT linear solvers) land on different points of it.
"""
import numpy as np

import cmpc_amd as cm


def _internal_force_direction(L, p):
    N = L.N
    gam = np.concatenate([p[L.p_gam[c]:L.p_gam[c] + N] for c in range(2)])
    if not (gam > 0.5).all():
        return None
    e = p[L.p_cur[0]:L.p_cur[0] + 3] - p[L.p_cur[1]:L.p_cur[1] + 3]
    e = e / np.linalg.norm(e)
    n = np.zeros(L.nx)
    for c, sgn in ((0, 1.0), (1, -1.0)):
        for j in range(4):
            n[L.f[c][j]:L.f[c][j] + 3 * N] = np.tile(sgn * e, N)
    return n / np.linalg.norm(n)


def errors(N, p, x, x_ref):
    """-> dict(com, force0, forces, pos, dcom, h): relative (max-norm) errors; pos absolute [m]."""
    L = cm.Layout(N)
    x = np.asarray(x, np.float64)
    x_ref = np.asarray(x_ref, np.float64)
    d = x - x_ref
    n = _internal_force_direction(L, np.asarray(p, np.float64))
    if n is not None:
        d = d - n * (n @ d)

    def rel(a, b):
        return float(np.abs(a).max() / max(np.abs(b).max(), 1e-12))

    fall = np.concatenate([np.arange(L.f[c][j], L.f[c][j] + 3 * N) for c in range(2) for j in range(4)])
    return dict(
        com=rel(L.x_com(d), L.x_com(x_ref)),
        dcom=rel(L.x_dcom(d), np.maximum(np.abs(L.x_dcom(x_ref)), 1e-2)),
        h=float(np.abs(L.x_h(d)).max()),
        pos=float(max(np.abs(L.x_pos(d, c)).max() for c in range(2))),
        force0=rel(L.first_forces(d), L.first_forces(x_ref)),
        forces=rel(d[fall], x_ref[fall]),
    )


TOL = 1e-4   # north_star: relative error on the CoM trajectory and the contact forces


def limits(N):
    """Tolerance per quantity: north_star's 1e-4 on the CoM trajectory, the contact forces (first knot AND every knot), the footsteps AND the
    CoM velocity at every horizon; 3e-5 absolute (mass-normalised, m^2/s) on the angular momentum, which north_star does not name (measured
    worst of 2 560 problems per config: 3e-6 / 2.1e-5 / 5e-6, profiles/r04_accuracy_sweep.txt).  History of the N = 30 CoM velocity: round 2 allowed 5e-4 and measured 2.3e-4 (unloaded corners of the LAST stages sit
    sqrt(mu / curvature) inside their friction pyramid at the barrier floor: the tail polish of round 3 removed that); round 3 allowed 1.3e-4 and
    measured 1.05e-4 (what is left is fed by complementarity products that lag above the floor at termination); since round 4 the default
    tolerance is 3e-7 beyond N = 20 (cmpc_create) and the limit is 1e-4 like everything else (profiles/r04_accuracy_sweep.txt)."""
    return dict(com=TOL, force0=TOL, pos=TOL, forces=TOL, dcom=TOL, h=3e-5)


def assert_no_sync_giveups(info):
    """info[:, 3] carries 1e6 per give-up of a wave of the streaming backward stage at a hand-off word (include/cmpc.h): a protocol bug would
    show there and nowhere else."""
    assert (np.asarray(info)[:, 3] < 1e6).all(), np.asarray(info)[:, 3].max()


def worst_errors(N, P, X, Xref):
    worst = dict(com=0.0, force0=0.0, forces=0.0, dcom=0.0, pos=0.0, h=0.0)
    for b in range(P.shape[0]):
        e = errors(N, P[b], X[b], Xref[b])
        for k in worst:
            worst[k] = max(worst[k], e[k])
    return worst


def assert_within(N, worst):
    lim = limits(N)
    bad = {k: (worst[k], lim[k]) for k in lim if not worst[k] < lim[k]}
    assert not bad, (bad, worst)
