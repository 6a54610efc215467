"""Contact schedule -> MPC parameter tensors (the job of CentroidalMPC::setContactPhaseList).

Call site in the reference: src/centroidal-mpc-walking/src/CentroidalMPCBlock.cpp:609 (the list is
built at :586-607).  The sampling rule lives inside BipedalLocomotionFramework, whose source is not
in the reference tree (parity unpinned), so the rule is defined and documented here:

  knot k (time t0 + k*dt), stage k = [t_k, t_{k+1}):
    * Gamma_k = 1 iff a contact of that foot is active at t_k (activation <= t_k < deactivation);
    * the "owner" of stage k is that active contact, else the next contact to activate (else the
      last one);  R_k, the bounding-box limits of row k and nominal_{k+1} come from the owner, so a
      row always constrains pos_{k+1} against the contact that position belongs to;
    * nominal_0 = owner of stage 0;  currentPos = position of the contact active at t_0 (or the
      owner's when the foot is in the air).
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import Dict, List, Sequence

import numpy as np

TIME_EPS = 1e-9


@dataclasses.dataclass
class PlannedContact:
    """Subset of BipedalLocomotion::Contacts::PlannedContact used on this path."""
    activation_time: float
    deactivation_time: float
    position: Sequence[float]
    yaw: float = 0.0
    rotation: np.ndarray | None = None  # overrides yaw when given

    def R(self) -> np.ndarray:
        if self.rotation is not None:
            return np.asarray(self.rotation, float)
        c, s = np.cos(self.yaw), np.sin(self.yaw)
        return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def _owner(contacts: List[PlannedContact], t: float):
    for c in contacts:
        if c.activation_time <= t + 1e-9 and t + 1e-9 < c.deactivation_time:
            return c, True
    nxt = [c for c in contacts if c.activation_time > t + 1e-9]
    if nxt:
        return min(nxt, key=lambda c: c.activation_time), False
    return contacts[-1], False


def sample_schedule(cfg, lists: Dict[str, List[PlannedContact]], t0: float = 0.0):
    """One problem: returns dict(R[2,N,3,3], upper[2,N,3], lower[2,N,3], enabled[2,N],
    nominal[2,N+1,3], current[2,3]) for the contacts in cfg order (alphabetical)."""
    N, dt = cfg.N, cfg.sampling_time
    out = dict(R=np.zeros((2, N, 3, 3)), upper=np.zeros((2, N, 3)), lower=np.zeros((2, N, 3)),
               enabled=np.zeros((2, N)), nominal=np.zeros((2, N + 1, 3)), current=np.zeros((2, 3)))
    for ci, cc in enumerate(cfg.contacts):
        lst = sorted(lists[cc.contact_name], key=lambda c: c.activation_time)
        own0, act0 = _owner(lst, t0)
        out["nominal"][ci, 0] = own0.position
        out["current"][ci] = own0.position
        for k in range(N):
            own, act = _owner(lst, t0 + k * dt)
            out["enabled"][ci, k] = 1.0 if act else 0.0
            out["R"][ci, k] = own.R()
            out["upper"][ci, k] = cc.bounding_box_upper_limit
            out["lower"][ci, k] = cc.bounding_box_lower_limit
            out["nominal"][ci, k + 1] = own.position
    return out


# ---------------------------------------------------------------------------------------------------------------
# Batched form.  One foot of one problem = at most M contacts sorted by activation time:
#   t[B,2,M,2] (activation, deactivation; float64 seconds), pose[B,2,M,7] (x y z, quaternion w x y z; float32),
#   n[B,2] contacts in use -- the layout of include/cmpc.h (cmpc_contacts_*).
# ---------------------------------------------------------------------------------------------------------------
def quat_from_R(R) -> np.ndarray:
    """Rotation matrix -> unit quaternion (w, x, y, z), w >= 0."""
    R = np.asarray(R, float)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = 2.0 * np.sqrt(1.0 + tr)
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax([R[0, 0], R[1, 1], R[2, 2]]))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = 2.0 * np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k])
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q if q[0] >= 0 else -q


def R_from_quat(q) -> np.ndarray:
    """[..., 4] (w, x, y, z) -> [..., 3, 3], the float32 formula of csrc/cmpc_contacts.h evaluated in the dtype of q."""
    q = np.asarray(q)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    one, two = q.dtype.type(1), q.dtype.type(2)
    R = np.empty(q.shape[:-1] + (3, 3), q.dtype)
    R[..., 0, 0] = one - two * (y * y + z * z); R[..., 0, 1] = two * (x * y - w * z); R[..., 0, 2] = two * (x * z + w * y)
    R[..., 1, 0] = two * (x * y + w * z); R[..., 1, 1] = one - two * (x * x + z * z); R[..., 1, 2] = two * (y * z - w * x)
    R[..., 2, 0] = two * (x * z - w * y); R[..., 2, 1] = two * (y * z + w * x); R[..., 2, 2] = one - two * (x * x + y * y)
    return R


def pack_lists(cfg, lists: Sequence[Dict[str, List[PlannedContact]]], max_contacts: int | None = None):
    """Per-problem dicts {contact_name: [PlannedContact, ...]} -> (t[B,2,M,2], pose[B,2,M,7], n[B,2])."""
    B = len(lists)
    names = [c.contact_name for c in cfg.contacts]
    M = max_contacts or max(len(l[nm]) for l in lists for nm in names)
    t = np.zeros((B, 2, M, 2))
    pose = np.zeros((B, 2, M, 7), np.float32)
    pose[..., 3] = 1.0
    n = np.zeros((B, 2), np.int32)
    for b, l in enumerate(lists):
        for ci, nm in enumerate(names):
            lst = sorted(l[nm], key=lambda c: c.activation_time)
            n[b, ci] = len(lst)
            for m, ct in enumerate(lst):
                t[b, ci, m] = (ct.activation_time, ct.deactivation_time)
                pose[b, ci, m, :3] = ct.position
                pose[b, ci, m, 3:] = quat_from_R(ct.R())
    return t, pose, n


def unpack_lists(cfg, t, pose, n) -> List[Dict[str, List[PlannedContact]]]:
    names = [c.contact_name for c in cfg.contacts]
    out = []
    for b in range(t.shape[0]):
        d = {}
        for ci, nm in enumerate(names):
            d[nm] = [PlannedContact(float(t[b, ci, m, 0]), float(t[b, ci, m, 1]), tuple(float(v) for v in pose[b, ci, m, :3]),
                                    rotation=R_from_quat(pose[b, ci, m, 3:].astype(np.float64))) for m in range(int(n[b, ci]))]
        out.append(d)
    return out


def _first_true(mask):
    """index of the first True along the last axis, and whether there is one"""
    return mask.argmax(-1), mask.any(-1)


def sample_schedule_batch(cfg, t, pose, n, t0: float = 0.0):
    """setContactPhaseList for a batch, vectorised over the problems (loops over the N knots and the two feet only):
    the rule of sample_schedule / csrc/cmpc_contacts.h.  Returns dict(R[B,2,N,3,3], upper, lower[B,2,N,3],
    enabled[B,2,N], nominal[B,2,N+1,3], current[B,2,3]) float32 and land[B,2] (landing knot, N, or -1)."""
    N, dt = cfg.N, cfg.sampling_time
    B, _, M, _ = t.shape
    idx = np.arange(M)[None, :]
    out = dict(R=np.zeros((B, 2, N, 3, 3), np.float32), upper=np.zeros((B, 2, N, 3), np.float32), lower=np.zeros((B, 2, N, 3), np.float32),
               enabled=np.zeros((B, 2, N), np.float32), nominal=np.zeros((B, 2, N + 1, 3), np.float32), current=np.zeros((B, 2, 3), np.float32))
    land = np.full((B, 2), -1, np.int32)
    rows = np.arange(B)
    for ci, cc in enumerate(cfg.contacts):
        valid = idx < n[:, ci, None]
        act_t, deact_t = t[:, ci, :, 0], t[:, ci, :, 1]
        Rall = R_from_quat(pose[:, ci, :, 3:].astype(np.float32))
        prev = np.ones(B, bool)
        for k in range(N):
            tk = t0 + k * dt + TIME_EPS
            ia, has_a = _first_true(valid & (act_t <= tk) & (tk < deact_t))
            inx, has_n = _first_true(valid & (act_t > tk))
            own = np.where(has_a, ia, np.where(has_n, inx, n[:, ci] - 1))
            out["enabled"][:, ci, k] = has_a
            out["R"][:, ci, k] = Rall[rows, own]
            out["upper"][:, ci, k] = cc.bounding_box_upper_limit
            out["lower"][:, ci, k] = cc.bounding_box_lower_limit
            out["nominal"][:, ci, k + 1] = pose[rows, ci, own, :3]
            if k == 0:
                out["nominal"][:, ci, 0] = pose[rows, ci, own, :3]
                out["current"][:, ci] = pose[rows, ci, own, :3]
            newly = has_a & ~prev & (land[:, ci] < 0)
            land[newly, ci] = k
            prev = has_a
        land[(~prev) & (land[:, ci] < 0), ci] = N
    return out, land


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def update_contact_phase_list(now: float, plan, mpc):
    """updateContactPhaseList (CentroidalMPCBlock.cpp:32-110) for a batch, through the C ABI (cmpc_contacts_merge):
    plan / mpc = (t, pose, n) of the planner's lists and of the MPC's previous output.  Returns ((t, pose, n), ok[B]):
    ok[b] is False where the reference returns false (the planner has no active contact under an active MPC contact)."""
    from . import _capi
    pt, pp, pn = (np.ascontiguousarray(plan[0], np.float64), np.ascontiguousarray(plan[1], np.float32), np.ascontiguousarray(plan[2], np.int32))
    mt, mp, mn = (np.ascontiguousarray(mpc[0], np.float64), np.ascontiguousarray(mpc[1], np.float32), np.ascontiguousarray(mpc[2], np.int32))
    B, _, M, _ = pt.shape
    assert mt.shape == pt.shape and pp.shape == (B, 2, M, 7) and mp.shape == pp.shape
    ot, op, on = np.zeros_like(pt), np.zeros_like(pp), np.zeros_like(pn)
    op[..., 3] = 1.0
    ok = np.zeros(B, np.int32)
    rc = _capi.lib().cmpc_contacts_merge(B, M, float(now), _ptr(pt), _ptr(pp), _ptr(pn), _ptr(mt), _ptr(mp), _ptr(mn), _ptr(ot), _ptr(op),
                                         _ptr(on), _ptr(ok))
    if rc not in (0, -1):
        raise RuntimeError(f"cmpc_contacts_merge failed ({rc})")
    return (ot, op, on), ok.astype(bool)
