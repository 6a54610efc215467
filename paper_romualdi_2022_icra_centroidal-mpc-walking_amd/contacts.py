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

import dataclasses
from typing import Dict, List, Sequence

import numpy as np


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
