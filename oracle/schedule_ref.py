"""TEST INFRASTRUCTURE ONLY -- float64 / integer-nanosecond restatement, one problem, plain Python, of what happens to a
ContactPhaseList on either side of the solve:

  * sample_contact_phase_list   CentroidalMPC::setContactPhaseList, call site
                                src/centroidal-mpc-walking/src/CentroidalMPCBlock.cpp:609 (list built at :586-607);
  * adjust_contact_phase_list   what getOutput().contactPhaseList carries back (:598, :626): the next contact of a foot
                                that lands inside the horizon takes the position the MPC chose for it.

PARITY UNPINNED for the sampling rule itself: it lives inside BipedalLocomotionFramework, whose source is not under
/root/reference, and the reference holds no fixture for it.  What the reference does pin, and what this
file follows, are the two ContactList queries it makes and their documented meaning (the same the reference's own
updateContactPhaseList relies on, CentroidalMPCBlock.cpp:44, :61, :69 -- restated in oracle/contacts_ref.py):
    getActiveContact(t) = the contact with activationTime <= t < deactivationTime
    getNextContact(t)   = the contact with the lowest activationTime > t
and the clock: the caller's time is a std::chrono::nanoseconds (CentroidalMPCBlock.cpp:32, :631 `m_absoluteTime += m_dT`),
so every comparison here is made on integer nanoseconds -- knot k of the horizon is now + k * dT exactly.

The rule (stated in include/cmpc.h at cmpc_contacts_sample; this file is its independent restatement -- it shares no code
with the product's three statements of it, csrc/cmpc_contacts.h, contacts.sample_schedule, contacts.sample_schedule_batch):
    stage k covers [t_k, t_k+1), t_k = now + k dT
    Gamma_k      = 1 iff getActiveContact(t_k) exists
    owner_k      = getActiveContact(t_k), else getNextContact(t_k), else the last contact of the list
    R_k, upper_k, lower_k come from owner_k (the limits are the contact's bounding box: one per foot);
    nominalPos_{k+1} = position of owner_k; nominalPos_0 = currentPos = position of owner_0
    landing knot = the first k with Gamma_k = 1 and Gamma_{k-1} = 0; N if the foot is in the air at the end of the horizon and
                   has not landed before; -1 if it never leaves the ground.

A contact list is a list of dicts {activation, deactivation (seconds, float), position (3), quaternion (w x y z)} ordered by
activation, as in oracle/contacts_ref.py.  Only tests/ may import this."""
from .contacts_ref import get_active_contact, get_next_contact_index

NS_PER_S = 1_000_000_000


def _ns(t):
    return int(round(float(t) * NS_PER_S))


def _in_ns(contact_list):
    return [dict(c, activation=_ns(c["activation"]), deactivation=_ns(c["deactivation"])) for c in contact_list]


def quaternion_to_rotation(q):
    """unit quaternion (w, x, y, z) -> 3x3 rotation matrix (rows), float64"""
    w, x, y, z = (float(v) for v in q)
    return [[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]]


def stage_owner(contact_list_ns, t_ns):
    """-> (contact, is_active) of the stage that starts at t_ns (integer nanoseconds)."""
    c = get_active_contact(contact_list_ns, t_ns)                    # activation <= t < deactivation
    if c is not None:
        return c, True
    i = get_next_contact_index(contact_list_ns, t_ns)               # lowest activation > t
    if i < len(contact_list_ns):
        return contact_list_ns[i], False
    return contact_list_ns[-1], False


def sample_contact_list(N, dt, now, contact_list, box_upper, box_lower):
    """One foot.  -> dict(gamma[N], R[N] (3x3 rows), upper[N], lower[N], nominal[N+1], current[3], land) or None when the
    list is empty (setContactPhaseList has nothing to sample: the caller must not solve)."""
    if not contact_list:
        return None
    lst = _in_ns(contact_list)
    now_ns, dt_ns = _ns(now), _ns(dt)
    out = dict(gamma=[], R=[], upper=[], lower=[], nominal=[], current=None, land=-1)
    prev_active = True
    for k in range(N):
        owner, active = stage_owner(lst, now_ns + k * dt_ns)
        if k == 0:
            out["nominal"].append([float(v) for v in owner["position"]])
            out["current"] = [float(v) for v in owner["position"]]
        out["gamma"].append(1.0 if active else 0.0)
        out["R"].append(quaternion_to_rotation(owner["quaternion"]))
        out["upper"].append([float(v) for v in box_upper])
        out["lower"].append([float(v) for v in box_lower])
        out["nominal"].append([float(v) for v in owner["position"]])
        if active and not prev_active and out["land"] < 0:
            out["land"] = k
        prev_active = active
    if not prev_active and out["land"] < 0:
        out["land"] = N
    return out


def sample_contact_phase_list(N, dt, now, phase_list, boxes):
    """phase_list {foot name: contact list}, boxes {foot name: (upper[3], lower[3])} -> {foot name: sample_contact_list(...)}
    in the map's (alphabetical) order, the order of the reference's parameter blocks."""
    return {name: sample_contact_list(N, dt, now, phase_list[name], *boxes[name]) for name in sorted(phase_list)}


def adjust_contact_list(now, contact_list, land, landing_position):
    """getOutput().contactPhaseList for one foot: a copy of the list in which the next contact (getNextContact(now)) has taken
    `landing_position` -- if the foot lands inside the horizon (land >= 0) and there is a next contact; else an unchanged copy."""
    out = [dict(c) for c in contact_list]
    if land < 0:
        return out
    i = get_next_contact_index(_in_ns(contact_list), _ns(now))
    if i < len(out):
        out[i]["position"] = [float(v) for v in landing_position]
    return out
