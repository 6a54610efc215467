"""TEST INFRASTRUCTURE ONLY -- float64 restatement of the reference's updateContactPhaseList
(src/centroidal-mpc-walking/src/CentroidalMPCBlock.cpp:32-110; call site :594-607), one problem, plain Python.

A contact list is a list of dicts {activation, deactivation, position(3), quaternion(4: w x y z)} ordered by
activation time, as BipedalLocomotion::Contacts::ContactList keeps them; a phase list is {foot name: contact list}.
ContactList's two queries the reference uses are restated from their documented meaning (BLF's source is not in the
reference tree): getActiveContact(t) = the contact with activationTime <= t < deactivationTime, getNextContact(t) =
the contact with the lowest activationTime > t.

Only tests/ may import this."""


def get_active_contact(contact_list, t):
    for c in contact_list:
        if c["activation"] <= t < c["deactivation"]:
            return c
    return None


def get_next_contact_index(contact_list, t):
    for i, c in enumerate(contact_list):
        if c["activation"] > t:
            return i
    return len(contact_list)


def update_contact_phase_list(current_time, mann_phase_list, mpc_phase_list):
    """-> (ok, contact_phase_list).  Line numbers: CentroidalMPCBlock.cpp."""
    contact_list_map = {}
    for name, contact_list in mann_phase_list.items():                       # :41
        new_list = contact_list_map.setdefault(name, [])
        # every contact of the planner that activates after now             # :44-58
        for c in contact_list[get_next_contact_index(contact_list, current_time):]:
            new_list.append(dict(c))
        mpc_list = mpc_phase_list[name]                                      # :60
        mpc_present = get_active_contact(mpc_list, current_time)             # :61
        if mpc_present is None:                                              # :63-67  nothing to do
            continue
        mann_present = get_active_contact(contact_list, current_time)        # :69
        if mann_present is None:                                             # :70-77
            return False, None
        contact = dict(mpc_present)                                          # :79  pose from the MPC ...
        contact["activation"] = mann_present["activation"]                   # :80  ... timing from the planner
        contact["deactivation"] = mann_present["deactivation"]               # :81
        new_list.append(contact)                                             # :82  (ContactList orders by time)
        new_list.sort(key=lambda c: c["activation"])
    return True, contact_list_map                                            # :107
