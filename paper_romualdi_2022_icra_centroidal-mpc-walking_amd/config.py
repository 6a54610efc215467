"""Configuration of the centroidal MPC: the keys CentroidalMPC::initialize() reads.

Mirrors the `[CENTROIDAL_MPC]` group the reference passes at
src/centroidal-mpc-walking/src/CentroidalMPCBlock.cpp:144 (files
src/centroidal-mpc-walking/config/robots/*/centroidal_mpc.ini); both key generations are accepted
(`sampling_time`/`time_horizon`, ergoCubGazeboV1/centroidal_mpc.ini:3-4, and the older
`controller_sampling_time`/`controller_horizon`, iCubGazeboV3/centroidal_mpc.ini:3-4).
"""
from __future__ import annotations

import dataclasses
import re
from typing import Dict, List, Sequence, Tuple

GRAVITY = 9.80665  # BLF StandardAccelerationOfGravitation


@dataclasses.dataclass
class ContactConfig:
    contact_name: str
    corners: List[Tuple[float, float, float]]
    bounding_box_upper_limit: Tuple[float, float, float]
    bounding_box_lower_limit: Tuple[float, float, float]


@dataclasses.dataclass
class CentroidalMPCConfig:
    sampling_time: float
    horizon_steps: int
    static_friction_coefficient: float = 0.33
    number_of_slices: int = 1
    com_weight: Tuple[float, float, float] = (10.0, 10.0, 200.0)
    contact_position_weight: float = 2e3
    force_rate_of_change_weight: Tuple[float, float, float] = (10.0, 10.0, 10.0)
    angular_momentum_weight: float = 1e2
    contact_force_symmetry_weight: float = 100.0
    contacts: List[ContactConfig] = dataclasses.field(default_factory=list)
    # solver options (ipopt_* names kept so reference ini files load unchanged)
    ipopt_tolerance: float = 1e-4
    ipopt_max_iteration: int = 40
    is_warm_start_enabled: bool = True
    solver_verbosity: int = 0

    def __post_init__(self):
        if self.number_of_slices != 1:
            raise ValueError("only number_of_slices == 1 (4-face friction pyramid) is supported, "
                             "as in every shipped centroidal_mpc.ini")
        if len(self.contacts) != 2 or any(len(c.corners) != 4 for c in self.contacts):
            raise ValueError("exactly 2 contacts with 4 corners each are supported "
                             "(number_of_maximum_contacts 2, number_of_corners 4)")
        # std::map<std::string, ...> ordering of the reference: lexicographic by contact name
        self.contacts = sorted(self.contacts, key=lambda c: c.contact_name)

    @property
    def N(self) -> int:
        return self.horizon_steps


def _foot(name, cy, up, lo, cx=0.08):
    return ContactConfig(name, [(cx, cy, 0.0), (cx, -cy, 0.0), (-cx, -cy, 0.0), (-cx, cy, 0.0)], up, lo)


def ergocub_gazebo_v1(horizon_steps: int = 20, sampling_time: float = 0.06) -> CentroidalMPCConfig:
    """config/robots/ergoCubGazeboV1/centroidal_mpc.ini:3-42"""
    return CentroidalMPCConfig(
        sampling_time=sampling_time, horizon_steps=horizon_steps,
        com_weight=(10.0, 10.0, 200.0), contact_position_weight=2e3,
        force_rate_of_change_weight=(10.0, 10.0, 10.0), angular_momentum_weight=1e2,
        contact_force_symmetry_weight=100.0,
        contacts=[_foot("left_foot", 0.01, (0.01, 0.05, 0.0), (-0.01, -0.0, 0.0)),
                  _foot("right_foot", 0.01, (0.01, 0.0, 0.0), (-0.01, -0.05, 0.0))])


def icub_gazebo_v3(horizon_steps: int = 10, sampling_time: float = 0.1) -> CentroidalMPCConfig:
    """config/robots/iCubGazeboV3/centroidal_mpc.ini:3-41 (no symmetry-weight key there -> 0)."""
    return CentroidalMPCConfig(
        sampling_time=sampling_time, horizon_steps=horizon_steps,
        com_weight=(1.0, 1.0, 200.0), contact_position_weight=2e2,
        force_rate_of_change_weight=(10.0, 10.0, 10.0), angular_momentum_weight=1e2,
        contact_force_symmetry_weight=0.0,
        contacts=[_foot("left_foot", 0.03, (0.01, 0.05, 0.0), (-0.01, -0.0, 0.0)),
                  _foot("right_foot", 0.03, (0.01, 0.0, 0.0), (-0.01, -0.05, 0.0))])


def generated_code_weights(which: str = "tmp", horizon_steps: int = 12,
                           sampling_time: float = 0.1) -> CentroidalMPCConfig:
    """The two weight sets baked into the reference's generated NLP code
    (config/robots/ergoCubGazeboV1/tmp.c and jit_tmpComMiH.c)."""
    cfg = ergocub_gazebo_v1(horizon_steps, sampling_time)
    cfg.contact_position_weight = 200.0
    if which == "tmp":
        cfg.com_weight, cfg.contact_force_symmetry_weight = (10.0, 10.0, 200.0), 10.0
    else:
        cfg.com_weight, cfg.contact_force_symmetry_weight = (10.0, 100.0, 200.0), 100.0
    return cfg


# ---------------------------------------------------------------- YARP .ini reader (key subset)
_NUM = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?"


def _parse_value(txt: str):
    txt = txt.strip()
    if txt.startswith("("):
        return tuple(float(v) for v in re.findall(_NUM, txt))
    if txt.startswith('"'):
        return txt.strip('"')
    if txt in ("true", "false"):
        return txt == "true"
    try:
        return int(txt)
    except ValueError:
        try:
            return float(txt)
        except ValueError:
            return txt


def parse_ini(text: str) -> Dict[str, dict]:
    """Minimal reader for the YARP ResourceFinder ini subset the MPC files use:
    `key value`, `key (a, b, c)`, `[GROUP]`, `#` comments."""
    groups: Dict[str, dict] = {"": {}}
    cur = groups[""]
    for raw in text.splitlines():
        line = raw.split("#", 1)[0].strip()
        if not line:
            continue
        m = re.match(r"\[(\w+)\]$", line)
        if m:
            cur = groups.setdefault(m.group(1), {})
            continue
        parts = line.split(None, 1)
        if len(parts) == 2:
            cur[parts[0]] = _parse_value(parts[1])
    return groups


def from_ini(text: str) -> CentroidalMPCConfig:
    g = parse_ini(text)
    top = g[""]
    if "sampling_time" in top:
        dt = float(top["sampling_time"])
        n = int(round(float(top["time_horizon"]) / dt))
    elif "controller_sampling_time" in top:
        dt = float(top["controller_sampling_time"])
        n = int(top["controller_horizon"])
    else:
        raise KeyError("sampling_time / controller_sampling_time missing")
    ncontacts = int(top.get("number_of_maximum_contacts", 2))
    contacts = []
    for i in range(ncontacts):
        c = g[f"CONTACT_{i}"]
        nc = int(c["number_of_corners"])
        contacts.append(ContactConfig(
            c["contact_name"], [tuple(c[f"corner_{j}"]) for j in range(nc)],
            tuple(c["bounding_box_upper_limit"]), tuple(c["bounding_box_lower_limit"])))
    return CentroidalMPCConfig(
        sampling_time=dt, horizon_steps=n,
        static_friction_coefficient=float(top.get("static_friction_coefficient", 0.33)),
        number_of_slices=int(top.get("number_of_slices", 1)),
        com_weight=tuple(top["com_weight"]),
        contact_position_weight=float(top["contact_position_weight"]),
        force_rate_of_change_weight=tuple(top["force_rate_of_change_weight"]),
        angular_momentum_weight=float(top["angular_momentum_weight"]),
        contact_force_symmetry_weight=float(top.get("contact_force_symmetry_weight", 0.0)),
        contacts=contacts,
        ipopt_tolerance=float(top.get("ipopt_tolerance", 1e-4)),
        ipopt_max_iteration=int(top.get("ipopt_max_iteration", 40)),
        is_warm_start_enabled=bool(top.get("is_warm_start_enabled", True)),
        solver_verbosity=int(top.get("solver_verbosity", 0)))
