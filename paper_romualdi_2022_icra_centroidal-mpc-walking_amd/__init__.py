"""MI355X-native batched centroidal-MPC solver (drop-in for the CentroidalMPC::advance() path of
GiulioRomualdi/paper_romualdi_2022_icra_centroidal-mpc-walking).

The directory name contains '-', so import it with
    importlib.import_module("paper_romualdi_2022_icra_centroidal-mpc-walking_amd")
(tests and bench.py do that through `cmpc_amd.py` at the repo root).
"""
from . import _capi, config, contacts, distributed, layout, rollout, solver, synthetic  # noqa: F401
from .config import CentroidalMPCConfig, ContactConfig  # noqa: F401
from .layout import Layout, cold_start, pack_parameters  # noqa: F401
from .solver import BatchSolver, CentroidalMPC  # noqa: F401
