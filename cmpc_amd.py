"""Import alias: the package directory name has a '-' in it (mandated by the repo layout)."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_pkg = importlib.import_module("paper_romualdi_2022_icra_centroidal-mpc-walking_amd")
sys.modules[__name__] = _pkg
