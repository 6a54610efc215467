"""The C++ drop-in class (include/BipedalLocomotion/ReducedModelControllers/CentroidalMPC.h) linked
against libcmpc_hip.so: one MPC tick driven like CentroidalMPCBlock.cpp:407-622."""
import os
import subprocess

import pytest

import cmpc_amd as cm

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_facade_links_and_runs(tmp_path):
    pkg = os.path.dirname(cm._capi.LIB_PATH)
    exe = str(tmp_path / "facade_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(pkg, "csrc", "shim"),
                           os.path.join(ROOT, "examples", "facade_demo.cpp"), "-L", pkg, "-lcmpc_hip", f"-Wl,-rpath,{pkg}",
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    vals = {l.split()[0]: [float(x) for x in l.split()[1:]] for l in out.stdout.splitlines() if not l.startswith("contact")}
    assert abs(vals["total_fz"][0] - 9.8) < 1.0          # first-knot forces carry the (unit-mass) weight
    nx, ny, nz = vals["next_left"]
    assert abs(nx - 0.1) <= 0.01 + 1e-5 and -1e-5 <= ny - 0.08 <= 0.05 + 1e-5 and abs(nz) < 1e-6
