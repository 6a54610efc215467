"""The C++ drop-in class (include/BipedalLocomotion/ReducedModelControllers/CentroidalMPC.h) linked against
libcmpc_hip.so and driven for 12 ticks across a lift-off exactly the way the reference's block drives it
(examples/facade_demo.cpp = the Block of csrc/facade_check.cpp, shaped after CentroidalMPCBlock.cpp:396-411, 579-631):
no call the reference does not make -- in particular nobody tells the controller the time."""
import os
import subprocess

import pytest

import cmpc_amd as cm

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_facade_links_and_walks(tmp_path):
    pkg = os.path.dirname(cm._capi.LIB_PATH)
    exe = str(tmp_path / "facade_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(pkg, "csrc", "shim"),
                           os.path.join(ROOT, "examples", "facade_demo.cpp"), "-L", pkg, "-lcmpc_hip", f"-Wl,-rpath,{pkg}",
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    ticks = [l.split() for l in out.stdout.splitlines() if l.startswith("tick ")]
    assert len(ticks) == 12
    ncontacts = [int(t[3]) for t in ticks]
    # the left foot lifts at 0.36 s = tick 6: the class's own clock (never set by the caller) must see it
    assert ncontacts[:6] == [2] * 6 and ncontacts[6:] == [1] * 6, ncontacts
    for t in ticks:
        assert abs(float(t[5]) - 9.8) < 1.5          # first-knot forces carry the (unit-mass) weight
        assert abs(float(t[9]) - 0.7) < 0.02 and abs(float(t[7])) < 0.1 and abs(float(t[8])) < 0.1   # the CoM stays put
    vals = {l.split()[0]: [float(x) for x in l.split()[1:]] for l in out.stdout.splitlines() if not l.startswith("tick ")}
    nx, ny, nz = vals["next_left"]
    assert abs(nx - 0.1) <= 0.01 + 1e-5 and -1e-5 <= ny - 0.08 <= 0.05 + 1e-5 and abs(nz) < 1e-6
