"""The stream-safety clause of the C ABI (include/trs_solver.h): "concurrent calls on distinct
streams with distinct buffers do not interact".  Rounds 3-4 broke it - a race in trs_joint_order's breadth-first sweep
that concurrent kernels brought out (EXPERIMENTS R5.1) - so it is asserted here, each test in a PROCESS OF ITS OWN under
a timeout (a device stall must end the test, not the suite):

* the torch-free reproducer `tools/repro_streams.cpp` (hipMalloc + hipStream_t + the C ABI, nothing else): the launch
  sequences of a ragged batch's buckets on two and three streams, 200 steps, every step compared bit for bit with a
  one-stream step and every joint order checked for being a permutation;
* `RaggedSolver(lanes=)` through PyTorch streams: 1 / 2 / 3 lanes, two section variants, a second solver on the same
  workspaces right behind, on the default and on a side stream (`tools/lanes_check.py`)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "python_stable_3d_truss_analysis_amd")
BINARY = os.path.join(ROOT, "tools", "repro_streams")
SOURCE = os.path.join(ROOT, "tools", "repro_streams.cpp")


def _binary():
    """The reproducer, built by `__graft_entry__.build()`; compiled here if the box has none (hipcc is part of the image)."""
    if not os.path.exists(BINARY) or os.path.getmtime(BINARY) < os.path.getmtime(SOURCE):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", SOURCE, "-o", BINARY,
                        "-ldl", "-lpthread"], check=True, timeout=300)
    return BINARY


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("lanes", [2, 3])
def test_streams_torch_free_bitwise_equal_over_200_steps(lanes):
    run = subprocess.run([_binary(), os.path.join(PKG, "libtrs_hip.so"), "--trusses", "8192", "--lanes", str(lanes),
                          "--steps", "200", "--variants", "2", "--noise", "1" if lanes == 3 else "0"],
                         capture_output=True, text=True, timeout=500, cwd=ROOT)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-2000:]
    result = [l for l in run.stdout.splitlines() if l.startswith("RESULT")]
    assert result and "steps=200: 0 steps with differences, 0 truss results, 0 non-permutation orders" in result[-1], run.stdout[-2000:]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_ragged_solver_lanes_give_the_same_bits():
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lanes_check.py")], capture_output=True, text=True,
                         timeout=500, cwd=ROOT)
    assert run.returncode == 0 and "lanes ok" in run.stdout, run.stdout[-2000:] + run.stderr[-3000:]
