"""The stream-safety clause of the C ABI (include/trs_solver.h): "concurrent calls on distinct
streams with distinct buffers do not interact".  Rounds 3-4 broke it - a race in trs_joint_order's breadth-first sweep
that concurrent kernels brought out (EXPERIMENTS R5.1) - so it is asserted here, each test in a PROCESS OF ITS OWN under
a timeout (a device stall must end the test, not the suite):

* the torch-free reproducer `tools/repro_streams.cpp` (hipMalloc + hipStream_t + the C ABI, nothing else): the launch
  sequences of a ragged batch's buckets on two, three and four streams, 200 steps, every step compared bit for bit
  with a one-stream step and every joint order checked for being a permutation;
* `tools/repro_families.cpp`, torch-free as well: every OTHER kernel family (fused small-system kernel with fitness, GA
  sections, graph features, fitness, row copies on a CU-masked stream through page-locked host memory, the generator)
  on streams of their own beside that pipeline, 200 steps, every output bit for bit that of a serial run;
* `RaggedSolver(lanes=)` through PyTorch streams: 1 / 2 / 3 / 4 lanes, two section variants, a second solver on the same
  workspaces right behind, on the default and on a side stream (`tools/lanes_check.py`)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "python_stable_3d_truss_analysis_amd")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _binary(name="repro_streams"):
    """A reproducer under tools/ (test infrastructure, a plain HIP program), compiled here when the box has none or an
    older one than its source (hipcc is part of the image; `__graft_entry__.build()` tries to build them ahead)."""
    source, binary = os.path.join(ROOT, "tools", name + ".cpp"), os.path.join(ROOT, "tools", name)
    if not os.path.exists(binary) or os.path.getmtime(binary) < os.path.getmtime(source):
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", source, "-o", binary,
                        "-ldl", "-lpthread"], check=True, timeout=300)
    return binary


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("lanes", [2, 3, 4])
def test_streams_torch_free_bitwise_equal_over_200_steps(lanes):
    """(4 = `batch.DEFAULT_LANES`, with the noise kernel beside the lanes as for 3)"""
    run = subprocess.run([_binary(), os.path.join(PKG, "libtrs_hip.so"), "--trusses", "8192", "--lanes", str(lanes),
                          "--steps", "200", "--variants", "2", "--noise", "1" if lanes >= 3 else "0"],
                         capture_output=True, text=True, timeout=500, cwd=ROOT)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-2000:]
    result = [l for l in run.stdout.splitlines() if l.startswith("RESULT")]
    assert result and "steps=200: 0 steps with differences, 0 truss results, 0 non-permutation orders" in result[-1], run.stdout[-2000:]


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("noise", [0, 1])
def test_every_kernel_family_beside_the_others_bitwise_equal_over_200_steps(noise):
    """`tools/repro_families.cpp`: the ragged pipeline, `trs_solve_small` + fitness, a GA generation (`trs_ga_sections`
    -> `trs_solve_small`), `trs_graph_features_packed` + `trs_fitness`, and - on a CU-masked stream of
    `trs_stream_create_masked` - `trs_copy_rows` pulls / pushes through page-locked host memory and `trs_cubegen_dev`,
    each on a stream of its own at the same time (with and without the spinning noise kernel): every output of every
    step equals, bit for bit, what the same calls give one after the other on one stream."""
    try:
        run = subprocess.run([_binary("repro_families"), os.path.join(PKG, "libtrs_hip.so"), "--trusses", "8192", "--small",
                              "8192", "--steps", "200", "--noise", str(noise)],
                             capture_output=True, text=True, timeout=240, cwd=ROOT)
    except subprocess.TimeoutExpired as exc:   # (the tool's own watchdog names the phase of a stall; this is the backstop)
        raise AssertionError(f"repro_families did not finish: {(exc.stdout or b'')[-2000:]!r}") from None
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-2000:]
    result = [l for l in run.stdout.splitlines() if l.startswith("RESULT")]
    assert result and "steps=200: 0 steps with differences, 0 outputs" in result[-1], run.stdout[-2000:]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_ragged_solver_lanes_give_the_same_bits():
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lanes_check.py")], capture_output=True, text=True,
                         timeout=500, cwd=ROOT)
    assert run.returncode == 0 and "lanes ok" in run.stdout, run.stdout[-2000:] + run.stderr[-3000:]
