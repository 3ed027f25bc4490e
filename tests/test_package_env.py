"""The package asks the HIP runtime for eight hardware queues unless the caller has chosen (EXPERIMENTS R5.5: streams
beyond `GPU_MAX_HW_QUEUES` share a queue, and the package's own default uses more than four streams)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _queues_after_import(preset):
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    if preset is not None:
        env["GPU_MAX_HW_QUEUES"] = preset
    code = f"import sys, os; sys.path.insert(0, {ROOT!r}); import python_stable_3d_truss_analysis_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=60).stdout.strip()


def test_hardware_queue_default_respects_the_callers_choice():
    assert _queues_after_import(None) == "8"
    assert _queues_after_import("2") == "2"
