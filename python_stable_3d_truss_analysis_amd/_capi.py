"""ctypes binding of the C ABI in `include/trs_solver.h` (library: `libtrs_hip.so`, in-tree).

There is no fallback: if the library is missing, `load()` raises `HipExtensionError`.
"""
import ctypes
import os

from .utils import HipExtensionError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtrs_hip.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

_P = ctypes.c_void_p
_I = ctypes.c_int
_D = ctypes.c_double

#: every symbol `include/trs_solver.h` declares -> (restype, argtypes)
SIGNATURES = {
    "trs_abi_version": (_I, []),
    "trs_slab_ld": (_I, [_I]),
    "trs_slab_rows": (_I, [_I]),
    "trs_dofmap": (_I, [_I, _I, _P, _P, _P, _P, _P]),
    "trs_env_ints": (_I, [_I]),
    "trs_assemble_work_bytes": (ctypes.c_size_t, [_I, _I, _I]),
    "trs_assemble": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P, _P, _P, _I, _P]),
    "trs_potrf_batched": (_I, [_I, _P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _P]),
    "trs_potrs_batched": (_I, [_I, _P, _I, _I, _P, _P, _I, _P, _I, _P]),
    "trs_recover": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P]),
    "trs_ga_sections": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "trs_fitness": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _D, _D, _P, _P, _P, _P]),
    "trs_solve_small_fits": (_I, [_I, _I, _I]),
    "trs_solve_small": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                             _D, _D, _P, _P, _P, _P]),
    "trs_graph_features_dev": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _D, _D, _D, _D,
                                    _I, _P, _P, _P, _P, _P, _P]),
    "trs_graph_features_packed": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _D, _D, _D, _D,
                                       _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "trs_joint_order_fits": (_I, [_I, _I]),
    "trs_joint_order": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "trs_joint_order_rows": (_I, [_I, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                  _P, _P, _I, _P]),
    "trs_recover_rows": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P,
                              _I, _P]),
    "trs_cubegen_dev": (_I, [_I, ctypes.c_uint64, _I, _I, _I, _P, _I, _I, _I, _D, _D, _P, _I, _I, _P, _I, _I, _I,
                             _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_int64, _P]),
    "trs_copy_rows": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _P]),
    "trs_stream_create_masked": (_I, [_P, _I, _P]),
    "trs_stream_destroy": (_I, [_P]),
    "trs_solve_rows": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I,
                            _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P]),
    "trs_solve": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I,
                       _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    # table member form (ABI 10): (conn16, type_idx, types) in place of (conn, E, A)
    "trs_assemble_tab": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P, _P, _P, _I, _P]),
    "trs_recover_tab": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P]),
    "trs_recover_rows_tab": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P,
                                  _I, _P]),
    "trs_solve_small_tab": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                 _D, _D, _P, _P, _P, _P]),
    "trs_joint_order_tab": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "trs_joint_order_rows_tab": (_I, [_I, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                      _P, _P, _I, _P]),
    "trs_solve_tab": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I,
                           _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "trs_solve_rows_tab": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I,
                                _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P]),
}

#: must equal TRS_ABI_VERSION of include/trs_solver.h
ABI_VERSION = 10

_lib = None


def build(verbose=False):
    """Compile the HIP sources for gfx950 into `libtrs_hip.so` (hipcc cross-compiles without a GPU)."""
    import subprocess
    cmd = ["make", "-C", CSRC_DIR, "-j4"]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or proc.returncode != 0:
        print(proc.stdout + proc.stderr)
    if proc.returncode != 0:
        raise HipExtensionError("building libtrs_hip.so failed:\n" + proc.stderr[-2000:])
    return LIB_PATH


def load():
    """Load the library once and attach the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C python_stable_3d_truss_analysis_amd/csrc`). There is no CPU fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:
        raise HipExtensionError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.trs_abi_version() != ABI_VERSION:
        raise HipExtensionError("libtrs_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise HipExtensionError(f"{what} failed with hipError_t {rc}")
