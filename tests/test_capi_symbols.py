"""The C-ABI shared library loads on a CPU-only box and exports every symbol that
include/trs_solver.h declares (no compute calls here)."""
import ctypes
import os
import re

from python_stable_3d_truss_analysis_amd import _capi
from tests.helpers import ROOT


def declared_symbols(header="trs_solver.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(trs_[a-z_0-9]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    if not os.path.exists(_capi.LIB_PATH):
        _capi.build()
    lib = ctypes.CDLL(_capi.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 10
    for name in names:
        assert hasattr(lib, name), f"{name} declared in trs_solver.h but not exported"
    assert sorted(_capi.SIGNATURES) == names        # the ctypes table covers the whole header


def test_host_side_helpers_of_the_abi():
    lib = _capi.load()
    assert lib.trs_abi_version() == _capi.ABI_VERSION == 9
    assert lib.trs_assemble_work_bytes(244, 942, 696) % 256 == 0
    assert lib.trs_slab_rows(696) == 704 and lib.trs_slab_ld(696) == 720
    assert lib.trs_slab_rows(64) == 64 and lib.trs_slab_rows(65) == 128 and lib.trs_slab_rows(0) == 64
    # argument validation happens before any launch: bad leading dimension is refused
    assert lib.trs_potrf_batched(1, None, 100, 64, None, None, None, None, None, 64, 0, None) != 0
    assert not hasattr(lib, "trs_set_option")        # ABI 7: no process-wide switches, flags per call
    assert lib.trs_env_ints(696) == 3 * (704 // 16) + 704 // 64 + 8   # ft, last, routing word + 7, cend, kmask (ABI 9)
    # ABI 8: masked streams refuse an empty mask (no compute unit) before touching the runtime
    import ctypes
    handle, empty = ctypes.c_void_p(), (ctypes.c_uint32 * 8)()
    assert lib.trs_stream_create_masked(empty, 8, ctypes.byref(handle)) != 0 and handle.value is None
    assert lib.trs_stream_create_masked(None, 0, None) != 0 and lib.trs_stream_destroy(None) != 0


def test_host_library_exports_every_symbol_of_its_header():
    """libtrs_host.so (native generator, RCM, joint permutation, graph features) against include/trs_host.h."""
    from python_stable_3d_truss_analysis_amd import generate
    lib = generate._load()
    names = declared_symbols("trs_host.h")
    assert names == ["trs_apply_joint_order", "trs_cubegen", "trs_cubegen_bounds", "trs_envelope_reach", "trs_ga_update_pop", "trs_graph_features", "trs_host_threads",
                     "trs_json_free_files", "trs_json_pack", "trs_json_read_files", "trs_profile_order", "trs_rcm_order"]
    for name in names:
        assert hasattr(lib, name), f"{name} declared in trs_host.h but not exported"
