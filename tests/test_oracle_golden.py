"""Pins the CPU oracle (oracle/truss_oracle.py) against
 (a) the reference's own stored answers (data/*_output_*.json, generate/cube-7_case_*.json),
 (b) dense outputs captured from the imported reference (tests/golden/make_golden.py).
CPU only."""
import numpy as np
import pytest

from oracle import truss_oracle as orc
from tests import helpers as H

# Tolerances: the oracle repeats the reference's operations in the same order, so it agrees
# to rounding with the captured dense vectors; the stored JSON answers were produced on other
# hardware/BLAS and agree to 1e-9 relative (bar-942: cond(K_ff) = 6e6).
TOL_CAPTURED = 1e-12
TOL_STORED = 1e-9


@pytest.mark.parametrize("name", H.data_case_names())
def test_oracle_matches_captured_dense_reference(name):
    data = H.load_json(name)
    z = H.dense_golden()
    res = orc.solve(data)
    assert H.max_scaled_err(res["u"], z[f"{name}/u"]) <= TOL_CAPTURED
    assert H.max_scaled_err(res["f_ext"], z[f"{name}/f_ext"]) <= TOL_CAPTURED
    assert H.max_scaled_err(res["N"], z[f"{name}/N"]) <= TOL_CAPTURED
    assert res["weight"] == pytest.approx(float(z[f"{name}/weight"]), rel=1e-15)
    if f"{name}/K_ff" in z.files:
        assert H.max_scaled_err(res["K_ff"], z[f"{name}/K_ff"]) <= 1e-15


@pytest.mark.parametrize("name", H.data_case_names())
def test_oracle_matches_reference_stored_outputs(name):
    data = H.load_json(name)
    stored = H.load_json(name.replace("_input_", "_output_"))
    dim, nJ, nM = orc.truss_dim(data), len(data["joint"]), len(data["member"])
    res = orc.solve(data)
    assert H.max_scaled_err(res["u"], orc.densify(stored["displace"], nJ, dim)) <= TOL_STORED
    assert H.max_scaled_err(res["f_ext"], orc.densify(stored["external"], nJ, dim)) <= TOL_STORED
    assert H.max_scaled_err(res["N"], orc.densify(stored["internal"], nM)) <= TOL_STORED
    assert res["weight"] == pytest.approx(stored["weight"], rel=1e-13)


@pytest.mark.parametrize("name", H.cube7_case_names())
def test_oracle_matches_reference_cube7_files(name):
    stored = H.load_json(name)
    nJ, nM = len(stored["joint"]), len(stored["member"])
    res = orc.solve(stored)
    assert H.max_scaled_err(res["u"], orc.densify(stored["displace"], nJ, 3)) <= TOL_STORED
    assert H.max_scaled_err(res["f_ext"], orc.densify(stored["external"], nJ, 3)) <= TOL_STORED
    assert H.max_scaled_err(res["N"], orc.densify(stored["internal"], nM)) <= TOL_STORED


def test_oracle_matches_ragged_cube_fixtures():
    for name, data, gold in H.ragged_cube_cases():
        if len(data["member"]) > 1200:   # keep the CPU suite short; big ones are GPU parity cases
            continue
        res = orc.solve(data)
        for key in ("u", "f_ext", "N"):
            assert H.max_scaled_err(res[key], gold[key]) <= 1e-11, (name, key)


def test_oracle_edge_cases():
    for name, entry in H.edge_cases().items():
        data = entry["input"]
        if entry.get("raises") == "TrussNotStableError":
            with pytest.raises(orc.OracleNotStable):
                orc.solve(data)
        elif entry.get("raises") == "LinAlgError":
            with pytest.raises(np.linalg.LinAlgError):
                orc.solve(data)
        else:
            res = orc.solve(data)
            for key in ("u", "f_ext", "N"):
                assert H.max_scaled_err(res[key], entry[key]) <= TOL_CAPTURED, (name, key)
            assert res["weight"] == pytest.approx(entry["weight"], rel=1e-15)
            # sparse views: same key sets and values as the genuine reference dicts
            u, f, n = orc.sparsify(res)
            assert sorted(u) == [j for j, _ in entry["sparse"]["displace"]]
            assert sorted(f) == [j for j, _ in entry["sparse"]["external"]]
            assert sorted(n) == [m for m, _ in entry["sparse"]["internal"]]
