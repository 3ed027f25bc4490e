"""Native bulk JSON reader (csrc/jsonpack.c; reference Truss.LoadFromJSON, truss.py:401-421) against the
Python packer on every bundled file, its edge semantics, and its error reporting.  CPU only."""
import glob
import json
import os

import numpy as np
import pytest

from python_stable_3d_truss_analysis_amd import Truss, batch
from python_stable_3d_truss_analysis_amd import generate as gen
from tests import helpers as H


def _same(a, b):
    for f in a.__dataclass_fields__:
        np.testing.assert_array_equal(getattr(a, f), getattr(b, f), err_msg=f)


def test_every_bundled_file_packs_like_the_python_path():
    paths = sorted(glob.glob(os.path.join(H.GOLDEN, "data", "*.json")))
    assert len(paths) >= 30                                     # 2D and 3D inputs, output files, cube-7 cases
    texts = [open(p, "rb").read() for p in paths]
    want = batch.pack_json([json.loads(t) for t in texts])
    _same(batch.pack_json_texts(texts), want)
    _same(batch.pack_json_files(paths), want)
    _same(batch.pack_json_texts([t.decode() for t in texts[:3]]), batch.pack_json([json.loads(t) for t in texts[:3]]))
    assert sorted(set(want.dim.tolist())) == [2, 3]
    # ... and like the reference-compatible object path
    trusses = [Truss(int(d)).LoadFromJSON(p) for p, d in zip(paths[:6], want.dim[:6])]
    _same(batch.pack_json_files(paths[:6]), batch.pack_trusses(trusses))


def test_generated_trusses_roundtrip_bit_exactly():
    p = gen.generate_cube_batch([3, 7, 20, 60], gridRange=(6, 6, 6), seed=4,
                                memberTypes=[[1.25, 2.05e7, 0.3], [3.0, 1e7, 0.1]])
    docs = [json.dumps(gen.packed_to_json(p, b)) for b in range(p.B)]
    _same(batch.pack_json_texts(docs), p)                       # repr(float) -> strtod is the identity


def test_reference_semantics_of_the_loader():
    doc = {"force": [[1, [0.0, 1e-12, 0.0]], [2, [5.0, 0.0, -1.5]], [2, [7.0, 0.0, 0.0]], [1, [0.0, 0.0, 0.0]]],
           "weight": 12.5, "displace": [[1, [0.1, 0.2, 0.3]]], "internal": [[0, 3.5]], "external": [],
           "member": [[[0, 1], [1, 1e7, 0.1]], [[1, 2], [2.5, 2e7, 0.2]], [[0, 2], [1.0, 1E+7, 1e-1]]],
           "joint": [[[0, 0, 0], "PIN"], [[1.5, 0, 0], "ROLLER_Z"], [[0, 2.25, -1e-3], "NO"]]}
    text = json.dumps(doc, indent=2)                            # keys out of order, whitespace, result keys
    got = batch.pack_json_texts([text])
    _same(got, batch.pack_json([doc]))
    assert got.loads[0, 1].tolist() == [0.0, 0.0, 0.0]          # a zero load is dropped (truss.py:181-182)
    assert got.loads[0, 2].tolist() == [7.0, 0.0, 0.0]          # a later load replaces an earlier one
    assert got.cbits[0].tolist() == [7, 4, 0] and got.n_free[0] == 5
    two_d = {"joint": [[[0, 0], "PIN"], [[1, 0], "ROLLER_Y"], [[0.5, 1], "NO"]], "force": [[2, [1, -2]]],
             "member": [[[0, 1], [1, 1, 1]], [[1, 2], [1, 1, 1]], [[0, 2], [1, 1, 1]]]}
    got = batch.pack_json_texts([json.dumps(two_d), text])      # mixed dimensions in one batch
    _same(got, batch.pack_json([two_d, doc]))
    assert got.dim.tolist() == [2, 3] and got.cbits[0].tolist() == [7, 6, 4]
    empty = batch.pack_json_texts(['{"joint": [], "force": [], "member": []}'])
    assert empty.nJ.tolist() == [0] and empty.nM.tolist() == [0]


@pytest.mark.parametrize("doc,what", [
    ('{"joint": [[[0,0,0], "PIN"]', "syntax"),
    ('{"joint": [[[0,0,0], "WELD"]], "force": [], "member": []}', "support"),
    ('{"joint": [[[0,0], "ROLLER_Z"]], "force": [], "member": []}', "support"),
    ('{"joint": [[[0,0,0], "PIN"], [[1,0], "NO"]], "force": [], "member": []}', "dimension"),
    ('{"joint": [[[0,0,0], "PIN"]], "force": [[3, [1,0,0]]], "member": []}', "joint id"),
    ('{"joint": [[[0,0,0], "PIN"], [[1,0,0], "NO"]], "force": [], "member": [[[0, 5], [1,1,1]]]}', "joint id"),
])
def test_errors_name_the_document(doc, what):
    good = json.dumps(H.load_json("bar-6_input_0"))
    with pytest.raises(ValueError, match=r"#1: .*" + what):
        batch.pack_json_texts([good, doc, good])


def test_unreadable_file_is_reported(tmp_path):
    path = tmp_path / "a.json"
    path.write_text(json.dumps(H.load_json("bar-6_input_0")))
    with pytest.raises(ValueError, match="#1: file cannot be read"):
        batch.pack_json_files([str(path), str(tmp_path / "missing.json")])


def test_numbers_are_parsed_exactly_like_python_including_rounding_boundaries():
    """Every float text -> the same double as Python's float(): short decimals (first fast path), 16-19
    digit texts as repr() prints them (extended-precision path with its guard band), texts within a hair
    of a rounding boundary, and magnitudes outside either fast path (strtod)."""
    from fractions import Fraction
    rng = np.random.default_rng(11)
    texts_of = []
    bits = rng.integers(0x0010000000000000, 0x7fe0000000000000, size=6000, dtype=np.uint64)
    texts_of += [repr(float(v)) for v in bits.view(np.float64)]                      # the whole normal range
    texts_of += [repr(float(v)) for v in rng.uniform(-500, 500, 20000)]              # what a geometry file holds
    texts_of += [repr(float(v)) for v in rng.normal(0, 1, 6000) * 10.0 ** rng.integers(-25, 25, 6000)]
    texts_of += ["0.1", "62.5", "1e7", "-0.0", "123456789012345678", "1e-320", "1.7976931348623157e308", "5e-324"]
    for a in rng.uniform(1, 10, 4000):                                               # 19 digits around a midpoint
        f = float(a)
        mid = (Fraction(f) + Fraction(float(np.nextafter(f, 100)))) / 2
        d = str(int(mid * 10 ** 18))
        texts_of += [d[0] + "." + d[1:], d[0] + "." + d[1:-1] + str((int(d[-1]) + 1) % 10)]
    n = len(texts_of) // 500 * 500
    docs = ['{"joint":[%s],"force":[],"member":[]}' % ",".join('[[%s,0.0,0.0],"PIN"]' % t for t in texts_of[i:i + 500])
            for i in range(0, n, 500)]
    got = batch.pack_json_texts(docs).xyz[:, :, 0].ravel()
    want = np.array([float(t) for t in texts_of[:n]])
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))

