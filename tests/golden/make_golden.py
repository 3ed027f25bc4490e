"""Regenerate the golden fixtures in this directory by IMPORTING the real reference.

Runs only in the build container (needs /root/reference); nothing here is used at test
time.  The reference's Python never travels: only inputs and expected outputs (data) are
written.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

In-process shims (none touches arithmetic):
  * `turtle` stub module            - reference utils.py:1 imports it, tkinter is absent
  * `np.bool8 = np.bool_`           - reference truss.py:321, removed in numpy 2
  * `plt.style.use("seaborn")`      - reference plot.py:9, renamed in matplotlib >= 3.6
Dense (un-sparsified) results are obtained by running the reference's own `Truss.Solve()`
with its `IsZero` / `IsZeroVector` names rebound to "never zero" inside the reference's
truss module, so every joint and member lands in the result dicts.

Written files:
  data/*.json            verbatim copies of the reference's data/ and generate/ JSON files
  dense_data.npz         dense u / f_ext / N / weight (+ K_ff up to 120 bars) per data case
  cube_ragged.npz        12 seeded GenerateRandomCubeTrusses cases (inputs + dense outputs)
  edge_cases.json        synthetic edge cases (inputs + dense outputs or expected exception)
  ga_trace.json          seeded GA run on bar-120 (generation-0 fitness triples, history)
  hetero_bar25.npz       HeteroData tensors of bar-25 for the four (task, metapath) combinations
  augment.json           the reference's data augmenters (generate.py:13-148) applied to bar-25 with
                         seeded `random`: expected joint lists per augmenter
  cubegrid.json          CubeGrid / CubeTruss (generate.py:150-311) under seeded `random`: cube joint ids, truss dicts
  cube_stats.npz         integer statistics of 200 reference samples per (GenerateMethod, LinkType, polycube
                         size): the distribution the native generator (csrc/cubegen.c) must reproduce
"""
import glob
import json
import os
import random
import shutil
import sys
import types

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    turtle = types.ModuleType("turtle")
    turtle.position = None
    sys.modules.setdefault("turtle", turtle)
    if not hasattr(np, "bool8"):
        np.bool8 = np.bool_
    import matplotlib.pyplot as plt
    real_use = plt.style.use
    plt.style.use = lambda name: real_use("seaborn-v0_8" if name == "seaborn" else name)
    sys.path.insert(0, REF)
    import slientruss3d.truss as rt
    import slientruss3d.type as rty
    import slientruss3d.generate as rg
    import slientruss3d.ga as rga
    return rt, rty, rg, rga


def dense_solve(rt, truss):
    """Run the reference Solve() with sparsification disabled -> dense arrays."""
    keep_zero, keep_vec = rt.IsZero, rt.IsZeroVector
    rt.IsZero = lambda *a, **k: False
    rt.IsZeroVector = lambda *a, **k: False
    try:
        truss.Solve()
    finally:
        rt.IsZero, rt.IsZeroVector = keep_zero, keep_vec
    dim, nJ, nM = truss.dim, truss.nJoint, truss.nMember
    u = np.array([truss.GetDisplacements()[j] for j in range(nJ)]).reshape(nJ, dim)
    f = np.array([truss.GetExternalForces()[j] for j in range(nJ)]).reshape(nJ, dim)
    n = np.array([truss.GetInternalForces()[m] for m in range(nM)])
    return u, f, n


def capture_data_cases(rt):
    os.makedirs(os.path.join(HERE, "data"), exist_ok=True)
    out = {}
    for path in sorted(glob.glob(os.path.join(REF, "data", "*.json"))
                       + glob.glob(os.path.join(REF, "generate", "*.json"))):
        shutil.copyfile(path, os.path.join(HERE, "data", os.path.basename(path)))
        os.chmod(os.path.join(HERE, "data", os.path.basename(path)), 0o644)
    for path in sorted(glob.glob(os.path.join(REF, "data", "*_input_*.json"))):
        name = os.path.basename(path)[:-5]
        with open(path) as fh:
            data = json.load(fh)
        dim = len(data["joint"][0][0])
        truss = rt.Truss(dim).LoadFromJSON(path)
        u, f, n = dense_solve(rt, truss)
        out[f"{name}/u"], out[f"{name}/f_ext"], out[f"{name}/N"] = u, f, n
        out[f"{name}/weight"] = np.array(truss.weight)
        if truss.nMember <= 120:
            mask = truss.GetDisplacementUnknownMask()
            out[f"{name}/K_ff"] = truss.GetKMatrix()[mask, :][:, mask]
    np.savez_compressed(os.path.join(HERE, "dense_data.npz"), **out)
    print("dense_data.npz:", len(out), "arrays")


SUPPORT_CODE = {"NO": 0, "PIN": 1, "ROLLER_X": 2, "ROLLER_Y": 3, "ROLLER_Z": 4}


def capture_cube_ragged(rt, rty, rg):
    out = {}
    cube_counts = [8, 12, 20, 30, 45, 60, 80, 100, 130, 160, 190, 216]
    for idx, num in enumerate(cube_counts):
        trusses = rg.GenerateRandomCubeTrusses(
            gridRange=(6, 6, 6), numCubeRange=(num, num), numEachRange=(1, 1),
            lengthRange=(50, 150), isDoStructuralAnalysis=False, isPrintMessage=False,
            seed=1000 + idx)
        truss = trusses[0]
        data = truss.Serialize()
        u, f, n = dense_solve(rt, truss)
        key = f"cube{idx:02d}"
        out[f"{key}/xyz"] = np.array([p for p, _ in data["joint"]], dtype=np.float64)
        out[f"{key}/support"] = np.array([SUPPORT_CODE[s] for _, s in data["joint"]], dtype=np.uint8)
        loads = np.zeros([len(data["joint"]), 3])
        for j, v in data["force"]:
            loads[j] = v
        out[f"{key}/loads"] = loads
        out[f"{key}/conn"] = np.array([c for c, _ in data["member"]], dtype=np.int32)
        out[f"{key}/mtype"] = np.array([t for _, t in data["member"]], dtype=np.float64)
        out[f"{key}/u"], out[f"{key}/f_ext"], out[f"{key}/N"] = u, f, n
        print(f"  {key}: numCube {num} nJ {truss.nJoint} nM {truss.nMember}")
    np.savez_compressed(os.path.join(HERE, "cube_ragged.npz"), **out)


def edge_case_inputs():
    mt = [1.0, 1e7, 0.1]
    cases = {}
    # 2D: PIN + ROLLER_Y (roller constrains y only), triangle with a load.
    cases["2d_pin_rollerY"] = {
        "joint": [[[0, 0], "PIN"], [[4, 0], "ROLLER_Y"], [[2, 3], "NO"]],
        "force": [[2, [1000.0, -2500.0]]],
        "member": [[[0, 1], mt], [[0, 2], mt], [[1, 2], [2.0, 2e7, 0.3]]]}
    # 2D: ROLLER_X on one support and a load on a supported joint (ignored at its
    # constrained DOF, overwritten in `external`).
    cases["2d_rollerX_load_on_support"] = {
        "joint": [[[0, 0], "PIN"], [[5, 0], "ROLLER_Y"], [[5, 4], "NO"], [[0, 4], "ROLLER_X"]],
        "force": [[2, [300.0, -700.0]], [1, [150.0, 999.0]], [3, [123.0, -50.0]]],
        "member": [[[0, 1], mt], [[1, 2], mt], [[2, 3], mt], [[3, 0], mt], [[0, 2], mt], [[1, 3], mt]]}
    # 3D: all three roller kinds + one pin, tetrahedron-like frame.
    cases["3d_rollers_xyz"] = {
        "joint": [[[0, 0, 0], "PIN"], [[3, 0, 0], "ROLLER_Z"], [[0, 3, 0], "ROLLER_Z"],
                  [[1, 1, 4], "NO"], [[3, 3, 0], "ROLLER_X"], [[1.5, 1.5, 0], "ROLLER_Y"]],
        "force": [[3, [500.0, -300.0, -2000.0]], [1, [10.0, 20.0, 30.0]]],
        "member": [[[0, 1], mt], [[0, 2], mt], [[1, 2], mt], [[0, 3], mt], [[1, 3], mt], [[2, 3], mt],
                   [[4, 3], mt], [[4, 1], mt], [[4, 2], mt], [[5, 3], mt], [[5, 0], mt], [[5, 4], mt],
                   [[5, 1], mt]]}
    # 3D: parallel (duplicated) members, different sections, both orientations.
    cases["3d_parallel_members"] = {
        "joint": [[[0, 0, 0], "PIN"], [[2, 0, 0], "PIN"], [[0, 2, 0], "PIN"], [[1, 1, 3], "NO"],
                  [[1, 1, 6], "NO"]],
        "force": [[3, [100.0, 200.0, -300.0]], [4, [-50.0, 75.0, -500.0]]],
        "member": [[[0, 3], mt], [[1, 3], mt], [[2, 3], mt], [[3, 0], [2.0, 1e7, 0.2]],
                   [[0, 3], [0.5, 3e7, 0.2]], [[3, 4], mt], [[4, 3], mt], [[0, 4], mt],
                   [[1, 4], mt], [[2, 4], mt]]}
    # 3D mechanism that passes the counting test: K_ff exactly singular -> LinAlgError.
    cases["3d_mechanism_singular"] = {
        "joint": [[[0, 0, 0], "PIN"], [[1, 0, 0], "PIN"], [[0, 1, 0], "NO"], [[5, 5, 5], "NO"]],
        "force": [[2, [0.0, 0.0, -10.0]]],
        "member": [[[0, 2], mt], [[1, 2], mt], [[0, 1], mt], [[0, 1], mt], [[0, 1], mt], [[0, 1], mt]]}
    # Counting test fails -> TrussNotStableError before arithmetic.
    cases["3d_count_unstable"] = {
        "joint": [[[0, 0, 0], "PIN"], [[1, 0, 0], "NO"], [[0, 1, 0], "NO"]],
        "force": [[1, [1.0, 0.0, 0.0]]],
        "member": [[[0, 1], mt], [[0, 2], mt]]}
    return cases


def capture_edge_cases(rt):
    out = {}
    for name, data in edge_case_inputs().items():
        dim = len(data["joint"][0][0])
        entry = {"input": data}
        try:
            truss = rt.Truss(dim).LoadFromJSON(data=data)
            u, f, n = dense_solve(rt, truss)
            entry.update(u=u.tolist(), f_ext=f.tolist(), N=n.tolist(), weight=truss.weight,
                         resist={str(k): v.tolist() for k, v in truss.GetResistances().items()})
            # also the genuine sparse views, for dict-semantics tests
            truss2 = rt.Truss(dim).LoadFromJSON(data=data)
            truss2.Solve()
            entry["sparse"] = {k: truss2.Serialize()[k] for k in ("displace", "external", "internal")}
        except Exception as exc:  # noqa: BLE001 - the exception type IS the golden value
            entry["raises"] = type(exc).__name__
        out[name] = entry
        print(f"  {name}: {'raises ' + entry['raises'] if 'raises' in entry else 'ok'}")
    with open(os.path.join(HERE, "edge_cases.json"), "w") as fh:
        json.dump(out, fh)


def capture_ga_trace(rt, rty, rga):
    """Seeded GA on bar-120 with the 20 member types of example.py:186."""
    random.seed(0)
    types_ = [rty.MemberType(inch, random.uniform(1e7, 3e7), random.uniform(0.1, 1.0))
              for inch in range(1, 21)]
    truss = rt.Truss(3).LoadFromJSON(os.path.join(REF, "data", "bar-120_input_0.json"))
    ga = rga.GA(truss, types_, 30000., 10., nIteration=5, nPatience=50, nPop=64, nElite=16)
    state = random.getstate()
    pop0 = ga.Initialize()
    gen0 = [list(ga.GetFitness(g)) for g in pop0]
    random.setstate(state)
    minGene, minInfo, pop, history = ga.Evolve(isPrintMessage=False)
    out = {"seed": 0, "memberTypes": [t.Serialize() for t in types_],
           "nPop": 64, "nElite": 16, "nIteration": 5, "allowStress": 30000., "allowDisplace": 10.,
           "pop0": pop0, "gen0": gen0, "bestFitnessHistory": history,
           "minGene": minGene, "minInfo": list(minInfo), "finalPop": pop}
    with open(os.path.join(HERE, "ga_trace.json"), "w") as fh:
        json.dump(out, fh)
    print("  ga history:", history)


def capture_hetero(rt, rty):
    """HeteroData tensors of bar-25 under the four (task, metapath) combinations (data.py:11-282).
    torch_geometric is not installed: a dict-of-stores stand-in for HeteroData lets data.py import."""
    import torch

    class _Store(dict):
        __getattr__ = dict.get
        def __setattr__(self, k, v):
            self[k] = v

    class HeteroData(dict):
        def __getitem__(self, key):
            if key not in self:
                dict.__setitem__(self, key, _Store())
            return dict.__getitem__(self, key)

    tg, tgd = types.ModuleType("torch_geometric"), types.ModuleType("torch_geometric.data")
    tgd.HeteroData = HeteroData
    tg.data = tgd
    sys.modules["torch_geometric"], sys.modules["torch_geometric.data"] = tg, tgd
    import slientruss3d.data as rd
    out = {}
    path = os.path.join(REF, "data", "bar-25_input_0.json")
    for task_name, task in (("opt", rty.TaskType.OPTIMIZATION), ("reg", rty.TaskType.REGRESSION)):
        for meta_name, meta in (("noimp", rty.MetapathType.NO_IMPLICIT), ("imp", rty.MetapathType.USE_IMPLICIT)):
            creator = rd.TrussHeteroDataCreator(meta, task)
            g = creator.FromJSON(path, 3, forceScale=1000., displaceScale=0.1, positionScale=100.)
            key = f"{task_name}_{meta_name}"
            for store_key, store in g.items():
                if not isinstance(store, dict):
                    continue
                name = store_key if isinstance(store_key, str) else "__".join(store_key)
                for field, value in store.items():
                    if torch.is_tensor(value):
                        out[f"{key}/{name}/{field}"] = value.numpy()
            out[f"{key}/originWeight"] = np.array(g["originWeight"])
    np.savez_compressed(os.path.join(HERE, "hetero_bar25.npz"), **out)
    print("  hetero:", len(out), "arrays;", {k: v.shape for k, v in out.items() if k.startswith("reg_imp")})


def capture_augmenters(rg):
    """Seeded runs of every augmenter of the reference on the bar-25 JSON dict (the form in which
    GenerateRandomCubeTrusses applies them, generate.py:357)."""
    with open(os.path.join(REF, "data", "bar-25_input_0.json")) as fh:
        base = json.load(fh)
    cases = {
        "NoChange": (lambda: rg.NoChange(), 1),
        "AddJointNoise": (lambda: rg.AddJointNoise([0.5, -1.0, 0.0], [0.1, 2.0, 0.5]), 7),
        "MoveToCentroid": (lambda: rg.MoveToCentroid(), 1),
        "Translation": (lambda: rg.Translation([1.5, -2.0, 10.0]), 1),
        "RandomTranslation": (lambda: rg.RandomTranslation([-3.0, 8.0]), 11),
        "RandomResetPin": (lambda: rg.RandomResetPin(3, 0.7), 13),
        "RandomResetPin_default": (lambda: rg.RandomResetPin(), 17),
        "List": (lambda: rg.TrussDataAugmenterList(rg.MoveToCentroid(), rg.AddJointNoise(),
                                                   rg.RandomResetPin(4, None), rg.RandomTranslation()), 19),
    }
    out = {}
    for name, (make, seed) in cases.items():
        random.seed(seed)
        res = make()(json.loads(json.dumps(base)))
        out[name] = {"seed": seed, "joint": res["joint"], "member_unchanged": res["member"] == base["member"],
                     "force_unchanged": res["force"] == base["force"]}
    with open(os.path.join(HERE, "augment.json"), "w") as fh:
        json.dump(out, fh)
    print("augment.json:", sorted(out))


CUBE_STAT_NAMES = ("nJ", "nM", "nPin", "nLoad", "maxDegree", "sumDegree2", "extentX", "extentY", "extentZ",
                   "nDiagonal")
CUBE_STAT_SIZES = (8, 30, 100)
CUBE_STAT_SAMPLES = 200


def cube_statistics(data):
    """Integer statistics of one generated cube truss (JSON dict): sizes, supports, loads, the joint-degree
    distribution (maximum and sum of squares), the polycube's extent in cells along each axis (the growth
    method shows there: depth-first snakes vs breadth-first lumps) and the number of face diagonals (the link
    type shows there).  Lengths are divided out through the three cell edge lengths = the smallest gap
    between distinct coordinate values of an axis (a polycube is connected, so neighbouring planes are one
    cell apart)."""
    xyz = np.array([p for p, _ in data["joint"]], dtype=float)
    conn = np.array([c for c, _ in data["member"]], dtype=int)
    nJ, nM = len(xyz), len(conn)
    deg = np.bincount(conn.ravel(), minlength=nJ)
    cell = [float(np.diff(np.unique(xyz[:, a])).min()) for a in range(3)]
    grid = np.rint(xyz / np.array(cell)).astype(int)
    ext = grid.max(axis=0) - grid.min(axis=0)
    step = np.abs(grid[conn[:, 0]] - grid[conn[:, 1]]).sum(axis=1)   # 1 = cube edge, 2 = face diagonal
    assert set(np.unique(step)) <= {1, 2}
    return [nJ, nM, sum(1 for _, s in data["joint"] if s == "PIN"), len(data["force"]), int(deg.max()),
            int((deg.astype(np.int64) ** 2).sum()), int(ext[0]), int(ext[1]), int(ext[2]), int((step == 2).sum())]


def capture_cube_stats(rty, rg):
    """200 reference samples per GenerateMethod x LinkType x polycube size on the 6x6x6 grid
    (generate.py:186-231,266-286,314-376), reduced to integers (a few KB)."""
    out = {"names": np.array(CUBE_STAT_NAMES), "sizes": np.array(CUBE_STAT_SIZES)}
    methods = {"DFS": rty.GenerateMethod.DFS, "BFS": rty.GenerateMethod.BFS, "Random": rty.GenerateMethod.Random}
    links = {"LBRT": rty.LinkType.LeftBottom_RightTop, "RBLT": rty.LinkType.RightBottom_LeftTop,
             "Cross": rty.LinkType.Cross, "Random": rty.LinkType.Random}
    for mi, (mname, method) in enumerate(methods.items()):
        for li, (lname, link) in enumerate(links.items()):
            for num in CUBE_STAT_SIZES:
                trusses = rg.GenerateRandomCubeTrusses(
                    gridRange=(6, 6, 6), numCubeRange=(num, num), numEachRange=(1, CUBE_STAT_SAMPLES),
                    method=method, linkType=link, isPrintMessage=False, seed=5000 + 100 * mi + 10 * li + num)
                rows = [cube_statistics(t.Serialize()) for t in trusses]
                out[f"{mname}/{lname}/{num}"] = np.array(rows, dtype=np.int32)
            print(f"  cube stats {mname}/{lname}: nM mean", [float(out[f'{mname}/{lname}/{n}'][:, 1].mean())
                                                               for n in CUBE_STAT_SIZES])
    np.savez_compressed(os.path.join(HERE, "cube_stats.npz"), **out)


def capture_cube_grid(rty, rg):
    """The reference's object-level generator (CubeGrid / CubeTruss, generate.py:150-311) under seeded `random`:
    the cubes' joint ids and the truss dict of CubesToTruss, per growth method and link type."""
    out = []
    for method in (rty.GenerateMethod.DFS, rty.GenerateMethod.BFS, rty.GenerateMethod.Random):
        for link in (rty.LinkType.LeftBottom_RightTop, rty.LinkType.RightBottom_LeftTop, rty.LinkType.Cross,
                     rty.LinkType.Random):
            for parallel in (False, True):
                seed = 900 + 100 * method + 10 * link + int(parallel)
                random.seed(seed)
                grid = rg.CubeGrid(3, 4, 2)
                cubes = grid.RandomGenerateCubes(6, method)
                truss = grid.CubesToTruss(cubes, [10.0, 20.0, 30.0], True, parallel, link)
                nxt = grid.GetNextFeasibles((1, 1, 0))
                out.append({"seed": seed, "method": method, "link": link, "parallel": parallel,
                            "cubes": [list(c.jointIDs) for c in cubes], "joint": truss["joint"],
                            "member": truss["member"], "next": [list(c) for c in nxt],
                            "vertices": [list(v) for v in cubes[0].GetCubeVertices()],
                            "out_of_range": [grid.IsOutOfRange(c) for c in ((0, 0, 0), (3, 0, 0), (2, 3, 1), (0, -1, 0))]})
    random.seed(77)
    grid = rg.CubeGrid(2, 2, 2)
    cubes = grid.RandomGenerateCubes(None, rty.GenerateMethod.Random)      # numCube drawn by the grid itself
    out.append({"seed": 77, "auto": True, "cubes": [list(c.jointIDs) for c in cubes],
                "joint": grid.ProcessPinSupport(False, [1, 1, 1])})
    with open(os.path.join(HERE, "cubegrid.json"), "w") as fh:
        json.dump(out, fh)
    print("cubegrid.json:", len(out), "cases")


def main():
    rt, rty, rg, rga = import_reference()
    which = sys.argv[1:] or ["data", "cube", "edge", "ga", "hetero", "augment", "stats", "cubegrid"]
    if "stats" in which:
        capture_cube_stats(rty, rg)
    if "cubegrid" in which:
        capture_cube_grid(rty, rg)
    if "augment" in which:
        capture_augmenters(rg)
    if "data" in which:
        capture_data_cases(rt)
    if "cube" in which:
        capture_cube_ragged(rt, rty, rg)
    if "edge" in which:
        capture_edge_cases(rt)
    if "ga" in which:
        capture_ga_trace(rt, rty, rga)
    if "hetero" in which:
        capture_hetero(rt, rty)


if __name__ == "__main__":
    main()
