#!/usr/bin/env python3
"""The reference's own benchmark protocol (example.py:1-25, README.md:86-94): load a truss once, call
`truss.Solve()` 30 times, report the mean wall time - through the drop-in `Truss` of this package (every call:
pack, upload, kernels, download, sparse result dicts), for the seven bundled cases of BASELINE.md section 1."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from python_stable_3d_truss_analysis_amd.truss import Truss

PUBLISHED = {"bar-6": 0.00037, "bar-10": 0.00050, "bar-25": 0.00126, "bar-47": 0.00253, "bar-72": 0.00323,
             "bar-120": 0.00557, "bar-942": 0.05253}          # README.md:88-94 (i7-10750H), seconds
SURVEY = {"bar-6": 0.00044, "bar-10": 0.00060, "bar-25": 0.00147, "bar-47": 0.00241, "bar-72": 0.00395,
          "bar-120": 0.00711, "bar-942": 0.06434}             # BASELINE.md section 2 (survey container), seconds
DIM = {"bar-10": 2, "bar-47": 2}


def measure(case, repeats=30):
    truss = Truss(dim=DIM.get(case, 3))
    truss.LoadFromJSON(os.path.join(ROOT, "tests", "golden", "data", f"{case}_input_0.json"))
    truss.Solve()                                   # first call: library load, first launches
    ts = []
    for _ in range(repeats):
        t0 = time.time()
        truss.Solve()
        ts.append(time.time() - t0)
    return sum(ts) / len(ts), min(ts)


if __name__ == "__main__":
    out = {}
    for case in PUBLISHED:
        mean, best = measure(case)
        out[case] = {"mean_s": mean, "min_s": best, "reference_published_s": PUBLISHED[case],
                     "reference_survey_container_s": SURVEY[case], "speedup_vs_published": PUBLISHED[case] / mean}
        print(f"{case:8s} mean of 30 {mean * 1e3:8.3f} ms  min {best * 1e3:8.3f} ms  reference {PUBLISHED[case] * 1e3:8.3f} ms "
              f"(published) {SURVEY[case] * 1e3:8.3f} ms (survey container)  -> {PUBLISHED[case] / mean:6.1f} x")
    print(json.dumps(out))
