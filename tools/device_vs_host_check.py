import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from python_stable_3d_truss_analysis_amd import batch, generate as gen
from python_stable_3d_truss_analysis_amd.type import GenerateMethod, LinkType
rng = np.random.default_rng(11)
ok = True
for k, (grid, lo, hi, kw) in enumerate([((6, 6, 6), 8, 190, {}), ((5, 5, 5), 1, 125, dict(method=GenerateMethod.DFS)),
                                       ((8, 8, 8), 50, 300, dict(method=GenerateMethod.BFS, linkType=LinkType.Cross)),
                                       ((4, 9, 3), 5, 100, dict(isAllowParallel=True))]):
    num = rng.integers(lo, hi + 1, size=32768)
    host = gen.generate_cube_batch(num, gridRange=grid, seed=100 + k, **kw)
    sizes, t = gen.generate_cube_batch_device(num, gridRange=grid, seed=100 + k, **kw)
    same = all(np.array_equal(t[f].cpu().numpy(), getattr(host, f)) for f in ("xyz", "conn", "E", "A", "rho", "cbits", "loads", "nJ", "nM"))
    plan = batch.order_plan(True, host.nJ_max, host.nM_max)
    line = f"grid {grid} {kw}: generator equal {same}, nJ_max {host.nJ_max}, nM_max {host.nM_max}, plan {plan}"
    if plan[0] == "device":
        a = batch.joint_order_device(torch, t, want_choice=True); torch.cuda.synchronize()
        b = batch.joint_order_device(torch, t, want_choice=True); torch.cuda.synchronize()
        perm = batch.profile_permutation(host)
        eq_host = bool((a["perm"].cpu().numpy() == perm).all())
        determ = all(torch.equal(a[x], b[x]) for x in a)
        line += f", order = host {eq_host}, deterministic {determ}"
        ok &= eq_host and determ
    ok &= same
    print(line)
print("ALL OK" if ok else "MISMATCH")
