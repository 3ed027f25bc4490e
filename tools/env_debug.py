import sys, json, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from python_stable_3d_truss_analysis_amd import batch
data = json.load(open('tests/golden/data/bar-942_input_0.json'))
dev = batch.DeviceBatch(batch.pack_json([data]))
dev.S.zero_()
dev.dofmap(); dev.assemble()
S0 = dev.S.clone(); env = dev.env[0].cpu().numpy()
nch = 44; ft = env[:nch]; last = env[nch:nch+11]; slack = int(env[nch+11]); cend = env[nch+11+8:nch+11+8+nch]
hits = []
for t in range(12):
    for q in range(t, nch + 1):           # nch = the rhs chunk columns 704..719
        if q < nch and q < cend[t]: continue   # written region
        if q == nch: continue
        dev.S.copy_(S0)
        dev.S[0, 16*t:16*t+16, 16*q:16*q+16] = float('nan')
        dev.potrf(); torch.cuda.synchronize()
        if int(dev.info[0]) != 0 or bool(torch.isnan(dev.S[0, :704, 704]).any()):
            hits.append((t, q, int(dev.info[0])))
print("unwritten tiles whose NaN reaches the result (row-tile, chunk, info):", hits[:20])
print("ft", ft[:12].tolist(), "last", last[:3].tolist())
