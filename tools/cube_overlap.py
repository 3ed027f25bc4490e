#!/usr/bin/env python3
"""How much of the four-lane cube step runs with kernels of SEVERAL lanes on the chip at once: from a rocprofv3 kernel
trace (start / end timestamps of every dispatch) of tools/cube_step.py - the share of the traced steps' time with
0, 1, 2, ... kernels in flight, and per kernel name the time it ran alone / beside others.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/x -- python3 tools/cube_step.py --steps 3
    python3 tools/cube_overlap.py gpurun_out/x"""
import csv, glob, sys
from collections import defaultdict
rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(path)))
ev = []
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if not name.startswith("trs_"):
        continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "")))
ev.sort()
# the LAST burst of dispatches = the timed steps (the run's phases are separated by host synchronisations: gaps with
# nothing in flight for more than a millisecond)
cut, reach = 0, ev[0][1]
for k, (s0, e0, _, _) in enumerate(ev):
    if s0 - reach > 1_000_000:
        cut = k
    reach = max(reach, e0)
ev = ev[cut:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
points = sorted([(s, 1) for s, _, _, _ in ev] + [(e, -1) for _, e, _, _ in ev])
hist, level, last = defaultdict(int), 0, t0
for t, d in points:
    hist[level] += t - last
    last, level = t, level + d
total = float(t1 - t0)
print(f"{len(ev)} dispatches on {len(set(e[3] for e in ev))} queues over {total / 1e6:.2f} ms")
for k in sorted(hist):
    print(f"  {k} kernels in flight: {hist[k] / total:6.1%} of the time")
busy = sum(e - s for s, e, _, _ in ev)
print(f"sum of the kernels' durations / elapsed = {busy / total:.2f}")
