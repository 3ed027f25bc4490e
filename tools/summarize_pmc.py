#!/usr/bin/env python3
"""Summarise the CSVs written by tools/profile_pmc.sh into one small JSON per tag.

    python tools/summarize_pmc.py gpurun_out/<tag> profiles/<name>.json

Per kernel: average duration from the kernel trace, and per-launch averages of every PMC counter.
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream, so the
read side is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]   # template arguments stay: trs_potrf_narrow_kernel<false> / <true>


def main(src, dst):
    out = defaultdict(dict)
    for path in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            out[short(row["Name"])].update(calls=int(row["Calls"]), avg_ns=float(row["AverageNs"]),
                                           pct=float(row["Percentage"]))
    for path in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        sums, counts = defaultdict(float), defaultdict(int)
        for row in csv.DictReader(open(path)):
            key = (short(row["Kernel_Name"]), row["Counter_Name"])
            sums[key] += float(row["Counter_Value"])
            counts[key] += 1
        for (kern, ctr), total in sums.items():
            out[kern].setdefault("pmc_per_launch", {})[ctr] = total / counts[(kern, ctr)]
    for kern, rec in out.items():
        pmc = rec.get("pmc_per_launch", {})
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            rec["hbm_bytes_per_launch"] = 2.0 * pmc["FETCH_SIZE"] * 1024 + pmc["WRITE_SIZE"] * 1024
            rec["hbm_bytes_note"] = "2*FETCH_SIZE KiB (gfx950 half-count correction) + WRITE_SIZE KiB"
        # matrix-core busy FRACTION: SQ_VALU_MFMA_BUSY_CYCLES is in shader cycles summed over the chip's 1024
        # SIMDs (256 CUs x 4; guide: "counts cycles"), GRBM_GUI_ACTIVE (another pass, same launches) is the
        # kernel's duration in cycles summed over the 8 XCDs -> busy / (1024 * GUI_ACTIVE / 8) lies in [0, 1]
        if pmc.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and pmc.get("GRBM_GUI_ACTIVE"):
            rec["mfma_busy_frac_per_simd"] = pmc["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * pmc["GRBM_GUI_ACTIVE"] / 8.0)
            rec["mfma_busy_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)"
        if pmc.get("SQ_WAIT_ANY") is not None and pmc.get("SQ_WAVE_CYCLES"):
            rec["wait_any_over_wave_cycles"] = pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
