#!/usr/bin/env python3
"""profiles/potrf_traffic.json from a PMC summary (tools/summarize_pmc.py output), stamped with the
fingerprint of the kernel sources it was measured on (bench.kernel_source_sha): bench.py quotes the
record as `roofline.traffic` only while the sources are unchanged.

    python tools/make_traffic_record.py profiles/r02_xxx_pmc_summary.json <kernel name> [batch] [joint order of the profiled run: profile | rcm | given]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main(summary_path, kernel, batch=4096, joint_order="profile"):
    with open(summary_path) as fh:
        rec = json.load(fh)[kernel]
    pmc = rec["pmc_per_launch"]
    out = {"kernel": kernel, "workload": f"bar-942 x {batch}", "envelope": True, "batch": int(batch), "joint_order": joint_order,
           "hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "fetch_size_kib": pmc["FETCH_SIZE"],
           "write_size_kib": pmc["WRITE_SIZE"], "avg_ns": rec.get("avg_ns"),
           "source_sha": bench.kernel_source_sha(),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_pmc.sh); hbm = "
                   "2*FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950 half-count of wide reads, MI355X_MICROARCH.md "
                   "section HBM; calibrated for this access pattern in profiles/r01_fetch_size_calibration.txt)",
           "source": os.path.relpath(summary_path, ROOT)}
    with open(os.path.join(ROOT, "profiles", "potrf_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
