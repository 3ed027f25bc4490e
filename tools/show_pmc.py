#!/usr/bin/env python3
"""Print one line per kernel from a summary written by tools/summarize_pmc.py."""
import json
import sys

d = json.load(open(sys.argv[1]))
for k, v in sorted(d.items()):
    if not k.startswith("trs"):
        continue
    p = v.get("pmc_per_launch", {})
    ns = max(v.get("avg_ns", 1), 1)
    print(f"{k:26s} ms {ns / 1e6:7.3f} hbm_GB {v.get('hbm_bytes_per_launch', 0) / 1e9:7.2f} "
          f"wr_GB {p.get('WRITE_SIZE', 0) * 1024 / 1e9:6.2f} TB/s {v.get('hbm_bytes_per_launch', 0) / ns / 1e3:5.2f} "
          f"valu {p.get('SQ_INSTS_VALU', 0) / 1e6:7.1f}M lds {p.get('SQ_INSTS_LDS', 0) / 1e6:6.1f}M "
          f"mfma {p.get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0) / 4e6:6.1f}M clk {p.get('GRBM_GUI_ACTIVE', 0) / 8 / ns:4.2f}")
