#!/usr/bin/env python3
"""Host <-> device copy rates on this box: pageable vs pinned, by transfer size, one stream."""
import time
import torch
dev = torch.device("cuda:0")
for mb in (1, 16, 64, 256):
    n = mb << 20
    d = torch.empty([n], dtype=torch.uint8, device=dev)
    for kind in ("pageable", "pinned"):
        h = torch.empty([n], dtype=torch.uint8)
        if kind == "pinned":
            h = h.pin_memory()
        for direction in ("h2d", "d2h"):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                if direction == "h2d":
                    d.copy_(h, non_blocking=True)
                else:
                    h.copy_(d, non_blocking=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            print(f"{mb:4d} MiB {kind:8s} {direction}: {n / dt / 1e9:6.1f} GB/s")
