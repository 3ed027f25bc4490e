#!/usr/bin/env python3
"""Does the slab's access pattern (rows of a few hundred contiguous bytes at a stride of ld * 8 bytes) cost
HBM bandwidth against the same bytes stored back to back?  torch reductions / copies over a strided view of a
[B, 704, 720] slab and over a contiguous tensor of the same payload."""
import sys, time
import torch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = "cuda"
A = torch.empty([B, 704, 720], dtype=torch.float64, device=dev)
A.zero_()


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for cols in (48, 64, 80, 128):
    view = A[:, :, :cols]
    C = torch.zeros([B, 704, cols], dtype=torch.float64, device=dev)
    D = torch.empty_like(C)
    nbytes = C.numel() * 8
    t_rs = timed(lambda: view.sum())
    t_rc = timed(lambda: C.sum())
    t_cs = timed(lambda: D.copy_(view))      # strided read, contiguous write
    t_cc = timed(lambda: D.copy_(C))
    t_ws = timed(lambda: view.copy_(C))      # contiguous read, strided write
    print(f"{cols * 8:5d} B rows: sum strided {nbytes / t_rs / 1e12:.2f} TB/s, contiguous {nbytes / t_rc / 1e12:.2f} | "
          f"copy strided->contig {2 * nbytes / t_cs / 1e12:.2f}, contig->contig {2 * nbytes / t_cc / 1e12:.2f}, "
          f"contig->strided {2 * nbytes / t_ws / 1e12:.2f} TB/s (read+write)")
