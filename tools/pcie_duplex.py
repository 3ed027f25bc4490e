#!/usr/bin/env python3
"""Host <-> device copy rates with page-locked buffers: one direction at a time, both at once on two streams,
and split into several copies per direction (what StreamedSolver does: 8 input fields up, 4 result fields down)."""
import time, torch
dev = torch.device("cuda:0")
MB = 1 << 20
def rate(fn, nbytes, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t0) / 1e9
for size in (16 * MB, 128 * MB, 512 * MB):
    h_up, h_dn = torch.empty(size, dtype=torch.uint8).pin_memory(), torch.empty(size, dtype=torch.uint8).pin_memory()
    d_up, d_dn = torch.empty(size, dtype=torch.uint8, device=dev), torch.empty(size, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    def up():
        with torch.cuda.stream(s1): d_up.copy_(h_up, non_blocking=True)
    def dn():
        with torch.cuda.stream(s2): h_dn.copy_(d_dn, non_blocking=True)
    def both():
        up(); dn()
    def both_split(k=8):
        n = size // k
        with torch.cuda.stream(s1):
            for i in range(k): d_up[i * n:(i + 1) * n].copy_(h_up[i * n:(i + 1) * n], non_blocking=True)
        with torch.cuda.stream(s2):
            for i in range(k): h_dn[i * n:(i + 1) * n].copy_(d_dn[i * n:(i + 1) * n], non_blocking=True)
    print(f"{size // MB:4d} MB: up {rate(up, size):5.1f} GB/s | down {rate(dn, size):5.1f} GB/s | both at once "
          f"{rate(both, 2 * size):5.1f} GB/s combined | both, 8 copies each {rate(both_split, 2 * size):5.1f} GB/s combined")
