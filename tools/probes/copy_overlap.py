#!/usr/bin/env python3
"""Does a host-to-device copy from pinned memory run beside kernels of another stream on this box?  (torch streams only)"""
import time, torch
dev = torch.device("cuda:0")
a = torch.randn(8192, 8192, device=dev)
h = torch.empty(77_000_000 // 4, dtype=torch.float32).pin_memory()
d = torch.empty_like(h, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def kern(n):
    with torch.cuda.stream(s2):
        for _ in range(n): b = a @ a
def copy(n):
    with torch.cuda.stream(s1):
        for _ in range(n): d.copy_(h, non_blocking=True)
for f, label in ((lambda: kern(4), "kernels"), (lambda: copy(4), "copies"), (lambda: (kern(4), copy(4)), "both")):
    f(); torch.cuda.synchronize()
    t = time.perf_counter(); f(); torch.cuda.synchronize()
    print(label, "%.2f ms" % ((time.perf_counter() - t) * 1e3))
