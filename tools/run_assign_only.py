"""Runs only the assign sweep a few times -- a small target for rocprofv3 --pmc passes.
argv: rows reps mode [d K]   (mode: exact | filter)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import acav100m_amd
acav100m_amd.configure_runtime(quiet=True)
from acav100m_amd.clustering import KMeans

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
d = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
k = int(sys.argv[5]) if len(sys.argv) > 5 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = sys.argv[3] if len(sys.argv) > 3 else "exact"
gen = torch.Generator(device="cuda").manual_seed(0)
cen = torch.randn(k, d, device="cuda", generator=gen)
x = cen[torch.randint(0, k, (n,), device="cuda", generator=gen)] + 0.3 * torch.randn(n, d, device="cuda", generator=gen)
torch.cuda.synchronize()
km = KMeans(None, d, k)
km.centers, km.counts, km.count = cen.cpu().numpy(), np.full(k, 1000, np.float32), 10 * k + n
km.to("cuda:0")
for _ in range(reps):
    t0 = time.perf_counter()
    lab, _ = km.calc_best(x, need_mean=(mode == "exact"))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
if mode != "exact":
    import ctypes as C
    from acav100m_amd import _lib
    fm = C.c_float(0)
    _lib.check(_lib._lib.acav_kmeans_filter_time(km._h, C.byref(fm)))
    print("filter kernel alone: %.4f ms -> %.1f GB/s = %.3f of 8 TB/s" % (fm.value, (n * d * 4 + n * 8) / fm.value / 1e6, (n * d * 4 + n * 8) / fm.value / 1e6 / 8000))
print(mode, "assign", n, "rows:", dt * 1e3, "ms ->", 2.0 * n * k * d / dt / 1e12, "TFLOP/s,", (n * d * 4 + n * 8) / dt / 1e9, "GB/s",
      km.filter_stats() if mode != "exact" else "")
