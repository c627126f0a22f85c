"""us per SGD step of acav_kmeans_train at several batch sizes (the multi-GPU path trains on global batches of 32 W rows).
argv: batch sizes; BENCH_D / BENCH_K select the shape."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acav100m_amd
acav100m_amd.configure_runtime(quiet=True)
from acav100m_amd.clustering import KMeans

n = 262144
d = int(os.environ.get("BENCH_D", "1024"))
k = int(os.environ.get("BENCH_K", "256"))
g = torch.Generator(device="cuda").manual_seed(0)
cen = torch.randn(k, d, device="cuda", generator=g) * 4
x = cen[torch.randint(0, k, (n,), device="cuda", generator=g)] + 0.3 * torch.randn(n, d, device="cuda", generator=g)
torch.cuda.synchronize()
for b in [int(a) for a in sys.argv[1:]] or [32, 64, 128, 256]:
    acav100m_amd.manual_seed(0)
    km = KMeans(None, d, k).to("cuda:0")
    km.train_epoch(x, b, lr=0.01)
    km.synchronize()
    t0 = time.perf_counter()
    km.train_epoch(x, b, lr=0.01)
    km.synchronize()
    dt = time.perf_counter() - t0
    print(f"b={b}: {dt / (n // b) * 1e6:.2f} us/step, {n / dt / 1e6:.2f} M rows/s; persistent launches / fallbacks {km.train_stats()}")
