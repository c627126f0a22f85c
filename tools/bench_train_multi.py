"""Do two clusterings of one shape really train side by side?  One epoch through acav_kmeans_train (one handle) against one
epoch of TWO handles through acav_kmeans_train_multi, same shape, same box.  BENCH_D / BENCH_K / BENCH_N select the shape.
side by side: the pair takes about as long as one; not side by side: twice as long."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acav100m_amd
acav100m_amd.configure_runtime(quiet=True)
from acav100m_amd import _lib
from acav100m_amd.clustering import KMeans

if os.environ.get("ACAV_LIB"):  # A/B of another build of the library (tools only, as tools/ab_train.py)
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.__file__), os.environ["ACAV_LIB"])

n = int(os.environ.get("BENCH_N", "262144"))
d = int(os.environ.get("BENCH_D", "1024"))
k = int(os.environ.get("BENCH_K", "1024"))
d2 = int(os.environ.get("BENCH_D2", str(d)))  # the second clustering's width (cfg4: 2048 + 128)
b = 32
g = torch.Generator(device="cuda").manual_seed(0)
xs = []
for dv in (d, d2):
    cen = torch.randn(k, dv, device="cuda", generator=g) * 4
    xs.append(cen[torch.randint(0, k, (n,), device="cuda", generator=g)] + 0.3 * torch.randn(n, dv, device="cuda", generator=g))
torch.cuda.synchronize()
acav100m_amd.manual_seed(0)
kms = [KMeans(None, dv, k).to("cuda:0") for dv in (d, d2)]


def one(i=0):
    t0 = time.perf_counter()
    kms[i].train_epoch(xs[i], b, lr=0.01)
    kms[i].synchronize()
    return time.perf_counter() - t0


def pair():
    t0 = time.perf_counter()
    KMeans.train_epoch_multi(kms, xs, b, lr=0.01)
    for km in kms:
        km.synchronize()
    return time.perf_counter() - t0


one(0), one(1), pair()
steps = n // b
for rep in range(3):
    t1, t1b, t2 = one(0), one(1), pair()
    print(f"d={d}/{d2} K={k}: one handle {t1 / steps * 1e6:.2f} / {t1b / steps * 1e6:.2f} us/step; two handles in one call {t2 / steps * 1e6:.2f} "
          f"us per step of the pair ({t2 / (2 * steps) * 1e6:.2f} effective); launches / fallbacks {[km.train_stats() for km in kms]}")
