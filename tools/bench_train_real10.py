"""The reference's real pipeline trains TEN clusterings over one batch stream (VGGish 64 / 128 / 256 / 512 / 128, SlowFast
88 / 352 / 704 / 1408 / 2304; models/vggish.py:20, models/slowfast.py:31).  One epoch of each clustering alone
(acav_kmeans_train) against one epoch of all ten in one acav_kmeans_train_multi call, same box.  BENCH_K (32), BENCH_N."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acav100m_amd
acav100m_amd.configure_runtime(quiet=True)
import torch
from acav100m_amd.clustering import KMeans

n = int(os.environ.get("BENCH_N", "262144"))
k = int(os.environ.get("BENCH_K", "32"))
dims = [int(v) for v in os.environ.get("BENCH_DIMS", "64,128,256,512,128,88,352,704,1408,2304").split(",")]
b = 32
g = torch.Generator(device="cuda").manual_seed(0)
xs = []
for dv in dims:
    cen = torch.randn(k, dv, device="cuda", generator=g) * 4
    xs.append(cen[torch.randint(0, k, (n,), device="cuda", generator=g)] + 0.3 * torch.randn(n, dv, device="cuda", generator=g))
torch.cuda.synchronize()
acav100m_amd.manual_seed(0)
kms = [KMeans(None, dv, k).to("cuda:0") for dv in dims]
steps = n // b


def one(i):
    t0 = time.perf_counter()
    kms[i].train_epoch(xs[i], b, lr=0.01)
    kms[i].synchronize()
    return time.perf_counter() - t0


def all_of(idx):
    t0 = time.perf_counter()
    KMeans.train_epoch_multi([kms[i] for i in idx], [xs[i] for i in idx], b, lr=0.01)
    for i in idx:
        kms[i].synchronize()
    return time.perf_counter() - t0


for i in range(len(dims)):
    one(i)
all_of(range(len(dims)))
for rep in range(2):
    alone = [one(i) / steps * 1e6 for i in range(len(dims))]
    t = all_of(range(len(dims))) / steps * 1e6
    print(f"K={k}: alone us/step " + " ".join(f"{dv}:{a:.2f}" for dv, a in zip(dims, alone)) + f" | sum {sum(alone):.1f} max {max(alone):.1f}"
          f" | all ten in one call {t:.2f} us per step of the ten ({t / len(dims):.2f} effective); "
          f"launches / fallbacks {[km.train_stats() for km in kms]}", flush=True)
for sub in ([0, 1, 2, 3, 4], [5, 6, 7], [8, 9], [0, 1, 2, 3, 4, 5, 6, 7]):
    t = all_of(sub) / steps * 1e6
    print(f"   subset {[dims[i] for i in sub]}: {t:.2f} us per step of the subset", flush=True)
