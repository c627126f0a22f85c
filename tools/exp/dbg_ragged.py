import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import acav100m_amd
acav100m_amd.configure_runtime(quiet=True)
from acav100m_amd.clustering import KMeans
from oracle import oracle as O
rs = np.random.RandomState(1)
for d, K in ((96, 32), (88, 32), (64, 32), (160, 32), (88, 256)):
    n = 3000
    cen = (2.0 * rs.randn(K, d)).astype(np.float32)
    x = (cen[rs.randint(0, K, n)] + 0.3 * rs.randn(n, d)).astype(np.float32)
    centers = (cen + 0.05 * rs.randn(K, d)).astype(np.float32)
    counts = np.full(K, 1000, np.float32); count = 10 * K + 2000
    ref = O.KMeans(d, K, O.Rng(0), centers=centers); ref.set_state(None, counts, count)
    lab_ref = ref.calc_best(x)[0]
    km = KMeans(None, d, K); km.centers, km.counts, km.count = centers, counts, count; km.to("cuda:0")
    lab = km.calc_best(torch.from_numpy(x).cuda(), need_mean=False)[0].cpu().numpy()
    print(d, K, "filter path: differ", (lab != lab_ref).sum(), "stats", km.filter_stats(), km.recheck_stats(), flush=True)
    lab2 = km.calc_best(torch.from_numpy(x).cuda())[0].cpu().numpy()
    print(d, K, "exact path: differ", (lab2 != lab_ref).sum(), flush=True)
    dp = (d + 31) // 32 * 32
    if dp != d:
        xp = np.zeros((n, dp), np.float32); xp[:, :d] = x
        cp = np.zeros((K, dp), np.float32); cp[:, :d] = centers
        km2 = KMeans(None, dp, K); km2.centers, km2.counts, km2.count = cp, counts, count; km2.to("cuda:0")
        lab3 = km2.calc_best(torch.from_numpy(xp).cuda(), need_mean=False)[0].cpu().numpy()
        print(d, K, "host-padded natural", dp, ": differ", (lab3 != lab_ref).sum(), km2.filter_stats(), flush=True)
