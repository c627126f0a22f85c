"""Which centre spread leaves the half-precision filter most rows undecided with <= 16 candidates each (test data calibration)."""
import numpy as np, torch, sys
sys.path.insert(0, ".")
import acav100m_amd as acav
acav100m_amd.configure_runtime(quiet=True)
from acav100m_amd.clustering import KMeans
n, d, K = 40_000, 1024, 256
for spread in (0.016, 0.013, 0.011, 0.0095, 0.008, 0.007):
    rs = np.random.RandomState(21)
    cen = (spread * rs.randn(K, d)).astype(np.float32)
    x = (cen[rs.randint(0, K, n)] + 0.3 * rs.randn(n, d)).astype(np.float32)
    centers = (cen + 0.2 * spread * rs.randn(K, d)).astype(np.float32)
    for disc in (False, True):
        counts = np.full(K, 1000, np.float32)
        if disc:
            counts[rs.randint(0, K, 40)] = 3.0
        km = KMeans(None, d, K)
        km.centers, km.counts, km.count = centers, counts, 10 * K + 200_000
        km.to("cuda:0")
        lab, _ = km.calc_best(torch.from_numpy(x).cuda(), need_mean=False)
        print(spread, "discounted" if disc else "plain", km.filter_stats()[2], km.recheck_stats(), flush=True)
