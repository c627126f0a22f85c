import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import acav100m_amd
acav100m_amd.configure_runtime(quiet=True)
from acav100m_amd.clustering import KMeans
def mixture(rs, n, d, k, spread):
    cen = rs.randn(k, d).astype(np.float32) * spread
    return (cen[rs.randint(0, k, n)] + rs.randn(n, d).astype(np.float32)).astype(np.float32)
for (n, d, k) in ((129, 128, 257), (129, 88, 600), (3000, 128, 257), (129, 128, 256), (129, 10, 600)):
    for scale in (1.0, 1e-6, 1e3):
        for offset in (0.0, 3.0):
            rs = np.random.RandomState(5)
            x = ((mixture(rs, n, d, k, 1.0) + offset) * np.float32(scale)).astype(np.float32)
            cen = ((mixture(rs, k, d, k, 1.0) + offset) * np.float32(scale)).astype(np.float32)
            cnts = np.full(k, 500, np.float32)
            km = KMeans(None, d, k)
            km.centers, km.counts, km.count = cen, cnts, 10 * k + int(cnts.sum())
            km.to("cuda:0")
            xt = torch.from_numpy(x).cuda()
            a, _ = km.calc_best(xt, need_mean=False)
            st = km.filter_stats(), km.recheck_stats()
            b, _ = km.calc_best(xt, need_mean=True)
            bad = (a != b).nonzero().flatten().cpu().numpy()
            print(n, d, k, "scale", scale, "offset", offset, "wrong", len(bad), bad[:8], "stats", st, flush=True)
