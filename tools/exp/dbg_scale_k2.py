import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import acav100m_amd
acav100m_amd.configure_runtime(quiet=True)
from acav100m_amd.clustering import KMeans
def mixture(rs, n, d, k, spread):
    cen = rs.randn(k, d).astype(np.float32) * spread
    return (cen[rs.randint(0, k, n)] + rs.randn(n, d).astype(np.float32)).astype(np.float32)
cases = [(3000, 128, 257), (3000, 128, 200), (3000, 1024, 257)]
for (n, d, k) in cases:
    for scale in (1.0, 2.0 ** -20, 1e-6, 1e3):
        rs = np.random.RandomState(5)
        x = (mixture(rs, n, d, k, 1.0) * np.float32(scale)).astype(np.float32)
        cen = (mixture(rs, k, d, k, 1.0) * np.float32(scale)).astype(np.float32)
        cnts = np.full(k, 500, np.float32)
        km = KMeans(None, d, k)
        km.centers, km.counts, km.count = cen, cnts, 10 * k + int(cnts.sum())
        km.to("cuda:0")
        xt = torch.from_numpy(x).cuda()
        a, _ = km.calc_best(xt, need_mean=False)
        st = km.filter_stats(), km.recheck_stats()
        b, _ = km.calc_best(xt, need_mean=True)
        bad = (a != b).nonzero().flatten().cpu().numpy()
        print(os.environ.get("TAG", ""), n, d, k, "scale", scale, "wrong", len(bad), "stats", st, flush=True)
