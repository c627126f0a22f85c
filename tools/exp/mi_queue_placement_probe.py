"""NOTES_r05 section 13: does the greedy loop slow down when k-means handles of a previous pass outlive it (streams created in a
different order)?  Three passes of: two KMeans handles (a short epoch each, side by side) + 8 000 greedy iterations at V = 10^6.
KEEP=1 keeps the previous pass's handles alive while the new ones are created (the accidental bench.py pattern)."""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import acav100m_amd
acav100m_amd.configure_runtime(quiet=True)
from acav100m_amd.clustering import KMeans
from acav100m_amd.subset_selection import get_measure
keep = os.environ.get("KEEP") == "1"
n, d, k = 65536, 1024, 256
gen = torch.Generator(device="cuda").manual_seed(0)
xs = [torch.randn(n, d, device="cuda", generator=gen) for _ in range(2)]
v, c, dd = 1_000_000, 256, 2
rs = np.random.RandomState(0)
comp = rs.randint(0, c, v)
a = np.stack([np.where(rs.rand(v) < 0.5, comp, rs.randint(0, c, v)) for _ in range(dd)], 1).astype(np.int64)
pairs = list(itertools.combinations(range(dd), 2))
cand = [int(i) for i in rs.permutation(v)]
held = None
for p in range(4):
    if not keep:
        held = None
    acav100m_amd.manual_seed(0)
    kms = [KMeans(None, d, k).to("cuda:0") for _ in range(2)]
    for km in kms:
        km.initialize()
    KMeans.train_epoch_multi(kms, xs, 32, lr=0.01)
    for km, x in zip(kms, xs):
        km.calc_best(x, need_mean=False)
        km.synchronize()
    held = kms
    m = get_measure("batch_mi")(a, ncentroids=c, batch_size=20, selection_size=4, device="cuda:0", keep_unselected=True)
    m.init(pairs, cand[1:])
    iters = 8000
    t0 = time.perf_counter()
    m.run_greedy(round(0.2 * v), cand[:1], None, max_iters=iters)
    dt = time.perf_counter() - t0
    print("KEEP=%d pass %d: %.2f us per iteration (incl. set-up)" % (keep, p, dt / iters * 1e6), flush=True)
    del m
