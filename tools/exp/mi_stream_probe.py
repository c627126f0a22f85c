"""Is the greedy loop's speed a matter of WHICH hardware queues its streams land on?  Creates MI_EXTRA_STREAMS live HIP streams (each
touched once) before the selection object creates its own, then times 6 000 iterations at V = 10^6.  One process per setting."""
import itertools, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import acav100m_amd
acav100m_amd.configure_runtime(quiet=True)
from acav100m_amd.subset_selection import get_measure
extra = int(os.environ.get("MI_EXTRA_STREAMS", "0"))
keep = []
for _ in range(extra):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        keep.append(torch.zeros(16, device="cuda") + 1)
    keep.append(s)
torch.cuda.synchronize()
v, c, dd = 1_000_000, 256, 2
rs = np.random.RandomState(0)
comp = rs.randint(0, c, v)
a = np.stack([np.where(rs.rand(v) < 0.5, comp, rs.randint(0, c, v)) for _ in range(dd)], 1).astype(np.int64)
pairs = list(itertools.combinations(range(dd), 2))
cand = [int(i) for i in rs.permutation(v)]
acav100m_amd.manual_seed(0)
m = get_measure("batch_mi")(a, ncentroids=c, batch_size=20, selection_size=4, device="cuda:0", keep_unselected=True)
m.init(pairs, cand[1:])
iters = 6000
t0 = time.perf_counter()
S, G, _, _ = m.run_greedy(round(0.2 * v), cand[:1], None, max_iters=iters)
dt = time.perf_counter() - t0
print("GPU_MAX_HW_QUEUES %s, %d extra live streams: %.2f us per iteration" % (os.environ.get("GPU_MAX_HW_QUEUES"), extra, dt / iters * 1e6), flush=True)
