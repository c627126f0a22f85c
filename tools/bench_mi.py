"""MI greedy selection timing (SURVEY 8(d) 'Roofline -- MI greedy'): iterations/s, selected clips/s,
permutation-stream GB/s; optional oracle (CPU port) timing on a bounded number of iterations."""
import itertools
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acav100m_amd
acav100m_amd.configure_runtime(quiet=True)
from acav100m_amd.subset_selection import get_measure

v = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
c = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dd = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cpu_iters = int(sys.argv[4]) if len(sys.argv) > 4 else 0
max_iters = int(sys.argv[5]) if len(sys.argv) > 5 else -1  # bound the run (profiling at large V)
rs = np.random.RandomState(0)
comp = rs.randint(0, c, v)
a = np.stack([np.where(rs.rand(v) < 0.5, comp, rs.randint(0, c, v)) for _ in range(dd)], 1).astype(np.int64)
a[0] = c - 1
pairs = list(itertools.combinations(range(dd), 2))
cand = [int(i) for i in rs.permutation(v)]
subset = round(0.2 * v)
acav100m_amd.manual_seed(0)
m = get_measure("batch_mi")(a, ncentroids=c, batch_size=20, selection_size=4, device="cuda:0", keep_unselected=True)
m.init(pairs, cand[1:])
t0 = time.perf_counter()
S, G, _, _ = m.run_greedy(subset, cand[:1], None, max_iters=max_iters)
dt = time.perf_counter() - t0
iters = (subset + 3) // 4 if max_iters < 0 else min(max_iters, (subset + 3) // 4)
out = {"V": v, "C": c, "D": dd, "P": len(pairs), "selected": len(S), "iters": iters, "seconds": dt,
       "us_per_iter": dt / iters * 1e6, "selected_clips_per_s": len(S) / dt, "curated_clips_per_s": v / dt,
       "perm_stream_GBs": sum(16 * (v - 1 - 4 * t) for t in range(iters)) / dt / 1e9}
if cpu_iters:
    from oracle import oracle as O
    om = O.BatchMI(a, c, pairs)
    t0 = time.perf_counter()
    om.run_greedy(cand[1:], cand[:1], subset, 20, 4, O.Rng(0), max_iters=cpu_iters)
    ct = time.perf_counter() - t0
    out["oracle_us_per_iter"] = ct / cpu_iters * 1e6
    out["speedup_vs_oracle"] = out["oracle_us_per_iter"] / out["us_per_iter"]
print(json.dumps(out))
