"""Exact greedy ('mi' / 'mem_mi') timing: one launch per iteration, all remaining candidates scored."""
import itertools
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acav100m_amd.subset_selection import get_measure

v = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
c = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dd = int(sys.argv[3]) if len(sys.argv) > 3 else 2
subset = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
rs = np.random.RandomState(0)
comp = rs.randint(0, c, v)
a = np.stack([np.where(rs.rand(v) < 0.5, comp, rs.randint(0, c, v)) for _ in range(dd)], 1).astype(np.int64)
a[0] = c - 1
pairs = list(itertools.combinations(range(dd), 2))
cand = [int(i) for i in rs.permutation(v)]
m = get_measure("mem_mi")(a, ncentroids=c, device="cuda:0")
m.init(pairs, cand[1:])
m.run_greedy(10, cand[:1])  # warm-up
m.init(pairs, cand[1:])
t0 = time.perf_counter()
S, G, _, _ = m.run_greedy(subset, cand[:1])
dt = time.perf_counter() - t0
it = len(G)
print(json.dumps({"V": v, "C": c, "D": dd, "P": len(pairs), "picks": it, "seconds": dt, "us_per_iteration": dt / it * 1e6,
                  "candidate_scores_per_s": sum(v - 1 - t for t in range(it)) * len(pairs) / dt}))
