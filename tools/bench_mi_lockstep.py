"""MI greedy selection with C independent chunks in lockstep on ONE GPU (acav_mi_run_greedy_multi): aggregate
throughput vs one chunk alone.   usage: bench_mi_lockstep.py V C_centroids D chunks"""
import itertools
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acav100m_amd.rng import Generator
from acav100m_amd.subset_selection import get_measure
from acav100m_amd.subset_selection.measures.batch import EfficientBatchMI

v = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
c = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dd = int(sys.argv[3]) if len(sys.argv) > 3 else 2
chunks = int(sys.argv[4]) if len(sys.argv) > 4 else 8
pairs = list(itertools.combinations(range(dd), 2))
subset = round(0.2 * v)


def make(i):
    rs = np.random.RandomState(i)
    comp = rs.randint(0, c, v)
    a = np.stack([np.where(rs.rand(v) < 0.5, comp, rs.randint(0, c, v)) for _ in range(dd)], 1).astype(np.int64)
    cand = [int(j) for j in rs.permutation(v)]
    m = get_measure("batch_mi")(a, ncentroids=c, batch_size=20, selection_size=4, device="cuda:0", keep_unselected=True,
                                generator=Generator(1000 + i))
    m.init(pairs, cand[1:])
    return m, cand[:1]


m, st = make(0)
m.run_greedy(400, st, None)  # warm-up
m, st = make(0)
t0 = time.perf_counter()
ref = m.run_greedy(subset, st, None)
single = time.perf_counter() - t0
ms = [make(i) for i in range(chunks)]
t0 = time.perf_counter()
out = EfficientBatchMI.run_greedy_multi([x[0] for x in ms], [subset] * chunks, [x[1] for x in ms])
dt = time.perf_counter() - t0
assert out[0][0] == ref[0]
iters = (subset + 3) // 4
print(json.dumps({"V": v, "C": c, "D": dd, "chunks": chunks, "seconds": dt, "single_chunk_seconds": single,
                  "us_per_iteration_single": single / iters * 1e6, "us_per_lockstep_iteration": dt / iters * 1e6,
                  "aggregate_us_per_chunk_iteration": dt / (iters * chunks) * 1e6,
                  "selected_clips_per_s": sum(len(o[0]) for o in out) / dt, "curated_clips_per_s": v * chunks / dt,
                  "speedup_vs_sequential": single * chunks / dt}))
