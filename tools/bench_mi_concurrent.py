"""MI greedy selection with several independent chunks in flight on ONE GPU (one host thread + one handle + one
generator per chunk; the reference runs one chunk per GPU at a time, chunk.py:26-53).  Reports aggregate throughput.
usage: bench_mi_concurrent.py V C D threads [chunks]"""
import itertools
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acav100m_amd
acav100m_amd.configure_runtime(quiet=True)
from acav100m_amd.rng import Generator
from acav100m_amd.subset_selection import get_measure

v = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
c = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dd = int(sys.argv[3]) if len(sys.argv) > 3 else 2
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 8
chunks = int(sys.argv[5]) if len(sys.argv) > 5 else threads
pairs = list(itertools.combinations(range(dd), 2))
subset = round(0.2 * v)


def make_chunk(i):
    rs = np.random.RandomState(i)
    comp = rs.randint(0, c, v)
    a = np.stack([np.where(rs.rand(v) < 0.5, comp, rs.randint(0, c, v)) for _ in range(dd)], 1).astype(np.int64)
    cand = [int(j) for j in rs.permutation(v)]
    return a, cand


data = [make_chunk(i) for i in range(chunks)]


def run(i):
    a, cand = data[i]
    m = get_measure("batch_mi")(a, ncentroids=c, batch_size=20, selection_size=4, device="cuda:0", keep_unselected=True,
                                generator=Generator(1000 + i))
    m.init(pairs, cand[1:])
    S, G, _, _ = m.run_greedy(subset, cand[:1], None)
    return S


run(0)  # warm-up (library load, first launches)
single0 = time.perf_counter()
ref = run(0)
single = time.perf_counter() - single0
t0 = time.perf_counter()
with ThreadPoolExecutor(threads) as ex:
    out = list(ex.map(run, range(chunks)))
dt = time.perf_counter() - t0
assert out[0] == ref, "a chunk's selection must not depend on what runs beside it"
iters = (subset + 3) // 4
print(json.dumps({"V": v, "C": c, "D": dd, "threads": threads, "chunks": chunks, "seconds": dt,
                  "single_chunk_seconds": single, "us_per_iter_single": single / iters * 1e6,
                  "aggregate_us_per_iter": dt / (iters * chunks) * 1e6,
                  "selected_clips_per_s": sum(len(s) for s in out) / dt, "curated_clips_per_s": v * chunks / dt,
                  "speedup_vs_sequential": single * chunks / dt}))
