"""Where does the lockstep selection's wall time go on the HOST side?  (bench.py --workload cfg5: 125 chunks of 100k clips, ten at a time;
the greedy loops add up to 1.74 s, the stage takes 2.5 s -- or 3.3-4.0 s in about one process of three.)  Same orchestration as
bench.select_chunked with a timer around every piece.   usage: prof_lockstep_host.py [clips] [C] [helpers]"""
import itertools
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acav100m_amd
acav100m_amd.configure_runtime(quiet=True)
import bench
from acav100m_amd.rng import Generator
from acav100m_amd.subset_selection.measures.batch import EfficientBatchMI
from acav100m_amd.subset_selection.run_greedy import _prepare

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
c = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
helpers = int(sys.argv[3]) if len(sys.argv) > 3 else 1
chunk, width = 100_000, 10
rs = np.random.RandomState(0)
comp = rs.randint(0, c, n)
a = np.stack([np.where(rs.rand(n) < 0.5, comp, rs.randint(0, c, n)) for _ in range(2)], 1).astype(np.int64)
types = [("audio_model", "layer_0"), ("visual_model", "layer_0")]
sargs = bench.select_args()
t_prep, t_loop, t_wait, t_clear = [], [], [], []


def prepare_group(g0):
    t0 = time.perf_counter()
    out = [_prepare(sargs, a[c0:c0 + chunk], types, None, 0.2, "batch_mi", "combination", True, False, generator=Generator(1 + g0 // chunk + i))
           for i, c0 in enumerate(range(g0, min(n, g0 + chunk * width), chunk))]
    t_prep.append(time.perf_counter() - t0)
    return out


def clear(prepared):
    t0 = time.perf_counter()
    prepared.clear()
    t_clear.append(time.perf_counter() - t0)


import contextlib, io
groups = list(range(0, n, chunk * width))
T0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()), ThreadPoolExecutor(helpers) as pool:
    nxt = pool.submit(prepare_group, groups[0])
    for gi, g0 in enumerate(groups):
        t0 = time.perf_counter()
        prepared = nxt.result()
        t_wait.append(time.perf_counter() - t0)
        if gi + 1 < len(groups):
            nxt = pool.submit(prepare_group, groups[gi + 1])
        t0 = time.perf_counter()
        res = EfficientBatchMI.run_greedy_multi([p[0] for p in prepared], [p[2] for p in prepared], [p[1] for p in prepared])
        t_loop.append(time.perf_counter() - t0)
        if os.environ.get("CLEAR_IN_HELPER"):
            pool.submit(clear, prepared)  # rounds 4-6: under the next group's loop
        else:
            clear(prepared)  # on the launching thread before the next set-up, as bench.select_chunked does now
        del prepared
total = time.perf_counter() - T0
f = lambda v: " ".join("%.0f" % (x * 1e3) for x in v)
print("helpers %d: total %.3f s for %d groups; ms per group: prepare [%s] | main waits for it [%s] | run_greedy_multi [%s] | clear [%s]" %
      (helpers, total, len(groups), f(t_prep), f(t_wait), f(t_loop), f(t_clear)))
