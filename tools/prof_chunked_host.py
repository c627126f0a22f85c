"""Where the HOST time of a chunked selection goes (cProfile around bench.select_chunked: 20 chunks of 100k clips, 10 in lockstep).
r4: per chunk 16.4 ms inside acav_mi_run_greedy_multi, ~8 ms outside it (handle set-up / release, array hand-over, lists)."""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
n, chunk, width = 2_000_000, 100_000, 10
rs = np.random.RandomState(0)
comp = rs.randint(0, 1024, n)
a = np.stack([np.where(rs.rand(n) < 0.5, comp, rs.randint(0, 1024, n)) for _ in range(2)], 1).astype(np.int64)
types = ["a", "b"]
bench.select_chunked(a[:200000], types, chunk, 2)  # warm
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
res = bench.select_chunked(a, types, chunk, width)
pr.disable()
print("total", time.perf_counter() - t0, "chunks", len(res))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
