"""Does the greedy loop's speed depend on what the process did before?  

  python tools/mi_history_probe.py [handles] [keep] [iterations]

Creates `handles` KMeans handles (each with its own stream; a short epoch + an assign sweep so that every stream really ran),
destroys all but `keep` of them, then times `iterations` iterations of the 1M-clip greedy loop (set-up excluded: the loop's own
device time through ACAV_MI_TIMING).  handles = 0: the clean process.  Prints one line:  us_per_iteration <value>"""
import itertools
import os
import re
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    handles = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    keep = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 8000
    import numpy as np
    import torch
    import acav100m_amd
    acav100m_amd.configure_runtime(quiet=True)
    from acav100m_amd.clustering import KMeans
    from acav100m_amd.subset_selection import get_measure
    held = []
    if handles:
        n, d, k = 16384, 256, 64
        gen = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn(n, d, device="cuda", generator=gen)
        acav100m_amd.manual_seed(0)
        for i in range(handles):
            km = KMeans(None, d, k).to("cuda:0")
            km.initialize()
            km.train_epoch(x, 32, lr=0.01)
            km.calc_best(x, need_mean=False)
            km.synchronize()
            held.append(km)
            if len(held) > keep and i % 2 == 1:  # destroy some while others are still alive: holes in the creation order
                held.pop(0)
        held = held[len(held) - keep:] if keep else []
    v, c, dd = 1_000_000, 256, 2
    rs = np.random.RandomState(0)
    comp = rs.randint(0, c, v)
    a = np.stack([np.where(rs.rand(v) < 0.5, comp, rs.randint(0, c, v)) for _ in range(dd)], 1).astype(np.int64)
    pairs = list(itertools.combinations(range(dd), 2))
    cand = [int(i) for i in rs.permutation(v)]
    acav100m_amd.manual_seed(0)
    m = get_measure("batch_mi")(a, ncentroids=c, batch_size=20, selection_size=4, device="cuda:0", keep_unselected=True)
    m.init(pairs, cand[1:])
    m.run_greedy(round(0.2 * v), cand[:1], None, max_iters=500)  # warm: code objects, buffers
    m = get_measure("batch_mi")(a, ncentroids=c, batch_size=20, selection_size=4, device="cuda:0", keep_unselected=True)
    m.init(pairs, cand[1:])
    t0 = time.perf_counter()
    m.run_greedy(round(0.2 * v), cand[:1], None, max_iters=iters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("wall incl. set-up: %.2f us per iteration; handles alive: %d" % (dt / iters * 1e6, len(held)), file=sys.stderr)


if __name__ == "__main__":
    if os.environ.get("ACAV_MI_TIMING") is None:  # re-run under ACAV_MI_TIMING and report the loop's own figure
        env = dict(os.environ, ACAV_MI_TIMING="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, capture_output=True, text=True, timeout=600)
        lines = re.findall(r"greedy loop: 1 chunk\(s\), (\d+) iterations, host enqueue [\d.]+ us/iteration, enqueue \+ drain ([\d.]+)[^\n]*", r.stderr)
        if r.returncode != 0 or not lines:
            sys.stderr.write(r.stderr[-2000:])
            sys.exit(1)
        its, us = lines[-1]
        rep = re.findall(r"replaced by the queue probe at create: (\d+)", r.stderr)
        print("us_per_iteration %s (iterations %s; streams replaced by the queue probe: %s)" % (us, its, rep[-1] if rep else "?"))
    else:
        main()
