"""Randomised parity stress (GPU vs oracle, bit-exact): random shapes for assign (both paths), bulk training
(persistent, wide persistent and per-step kernels, batches up to 512), several clusterings side by side, MI batch greedy
(start sets, partial candidate lists, B up to 64), several chunks in lockstep, exact greedy, and the multi-clustering DDP
epoch through the C-ABI communicator with a world of one.  usage: stress_parity.py [seconds] [seed]
tests/test_gpu_configs.py runs a fixed-seed slice of it (stress(seed, budget, max_cases)) inside `pytest -m gpu`."""
import itertools
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def stress(seed=0, budget=60.0, max_cases=None):
    import torch
    import acav100m_amd
    from acav100m_amd.clustering import KMeans
    from acav100m_amd.rng import Generator
    from acav100m_amd.subset_selection import get_measure
    from oracle import oracle as O
    rs = np.random.RandomState(seed)
    t_end = time.time() + budget
    counts = dict(assign=0, train=0, mi=0, exact=0, lockstep=0, train_multi=0, ddp=0)

    def mixture(n, d, k, spread):
        cen = rs.randn(k, d).astype(np.float32) * spread
        return (cen[rs.randint(0, k, n)] + rs.randn(n, d).astype(np.float32)).astype(np.float32)

    while time.time() < t_end and (max_cases is None or sum(counts.values()) < max_cases):
        which = rs.randint(0, 7)
        if which == 0:  # assign, filter and exact paths
            d = int(rs.choice([8, 10, 32, 64, 88, 96, 128, 130, 160, 352, 512, 704, 1001, 1024, 1056, 2304]))
            k = int(rs.choice([2, 3, 17, 64, 255, 256, 257, 300, 600, 1024, 1500]))
            n = int(rs.choice([1, 63, 128, 129, 1000, 4097, 20000]))
            if k > 600 and (n > 4097 or d > 1056):
                n, d = min(n, 4097), min(d, 1056)  # keep the oracle's share of a case short
            x = mixture(n, d, k, float(rs.choice([0.05, 1.0, 4.0])))
            offset = float(rs.choice([0.0, 0.0, 3.0, 40.0]))  # a large common component: the centred filter's case
            x = (x + offset).astype(np.float32)
            s = int(rs.randint(1 << 30))
            acav100m_amd.manual_seed(s)
            km = KMeans(None, d, k).to("cuda:0")
            ref = O.KMeans(d, k, O.Rng(s))
            cen = (mixture(k, d, k, 1.0) + offset).astype(np.float32)
            # the data's overall scale (round 5: half-precision filter operands with exact power-of-two scaling; rows and centres at
            # 1e-6 are all below half's smallest normal, at 3e4 all above its largest finite value) and, sometimes, centres that are
            # much smaller than the rows (noise-dominated rows: the row scale is taken from the centres)
            scale = np.float32(rs.choice([1.0, 1.0, 1e-3, 1e3, 1e-6, 3e4]))
            x = (x * scale).astype(np.float32)
            cen = (cen * scale * np.float32(rs.choice([1.0, 1.0, 1.0, 0.02]))).astype(np.float32)
            cnts = rs.randint(0, 50, k).astype(np.float32) if rs.rand() < 0.5 else np.full(k, 500, np.float32)
            km.centers, km.counts, km.count = cen, cnts, 10 * k + int(cnts.sum())
            ref.set_state(cen, cnts, 10 * k + int(cnts.sum()))
            xt = torch.from_numpy(x).cuda()
            want = ref.calc_best(x)[0]
            a, _ = km.calc_best(xt, need_mean=False)
            b, _ = km.calc_best(xt, need_mean=True)
            bad_a, bad_b = int((a.cpu().numpy() != want).sum()), int((b.cpu().numpy() != want).sum())
            assert bad_a == 0 and bad_b == 0, ("assign", n, d, k, s, "scale", float(scale), "offset", offset, "labels that differ: filter path",
                                               bad_a, "exact path", bad_b, "undecided", km.filter_stats())
            counts["assign"] += 1
        elif which == 1:  # bulk training
            d = int(rs.choice([8, 30, 64, 88, 128, 130, 256, 352, 704, 1000, 1024, 1280, 1408, 2048]))  # 1280 / 2048 with K >= 512: the column-split kernel
            k = int(rs.choice([3, 16, 40, 64, 100, 256, 300, 600, 1000, 1024, 1030, 2048]))
            b = int(rs.choice([1, 7, 16, 24, 32, 48, 64, 128, 200, 256, 512, 1024]))  # >= 128: several row groups per workgroup
            steps = int(rs.randint(3, 60)) if b <= 128 else int(rs.randint(3, 12))
            if k >= 1000:  # the warm-up alone is 10 K rows: a few real steps beyond it, no more
                steps = (10 * k) // b + int(rs.randint(2, 10))
                d = min(d, 256) if rs.rand() < 0.7 else d
            lr = float(rs.choice([0.01, 0.01, 0.3]))
            x = mixture(steps * b + int(rs.randint(0, b)), d, k, 3.0)
            s = int(rs.randint(1 << 30))
            acav100m_amd.manual_seed(s)
            km = KMeans(None, d, k).to("cuda:0")
            ref = O.KMeans(d, k, O.Rng(s))
            xt = torch.from_numpy(x).cuda()
            for _ in range(2):
                km.train_epoch(xt, b, lr=lr)
                ref.train_epoch(x, b, lr=lr)
            assert np.array_equal(km.centers.numpy(), ref.centers) and np.array_equal(km.counts.numpy(), ref.counts), ("train", d, k, b, steps, lr, s)
            assert km.count == ref.count and km.fallback == ref.fallback
            counts["train"] += 1
        elif which == 4:  # several chunks in lockstep (own sizes, tables and generators) == each chunk alone == the oracle
            from acav100m_amd.subset_selection.measures.batch import EfficientBatchMI
            nch = int(rs.randint(2, 6))
            B = int(rs.choice([4, 20, 33])); kk = int(rs.randint(1, B + 1))
            ms, want, subsets, starts = [], [], [], []
            for i in range(nch):
                v = int(rs.choice([200, 1000, 3000, 12000]))
                dd = int(rs.choice([2, 3, 5])); c = int(rs.choice([4, 40, 256]))
                a = rs.randint(0, c, (v, dd)).astype(np.int64)
                a[0] = c - 1
                pairs = list(itertools.combinations(range(dd), 2))
                cand = rs.permutation(v)
                subset = int(rs.randint(1, max(2, v // 8)))
                iters = (subset + kk - 1) // kk
                if (v - 1) - (iters - 1) * kk < B:
                    subset = kk
                s = int(rs.randint(1 << 30))
                m = get_measure("batch_mi")(a, ncentroids=c, batch_size=B, selection_size=kk, device="cuda:0",
                                            keep_unselected=True, generator=Generator(s))
                m.init(pairs, [int(j) for j in cand[1:]])
                ms.append(m); subsets.append(subset); starts.append([int(cand[0])])
                want.append(O.BatchMI(a, c, pairs).run_greedy(cand[1:], cand[:1], subset, B, m.k, O.Rng(s)))
            got = EfficientBatchMI.run_greedy_multi(ms, subsets, starts)
            for i in range(nch):
                assert got[i][0] == want[i]["S"].tolist() and np.array_equal(np.array(got[i][1]), want[i]["GAIN"]), ("lockstep", nch, i, B, kk)
            counts["lockstep"] += 1
        elif which == 5:  # several clusterings side by side (acav_kmeans_train_multi)
            ncl = int(rs.randint(2, 5))
            b = int(rs.choice([16, 32, 32, 64]))
            steps = int(rs.randint(20, 80))
            shapes = [(int(rs.choice([64, 128, 352, 704, 1024, 2048])), int(rs.choice([16, 64, 256, 300]))) for _ in range(ncl)]
            if rs.rand() < 0.35:  # wide pairs: K > 256 at 768 < d <= 1024 takes the two-row-pass form when another clustering shares the call
                ncl = int(rs.randint(2, 4))
                b = int(rs.choice([9, 20, 32, 32]))
                wide = [(1024, 1024), (1000, 520), (896, 1024), (800, 300), (128, 1024), (1024, 512), (2048, 1024)]
                shapes = [wide[int(rs.randint(len(wide)))] for _ in range(ncl)]
                steps = (10 * max(k for _, k in shapes)) // b + int(rs.randint(2, 12))  # a few synced steps beyond the longest warm-up
            xs = [mixture(steps * b, d, k, 3.0) for d, k in shapes]
            s = int(rs.randint(1 << 30))
            acav100m_amd.manual_seed(s)
            kms = [KMeans(None, d, k).to("cuda:0") for d, k in shapes]
            rng = O.Rng(s)
            refs = [O.KMeans(d, k, rng) for d, k in shapes]  # one generator: centres drawn clustering by clustering, as above
            xts = [torch.from_numpy(x).cuda() for x in xs]
            for _ in range(2):
                KMeans.train_epoch_multi(kms, xts, b, lr=0.01)  # warm-up labels drawn clustering by clustering
                for ref, x in zip(refs, xs):
                    ref.train_epoch(x, b, lr=0.01)
            for km, ref, sh in zip(kms, refs, shapes):
                assert np.array_equal(km.centers.numpy(), ref.centers) and np.array_equal(km.counts.numpy(), ref.counts), ("train_multi", shapes, b, steps, s, sh)
            counts["train_multi"] += 1
        elif which == 6:  # the DDP epoch of several clusterings through the C-ABI communicator (a world of one: the row exchange,
            # the interleave kernel, chunking and warm-up hand-over are all on the path; == plain epochs == the oracle)
            ncl = int(rs.randint(1, 4))
            b = int(rs.choice([16, 32, 32, 64]))
            steps = int(rs.randint(10, 40))
            chunk = int(rs.choice([8, 1024]))
            shapes = [(int(rs.choice([64, 128, 352, 1024])), int(rs.choice([16, 40, 64, 256]))) for _ in range(ncl)]
            xs = [mixture(steps * b + int(rs.randint(0, b)), d, k, 3.0) for d, k in shapes]
            n0 = min(len(x) for x in xs)
            xs = [x[:n0] for x in xs]
            s = int(rs.randint(1 << 30))
            acav100m_amd.manual_seed(s)
            kms = [KMeans(None, d, k).to("cuda:0") for d, k in shapes]
            rng = O.Rng(s)
            refs = [O.KMeans(d, k, rng) for d, k in shapes]
            xts = [torch.from_numpy(x).cuda() for x in xs]
            for _ in range(2):
                tr = KMeans.train_epoch_distributed_multi(kms, xts, b, lr=0.01, chunk_steps=chunk)
                for v, km in enumerate(kms):
                    km.broadcast_state_from(tr[v], comm_slot=v)
                for ref, x in zip(refs, xs):
                    ref.train_epoch(x, b, lr=0.01)
            for km, ref, sh in zip(kms, refs, shapes):
                assert np.array_equal(km.centers.numpy(), ref.centers) and km.count == ref.count, ("ddp", shapes, b, steps, chunk, s, sh)
            counts["ddp"] += 1
        else:
            v = int(rs.choice([60, 300, 1000, 5000, 5000, 30000]))
            dd = int(rs.choice([2, 3, 5, 10, 13]))
            c = int(rs.choice([2, 8, 40, 256, 700]))
            if c > 256 and dd > 5:
                dd = 3  # P C^2 table cells
            comp = rs.randint(0, c, v)
            a = np.stack([np.where(rs.rand(v) < 0.5, comp, rs.randint(0, c, v)) for _ in range(dd)], 1).astype(np.int64)
            a[0] = c - 1
            pairs = list(itertools.combinations(range(dd), 2))
            cand = rs.permutation(v)
            if which == 2:
                B = int(rs.choice([1, 2, 4, 20, 33, 64])); kk = int(rs.randint(1, B + 1)); keep = bool(rs.randint(0, 2))
                ns = int(rs.choice([1, 1, 3]))                       # start set (seeds the tables, never selected)
                L = v - ns if rs.rand() < 0.6 else max(2, int((v - ns) * rs.uniform(0.3, 1.0)))  # not every clip is a candidate
                B = min(B, L - 1); kk = min(kk, B)
                if B < 1:
                    continue
                subset = int(rs.randint(1, max(2, L // 5)))
                iters = (subset + kk - 1) // kk
                if L - (iters - 1) * (kk if keep else B) < B:
                    continue  # the list would run out of candidates (the reference raises there too: tests/test_gpu_mi.py)
                s = int(rs.randint(1 << 30))
                m = get_measure("batch_mi")(a, ncentroids=c, batch_size=B, selection_size=kk, device="cuda:0",
                                            keep_unselected=keep, generator=Generator(s))
                m.init(pairs, [int(i) for i in cand[ns:ns + L]])
                S, G, _, _ = m.run_greedy(subset, [int(i) for i in cand[:ns]], None)
                r = O.BatchMI(a, c, pairs).run_greedy(cand[ns:ns + L], cand[:ns], subset, B, m.k, O.Rng(s), keep_unselected=keep)
                assert S == r["S"].tolist() and np.array_equal(np.array(G), r["GAIN"]), ("mi", v, dd, c, B, kk, keep, subset, ns, L, s)
                counts["mi"] += 1
            else:
                if v > 5000:
                    continue  # the exact greedy scores every remaining candidate per pick: small lists only
                subset = int(rs.randint(2, max(3, v // 4)))
                m = get_measure("mi")(a, ncentroids=c, device="cuda:0")
                m.init(pairs, [int(i) for i in cand[1:]])
                S, G, _, _ = m.run_greedy(subset, [int(cand[0])])
                r = O.BatchMI(a, c, pairs).run_exact(cand[1:], cand[:1], subset)
                assert S[1:] == r["S"].tolist() and np.array_equal(np.array(G), r["GAIN"]), ("exact", v, dd, c, subset)
                counts["exact"] += 1
    return counts


if __name__ == "__main__":
    res = stress(int(sys.argv[2]) if len(sys.argv) > 2 else 0, float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
    print("stress ok", res, flush=True)
    print("stress ok", res, file=sys.stderr, flush=True)  # (RCCL prints its banner to stdout at exit: the last stdout line is not ours)
