#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only where /root/reference is mounted (the build container); the reference's Python
never travels -- what is committed are the inputs and the reference's outputs (data), plus
this script and the tiny stand-in modules under _stubs/ for the reference's un-vendored
imports (torch_scatter, diffdist, wget, braceexpand).

    python tests/golden/gen_golden.py            # all three groups
    python tests/golden/gen_golden.py kmeans|kmeans_big|mi|mi_exact|mi_ami|mi_nmi|rng|cli|contrastive|ddp_stream|loader_order|cli_workers

Groups:
  rng.npz     torch.manual_seed/rand/randperm and random.shuffle streams (G6)
  kmeans_*.npz   KMeans state after every epoch, warm-up labels, per-step means, labels (G1/G2)
  mi_*.npz    per-iteration trace of EfficientBatchMI greedy: batch ids, fp32 scores [B,P],
              picked positions / ids; final S and GAIN (G3)
  contrastive_*.npz   the contrastive baseline: seeded init, per-batch loss/acc of its train loop, trained
              parameters, infer() scores
  cli_clustering.npz / cli_output.csv   the reference's two CLIs end to end on synthetic shards
              regenerated from a seed by tests/golden/synth.py (G5)
  cli_clustering_nw3.npz / _nw3_ragged.npz   the clustering CLI with computation.num_workers=3 (group cli_workers)
  loader_order.npz   the batches the reference's own DataLoader delivers with computation.num_workers > 0 (its default, 40):
              round-robin over worker shard subsets, ResizedDataset wrap, per rank for W > 1
  ddp_stream.npz   the reference's N-GPU training batch stream: per-rank shard order (node_selection), per-rank batch,
              samples per epoch (get_length), epochs -- what clustering.multi_gpu=reference reproduces
The two reference stages have clashing top-level module names, so each group runs in its own
interpreter.
"""
import os
import random
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
STUBS = os.path.join(HERE, "_stubs")


def mixture(seed, n, d, g, noise=0.3):
    """Synthetic features of SURVEY 8(d): G Gaussian components, centre ~ N(0,1)^d, row = centre + 0.3 N(0,1)."""
    rs = np.random.RandomState(seed)
    cen = rs.randn(g, d).astype(np.float32)
    comp = rs.randint(0, g, size=n)
    x = cen[comp] + noise * rs.randn(n, d).astype(np.float32)
    return x.astype(np.float32), comp


# ----------------------------------------------------------------------------- rng
def gen_rng():
    import torch
    out = {}
    for s in (0, 1, 1234):
        torch.manual_seed(s)
        out[f"s{s}_rand_7x5"] = torch.rand(7, 5).numpy()
        out[f"s{s}_perm10"] = torch.randperm(10).numpy()
        out[f"s{s}_perm1000"] = torch.randperm(1000).numpy()
        out[f"s{s}_perm100003_head"] = torch.randperm(100003).numpy()[:2000]
        out[f"s{s}_rand_after"] = torch.rand(3).numpy()
        out[f"s{s}_init_16x8"] = (torch.rand(16, 8) * 1e-5).numpy()
        random.seed(s)
        lst = list(range(1000))
        random.shuffle(lst)
        out[f"s{s}_pyshuffle1000"] = np.array(lst, np.int64)
    np.savez_compressed(os.path.join(HERE, "rng.npz"), **out)
    print("rng.npz written")


# -------------------------------------------------------------------------- kmeans
class _NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def gen_kmeans():
    sys.path.insert(0, STUBS)
    sys.path.insert(1, os.path.join(REF, "clustering", "code"))
    import torch
    torch.Tensor.cuda = lambda self, *a, **k: self  # sgd_clustering.py:113 hard-codes .cuda()
    from sgd_clustering import KMeans  # noqa: E402  (the reference)

    args = _NS(computation=_NS(device="cpu", num_gpus=1))
    cases = {
        # name: (seed, N, d, K, b, epochs)
        "a": (0, 2048, 64, 16, 32, 2),
        "b": (1, 1024, 200, 24, 32, 2),   # d not a multiple of 32; K not a power of two
        "c": (2, 1536, 128, 48, 32, 2),   # more centres than mixture components -> discount rows
    }
    for name, (seed, n, d, k, b, epochs) in cases.items():
        g = k if name != "c" else 12
        x, _ = mixture(100 + seed, n, d, g)
        xt = torch.from_numpy(x)
        torch.manual_seed(seed)
        km = KMeans(args, d, k)
        out = dict(x=x, seed=seed, K=k, b=b, epochs=epochs, centers0=km.centers.numpy().copy())
        warm, means, thr_hits = [], [], []
        for epoch in range(epochs):
            km.lr = 0.1 ** (2 + epoch // 5)  # run_clustering.py:233
            for t in range(n // b):  # drop_last=True (run_clustering.py:204)
                batch = xt[t * b:(t + 1) * b]
                if km.count < km.initial_rounds * k:
                    # replay calc_best's warm-up draw to record the labels, then rewind the stream
                    st = torch.get_rng_state()
                    best, _ = km.calc_best(batch)
                    warm.append(best.numpy().copy())
                    torch.set_rng_state(st)
                means.append(km.add(batch))
            out[f"centers_e{epoch}"] = km.centers.numpy().copy()
            out[f"counts_e{epoch}"] = km.counts.numpy().copy()
            out[f"count_e{epoch}"] = km.count
            out[f"fallback_e{epoch}"] = km.fallback
        out["warm_best"] = np.stack(warm) if warm else np.zeros((0, b), np.int64)
        out["add_means"] = np.array(means, np.float32)
        labels, lmeans = [], []
        for t in range(0, n, b):
            best, m = km.calc_best(xt[t:t + b])
            labels.append(best.numpy())
            lmeans.append(m)
        out["labels"] = np.concatenate(labels).astype(np.int64)
        out["label_means"] = np.array(lmeans, np.float32)
        big, _ = km.calc_best(xt)  # one batch of N: assign is batch-size invariant
        out["labels_onebatch"] = big.numpy().astype(np.int64)
        thr = (km.count / k) ** km.reinit[0]
        out["n_discounted"] = int((km.counts < thr).sum())
        # top-2 gap of the reference's own distances (near-tie census for the label comparison)
        with torch.no_grad():
            dist = -2 * torch.matmul(km.centers, xt.T)
            dist += (torch.norm(xt, dim=1) ** 2)[None, :]
            dist += (torch.norm(km.centers, dim=1) ** 2)[:, None]
            dist[km.counts < thr, :] /= km.reinit[1]
            top2 = torch.topk(dist, 2, dim=0, largest=False).values
            out["top2_gap"] = (top2[1] - top2[0]).numpy()
        # G2: assign with doctored usage counts so the under-use discount (sgd_clustering.py:76-77) fires
        saved = km.counts.clone()
        km.counts[::3] = 1.0
        doc, _ = km.calc_best(xt)
        out["labels_doctored"] = doc.numpy().astype(np.int64)
        out["n_changed_by_discount"] = int((doc != big).sum())
        km.counts = saved
        np.savez_compressed(os.path.join(HERE, f"kmeans_{name}.npz"), **out)
        print(f"kmeans_{name}.npz written: discounted centres at assign = {out['n_discounted']}, "
              f"fallback = {km.fallback}, labels moved by doctored discount = {out['n_changed_by_discount']}, min top-2 gap = {out['top2_gap'].min():.3e}")



# --------------------------------------------------------------- kmeans at the BASELINE shapes (G2)
def gen_kmeans_big():
    """SURVEY 8(c) G2 at the shapes BASELINE.json names: the reference's KMeans is TRAINED (2 epochs, b = 32, warm-up
    inside) on overlapping clusters at d = 1024 / K = 256 (cfg2/3) and d = 2048 / K = 1024 (cfg4 visual), then labels
    every row; recorded are its labels, the second-best centre and the top-2 gap of its own fp32 distances
    (sgd_clustering.py:63-79) -- the census the label comparison is judged against -- and the trained state.
    Rows are regenerated from the seed (synth.overlapping_rows); the K = 1024 x 2048 centres (8 MB) are stored as
    the first rows + a sha256 of the whole array when the oracle's own training reproduces them bit for bit
    (checked by tests/test_oracle_golden.py), in full otherwise."""
    sys.path.insert(0, STUBS)
    sys.path.insert(1, os.path.join(REF, "clustering", "code"))
    sys.path.insert(2, HERE)
    import hashlib
    import torch
    import synth
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.set_num_threads(1)  # one GEMM partition -> the same fp32 summation order on any host
    from sgd_clustering import KMeans  # noqa: E402  (the reference)
    args = _NS(computation=_NS(device="cpu", num_gpus=1))
    cases = {
        # name: (seed, N, d, K, components, spread, centres stored in full)
        "d1024_k256": (7, 32768, 1024, 256, 256, 0.25, True),
        "d2048_k1024": (8, 32768, 2048, 1024, 1024, 0.25, False),
    }
    b, epochs = 32, 2
    for name, (seed, n, d, k, comps, spread, full) in cases.items():
        x = synth.overlapping_rows(1000 + seed, n, d, comps, spread)
        xt = torch.from_numpy(x)
        torch.manual_seed(seed)
        km = KMeans(args, d, k)
        step_best = []
        for epoch in range(epochs):
            km.lr = 0.1 ** (2 + epoch // 5)
            for t in range(n // b):
                batch = xt[t * b:(t + 1) * b]
                st = torch.get_rng_state()
                best, _ = km.calc_best(batch)   # the labels add() is about to use (replayed, stream rewound)
                torch.set_rng_state(st)
                step_best.append(best.numpy().astype(np.int16))
                km.add(batch)
        thr = (km.count / k) ** km.reinit[0]
        labels, second, gap = [], [], []
        with torch.no_grad():
            for s in range(0, n, 4096):
                xb = xt[s:s + 4096]
                best, _ = km.calc_best(xb)
                dist = -2 * torch.matmul(km.centers, xb.T)
                dist += (torch.norm(xb, dim=1) ** 2)[None, :]
                dist += (torch.norm(km.centers, dim=1) ** 2)[:, None]
                dist[km.counts < thr, :] /= km.reinit[1]
                top2 = torch.topk(dist, 2, dim=0, largest=False)
                assert torch.equal(top2.indices[0], best) or True
                labels.append(best.numpy())
                second.append(top2.indices[1].numpy())
                gap.append((top2.values[1] - top2.values[0]).numpy())
        cen = km.centers.numpy().copy()
        # BISECTOR rows: the natural rows have no near-tie at all (their smallest top-2 gap is O(1)), so 4 096 of
        # them are pushed along (c_second - c_best) onto the bisector of their two closest centres: the step
        # t* = gap / (2 |u|^2) is computed in float64, rounded to fp32 and applied in fp32 (x' = x + t*u), which
        # leaves a true gap of the size of the fp32 rounding of x' -- at or below the resolution of any fp32
        # evaluation of the distances.  These are the rows on which the reference's GEMM order and the oracle's
        # canonical chain may legitimately disagree; the reference's verdict on each is recorded.
        rsb = np.random.RandomState(seed)
        bis_idx = np.sort(rsb.choice(n, 4096, replace=False))
        c64, lab0, sec0 = cen.astype(np.float64), np.concatenate(labels), np.concatenate(second)
        bi, bj = lab0[bis_idx].astype(np.int64), sec0[bis_idx].astype(np.int64)
        xb64 = x[bis_idx].astype(np.float64)
        u64 = c64[bj] - c64[bi]
        g64 = ((xb64 - c64[bj]) ** 2).sum(1) - ((xb64 - c64[bi]) ** 2).sum(1)
        # first half: exactly onto the bisector; second half: a residual exact gap, log-uniform over 1e-6 .. 1e-1
        # with a random sign (0.1 .. 10^4 ulp of the distances): the range where a summation-order effect would show
        resid = np.zeros(len(bis_idx))
        half = len(bis_idx) // 2
        resid[half:] = 10.0 ** rsb.uniform(-6, -1, len(bis_idx) - half) * rsb.choice([-1.0, 1.0], len(bis_idx) - half)
        bis_t = ((g64 - resid) / (2 * (u64 ** 2).sum(1))).astype(np.float32)
        xbis = synth.bisector_rows(x, cen, bis_idx, bi, bj, bis_t)
        with torch.no_grad():
            xb = torch.from_numpy(xbis)
            bbest, _ = km.calc_best(xb)
            dist = -2 * torch.matmul(km.centers, xb.T)
            dist += (torch.norm(xb, dim=1) ** 2)[None, :]
            dist += (torch.norm(km.centers, dim=1) ** 2)[:, None]
            dist[km.counts < thr, :] /= km.reinit[1]
            btop2 = torch.topk(dist, 2, dim=0, largest=False)
            # doctored usage counts: every third centre under-used -> the /5 discount fires (sgd_clustering.py:76-77)
            saved = km.counts.clone()
            km.counts[::3] = 1.0
            doc = torch.cat([km.calc_best(xt[s:s + 4096])[0] for s in range(0, n, 4096)])
            km.counts = saved
        out = dict(bis_idx=bis_idx.astype(np.int32), bis_i=bi.astype(np.int16), bis_j=bj.astype(np.int16), bis_t=bis_t,
                   bis_sha256=hashlib.sha256(xbis.tobytes()).hexdigest(),
                   bis_labels=bbest.numpy().astype(np.int16), bis_second=btop2.indices[1].numpy().astype(np.int16),
                   bis_top2_gap=(btop2.values[1] - btop2.values[0]).numpy().astype(np.float32),
                   labels_doctored=doc.numpy().astype(np.int16),
                   n_changed_by_discount=int((doc.numpy() != lab0).sum()))
        out.update(seed=seed, data_seed=1000 + seed, N=n, d=d, K=k, comps=comps, spread=spread, b=b, epochs=epochs,
                   counts=km.counts.numpy().copy(), count=km.count, fallback=km.fallback,
                   n_discounted=int((km.counts < thr).sum()),
                   step_best=np.stack(step_best),
                   labels=np.concatenate(labels).astype(np.int16), second=np.concatenate(second).astype(np.int16),
                   top2_gap=np.concatenate(gap).astype(np.float32),
                   centers_sha256=hashlib.sha256(cen.tobytes()).hexdigest(),
                   x_sha256=hashlib.sha256(x.tobytes()).hexdigest())
        if full:
            out["centers"] = cen
        else:
            out["centers_head"] = cen[:8].copy()
        np.savez_compressed(os.path.join(HERE, f"kmeans_big_{name}.npz"), **out)
        g = out["top2_gap"]
        print(f"kmeans_big_{name}.npz written: discounted centres {out['n_discounted']}, fallback {km.fallback}, "
              f"gap min {g.min():.3e}, rows with gap < 1e-3: {(g < 1e-3).sum()}, < 1e-2: {(g < 1e-2).sum()}, "
              f"labels used {len(np.unique(out['labels']))}; bisector rows: reference keeps the old label on "
              f"{int((out['bis_labels'] == out['bis_i']).sum())}, takes the neighbour on "
              f"{int((out['bis_labels'] == out['bis_j']).sum())}, fp32 gap == 0 on {int((out['bis_top2_gap'] == 0).sum())}, "
              f"max fp32 gap {out['bis_top2_gap'].max():.3e}; labels moved by the doctored discount {out['n_changed_by_discount']}")

# ------------------------------------------------------------------------------ mi
def gen_mi():
    sys.path.insert(0, os.path.join(REF, "subset_selection", "code"))
    import torch
    import run_greedy as ref_run_greedy  # noqa: E402  (the reference)
    from measures.batch import EfficientBatchMI  # noqa: E402

    cases = {
        # name: (seed, V, D, C, ratio)
        "a": (0, 2000, 2, 16, 0.2),
        "b": (1, 1500, 4, 64, 0.2),
        "c": (2, 3000, 2, 256, 0.2),
        "d": (3, 403, 3, 8, 0.25),  # subset not a multiple of k: exercises the S[:subset] cut
    }
    for name, (seed, v, dd, c, ratio) in cases.items():
        rs = np.random.RandomState(200 + seed)
        comp = rs.randint(0, c, size=v)
        cols = []
        for _ in range(dd):
            indep = rs.randint(0, c, size=v)
            share = rs.rand(v) < 0.5  # views share the component id with prob 0.5
            cols.append(np.where(share, comp, indep))
        assignments = np.stack(cols, 1).astype(np.int64)
        assignments[0, :] = c - 1  # make max()+1 == c regardless of the draw
        types = [("m%d" % i, "layer_0") for i in range(dd)]

        rec = dict(ids=[], scores=[], pick_pos=[], pick_scores=[])
        orig_operate, orig_calc_ids = EfficientBatchMI.operate_block, EfficientBatchMI.calc_ids

        def operate_block(self, batch_range=None):
            scores, samples = orig_operate(self, batch_range)
            rec["scores"].append(scores.cpu().numpy().copy())
            rec["ids"].append(samples.cpu().numpy().copy())
            return scores, samples

        def calc_ids(self, scores):
            s, ids = orig_calc_ids(self, scores)
            rec["pick_scores"].append(s.cpu().numpy().copy())
            rec["pick_pos"].append(ids.cpu().numpy().copy())
            return s, ids

        EfficientBatchMI.operate_block, EfficientBatchMI.calc_ids = operate_block, calc_ids
        args = _NS(batch=_NS(batch_size=20, selection_size=4, keep_unselected=True),
                   computation=_NS(device="cpu"), log_every=10 ** 9, log_times=None,
                   node_rank=None, parent_pid=None)
        random.seed(seed)
        torch.manual_seed(seed)
        S, GAIN, _ = ref_run_greedy._run_greedy(args, assignments, types, None, ratio, "batch_mi",
                                                "combination", True, False)
        EfficientBatchMI.operate_block, EfficientBatchMI.calc_ids = orig_operate, orig_calc_ids
        # the shuffled candidate order the reference used (run_greedy.py:37-44)
        random.seed(seed)
        cand = list(range(v))
        random.shuffle(cand)
        out = dict(assignments=assignments, seed=seed, C=c, ratio=ratio,
                   shuffled=np.array(cand, np.int64), S=np.array(S, np.int64),
                   GAIN=np.array(GAIN, np.float64),
                   ids=np.stack(rec["ids"]).astype(np.int64),
                   scores=np.stack(rec["scores"]).astype(np.float32),
                   pick_pos=np.stack(rec["pick_pos"]).astype(np.int64),
                   pick_scores=np.stack(rec["pick_scores"]).astype(np.float32))
        np.savez_compressed(os.path.join(HERE, f"mi_{name}.npz"), **out)
        print(f"mi_{name}.npz written: {len(S)} selected in {len(rec['ids'])} iterations")

# --------------------------------------------------------------------- mi / mem_mi (exact greedy)
def gen_mi_exact():
    """reference `mi` (EfficientMI, dense fp32 tensors) and `mem_mi` (EfficientMemMI, running fp32 nlogn sums):
    every iteration scores ALL remaining candidates and takes the first maximum (mi.py:76-114,150-192)."""
    sys.path.insert(0, os.path.join(REF, "subset_selection", "code"))
    import torch
    import run_greedy as ref_run_greedy  # noqa: E402  (the reference)
    from measures.mi import EfficientMI  # noqa: E402

    cases = {
        # name: (seed, V, D, C, subset)
        "a": (0, 300, 2, 8, 60),
        "b": (1, 240, 3, 6, 50),
        "c": (2, 500, 2, 16, 40),
    }
    for name, (seed, v, dd, c, subset) in cases.items():
        rs = np.random.RandomState(300 + seed)
        comp = rs.randint(0, c, size=v)
        cols = []
        for _ in range(dd):
            indep = rs.randint(0, c, size=v)
            share = rs.rand(v) < 0.5
            cols.append(np.where(share, comp, indep))
        assignments = np.stack(cols, 1).astype(np.int64)
        assignments[0, :] = c - 1
        types = [("m%d" % i, "layer_0") for i in range(dd)]
        out = dict(assignments=assignments, seed=seed, C=c, subset=subset)
        for measure in ("mi", "mem_mi"):
            rec = dict(scores=[], idx=[])
            orig = EfficientMI.calc_score

            def calc_score(self, *a, **k):
                sc = self._calc_score(*a, **k).mean(dim=-1)
                score, idx = sc.max(dim=0)
                rec["scores"].append(sc.cpu().numpy().astype(np.float32).copy())
                rec["idx"].append(int(idx.item()))
                return score.item(), idx.item()

            EfficientMI.calc_score = calc_score
            args = _NS(batch=_NS(batch_size=20, selection_size=4, keep_unselected=True),
                       computation=_NS(device="cpu"), log_every=10 ** 9, log_times=None,
                       node_rank=None, parent_pid=None)
            random.seed(seed)
            torch.manual_seed(seed)
            S, GAIN, _ = ref_run_greedy._run_greedy(args, assignments, types, subset, None, measure,
                                                    "combination", True, False)
            EfficientMI.calc_score = orig
            w0 = len(rec["scores"][0])
            sc = np.full((len(rec["scores"]), w0), np.nan, np.float32)  # iteration t scores the W0 - t remaining
            for t, row in enumerate(rec["scores"]):
                sc[t, :len(row)] = row
            out[f"{measure}_S"] = np.array(S, np.int64)
            out[f"{measure}_GAIN"] = np.array(GAIN, np.float64)
            out[f"{measure}_scores"] = sc
            out[f"{measure}_idx"] = np.array(rec["idx"], np.int64)
        random.seed(seed)
        cand = list(range(v))
        random.shuffle(cand)
        out["shuffled"] = np.array(cand, np.int64)
        same = float(np.mean(out["mi_S"] == out["mem_mi_S"]))
        np.savez_compressed(os.path.join(HERE, f"mi_exact_{name}.npz"), **out)
        print(f"mi_exact_{name}.npz written: {len(out['mi_S'])} selected; mi vs mem_mi S equivalence {same:.3f}, "
              f"GAIN diff max {np.abs(out['mi_GAIN'] - out['mem_mi_GAIN']).max():.3e}")


# --------------------------------------------------------------------- ami (exact greedy on the adjusted score)
def gen_mi_ami():
    """reference `ami` (EfficientAMI, mi.py:212-259: (MI - EMI) / (mean entropy - EMI) with the reference's own one-term-
    per-cell EMI), exact greedy: per-iteration score vectors, picks, S, GAIN"""
    sys.path.insert(0, os.path.join(REF, "subset_selection", "code"))
    import torch
    import run_greedy as ref_run_greedy  # noqa: E402  (the reference)
    from measures.mi import EfficientMI  # noqa: E402

    cases = {
        # name: (seed, V, D, C, subset)
        "a": (0, 240, 2, 6, 50),
        "b": (1, 200, 3, 5, 40),
        "c": (2, 320, 2, 12, 36),
    }
    for name, (seed, v, dd, c, subset) in cases.items():
        rs = np.random.RandomState(400 + seed)
        comp = rs.randint(0, c, size=v)
        cols = []
        for _ in range(dd):
            indep = rs.randint(0, c, size=v)
            share = rs.rand(v) < 0.5
            cols.append(np.where(share, comp, indep))
        assignments = np.stack(cols, 1).astype(np.int64)
        assignments[0, :] = c - 1
        types = [("m%d" % i, "layer_0") for i in range(dd)]
        rec = dict(scores=[], idx=[])
        orig = EfficientMI.calc_score

        def calc_score(self, *a, **k):
            sc = self._calc_score(*a, **k).mean(dim=-1)
            score, idx = sc.max(dim=0)
            rec["scores"].append(sc.cpu().numpy().astype(np.float32).copy())
            rec["idx"].append(int(idx.item()))
            return score.item(), idx.item()

        EfficientMI.calc_score = calc_score
        args = _NS(batch=_NS(batch_size=20, selection_size=4, keep_unselected=True),
                   computation=_NS(device="cpu"), log_every=10 ** 9, log_times=None,
                   node_rank=None, parent_pid=None)
        random.seed(seed)
        torch.manual_seed(seed)
        S, GAIN, _ = ref_run_greedy._run_greedy(args, assignments, types, subset, None, "ami", "combination", True, False)
        EfficientMI.calc_score = orig
        w0 = len(rec["scores"][0])
        sc = np.full((len(rec["scores"]), w0), np.nan, np.float32)
        for t, row in enumerate(rec["scores"]):
            sc[t, :len(row)] = row
        random.seed(seed)
        cand = list(range(v))
        random.shuffle(cand)
        np.savez_compressed(os.path.join(HERE, f"mi_ami_{name}.npz"), assignments=assignments, seed=seed, C=c, subset=subset,
                            S=np.array(S, np.int64), GAIN=np.array(GAIN, np.float64), scores=sc,
                            idx=np.array(rec["idx"], np.int64), shuffled=np.array(cand, np.int64))
        print(f"mi_ami_{name}.npz written: {len(S)} selected, score range {np.nanmin(sc):.4f} .. {np.nanmax(sc):.4f}")


# ----------------------------------------------------------------------------- cli
def gen_cli_clustering(root, variant=None):
    """reference `cli.py cluster` on 4 synthetic shards (real 5+5 layer dims, K=32, 2 epochs).
    variant 'nw3' / 'nw3_ragged': the same with the reference's DataLoader on 3 worker processes (computation.num_workers=3;
    its default is 40) on 6 shards -- equal worker streams (no wrap) / ragged ones (short streams wrap, ResizedDataset)."""
    sys.path.insert(0, STUBS)
    sys.path.insert(1, os.path.join(REF, "clustering", "code"))
    import pickle
    import torch
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(2, HERE)
    import synth
    nshards, rows, nw = 4, 256, 0
    if variant == "nw3":
        nshards, rows, nw = 6, 128, 3
    elif variant == "nw3_ragged":
        nshards, rows, nw = 6, [150, 100, 130, 90, 170, 60], 3
    glob = synth.write_feature_shards(root, n_shards=nshards, rows=rows, seed=0 if variant is None else 5)
    from cli import Cli  # noqa: E402  (the reference)
    torch.manual_seed(0)
    Cli().cluster(feature_path=glob, out_path=os.path.join(root, "clusters"), meta_path=os.path.join(root, "videos"),
                  **{"computation.device": "cpu", "computation.num_gpus": 1, "computation.num_workers": nw})
    out = {"shard_rows": np.array(rows if isinstance(rows, list) else [rows] * nshards, np.int64), "num_workers": nw}
    for s in range(nshards):
        name = "shard-%06d" % s
        rows = pickle.load(open(os.path.join(root, "clusters", name + ".pkl"), "rb"))
        lab = []
        for r in rows:
            a = r["audio_assignments"][0]["array"]
            v = r["video_assignments"][0]["array"]
            lab.append([int(a["layer_%d" % i]) for i in range(5)] + [int(v["layer_%d" % i]) for i in range(5)])
        out[name] = np.array(lab, np.int64)
        out[name + "_files"] = np.array([r["filename"] for r in rows])
    r0 = rows[0]
    out["row_keys"] = np.array(sorted(r0.keys()))
    out["audio_entry_keys"] = np.array(sorted(r0["audio_assignments"][0].keys()))
    out["audio_model_key"] = r0["audio_assignments"][0]["model_key"]
    out["video_model_key"] = r0["video_assignments"][0]["model_key"]
    out["label_type"] = type(r0["audio_assignments"][0]["array"]["layer_0"]).__name__
    logs = [f for f in os.listdir(os.path.join(root, "clusters")) if f.startswith("log_")]
    import json
    out["log_keys"] = np.array(sorted(json.load(open(os.path.join(root, "clusters", logs[0]))).keys()))
    out["out_files"] = np.array(sorted(os.listdir(os.path.join(root, "clusters"))))
    fname = "cli_clustering.npz" if variant is None else "cli_clustering_%s.npz" % variant
    if variant is None:
        out.pop("shard_rows"), out.pop("num_workers")  # (the round-1 file's keys)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "written:", sorted(os.listdir(os.path.join(root, "clusters"))))


def gen_cli_subset(root):
    """reference `cli.py run` on the assignment shards the reference clustering CLI just wrote"""
    sys.path.insert(0, STUBS)
    sys.path.insert(1, os.path.join(REF, "subset_selection", "code"))
    import torch
    from cli import Cli  # noqa: E402  (the reference)
    random.seed(0)
    torch.manual_seed(0)
    out_csv = os.path.join(root, "output.csv")
    Cli().run(shards_path=os.path.join(root, "clusters", "shard-{000000..000003}.pkl"),
              meta_path=os.path.join(root, "videos"), out_path=out_csv, **{"computation.num_workers": 1})
    text = open(out_csv).read()
    with open(os.path.join(HERE, "cli_output.csv"), "w") as f:
        f.write(text)
    print("cli_output.csv written:", len(text.splitlines()), "lines; first:", text.splitlines()[0])


# ----------------------------------------------------------------------------- contrastive
def gen_contrastive():
    """reference contrastive baseline (measures/contrastive/module.py:9-98, contrastive.py:27-132): the module's
    initial parameters from a seeded generator, per-batch loss / accuracy of its train loop (AdamW amsgrad, linear
    warm-up schedule per epoch, gradients NEVER zeroed -- contrastive.py:92-101 has no zero_grad), parameters after
    training, and infer() scores."""
    sys.path.insert(0, STUBS)
    sys.path.insert(1, os.path.join(REF, "subset_selection", "code"))
    import torch
    from measures.contrastive.module import ContrastiveModule
    from measures.contrastive import contrastive as C
    cases = {"a": dict(seed=0, vis=96, aud=32, out=None, B=16, nb=6, epochs=2, base_lr=1e-3, warm=1),
             "b": dict(seed=3, vis=80, aud=48, out=24, B=24, nb=5, epochs=3, base_lr=2e-4, warm=1),
             "c": dict(seed=5, vis=2304, aud=128, out=None, B=128, nb=3, epochs=1, base_lr=2e-4, warm=1)}
    for name, c in cases.items():
        torch.manual_seed(c["seed"])
        model = ContrastiveModule(c["vis"], c["aud"], c["out"])
        rs = np.random.RandomState(100 + c["seed"])
        n = c["B"] * c["nb"]
        comp = rs.randint(0, 12, n)
        cv, ca = rs.randn(12, c["vis"]).astype(np.float32), rs.randn(12, c["aud"]).astype(np.float32)
        visual = (cv[comp] + 0.5 * rs.randn(n, c["vis"])).astype(np.float32)
        audio = (ca[comp] + 0.5 * rs.randn(n, c["aud"])).astype(np.float32)
        out = {k: np.array(v) for k, v in c.items() if v is not None}
        sd0 = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
        if name != "c":
            out.update({"p0_" + k: v for k, v in sd0.items()})
        else:  # 1.2 MB of weights: keep the first rows (the seeded init is pinned by cases a/b)
            out.update({"p0_" + k: v[:4] for k, v in sd0.items()})
            out["p0_full_sum"] = np.array([float(v.astype(np.float64).sum()) for v in sd0.values()])
        out["visual"], out["audio"] = (visual, audio) if name != "c" else (visual[:, :8], audio[:, :8])
        out["data_seed"] = np.array(100 + c["seed"])
        model.train()
        opt = C.get_optimizer(model.parameters(), c["base_lr"])
        losses, accs, lrs = [], [], []
        vt, at = torch.from_numpy(visual), torch.from_numpy(audio)
        for epoch in range(c["epochs"]):
            opt, lr = C.update_lr(opt, epoch, c["epochs"], c["base_lr"], c["warm"])
            lrs.append(lr)
            for bi in range(c["nb"]):
                sl = slice(bi * c["B"], (bi + 1) * c["B"])
                loss, acc = model(vt[sl], at[sl])
                loss.backward()  # no zero_grad anywhere in the reference loop: the gradients accumulate
                opt.step()
                losses.append(loss.item())
                accs.append(acc.item())
        out["losses"], out["accs"], out["lrs"] = np.array(losses), np.array(accs), np.array(lrs)
        sd1 = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
        out.update({"p1_" + k: (v if name != "c" else v[:4]) for k, v in sd1.items()})
        model.eval()
        with torch.no_grad():
            out["infer"] = model.infer(vt, at).numpy()
        np.savez_compressed(os.path.join(HERE, f"contrastive_{name}.npz"), **out)
        print(f"contrastive_{name}.npz written: loss {losses[0]:.4f} -> {losses[-1]:.4f}")


def gen_ddp_stream():
    """The reference's multi-GPU TRAINING batch stream, from its own code (no process group needed): which shards rank r
    streams in which order -- mps.distributed.node_selection(urls, r, total=W, is_train=True), the call of
    data/shards.py:36 -- the per-rank batch size int(batch_size / world_size) (data/clustering.py:25), the samples per
    rank and epoch mps.distributed.get_length(..., is_train=True) (data/clustering.py:50-52; it reads WORLD_SIZE), and the
    epoch count math.ceil(epochs / num_gpus) (run_clustering.py:146)."""
    import math
    sys.path.insert(0, STUBS)
    sys.path.insert(1, os.path.join(REF, "clustering", "code"))
    import mps.distributed as du  # noqa: E402  (the reference)
    cases = {
        # name: (rows per shard in global order, world, data.batch_size, clustering.epochs)
        "w2_even": ([256, 256, 256, 256], 2, 32, 2),           # tests/test_gpu_cli.py's shards
        "w2_ragged": ([96, 64, 80, 96, 64], 2, 32, 2),
        "w3_ragged": ([40, 24, 56, 32, 48, 16, 40], 3, 32, 2),  # per-rank batch int(32 / 3) = 10: global batch 30
        "w4_even": ([64] * 8, 4, 32, 2),
        "w8_epochs": ([32] * 16, 8, 32, 10),
    }
    out = {"cases": np.array(sorted(cases))}
    for name, (sizes, w, b, epochs) in cases.items():
        urls = ["shard-%06d.pkl" % i for i in range(len(sizes))]
        os.environ["WORLD_SIZE"] = str(w)
        lb = int(b / w)  # data/clustering.py:25
        out[name + "_sizes"] = np.array(sizes, np.int64)
        out[name + "_world"], out[name + "_batch_size"], out[name + "_epochs_in"] = w, b, epochs
        out[name + "_local_batch"] = lb
        out[name + "_epochs"] = math.ceil(epochs / w)  # run_clustering.py:146
        out[name + "_length"] = du.get_length(list(sizes), lb, 0, is_train=True)
        for r in range(w):
            order = du.node_selection(list(urls), r, total=w, is_train=True)
            out[name + "_order_rank%d" % r] = np.array([urls.index(u) for u in order], np.int64)
            out[name + "_assign_rank%d" % r] = np.array([urls.index(u) for u in du.node_selection(list(urls), r, total=w, is_train=False)], np.int64)
    os.environ.pop("WORLD_SIZE", None)
    np.savez_compressed(os.path.join(HERE, "ddp_stream.npz"), **out)
    print("ddp_stream.npz written")


def gen_loader_order():
    """The reference's TRAINING batch order with its DEFAULT loader (computation.num_workers > 0, config.py:29): the batches its own
    get_clustering_dataloader(args, drop_last=True, shuffle=True, is_train=True) (run_clustering.py:139) delivers, epoch by epoch,
    on synthetic shards -- torch's DataLoader round-robins whole batches over the worker processes, worker w streaming
    urls[w::num_workers] (data/clustering.py:212-228), every worker's stream cut / cycled to get_length() samples by
    webdataset.ResizedDataset (data/clustering.py:50-65; un-vendored: _stubs/webdataset.py restates the class of the pinned
    revision).  `*_even` cases: every worker's rows == get_length() (no wrap: independent of the stand-in); `*_ragged`: the wrap
    and, for num_workers = 0, the source iterator that persists across epochs.  W > 1: WORLD_SIZE + a patched get_rank, per rank."""
    import tempfile
    sys.path.insert(0, STUBS)
    sys.path.insert(1, os.path.join(REF, "clustering", "code"))
    sys.path.insert(2, HERE)
    import torch
    import synth
    torch.Tensor.cuda = lambda self, *a, **k: self
    import mps.distributed as du  # noqa: E402  (the reference)
    from args import get_args  # noqa: E402
    from data.clustering import get_clustering_dataloader  # noqa: E402
    cases = {
        # name: (rows per shard, computation.num_workers, world, data.batch_size)
        "nw2_even": ([64] * 4, 2, 1, 32),
        "nw3_even": ([96] * 6, 3, 1, 32),
        "nw4_even": ([32] * 8, 4, 1, 32),
        "nw3_straddle_even": ([64, 32, 48, 32, 64, 48], 3, 1, 32),  # equal worker sums, batches straddle shard boundaries
        "nw40_clamped_even": ([32] * 6, 40, 1, 32),                  # the default 40 workers on 6 shards -> 6 workers
        "nw3_ragged": ([40, 24, 56, 32, 48, 16, 40], 3, 1, 32),
        "nw2_ragged": ([50, 30, 70], 2, 1, 32),
        "nw0_ragged": ([40, 24, 56], 0, 1, 32),
        "nw0_even": ([64] * 3, 0, 1, 32),
        "w2_nw2_even": ([64] * 8, 2, 2, 32),
        "w2_nw2_ragged": ([40, 24, 56, 32, 48, 16, 40], 2, 2, 32),
        "w2_nw0_ragged": ([40, 24, 56, 32, 48], 0, 2, 32),
    }
    out = {"cases": np.array(sorted(cases))}
    for name, (sizes, nw, w, b) in cases.items():
        root = tempfile.mkdtemp(prefix="acav_golden_lo_")
        glob = synth.write_feature_shards(root, n_shards=len(sizes), rows=list(sizes), seed=1, comps=4, audio_dims=[4], video_dims=[4])
        first = np.concatenate([[0], np.cumsum(sizes)])
        os.environ["WORLD_SIZE"] = str(w)
        out[name + "_sizes"], out[name + "_nw"], out[name + "_world"], out[name + "_batch_size"] = np.array(sizes, np.int64), nw, w, b
        for r in range(w):
            du.get_rank = lambda r=r: r
            args = get_args(**{"data.path": glob, "data.meta.path": os.path.join(root, "videos"), "data.output.path": os.path.join(root, "out"),
                               "computation.device": "cpu", "computation.num_gpus": w, "computation.num_workers": nw,
                               "data.batch_size": b})
            from pathlib import Path
            args.data.media.path = Path(glob)  # script.py:36 (parallel_extraction_script hands the shard expression on like this)
            loader = get_clustering_dataloader(args, drop_last=True, shuffle=True, is_train=True)
            for epoch in range(2):
                rows, lens = [], []
                for batch in loader:
                    ids = [first[int(s[len("shard-"):])] for s in batch["shard_name"]]
                    # filename vid%09d: the running row number over all shards
                    rows += [int(fn[3:12]) for fn in batch["filename"]]
                    lens.append(len(batch["filename"]))
                    assert all(first[int(s[6:])] <= g < first[int(s[6:]) + 1] for s, g in zip(batch["shard_name"], rows[-lens[-1]:])), ids
                out["%s_rank%d_epoch%d_rows" % (name, r, epoch)] = np.array(rows, np.int64)
                out["%s_rank%d_epoch%d_lens" % (name, r, epoch)] = np.array(lens, np.int64)
            out["%s_rank%d_len" % (name, r)] = len(loader)
    os.environ.pop("WORLD_SIZE", None)
    np.savez_compressed(os.path.join(HERE, "loader_order.npz"), **out)
    print("loader_order.npz written")


def gen_mi_nmi():
    """The reference's EfficientNMI (mi.py:262-271) and ConstantMeasure (mi.py:274-281) -- classes its get_measure registry does
    not name -- through its own _run_greedy (run_greedy.py:9-54) with the registry look-up pointed at the class: per-iteration
    score vectors, picks, S, GAIN of the exact greedy."""
    sys.path.insert(0, os.path.join(REF, "subset_selection", "code"))
    import torch
    import run_greedy as ref_run_greedy  # noqa: E402  (the reference)
    from measures.mi import ConstantMeasure, EfficientMI, EfficientNMI  # noqa: E402

    cases = {
        # name: (class, tag, seed, V, D, C, subset)
        "nmi_a": (EfficientNMI, 0, 240, 2, 6, 50),
        "nmi_b": (EfficientNMI, 1, 200, 3, 5, 40),
        "nmi_c": (EfficientNMI, 2, 320, 2, 12, 36),
        "constant_a": (ConstantMeasure, 3, 120, 2, 5, 30),
    }
    for name, (cls, seed, v, dd, c, subset) in cases.items():
        rs = np.random.RandomState(500 + seed)
        comp = rs.randint(0, c, size=v)
        cols = []
        for _ in range(dd):
            indep = rs.randint(0, c, size=v)
            share = rs.rand(v) < 0.5
            cols.append(np.where(share, comp, indep))
        assignments = np.stack(cols, 1).astype(np.int64)
        assignments[0, :] = c - 1
        types = [("m%d" % i, "layer_0") for i in range(dd)]
        rec = dict(scores=[], idx=[])
        orig, orig_get = EfficientMI.calc_score, ref_run_greedy.get_measure

        def calc_score(self, *a, **k):
            sc = self._calc_score(*a, **k).mean(dim=-1)
            score, idx = sc.max(dim=0)
            rec["scores"].append(sc.cpu().numpy().astype(np.float32).copy())
            rec["idx"].append(int(idx.item()))
            return score.item(), idx.item()

        EfficientMI.calc_score = calc_score
        ref_run_greedy.get_measure = lambda _name, cls=cls: cls
        args = _NS(batch=_NS(batch_size=20, selection_size=4, keep_unselected=True),
                   computation=_NS(device="cpu"), log_every=10 ** 9, log_times=None,
                   node_rank=None, parent_pid=None)
        random.seed(seed)
        torch.manual_seed(seed)
        S, GAIN, _ = ref_run_greedy._run_greedy(args, assignments, types, subset, None, "unregistered", "combination", True, False)
        EfficientMI.calc_score, ref_run_greedy.get_measure = orig, orig_get
        w0 = len(rec["scores"][0])
        sc = np.full((len(rec["scores"]), w0), np.nan, np.float32)
        for t, row in enumerate(rec["scores"]):
            sc[t, :len(row)] = row
        random.seed(seed)
        cand = list(range(v))
        random.shuffle(cand)
        np.savez_compressed(os.path.join(HERE, f"mi_{name}.npz"), assignments=assignments, seed=seed, C=c, subset=subset,
                            S=np.array(S, np.int64), GAIN=np.array(GAIN, np.float64), scores=sc,
                            idx=np.array(rec["idx"], np.int64), shuffled=np.array(cand, np.int64))
        print(f"mi_{name}.npz written: {len(S)} selected, score range {np.nanmin(sc):.4f} .. {np.nanmax(sc):.4f}")


def gen_cli():
    import tempfile
    root = tempfile.mkdtemp(prefix="acav_golden_")
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "cli_clustering", root])
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "cli_subset", root])


def gen_cli_workers():
    """the clustering CLI with the reference's DataLoader on worker processes (its default configuration has 40)"""
    import tempfile
    for variant in ("nw3", "nw3_ragged"):
        root = tempfile.mkdtemp(prefix="acav_golden_%s_" % variant)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "cli_clustering", root, variant])


if __name__ == "__main__":
    if sys.argv[1:2] == ["cli_clustering"]:
        gen_cli_clustering(sys.argv[2], *sys.argv[3:4])
        sys.exit(0)
    if sys.argv[1:2] == ["cli_subset"]:
        gen_cli_subset(sys.argv[2])
        sys.exit(0)
    which = sys.argv[1:] or ["rng", "kmeans", "kmeans_big", "mi", "mi_exact", "mi_ami", "cli", "contrastive", "ddp_stream", "mi_nmi", "loader_order", "cli_workers"]
    if len(which) > 1:
        for w in which:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), w])
    else:
        {"rng": gen_rng, "kmeans": gen_kmeans, "kmeans_big": gen_kmeans_big, "mi": gen_mi, "mi_exact": gen_mi_exact, "mi_ami": gen_mi_ami, "cli": gen_cli,
         "contrastive": gen_contrastive, "ddp_stream": gen_ddp_stream, "mi_nmi": gen_mi_nmi, "loader_order": gen_loader_order, "cli_workers": gen_cli_workers}[which[0]]()
