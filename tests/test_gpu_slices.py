"""BASELINE.json configs[3] and configs[4] AT THEIR PER-GPU SIZE on one MI355X (both configs were only ever
run at 12 k / 32 k rows).  An 8-GPU job gives every GPU 1/8 of the clips:

  cfg4 slice   1.25M clips, visual 2048-d + audio 128-d, K = 1024          (11 GB of features)
  cfg5 slice   12.5M clips, two 1024-d views, K = 1024                     (102 GB of features, resident in one GPU's HBM)

Per slice: one training epoch of every view at b = 32 (the persistent epoch kernels -- asserted: no fallback to per-step
launches), both assign paths equal on ALL rows and equal to the oracle on a 16 k-row sample, then the selection the
reference would run on that slice (cfg4: one chunk; cfg5: chunks of 100 shards = 100k clips, 10 in lockstep --
subset_selection/code/chunk.py:21-53) with its size / uniqueness / contingency-table properties, the first chunk equal to
the oracle's run pick for pick.  Data from bench.py's generator, generated on the device block by block."""
import os
import random
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import acav100m_amd
    acav100m_amd.load_library()
    from oracle import oracle as O
    return torch, acav100m_amd, O


def _train_and_label(env, n, dims, K, seed):
    """one epoch of SGD per view at b = 32 (side by side, as run_clustering does), then labels by both assign paths"""
    torch, acav, O = env
    import bench
    from acav100m_amd.clustering import KMeans
    free, _ = torch.cuda.mem_get_info()
    need = sum(n * d * 4 for d in dims) + (4 << 30)
    if free < need:
        pytest.skip(f"needs {need >> 30} GB of free HBM, {free >> 30} GB available")
    xs = bench.synth_views(torch, n, dims, K, seed, "cuda:0")
    acav.manual_seed(seed)
    kms = [KMeans(None, d, K).to("cuda:0") for d in dims]
    KMeans.train_epoch_multi(kms, xs, 32, lr=0.01)
    for km in kms:
        km.synchronize()
        launches, fallbacks = km.train_stats()
        assert launches >= 1 and fallbacks == 0, f"persistent epoch kernel: {launches} launches, {fallbacks} fallbacks"
        assert km.count == (n // 32) * 32
    labels = []
    rs = np.random.RandomState(seed)
    idx = np.sort(rs.choice(n, 16384, replace=False))
    idx_t = torch.from_numpy(idx).cuda()
    for km, x, d in zip(kms, xs, dims):
        fast = km.calc_best(x, need_mean=False)[0]
        launches, rows, undecided = km.filter_stats()
        assert launches >= 1 and rows == n
        exact = km.calc_best(x, need_mean=True)[0]
        assert torch.equal(fast, exact), f"d={d}: filter path != exact sweep on {(fast != exact).sum().item()} of {n} rows"
        ref = O.KMeans(d, K, O.Rng(0))
        ref.set_state(km.centers.numpy(), km.counts.numpy(), km.count)
        want = ref.calc_best(x[idx_t].cpu().numpy())[0]
        assert np.array_equal(fast[idx_t].cpu().numpy(), want), f"d={d}: labels differ from the oracle on the sample"
        print(f"slice n={n} d={d} K={K}: {undecided} rows undecided by the filter; labels use {len(torch.unique(fast))} centres")
        labels.append(fast)
        del exact
    a = torch.stack(labels, 1).cpu().numpy()
    del xs, kms
    torch.cuda.empty_cache()
    return a


def _oracle_selection(O, a, seed, subset, py_seed):
    """the oracle's selection of one chunk, driven like _prepare(): python shuffle of the candidates, singleton start"""
    random.seed(py_seed)
    order = list(range(a.shape[0]))
    random.shuffle(order)
    C = int(a.max()) + 1
    return O.BatchMI(a, C, [(0, 1)]).run_greedy(order[1:], order[:1], subset, 20, 4, O.Rng(seed))


def test_cfg4_slice_full_size(env):
    """configs[3] per GPU: 1.25M x (2048 + 128), K = 1024 -- train, assign (both paths, all rows), select 250 000 in one chunk."""
    torch, acav, O = env
    import bench
    from acav100m_amd.subset_selection.run_greedy import _run_greedy
    n, K = 1_250_000, 1024
    a = _train_and_label(env, n, (2048, 128), K, 41)
    types = [("audio_model", "layer_0"), ("visual_model", "layer_0")]
    random.seed(0)
    acav.manual_seed(5)
    S, GAIN, _ = _run_greedy(bench.select_args(), a, types, None, 0.2, "batch_mi", "combination", True, False)
    subset = round(0.2 * n)
    assert len(S) == subset and len(set(S)) == subset and min(S) >= 0 and max(S) < n
    assert len(GAIN) == -(-subset // 4) * 4 and np.isfinite(GAIN).all()
    # the first 2 000 iterations against the oracle (same shuffle, same generator)
    random.seed(0)
    order = list(range(n))
    random.shuffle(order)
    assert order[0] not in set(S)  # the start index seeds the tables, it is never selected (batch.py:205-206)
    C = int(a.max()) + 1
    r = O.BatchMI(a, C, [(0, 1)]).run_greedy(order[1:], order[:1], subset, 20, 4, O.Rng(5), max_iters=2000)
    assert S[:8000] == r["S"].tolist() and np.array_equal(np.array(GAIN[:8000]), r["GAIN"])


def test_cfg5_slice_full_size(env):
    """configs[4] per GPU: 12.5M x 1024 x 2 views (102 GB), K = 1024 -- train, assign (both paths, all rows), then 125 chunks
    of 100k clips, 10 in lockstep, 20 % each."""
    torch, acav, O = env
    import bench
    n, K, chunk, width = 12_500_000, 1024, 100_000, 10
    a = _train_and_label(env, n, (1024, 1024), K, 43)
    types = [("audio_model", "layer_0"), ("visual_model", "layer_0")]
    random.seed(0)
    res = bench.select_chunked(a, types, chunk, width)
    assert len(res) == n // chunk
    per = round(0.2 * chunk)
    for ci, (S, GAIN) in enumerate(res):
        assert len(S) == per and len(set(S)) == per and min(S) >= 0 and max(S) < chunk, f"chunk {ci}"
        assert len(GAIN) == -(-per // 4) * 4 and np.isfinite(GAIN).all()
    # chunk 0 == the oracle's run on the same rows (generator seed 1, first python shuffle after random.seed(0))
    r = _oracle_selection(O, a[:chunk], 1, per, 0)
    assert res[0][0] == r["S"].tolist() and np.array_equal(np.array(res[0][1]), r["GAIN"])
    # the union over the chunks: 20 % of the slice, no clip twice
    total = sum(len(S) for S, _ in res)
    assert total == round(0.2 * n)
