"""GPU parity tests AT THE SHAPES BASELINE.json names (VERDICT r1 "configs not exercised"): cfg1 in full against the
oracle; the bf16 filter at cfg2's 1M x 1024 x K=256 with trained centres on overlapping data (non-empty re-check list);
cfg3's V = 1M single-chunk selection (first 2 000 iterations == oracle, then the size / uniqueness properties of the
full run); K = 1024 at d = 128 / 1024 / 2048 (cfg4 / cfg5: training and both assign paths); and a fixed-seed slice of
the randomised stress of tools/stress_parity.py.  Integer outputs and centres bit-exact."""
import importlib.util
import itertools
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import acav100m_amd
    acav100m_amd.load_library()
    from oracle import oracle as O
    return torch, acav100m_amd, O


class _NS(dict):
    __getattr__ = dict.get


def _views(seed, n, d, k, nviews=2, rho=0.5, noise=0.3):
    """SURVEY 8(d) generator on the host (small n)"""
    rs = np.random.RandomState(seed)
    shared = rs.randint(0, k, n)
    out = []
    for _ in range(nviews):
        cen = rs.randn(k, d).astype(np.float32)
        comp = np.where(rs.rand(n) < rho, shared, rs.randint(0, k, n))
        out.append((cen[comp] + noise * rs.randn(n, d)).astype(np.float32))
    return out


def test_cfg1_full_pipeline_vs_oracle(env):
    """BASELINE configs[0]: 10k clips, 512-d audio + 512-d visual, K=64, 2 epochs at b=32, assign, select 20 % --
    the whole path on the GPU against the oracle, bit for bit (centres, counts, labels, S, GAIN)."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    from acav100m_amd.subset_selection.run_greedy import _run_greedy
    n, d, K, b = 10_000, 512, 64, 32
    xs = _views(1, n, d, K)
    acav.manual_seed(0)
    rng = O.Rng(0)
    kms = [KMeans(None, d, K) for _ in xs]          # RNG order: both inits first (run_clustering.py:32-44)
    refs = [O.KMeans(d, K, rng) for _ in xs]
    for km in kms:
        km.to("cuda:0")
    xts = [torch.from_numpy(x).cuda() for x in xs]
    steps = n // b
    for epoch in range(2):
        lr = 0.1 ** (2 + epoch // 5)
        # warm-up labels interleaved batch by batch across the clusterings (run_clustering.py:229-241)
        need = [km.warmup_steps(b, steps) for km in kms]
        warm = [np.empty((nd, b), np.int64) for nd in need]
        for t in range(max(need)):
            for v, km in enumerate(kms):
                if t < need[v]:
                    warm[v][t] = km.draw_warmup(b)
        for v, km in enumerate(kms):
            km.train_epoch(xts[v], b, lr=lr, warm_best=warm[v])
        for t in range(steps):  # the oracle follows the reference loop literally: per batch, every clustering adds
            for v, ref in enumerate(refs):
                ref.lr = lr
                ref.add(xs[v][t * b:(t + 1) * b])
        for km, ref in zip(kms, refs):
            assert np.array_equal(km.centers.numpy(), ref.centers), f"epoch {epoch}"
            assert np.array_equal(km.counts.numpy(), ref.counts) and km.count == ref.count
    labs = []
    for km, ref, xt, x in zip(kms, refs, xts, xs):
        want = ref.calc_best(x)[0]
        fast = km.calc_best(xt, need_mean=False)[0].cpu().numpy()
        exact = km.calc_best(xt, need_mean=True)[0].cpu().numpy()
        assert np.array_equal(fast, want) and np.array_equal(exact, want)
        labs.append(want)
    a = np.stack(labs, 1).astype(np.int64)
    args = _NS(batch=_NS(batch_size=20, selection_size=4, keep_unselected=True), computation=_NS(device="cuda:0"),
               log_every=1000, log_times=10)
    types = [("audio", "layer_0"), ("visual", "layer_0")]
    random.seed(0)
    acav.manual_seed(5)
    S, GAIN, _ = _run_greedy(args, a, types, None, 0.2, "batch_mi", "combination", True, False)
    random.seed(0)
    order = list(range(n))
    random.shuffle(order)
    C = int(a.max()) + 1
    r = O.BatchMI(a, C, [(0, 1)]).run_greedy(order[1:], order[:1], 2000, 20, 4, O.Rng(5))
    assert len(S) == 2000 and S == r["S"].tolist()
    assert np.array_equal(np.array(GAIN), r["GAIN"])


@pytest.mark.parametrize("spread", [0.04, 0.05, 0.06, 0.25])
def test_cfg2_filter_full_size_trained_centres(env, spread):
    """BASELINE configs[1] shape, the kernel the roofline is quoted on: 1M x 1024, K=256, centres that come out of
    real training on OVERLAPPING clusters (centre spread << noise radius), so the filter's acceptance test fails for
    part of the rows and the exact re-check pass runs with a non-empty list.  filter+re-check == exact sweep on all
    1M rows, == oracle on 16 384 sampled rows."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    n, d, K, b = 1_000_000, 1024, 256, 32
    gen = torch.Generator(device="cuda").manual_seed(7)
    cen = spread * torch.randn(K, d, device="cuda", generator=gen)
    comp = torch.randint(0, K, (n,), device="cuda", generator=gen)
    x = torch.empty(n, d, device="cuda")
    for s in range(0, n, 65536):
        e = min(n, s + 65536)
        x[s:e] = cen[comp[s:e]] + 0.3 * torch.randn(e - s, d, device="cuda", generator=gen)
    acav.manual_seed(3)
    km = KMeans(None, d, K).to("cuda:0")
    km.train_epoch(x[:262144], b, lr=0.01)  # 8 192 SGD steps, warm-up included
    fast = km.calc_best(x, need_mean=False)[0]
    launches, rows, rechecked = km.filter_stats()
    assert launches >= 1 and rows == n
    assert rechecked > 0, "the data was meant to leave ambiguous rows for the exact re-check"
    print(f"spread {spread}: {rechecked} of {n} rows re-checked exactly")
    exact = km.calc_best(x, need_mean=True)[0]
    assert torch.equal(fast, exact)
    idx = np.sort(np.random.RandomState(0).choice(n, 16384, replace=False))
    ref = O.KMeans(d, K, O.Rng(0))
    ref.set_state(km.centers.numpy(), km.counts.numpy(), km.count)
    want = ref.calc_best(x[torch.from_numpy(idx).cuda()].cpu().numpy())[0]
    assert np.array_equal(fast.cpu().numpy()[idx], want)


def test_cfg3_single_chunk_one_million(env):
    """BASELINE configs[2]'s selection: V = 1M in ONE chunk (the reference default).  The first 2 000 iterations
    (8 000 picks, each after a full 10^6-element randperm) equal the oracle; the generator is handed back in the
    oracle's state; the full 50 000-iteration run keeps that prefix and returns 200 000 distinct valid ids."""
    torch, acav, O = env
    from acav100m_amd.rng import Generator
    from acav100m_amd.subset_selection import get_measure
    v, c = 1_000_000, 256
    rs = np.random.RandomState(11)
    comp = rs.randint(0, c, v)
    a = np.stack([np.where(rs.rand(v) < 0.5, comp, rs.randint(0, c, v)) for _ in range(2)], 1).astype(np.int64)
    a[0] = c - 1
    cand = rs.permutation(v).astype(np.int64)
    subset = 200_000
    gen = Generator(21)
    m = get_measure("batch_mi")(a, ncentroids=c, batch_size=20, selection_size=4, device="cuda:0", keep_unselected=True,
                                generator=gen)
    m.init([(0, 1)], cand[1:])
    S, G, _, _ = m.run_greedy(subset, cand[:1], None, max_iters=2000)
    orng = O.Rng(21)
    r = O.BatchMI(a, c, [(0, 1)]).run_greedy(cand[1:], cand[:1], subset, 20, 4, orng, max_iters=2000)
    assert len(S) == 8000 and S == r["S"].tolist()
    assert np.array_equal(np.array(G), r["GAIN"])
    assert [gen.u32() for _ in range(4)] == [orng.u32() for _ in range(4)], "generator state after the run"
    m2 = get_measure("batch_mi")(a, ncentroids=c, batch_size=20, selection_size=4, device="cuda:0", keep_unselected=True,
                                 generator=Generator(21))
    m2.init([(0, 1)], cand[1:])
    S2, G2, _, _ = m2.run_greedy(subset, cand[:1], None)
    assert len(S2) == subset and len(set(S2)) == subset
    assert S2[:8000] == S and int(cand[0]) not in set(S2)
    assert min(S2) >= 0 and max(S2) < v and len(G2) == 50_000 * 4
    n_tab = m2.cache
    assert int(n_tab["n"]) == subset + 1 and int(n_tab["N"].sum()) == subset + 1


@pytest.mark.parametrize("d", [128, 1024, 2048])
def test_k1024_train_and_assign(env, d):
    """K = 1024 (cfg4: d = 2048 visual / 128 audio; cfg5: d = 1024): one epoch with the 320 warm-up steps inside, one
    epoch of real steps, then both assign paths -- all bit-exact against the oracle."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    K, b, n = 1024, 32, 12288
    rs = np.random.RandomState(d)
    cen = rs.randn(K, d).astype(np.float32)
    x = (cen[rs.randint(0, K, n)] + 0.3 * rs.randn(n, d)).astype(np.float32)
    acav.manual_seed(17)
    km = KMeans(None, d, K).to("cuda:0")
    ref = O.KMeans(d, K, O.Rng(17))
    xt = torch.from_numpy(x).cuda()
    for e in range(2):
        km.train_epoch(xt, b, lr=0.01)
        ref.train_epoch(x, b, lr=0.01)
        assert np.array_equal(km.centers.numpy(), ref.centers), f"epoch {e}"
        assert np.array_equal(km.counts.numpy(), ref.counts) and km.count == ref.count
    want = ref.calc_best(x)[0]
    assert np.array_equal(km.calc_best(xt, need_mean=False)[0].cpu().numpy(), want)
    assert np.array_equal(km.calc_best(xt, need_mean=True)[0].cpu().numpy(), want)
    # under-used centres (the /5 discount) at K = 1024: doctored counts
    cnt = km.counts.clone()
    cnt[::7] = 0.0
    km.counts = cnt
    ref.set_state(None, cnt.numpy(), ref.count)
    want = ref.calc_best(x)[0]
    assert np.array_equal(km.calc_best(xt, need_mean=False)[0].cpu().numpy(), want)
    assert np.array_equal(km.calc_best(xt, need_mean=True)[0].cpu().numpy(), want)


@pytest.mark.parametrize("name", ["d1024_k256", "d2048_k1024"])
def test_reference_goldens_at_baseline_shapes(env, name):
    """The REFERENCE's own results at BASELINE's shapes (tests/golden/kmeans_big_*.npz, recorded by gen_golden.py
    `kmeans_big`; 32 768 rows of overlapping clusters): GPU training from the reference's seed ends in the reference's
    centres bit for bit (sha256), both assign paths give the reference's label on every natural row (also under the
    doctored /5 discount), and on the 4 096 bisector rows the GPU equals the oracle exactly while every difference from
    the reference is an exact-arithmetic tie (census printed; tests/_census.py)."""
    import hashlib
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    from tests import _census as Z
    g, x, n, d, K = Z.load_case(name)
    b = int(g["b"])
    acav.manual_seed(int(g["seed"]))
    km = KMeans(None, d, K).to("cuda:0")
    xt = torch.from_numpy(x).cuda()
    for e in range(int(g["epochs"])):
        km.train_epoch(xt, b, lr=0.1 ** (2 + e // 5))
    c, cnt = km.centers.numpy(), km.counts.numpy()
    assert hashlib.sha256(c.tobytes()).hexdigest() == str(g["centers_sha256"]), "trained centres differ from the reference's"
    assert np.array_equal(cnt, g["counts"]) and km.count == int(g["count"]) and km.fallback == int(g["fallback"])
    reinit = (0.7, 5.0)
    fast = km.calc_best(xt, need_mean=False)[0].cpu().numpy()
    exact = km.calc_best(xt, need_mean=True)[0].cpu().numpy()
    assert np.array_equal(fast, exact)
    nm, _ = Z.census(x, c, cnt, km.count, reinit, fast, g["labels"], f"GPU, {name}, natural rows")
    assert nm == 0
    ref = O.KMeans(d, K, O.Rng(0))
    ref.set_state(c, cnt, km.count)
    xb = Z.bisector(g, x, c)
    xbt = torch.from_numpy(xb).cuda()
    want = ref.calc_best(xb)[0]
    fastb = km.calc_best(xbt, need_mean=False)[0].cpu().numpy()
    _, rows, rechecked = km.filter_stats()
    exactb = km.calc_best(xbt, need_mean=True)[0].cpu().numpy()
    assert np.array_equal(fastb, want) and np.array_equal(exactb, want)  # GPU == oracle bit for bit ON the ties
    print(f"{name}: {rechecked} of {rows} bisector rows went to the exact re-check")
    assert rechecked >= len(xb) // 2
    nb, _ = Z.census(xb, c, cnt, km.count, reinit, fastb, g["bis_labels"], f"GPU, {name}, bisector rows", max_ulps=8)
    assert nb > 0
    cnt2 = cnt.copy()
    cnt2[::3] = 1.0
    km.counts = torch.from_numpy(cnt2)
    doc = km.calc_best(xt, need_mean=False)[0].cpu().numpy()
    assert np.array_equal(doc, km.calc_best(xt, need_mean=True)[0].cpu().numpy())
    Z.census(x, c, cnt2, km.count, reinit, doc, g["labels_doctored"], f"GPU, {name}, doctored discount")


@pytest.mark.parametrize("v", [16_777_216, 17_000_000])
def test_longest_lists_first_iterations_vs_oracle(env, v):
    """Maximum sizes of the permutation: the longest list the tiled Fisher-Yates takes (16 Mi candidates: ~3 700 tiles,
    the largest tile table and buckets) and the first size beyond it (global-atomic kernels, chosen automatically) --
    the first iterations of a selection equal the oracle's, ids and float64 gains."""
    torch, acav, O = env
    from acav100m_amd.subset_selection import get_measure
    c, iters = 64, 10
    rs = np.random.RandomState(1)
    a = rs.randint(0, c, (v, 2)).astype(np.int64)
    a[0] = c - 1
    cand = np.arange(v, dtype=np.int64)
    acav.manual_seed(3)
    m = get_measure("batch_mi")(a, ncentroids=c, batch_size=20, selection_size=4, device="cuda:0", keep_unselected=True)
    m.init([(0, 1)], cand[1:])
    S, G, _, _ = m.run_greedy(4 * iters, [0], None, max_iters=iters)
    ref = O.BatchMI(a, c, [(0, 1)]).run_greedy(cand[1:], cand[:1], 4 * iters, 20, 4, O.Rng(3), max_iters=iters)
    assert S == list(ref["S"]) and np.array_equal(np.array(G), ref["GAIN"])


@pytest.mark.parametrize("seed", [2024, 601, 7])
def test_stress_parity_fixed_seed_slice(env, seed):
    """tools/stress_parity.py with three fixed seeds (45 s each): random shapes (d 8..2304, K 2..2048, b 7..512, ragged everything,
    data scales 1e-6 .. 3e4) for both assign paths, persistent / wide / per-step training alone and side by side, batch greedy alone
    and in lockstep, exact greedy, the DDP epoch through the C-ABI communicator (a world of one) -- every case bit-identical.  (The
    tool run longer is what found the 257..500-centre clusterings on the wrong persistent kernel, and in round 5 the inline-asm pack
    conversion that was wrong in the scaled-row 8-wave instantiations.)"""
    spec = importlib.util.spec_from_file_location("stress_parity", os.path.join(ROOT, "tools", "stress_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    counts = mod.stress(seed=seed, budget=45.0, max_cases=70)
    assert sum(counts.values()) >= 12 and sum(1 for v in counts.values() if v > 0) >= 4, counts
