"""GPU parity tests for the greedy batch-MI selection: HIP (C ABI / EfficientBatchMI mirror) vs the
oracle (bit-exact ids, identical float64 gains) and vs the golden traces recorded from the
reference (teacher-forced: identical batch ids every iteration, scores within 1e-5)."""
import itertools
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import acav100m_amd
    acav100m_amd.load_library()
    from oracle import oracle as O
    return torch, acav100m_amd, O


def _measure(a, c, pairs, cand, B=20, k=4, keep=True):
    from acav100m_amd.subset_selection import get_measure
    m = get_measure("batch_mi")(a, ncentroids=c, batch_size=B, selection_size=k, device="cuda:0",
                                keep_unselected=keep)
    m.init(pairs, cand)
    return m


def _correlated(seed, v, dd, c):
    rs = np.random.RandomState(seed)
    comp = rs.randint(0, c, size=v)
    cols = [np.where(rs.rand(v) < 0.5, comp, rs.randint(0, c, size=v)) for _ in range(dd)]
    a = np.stack(cols, 1).astype(np.int64)
    a[0] = c - 1
    return a


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_golden_trace_teacher_forced(env, golden_dir, name):
    torch, acav, O = env
    g = np.load(os.path.join(golden_dir, f"mi_{name}.npz"))
    a, c, seed = g["assignments"], int(g["C"]), int(g["seed"])
    v, dd = a.shape
    pairs = list(itertools.combinations(range(dd), 2))
    subset = round(float(g["ratio"]) * v)
    cand = list(g["shuffled"])
    start, cand = [cand[0]], cand[1:]
    acav.manual_seed(seed)
    m = _measure(a, c, pairs, cand)
    S, GAIN, _, _ = m.run_greedy(subset, start, None, record_trace=True, forced_pos=g["pick_pos"])
    # permutation stream, batch slicing and re-queue order are exactly the reference's
    assert np.array_equal(m.trace["ids"], g["ids"])
    assert S == list(g["S"])
    ref_mean = g["scores"].astype(np.float64).mean(-1)
    np.testing.assert_allclose(m.trace["scores"], ref_mean, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(GAIN, g["GAIN"], rtol=1e-5, atol=1e-6)
    # our own picks differ from the reference's only at fp32 near-ties of the reference scores
    diff = 0
    for t in range(len(g["ids"])):
        if set(m.trace["pos"][t]) != set(g["pick_pos"][t]):
            srt = np.sort(ref_mean[t])[::-1]
            k = g["pick_pos"].shape[1]
            assert abs(srt[k - 1] - srt[k]) <= 2e-6 * max(abs(srt[k - 1]), 1e-3), (t, srt[:k + 1])
            diff += 1
    print(f"mi_{name}: {diff}/{len(g['ids'])} iterations differ from the reference picks, all at near-ties")


@pytest.mark.parametrize("v,dd,c,keep,subset", [(600, 3, 16, True, 120), (2000, 2, 64, True, 401),
                                                (900, 10, 32, True, 180), (500, 2, 8, False, 60),
                                                (5000, 2, 256, True, 1000)])
def test_free_running_equals_oracle(env, v, dd, c, keep, subset):
    torch, acav, O = env
    a = _correlated(v + dd, v, dd, c)
    pairs = list(itertools.combinations(range(dd), 2))
    rs = np.random.RandomState(v)
    cand = [int(i) for i in rs.permutation(v)]
    start, cand = [cand[0]], cand[1:]
    acav.manual_seed(9)
    m = _measure(a, c, pairs, cand, keep=keep)
    S, GAIN, _, _ = m.run_greedy(subset, start, None, record_trace=True)
    rng = O.Rng(9)
    om = O.BatchMI(a, c, pairs)
    k = m.k
    ref = om.run_greedy(cand, start, subset, 20, k, rng, keep_unselected=keep, trace=True)
    assert np.array_equal(m.trace["ids"], ref["ids"])
    assert np.array_equal(m.trace["pos"], ref["pos"])
    assert S == list(ref["S"])
    assert np.array_equal(np.array(GAIN), ref["GAIN"])  # same float64 operations, same order
    assert len(set(S)) == len(S) == subset and start[0] not in S
    # the tables hold exactly the start sample + the committed picks
    Nc, ac, bc, nc = om.counts()
    cache = m.cache
    assert np.array_equal(cache["N"], Nc) and np.array_equal(cache["a"], ac) and np.array_equal(cache["b"], bc)
    assert cache["n"] == nc == 1 + len(GAIN)
    # the MT19937 stream continues where the device left it
    mt_o, idx_o = rng.get_state()
    mt_p, idx_p = acav.default_generator.get_state()
    assert idx_o == idx_p and np.array_equal(mt_o, mt_p)


@pytest.mark.parametrize("switch", [("ACAV_FY_LEGACY", "1"), ("ACAV_FY_ECAP", "64"), ("ACAV_FY_CAP", "256"),
                                    ("ACAV_FY_PART_DIRECT", "1")])
def test_permutation_variants_equal_oracle(env, switch, monkeypatch):
    """The other evaluations of the same swap sequence give the same selection: the global-atomic Fisher-Yates kernels
    (ACAV_FY_LEGACY=1, also the path of lists beyond the tile table), the tiled kernels with the tile's LDS capacity cut
    so that every loaded tile takes the sub-ranged overload path (ACAV_FY_ECAP), smaller tiles (ACAV_FY_CAP), and the bucket
    appends written one by one instead of as LDS-sorted runs (ACAV_FY_PART_DIRECT: the form lists of millions of candidates
    fall back to when the tile table leaves no room for the staging area)."""
    torch, acav, O = env
    monkeypatch.setenv(*switch)
    v, dd, c, subset = 20000, 2, 32, 1200
    a = _correlated(77, v, dd, c)
    pairs = list(itertools.combinations(range(dd), 2))
    cand = [int(i) for i in np.random.RandomState(5).permutation(v)]
    start, cand = [cand[0]], cand[1:]
    acav.manual_seed(21)
    m = _measure(a, c, pairs, cand)
    S, GAIN, _, _ = m.run_greedy(subset, start, None, record_trace=True)
    rng = O.Rng(21)
    ref = O.BatchMI(a, c, pairs).run_greedy(cand, start, subset, 20, m.k, rng, keep_unselected=True, trace=True)
    assert np.array_equal(m.trace["ids"], ref["ids"]) and S == list(ref["S"])
    assert np.array_equal(np.array(GAIN), ref["GAIN"])
    mt_o, idx_o = rng.get_state()
    mt_p, idx_p = acav.default_generator.get_state()
    assert idx_o == idx_p and np.array_equal(mt_o, mt_p)


def test_scores_vs_oracle_and_sklearn(env):
    torch, acav, O = env
    from sklearn.metrics import mutual_info_score
    v, dd, c = 3000, 2, 32
    a = _correlated(1, v, dd, c)
    pairs = [(0, 1)]
    m = _measure(a, c, pairs, list(range(1, v)))
    sel = list(range(0, 1000))
    m.add_samples(sel)
    om = O.BatchMI(a, c, pairs)
    om.add_samples(sel)
    ids = np.arange(1000, 1020)
    s = m.score_batch(ids)
    assert np.array_equal(s, om.scores_canon(ids))
    np.testing.assert_allclose(s, om.scores_dense(ids)[:, 0], rtol=1e-6)
    for w, i in enumerate(ids):
        rows = a[sel + [int(i)]]
        kat = mutual_info_score(rows[:, 0], rows[:, 1])
        assert abs(s[w] - kat) <= 1e-6 * kat


def test_larger_run_properties(env):
    """V = 100k (a chunk of the reference's chunked mode): selection is a set of distinct ids of
    the right size, and matches the oracle on the first 300 iterations."""
    torch, acav, O = env
    v, dd, c = 100_000, 2, 64
    a = _correlated(7, v, dd, c)
    pairs = [(0, 1)]
    rs = np.random.RandomState(7)
    cand = [int(i) for i in rs.permutation(v)]
    start, cand = [cand[0]], cand[1:]
    acav.manual_seed(21)
    m = _measure(a, c, pairs, cand)
    S, GAIN, _, _ = m.run_greedy(4000, start, None)
    assert len(S) == len(set(S)) == 4000 and min(S) >= 0 and max(S) < v
    ref = O.BatchMI(a, c, pairs).run_greedy(cand, start, 1200, 20, 4, O.Rng(21))
    assert S[:1200] == list(ref["S"])


def test_mi_errors(env):
    torch, acav, O = env
    a = _correlated(0, 50, 2, 4)
    m = _measure(a, 4, [(0, 1)], list(range(1, 50)))
    with pytest.raises(RuntimeError):
        m.run_greedy(45, [0], None)  # candidates run out below batch_size: the reference's topk raises
    with pytest.raises(ValueError):
        bad = a.copy()
        bad[3, 0] = 9
        _measure(bad, 4, [(0, 1)], list(range(1, 50)))


# ------------------------------------------------------------------ exact greedy ('mi' / 'mem_mi')
def _remaining_to_original(idx, L):
    """positions in the shrinking list (what the reference records) -> positions in the initial list"""
    alive = list(range(L))
    return np.array([alive.pop(int(i)) for i in idx], np.int64)


@pytest.mark.parametrize("name", ["a", "b", "c"])
@pytest.mark.parametrize("measure", ["mi", "mem_mi"])
def test_exact_greedy_golden_teacher_forced(env, golden_dir, name, measure):
    """HIP exact greedy replaying the reference's recorded picks: S / GAIN / per-iteration score vectors equal the
    oracle's bit for bit, hence (test_oracle_golden) the reference's to 2e-6 with its picks inside exact ties."""
    torch, acav, O = env
    from acav100m_amd.subset_selection import get_measure
    g = np.load(os.path.join(golden_dir, f"mi_exact_{name}.npz"))
    a, c, subset, cand = g["assignments"], int(g["C"]), int(g["subset"]), g["shuffled"]
    pairs = list(itertools.combinations(range(a.shape[1]), 2))
    idx = g[f"{measure}_idx"]
    L = len(cand) - 1
    m = get_measure(measure)(a, ncentroids=c, batch_size=20, selection_size=4, device="cuda:0", keep_unselected=True)
    m.init(pairs, [int(i) for i in cand[1:]])
    S, GAIN, timelapse, lookups = m.run_greedy(subset, [int(cand[0])], None, record_trace=True,
                                               forced_pos=_remaining_to_original(idx, L))
    assert S == g[f"{measure}_S"].tolist() and len(GAIN) == len(timelapse) == len(lookups) == subset - 2
    ref = O.BatchMI(a, c, pairs).run_exact(cand[1:], cand[:1], subset, forced_idx=idx, trace=True)
    assert np.array_equal(np.array(GAIN), ref["GAIN"])  # float64, bit for bit
    alive = list(range(L))
    for t in range(subset - 2):
        row = m.trace["scores"][t]
        assert np.array_equal(row[alive], ref["scores"][t, :len(alive)])
        dead = np.setdiff1d(np.arange(L), alive)
        assert np.isnan(row[dead]).all()
        assert m.trace["argmax"][t] == alive[int(ref["argmax"][t])]
        alive.pop(int(idx[t]))
    assert len(m.candidate_ids) == L - (subset - 2)


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_ami_exact_greedy_golden_and_free_running(env, golden_dir, name):
    """'ami' (measures/mi.py:212-259) on the GPU: replaying the reference's recorded picks, S / GAIN / every score vector
    equal the oracle's canonical float64 form bit for bit (table look-ups + the shared canon_exp), hence
    (test_oracle_golden) the reference's fp32 scores to 4e-7 relative; free-running == the oracle, pick for pick."""
    torch, acav, O = env
    from acav100m_amd.subset_selection import get_measure
    g = np.load(os.path.join(golden_dir, f"mi_ami_{name}.npz"))
    a, c, subset, cand = g["assignments"], int(g["C"]), int(g["subset"]), g["shuffled"]
    pairs = list(itertools.combinations(range(a.shape[1]), 2))
    idx = g["idx"]
    L = len(cand) - 1
    m = get_measure("ami")(a, ncentroids=c, batch_size=20, selection_size=4, device="cuda:0", keep_unselected=True)
    m.init(pairs, [int(i) for i in cand[1:]])
    S, GAIN, _, _ = m.run_greedy(subset, [int(cand[0])], None, record_trace=True, forced_pos=_remaining_to_original(idx, L))
    assert S == g["S"].tolist()
    om = O.BatchMI(a, c, pairs)
    om.set_measure("ami")
    ref = om.run_exact(cand[1:], cand[:1], subset, forced_idx=idx, trace=True)
    assert np.array_equal(np.array(GAIN), ref["GAIN"])  # float64, bit for bit
    alive = list(range(L))
    for t in range(subset - 2):
        row = m.trace["scores"][t]
        assert np.array_equal(row[alive], ref["scores"][t, :len(alive)])
        assert m.trace["argmax"][t] == alive[int(ref["argmax"][t])]
        alive.pop(int(idx[t]))
    np.testing.assert_allclose(np.array(GAIN), g["GAIN"], rtol=1e-5, atol=1e-7)  # the reference's own fp32 gains
    m2 = get_measure("ami")(a, ncentroids=c, device="cuda:0")
    m2.init(pairs, [int(i) for i in cand[1:]])
    S2, G2, _, _ = m2.run_greedy(subset, [int(cand[0])])
    om2 = O.BatchMI(a, c, pairs)
    om2.set_measure("ami")
    free = om2.run_exact(cand[1:], cand[:1], subset)
    assert S2[1:] == free["S"].tolist() and np.array_equal(np.array(G2), free["GAIN"])


@pytest.mark.parametrize("name", ["nmi_a", "nmi_b", "nmi_c", "constant_a"])
def test_nmi_and_constant_golden_and_free_running(env, golden_dir, name):
    """'nmi' / 'constant' (EfficientNMI / ConstantMeasure, measures/mi.py:262-281) on the GPU: replaying the reference's recorded
    picks, S / GAIN / every score vector equal the oracle's canonical float64 form bit for bit (hence, test_oracle_golden, the
    reference's fp32 scores to 1e-5); free-running == the oracle, pick for pick."""
    torch, acav, O = env
    from acav100m_amd.subset_selection import get_measure
    g = np.load(os.path.join(golden_dir, f"mi_{name}.npz"))
    a, c, subset, cand = g["assignments"], int(g["C"]), int(g["subset"]), g["shuffled"]
    pairs = list(itertools.combinations(range(a.shape[1]), 2))
    idx = g["idx"]
    measure = name.split("_")[0]
    L = len(cand) - 1
    m = get_measure(measure)(a, ncentroids=c, batch_size=20, selection_size=4, device="cuda:0", keep_unselected=True)
    m.init(pairs, [int(i) for i in cand[1:]])
    S, GAIN, _, _ = m.run_greedy(subset, [int(cand[0])], None, record_trace=True, forced_pos=_remaining_to_original(idx, L))
    assert S == g["S"].tolist()
    om = O.BatchMI(a, c, pairs)
    om.set_measure(measure)
    ref = om.run_exact(cand[1:], cand[:1], subset, forced_idx=idx, trace=True)
    assert np.array_equal(np.array(GAIN), ref["GAIN"])  # float64, bit for bit
    alive = list(range(L))
    for t in range(subset - 2):
        row = m.trace["scores"][t]
        assert np.array_equal(row[alive], ref["scores"][t, :len(alive)])
        assert m.trace["argmax"][t] == alive[int(ref["argmax"][t])]
        alive.pop(int(idx[t]))
    np.testing.assert_allclose(np.array(GAIN), g["GAIN"], rtol=1e-5, atol=1e-7)  # the reference's own fp32 gains
    m2 = get_measure(measure)(a, ncentroids=c, device="cuda:0")
    m2.init(pairs, [int(i) for i in cand[1:]])
    S2, G2, _, _ = m2.run_greedy(subset, [int(cand[0])])
    om2 = O.BatchMI(a, c, pairs)
    om2.set_measure(measure)
    free = om2.run_exact(cand[1:], cand[:1], subset)
    assert S2[1:] == free["S"].tolist() and np.array_equal(np.array(G2), free["GAIN"])
    if measure == "constant":
        assert S2 == g["S"].tolist() and all(v == 1.0 for v in G2)


def test_ami_larger_tables_free_running_equals_oracle(env):
    """ami beyond the golden sizes: D = 4 (P = 6), C = 24, 2 000 candidates, 80 picks -- GPU == oracle bit for bit"""
    torch, acav, O = env
    from acav100m_amd.subset_selection import get_measure
    v, dd, c, subset = 2000, 4, 24, 80
    a = _correlated(77, v, dd, c)
    pairs = list(itertools.combinations(range(dd), 2))
    cand = np.random.RandomState(5).permutation(v)
    m = get_measure("ami")(a, ncentroids=c, device="cuda:0")
    m.init(pairs, [int(i) for i in cand[1:]])
    S, GAIN, _, _ = m.run_greedy(subset, [int(cand[0])])
    om = O.BatchMI(a, c, pairs)
    om.set_measure("ami")
    ref = om.run_exact(cand[1:], cand[:1], subset)
    assert S[1:] == ref["S"].tolist() and np.array_equal(np.array(GAIN), ref["GAIN"])


@pytest.mark.parametrize("v,dd,c,subset,pairing", [(3000, 2, 16, 200, "combination"), (1500, 4, 40, 120, "combination"),
                                                    (5000, 10, 12, 60, "bipartite"), (700, 3, 300, 90, "combination")])
def test_exact_greedy_free_running_equals_oracle(env, v, dd, c, subset, pairing):
    """free-running: same picks in the same order, identical float64 gains (first-maximum rule, ties included --
    many candidates share an assignment row, so exact ties are the norm); D=10 views with the reference's
    bipartite pairing (P=25) as in the real 5+5-layer pipeline; C > 256."""
    torch, acav, O = env
    from acav100m_amd.subset_selection import get_measure
    from acav100m_amd.subset_selection.pairing import get_cluster_pairing
    a = _correlated(50 + v, v, dd, c)
    keys = [("audio" if i < dd // 2 else "video", f"layer_{i}") for i in range(dd)]
    pairs = get_cluster_pairing(keys, pairing) if pairing != "combination" else list(itertools.combinations(range(dd), 2))
    rs = np.random.RandomState(v)
    cand = rs.permutation(v)
    m = get_measure("mem_mi")(a, ncentroids=c, device="cuda:0")
    m.init(pairs, [int(i) for i in cand[1:]])
    S, GAIN, _, _ = m.run_greedy(subset, [int(cand[0])])
    ref = O.BatchMI(a, c, pairs).run_exact(cand[1:], cand[:1], subset)
    assert S[0] == cand[0] and S[1:] == ref["S"].tolist() and len(S) == subset - 1
    assert np.array_equal(np.array(GAIN), ref["GAIN"])
    assert len(set(S)) == len(S)
    # the tables hold exactly the picks (the start index is not added: mi.py never does)
    om = O.BatchMI(a, c, pairs)
    om.add_samples(np.array(S[1:], np.int64))
    Nc, ac, bc, nc = om.counts()
    cache = m.cache
    assert np.array_equal(cache["N"], Nc) and np.array_equal(cache["a"], ac) and np.array_equal(cache["b"], bc)
    assert cache["n"] == nc == subset - 2


def test_exact_greedy_through_run_greedy(env):
    """run_greedy._run_greedy(measure_name='mi') end to end, and the edge cases of the loop bounds (mi.py:161)"""
    torch, acav, O = env
    import random
    import importlib
    rg = importlib.import_module("acav100m_amd.subset_selection.run_greedy")
    from acav100m_amd.config import Namespace
    a = _correlated(3, 400, 2, 8)
    args = Namespace(batch=Namespace(batch_size=20, selection_size=4, keep_unselected=True),
                     computation=Namespace(device="cuda"), log_every=10 ** 9, log_times=None, node_rank=None,
                     parent_pid=None)
    random.seed(5)
    S, GAIN, _ = rg._run_greedy(args, a, [("m0", "layer_0"), ("m1", "layer_0")], 30, None, "mi", "combination", True, False)
    random.seed(5)
    cand = list(range(400))
    random.shuffle(cand)
    ref = O.BatchMI(a, 8, [(0, 1)]).run_exact(np.array(cand[1:]), np.array(cand[:1]), 30)
    assert S == [cand[0]] + ref["S"].tolist() and np.array_equal(np.array(GAIN), ref["GAIN"])
    from acav100m_amd.subset_selection import get_measure
    m = get_measure("mi")(a, ncentroids=8, device="cuda:0")
    m.init([(0, 1)], cand[1:])
    assert m.run_greedy(2, [cand[0]])[0] == [cand[0]]          # subset - 1 - ns = 0 iterations
    assert m.run_greedy(1, [cand[0]])[0] == [cand[0]]
    S_all = m.run_greedy(10 ** 6, [cand[0]])[0]                # more than there are candidates: everything, once
    assert sorted(S_all) == list(range(400))


def test_lockstep_chunks_equal_individual_runs(env):
    """acav_mi_run_greedy_multi: several chunks (different sizes, different D / P / C tables, own generators) driven
    by one set of launches per iteration give exactly what each chunk gives alone -- and hence the oracle's result;
    every generator ends where its own run would have left it."""
    torch, acav, O = env
    from acav100m_amd.rng import Generator
    from acav100m_amd.subset_selection import get_measure
    from acav100m_amd.subset_selection.measures.batch import EfficientBatchMI
    specs = [(3000, 2, 16, 300), (1200, 3, 40, 100), (5000, 2, 256, 37), (800, 4, 8, 160), (2500, 2, 64, 1)]
    data = []
    for i, (v, dd, c, subset) in enumerate(specs):
        a = _correlated(900 + i, v, dd, c)
        cand = np.random.RandomState(i).permutation(v)
        data.append((a, c, list(itertools.combinations(range(dd), 2)), cand, subset))

    def build(i, seed):
        a, c, pairs, cand, subset = data[i]
        m = get_measure("batch_mi")(a, ncentroids=c, batch_size=20, selection_size=4, device="cuda:0",
                                    keep_unselected=True, generator=Generator(seed))
        m.init(pairs, [int(j) for j in cand[1:]])
        return m

    for keep in (True,):
        alone, tails = [], []
        for i in range(len(specs)):
            m = build(i, 50 + i)
            alone.append(m.run_greedy(data[i][4], [int(data[i][3][0])], None))
            tails.append(m._generator.u32())
        ms = [build(i, 50 + i) for i in range(len(specs))]
        multi = EfficientBatchMI.run_greedy_multi(ms, [d[4] for d in data], [[int(d[3][0])] for d in data])
        for i in range(len(specs)):
            assert multi[i][0] == alone[i][0] and multi[i][1] == alone[i][1], f"chunk {i}"
            assert ms[i]._generator.u32() == tails[i]
            a, c, pairs, cand, subset = data[i]
            r = O.BatchMI(a, c, pairs).run_greedy(cand[1:], cand[:1], subset, 20, 4, O.Rng(50 + i))
            assert multi[i][0] == r["S"].tolist()
            cache = ms[i].cache
            Nc, ac, bc, nc = O.BatchMI(a, c, pairs).counts()
            assert cache["n"] == 1 + len(multi[i][1])
    with pytest.raises(AssertionError):
        g = Generator(1)
        m1, m2 = build(0, 1), build(1, 2)
        m1._generator = m2._generator = g
        EfficientBatchMI.run_greedy_multi([m1, m2], [10, 10], [[0], [0]])


def test_recycled_device_blocks_change_nothing_and_can_be_trimmed(env):
    """Handles are re-created per chunk (chunk.py:21-53) and their device blocks come from the library's pool of parked blocks
    (acav_trim_device_cache, include/acav_hip.h): a selection whose every buffer is a RECYCLED block -- still holding another
    chunk's lists, tables and counters -- gives the oracle's result, and trimming returns the parked bytes to the driver."""
    import gc
    torch, acav, O = env
    from acav100m_amd.rng import Generator
    from acav100m_amd.subset_selection import get_measure
    specs = [(6000, 2, 64, 240), (6000, 2, 64, 240), (4000, 3, 32, 120)]

    def run(i, v, dd, c, subset):
        a = _correlated(700 + i, v, dd, c)
        cand = np.random.RandomState(70 + i).permutation(v)
        pairs = list(itertools.combinations(range(dd), 2))
        m = get_measure("batch_mi")(a, ncentroids=c, batch_size=20, selection_size=4, device="cuda:0", keep_unselected=True,
                                    generator=Generator(90 + i))
        m.init(pairs, [int(j) for j in cand[1:]])
        got = m.run_greedy(subset, [int(cand[0])], None)
        r = O.BatchMI(a, c, pairs).run_greedy(cand[1:], cand[:1], subset, 20, 4, O.Rng(90 + i))
        assert got[0] == r["S"].tolist(), f"run {i}"
        del m

    acav.trim_device_cache()
    for i, sp in enumerate(specs):  # runs 1 and 2 take the blocks run 0 parked (same sizes; a smaller third one takes what fits)
        run(i, *sp)
        gc.collect()
    freed = acav.trim_device_cache()
    assert freed > 0, "destroyed handles parked nothing"
    assert acav.trim_device_cache() == 0
    run(0, *specs[0])  # and from an empty pool again


def test_greedy_loop_speed_does_not_depend_on_the_process_history():
    """The loop's three streams (content, positions, generator) need three HARDWARE queues; which queue the runtime gives a new
    stream depends on every stream the process created and destroyed before (round 5: two idle k-means handles cost the loop a
    third of its speed).  acav_mi_create now measures whether two of its streams serialise and replaces the one that does
    (mi_separate_queues).  Ten k-means handles created, used and partly destroyed before the handle exists, two of them still alive:
    the 1M-clip loop must run within 10 % of the clean process's speed (tools/mi_history_probe.py, loop time by ACAV_MI_TIMING;
    best of two runs each, fresh processes)."""
    import re
    import subprocess
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "mi_history_probe.py")

    def run(*argv):
        best = None
        for _ in range(2):
            r = subprocess.run([sys.executable, tool] + [str(a) for a in argv], capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr[-1500:]
            us = float(re.search(r"us_per_iteration ([\d.]+)", r.stdout).group(1))
            best = us if best is None else min(best, us)
        return best

    clean = run(0, 0, 6000)
    history = run(10, 2, 6000)
    assert history <= 1.10 * clean + 0.5, (clean, history)
