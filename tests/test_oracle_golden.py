"""The oracle (oracle/acav_oracle.c) pinned against the golden vectors recorded from the reference
itself (tests/golden/gen_golden.py) and against sklearn's mutual_info_score.  CPU only."""
import itertools
import os

import numpy as np
import pytest

from oracle import oracle as O


def test_rng_streams(golden_dir):
    g = np.load(os.path.join(golden_dir, "rng.npz"))
    for s in (0, 1, 1234):
        r = O.Rng(s)
        assert np.array_equal(r.rand(7, 5), g[f"s{s}_rand_7x5"])
        assert np.array_equal(r.randperm(10), g[f"s{s}_perm10"])
        assert np.array_equal(r.randperm(1000), g[f"s{s}_perm1000"])
        assert np.array_equal(r.randperm(100003)[:2000], g[f"s{s}_perm100003_head"])
        assert np.array_equal(r.rand(3), g[f"s{s}_rand_after"])
        assert np.array_equal((r.rand(16, 8) * np.float32(1e-5)).astype(np.float32), g[f"s{s}_init_16x8"])


def test_python_shuffle_is_stdlib(golden_dir):
    import random
    g = np.load(os.path.join(golden_dir, "rng.npz"))
    for s in (0, 1, 1234):
        random.seed(s)
        lst = list(range(1000))
        random.shuffle(lst)
        assert lst == list(g[f"s{s}_pyshuffle1000"])


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_kmeans_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"kmeans_{name}.npz"))
    x, K, b = g["x"], int(g["K"]), int(g["b"])
    n, d = x.shape
    km = O.KMeans(d, K, O.Rng(int(g["seed"])))
    assert np.array_equal(km.centers, g["centers0"])
    warm, means = [], []
    for e in range(int(g["epochs"])):
        km.lr = 0.1 ** (2 + e // 5)
        for t in range(n // b):
            w = km.count < 10 * K
            m, best = km.add(x[t * b:(t + 1) * b], return_best=True)
            means.append(m)
            if w:
                warm.append(best)
        c, cnt, count, fb = km.get_state()
        # identical labels at every step => identical update arithmetic => bit-identical centres
        assert np.array_equal(c, g[f"centers_e{e}"])
        assert np.array_equal(cnt, g[f"counts_e{e}"])
        assert count == int(g[f"count_e{e}"]) and fb == int(g[f"fallback_e{e}"])
    assert np.array_equal(np.stack(warm), g["warm_best"])
    np.testing.assert_allclose(np.array(means), g["add_means"], rtol=2e-6)
    lab, _ = km.calc_best(x)
    near_ties = int((g["top2_gap"] < 1e-3).sum())
    assert np.array_equal(lab, g["labels"]) and np.array_equal(lab, g["labels_onebatch"]), \
        f"labels differ from the reference ({near_ties} reference near-ties below 1e-3)"
    c, cnt, count, fb = km.get_state()
    cnt2 = cnt.copy()
    cnt2[::3] = 1.0
    km.set_state(None, cnt2, count, fb)
    lab2, _ = km.calc_best(x)
    assert np.array_equal(lab2, g["labels_doctored"])
    assert int(g["n_changed_by_discount"]) > 0  # the discount path really was exercised


@pytest.mark.parametrize("name", ["d1024_k256", "d2048_k1024"])
def test_kmeans_golden_at_baseline_shapes(name):
    """SURVEY 8(c) G2 at BASELINE's shapes (cfg2/3: d = 1024, K = 256; cfg4 visual: d = 2048, K = 1024), 32 768 rows of
    overlapping clusters: the oracle's TRAINING from the reference's seed follows the reference step for step (labels of
    all 2 048 SGD steps) to bit-identical centres (sha256 of the reference's trained centres), its labels equal the
    reference's on every natural row (also under the doctored under-use discount), and on 4 096 rows pushed onto the
    bisector of their two closest centres -- where the reference's GEMM order and the canonical multi-segment fold
    (oracle/acav_oracle.c orc_dot) are free to disagree -- every disagreement is an exact-arithmetic tie within the
    stated fp32 bound (tests/_census.py).  The count is printed, not hidden."""
    import hashlib
    from tests import _census as Z
    g, x, n, d, K = Z.load_case(name)
    b = int(g["b"])
    km = O.KMeans(d, K, O.Rng(int(g["seed"])))
    t = 0
    for e in range(int(g["epochs"])):
        km.lr = 0.1 ** (2 + e // 5)
        for i in range(n // b):
            _, best = km.add(x[i * b:(i + 1) * b], return_best=True)
            assert np.array_equal(best, g["step_best"][t]), f"SGD step {t}: labels differ from the reference's"
            t += 1
    c, cnt, count, fb = km.get_state()
    assert hashlib.sha256(c.tobytes()).hexdigest() == str(g["centers_sha256"]), "trained centres differ from the reference's"
    if "centers" in g:
        assert np.array_equal(c, g["centers"])
    else:
        assert np.array_equal(c[:8], g["centers_head"])
    assert np.array_equal(cnt, g["counts"]) and count == int(g["count"]) and fb == int(g["fallback"])
    reinit = (0.7, 5.0)
    lab, _ = km.calc_best(x)
    nm, _ = Z.census(x, c, cnt, count, reinit, lab, g["labels"], f"oracle, {name}, natural rows")
    assert nm == 0  # smallest reference top-2 gap on these rows is O(1): nothing to excuse
    xb = Z.bisector(g, x, c)
    labb, _ = km.calc_best(xb)
    nb, _ = Z.census(xb, c, cnt, count, reinit, labb, g["bis_labels"], f"oracle, {name}, bisector rows", max_ulps=8)
    assert nb > 0  # the two fp32 evaluations DO disagree on exact ties: the census is not vacuous
    assert int((g["bis_top2_gap"] == 0).sum()) > 1000  # the reference itself saw exact fp32 ties on these rows
    cnt2 = cnt.copy()
    cnt2[::3] = 1.0
    km.set_state(None, cnt2, count, fb)
    labd, _ = km.calc_best(x)
    Z.census(x, c, cnt2, count, reinit, labd, g["labels_doctored"], f"oracle, {name}, doctored discount")
    assert int(g["n_changed_by_discount"]) > 0


def _mi_case(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"mi_{name}.npz"))
    a, c, seed = g["assignments"], int(g["C"]), int(g["seed"])
    v, dd = a.shape
    assert a.max() + 1 == c
    pairs = list(itertools.combinations(range(dd), 2))
    subset = round(float(g["ratio"]) * v)
    cand = list(g["shuffled"])
    return g, a, c, seed, pairs, subset, [cand[0]], cand[1:]


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_mi_golden_teacher_forced(golden_dir, name):
    g, a, c, seed, pairs, subset, start, cand = _mi_case(golden_dir, name)
    mi = O.BatchMI(a, c, pairs)
    res = mi.run_greedy(cand, start, subset, 20, 4, O.Rng(seed), trace=True, forced_pos=g["pick_pos"])
    # randperm stream + in-place Fisher-Yates + drop-B + ascending re-queue == the reference
    assert np.array_equal(res["ids"], g["ids"])
    assert np.array_equal(res["S"], g["S"]) and len(res["GAIN"]) == len(g["GAIN"])
    ref_mean = g["scores"].astype(np.float64).mean(-1)
    np.testing.assert_allclose(res["scores"], ref_mean, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(res["GAIN"], g["GAIN"], rtol=1e-5, atol=1e-6)
    k = g["pick_pos"].shape[1]
    for t in range(len(g["ids"])):
        if set(res["pos"][t]) != set(g["pick_pos"][t]):
            srt = np.sort(ref_mean[t])[::-1]
            assert abs(srt[k - 1] - srt[k]) <= 2e-6 * max(abs(srt[k - 1]), 1e-3)


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_mi_dense_literal_scores(golden_dir, name):
    """the literal fp32 dense calc_MI restatement reproduces the reference scores to fp32 round-off,
    and the canonical float64 closed form agrees with it"""
    g, a, c, seed, pairs, subset, start, cand = _mi_case(golden_dir, name)
    mi = O.BatchMI(a, c, pairs)
    mi.add_samples(start)
    for t in range(len(g["ids"])):
        ids = g["ids"][t]
        dense = mi.scores_dense(ids)
        np.testing.assert_allclose(dense, g["scores"][t], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(mi.scores_canon(ids), dense.astype(np.float64).mean(-1), rtol=2e-6, atol=1e-7)
        mi.add_samples(ids[g["pick_pos"][t]])


def test_mi_known_answer_sklearn():
    from sklearn.metrics import mutual_info_score
    rs = np.random.RandomState(0)
    v, c = 1500, 32
    comp = rs.randint(0, c, v)
    a = np.stack([comp, np.where(rs.rand(v) < 0.5, comp, rs.randint(0, c, v))], 1).astype(np.int64)
    mi = O.BatchMI(a, c, [(0, 1)])
    sel = list(range(1000))
    mi.add_samples(sel)
    ids = np.arange(1000, 1020)
    s = mi.scores_canon(ids)
    for w, i in enumerate(ids):
        rows = a[sel + [int(i)]]
        kat = mutual_info_score(rows[:, 0], rows[:, 1])
        assert abs(s[w] - kat) <= 1e-6 * kat


def test_mi_free_running_invariants():
    rs = np.random.RandomState(4)
    v, dd, c = 1200, 3, 16
    a = rs.randint(0, c, (v, dd)).astype(np.int64)
    pairs = list(itertools.combinations(range(dd), 2))
    cand = [int(i) for i in rs.permutation(v)]
    res = O.BatchMI(a, c, pairs).run_greedy(cand[1:], cand[:1], 241, 20, 4, O.Rng(5), trace=True)
    assert len(res["S"]) == 241 == len(set(res["S"])) and cand[0] not in res["S"]   # batch.py:238-241
    assert len(res["GAIN"]) == 244                                                  # GAIN is not cut
    with pytest.raises(RuntimeError):
        O.BatchMI(a[:30], c, pairs).run_greedy(list(range(1, 30)), [0], 28, 20, 4, O.Rng(0))


def test_canonical_arithmetic_definitions():
    """the oracle's canonical dot / sumsq are what their header says (pure-Python restatement)"""
    rs = np.random.RandomState(1)
    for d in (1, 31, 32, 33, 200, 256, 257, 1024, 1300):
        v = rs.randn(d).astype(np.float32)
        w = rs.randn(d).astype(np.float32)
        tot = None
        for j0 in range(0, d, 256):  # 256-column segments, each ONE sequential FMA chain, folded left to right
            acc = np.float32(0)
            for j in range(j0, min(d, j0 + 256)):  # exact product in float64, one rounding
                acc = np.float32(np.float64(v[j]) * np.float64(w[j]) + np.float64(acc))
            tot = acc if tot is None else np.float32(tot + acc)
        assert O.dot(v, w) == float(tot)
        p = np.zeros(32, np.float32)
        for j in range(d):
            p[j & 31] = np.float32(np.float64(v[j]) * np.float64(v[j]) + np.float64(p[j & 31]))
        gq = [(p[4 * q] + p[4 * q + 1]) + (p[4 * q + 2] + p[4 * q + 3]) for q in range(8)]
        ss = ((gq[0] + gq[1]) + (gq[2] + gq[3])) + ((gq[4] + gq[5]) + (gq[6] + gq[7]))
        assert O.sumsq(v) == float(ss)
        assert O.norm2(v) == float(np.float32(np.sqrt(np.float32(ss))) ** 2)


@pytest.mark.parametrize("name", ["a", "b", "c"])
@pytest.mark.parametrize("measure", ["mi", "mem_mi"])
def test_mi_exact_greedy_golden(golden_dir, name, measure):
    """'mi' / 'mem_mi' (measures/mi.py:150-192, 284-412): exact greedy, all remaining candidates scored per
    iteration, first maximum committed.  The reference's two formulations already disagree with each other on
    near-ties, so the pin is teacher-forced: replaying the reference's picks, (1) the canonical float64 scores equal
    its fp32 score vectors to 2e-6 absolute at every iteration, (2) every pick of the reference IS a canonical
    maximum (gap <= 1e-12: it only ever differs inside exact ties of the canonical score), (3) S and GAIN follow."""
    g = np.load(os.path.join(golden_dir, f"mi_exact_{name}.npz"))
    a, c, subset, cand = g["assignments"], int(g["C"]), int(g["subset"]), g["shuffled"]
    pairs = list(itertools.combinations(range(a.shape[1]), 2))
    idx, ref_sc = g[f"{measure}_idx"], g[f"{measure}_scores"]
    r = O.BatchMI(a, c, pairs).run_exact(cand[1:], cand[:1], subset, forced_idx=idx, trace=True)
    n = subset - 2
    assert r["iters"] == n == len(idx)
    assert np.array_equal(r["S"], g[f"{measure}_S"][1:]) and g[f"{measure}_S"][0] == cand[0]
    assert np.allclose(r["GAIN"], g[f"{measure}_GAIN"], rtol=0, atol=2e-6)
    for t in range(n):
        L = ref_sc.shape[1] - t
        mine = r["scores"][t, :L]
        assert np.isnan(r["scores"][t, L:]).all() and np.isnan(ref_sc[t, L:]).all()
        assert np.max(np.abs(mine - ref_sc[t, :L].astype(np.float64))) <= 2e-6
        assert mine.max() - mine[idx[t]] <= 1e-12 * max(1.0, abs(mine.max()))
        assert r["argmax"][t] == int(np.argmax(mine))  # first maximum
    # free-running: the tie-free mem_mi runs are reproduced exactly
    if measure == "mem_mi" and name in ("a", "b"):
        free = O.BatchMI(a, c, pairs).run_exact(cand[1:], cand[:1], subset)
        assert np.array_equal(free["S"], g["mem_mi_S"][1:])


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_ami_exact_greedy_golden(golden_dir, name):
    """'ami' (measures/mi.py:212-259, registered in measures/__init__.py): exact greedy on the adjusted score
    (MI - EMI) / (mean entropy - EMI) with the reference's own one-term-per-cell EMI.  Teacher-forced on the reference's
    picks: the canonical float64 scores (table look-ups + canon_exp over integer counts) equal its fp32 score vectors to
    1e-5 relative (BASELINE's bar; observed 3.6e-7) at every iteration, every reference pick is within that band of the
    canonical maximum (the scores tie in fp32 exactly as those of `mi`), S and GAIN follow."""
    g = np.load(os.path.join(golden_dir, f"mi_ami_{name}.npz"))
    a, c, subset, cand = g["assignments"], int(g["C"]), int(g["subset"]), g["shuffled"]
    pairs = list(itertools.combinations(range(a.shape[1]), 2))
    idx, ref_sc = g["idx"], g["scores"]
    m = O.BatchMI(a, c, pairs)
    m.set_measure("ami")
    r = m.run_exact(cand[1:], cand[:1], subset, forced_idx=idx, trace=True)
    n = subset - 2
    assert r["iters"] == n == len(idx)
    assert np.array_equal(r["S"], g["S"][1:]) and g["S"][0] == cand[0]
    np.testing.assert_allclose(r["GAIN"], g["GAIN"], rtol=1e-5, atol=1e-7)
    worst = 0.0
    for t in range(n):
        L = ref_sc.shape[1] - t
        mine, ref = r["scores"][t, :L], ref_sc[t, :L].astype(np.float64)
        assert np.isnan(r["scores"][t, L:]).all()
        rel = np.abs(mine - ref) / np.maximum(np.abs(ref), 1e-6)
        worst = max(worst, float(rel.max()))
        assert rel.max() <= 1e-5
        assert mine.max() - mine[idx[t]] <= 1e-5 * max(1e-6, abs(mine.max()))
        assert r["argmax"][t] == int(np.argmax(mine))
    print(f"ami_{name}: max relative score difference from the reference {worst:.2e}")
    # the deterministic exp of the canonical score against libm
    import math
    for x in (0.0, -1e-9, -0.3, -3.7, -50.0, -700.0):
        assert abs(O.lib().orc_canon_exp(x) - math.exp(x)) <= 4e-16 * math.exp(x)


def _contrastive_data(seed, n, vis, aud):
    """the generator recipe of tests/golden/gen_golden.py:gen_contrastive (case c only stores the first columns)"""
    rs = np.random.RandomState(seed)
    comp = rs.randint(0, 12, n)
    cv, ca = rs.randn(12, vis).astype(np.float32), rs.randn(12, aud).astype(np.float32)
    return (cv[comp] + 0.5 * rs.randn(n, vis)).astype(np.float32), (ca[comp] + 0.5 * rs.randn(n, aud)).astype(np.float32)


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_contrastive_oracle_vs_reference(golden_dir, name):
    """oracle/contrastive_ref.py against the reference module's own run: nn.Linear init from the seeded torch stream
    (bit-exact), per-batch loss / accuracy over the epochs (gradients accumulating, AdamW amsgrad, per-epoch lr),
    trained parameters and infer() scores within 1e-4 relative (the reference sums its GEMMs in MKL's order)."""
    from oracle import oracle as O
    from oracle import contrastive_ref as CR
    g = np.load(os.path.join(golden_dir, f"contrastive_{name}.npz"))
    vis, aud, B, nb, epochs = int(g["vis"]), int(g["aud"]), int(g["B"]), int(g["nb"]), int(g["epochs"])
    out = int(g["out"]) if "out" in g.files else min(vis, aud)
    rng = O.Rng(int(g["seed"]))
    wv, bv = CR.linear_init(lambda n: rng.rand(n), out, vis)
    wa, ba = CR.linear_init(lambda n: rng.rand(n), out, aud)
    init = {"visual_linear.weight": wv, "visual_linear.bias": bv, "audio_linear.weight": wa, "audio_linear.bias": ba}
    for k, v in init.items():
        ref = g["p0_" + k]
        assert np.array_equal(v[:len(ref)], ref), f"{k}: seeded init differs from torch"
    if name == "c":
        visual, audio = _contrastive_data(int(g["data_seed"]), B * nb, vis, aud)
        assert np.array_equal(visual[:, :8], g["visual"]) and np.array_equal(audio[:, :8], g["audio"])
        sums = [float(v.astype(np.float64).sum()) for v in init.values()]
        np.testing.assert_allclose(sums, g["p0_full_sum"], rtol=1e-12)
    else:
        visual, audio = g["visual"], g["audio"]
    m = CR.Contrastive(wv, bv, wa, ba)
    losses, accs = [], []
    for epoch in range(epochs):
        lr = CR.epoch_lr(epoch, epochs, float(g["base_lr"]), int(g["warm"]))
        assert abs(lr - g["lrs"][epoch]) < 1e-15
        for bi in range(nb):
            sl = slice(bi * B, (bi + 1) * B)
            lo, ac = m.train_batch(visual[sl], audio[sl], lr)
            losses.append(lo)
            accs.append(ac)
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-5)
    np.testing.assert_allclose(accs, g["accs"], atol=1e-4)
    for k, v in zip(init, m.p):
        ref = g["p1_" + k]
        np.testing.assert_allclose(v[:len(ref)], ref, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(m.infer(visual, audio), g["infer"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", ["nmi_a", "nmi_b", "nmi_c", "constant_a"])
def test_nmi_and_constant_exact_greedy_golden(golden_dir, name):
    """EfficientNMI (2 MI / max(mean entropy, eps), measures/mi.py:262-271) and ConstantMeasure (every candidate scores 1,
    mi.py:274-281): classes of the reference that its get_measure registry does not name, run through its own _run_greedy
    (tests/golden/gen_golden.py mi_nmi).  Teacher-forced on the reference's picks: the canonical float64 scores equal its fp32
    score vectors to 1e-5 relative at every iteration, every reference pick is within that band of the canonical maximum, S and
    GAIN follow; the constant measure takes the first remaining candidate every time, free-running too."""
    g = np.load(os.path.join(golden_dir, f"mi_{name}.npz"))
    a, c, subset, cand = g["assignments"], int(g["C"]), int(g["subset"]), g["shuffled"]
    pairs = list(itertools.combinations(range(a.shape[1]), 2))
    idx, ref_sc = g["idx"], g["scores"]
    measure = name.split("_")[0]
    m = O.BatchMI(a, c, pairs)
    m.set_measure(measure)
    r = m.run_exact(cand[1:], cand[:1], subset, forced_idx=idx, trace=True)
    n = subset - 2
    assert r["iters"] == n == len(idx)
    assert np.array_equal(r["S"], g["S"][1:]) and g["S"][0] == cand[0]
    np.testing.assert_allclose(r["GAIN"], g["GAIN"], rtol=1e-5, atol=1e-7)
    worst = 0.0
    for t in range(n):
        L = ref_sc.shape[1] - t
        mine, ref = r["scores"][t, :L], ref_sc[t, :L].astype(np.float64)
        rel = np.abs(mine - ref) / np.maximum(np.abs(ref), 1e-6)
        worst = max(worst, float(rel.max()))
        assert rel.max() <= 1e-5
        assert mine.max() - mine[idx[t]] <= 1e-5 * max(1e-6, abs(mine.max()))
    print(f"{name}: max relative score difference from the reference {worst:.2e}")
    if measure == "constant":
        assert (idx == 0).all() and np.array_equal(g["S"][1:], cand[1:subset - 1])
        m2 = O.BatchMI(a, c, pairs)
        m2.set_measure("constant")
        free = m2.run_exact(cand[1:], cand[:1], subset)
        assert np.array_equal(free["S"], g["S"][1:]) and (free["GAIN"] == 1.0).all()
