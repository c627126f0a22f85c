"""GPU parity tests for the k-means hot path: HIP (through the C ABI / KMeans mirror) vs the
oracle and vs the golden vectors captured from the reference.  Bit-exact for labels, centres,
counts (integer / canonical-fp32 paths); 1e-5 relative for the returned mean distances."""
import os

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


def _mixture(seed, n, d, g, noise=0.3):
    rs = np.random.RandomState(seed)
    cen = rs.randn(g, d).astype(np.float32)
    return (cen[rs.randint(0, g, n)] + noise * rs.randn(n, d)).astype(np.float32)


def test_single_hip_runtime(env):
    """libacav_hip.so must share torch's HIP runtime (one libamdhip64 in the process)."""
    with open("/proc/self/maps") as f:
        libs = {line.split()[-1] for line in f if "libamdhip64" in line}
    assert len(libs) == 1, libs


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_golden_train_and_assign(env, golden_dir, name):
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    g = np.load(os.path.join(golden_dir, f"kmeans_{name}.npz"))
    x, K, b = g["x"], int(g["K"]), int(g["b"])
    n, d = x.shape
    acav.manual_seed(int(g["seed"]))
    km = KMeans(None, d, K)
    assert np.array_equal(km._centers0, g["centers0"])
    km.to("cuda:0")
    xt = torch.from_numpy(x).cuda()
    means = []
    for e in range(int(g["epochs"])):
        km.lr = 0.1 ** (2 + e // 5)
        for t in range(n // b):
            means.append(km.add(xt[t * b:(t + 1) * b]))
        assert np.array_equal(km.centers.numpy(), g[f"centers_e{e}"]), f"epoch {e}: centres differ from the reference"
        assert np.array_equal(km.counts.numpy(), g[f"counts_e{e}"])
        assert km.count == int(g[f"count_e{e}"]) and km.fallback == int(g[f"fallback_e{e}"])
    np.testing.assert_allclose(np.array(means), g["add_means"], rtol=1e-5)
    lab, _ = km.calc_best(xt)
    assert np.array_equal(lab.cpu().numpy(), g["labels"])
    # per-batch calls give the same labels and the reference's per-batch means
    lm = [km.calc_best(xt[t:t + b])[1] for t in range(0, n, b)]
    np.testing.assert_allclose(np.array(lm), g["label_means"], rtol=1e-5)
    # doctored usage counts: the under-use discount must fire exactly as in the reference
    cnt = km.counts.clone()
    cnt[::3] = 1.0
    km.counts = cnt
    lab2, _ = km.calc_best(xt)
    assert np.array_equal(lab2.cpu().numpy(), g["labels_doctored"])


def test_bulk_train_equals_stepwise_and_oracle(env):
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    n, d, K, b = 4096, 160, 40, 32
    x = _mixture(5, n, d, K)
    acav.manual_seed(11)
    km = KMeans(None, d, K).to("cuda:0")
    ref = O.KMeans(d, K, O.Rng(11))
    xt = torch.from_numpy(x).cuda()
    for e in range(2):
        km.train_epoch(xt, b, lr=0.01)
        ref.train_epoch(x, b, lr=0.01)
    assert np.array_equal(km.centers.numpy(), ref.centers)
    assert np.array_equal(km.counts.numpy(), ref.counts)
    assert km.count == ref.count == 2 * n


@pytest.mark.parametrize("n,d,K,b", [(2048, 256, 40, 32), (4096, 1024, 256, 32), (1536, 512, 20, 24),
                                     (2048, 768, 64, 32), (4096, 256, 40, 64), (8192, 1024, 256, 256),
                                     (2048, 88, 40, 32), (2048, 704, 64, 32), (1024, 64, 16, 32), (1536, 352, 24, 32),
                                     (2048, 128, 32, 32), (1024, 1000, 24, 32),
                                     # more than 32 centre groups (256 < K): the wide kernel's exchange, not the narrow one's
                                     (4096, 256, 264, 32), (4096, 256, 300, 32), (8192, 64, 512, 32),
                                     # round 6: every pass width of the lean sweep (tp_sweep_lean<NU>), full and masked: 9-16 centre
                                     # groups (NU = 8), a ragged batch under it, 63 groups of 16 centres with a ragged batch (NU = 32,
                                     # masked), 64 full (NU = 32, the unmasked path)
                                     (2048, 256, 128, 32), (2000, 512, 100, 20), (3072, 128, 1000, 24), (4096, 128, 1024, 32)])
def test_persistent_epoch_kernel(env, n, d, K, b):
    """acav_kmeans_train takes the persistent one-launch path for d % 4 == 0, d <= 1024, b <= 32:
    centres resident in LDS, per-step device-scope key exchange.  Must equal the oracle bit for bit
    (ragged centre groups K=40/20, ragged row group b=24, ragged column blocks d=88/352/704/1000 -- the audio /
    visual layer widths of the real pipeline -- and warm-up steps inside the launch); b > 32 takes the per-step
    kernels."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    x = _mixture(d + K, n, d, K)
    acav.manual_seed(13)
    km = KMeans(None, d, K).to("cuda:0")
    ref = O.KMeans(d, K, O.Rng(13))
    xt = torch.from_numpy(x).cuda()
    for e in range(2):
        km.train_epoch(xt, b, lr=0.01)
        ref.train_epoch(x, b, lr=0.01)
        assert np.array_equal(km.centers.numpy(), ref.centers), f"epoch {e}"
        assert np.array_equal(km.counts.numpy(), ref.counts)
    assert km.count == ref.count
    lab, _ = km.calc_best(xt)
    assert np.array_equal(lab.cpu().numpy(), ref.calc_best(x)[0])
    # the lr fallback inside the persistent kernel
    km2 = KMeans(None, d, K).to("cuda:0")
    ref2 = O.KMeans(d, K, O.Rng(13))
    km2.centers, km2.counts, km2.count = ref2.centers, ref2.counts, 0
    acav.manual_seed(14)
    rng2 = O.Rng(14)
    ref2.rng = rng2
    km2.train_epoch(xt, b, lr=0.3)
    ref2.train_epoch(x, b, lr=0.3)
    assert km2.fallback == ref2.fallback
    assert km2.fallback > 0 or K >= 1000  # (with a thousand centres no batch of 24-32 rows puts four on one centre: the path is not taken)
    assert np.array_equal(km2.centers.numpy(), ref2.centers)


@pytest.mark.parametrize("n,d,K", [(1000, 88, 32), (777, 130, 70), (513, 64, 300), (4096, 1024, 256),
                                   (300, 2304, 32), (64, 8, 3), (1, 32, 5)])
def test_assign_matches_oracle(env, n, d, K):
    """calc_best over ragged shapes: d not a multiple of 32 or 4, K not a multiple of 32, K > 256
    (several centre groups), n not a multiple of the row tile, single row."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    x = _mixture(n + d, n, d, max(2, K // 2))
    rs = np.random.RandomState(K)
    centers = (x[rs.randint(0, n, K)] + 0.05 * rs.randn(K, d)).astype(np.float32)
    counts = rs.randint(0, 60, K).astype(np.float32)  # some centres fall under the threshold
    count = 10 * K + 1000
    km = KMeans(None, d, K)
    km.centers, km.counts, km.count = centers, counts, count
    km.to("cuda:0")
    ref = O.KMeans(d, K, O.Rng(0), centers=centers)
    ref.set_state(None, counts, count)
    assert (counts < ref.threshold()).any()
    lab, mean = km.calc_best(torch.from_numpy(x).cuda())
    lab_ref, mean_ref = ref.calc_best(x)
    assert np.array_equal(lab.cpu().numpy(), lab_ref)
    assert abs(mean - mean_ref) <= 1e-5 * abs(mean_ref)
    # host-pointer entry (the library stages the batch itself)
    lab_h, _ = km.calc_best(x)
    assert np.array_equal(lab_h.numpy(), lab_ref)


@pytest.mark.parametrize("case", ["clustered", "unstructured", "duplicates", "two_groups", "tiny_gap",
                                  "offset", "offset_unstructured", "offset_tiny_gap", "offset_discounted"])
def test_bf16_filter_path_is_bit_identical(env, case):
    """calc_best(need_mean=False) = bf16-MFMA filter + exact re-check of the rows whose top-2 gap is below
    the proven bound.  Labels must equal the oracle's for ANY data: well separated clusters (no re-check),
    unstructured data (most rows re-checked), duplicated centres (exact ties -> first index), K > 256,
    near-ties far below bf16 resolution, and embeddings with a large common component ("offset": every feature
    shifted by +25, norms 25x the cluster spread) -- there the filter multiplies by the CENTRED centres, whose
    bound scales with the spread of the centres, unless a centre is under-used (the row constant does not survive
    the division by r: "offset_discounted" takes the raw filter and re-checks nearly everything)."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    rs = np.random.RandomState(7)
    n, d, K = 3000, 256, 64
    if case == "clustered":
        rs2 = np.random.RandomState(11)
        cen = (4.0 * rs2.randn(K, d)).astype(np.float32)
        x = (cen[rs2.randint(0, K, n)] + 0.3 * rs2.randn(n, d)).astype(np.float32)
        centers = (cen + 0.05 * rs2.randn(K, d)).astype(np.float32)
    elif case == "unstructured":
        x = rs.randn(n, d).astype(np.float32)
        centers = rs.randn(K, d).astype(np.float32)
    elif case == "duplicates":
        x = _mixture(2, n, d, K // 2)
        centers = np.stack([x[rs.randint(0, n)] for _ in range(K)]).astype(np.float32)
        centers[40] = centers[3]
        centers[41] = centers[3]
        x[:200] = centers[3] + 0.01 * rs.randn(200, d).astype(np.float32)
    elif case == "two_groups":
        K, d = 300, 128
        x = _mixture(3, n, d, 100)
        centers = np.stack([x[rs.randint(0, n)] for _ in range(K)]).astype(np.float32)
    elif case in ("offset", "offset_discounted"):
        rs2 = np.random.RandomState(12)
        cen = (1.0 * rs2.randn(K, d)).astype(np.float32)
        x = (cen[rs2.randint(0, K, n)] + 0.1 * rs2.randn(n, d) + 25.0).astype(np.float32)
        centers = (cen + 0.02 * rs2.randn(K, d) + 25.0).astype(np.float32)
    elif case == "offset_unstructured":
        x = (rs.randn(n, d) + 25.0).astype(np.float32)
        centers = (rs.randn(K, d) + 25.0).astype(np.float32)
    elif case == "offset_tiny_gap":
        x = (_mixture(4, n, d, K // 2) + 25.0).astype(np.float32)
        base = np.stack([x[rs.randint(0, n)] for _ in range(K // 2)]).astype(np.float32)
        centers = np.concatenate([base, base + (1e-4 * rs.randn(K // 2, d)).astype(np.float32)])
    else:  # tiny_gap: pairs of centres 1e-4 apart -- invisible to bf16, decided by the exact pass
        x = _mixture(4, n, d, K // 2)
        base = np.stack([x[rs.randint(0, n)] for _ in range(K // 2)]).astype(np.float32)
        centers = np.concatenate([base, base + (1e-4 * rs.randn(K // 2, d)).astype(np.float32)])
    counts = rs.randint(0, 80, K).astype(np.float32)  # some centres under the usage threshold -> discount
    if case in ("offset", "offset_unstructured", "offset_tiny_gap"):
        counts = np.full(K, 1000, np.float32)          # nobody under-used: the centred filter applies
    count = 10 * K + 2000
    km = KMeans(None, d, K)
    km.centers, km.counts, km.count = centers, counts, count
    km.to("cuda:0")
    ref = O.KMeans(d, K, O.Rng(0), centers=centers)
    ref.set_state(None, counts, count)
    xt = torch.from_numpy(x).cuda()
    lab, mean = km.calc_best(xt, need_mean=False)
    assert mean is None
    launches, rows, rechecked = km.filter_stats()
    assert launches == 1 and rows == n
    lab_ref, _ = ref.calc_best(x)
    assert np.array_equal(lab.cpu().numpy(), lab_ref), f"{(lab.cpu().numpy() != lab_ref).sum()} labels differ"
    lab_exact, _ = km.calc_best(xt)  # exact path
    assert np.array_equal(lab_exact.cpu().numpy(), lab_ref)
    print(f"{case}: {rechecked}/{n} rows needed the exact re-check")
    if case in ("tiny_gap", "offset_tiny_gap", "offset_discounted"):
        assert rechecked > n // 2   # the exact pass really decided these
    if case == "unstructured":      # (half-precision operands since round 5: the window is 8 x narrower than bf16's)
        assert rechecked > n // 10
    if case == "offset":
        assert rechecked < n // 20  # centred centres: the common component costs nothing


@pytest.mark.parametrize("d,K,kind", [(88, 32, "clustered"), (88, 256, "clustered"), (88, 64, "unstructured"), (200, 48, "tiny_gap"),
                                      (100, 300, "clustered"), (36, 16, "discounted"), (2308, 32, "clustered")])
def test_filter_on_widths_that_are_not_stage_multiples(env, d, K, kind, monkeypatch):
    """Round 5: a view whose width is not a multiple of the filter's 32-column stage -- SlowFast's 88-wide layer
    (clustering/code/models/slowfast.py:31) -- used to take the guarded exact sweep.  Now calc_best(need_mean=False) runs the
    bf16 filter + candidate / exact re-check on zero-padded copies of the rows and the centres (88 -> 96; fmaf(0, 0, s) == s in
    every canonical chain).  Labels must equal the oracle's on the UNPADDED data: separated clusters (filter decides), unstructured
    rows and centre pairs 1e-4 apart (the re-check decides, on the padded copies), K > 256 (group pairs + merge), under-used
    centres (no centring), a width past 2304; and equal what the guarded exact sweep gives (ACAV_FILTER_PAD=0)."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    rs = np.random.RandomState(d + K)
    n = 3000
    if kind == "unstructured":
        x = rs.randn(n, d).astype(np.float32)
        centers = rs.randn(K, d).astype(np.float32)
    elif kind == "tiny_gap":
        x = _mixture(4, n, d, K // 2)
        base = np.stack([x[rs.randint(0, n)] for _ in range(K // 2)]).astype(np.float32)
        centers = np.concatenate([base, base + (1e-4 * rs.randn(K // 2, d)).astype(np.float32)])
    else:
        cen = (2.0 * rs.randn(K, d)).astype(np.float32)
        x = (cen[rs.randint(0, K, n)] + 0.3 * rs.randn(n, d)).astype(np.float32)
        centers = (cen + 0.05 * rs.randn(K, d)).astype(np.float32)
    counts = np.full(K, 1000, np.float32)
    if kind == "discounted":
        counts = rs.randint(0, 80, K).astype(np.float32)
    count = 10 * K + 2000
    km = KMeans(None, d, K)
    km.centers, km.counts, km.count = centers, counts, count
    km.to("cuda:0")
    ref = O.KMeans(d, K, O.Rng(0), centers=centers)
    ref.set_state(None, counts, count)
    lab_ref, _ = ref.calc_best(x)
    xt = torch.from_numpy(x).cuda()
    lab, mean = km.calc_best(xt, need_mean=False)
    assert mean is None
    launches, rows, rechecked = km.filter_stats()
    assert launches == 1 and rows == n, "the padded filter did not run"
    assert np.array_equal(lab.cpu().numpy(), lab_ref), f"{(lab.cpu().numpy() != lab_ref).sum()} labels differ"
    print(f"d={d} K={K} {kind}: {rechecked}/{n} rows needed the exact re-check")
    if kind == "clustered":
        assert rechecked < n // 10
    if kind == "unstructured":
        assert rechecked > n // 50  # the re-check (on the padded copies) really decided rows
    if kind == "tiny_gap":
        assert rechecked > n // 2
    # host rows go through the same path; a second sweep after a state change rebuilds the padded copies
    lab_h, _ = km.calc_best(x, need_mean=False)
    assert np.array_equal(lab_h.numpy(), lab_ref)
    km.centers = centers[::-1].copy()
    ref2 = O.KMeans(d, K, O.Rng(0), centers=centers[::-1].copy())
    ref2.set_state(None, counts, count)
    assert np.array_equal(km.calc_best(xt, need_mean=False)[0].cpu().numpy(), ref2.calc_best(x)[0])
    monkeypatch.setenv("ACAV_FILTER_PAD", "0")  # the guarded exact sweep, as before round 5
    assert np.array_equal(km.calc_best(xt, need_mean=False)[0].cpu().numpy(), ref2.calc_best(x)[0])
    assert km.filter_stats()[0] == 3


@pytest.mark.parametrize("scale,K", [(1.0, 64), (1e-4, 64), (3e4, 64), (2e-7, 32), (1e-4, 300), (1.0, 300)])
def test_filter_half_precision_operands_any_data_scale(env, scale, K):
    """Round 5: the filter's MFMA operands are IEEE half (unit roundoff 2^-11; bf16: 2^-8), scaled by exact powers of two so that
    the narrow exponent range of half does not matter: centres always (the copy holds sc c'), rows only when the data's own scale
    sits outside half's comfortable range (sx != 1: the XS instantiations).  Labels must equal the oracle's at EVERY data scale --
    1e-4 and 2e-7 (every element far below half's smallest normal: all of it would flush without the scaling), 3e4 (every element
    beyond half's largest finite value), with K <= 256 and K > 256 -- and for rows with a few ENORMOUS elements (10^9 x the data's scale: half has no
    such number after any scaling; the bound refuses the row and the exact path labels it)."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    rs = np.random.RandomState(int(K + 1000 * abs(np.log10(scale))))
    n, d = 3000, 256
    cen = (2.0 * rs.randn(K, d)).astype(np.float32)
    x = (cen[rs.randint(0, K, n)] + 0.5 * rs.randn(n, d)).astype(np.float32)
    centers = (cen + 0.05 * rs.randn(K, d)).astype(np.float32)
    x[:200] = centers[rs.randint(0, K, 200)] + (1e-4 * rs.randn(200, d)).astype(np.float32)  # near-ties for the re-check
    x = (x * np.float32(scale)).astype(np.float32)
    centers = (centers * np.float32(scale)).astype(np.float32)
    x[7, 3] = np.float32(1e9) * np.float32(scale)    # outliers 10^9 x the data's scale: half cannot hold them whatever the scaling
    x[8, :5] = np.float32(-3e8) * np.float32(scale)  # (fp32 itself still can: 1e30 would overflow ||x||^2 and the dots in fp32 too)
    counts = np.full(K, 1000, np.float32)
    count = 10 * K + 2000
    km = KMeans(None, d, K)
    km.centers, km.counts, km.count = centers, counts, count
    km.to("cuda:0")
    ref = O.KMeans(d, K, O.Rng(0), centers=centers)
    ref.set_state(None, counts, count)
    lab_ref, _ = ref.calc_best(x)
    xt = torch.from_numpy(x).cuda()
    lab, _ = km.calc_best(xt, need_mean=False)
    launches, rows, rechecked = km.filter_stats()
    assert launches == 1 and rows == n
    assert np.array_equal(lab.cpu().numpy(), lab_ref), f"{(lab.cpu().numpy() != lab_ref).sum()} labels differ"
    print(f"scale {scale} K {K}: {rechecked}/{n} rows needed the exact re-check")
    assert 2 <= rechecked < n // 2   # the two outlier rows always; the separated rows never
    # under-used centres (no centring, raw norms in the bound) at this scale too
    counts2 = rs.randint(0, 80, K).astype(np.float32)
    km.counts = counts2
    ref.set_state(None, counts2, count)
    assert np.array_equal(km.calc_best(xt, need_mean=False)[0].cpu().numpy(), ref.calc_best(x)[0])


@pytest.mark.parametrize("K", [2, 300])
@pytest.mark.parametrize("scale", [1.0, 2.0 ** -18])
def test_filter_bound_against_aligned_worst_case_roundings(env, scale, K):
    """Random data never comes near the filter's acceptance bound (a dot's rounding errors add up like a random walk, ~sqrt(d) below
    the worst case) -- which is how a bound HALF as large as the proof needs survived four rounds of tests.  This data is built so
    that every operand rounding of the half-precision filter pushes the same way: 448 coordinates with x = c' = 1 + 2^-11 (a tie,
    rounds DOWN to 1), 448 with x = -c' = 1 + 3 2^-11 (a tie, rounds UP to 1 + 2^-9), and 128 exactly representable coordinates that
    move the TRUE x.c' across zero from row to row.  Two centres +c', -c' (mean 0: the centred copy is c' itself, equal norms).  For
    ~220 rows the filter's own distances name the WRONG centre with an apparent gap of up to 0.87 of twice the proven bound: they
    must all come out undecided and be labelled by the exact path.  A leading constant a factor 2 too small accepts ~90 of them.
    K = 300 (the (tile, group) pair kernel, k_assign_merge and the emission pass): 149 more pairs +-w of centres, w = +-1.25 in a sign
    pattern that is balanced inside both rounding blocks (x.w = 0 exactly, every partial sum of the mean exact: the centred copy is
    still c' to the bit) and 376 further away than the two that matter."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    d = 1024
    e = 2.0 ** -11
    cp = np.concatenate([np.full(448, 1 + e), np.full(448, -(1 + 3 * e)), np.ones(128)]).astype(np.float32)
    ms = np.arange(0, 2000, 2)  # row i: the exact block holds m 2^-16 (a half number); true tie at m = 448.4, the filter's at m = 896.9;
    # below m ~ 300 and above ~ 1500 the gap exceeds twice the bound: those rows the filter decides itself
    n = len(ms)
    x = np.empty((n, d), np.float32)
    x[:, :448], x[:, 448:896] = 1 + e, 1 + 3 * e
    x[:, 896:] = (ms * 2.0 ** -16)[:, None]
    x[1::2] *= -1  # every other row mirrored: the other centre wins
    cen = [cp, -cp]
    rs = np.random.RandomState(3)
    for _ in range((K - 2) // 2):
        w = np.zeros(d, np.float32)
        w[:448] = 1.25 * rs.permutation(np.repeat([1.0, -1.0], 224))
        w[448:896] = 1.25 * rs.permutation(np.repeat([1.0, -1.0], 224))
        cen += [w, -w]
    centers = (np.stack(cen) * np.float32(scale)).astype(np.float32)
    x = (x * np.float32(scale)).astype(np.float32)
    # what an unguarded half-precision filter would say, and the truth (float64)
    x16, c16 = (x / np.float32(scale)).astype(np.float16).astype(np.float64), cp.astype(np.float16).astype(np.float64)
    apparent = np.where(x16 @ c16 > 0, 0, 1)
    truth = np.where((x / np.float32(scale)).astype(np.float64) @ cp.astype(np.float64) > 0, 0, 1)
    fooled = int((apparent != truth).sum())
    assert fooled > 200, fooled  # the construction works: these rows' filter distances name the wrong centre
    counts = np.full(K, 1000, np.float32)
    km = KMeans(None, d, K)
    km.centers, km.counts, km.count = centers, counts, 10 * K + int(counts.sum())
    km.to("cuda:0")
    lab, _ = km.calc_best(torch.from_numpy(x).cuda(), need_mean=False)
    launches, rows, undecided = km.filter_stats()
    assert launches == 1 and rows == n
    ref = O.KMeans(d, K, O.Rng(0), centers=centers)
    ref.set_state(None, counts, 10 * K + int(counts.sum()))
    want = ref.calc_best(x)[0]
    got = lab.cpu().numpy()
    assert np.array_equal(got, want), f"{int((got != want).sum())} labels differ from the oracle"
    assert np.array_equal(want, truth)  # (no row of this sweep is a float32-level tie)
    assert fooled <= undecided < n - 200, (undecided, fooled)  # ... and the far ends of the sweep were decided by the filter
    if K > 2:
        cand_rows, cand_pairs, full_rows = km.recheck_stats()
        assert cand_rows + full_rows == undecided and cand_pairs >= 2 * cand_rows  # both centres of the pair are candidates of every such row
    print(f"scale {scale} K {K}: {fooled} rows whose filter distances name the wrong centre, {undecided} of {n} undecided")


@pytest.mark.parametrize("K", [4, 300])
def test_candidate_threshold_against_aligned_worst_case_roundings(env, K):
    """The second "proven" inequality of the filter (cand_threshold, acav_kmeans_assign.hip): a centre may be left out of an undecided
    row's candidate list only if its filter value lies more than 2 E + slack above the row's filter minimum.  Random data never tests
    the 2: a true minimiser sits within a fraction of E of the filter's minimum.  Here the true minimiser T is the THIRD centre by the
    filter's values, 1.15 .. 1.64 E above the minimum: the aligned-rounding construction of the test above (+-c': every operand rounding
    pushes the filter's value of the true winner T UP by ~0.82 E and that of the named centre N DOWN by as much) plus a DECOY centre D
    whose operands are exactly representable (its filter value is its true value) placed at 0.6 of the way from N to T -- the filter's
    top two are (N, D), truth is T < D < N.  The row is undecided (D - N < 2 E) and T must be among the emitted candidates; with a
    window of E instead of 2 E it is not, and the exact evaluation of {N, D} returns D.  K = 300: the same through the (tile, group)
    pairs, k_assign_merge's thresholds and the emission pass."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    d = 1024
    e = 2.0 ** -11
    g = 4.0
    cp = np.concatenate([np.full(448, 1 + e), np.full(448, -(1 + 3 * e)), np.ones(64), np.zeros(64)]).astype(np.float32)
    dc = np.concatenate([np.zeros(960), np.full(64, g)]).astype(np.float32)  # the decoy: support where +-c' is zero
    ms = np.arange(900, 1160, 2)  # the exact block holds m 2^-16 in 64 coordinates: x.c' = -0.8758 + m / 1024 (true), -1.7517 + m / 1024 (filter)
    n = len(ms)
    qf = -(448 * (8 * e + 16 * e * e)) + ms / 1024.0  # the filter's dot of the un-mirrored row with c' (exact arithmetic of rounded operands)
    qt = -(448 * (4 * e + 8 * e * e)) + ms / 1024.0   # the true dot
    assert (qf < 0).all() and (qt > 0).all()           # every row: the filter names -c', the truth is +c'
    cnA = float(np.float64(cp.astype(np.float64) @ cp.astype(np.float64)))
    gap = -4.0 * qf                                     # v(T) - v(N) by the filter
    t = (g * g * 64 - cnA + 0.4 * qf) / (2 * 64 * g)    # v(D) = v(N) + 0.6 gap  <=>  -2 * 64 g t + (|D|^2 - |c'|^2) = -0.4 qf
    t = np.round(t * 2.0 ** 14) / 2.0 ** 14             # on half's grid at 2^-4 .. 2^-3: the products t g and their sums are exact
    x = np.empty((n, d), np.float32)
    x[:, :448], x[:, 448:896] = 1 + e, 1 + 3 * e
    x[:, 896:960] = (ms * 2.0 ** -16)[:, None]
    x[:, 960:] = 0.0
    x[1::2] *= -1  # every other row mirrored: the roles of +c' and -c' swap (the decoy coordinates stay positive)
    x[:, 960:] = t[:, None].astype(np.float32)
    cen = [cp, -cp, dc, -dc]
    rs = np.random.RandomState(3)
    for _ in range((K - 4) // 2):  # far pairs +-w, balanced inside both rounding blocks: x.w = 0, the centres' mean stays exactly 0
        w = np.zeros(d, np.float32)
        w[:448] = 1.25 * rs.permutation(np.repeat([1.0, -1.0], 224))
        w[448:896] = 1.25 * rs.permutation(np.repeat([1.0, -1.0], 224))
        cen += [w, -w]
    centers = np.stack(cen).astype(np.float32)
    # the construction, checked in float64: truth and what a half-precision filter sees
    x64, c64 = x.astype(np.float64), centers.astype(np.float64)
    true_d = (x64 * x64).sum(1)[:, None] - 2 * x64 @ c64.T + (c64 * c64).sum(1)[None]
    truth = true_d.argmin(1)
    xh, ch = x.astype(np.float16).astype(np.float64), centers.astype(np.float16).astype(np.float64)
    filt = -2 * xh @ ch.T + (c64 * c64).sum(1)[None]
    order = np.argsort(filt, axis=1)
    named, second, third = order[:, 0], order[:, 1], order[:, 2]
    want_named = np.where(np.arange(n) % 2 == 0, 1, 0)
    assert np.array_equal(named, want_named) and (second == 2).all() and np.array_equal(third, 1 - want_named)
    assert np.array_equal(truth, 1 - want_named)  # the true minimiser is the filter's THIRD
    E = (2.103e-3 + 1.248e-4) * 32.0 * np.sqrt((x64 * x64).sum(1)) + 2.0 ** -20 * (np.sqrt((x64 * x64).sum(1)) + 32.0) ** 2  # filter_bound, to ~1 %
    rel = (filt[np.arange(n), third] - filt[np.arange(n), named]) / E
    assert rel.min() > 1.1 and rel.max() < 1.8, (rel.min(), rel.max())  # beyond a window of E, inside the proven 2 E
    counts = np.full(K, 1000, np.float32)
    km = KMeans(None, d, K)
    km.centers, km.counts, km.count = centers, counts, 10 * K + int(counts.sum())
    km.to("cuda:0")
    lab, _ = km.calc_best(torch.from_numpy(x).cuda(), need_mean=False)
    launches, rows, undecided = km.filter_stats()
    assert launches == 1 and rows == n and undecided == n
    ref = O.KMeans(d, K, O.Rng(0), centers=centers)
    ref.set_state(None, counts, 10 * K + int(counts.sum()))
    want = ref.calc_best(x)[0]
    got = lab.cpu().numpy()
    assert np.array_equal(want, truth)
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {n} labels differ: the true minimiser was not among the candidates"
    cand_rows, cand_pairs, full_rows = km.recheck_stats()
    assert cand_rows + full_rows == n and cand_pairs >= 3 * cand_rows  # N, D and T of every row
    print(f"K {K}: true minimiser third by the filter at {rel.min():.2f} .. {rel.max():.2f} E above its minimum in {n} rows; "
          f"{cand_rows} rows by candidates ({cand_pairs} pairs), {full_rows} by the full exact sweep")


def test_position_tag_may_reorder_the_filters_top_two(env):
    """The filter's compare-free top-2 scan overwrites the 7 low mantissa bits of every distance with the centre's position in the
    lane (acav_kmeans_assign.hip: "tagged", c = 1.6e-5 of |d1| + |d2| in the acceptance test).  Two centres whose filter values are
    closer than the tag's reach come out of the scan in the order of their POSITIONS, not of their values.  Integer data (every product
    and partial sum exact, filter value == true value): pairs of centres 1 .. 100 units of the last place apart at ~2^21, the nearer
    one at the higher position -- the scan names the wrong one of every pair; the rows must be undecided and the exact path must
    return the true minimiser (ties: the first index)."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    d, K = 64, 130
    rs = np.random.RandomState(11)
    centers = np.zeros((K, d), np.float32)
    base = rs.randint(-8, 9, size=d).astype(np.float32) * 64.0
    for k in range(K):
        centers[k] = base
        centers[k, k % d] += 16.0 * (K - k)  # all centres close together, each in its own direction; later centres nearer to `base`
    n = 512
    x = (base[None, :] + rs.randint(-2, 3, size=(n, d)).astype(np.float32)) * 1.0
    x[:, 0] += 512.0  # a common offset: distances ~ 2.6e5 .. with ulp-level differences between neighbours in k
    counts = np.full(K, 1000, np.float32)
    km = KMeans(None, d, K)
    km.centers, km.counts, km.count = centers, counts, 10 * K + int(counts.sum())
    km.to("cuda:0")
    lab, _ = km.calc_best(torch.from_numpy(x).cuda(), need_mean=False)
    ref = O.KMeans(d, K, O.Rng(0), centers=centers)
    ref.set_state(None, counts, 10 * K + int(counts.sum()))
    want = ref.calc_best(x)[0]
    assert np.array_equal(lab.cpu().numpy(), want)
    x64, c64 = x.astype(np.float64), centers.astype(np.float64)
    true_d = (x64 * x64).sum(1)[:, None] - 2 * x64 @ c64.T + (c64 * c64).sum(1)[None]
    assert np.array_equal(want, true_d.argmin(1))  # integer data: float32 evaluates these distances exactly


@pytest.mark.parametrize("d,K,switch", [(1024, 300, None), (128, 257, None), (1024, 200, ("ACAV_FILTER_NW", "8")),
                                        (1024, 200, ("ACAV_ASSIGN_EMIT", "1")), (1024, 600, ("ACAV_FILTER_GS", "0")),
                                        (96, 1024, None)])
@pytest.mark.parametrize("scale", [2.0 ** -20, 1e3])
def test_filter_scaled_rows_every_instantiation(env, d, K, switch, scale, monkeypatch):
    """The scaled-row (XS) instantiations of EVERY tile form of the filter -- 8 waves / (tile, group) pairs (K > 256, wide rows), 4-wave
    pairs (narrow rows), 8 waves with K <= 256, the emission pass, the group loop -- on overlapping mixtures (a tenth of the rows
    undecided, candidates emitted) at data scales that switch the row scaling on.  An inline-asm pack conversion in round 5 was wrong
    exactly here (a third of the labels with 8 waves, a handful with 4-wave pairs; right in the default K <= 256 form, which is all the
    tests of that day covered): found by tools/stress_parity.py when it began to vary the data's scale."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    if switch:
        monkeypatch.setenv(*switch)
    n = 3000
    rs = np.random.RandomState(5)
    cen0 = rs.randn(K, d).astype(np.float32)
    x = ((cen0[rs.randint(0, K, n)] + rs.randn(n, d).astype(np.float32)) * np.float32(scale)).astype(np.float32)
    centers = ((rs.randn(K, d).astype(np.float32)[rs.randint(0, K, K)] + rs.randn(K, d).astype(np.float32)) * np.float32(scale)).astype(np.float32)
    counts = np.full(K, 500, np.float32)
    km = KMeans(None, d, K)
    km.centers, km.counts, km.count = centers, counts, 10 * K + int(counts.sum())
    km.to("cuda:0")
    xt = torch.from_numpy(x).cuda()
    ref = O.KMeans(d, K, O.Rng(0), centers=centers)
    ref.set_state(None, counts, 10 * K + int(counts.sum()))
    want = ref.calc_best(x)[0]
    for rep in range(2):
        lab, _ = km.calc_best(xt, need_mean=False)
        assert km.filter_stats()[0] == rep + 1, "the filter did not run"
        bad = int((lab.cpu().numpy() != want).sum())
        assert bad == 0, f"{bad} of {n} labels differ from the oracle (undecided {km.filter_stats()[2]}, re-check {km.recheck_stats()})"
    assert np.array_equal(km.calc_best(xt)[0].cpu().numpy(), want)


@pytest.mark.parametrize("spread", [0.02, 0.006, 0.0004])
def test_filter_underflow_unit_and_row_headroom(env, spread, monkeypatch):
    """The half-precision filter's bound charges 2^-25 per operand element below half's normal range when the device keeps half
    subnormals (checked once per device with the product's own instructions), 2^-14 when it is told to assume a flush
    (ACAV_FILTER_SUBNORMAL=0): same labels, and never MORE undecided rows with the tighter unit.  Rows that are noise-dominated over
    tiny centres (row elements 50 ... 750 x the largest centre element: the scaled-row instantiation at 0.006 and below) must not fall
    off the filter wholesale -- the first round-5 version refused every row of the 0.006 case (scaled rows left half's range)."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    n, d, K = 20_000, 1024, 256
    rs = np.random.RandomState(5)
    cen = (spread * rs.randn(K, d)).astype(np.float32)
    x = (cen[rs.randint(0, K, n)] + 0.3 * rs.randn(n, d)).astype(np.float32)
    centers = (cen + 0.2 * spread * rs.randn(K, d)).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    und = {}
    for unit in ("kept", "flush"):
        if unit == "flush":
            monkeypatch.setenv("ACAV_FILTER_SUBNORMAL", "0")
        km = KMeans(None, d, K)
        km.centers, km.counts, km.count = centers, np.full(K, 1000, np.float32), 10 * K + 200_000
        km.to("cuda:0")
        lab, _ = km.calc_best(xt, need_mean=False)
        und[unit] = km.filter_stats()[2]
        exact, _ = km.calc_best(xt)
        assert torch.equal(lab, exact)
    print(f"spread {spread}: undecided {und}")
    assert und["kept"] <= und["flush"] and und["kept"] < n // 3
    ref = O.KMeans(d, K, O.Rng(0), centers=centers)
    ref.set_state(None, np.full(K, 1000, np.float32), 10 * K + 200_000)
    assert np.array_equal(lab.cpu().numpy()[:2048], ref.calc_best(x[:2048])[0])


@pytest.mark.parametrize("mode", ["plain", "pool_overflow", "cand_off", "discounted", "many_ties", "emission_pass"])
def test_candidate_restricted_recheck(env, mode, monkeypatch):
    """Round 4: a row the bf16 filter cannot decide is settled by the exact canonical distances of its CANDIDATE centres only
    (those the acceptance inequality cannot rule out against the filter's minimum; k_assign_cand), not by a sweep over all
    K.  Heavily overlapping clusters (a fifth of the rows undecided, a handful of candidates each): labels == the exact sweep on every
    row and == the oracle; the statistics show the candidate path did the work.  `pool_overflow`: a pair pool of 1 000
    entries -- the rows that do not fit take the full exact sweep, same labels.  `cand_off`: ACAV_ASSIGN_CAND=0 restores the
    round-3 behaviour.  `discounted`: under-used centres (distance / r, raw filter).  `many_ties`: 40 identical centres --
    more than 16 candidates per row -> full sweep, first index wins.  `emission_pass`: ACAV_ASSIGN_EMIT=1, the two-pass form."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    if mode == "pool_overflow":
        monkeypatch.setenv("ACAV_CAND_PAIR_CAP", "1000")
    if mode == "cand_off":
        monkeypatch.setenv("ACAV_ASSIGN_CAND", "0")
    if mode == "emission_pass":  # lean filter + a second pass over the listed rows instead of the in-place emission
        monkeypatch.setenv("ACAV_ASSIGN_EMIT", "1")
    n, d, K = 40_000, 1024, 256
    rs = np.random.RandomState(21)
    # centre spread << noise radius: the bf16 filter of rounds 1-4 left most of these rows undecided, the half-precision one a
    # fifth (the fraction does not depend on the spread: gap and bound scale together -- tools/exp/recheck_spread_probe.py)
    cen = (0.02 * rs.randn(K, d)).astype(np.float32)
    x = (cen[rs.randint(0, K, n)] + 0.3 * rs.randn(n, d)).astype(np.float32)
    centers = (cen + 0.004 * rs.randn(K, d)).astype(np.float32)
    if mode == "many_ties":
        centers[100:140] = centers[7]
    counts = np.full(K, 1000, np.float32)
    if mode == "discounted":
        counts[rs.randint(0, K, 40)] = 3.0
    count = 10 * K + 200_000
    km = KMeans(None, d, K)
    km.centers, km.counts, km.count = centers, counts, count
    km.to("cuda:0")
    xt = torch.from_numpy(x).cuda()
    for rep in range(2):  # twice: the sweep's control block must come back zeroed
        lab, _ = km.calc_best(xt, need_mean=False)
        _, rows, rechecked = km.filter_stats()
        cand_rows, cand_pairs, full_rows = km.recheck_stats()
        exact, _ = km.calc_best(xt)
        assert torch.equal(lab, exact), f"{mode}: {(lab != exact).sum().item()} labels differ from the exact sweep"
        assert rows == n and rechecked == cand_rows + full_rows
        print(f"{mode}: undecided {rechecked}/{n}: {cand_rows} rows by {cand_pairs} candidate pairs, {full_rows} by the full sweep")
        if mode in ("plain", "emission_pass"):
            assert cand_rows > n // 8 and full_rows < cand_rows // 10 and cand_pairs >= cand_rows
        if mode == "discounted":  # the raw (uncentred) filter's bound is wide: many rows exceed 16 candidates
            assert cand_rows > 0 and cand_rows + full_rows > n // 4
        if mode == "pool_overflow":
            assert 0 < cand_rows < 1000 and full_rows > n // 8
        if mode == "cand_off":
            assert cand_rows == 0 and full_rows > n // 8
    ref = O.KMeans(d, K, O.Rng(0), centers=centers)
    ref.set_state(None, counts, count)
    idx = np.sort(rs.choice(n, 4096, replace=False))
    assert np.array_equal(lab.cpu().numpy()[idx], ref.calc_best(x[idx])[0])
    if mode == "many_ties":
        assert not np.isin(lab.cpu().numpy(), np.arange(100, 140)).any()  # the duplicates lose to index 7


@pytest.mark.parametrize("switch", [("ACAV_FILTER_NT", "0"), ("ACAV_ASSIGN_EXACT_ONLY", "1"),
                                    ("ACAV_NO_PERSISTENT", "1")])
def test_diagnostic_switches_keep_the_results(env, switch, monkeypatch):
    """The A/B switches select other kernels, never other results: the round-1 wave layout of the filter, the default
    cache policy on its row DMA, the exact sweep alone, and per-step launches instead of the persistent epoch kernel."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    monkeypatch.setenv(*switch)
    n, d, K, b = 4096, 256, 64, 32
    x = _mixture(21, n, d, 40)
    acav.manual_seed(4)
    km = KMeans(None, d, K).to("cuda:0")
    ref = O.KMeans(d, K, O.Rng(4))
    xt = torch.from_numpy(x).cuda()
    for epoch in range(2):
        km.train_epoch(xt, b, lr=0.01)
        ref.train_epoch(x, b, 0.01)
    assert np.array_equal(km.centers.numpy(), ref.centers) and np.array_equal(km.counts.numpy(), ref.counts)
    lab, mean = km.calc_best(xt, need_mean=False)
    lab_ref, _ = ref.calc_best(x)
    assert np.array_equal(lab.cpu().numpy(), lab_ref)


def test_exact_ties_take_first_index(env):
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    d, K = 64, 40
    rs = np.random.RandomState(3)
    centers = rs.randn(K, d).astype(np.float32)
    centers[17] = centers[5]
    centers[33] = centers[5]  # three identical centres: first index must win
    x = (centers[5][None, :] + 0.01 * rs.randn(50, d)).astype(np.float32)
    km = KMeans(None, d, K)
    km.centers, km.counts, km.count = centers, np.full(K, 100, np.float32), 10 * K + 5
    km.to("cuda:0")
    lab, _ = km.calc_best(torch.from_numpy(x).cuda())
    assert (lab.cpu().numpy() == 5).all()


@pytest.mark.parametrize("b,lr,d,K", [(32, 0.01, 96, 24), (16, 0.01, 96, 24), (48, 0.01, 256, 40),
                                      (32, 0.2, 96, 24), (7, 0.5, 96, 24), (32, 0.01, 130, 24),
                                      (32, 0.01, 2304, 20), (20, 0.01, 200, 70), (256, 0.01, 64, 33),
                                      # large (DDP global) batches: several row groups per workgroup (2 / 4 / 8), ragged
                                      # column blocks, centre groups and row groups
                                      (128, 0.01, 1024, 256), (256, 0.01, 256, 256), (250, 0.01, 132, 300),
                                      (1024, 0.0005, 512, 64)])
def test_step_matches_oracle(env, b, lr, d, K):
    """add() at several batch sizes / shapes: DMA path (d % 4 == 0, incl. d > 1024 = two LDS stages and
    ragged 256-column blocks), MFMA fallback (d = 130), ragged centre / row groups; lr large enough to
    trigger the fallback (sgd_clustering.py:116-119)."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    steps = (40 if d < 1000 else 16) if b < 128 else 12
    x = _mixture(b, b * steps, d, K)
    acav.manual_seed(2)
    km = KMeans(None, d, K).to("cuda:0")
    km.lr = lr
    ref = O.KMeans(d, K, O.Rng(2), lr=lr)
    for t in range(steps):
        m = km.add(torch.from_numpy(x[t * b:(t + 1) * b]).cuda())
        m_ref = ref.add(x[t * b:(t + 1) * b])
        assert abs(m - m_ref) <= 1e-5 * abs(m_ref) + 1e-30
    assert np.array_equal(km.centers.numpy(), ref.centers)
    assert np.array_equal(km.counts.numpy(), ref.counts)
    assert km.fallback == ref.fallback
    if lr >= 0.2:
        assert km.fallback > 0


def test_full_size_assign_properties(env):
    """BASELINE config 2 shape (1M x 1024, K=256): checked through size-independent properties --
    labels of a random row subset recomputed on their own are identical (batch invariance), and
    equal the oracle's on a sample."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    n, d, K = 1_000_000, 1024, 256
    gen = torch.Generator(device="cuda").manual_seed(0)
    cen = torch.randn(K, d, device="cuda", generator=gen)
    comp = torch.randint(0, K, (n,), device="cuda", generator=gen)
    x = torch.empty(n, d, device="cuda")
    for s in range(0, n, 100_000):
        x[s:s + 100_000] = cen[comp[s:s + 100_000]] + 0.3 * torch.randn(100_000, d, device="cuda", generator=gen)
    rs = np.random.RandomState(0)
    centers = (cen.cpu().numpy() + 0.1 * rs.randn(K, d)).astype(np.float32)
    km = KMeans(None, d, K)
    km.centers, km.counts, km.count = centers, np.full(K, 3000, np.float32), 10 * K + n
    km.to("cuda:0")
    lab, mean = km.calc_best(x)
    lab = lab.cpu().numpy()
    assert lab.min() >= 0 and lab.max() < K and np.isfinite(mean)
    idx = np.sort(rs.choice(n, 3000, replace=False))
    sub = x[torch.from_numpy(idx).cuda()].contiguous()
    lab_sub, _ = km.calc_best(sub)
    assert np.array_equal(lab_sub.cpu().numpy(), lab[idx])
    ref = O.KMeans(d, K, O.Rng(0), centers=centers)
    ref.set_state(None, np.full(K, 3000, np.float32), 10 * K + n)
    lab_ref, _ = ref.calc_best(sub.cpu().numpy())
    assert np.array_equal(lab_ref, lab[idx])
    # the planted component is recovered
    assert (lab == comp.cpu().numpy()).mean() > 0.999


def test_errors_are_loud(env):
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    km = KMeans(None, 8, 4)
    with pytest.raises(acav.AcavError):
        km.add(np.zeros((4, 8), np.float32))  # not on a GPU
    with pytest.raises(acav.AcavError):
        km.to("cpu")


def test_edge_cases_empty_single_and_limits(env):
    """empty inputs, single rows / single centre-adjacent sizes, and the documented limits fail cleanly (no crash,
    no hang): the C ABI reports ACAV_EINVAL -> ValueError like the reference's asserts."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    d, K = 16, 5
    acav.manual_seed(3)
    km = KMeans(None, d, K).to("cuda:0")
    ref = O.KMeans(d, K, O.Rng(3))
    x = _mixture(1, 200, d, K)
    xt = torch.from_numpy(x).cuda()
    km.train_epoch(xt, 32, lr=0.01)
    ref.train_epoch(x, 32, lr=0.01)
    # empty batch: no labels, state untouched
    empty = torch.empty((0, d), dtype=torch.float32, device="cuda")
    lab, _ = km.calc_best(empty, need_mean=False)
    assert lab.numel() == 0
    km.train_epoch(empty, 32, lr=0.01)             # floor(0 / 32) = 0 steps
    km.train_epoch(xt[:31], 32, lr=0.01)           # fewer rows than one batch: drop_last -> 0 steps
    assert np.array_equal(km.centers.numpy(), ref.centers) and km.count == ref.count
    # one row, one step of one row
    one, _ = km.calc_best(xt[:1])
    assert one.cpu().numpy().tolist() == ref.calc_best(x[:1])[0].tolist()
    km.train_epoch(xt[:3], 1, lr=0.01)
    ref.train_epoch(x[:3], 1, lr=0.01)
    assert np.array_equal(km.centers.numpy(), ref.centers) and km.count == ref.count
    # limits
    with pytest.raises(ValueError):
        km.train_epoch(xt, 2048, lr=0.01)          # batch above the supported 1024
    with pytest.raises((ValueError, acav.AcavError)):
        km.calc_best(torch.zeros((4, d + 1), device="cuda"))  # wrong feature width
    with pytest.raises((ValueError, acav.AcavError)):
        KMeans(None, d, 0).to("cuda:0")            # the handle is created on the move to the device


def test_train_epoch_multi_equals_one_by_one(env):
    """acav_kmeans_train_multi: several clusterings' persistent kernels in flight together (two 1024-d views at K=256 =
    2 x 128 workgroups = every CU; plus a third that no longer fits and waits its turn, and one that is not eligible for
    the persistent kernel at all).  State of every clustering == its own train_epoch == the oracle."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    shapes = [(1024, 256), (1024, 256), (512, 64), (2048, 40)]  # d=2048: per-step path
    n, b = 4096 + 2560, 32
    xs = [_mixture(40 + i, n, d, K) for i, (d, K) in enumerate(shapes)]
    xts = [torch.from_numpy(x).cuda() for x in xs]
    acav.manual_seed(23)
    kms = [KMeans(None, d, K).to("cuda:0") for d, K in shapes]
    refs = [O.KMeans(d, K, O.Rng(0), centers=km._centers0.copy()) for (d, K), km in zip(shapes, kms)]
    lab_rs = np.random.RandomState(9)
    for epoch in range(2):
        warm = []
        for km, (d, K) in zip(kms, shapes):
            need = km.warmup_steps(b, n // b)
            warm.append(lab_rs.randint(0, K, (need, b)).astype(np.int64))
        KMeans.train_epoch_multi(kms, xts, b, lr=0.01, warm_bests=warm)
        for km, ref, x, w in zip(kms, refs, xs, warm):
            for t in range(n // b):
                xb = x[t * b:(t + 1) * b]
                if t < len(w):
                    ref.apply_update(xb, w[t], 0.01)  # counts the batch as add() does
                else:
                    ref.lr = 0.01
                    ref.add(xb)
            assert np.array_equal(km.centers.numpy(), ref.centers), f"epoch {epoch}"
            assert np.array_equal(km.counts.numpy(), ref.counts) and km.count == ref.count


@pytest.mark.parametrize("n,d,K,b", [(4096, 2048, 1024, 32), (2048, 2048, 256, 32), (2048, 1280, 300, 32), (1600, 1536, 40, 20),
                                     (2048, 1792, 64, 16), (700, 2048, 24, 7)])
def test_split_column_persistent_kernel(env, n, d, K, b):
    """k_train_persistent_split (1024 < d <= 2048, d % 256 == 0): pairs of workgroups own the two column halves of 16
    centres; the canonical segment fold and the canonical ||c||^2 chains run across the pair through tagged hand-offs.
    Two epochs (warm-up inside the first launch) == the oracle bit for bit, in ONE launch per epoch (no per-step
    launches); ragged centre groups (K = 300 / 40 / 24), a second half of 1 / 2 / 3 column blocks (d = 1280 / 1536 /
    1792), ragged row groups (b = 20 / 7) and the lr fallback."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    os.environ["ACAV_SPLIT_MINK"] = "1"  # the product takes this kernel from K = 512 on; the small shapes exercise its edges
    os.environ["ACAV_TALL"] = "0"         # ... and where both fit, the one-workgroup tall kernel (test below) would come first
    x = _mixture(d + K, n, d, K)
    acav.manual_seed(21)
    km = KMeans(None, d, K).to("cuda:0")
    ref = O.KMeans(d, K, O.Rng(21))
    xt = torch.from_numpy(x).cuda()
    for e in range(2):
        km.train_epoch(xt, b, lr=0.01)
        ref.train_epoch(x, b, lr=0.01)
        assert np.array_equal(km.centers.numpy(), ref.centers), f"epoch {e}"
        assert np.array_equal(km.counts.numpy(), ref.counts)
    assert km.count == ref.count
    launches, gave_up = km.train_stats()
    assert launches == 2 and gave_up == 0, f"persistent launches {launches}, fallbacks {gave_up}: the split kernel was not taken (or gave up)"
    lab, _ = km.calc_best(xt)
    assert np.array_equal(lab.cpu().numpy(), ref.calc_best(x)[0])
    km2 = KMeans(None, d, K).to("cuda:0")
    ref2 = O.KMeans(d, K, O.Rng(13))
    km2.centers, km2.counts, km2.count = ref2.centers, ref2.counts, 0
    acav.manual_seed(14)
    ref2.rng = O.Rng(14)
    km2.train_epoch(xt, b, lr=0.3)
    ref2.train_epoch(x, b, lr=0.3)
    assert km2.fallback == ref2.fallback and (km2.fallback > 0 or K > 64)  # many centres: the batch never piles up on one
    assert np.array_equal(km2.centers.numpy(), ref2.centers)
    os.environ.pop("ACAV_SPLIT_MINK", None)
    os.environ.pop("ACAV_TALL", None)


@pytest.mark.parametrize("n,d,K,b", [(2048, 1408, 256, 32), (1536, 2304, 64, 32), (2048, 2048, 512, 32), (3072, 1408, 1024, 32),
                                     (1280, 1100, 40, 20), (1024, 2304, 300, 7), (1024, 1056, 24, 32)])
def test_tall_persistent_kernel(env, n, d, K, b):
    """Round 4: rows wider than 1024 columns on ONE workgroup per centre group (k_train_persistent_wide with one or two centre
    passes, its waves looping over the 256-column blocks; one batch-row buffer where two do not fit) -- the real SlowFast
    widths 1408 / 2304, d = 2048 below K = 1024, ragged widths (d % 256 != 0, d % 32 != 0), ragged centre and row groups.
    Two epochs (warm-up inside the first launch) == the oracle bit for bit in ONE launch per epoch, the lr fallback too."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    x = _mixture(d + K, n, d, K)
    acav.manual_seed(23)
    km = KMeans(None, d, K).to("cuda:0")
    ref = O.KMeans(d, K, O.Rng(23))
    xt = torch.from_numpy(x).cuda()
    for e in range(2):
        km.train_epoch(xt, b, lr=0.01)
        ref.train_epoch(x, b, lr=0.01)
        assert np.array_equal(km.centers.numpy(), ref.centers), f"epoch {e}"
        assert np.array_equal(km.counts.numpy(), ref.counts)
    assert km.count == ref.count
    launches, gave_up = km.train_stats()
    assert launches == 2 and gave_up == 0, f"persistent launches {launches}, fallbacks {gave_up}: the tall kernel was not taken (or gave up)"
    lab, _ = km.calc_best(xt)
    assert np.array_equal(lab.cpu().numpy(), ref.calc_best(x)[0])
    km2 = KMeans(None, d, K).to("cuda:0")
    ref2 = O.KMeans(d, K, O.Rng(13))
    km2.centers, km2.counts, km2.count = ref2.centers, ref2.counts, 0
    acav.manual_seed(14)
    ref2.rng = O.Rng(14)
    km2.train_epoch(xt, b, lr=0.3)
    ref2.train_epoch(x, b, lr=0.3)
    assert km2.fallback == ref2.fallback and (km2.fallback > 0 or K > 64)
    assert np.array_equal(km2.centers.numpy(), ref2.centers)


@pytest.mark.parametrize("n,d,K,b", [(3072, 1024, 1024, 32), (2048, 1000, 520, 32), (1536, 800, 300, 20), (1024, 1024, 1024, 7),
                                     (1280, 900, 264, 9)])
def test_two_row_pass_wide_kernel(env, n, d, K, b):
    """Round 4 (late): the wide persistent kernel with TWO row passes (16 centres x 16 rows per workgroup, one row buffer, the four
    quadrants of a column block as four interleaved chains) -- K = 1024 at 768 < d <= 1024 on 128 workgroups instead of 256, so that
    cfg5's two views train side by side.  Forced here for ONE clustering (ACAV_WIDE_NRP=2): two epochs == the oracle bit for bit in
    one launch per epoch; ragged widths, ragged centre groups (K = 520 / 300 / 264), ragged row groups (b = 20: 16 + 4, b = 9: 8 + 1
    valid row in the second pass, b = 7: none), the lr fallback."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    os.environ["ACAV_WIDE_NRP"] = "2"
    try:
        x = _mixture(d + K, n, d, K)
        acav.manual_seed(29)
        km = KMeans(None, d, K).to("cuda:0")
        ref = O.KMeans(d, K, O.Rng(29))
        xt = torch.from_numpy(x).cuda()
        for e in range(2):
            km.train_epoch(xt, b, lr=0.01)
            ref.train_epoch(x, b, lr=0.01)
            assert np.array_equal(km.centers.numpy(), ref.centers), f"epoch {e}"
            assert np.array_equal(km.counts.numpy(), ref.counts)
        assert km.count == ref.count
        launches, gave_up = km.train_stats()
        assert launches == 2 and gave_up == 0, f"persistent launches {launches}, fallbacks {gave_up}"
        km2 = KMeans(None, d, K).to("cuda:0")
        ref2 = O.KMeans(d, K, O.Rng(13))
        km2.centers, km2.counts, km2.count = ref2.centers, ref2.counts, 0
        acav.manual_seed(14)
        ref2.rng = O.Rng(14)
        km2.train_epoch(xt, b, lr=0.3)
        ref2.train_epoch(x, b, lr=0.3)
        assert km2.fallback == ref2.fallback
        assert np.array_equal(km2.centers.numpy(), ref2.centers)
    finally:
        os.environ.pop("ACAV_WIDE_NRP", None)


def test_two_wide_clusterings_side_by_side(env):
    """cfg5's pair: two K = 1024 clusterings of 1024-d rows in one acav_kmeans_train_multi call take the two-row-pass form (128
    workgroups each) and are in flight together; each == its own oracle, one persistent launch per epoch and handle."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    n, d, K, b = 4096, 1024, 1024, 32
    xs = [_mixture(70 + i, n, d, K) for i in range(2)]
    xts = [torch.from_numpy(x).cuda() for x in xs]
    acav.manual_seed(31)
    kms = [KMeans(None, d, K).to("cuda:0") for _ in range(2)]
    refs = [O.KMeans(d, K, O.Rng(0), centers=km._centers0.copy()) for km in kms]
    lab_rs = np.random.RandomState(3)
    for epoch in range(2):
        warm = [lab_rs.randint(0, K, (km.warmup_steps(b, n // b), b)).astype(np.int64) for km in kms]
        KMeans.train_epoch_multi(kms, xts, b, lr=0.01, warm_bests=warm)
        for km, ref, x, w in zip(kms, refs, xs, warm):
            for t in range(n // b):
                xb = x[t * b:(t + 1) * b]
                if t < len(w):
                    ref.apply_update(xb, w[t], 0.01)
                else:
                    ref.lr = 0.01
                    ref.add(xb)
            assert np.array_equal(km.centers.numpy(), ref.centers), f"epoch {epoch}"
            assert np.array_equal(km.counts.numpy(), ref.counts) and km.count == ref.count
    assert [km.train_stats() for km in kms] == [(2, 0), (2, 0)]


def test_cfg4_pair_in_one_multi_call(env):
    """cfg4's pair in one acav_kmeans_train_multi call: the column-split kernel of the 2048-d view (one workgroup on every CU)
    and the 128-d view's K = 1024 clustering, which has to wait its turn (the split kernel's waves take a SIMD's whole register file:
    nothing becomes resident beside it -- tools/exp/NOTES_r04.md section 12).  Each == its own oracle, one persistent launch per epoch
    and handle, none given up."""
    torch, acav, O = env
    from acav100m_amd.clustering import KMeans
    n, K, b = 3072, 1024, 32
    dims = [2048, 128]
    xs = [_mixture(80 + i, n, d, K) for i, d in enumerate(dims)]
    xts = [torch.from_numpy(x).cuda() for x in xs]
    acav.manual_seed(37)
    kms = [KMeans(None, d, K).to("cuda:0") for d in dims]
    refs = [O.KMeans(d, K, O.Rng(0), centers=km._centers0.copy()) for d, km in zip(dims, kms)]
    lab_rs = np.random.RandomState(5)
    for epoch in range(2):
        warm = [lab_rs.randint(0, K, (km.warmup_steps(b, n // b), b)).astype(np.int64) for km in kms]
        KMeans.train_epoch_multi(kms, xts, b, lr=0.01, warm_bests=warm)
        for km, ref, x, w in zip(kms, refs, xs, warm):
            for t in range(n // b):
                xb = x[t * b:(t + 1) * b]
                if t < len(w):
                    ref.apply_update(xb, w[t], 0.01)
                else:
                    ref.lr = 0.01
                    ref.add(xb)
            assert np.array_equal(km.centers.numpy(), ref.centers), f"epoch {epoch}"
            assert np.array_equal(km.counts.numpy(), ref.counts) and km.count == ref.count
    assert [km.train_stats() for km in kms] == [(2, 0), (2, 0)]
