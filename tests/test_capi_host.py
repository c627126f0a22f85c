"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol
include/acav_hip.h declares, the host MT19937 stream equals torch's (golden), the mirror classes
keep the reference's surface, and nothing in the product imports the oracle."""
import ast
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def acav():
    import acav100m_amd
    acav100m_amd.load_library()
    return acav100m_amd


def test_header_symbols_exported(acav):
    from acav100m_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "acav_hip.h")).read()
    declared = set(re.findall(r"\b(acav_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    lib = _lib.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/acav_hip.h but not exported"
    assert declared - {"acav_last_error"} == set(_lib.SIGNATURES), "ctypes table and header drifted"
    assert lib.acav_version() >= 100


def test_trim_device_cache_is_callable_without_a_gpu(acav):
    import acav100m_amd
    assert acav100m_amd.trim_device_cache() >= 0  # nothing parked in a process that never allocated


def test_no_gpu_is_a_loud_error(acav):
    from acav100m_amd.clustering import KMeans
    if acav.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(acav.AcavError, match="no HIP device"):
        KMeans(None, 8, 4).to("cuda")
    from acav100m_amd.subset_selection import get_measure
    m = get_measure("batch_mi")(np.zeros((10, 2), np.int64), ncentroids=1, batch_size=2, selection_size=1,
                                device="cuda")
    with pytest.raises(acav.AcavError, match="no HIP device"):
        m.init([(0, 1)], list(range(1, 10)))


def test_product_rng_equals_torch_stream(acav, golden_dir):
    g = np.load(os.path.join(golden_dir, "rng.npz"))
    for s in (0, 1, 1234):
        r = acav.Generator(s)
        assert np.array_equal(r.rand(7, 5), g[f"s{s}_rand_7x5"])
        assert np.array_equal(r.randperm(10), g[f"s{s}_perm10"])
        assert np.array_equal(r.randperm(1000), g[f"s{s}_perm1000"])
        assert np.array_equal(r.randperm(100003)[:2000], g[f"s{s}_perm100003_head"])
        assert np.array_equal(r.rand(3), g[f"s{s}_rand_after"])


def test_product_rng_live_against_torch(acav):
    import torch
    torch.manual_seed(77)
    r = acav.Generator(77)
    assert np.array_equal(r.rand(33, 3), torch.rand(33, 3).numpy())
    assert np.array_equal(r.randperm(5001), torch.randperm(5001).numpy())
    # default generator == an unseeded torch process (default seed 67280421310721)
    assert torch.initial_seed() == 77
    g2 = acav.Generator()
    torch.manual_seed(67280421310721)
    assert np.array_equal(g2.rand(5), torch.rand(5).numpy())


def test_warmup_best_matches_reference_formula(acav):
    import torch
    torch.manual_seed(5)
    dist = torch.rand(24, 32)
    v, i = dist.min(axis=0)
    best, mean = acav.Generator(5).warmup_best(24, 32)
    assert np.array_equal(best, i.numpy())
    assert abs(mean - v.mean().item()) < 1e-6


def test_kmeans_surface_and_checkpoint_roundtrip(acav, golden_dir):
    """constructor / attributes / get_attrs / load of sgd_clustering.py:18-57 without a GPU"""
    import pickle
    from acav100m_amd.clustering import KMeans
    import torch
    acav.manual_seed(0)
    km = KMeans(None, 8, 16)
    torch.manual_seed(0)
    assert np.array_equal(km.centers.numpy(), (torch.rand(16, 8) * 1e-5).numpy())  # sgd_clustering.py:24
    assert km.count == 0 and km.fallback == 0 and km.lr == 1e-2 and km.initial_rounds == 10
    assert km.reinit == (.7, 5.0) and km.sequential is False and not km.is_distributed
    km.count = 320
    km.counts = np.arange(16, dtype=np.float32)
    dt = km.get_attrs()
    assert set(dt) == {'args', 'count', 'lr', 'initial_rounds', 'reinit', 'fallback', 'sequential', 'centers',
                       'counts'}
    km2 = KMeans.load(dt)
    assert km2.count == 320 and np.array_equal(km2.counts.numpy(), np.arange(16, dtype=np.float32))
    km3 = pickle.loads(pickle.dumps(km))
    assert np.array_equal(km3.centers.numpy(), km.centers.numpy()) and km3.count == 320
    for name in ("to", "initialize", "add", "calc_best", "get_attrs", "load_from_saves", "load"):
        assert callable(getattr(KMeans, name))


def test_run_greedy_host_logic(acav, monkeypatch):
    """_run_greedy (run_greedy.py:9-54): C = max+1, subset = round(ratio*V), B/k clamps, python shuffle,
    start index removed from the candidates -- checked with a recording stand-in measure."""
    import random
    import importlib
    rg = importlib.import_module("acav100m_amd.subset_selection.run_greedy")
    seen = {}

    class Rec:
        def __init__(self, assignments, **kw):
            seen.update(kw, V=assignments.shape[0])

        def init(self, pairs, candidates):
            seen.update(pairs=pairs, candidates=list(candidates))

        def run_greedy(self, subset_size, start_indices, target, **kw):
            seen.update(subset=subset_size, start=list(start_indices))
            return list(range(subset_size)), [0.0] * subset_size, [], []

    monkeypatch.setattr(rg, "get_measure", lambda name: Rec)

    class NS:
        def __init__(self, **kw):
            self.__dict__.update(kw)
    args = NS(batch=NS(batch_size=20, selection_size=4, keep_unselected=True), computation=NS(device="cuda"),
              log_every=1, log_times=None, node_rank=None, parent_pid=None)
    a = np.zeros((15, 3), np.int64)
    a[4, 1] = 6
    random.seed(3)
    rg._run_greedy(args, a, [("a", "l0"), ("a", "l1"), ("v", "l0")], None, 0.2, "batch_mi", "combination", True)
    random.seed(3)
    expect = list(range(15))
    random.shuffle(expect)
    assert seen["ncentroids"] == 7 and seen["subset"] == 3 and seen["batch_size"] == 14
    assert seen["selection_size"] == 4 and seen["start"] == expect[:1] and seen["candidates"] == expect[1:]
    assert seen["pairs"] == [(0, 1), (0, 2), (1, 2)]
    rows = rg.run_greedy(args, a, ["s"] * 15, [f"f{i}" for i in range(15)], [("a", "l0"), ("a", "l1"), ("v", "l0")],
                         None, 0.2, "batch_mi")
    assert [r["filename"] for r in rows] == ["f0", "f1", "f2"]  # sorted(S) (run_greedy.py:72)


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "acav100m_amd")):
        for f in files:
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(dirpath, f)).read())
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or ""]
                bad += [(f, n) for n in names if n.split(".")[0] == "oracle"]
    assert not bad, bad
    # the native side neither includes nor loads anything from oracle/ (comments may cite it as the spec)
    for f in os.listdir(os.path.join(ROOT, "acav100m_amd", "csrc")):
        src = open(os.path.join(ROOT, "acav100m_amd", "csrc", f)).read()
        assert "libacav_oracle" not in src and "orc_" not in src, f
        assert not re.search(r'#include\s*[<"][^>"]*oracle', src), f


def test_rng_jump_equals_sequential_draws(acav):
    """GF(2) jump-ahead of the MT19937 stream (acav_mtjump.hip: characteristic polynomial by Berlekamp-Massey,
    t^J mod phi, XOR-combination of windows) == drawing the words one by one.  The GPU generator lanes hop over each
    other's blocks with the same polynomials."""
    for seed, pre, n in [(0, 0, 1), (0, 0, 624), (0, 0, 625), (1, 100, 5000), (2, 7, 300_007), (3, 624, 19937 * 3 + 11)]:
        a, b = acav.Generator(seed), acav.Generator(seed)
        for _ in range(pre):
            a.u32(), b.u32()
        a.jump(n)
        for _ in range(n):
            b.u32()
        (ma, ia), (mb, ib) = a.get_state(), b.get_state()
        assert ia == ib and np.array_equal(ma, mb), (seed, pre, n)
        assert [a.u32() for _ in range(700)] == [b.u32() for _ in range(700)]
    # composition: two half jumps == one jump, far beyond anything one would draw sequentially
    a, b = acav.Generator(9), acav.Generator(9)
    n = 624 * 4096 * 31 + 12345
    a.jump(n)
    b.jump(n // 3).jump(n - n // 3)
    assert a.get_state()[1] == b.get_state()[1] and np.array_equal(a.get_state()[0], b.get_state()[0])


def test_python_shuffle_in_library_equals_random_shuffle(acav):
    """run_greedy.py:38-41 shuffles the candidates with Python's generator; acav_rng_py_shuffle reproduces
    random.shuffle(list(range(n))) and leaves `random` in the state the interpreter's own loop would."""
    import random
    from acav100m_amd.rng import python_shuffled_range
    for seed, n in [(0, 1), (0, 2), (0, 10), (1, 1000), (5, 100003), (7, 65536)]:
        random.seed(seed)
        want = list(range(n))
        random.shuffle(want)
        after = [random.random(), random.getrandbits(40)]
        random.seed(seed)
        got = python_shuffled_range(n)
        assert got.tolist() == want, (seed, n)
        assert [random.random(), random.getrandbits(40)] == after, "generator state after the shuffle"


def test_contrastive_rank_rows_short_batches(acav):
    """a batch with fewer rows than ranks (the tail batch of drop_last=False) must not raise on a SUBSET of the
    ranks in front of a collective.  Training drops it on every rank alike (decided from the offsets), inference scores an
    empty slice on the ranks past its end."""
    from acav100m_amd.subset_selection.measures.contrastive import Contrastive
    off = [0, 128, 129]  # a 128-row batch and a 1-row tail, 2 ranks
    per_rank = [Contrastive.rank_rows(off, r, 2) for r in range(2)]
    assert [o.tolist() for _, o in per_rank] == [[0, 64], [0, 64]]          # same number of batches on both ranks
    assert per_rank[0][0].tolist() == list(range(0, 128, 2)) and per_rank[1][0].tolist() == list(range(1, 128, 2))
    inf = [Contrastive.rank_rows(off, r, 2, training=False) for r in range(2)]
    assert inf[0][0].tolist() == list(range(0, 128, 2)) + [128] and inf[0][1].tolist() == [0, 64, 65]
    assert inf[1][0].tolist() == list(range(1, 128, 2)) and inf[1][1].tolist() == [0, 64, 64]
    assert sorted(inf[0][0].tolist() + inf[1][0].tolist()) == list(range(129))  # every row scored exactly once
