"""GPU parity of the contrastive baseline measure: HIP (through the C ABI / Contrastive mirror) vs the
golden vectors produced by the reference module itself and vs the numpy oracle.  fp32 with a different GEMM summation
order than the reference's MKL: 1e-4 relative on trained parameters and scores, 5e-5 on per-batch losses (stated per
assertion); the seeded initial parameters are bit-exact."""
import csv
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
    return torch, acav100m_amd


def _data(seed, n, vis, aud):
    rs = np.random.RandomState(seed)
    comp = rs.randint(0, 12, n)
    cv, ca = rs.randn(12, vis).astype(np.float32), rs.randn(12, aud).astype(np.float32)
    return (cv[comp] + 0.5 * rs.randn(n, vis)).astype(np.float32), (ca[comp] + 0.5 * rs.randn(n, aud)).astype(np.float32)


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_contrastive_vs_reference_and_oracle(env, golden_dir, name):
    torch, acav = env
    from acav100m_amd.rng import Generator
    from acav100m_amd.subset_selection.measures import get_measure
    from acav100m_amd.subset_selection.measures.contrastive import PARAM_NAMES, lr_func_linear
    from oracle import contrastive_ref as CR
    g = np.load(os.path.join(golden_dir, f"contrastive_{name}.npz"))
    vis, aud, B, nb, epochs = int(g["vis"]), int(g["aud"]), int(g["B"]), int(g["nb"]), int(g["epochs"])
    out = int(g["out"]) if "out" in g.files else None
    m = get_measure("contrastive")(epochs, "cuda:0", float(g["base_lr"]), int(g["warm"]), sizes=(vis, aud), out_size=out,
                                   generator=Generator(int(g["seed"])))
    sd0 = m.state_dict()
    for k in PARAM_NAMES:
        ref = g["p0_" + k]
        assert np.array_equal(sd0[k][:len(ref)], ref), f"{k}: seeded init differs from torch's nn.Linear"
    if name == "c":
        visual, audio = _data(int(g["data_seed"]), B * nb, vis, aud)
    else:
        visual, audio = g["visual"], g["audio"]
    orc = CR.Contrastive(*[sd0[k] for k in PARAM_NAMES])
    offsets = np.arange(nb + 1, dtype=np.int64) * B
    losses, accs, olosses = [], [], []
    for epoch in range(epochs):
        lr = lr_func_linear(epoch + 1, epochs + 1, int(g["warm"])) * float(g["base_lr"])
        lo, ac = m.train_batches(visual, audio, offsets, lr)
        losses.extend(lo.tolist())
        accs.extend(ac.tolist())
        for bi in range(nb):
            olosses.append(orc.train_batch(visual[bi * B:(bi + 1) * B], audio[bi * B:(bi + 1) * B], lr)[0])
    np.testing.assert_allclose(losses, g["losses"], rtol=5e-5)   # vs the reference
    np.testing.assert_allclose(losses, olosses, rtol=5e-5)       # vs the oracle
    np.testing.assert_allclose(accs, g["accs"], atol=1e-3)
    sd1 = m.state_dict()
    for k, op in zip(PARAM_NAMES, orc.p):
        ref = g["p1_" + k]
        np.testing.assert_allclose(sd1[k][:len(ref)], ref, rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(sd1[k], op, rtol=1e-4, atol=2e-6)
    scores = m.infer_scores(visual, audio)
    np.testing.assert_allclose(scores, g["infer"], rtol=1e-4, atol=2e-6)
    # device-resident inputs and ragged batches (the de-duplicated batches of the real loader) take the same path
    m2 = get_measure("contrastive")(1, "cuda:0", 1e-3, 1, sizes=(vis, aud), out_size=out, generator=Generator(1))
    o2 = CR.Contrastive(*[m2.state_dict()[k] for k in PARAM_NAMES])
    off = np.array([0, B - 3, 2 * B - 3, 2 * B], np.int64)
    lo, _ = m2.train_batches(torch.from_numpy(visual[:2 * B]).cuda(), torch.from_numpy(audio[:2 * B]).cuda(), off, 1e-3)
    want = [o2.train_batch(visual[off[i]:off[i + 1]], audio[off[i]:off[i + 1]], 1e-3)[0] for i in range(3)]
    np.testing.assert_allclose(lo, want, rtol=5e-5, atol=1e-6)  # the third loss is ~2e-3: a difference of logsumexp and logit


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_contrastive_random_shapes_vs_oracle(env, seed):
    """Random feature widths (not multiples of the 64 x 64 GEMM tile), projection sizes and ragged batch lists, a few
    accumulating steps each: losses 1e-4 relative (+1e-6), parameters 2e-4 relative, scores (cosines) 2e-4 relative + 1e-5
    vs the numpy oracle."""
    torch, acav = env
    from acav100m_amd.rng import Generator
    from acav100m_amd.subset_selection.measures import get_measure
    from acav100m_amd.subset_selection.measures.contrastive import PARAM_NAMES
    from oracle import contrastive_ref as CR
    rs = np.random.RandomState(100 + seed)
    vis, aud = int(rs.choice([5, 37, 64, 130, 513, 1000])), int(rs.choice([3, 64, 100, 257]))
    out = int(rs.choice([2, 19, 64, 128, 200]))
    sizes = [int(rs.choice([1, 2, 7, 33, 64, 65, 150])) for _ in range(int(rs.randint(2, 6)))]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    visual, audio = _data(seed, int(off[-1]), vis, aud)
    m = get_measure("contrastive")(2, "cuda:0", 1e-3, 1, sizes=(vis, aud), out_size=out, generator=Generator(seed))
    orc = CR.Contrastive(*[m.state_dict()[k] for k in PARAM_NAMES])
    for lr in (5e-4, 1e-3):
        lo, ac = m.train_batches(visual, audio, off, lr)
        want = [orc.train_batch(visual[off[i]:off[i + 1]], audio[off[i]:off[i + 1]], lr) for i in range(len(sizes))]
        np.testing.assert_allclose(lo, [w[0] for w in want], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(ac, [w[1] for w in want], atol=1e-3)
    sd = m.state_dict()
    for k, op in zip(PARAM_NAMES, orc.p):
        np.testing.assert_allclose(sd[k], op, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(m.infer_scores(visual, audio), orc.infer(visual, audio), rtol=2e-4, atol=1e-5)  # cosines in [-1, 1]


def test_contrastive_cli_end_to_end(env, tmp_path_factory, golden_dir):
    """`cli.py run --measure_name=contrastive` + `cli.py merge_contrastive` on synthetic feature shards (the real
    2304-d / 128-d penultimate layers): model caches per epoch in the reference's torch.save layout, one inference cache
    line per clip, output.csv = every clip once, best score first; scores equal the oracle's run of the same schedule."""
    torch, acav = env
    sys.path.insert(0, golden_dir)
    import synth
    from acav100m_amd.subset_selection.cli import Cli
    from acav100m_amd.subset_selection.measures.contrastive import PARAM_NAMES, lr_func_linear
    from acav100m_amd.subset_selection.run_contrastive import feature_batches
    from acav100m_amd import shards as io
    from oracle import contrastive_ref as CR
    from oracle import oracle as O
    root = str(tmp_path_factory.mktemp("acav_ctr"))
    glob = synth.write_feature_shards(root, n_shards=3, rows=200, seed=5)
    out_csv = os.path.join(root, "sel", "output.csv")
    acav.manual_seed(0)
    kw = dict(shards_path=glob, meta_path=os.path.join(root, "videos"), out_path=out_csv, measure_name="contrastive")
    Cli().run(**kw)
    caches = sorted(os.listdir(os.path.join(root, "sel", "caches")))
    assert sum(c.startswith("contrastive_model_cache_epoch_") and c.endswith(".pkl") for c in caches) == 3
    ck = torch.load(os.path.join(root, "sel", "caches", [c for c in caches if c.startswith("contrastive_model_cache_epoch_2") and c.endswith(".pkl")][0]),
                    weights_only=False)
    assert sorted(ck.keys()) == ["base_lr", "epoch", "model"] and sorted(ck["model"].keys()) == sorted(PARAM_NAMES)
    assert tuple(ck["model"]["visual_linear.weight"].shape) == (128, 2304)
    inf = [c for c in caches if "contrastive_inferred_cache" in c]
    assert len(inf) == 1
    rows = list(csv.reader(open(os.path.join(root, "sel", "caches", inf[0]))))
    assert len(rows) == 600 and len(rows[0]) == 5
    # the oracle, same seeded stream, same batch stream and schedule, the reference's object flow (run_contrastive.py:16-116):
    # the first Contrastive of _run only consumes its draws, _train builds and trains the second, _infer builds a third and
    # copies the trained WEIGHTS into it -- the scores come from the trained weights with the third object's fresh biases
    rng = O.Rng(0)

    def draw():
        wv, bv = CR.linear_init(lambda n: rng.rand(n), 128, 2304)
        wa, ba = CR.linear_init(lambda n: rng.rand(n), 128, 128)
        return wv, bv, wa, ba
    draw()
    orc = CR.Contrastive(*draw())
    table = io.load_feature_shards(sorted(io.brace_expand(glob)))
    visual, audio, off, meta_rows = feature_batches(table, 128)
    for epoch in range(3):
        lr = lr_func_linear(epoch + 1, 4, 1) * 2e-4
        for i in range(len(off) - 1):
            orc.train_batch(visual[off[i]:off[i + 1]], audio[off[i]:off[i + 1]], lr)
    _, bv3, _, ba3 = draw()
    want = CR.Contrastive(orc.p[0], bv3, orc.p[2], ba3).infer(visual, audio)
    got = np.array([float(r[0]) for r in rows], np.float32)
    assert [r[2] for r in rows] == [m["filename"] for m in meta_rows]
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5)
    Cli().merge_contrastive(**kw)
    final = list(csv.reader(open(out_csv)))
    assert len(final) == 600 and len({r[1] for r in final}) == 600 and len(final[0]) == 4
    order = np.argsort(-got, kind="stable")
    assert [r[1] for r in final[:50]] == [rows[i][2] for i in order[:50]]


def test_contrastive_chunked_cli(env, tmp_path_factory, golden_dir):
    """`cli.py run --measure_name=contrastive --chunk_size=2`: every chunk of shards trains its own model and scores its own
    clips (chunk_contrastive.py:17-49,115-131); the rank's inference cache collects all chunks, merge_contrastive ranks them.
    The first chunk's scores equal a plain run on those two shards with the same seed."""
    torch, acav = env
    sys.path.insert(0, golden_dir)
    import synth
    from acav100m_amd.subset_selection.cli import Cli
    root = str(tmp_path_factory.mktemp("acav_ctr_chunks"))
    glob = synth.write_feature_shards(root, n_shards=4, rows=150, seed=9)
    out_csv = os.path.join(root, "sel", "output.csv")
    kw = dict(shards_path=glob, meta_path=os.path.join(root, "videos"), out_path=out_csv, measure_name="contrastive")
    acav.manual_seed(3)
    done = Cli().run(chunk_size=2, **kw)
    assert done == [0, 1]
    caches = sorted(os.listdir(os.path.join(root, "sel", "caches")))
    models = [c for c in caches if c.startswith("contrastive_model_cache_epoch_") and c.endswith(".pkl")]
    assert len(models) == 6 and {c[:-4].rsplit("_", 1)[1] for c in models} == {"0", "1"}  # 3 epochs x 2 chunks, chunk_num last
    inf = [c for c in caches if "contrastive_inferred_cache" in c]
    assert len(inf) == 1
    rows = list(csv.reader(open(os.path.join(root, "sel", "caches", inf[0]))))
    assert len(rows) == 600 and len({r[2] for r in rows}) == 600
    Cli().merge_contrastive(**kw)
    final = list(csv.reader(open(out_csv)))
    assert len(final) == 600 and len({r[1] for r in final}) == 600
    got = np.array([float(r[0]) for r in rows], np.float32)
    order = np.argsort(-got, kind="stable")
    assert [r[1] for r in final[:50]] == [rows[i][2] for i in order[:50]]
    # chunk 0 alone, same seed: the same 300 scores
    assert "{000000..000003}" in glob
    root2 = str(tmp_path_factory.mktemp("acav_ctr_chunk0"))
    out2 = os.path.join(root2, "sel", "output.csv")
    acav.manual_seed(3)
    Cli().run(shards_path=glob.replace("{000000..000003}", "{000000..000001}"), meta_path=os.path.join(root, "videos"), out_path=out2,
              measure_name="contrastive")
    c2 = os.path.join(root2, "sel", "caches")
    rows2 = list(csv.reader(open(os.path.join(c2, [c for c in os.listdir(c2) if "contrastive_inferred_cache" in c][0]))))
    assert [r[2] for r in rows2] == [r[2] for r in rows[:300]]
    np.testing.assert_array_equal(np.array([float(r[0]) for r in rows2]), got[:300].astype(np.float64))


def test_contrastive_distributed_two_workers(env, tmp_path_factory, golden_dir):
    """`cli.py run --measure_name=contrastive` with computation.use_distributed (the reference's default) on TWO workers
    (both on cuda:0, gloo carries the gradient average): every worker trains on rows rank::2 of every batch, the averaged
    gradients drive identical AdamW steps (run_contrastive.py:118-168, contrastive.py:92-124, module.py:96-101); each
    worker scores its rows and writes its own inference cache.  Against the oracle's replica simulation."""
    import subprocess
    torch, acav = env
    sys.path.insert(0, golden_dir)
    import synth
    from acav100m_amd.subset_selection.cli import Cli
    from acav100m_amd.subset_selection.measures.contrastive import PARAM_NAMES, lr_func_linear
    from acav100m_amd.subset_selection.run_contrastive import feature_batches
    from acav100m_amd import shards as io
    from oracle import contrastive_ref as CR
    from oracle import oracle as O
    root = str(tmp_path_factory.mktemp("acav_ctr_ddp"))
    glob = synth.write_feature_shards(root, n_shards=3, rows=200, seed=5)
    out_csv = os.path.join(root, "sel", "output.csv")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    envp = dict(os.environ, ACAV_OVERSUBSCRIBE="1", ACAV_DIST_BACKEND="gloo", ACAV_SEED="0", PYTHONPATH=repo)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        envp.pop(k, None)
    cmd = [sys.executable, "-m", "acav100m_amd.subset_selection.cli", "run", "--shards_path=" + glob,
           "--meta_path=" + os.path.join(root, "videos"), "--out_path=" + out_csv, "--measure_name=contrastive",
           "--computation.num_gpus=2"]
    res = subprocess.run(cmd, env=envp, cwd=repo, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "training contrastive loss with distributed" in res.stdout and "(node 1) running inference" in res.stdout
    cdir = os.path.join(root, "sel", "caches")
    caches = sorted(os.listdir(cdir))
    trained = [c for c in caches if c.startswith("contrastive_trained_model_cache_")]
    assert len(trained) == 1
    ck = torch.load(os.path.join(cdir, trained[0]), weights_only=False)
    assert sorted(ck.keys()) == ["base_lr", "model"] and sorted(ck["model"].keys()) == sorted(PARAM_NAMES)
    inf = sorted(c for c in caches if "contrastive_inferred_cache" in c)
    assert len(inf) == 2 and inf[0].endswith("_0.csv") and inf[1].endswith("_1.csv")
    rows = [list(csv.reader(open(os.path.join(cdir, f)))) for f in inf]
    # the oracle: same seeded stream and object flow as the single-process run, two replicas per batch
    rng = O.Rng(0)

    def draw():
        wv, bv = CR.linear_init(lambda n: rng.rand(n), 128, 2304)
        wa, ba = CR.linear_init(lambda n: rng.rand(n), 128, 128)
        return wv, bv, wa, ba
    draw()
    orc = CR.Contrastive(*draw())
    table = io.load_feature_shards(sorted(io.brace_expand(glob)))
    visual, audio, off, meta_rows = feature_batches(table, 128)
    for epoch in range(3):
        lr = lr_func_linear(epoch + 1, 4, 1) * 2e-4
        for i in range(len(off) - 1):
            orc.train_batch_ddp(visual[off[i]:off[i + 1]], audio[off[i]:off[i + 1]], lr, 2)
    for k, op in zip(PARAM_NAMES, orc.p):
        np.testing.assert_allclose(ck["model"][k].numpy(), op, rtol=3e-4, atol=3e-6)
    _, bv3, _, ba3 = draw()
    want = CR.Contrastive(orc.p[0], bv3, orc.p[2], ba3).infer(visual, audio)
    for rank in (0, 1):
        idx = np.concatenate([np.arange(off[i] + rank, off[i + 1], 2) for i in range(len(off) - 1)])
        assert [r[2] for r in rows[rank]] == [meta_rows[i]["filename"] for i in idx]
        np.testing.assert_allclose(np.array([float(r[0]) for r in rows[rank]], np.float32), want[idx], rtol=3e-4, atol=3e-5)
    kw = dict(shards_path=glob, meta_path=os.path.join(root, "videos"), out_path=out_csv, measure_name="contrastive")
    Cli().merge_contrastive(**kw)
    final = list(csv.reader(open(out_csv)))
    assert len(final) == 600 and len({r[1] for r in final}) == 600


def test_contrastive_pieces_and_world1_comm(env):
    """backward / get_grads / set_grads / step are the pieces of train_batches (bit for bit), and a world-1 RCCL communicator
    set on the handle (the gradient average runs through acav_comm) changes nothing"""
    torch, acav = env
    from acav100m_amd.subset_selection.measures.contrastive import Contrastive
    from acav100m_amd.parallel.rccl_comm import Comm
    rs = np.random.RandomState(7)
    n, vis, aud = 96, 80, 48
    visual, audio = rs.randn(n, vis).astype(np.float32), rs.randn(n, aud).astype(np.float32)
    off = np.array([0, 32, 64, 96], np.int64)
    models = []
    for mode in range(3):
        acav.manual_seed(4)
        m = Contrastive(2, "cuda", 1e-3, 1, sizes=(vis, aud), out_size=24)
        if mode == 0:
            lo, ac = m.train_batches(visual, audio, off, 1e-3)
        elif mode == 1:
            lo, ac = [], []
            for i in range(3):
                l, a = m.backward(visual[off[i]:off[i + 1]], audio[off[i]:off[i + 1]])
                g = m.get_grads()
                m.set_grads(g)  # the host round trip of the torch.distributed route
                m.step(1e-3)
                lo.append(l), ac.append(a)
        else:
            comm = Comm(0, 1, Comm.unique_id(), 0)
            m.set_comm(comm)
            lo, ac = m.train_batches(visual, audio, off, 1e-3)
            m.set_comm(None)
        models.append((np.asarray(lo, np.float32), np.asarray(ac, np.float32), m.state_dict(), m.get_grads()))
    for lo, ac, sd, g in models[1:]:
        assert np.array_equal(lo, models[0][0]) and np.array_equal(ac, models[0][1]) and np.array_equal(g, models[0][3])
        for k in sd:
            assert np.array_equal(sd[k], models[0][2][k])
    idx, o2 = Contrastive.rank_rows(off, 1, 3)
    assert idx.tolist() == [i for b in range(3) for i in range(32 * b + 1, 32 * (b + 1), 3)] and o2.tolist() == [0, 11, 22, 33]
