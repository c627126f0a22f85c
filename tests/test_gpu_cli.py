"""CLI-level parity (G5): our `cli.py cluster` / `cli.py run` against the files the reference's own
CLIs produced on the same synthetic shards (regenerated from the seed by tests/golden/synth.py)."""
import csv
import io as _io
import itertools
import json
import os
import pickle
import random
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workdir(tmp_path_factory, golden_dir):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, golden_dir)
    import synth
    root = str(tmp_path_factory.mktemp("acav_cli"))
    glob = synth.write_feature_shards(root, n_shards=4, rows=256, seed=0)
    return root, glob


def test_cluster_cli_matches_reference_files(workdir, golden_dir):
    import acav100m_amd
    from acav100m_amd.clustering.cli import Cli
    root, glob = workdir
    g = np.load(os.path.join(golden_dir, "cli_clustering.npz"))
    acav100m_amd.manual_seed(0)
    saved = Cli().cluster(feature_path=glob, out_path=os.path.join(root, "clusters"),
                          meta_path=os.path.join(root, "videos"), **{"computation.num_workers": 0})  # as the golden was made
    assert [p.name for p in saved] == ["shard-%06d.pkl" % s for s in range(4)]
    for s in range(4):
        name = "shard-%06d" % s
        rows = pickle.load(open(os.path.join(root, "clusters", name + ".pkl"), "rb"))
        assert [r["filename"] for r in rows] == list(g[name + "_files"])
        lab = np.array([[int(r["audio_assignments"][0]["array"]["layer_%d" % i]) for i in range(5)] +
                        [int(r["video_assignments"][0]["array"]["layer_%d" % i]) for i in range(5)] for r in rows])
        assert np.array_equal(lab, g[name]), f"{name}: {(lab != g[name]).sum()} labels differ from the reference CLI"
    r0 = rows[0]
    assert sorted(r0.keys()) == list(g["row_keys"])
    assert sorted(r0["audio_assignments"][0].keys()) == list(g["audio_entry_keys"])
    assert r0["audio_assignments"][0]["model_key"] == str(g["audio_model_key"])
    assert r0["video_assignments"][0]["model_key"] == str(g["video_model_key"])
    assert type(r0["audio_assignments"][0]["array"]["layer_0"]).__name__ == str(g["label_type"])
    files = sorted(os.listdir(os.path.join(root, "clusters")))
    ref_files = list(g["out_files"])
    strip = lambda fs: sorted("log" if f.startswith("log_") else f for f in fs)  # noqa: E731  (log name = host_pid_ts)
    assert strip(files) == strip(ref_files), (files, ref_files)
    log = [f for f in files if f.startswith("log_")][0]
    assert sorted(json.load(open(os.path.join(root, "clusters", log))).keys()) == list(g["log_keys"])
    assert os.path.isfile(os.path.join(root, "videos", "meta_cache.pkl"))
    # a second run finds every shard already written
    assert Cli().cluster(feature_path=glob, out_path=os.path.join(root, "clusters"),
                         meta_path=os.path.join(root, "videos")) == []


@pytest.mark.parametrize("variant", ["nw3", "nw3_ragged"])
def test_cluster_cli_matches_reference_files_with_loader_workers(tmp_path_factory, golden_dir, variant):
    """The reference's DEFAULT loader has worker processes (computation.num_workers = 40, config.py:29): its DataLoader hands out
    whole batches round-robin over the workers, worker w streaming shards [w::num_workers] (data/clustering.py:17-66,212-228).
    The goldens are the reference CLI's own files with --computation.num_workers=3 on six shards: `nw3` equal worker streams
    (nothing wraps), `nw3_ragged` streams of 320 / 270 / 110 rows cut / cycled to get_length() = 320 samples each
    (ResizedDataset).  Our CLI with the same flag -- the batch order is a plan, the shards are read as ever -- must write the
    same assignment files, bit for bit; with ACAV_LOADER_ORDER=single it must NOT (the order matters)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, golden_dir)
    import synth
    import acav100m_amd
    from acav100m_amd.clustering.cli import Cli
    g = np.load(os.path.join(golden_dir, "cli_clustering_%s.npz" % variant))
    sizes = [int(x) for x in g["shard_rows"]]
    root = str(tmp_path_factory.mktemp("acav_cli_" + variant))
    glob = synth.write_feature_shards(root, n_shards=len(sizes), rows=sizes, seed=5)

    def labels_of(out):
        res = {}
        for s in range(len(sizes)):
            name = "shard-%06d" % s
            rows = pickle.load(open(os.path.join(out, name + ".pkl"), "rb"))
            assert [r["filename"] for r in rows] == list(g[name + "_files"])
            res[name] = np.array([[int(r["audio_assignments"][0]["array"]["layer_%d" % i]) for i in range(5)] +
                                  [int(r["video_assignments"][0]["array"]["layer_%d" % i]) for i in range(5)] for r in rows])
        return res

    acav100m_amd.manual_seed(0)
    out = os.path.join(root, "clusters")
    saved = Cli().cluster(feature_path=glob, out_path=out, meta_path=os.path.join(root, "videos"),
                          **{"computation.num_workers": int(g["num_workers"])})
    assert [p.name for p in saved] == ["shard-%06d.pkl" % s for s in range(len(sizes))]
    for name, lab in labels_of(out).items():
        assert np.array_equal(lab, g[name]), f"{variant} {name}: {(lab != g[name]).sum()} labels differ from the reference CLI"
    os.environ["ACAV_LOADER_ORDER"] = "single"
    try:
        acav100m_amd.manual_seed(0)
        out1 = os.path.join(root, "clusters_single")
        Cli().cluster(feature_path=glob, out_path=out1, meta_path=os.path.join(root, "videos"),
                      **{"computation.num_workers": int(g["num_workers"])})
        assert any(not np.array_equal(lab, g[name]) for name, lab in labels_of(out1).items())
    finally:
        os.environ.pop("ACAV_LOADER_ORDER", None)


def test_subset_cli_output_csv(workdir, golden_dir):
    """Same file format / size / ordering as the reference's output.csv; the selected set equals the
    oracle's (bit-exact); overlap with the reference's free-running selection is reported (its picks
    depend on fp32 summation noise and torch.topk's tie order -- DESIGN.md, parity)."""
    import acav100m_amd
    from acav100m_amd.subset_selection.cli import Cli
    from acav100m_amd import shards as io
    from oracle import oracle as O
    root, glob = workdir
    if not os.path.isfile(os.path.join(root, "clusters", "shard-000000.pkl")):
        pytest.skip("clustering test did not run")
    out_csv = os.path.join(root, "output.csv")
    random.seed(0)
    acav100m_amd.manual_seed(0)
    Cli().run(shards_path=os.path.join(root, "clusters", "shard-{000000..000003}.pkl"),
              meta_path=os.path.join(root, "videos"), out_path=out_csv)
    ours = open(out_csv).read().splitlines()
    ref = open(os.path.join(golden_dir, "cli_output.csv")).read().splitlines()
    assert len(ours) == len(ref) == 205
    parse = lambda lines: list(csv.reader(_io.StringIO("\n".join(lines))))  # noqa: E731
    po, pr = parse(ours), parse(ref)
    assert all(len(r) == 4 and r[3] == "[10, 20]" and r[2] == r[1][:12] for r in po)
    assert ours[0].count('"') == ref[0].count('"') == 2          # the list repr is quoted the same way
    assert [r[1] for r in po] == sorted(r[1] for r in po)         # sorted(S) order (run_greedy.py:72)
    # expected selection from the oracle on the same inputs / seeds
    paths = [os.path.join(root, "clusters", "shard-%06d.pkl" % s) for s in range(4)]
    a, types, shard_names, filenames = io.load_assignment_shards(paths)
    assert types == sorted(types) and len(types) == 10
    random.seed(0)
    cand = list(range(len(a)))
    random.shuffle(cand)
    pairs = list(itertools.combinations(range(10), 2))
    res = O.BatchMI(a, int(a.max()) + 1, pairs).run_greedy(cand[1:], cand[:1], 205, 20, 4, O.Rng(0))
    assert [r[1] for r in po] == [filenames[s] for s in sorted(res["S"])]
    overlap = len({r[1] for r in po} & {r[1] for r in pr}) / 205.0
    print(f"overlap with the reference's free-running selection: {overlap:.2f}")


def test_chunked_run_sync_and_async_prefetch(workdir):
    """chunk mode (chunk.py:21-53): per-chunk selections into caches/, merged by reduce_csvs; with
    computation.load_async the next chunk is loaded by a host thread while the GPU works -- same files."""
    import acav100m_amd
    from acav100m_amd.subset_selection.cli import Cli
    root, glob = workdir
    if not os.path.isfile(os.path.join(root, "clusters", "shard-000000.pkl")):
        pytest.skip("clustering test did not run")
    outs = {}
    for mode in (False, True):
        out_dir = os.path.join(root, "chunked_async" if mode else "chunked_sync")
        os.makedirs(out_dir, exist_ok=True)
        out_csv = os.path.join(out_dir, "output.csv")
        random.seed(1)
        acav100m_amd.manual_seed(1)
        Cli().run(shards_path=os.path.join(root, "clusters", "shard-{000000..000003}.pkl"),
                  meta_path=os.path.join(root, "videos"), out_path=out_csv, chunk_size=2,
                  **{"computation.load_async": mode})
        caches = sorted(os.listdir(os.path.join(out_dir, "caches")))
        assert len(caches) == 2 and all(c.startswith("cache_") and c.endswith("_output.csv") for c in caches)
        Cli().reduce_csvs(out_path=out_csv)
        outs[mode] = open(out_csv).read().splitlines()
        # each chunk (2 shards = 512 clips) selects round(0.2 * 512) = 102 clips
        assert len(outs[mode]) == 204
    assert outs[False] == outs[True]
    # pickle caches (save_cache_as_csvs=False) + `reduce_pkls` (chunk.py:56-112,134-141) end in the same output.csv
    out_dir = os.path.join(root, "chunked_pkl")
    os.makedirs(out_dir, exist_ok=True)
    out_csv = os.path.join(out_dir, "output.csv")
    random.seed(1)
    acav100m_amd.manual_seed(1)
    kw = dict(shards_path=os.path.join(root, "clusters", "shard-{000000..000003}.pkl"), meta_path=os.path.join(root, "videos"),
              out_path=out_csv, chunk_size=2, save_cache_as_csvs=False)
    Cli().run(**kw)
    caches = sorted(os.listdir(os.path.join(out_dir, "caches")))
    assert len(caches) == 2 and all(c.startswith("cache_") and c.endswith(".pkl") for c in caches)
    Cli().reduce_pkls(**kw)
    assert open(out_csv).read().splitlines() == outs[False]


def test_compare_measures_verb(workdir):
    """`cli.py compare_measures` (tests.py:10-46): the exact-greedy `mem_mi` and `mi` share one canonical evaluation here,
    so their selections agree completely (the reference's two formulations agree on 12-66 % of the picks)."""
    from acav100m_amd.subset_selection.cli import Cli
    root, glob = workdir
    if not os.path.isfile(os.path.join(root, "clusters", "shard-000000.pkl")):
        pytest.skip("clustering test did not run")
    rep = Cli().compare_measures(shards_path=os.path.join(root, "clusters", "shard-{000000..000001}.pkl"),
                                 meta_path=os.path.join(root, "videos"), out_path=os.path.join(root, "cmp", "output.csv"),
                                 **{"subset.size": 40})
    assert len(rep) == 1 and rep[0][1:3] == ("mem_mi", "mi") and rep[0][3] == 1.0 and rep[0][4] == 0.0


def test_chunked_run_lockstep_chunks(workdir):
    """computation.concurrent_chunks=2: both chunks selected in lockstep by one set of launches; every chunk gets
    its own generator (random_seed + 1 + chunk number), so the result equals running that chunk alone with it."""
    import acav100m_amd
    from acav100m_amd.rng import Generator
    from acav100m_amd.subset_selection.cli import Cli
    from acav100m_amd import shards as io
    from oracle import oracle as O
    root, glob = workdir
    if not os.path.isfile(os.path.join(root, "clusters", "shard-000000.pkl")):
        pytest.skip("clustering test did not run")
    out_dir = os.path.join(root, "chunked_lockstep")
    os.makedirs(out_dir, exist_ok=True)
    out_csv = os.path.join(out_dir, "output.csv")
    random.seed(1)
    Cli().run(shards_path=os.path.join(root, "clusters", "shard-{000000..000003}.pkl"),
              meta_path=os.path.join(root, "videos"), out_path=out_csv, chunk_size=2,
              **{"computation.concurrent_chunks": 2, "computation.random_seed": 7})
    Cli().reduce_csvs(out_path=out_csv)
    lines = open(out_csv).read().splitlines()
    assert len(lines) == 204
    # expected: each chunk alone, oracle, same Python shuffle order (chunk 0 then chunk 1) and generator seeds 8, 9
    random.seed(1)
    want = []
    for num in range(2):
        paths = [os.path.join(root, "clusters", "shard-%06d.pkl" % s) for s in (2 * num, 2 * num + 1)]
        a, types, shard_names, filenames = io.load_assignment_shards(paths)
        cand = list(range(len(a)))
        random.shuffle(cand)
        pairs = list(itertools.combinations(range(len(types)), 2))
        res = O.BatchMI(a, int(a.max()) + 1, pairs).run_greedy(cand[1:], cand[:1], round(0.2 * len(a)), 20, 4,
                                                               O.Rng(7 + 1 + num))
        want += [filenames[s] for s in sorted(res["S"])]
    got = [r[1] for r in csv.reader(_io.StringIO("\n".join(lines)))]
    assert got == want


def _run_cli(module, argv, extra_env):
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-m", module] + argv, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_cluster_cli_spawns_one_process_per_gpu(workdir):
    """`cli.py cluster --computation.num_gpus=2` started plainly re-executes itself once per GPU like the reference
    (script.py:52-65); the processes join a group, deal the 10 clusterings out between them (view-parallel training),
    broadcast the states and label their own shards.  Two processes on the one GPU of this box (gloo): the files must
    be those of the one-process run, byte for byte in the labels."""
    root, glob = workdir
    common = ["cluster", "--feature_path=" + glob, "--meta_path=" + os.path.join(root, "videos")]
    out1, out2 = os.path.join(root, "cl_one"), os.path.join(root, "cl_two")
    _run_cli("acav100m_amd.clustering.cli", common + ["--out_path=" + out1, "--computation.num_gpus=1"], {"ACAV_SEED": "0"})
    log = _run_cli("acav100m_amd.clustering.cli", common + ["--out_path=" + out2, "--computation.num_gpus=2"],
                   {"ACAV_SEED": "0", "ACAV_DIST_BACKEND": "gloo"})
    assert log.count("done") == 2
    for s in range(4):
        name = "shard-%06d.pkl" % s
        a = pickle.load(open(os.path.join(out1, name), "rb"))
        b = pickle.load(open(os.path.join(out2, name), "rb"))
        assert len(a) == len(b) == 256
        for ra, rb in zip(a, b):
            assert ra["filename"] == rb["filename"]
            for key in ("audio_assignments", "video_assignments"):
                assert {k: int(v) for k, v in ra[key][0]["array"].items()} == {k: int(v) for k, v in rb[key][0]["array"].items()}
    # the checkpoints are the reference's file: torch.save of {model_key: {layer: attrs}}, written by rank 0 only
    import torch
    caches = sorted(f for f in os.listdir(out2) if f.startswith("cache_epoch_"))
    assert len(caches) == 2
    c1 = torch.load(os.path.join(out1, caches[-1]), weights_only=False)
    c2 = torch.load(os.path.join(out2, caches[-1]), weights_only=False)
    assert sorted(c1.keys()) == sorted(c2.keys()) and len(c1) == 2
    for mk in c1:
        assert sorted(c1[mk].keys()) == ["layer_%d" % i for i in range(5)]
        for layer in c1[mk]:
            assert np.array_equal(c1[mk][layer]["centers"], c2[mk][layer]["centers"])
            assert c1[mk][layer]["count"] == c2[mk][layer]["count"]
    logs = [f for f in os.listdir(out2) if f.startswith("log_")]
    assert len(logs) == 2  # one manifest per process, each listing its own shards (save.py:9-17)


def test_cluster_cli_striped_mode_two_workers(workdir):
    """`cli.py cluster --clustering.multi_gpu=striped --computation.num_gpus=2`: SURVEY 8(e)'s row-striped partition -- every
    worker HOLDS only the rows of its own shards (rank::2), the global batch stream is the one-GPU run's (shards in global order,
    32 rows per step, 2 epochs), the rows travel in bulk to the worker that runs a clustering's chain.  The files must be those
    of the one-process run (and of the default `views` mode): every label, and the checkpoints' centres / counts bit for bit."""
    import torch
    root, glob = workdir
    out1 = os.path.join(root, "cl_one")
    if not os.path.isfile(os.path.join(out1, "shard-000000.pkl")):
        pytest.skip("the one-process run of the spawn test is missing")
    out = os.path.join(root, "cl_striped")
    log = _run_cli("acav100m_amd.clustering.cli",
                   ["cluster", "--feature_path=" + glob, "--meta_path=" + os.path.join(root, "videos"), "--out_path=" + out,
                    "--computation.num_gpus=2", "--clustering.multi_gpu=striped"], {"ACAV_SEED": "0", "ACAV_DIST_BACKEND": "gloo"})
    assert log.count("done") == 2 and "mode striped: 32 steps of 1 x 32 rows per epoch, 2 epochs" in log
    for s in range(4):
        name = "shard-%06d.pkl" % s
        a = pickle.load(open(os.path.join(out1, name), "rb"))
        b = pickle.load(open(os.path.join(out, name), "rb"))
        assert len(a) == len(b) == 256
        for ra, rb in zip(a, b):
            assert ra["filename"] == rb["filename"]
            for key in ("audio_assignments", "video_assignments"):
                assert {k: int(v) for k, v in ra[key][0]["array"].items()} == {k: int(v) for k, v in rb[key][0]["array"].items()}
    caches = sorted(f for f in os.listdir(out) if f.startswith("cache_epoch_"))
    assert len(caches) == 2
    for name in caches:
        c1 = torch.load(os.path.join(out1, name), weights_only=False)
        c2 = torch.load(os.path.join(out, name), weights_only=False)
        for mk in c1:
            for layer in c1[mk]:
                assert np.array_equal(c1[mk][layer]["centers"], c2[mk][layer]["centers"]), (name, mk, layer)
                assert np.array_equal(c1[mk][layer]["counts"], c2[mk][layer]["counts"])
                assert c1[mk][layer]["count"] == c2[mk][layer]["count"]


def test_cluster_cli_rows_mode_two_workers(workdir):
    """`cli.py cluster --clustering.multi_gpu=rows --computation.num_gpus=2`: the LARGE-BATCH multi-GPU mode through the CLI
    (not the reference's N-GPU run -- that is test_cluster_cli_reference_mode_two_workers): every worker holds the rows of
    its own shards (rank::2), a step's global batch is 32 rows of worker 0 followed by 32 rows of worker 1 (global batch
    64), ceil(2 / 2) = 1 epoch.  Two workers on the one GPU of this box (gloo).  Checked against the ORACLE fed that batch
    stream: the saved centres of all ten clusterings bit for bit, and every written label."""
    import torch
    from oracle import oracle as O
    from acav100m_amd import shards as io
    root, glob = workdir
    out = os.path.join(root, "cl_rows")
    log = _run_cli("acav100m_amd.clustering.cli",
                   ["cluster", "--feature_path=" + glob, "--meta_path=" + os.path.join(root, "videos"), "--out_path=" + out,
                    "--computation.num_gpus=2", "--clustering.multi_gpu=rows"], {"ACAV_SEED": "0", "ACAV_DIST_BACKEND": "gloo"})
    assert log.count("done") == 2 and "mode rows: 16 steps of 2 x 32 rows per epoch, 1 epochs" in log
    caches = sorted(f for f in os.listdir(out) if f.startswith("cache_epoch_"))
    assert len(caches) == 1  # ceil(epochs / num_gpus) = 1 epoch, written by worker 0
    saved = torch.load(os.path.join(out, caches[0]), weights_only=False)
    # the oracle on the same stream: worker r's rows = its shards in order
    paths = sorted(io.brace_expand(glob))
    models, audio = ['layer_vggish', 'layer_slow_fast'], ('vggish', 'layer_vggish')
    tabs = [io.load_feature_shards([__import__("pathlib").Path(q) for q in paths[r::2]], model_order=models, audio_models=audio)
            for r in range(2)]
    views = list(tabs[0].views)
    K, b, lr = 32, 32, 0.1 ** 2  # run_clustering.py:168: lr = 0.1 ** (2 + epoch // 5)
    rng = O.Rng(0)
    refs = [O.KMeans(tabs[0].views[v].shape[1], K, rng) for v in views]  # ten inits in view order (run_clustering.py:32-44)
    rng.u32(), rng.u32()                                                   # the DataLoader iterator's seed draw
    steps = min(len(t) for t in tabs) // b
    # warm-up labels: each worker draws the labels of ITS rows batch by batch across the clusterings (the reference loop
    # steps every clustering per batch); both workers are seeded alike and draw the same stream
    need = min(steps, -(-(10 * K) // (2 * b)))
    warms = [[None] * need for _ in views]
    for t in range(need):
        for i in range(len(views)):
            warms[i][t] = np.argmin(rng.rand(K, b), axis=0)
    for ref, v, warm in zip(refs, views, warms):
        for t in range(steps):
            xg = np.concatenate([tabs[r].views[v][t * b:(t + 1) * b] for r in range(2)])
            if t < need:
                ref.apply_update(xg, np.concatenate([warm[t], warm[t]]), lr)
            else:
                ref.lr = lr
                ref.add(xg)
    for ref, (kind, mk, layer) in zip(refs, views):
        got = saved[mk][layer]
        assert np.array_equal(np.asarray(got["centers"]), ref.centers), (mk, layer)
        assert got["count"] == ref.count == steps * 2 * b
    for s in range(4):
        rows = pickle.load(open(os.path.join(out, "shard-%06d.pkl" % s), "rb"))
        tab = tabs[s % 2]
        ids = tab.shard_rows["shard-%06d" % s]
        assert len(rows) == len(ids) == 256
        for ref, view in zip(refs, views):
            kind, mk, layer = view
            want = ref.calc_best(tab.views[view][ids])[0]
            key = "audio_assignments" if kind == "audio" else "video_assignments"
            got = np.array([int(r[key][0]["array"][layer]) for r in rows])
            assert np.array_equal(got, want), (s, mk, layer)


def test_cluster_cli_reference_mode_two_workers(workdir, golden_dir):
    """`cli.py cluster --clustering.multi_gpu=reference --computation.num_gpus=2`: the reference's OWN two-GPU training run
    (sgd_clustering.py:94-129 under is_distributed) -- every worker feeds int(32 / 2) = 16 rows per step
    (data/clustering.py:25) of its stream over ALL four shards in the rotated order of mps/distributed.py:433-437
    (worker 0: shards 0 2 1 3, worker 1: 1 3 0 2 -- read from tests/golden/ddp_stream.npz, which the reference's own
    node_selection produced), ceil(2 / 2) = 1 epoch of 2 * 1024 / 32 = 64 steps at the global batch of 32.  Two workers on
    the one GPU of this box (gloo), each HOLDING only its own shards.  Checked against the ORACLE fed the reference's
    batch stream: the saved centres of all ten clusterings bit for bit, and every written label."""
    import torch
    from pathlib import Path
    from oracle import oracle as O
    from acav100m_amd import shards as io
    root, glob = workdir
    out = os.path.join(root, "cl_reference")
    log = _run_cli("acav100m_amd.clustering.cli",
                   ["cluster", "--feature_path=" + glob, "--meta_path=" + os.path.join(root, "videos"), "--out_path=" + out,
                    "--computation.num_gpus=2", "--clustering.multi_gpu=reference", "--computation.num_workers=0"], {"ACAV_SEED": "0", "ACAV_DIST_BACKEND": "gloo"})
    assert log.count("done") == 2 and "mode reference: 64 steps of 2 x 16 rows per epoch, 1 epochs" in log
    caches = sorted(f for f in os.listdir(out) if f.startswith("cache_epoch_"))
    assert len(caches) == 1  # ceil(epochs / num_gpus) = 1 epoch, written by worker 0
    saved = torch.load(os.path.join(out, caches[0]), weights_only=False)
    g = np.load(os.path.join(golden_dir, "ddp_stream.npz"))
    case = "w2_even"
    assert [int(x) for x in g[case + "_sizes"]] == [256] * 4 and int(g[case + "_world"]) == 2
    lb, epochs = int(g[case + "_local_batch"]), int(g[case + "_epochs"])
    paths = [Path(q) for q in sorted(io.brace_expand(glob))]
    models, audio = ['layer_vggish', 'layer_slow_fast'], ('vggish', 'layer_vggish')
    shard_tabs = [io.load_feature_shards([q], model_order=models, audio_models=audio) for q in paths]
    views = list(shard_tabs[0].views)
    K, lr = 32, 0.1 ** 2  # run_clustering.py:168: lr = 0.1 ** (2 + epoch // 5)
    rng = O.Rng(0)
    refs = [O.KMeans(shard_tabs[0].views[v].shape[1], K, rng) for v in views]  # ten inits in view order (run_clustering.py:32-44)
    rng.u32(), rng.u32()                                                         # the DataLoader iterator's seed draw
    steps = 1024 // lb
    assert epochs == 1 and steps * lb == int(g[case + "_length"])
    need = min(steps, -(-(10 * K) // (2 * lb)))
    warms = [[None] * need for _ in views]
    for t in range(need):  # each worker labels its own 16 rows (calc_best(batch): torch.rand(k, 16)); same seed, same draws
        for i in range(len(views)):
            warms[i][t] = np.argmin(rng.rand(K, lb), axis=0)
    for ref, v, warm in zip(refs, views, warms):
        streams = [np.concatenate([shard_tabs[int(s)].views[v] for s in g[case + "_order_rank%d" % r]]) for r in range(2)]
        for t in range(steps):
            xg = np.concatenate([streams[r][t * lb:(t + 1) * lb] for r in range(2)])  # all_gather order: rank-major
            if t < need:
                ref.apply_update(xg, np.concatenate([warm[t], warm[t]]), lr)
            else:
                ref.lr = lr
                ref.add(xg)
    for ref, (kind, mk, layer) in zip(refs, views):
        got = saved[mk][layer]
        assert np.array_equal(np.asarray(got["centers"]), ref.centers), (mk, layer)
        assert got["count"] == ref.count == steps * 2 * lb
    for s in range(4):
        rows = pickle.load(open(os.path.join(out, "shard-%06d.pkl" % s), "rb"))
        assert len(rows) == 256
        for ref, view in zip(refs, views):
            kind, mk, layer = view
            want = ref.calc_best(shard_tabs[s].views[view])[0]
            key = "audio_assignments" if kind == "audio" else "video_assignments"
            got = np.array([int(r[key][0]["array"][layer]) for r in rows])
            assert np.array_equal(got, want), (s, mk, layer)
    # more GPUs than shards: the reference clamps num_gpus (script.py:22,37); here the run refuses instead of writing
    # checkpoints and labels of untrained clusterings
    one = os.path.join(os.path.dirname(glob), "shard-{000000..000000}.pkl")
    with pytest.raises(AssertionError):
        _run_cli("acav100m_amd.clustering.cli",
                 ["cluster", "--feature_path=" + one, "--meta_path=" + os.path.join(root, "videos"),
                  "--out_path=" + os.path.join(root, "cl_reference_none"), "--computation.num_gpus=2",
                  "--clustering.multi_gpu=reference"], {"ACAV_SEED": "0", "ACAV_DIST_BACKEND": "gloo"})
    assert not os.path.isdir(os.path.join(root, "cl_reference_none")) or not any(
        f.startswith("cache_epoch_") or f.endswith(".pkl") for f in os.listdir(os.path.join(root, "cl_reference_none")))


def test_subset_cli_chunks_spawn_per_gpu(workdir):
    """`cli.py run --chunk_size=2 --computation.num_gpus=2`: one process per GPU (chunk.py:28,53), no process group,
    every process selects from its own block of chunks into caches/cache_{parent pid}_{rank}_{i}_output.csv;
    reduce_csvs merges them."""
    root, glob = workdir
    src = os.path.join(root, "cl_one")
    if not os.path.isfile(os.path.join(src, "shard-000000.pkl")):
        pytest.skip("clustering spawn test did not run")
    out = os.path.join(root, "sel_two", "output.csv")
    args = ["--shards_path=" + os.path.join(src, "shard-{000000..000003}.pkl"), "--meta_path=" + os.path.join(root, "videos"),
            "--out_path=" + out, "--chunk_size=2", "--computation.num_gpus=2"]
    _run_cli("acav100m_amd.subset_selection.cli", ["run"] + args, {"ACAV_SEED": "0", "ACAV_OVERSUBSCRIBE": "1"})
    caches = sorted(os.listdir(os.path.join(root, "sel_two", "caches")))
    assert len(caches) == 2 and {c.split("_")[2] for c in caches} == {"0", "1"}, caches  # ranks 0 and 1, same parent pid
    assert len({c.split("_")[1] for c in caches}) == 1
    _run_cli("acav100m_amd.subset_selection.cli", ["reduce_csvs"] + args, {})
    rows = list(csv.reader(open(out)))
    assert len(rows) == 2 * round(0.2 * 512) and len({tuple(r[:2]) for r in rows}) == len(rows)
    assert {r[0] for r in rows} == {"shard-%06d" % s for s in range(4)}


def test_streamed_clustering_equals_resident(tmp_path_factory, golden_dir):
    """Out-of-core clustering: with a device budget smaller than the data the shards stream
    through the GPU in row groups (next group loaded by a host thread meanwhile), the batch stream runs ACROSS group
    boundaries (250-row shards, b = 32: every group hands a partial batch to the next one) and nothing holds [N, d]
    for all N.  Checkpoints and assignment files must be those of the resident run."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, golden_dir)
    import synth
    import acav100m_amd
    from acav100m_amd.clustering.cli import Cli
    root = str(tmp_path_factory.mktemp("acav_stream"))
    glob = synth.write_feature_shards(root, n_shards=5, rows=250, seed=3)
    outs = {}
    # budgets: about one shard per row group, two, and most of the data
    modes = [("resident", None), ("streamed", str(10_000_000)), ("streamed_small", str(4_000_000)), ("streamed_big", str(30_000_000))]
    for mode, budget in modes:
        if budget is None:
            os.environ.pop("ACAV_RESIDENT_BYTES", None)
        else:
            os.environ["ACAV_RESIDENT_BYTES"] = budget
        try:
            acav100m_amd.manual_seed(0)
            out = os.path.join(root, "clusters_" + mode)
            saved = Cli().cluster(feature_path=glob, out_path=out, meta_path=os.path.join(root, "videos"))
            assert [p.name for p in saved] == ["shard-%06d.pkl" % s for s in range(5)]
            outs[mode] = out
        finally:
            os.environ.pop("ACAV_RESIDENT_BYTES", None)
    for streamed in [m for m, b in modes if b is not None]:
        for s in range(5):
            name = "shard-%06d.pkl" % s
            a = pickle.load(open(os.path.join(outs["resident"], name), "rb"))
            b = pickle.load(open(os.path.join(outs[streamed], name), "rb"))
            assert len(a) == len(b) == 250
            for ra, rb in zip(a, b):
                assert ra["filename"] == rb["filename"]
                for key in ("audio_assignments", "video_assignments"):
                    assert {k: int(v) for k, v in ra[key][0]["array"].items()} == {k: int(v) for k, v in rb[key][0]["array"].items()}
        for e in (0, 1):
            ca = [f for f in os.listdir(outs["resident"]) if f.startswith("cache_epoch_%d_" % e)][0]
            c1 = torch.load(os.path.join(outs["resident"], ca), weights_only=False)
            c2 = torch.load(os.path.join(outs[streamed], ca), weights_only=False)
            for mk in c1:
                for layer in c1[mk]:
                    assert np.array_equal(c1[mk][layer]["centers"], c2[mk][layer]["centers"]), (streamed, e, mk, layer)
                    assert np.array_equal(c1[mk][layer]["counts"], c2[mk][layer]["counts"])
                    assert c1[mk][layer]["count"] == c2[mk][layer]["count"]


def test_worker_process_loader_and_writer_equal_the_serial_run(tmp_path_factory, golden_dir):
    """round 3: with >= 16 shards the clustering CLI reads the shards with worker processes (straight into reused, pinned
    shared memory) and writes the assignment shards with the same pool.  Resident and streamed, the files must be the
    serial run's, byte for byte; a shard that does not match its metadata sends its group back to the plain loop."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, golden_dir)
    import synth
    import acav100m_amd
    from acav100m_amd.clustering.cli import Cli
    root = str(tmp_path_factory.mktemp("acav_workers"))
    glob = synth.write_feature_shards(root, n_shards=24, rows=48, seed=9, audio_dims=[64, 128], video_dims=[88, 352])
    outs = {}
    modes = [("serial", "0", None), ("workers", "4", None), ("workers_streamed", "4", str(1_500_000))]
    for mode, workers, budget in modes:
        os.environ["ACAV_LOAD_WORKERS"] = workers
        if budget is not None:
            os.environ["ACAV_RESIDENT_BYTES"] = budget
        try:
            acav100m_amd.manual_seed(0)
            out = os.path.join(root, "clusters_" + mode)
            saved = Cli().cluster(feature_path=glob, out_path=out, meta_path=os.path.join(root, "videos"),
                                  **{"clustering.ncentroids": 16})
            assert len(saved) == 24
            outs[mode] = out
        finally:
            os.environ.pop("ACAV_LOAD_WORKERS", None)
            os.environ.pop("ACAV_RESIDENT_BYTES", None)
    for mode in ("workers", "workers_streamed"):
        for s in range(24):
            name = "shard-%06d.pkl" % s
            assert open(os.path.join(outs["serial"], name), "rb").read() == open(os.path.join(outs[mode], name), "rb").read(), (mode, name)
