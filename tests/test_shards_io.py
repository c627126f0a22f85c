"""CPU tests of the shard / csv / config layer (the reference's file contract)."""
import os
import pickle
import sys

import numpy as np
import pytest

from acav100m_amd import shards as io
from acav100m_amd.config import CLUSTERING_DEFAULTS, SUBSET_DEFAULTS, merge, parse_cli


def test_brace_expand_and_cli_parsing():
    assert io.brace_expand("d/shard-{000008..000011}.pkl") == ["d/shard-%06d.pkl" % i for i in range(8, 12)]
    assert io.brace_expand("x{a,b}.pkl") == ["xa.pkl", "xb.pkl"] and io.brace_expand("plain") == ["plain"]
    assert io.to_brace(["a", "b"]) == "{a,b}" and io.to_brace(["a"]) == "a"
    cmd, kw = parse_cli(["cluster", "--feature_path=/x/s-{0..1}.pkl", "--clustering.ncentroids=64", "--debug",
                         "--subset.ratio", "0.3"])
    assert cmd == "cluster" and kw == {"feature_path": "/x/s-{0..1}.pkl", "clustering.ncentroids": 64,
                                       "debug": True, "subset.ratio": 0.3}
    a = merge(CLUSTERING_DEFAULTS, {"clustering.ncentroids": 64, "data.output.path": "rel/out", "brand.new": 1})
    assert a.clustering.ncentroids == 64 and a.clustering.epochs == 2 and a.brand.new == 1
    assert a.data.output.path.is_absolute() and a.no_such_key is None and a.data.batch_size == 32
    s = merge(SUBSET_DEFAULTS, {})
    assert (s.batch.batch_size, s.batch.selection_size, s.batch.keep_unselected) == (20, 4, True)
    assert s.subset.ratio == 0.2 and s.measure_name == "batch_mi" and s.clustering.pairing == "combination"


def test_feature_table_and_assignment_schema(tmp_path, golden_dir):
    sys.path.insert(0, golden_dir)
    import synth
    glob = synth.write_feature_shards(str(tmp_path), n_shards=2, rows=8, seed=1, audio_dims=[4, 6], video_dims=[5])
    paths = io.brace_expand(glob)
    table = io.load_feature_shards(paths, model_order=["layer_vggish", "layer_slow_fast"],
                                   audio_models=("vggish", "layer_vggish"))
    assert len(table) == 16 and list(table.shard_rows) == ["shard-000000", "shard-000001"]
    # KMeans construction order of the reference: args.models order, then layer index
    assert list(table.views) == [("audio", "layer_vggish", "layer_0"), ("audio", "layer_vggish", "layer_1"),
                                 ("video", "layer_slow_fast", "layer_0")]
    assert [m.shape for m in table.views.values()] == [(16, 4), (16, 6), (16, 5)]
    raw = pickle.load(open(paths[1], "rb"))
    assert np.array_equal(table.views[("audio", "layer_vggish", "layer_1")][8], raw[0]["audio_features"][0]["array"]["layer_1"])
    labels = {v: np.arange(16, dtype=np.int64) * (i + 1) for i, v in enumerate(table.views)}
    rows = io.assignment_rows(table, labels, table.shard_rows["shard-000001"])
    r = rows[3]
    assert set(r) == {"video_assignments", "audio_assignments", "filename", "shard_size", "shard_name"}
    assert r["audio_assignments"][0]["array"] == {"layer_0": 11, "layer_1": 22}
    assert isinstance(r["audio_assignments"][0]["array"]["layer_0"], np.int64)
    assert r["video_assignments"][0]["extractor_name"] == "SLOWFAST_8x8_R50" and r["shard_name"] == "shard-000001"
    out = tmp_path / "clusters" / "shard-000001.pkl"
    io.dump_pickle(rows, out)
    mat, types, shard_names, filenames = io.load_assignment_shards([out])
    assert types == [("layer_slow_fast", "layer_0"), ("layer_vggish", "layer_0"), ("layer_vggish", "layer_1")]
    assert np.array_equal(mat[:, 1], np.arange(8, 16)) and mat.dtype == np.int64 and filenames[0] == rows[0]["filename"]
    # corrupt shard: reported and skipped like the reference
    open(tmp_path / "features" / "shard-000002.pkl", "wb").close()
    t2 = io.load_feature_shards(paths + [str(tmp_path / "features" / "shard-000002.pkl")])
    assert len(t2) == 16
    assert io.shard_sizes_from_meta(paths, tmp_path / "videos") == {"shard-000000": 8, "shard-000001": 8}


def test_partitions_metas_and_output_csv(tmp_path, golden_dir):
    d = tmp_path / "clusters"
    io.dump_json({"hostname": "h", "pid": 1, "timestamp": 100, "time": "t", "shards": ["s0", "s1"]}, d / "log_h_1_100.json")
    io.dump_json({"hostname": "h", "pid": 2, "timestamp": 200, "time": "t", "shards": ["s1", "s2"]}, d / "log_h_2_200.json")
    assert io.load_partitions(d) == {"s0": 0, "s1": 1, "s2": 1}  # newer manifest wins
    io.dump_json([{"filename": "v1_010.mp4", "id": "v1", "segment": [10, 20]}], tmp_path / "m" / "s0.json")
    metas = io.load_metas([d / "s0.pkl", d / "s9.pkl"], tmp_path / "m")
    assert list(metas) == ["s0"] and metas["s0"]["v1_010"]["id"] == "v1"
    out = tmp_path / "o" / "output.csv"
    data = [{"filename": "v1_010.mp4", "shard_name": "s0"}, {"filename": "zz.mp4", "shard_name": "s7"}]
    p, n = io.append_output_csv(data, metas, out)
    p, n2 = io.append_output_csv(data[:1], metas, out)  # append mode ('a+')
    lines = open(p).read().splitlines()
    assert (n, n2) == (2, 1) and lines == ['s0,v1_010.mp4,v1,"[10, 20]"', 'zz,zz.mp4,-1,"[-1.0, -1.0]"'.replace("zz,", "s7,", 1),
                                           's0,v1_010.mp4,v1,"[10, 20]"']
    ref_first = open(os.path.join(golden_dir, "cli_output.csv")).readline().strip()
    assert ref_first == 'shard-000000,vid000000000_010.mp4,vid000000000,"[10, 20]"'  # the reference's own line format
    assert [list(b) for b in io.chunked(list(range(5)), 2)] == [[0, 1], [2, 3], [4]]


def test_columnar_sidecars_round_trip(tmp_path, golden_dir, monkeypatch):
    """the columnar twins of feature / assignment shards reproduce the pkl path exactly, are
    ignored once the pkl changes, and are never written unless asked for."""
    import sys
    import time
    sys.path.insert(0, golden_dir)
    import synth
    from acav100m_amd import shards as io
    glob = synth.write_feature_shards(str(tmp_path), n_shards=3, rows=64, seed=3)
    paths = io.brace_expand(glob)
    monkeypatch.delenv("ACAV_SHARD_SIDECAR", raising=False)
    t0 = time.perf_counter()
    ref = io.load_feature_shards(paths)                      # 'auto' with no sidecar: plain pkl path, nothing written
    t_pkl = time.perf_counter() - t0
    assert not any(io.feature_sidecar_dir(p).exists() for p in paths)
    built = io.load_feature_shards(paths, sidecar="write")   # builds the twins while loading
    assert all(io.feature_sidecar_dir(p).is_dir() for p in paths)
    t0 = time.perf_counter()
    fast = io.load_feature_shards(paths)                     # 'auto' now memory-maps the columns
    t_cols = time.perf_counter() - t0
    for other in (built, fast):
        assert list(other.views) == list(ref.views) and other.tags == ref.tags
        assert other.filename == ref.filename and other.shard_name == ref.shard_name and other.shard_size == ref.shard_size
        assert other.shard_rows == ref.shard_rows
        for v in ref.views:
            assert other.views[v].dtype == np.float32 and np.array_equal(other.views[v], ref.views[v])
    print(f"feature shards: pkl {t_pkl * 1e3:.1f} ms, sidecar {t_cols * 1e3:.1f} ms")
    # a rewritten pkl invalidates its twin (size / mtime stamp): the loader falls back to the pkl
    rows = io.load_pickle(paths[0])
    rows[0]['filename'] = 'changed.mp4'
    io.dump_pickle(rows, paths[0])
    assert io.read_feature_sidecar(paths[0]) is None and io.read_feature_sidecar(paths[1]) is not None
    assert io.load_feature_shards(paths).filename[0] == 'changed.mp4'
    assert io.load_feature_shards(paths, sidecar="off").filename[0] == 'changed.mp4'
    # assignment shards
    table = io.load_feature_shards(paths, sidecar="off")
    rs = np.random.RandomState(0)
    labels = {v: rs.randint(0, 32, len(table)).astype(np.int64) for v in table.views}
    apaths = []
    for shard, ids in table.shard_rows.items():
        p = tmp_path / "clusters" / (shard + ".pkl")
        io.dump_pickle(io.assignment_rows(table, labels, ids), p)
        apaths.append(p)
    ref_a = io.load_assignment_shards(apaths)
    for p in apaths:
        io.write_assignment_sidecar(p)
    fast_a = io.load_assignment_shards(apaths)
    assert np.array_equal(ref_a[0], fast_a[0]) and fast_a[0].dtype == np.int64
    assert ref_a[1] == fast_a[1] and ref_a[2] == fast_a[2] and ref_a[3] == fast_a[3]
    io.dump_pickle(io.load_pickle(apaths[0])[:-1], apaths[0])     # stale twin is ignored
    assert len(io.load_assignment_shards(apaths)[2]) == len(ref_a[2]) - 1
    with pytest.raises(ValueError):
        io.sidecar_mode("sometimes")


def test_parallel_loader_equals_serial(tmp_path):
    """shards read by worker processes straight into shared memory (shards.py _load_parallel) == the plain loop: same
    matrices, same row metadata, same shard -> rows map; a shard with an unexpected size makes the call fall back."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import synth
    from collections import OrderedDict
    glob = synth.write_feature_shards(str(tmp_path), n_shards=20, rows=12, seed=3, audio_dims=[8, 16], video_dims=[24])
    paths = sorted(io.brace_expand(glob))
    serial = io.load_feature_shards(paths, sidecar="off")
    sizes = {os.path.basename(p)[:-4]: 12 for p in paths}
    dims = OrderedDict((v, m.shape[1]) for v, m in serial.views.items())
    par = io.load_feature_shards(paths, sidecar="off", workers=3, expect_rows=sizes, expect_views=dims)
    assert getattr(par, "_shm", None), "the parallel path was not taken"
    assert list(par.views) == list(serial.views) and par.filename == serial.filename
    assert par.shard_name == serial.shard_name and par.shard_size == serial.shard_size
    assert par.shard_rows == serial.shard_rows and par.tags == serial.tags
    for v in serial.views:
        assert np.array_equal(par.views[v], serial.views[v])
    sizes[os.path.basename(paths[5])[:-4]] = 11  # metadata and shard disagree: plain loop, same result
    again = io.load_feature_shards(paths, sidecar="off", workers=3, expect_rows=sizes, expect_views=dims)
    assert getattr(again, "_shm", None) is None and again.filename == serial.filename
    for v in serial.views:
        assert np.array_equal(again.views[v], serial.views[v])


def test_parallel_loader_keeps_going_over_short_and_broken_shards(tmp_path):
    """an incomplete shard (fewer rows than its metadata says) or an unreadable one must not send the whole
    group back to the serial loop: the workers report what they found, the parent closes the gaps.  Result == the serial
    loop over the same files (which reports and skips the broken shard, clustering data/clustering.py:167-182)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import synth
    from collections import OrderedDict
    glob = synth.write_feature_shards(str(tmp_path), n_shards=20, rows=12, seed=5, audio_dims=[8], video_dims=[24, 16])
    paths = sorted(io.brace_expand(glob))
    sizes = {os.path.basename(p)[:-4]: 12 for p in paths}
    io.dump_pickle(io.load_pickle(paths[3])[:7], paths[3])      # a short shard in the middle
    io.dump_pickle(io.load_pickle(paths[19])[:11], paths[19])   # ... and at the end
    with open(paths[8], "wb") as f:                              # an unreadable one
        f.write(b"not a pickle")
    serial = io.load_feature_shards(paths, sidecar="off")
    dims = OrderedDict((v, m.shape[1]) for v, m in serial.views.items())
    par = io.load_feature_shards(paths, sidecar="off", workers=3, expect_rows=sizes, expect_views=dims)
    assert getattr(par, "_shm", None), "the parallel path was abandoned"
    assert len(par) == len(serial) == 12 * 17 + 7 + 11
    assert par.filename == serial.filename and par.shard_name == serial.shard_name and par.shard_rows == serial.shard_rows
    assert os.path.basename(paths[8])[:-4] not in par.shard_rows
    for v in serial.views:
        assert par.views[v].shape == serial.views[v].shape and np.array_equal(par.views[v], serial.views[v])
