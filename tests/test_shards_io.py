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
    # the reference's meta_cache.pkl is trusted for the shards it lists, the others are counted from their json (data/meta.py:11-20)
    io.dump_pickle({"shard-000001": 5}, tmp_path / "videos" / "meta_cache.pkl")
    assert io.shard_sizes_from_meta(paths, tmp_path / "videos", use_cache=True) == {"shard-000000": 8, "shard-000001": 5}
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


def _columns_equal(a, b):
    assert a["filename"] == b["filename"] and a["shard_name"] == b["shard_name"] and a["shard_size"] == b["shard_size"]
    # pickle.load gives rows that shared a str object ONE object again; the native reader reproduces the identities
    ida = [[i for i, y in enumerate(a["shard_name"]) if y is x][0] for x in a["shard_name"]]
    idb = [[i for i, y in enumerate(b["shard_name"]) if y is x][0] for x in b["shard_name"]]
    assert ida == idb
    assert list(a["tags"].items()) == list(b["tags"].items())
    assert list(a["views"]) == list(b["views"])
    for v in a["views"]:
        assert a["views"][v].dtype == np.float32 and a["views"][v].shape == b["views"][v].shape
        assert np.array_equal(a["views"][v], b["views"][v])


def _synthetic_rows(rs, n, layout, name="shard-000007"):
    rows = []
    for i in range(n):
        va, vb, vv = (rs.randn(d).astype(np.float32) for d in (8, 12, 20))
        if layout == "dict":
            audio, video = {"conv4": va, "fc": vb}, {"layer_0": vv}
        elif layout == "list":
            audio, video = [va, vb], [vv]
        elif layout == "tuple":
            audio, video = (va, vb), (vv,)
        else:  # the bare vector of a non-layer extractor
            audio, video = va, vv
        rows.append({"video_features": [{"model_key": "vis", "extractor_name": "V", "dataset": None, "array": video}],
                     "audio_features": [{"model_key": "aud", "extractor_name": "A", "dataset": "ds", "array": audio},
                                        {"model_key": "aud2", "array": {"only": vb[::-1].copy()}}],
                     "filename": "clip é中_%d.mp4" % i, "shard_size": n, "shard_name": name,
                     "segment": (10, 20.5), "extra": {"flag": True, "none": None, "big": 1 << 40, "neg": -5, "raw": b"\x00\x01",
                                                      "nested": [1, 2, (3,)], "long": "x" * 300}})
    return rows


def test_native_shard_reader_equals_pickle(tmp_path):
    """csrc/acav_shardio.hip (acav_pkl_shard_*: the shard file mapped, its pickle opcodes walked once, every vector copied
    straight into the table) against pickle.load + _shard_columns_from_rows -- what the reference's own loader does
    (clustering/code/data/clustering.py:78-113, 172): every array layout, pickle protocols 3-5 (2: refused), missing keys, shared and
    distinct shard_name objects; shards outside the reader's subset must be REFUSED (the caller then unpickles)."""
    from acav100m_amd import _lib
    lib = _lib.load_library()
    rs = np.random.RandomState(0)
    seen_native = 0
    for layout in ("dict", "list", "tuple", "bare"):
        for proto in (2, 3, 4, 5):
            rows = _synthetic_rows(rs, 9, layout)
            if layout == "list" and proto == 4:       # rows without the optional keys; distinct-but-equal name objects
                for i, r in enumerate(rows):
                    del r["shard_size"]
                    if i % 2:
                        del r["shard_name"]
                    else:
                        r["shard_name"] = "".join(["shard-", "x"])  # a fresh object per row
            p = tmp_path / ("s_%s_%d.pkl" % (layout, proto))
            with open(p, "wb") as f:
                pickle.dump(rows, f, protocol=proto)
            ref = io._shard_columns_from_rows(io.load_pickle(p), p.stem)
            nat = io.read_shard_native(lib, p)
            if proto == 2:  # Python 3 writes bytes as _codecs.encode(latin-1 text) there: not a payload to copy -- refused
                assert nat is None
                continue
            assert nat is not None, (layout, proto, lib.acav_last_error())
            seen_native += 1
            _columns_equal(nat, ref)
            # into a caller's table at an offset, with a wider row stride
            dest = {v: np.full((20, m.shape[1]), 7.0, np.float32) for v, m in ref["views"].items()}
            nat2 = io.read_shard_native(lib, p, dest, 5, 9)
            _columns_equal(nat2, ref)
            for v, m in dest.items():
                assert np.array_equal(m[5:14], ref["views"][v]) and (m[:5] == 7).all() and (m[14:] == 7).all()
            assert io.read_shard_native(lib, p, dest, 0, 8) == "layout"  # more rows than reserved
    assert seen_native == 12
    # outside the subset: refused, never guessed
    base = _synthetic_rows(rs, 4, "dict")
    cases = {}
    r = [dict(x) for x in base]
    r[2] = dict(r[2], audio_features=[dict(r[2]["audio_features"][0], array={"conv4": np.zeros(8, np.float64), "fc": np.zeros(12, np.float32)})]
                + r[2]["audio_features"][1:])
    cases["float64 vector"] = r
    r = [dict(x) for x in base]
    r[3] = dict(r[3], audio_features=r[3]["audio_features"][:1])
    cases["rows with different view lists"] = r
    r = [dict(x) for x in base]
    r[1] = dict(r[1], video_features=[dict(r[1]["video_features"][0], array={"layer_0": np.zeros((2, 10), np.float32)})])
    cases["2-d vector"] = r
    r = [dict(x) for x in base]
    r[0] = dict(r[0], video_features=[dict(r[0]["video_features"][0], array={"layer_0": np.zeros(21, np.float32)})])
    cases["rows with different vector lengths"] = r
    cases["big-endian vectors"] = [dict(x, video_features=[dict(x["video_features"][0], array={"layer_0": np.zeros(20, ">f4")})]) for x in base]
    cases["not a list"] = {"rows": base}
    for what, obj in cases.items():
        p = tmp_path / "bad.pkl"
        with open(p, "wb") as f:
            pickle.dump(obj, f)
        assert io.read_shard_native(lib, p) is None, what
    with open(tmp_path / "bad.pkl", "wb") as f:
        pickle.dump(base, f, protocol=0)  # text opcodes
    assert io.read_shard_native(lib, tmp_path / "bad.pkl") is None
    with open(tmp_path / "bad.pkl", "wb") as f:
        f.write(pickle.dumps(base)[:-40])  # truncated
    assert io.read_shard_native(lib, tmp_path / "bad.pkl") is None
    assert io.read_shard_native(lib, tmp_path / "missing.pkl") is None


def test_native_loader_equals_worker_processes(tmp_path, monkeypatch):
    """load_feature_shards through the native group reader (acav_pkl_load_group) == the same call with ACAV_SHARD_NATIVE=0 (worker processes +
    pickle.load) == the plain loop without the library's reader: a group with a short shard, an unreadable one and one the
    native reader refuses (float64 vectors: read by pickle.load)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import synth
    from collections import OrderedDict
    glob = synth.write_feature_shards(str(tmp_path), n_shards=18, rows=10, seed=9, audio_dims=[8, 4], video_dims=[16])
    paths = sorted(io.brace_expand(glob))
    sizes = {os.path.basename(p)[:-4]: 10 for p in paths}
    io.dump_pickle(io.load_pickle(paths[2])[:6], paths[2])
    with open(paths[11], "wb") as f:
        f.write(b"\x80\x04garbage")
    rows = io.load_pickle(paths[5])
    for r in rows:
        r["audio_features"][0]["array"]["layer_1"] = r["audio_features"][0]["array"]["layer_1"].astype(np.float64)
    io.dump_pickle(rows, paths[5])
    monkeypatch.setenv("ACAV_SHARD_NATIVE", "0")
    plain = io.load_feature_shards(paths, sidecar="off")
    dims = OrderedDict((v, m.shape[1]) for v, m in plain.views.items())
    procs = io.load_feature_shards(paths, sidecar="off", workers=3, expect_rows=sizes, expect_views=dims)
    monkeypatch.setenv("ACAV_SHARD_NATIVE", "1")
    calls, taken = [], []
    orig, orig_group = io.read_shard_native, io._load_native
    monkeypatch.setattr(io, "read_shard_native", lambda *a, **k: calls.append(1) or orig(*a, **k))
    monkeypatch.setattr(io, "_load_native", lambda *a, **k: taken.append(orig_group(*a, **k)) or taken[-1])
    native = io.load_feature_shards(paths, sidecar="off", workers=3, expect_rows=sizes, expect_views=dims)
    plain_native = io.load_feature_shards(paths, sidecar="off")
    assert len(calls) == len(paths) and taken == [native] and getattr(native, "_shm", None)  # the group call, then the plain loop
    for t in (procs, native, plain_native):
        assert len(t) == len(plain) == 10 * 16 + 6
        assert t.filename == plain.filename and t.shard_name == plain.shard_name and t.shard_size == plain.shard_size
        assert t.shard_rows == plain.shard_rows and t.tags == plain.tags and list(t.views) == list(plain.views)
        for v in plain.views:
            assert np.array_equal(t.views[v], plain.views[v])


def test_native_assignment_reader_equals_pickle(tmp_path, monkeypatch):
    """acav_pkl_assign_load_group (the selection stage's input shards parsed by the library) against the reference's row walk
    (subset_selection/code/dataloader.py:17-69 as restated in _assignment_rows_to_lists): dict and list 'array's, numpy.int64
    and plain-int labels, protocols 3-5, an empty shard; shards it must refuse fall back to pickle.load -- same result."""
    rs = np.random.RandomState(4)
    paths = []
    for s, proto in enumerate((3, 4, 5, 4)):
        n = 0 if s == 3 else 7 + s
        cols = [(("audio", "am", "layer_1"), rs.randint(0, 999, n)), (("audio", "am", "layer_0"), rs.randint(0, 999, n)),
                (("video", "vm", "layer_0"), rs.randint(0, 1 << 40, n))]
        rows = io._assignment_rows_from_columns(["clip_%d_%d é.mp4" % (s, i) for i in range(n)], [n] * n, ["shard-%06d" % s] * n,
                                                [(("audio", "am"), ("A", "ds")), (("video", "vm"), ("V", None))], cols)
        if s == 1:   # list arrays and plain ints
            for r in rows:
                r["audio_assignments"][0]["array"] = [int(v) for v in r["audio_assignments"][0]["array"].values()]
                r["video_assignments"][0]["array"] = [int(v) for v in r["video_assignments"][0]["array"].values()]
        p = tmp_path / ("shard-%06d.pkl" % s)
        with open(p, "wb") as f:
            pickle.dump(rows, f, protocol=proto)
        paths.append(p)
    nat = io._assignment_shards_native(paths)
    assert sorted(nat) == [0, 1, 2, 3]
    for i, p in enumerate(paths):
        per_row, sn, fn = io._assignment_rows_to_lists(io.load_pickle(p))
        mat, ty = io.rows_to_matrix(per_row)
        if per_row:
            assert nat[i][1] == ty
        assert np.array_equal(nat[i][0].reshape(mat.shape), mat) and nat[i][0].dtype == np.int64
        assert nat[i][2] == sn and nat[i][3] == fn
    # shards 0 and 2 share their clusterings: the loader's result with and without the library's reader
    monkeypatch.setenv("ACAV_SHARD_NATIVE", "0")
    ref = io.load_assignment_shards([paths[0], paths[2], paths[3]], sidecar="off")
    monkeypatch.setenv("ACAV_SHARD_NATIVE", "1")
    got = io.load_assignment_shards([paths[0], paths[2], paths[3]], sidecar="off")
    assert np.array_equal(ref[0], got[0]) and ref[1:] == got[1:]
    with pytest.raises(KeyError):  # list arrays name their layers layer_0 / layer_1: other clusterings than shard 0's? no: same names
        io.load_assignment_shards([paths[0], tmp_path / "other.pkl"], sidecar="off") if _write_other(tmp_path) else None
    # refused: float labels, rows with different clusterings, a missing shard_name
    rows = io.load_pickle(paths[0])
    bad = {"float label": [dict(r, video_assignments=[dict(r["video_assignments"][0], array={"layer_0": 1.5})]) for r in rows],
           "differing rows": [dict(r) if i else dict(r, audio_assignments=[]) for i, r in enumerate(rows)],
           "no shard_name": [{k: v for k, v in r.items() if k != "shard_name"} for r in rows]}
    for what, obj in bad.items():
        io.dump_pickle(obj, tmp_path / "bad.pkl")
        assert io._assignment_shards_native([tmp_path / "bad.pkl"]) == {}, what


def _write_other(tmp_path):
    rows = io._assignment_rows_from_columns(["x.mp4"], [1], ["other"], [(("audio", "zz"), ("A", "ds"))], [(("audio", "zz", "layer_0"), np.array([3]))])
    io.dump_pickle(rows, tmp_path / "other.pkl")
    return True
