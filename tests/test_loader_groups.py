"""The training batch stream as ROW GROUPS (acav100m_amd/clustering/run_clustering.py: _RowGroups.iterate_stream): resident
rows gathered into the loader's order, and the same stream cut into groups that fit a small device budget (each group reads
exactly the shards it touches) -- both must deliver the rows the REFERENCE's own DataLoader delivers
(tests/golden/loader_order.npz), epoch by epoch.  Host logic only: the device is 'cpu' here."""
import os
import sys
from collections import OrderedDict
from pathlib import Path

import numpy as np
import pytest
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN_DIR)


def _setup(tmp_path, monkeypatch, case, budget):
    import synth
    from acav100m_amd import shards as io
    from acav100m_amd.clustering import run_clustering as rc
    from acav100m_amd.config import CLUSTERING_DEFAULTS, merge
    g = np.load(os.path.join(GOLDEN_DIR, "loader_order.npz"))
    sizes = [int(x) for x in g[case + "_sizes"]]
    nw, b = int(g[case + "_nw"]), int(g[case + "_batch_size"])
    glob = synth.write_feature_shards(str(tmp_path), n_shards=len(sizes), rows=sizes, seed=1, comps=4, audio_dims=[4], video_dims=[6])
    monkeypatch.setattr(rc, "_device", lambda args: "cpu")
    monkeypatch.setenv("ACAV_LOAD_WORKERS", "0")
    args = merge(CLUSTERING_DEFAULTS, {"data.path": glob, "data.meta.path": os.path.join(str(tmp_path), "videos"),
                                       "computation.num_workers": nw})
    paths = [Path(p) for p in sorted(io.brace_expand(glob))]
    meta = io.shard_sizes_from_meta(paths, args.data.meta.path, use_cache=False)
    full = io.load_feature_shards(paths, model_order=list(args.models), audio_models=tuple(args.model_types.audio))
    row_bytes = 4 * sum(m.shape[1] for m in full.views.values())
    groups = rc._RowGroups(args, paths, meta, row_bytes, budget if budget else 1 << 40,
                           OrderedDict((v, m.shape[1]) for v, m in full.views.items()))
    return g, sizes, nw, b, args, paths, full, row_bytes, groups, rc


@pytest.mark.parametrize("budget", [None, 2600, 8000])  # resident; ~1 shard per half budget; a few shards
@pytest.mark.parametrize("case", ["nw3_even", "nw3_straddle_even", "nw3_ragged", "nw2_ragged", "nw0_ragged", "nw0_even", "nw40_clamped_even"])
def test_row_groups_deliver_the_references_batches(tmp_path, monkeypatch, case, budget):
    from acav100m_amd.parallel import make_plan
    g, sizes, nw, b, args, paths, full, row_bytes, groups, rc = _setup(tmp_path, monkeypatch, case, budget)
    assert groups.streamed == (budget is not None and sum(sizes) * row_bytes > budget)
    order_nw, tail = rc.loader_settings(args)
    assert (order_nw, tail) == (nw, "wrap")
    plan = make_plan("views", sizes, 1, b, 2, num_workers=order_nw, meta_rows=sizes, tail=tail)
    first = np.concatenate([[0], np.cumsum(sizes)])
    for epoch in range(2):
        pe = plan.at_epoch(epoch)
        ext = []
        for _o, f, n in pe.extents[0]:  # row ranges -> (shard, first, rows), as train_clusters does
            while n > 0:
                si = int(np.searchsorted(first, f, side="right")) - 1
                m = min(n, int(first[si + 1]) - f)
                ext.append((si, f - int(first[si]), m))
                f, n = f + m, n - m
        want = g["%s_rank0_epoch%d_rows" % (case, epoch)]
        got = {v: [] for v in full.views}
        n_groups = 0
        for table, part in groups.iterate_stream(paths, ext, b, row_bytes, groups.budget):
            n_groups += 1
            rows = next(iter(part.values())).shape[0]
            assert rows % b == 0 and rows > 0
            if groups.streamed:  # a group never holds more shards than the budget admits (one more where a cut falls into a shard)
                assert len(table) * row_bytes <= budget // 2 + 2 * max(sizes) * row_bytes
            for v, x in part.items():
                got[v].append(x.numpy().copy())
        for v, m in full.views.items():
            assert np.array_equal(np.concatenate(got[v]), m[want]), (case, budget, epoch, v)
        if groups.streamed and budget < 4000:
            assert n_groups > 1


def test_loader_settings_env_and_config(monkeypatch):
    from acav100m_amd.clustering import run_clustering as rc
    from acav100m_amd.config import CLUSTERING_DEFAULTS, merge
    args = merge(CLUSTERING_DEFAULTS, {})
    assert rc.loader_settings(args) == (40, "wrap")  # the reference's default configuration (config.py:29)
    assert rc.loader_settings(merge(CLUSTERING_DEFAULTS, {"computation.num_workers": 0})) == (0, "wrap")
    assert rc.loader_settings(merge(CLUSTERING_DEFAULTS, {"data.loader_order": "single", "data.loader_tail": "drop"})) == (0, "drop")
    monkeypatch.setenv("ACAV_LOADER_ORDER", "single")
    assert rc.loader_settings(args) == (0, "wrap")
    monkeypatch.setenv("ACAV_LOADER_ORDER", "shuffled")
    with pytest.raises(ValueError):
        rc.loader_settings(args)
