"""Deterministic synthetic feature shards in the reference's pkl / json schema:
used by gen_golden.py (to feed the reference CLIs) and by the tests (to feed ours) -- the inputs are
regenerated from the seed instead of being committed."""
import json
import os
import pickle

import numpy as np

AUDIO_DIMS = [64, 128, 256, 512, 128]          # LayerVggish.output_dims  (clustering models/vggish.py:20)
VIDEO_DIMS = [88, 352, 704, 1408, 2304]        # LayerSlowFast.output_dims (clustering models/slowfast.py:31)


def write_feature_shards(root, n_shards=4, rows=256, seed=0, comps=24, audio_dims=AUDIO_DIMS,
                         video_dims=VIDEO_DIMS):
    """root/features/shard-00000i.pkl + root/videos/shard-00000i.json; returns the shard glob."""
    feat_dir, meta_dir = os.path.join(root, "features"), os.path.join(root, "videos")
    os.makedirs(feat_dir, exist_ok=True)
    os.makedirs(meta_dir, exist_ok=True)
    rs = np.random.RandomState(seed)
    cen_a = [rs.randn(comps, d).astype(np.float32) for d in audio_dims]
    cen_v = [rs.randn(comps, d).astype(np.float32) for d in video_dims]
    vid = 0
    for s in range(n_shards):
        name = "shard-%06d" % s
        rows_out, meta = [], []
        rows_s = rows[s] if isinstance(rows, (list, tuple)) else rows  # (per-shard sizes: the loader-order goldens)
        for _ in range(rows_s):
            ga = rs.randint(0, comps)
            gv = ga if rs.rand() < 0.5 else rs.randint(0, comps)  # views share the component w.p. 0.5
            fn = "vid%09d_010.mp4" % vid
            audio = {"layer_%d" % i: (cen_a[i][ga] + 0.3 * rs.randn(d)).astype(np.float32)
                     for i, d in enumerate(audio_dims)}
            video = {"layer_%d" % i: (cen_v[i][gv] + 0.3 * rs.randn(d)).astype(np.float32)
                     for i, d in enumerate(video_dims)}
            rows_out.append({
                "video_features": [{"model_key": "layer_slow_fast", "extractor_name": "SLOWFAST_8x8_R50",
                                    "dataset": "kinetics-400", "array": video}],
                "audio_features": [{"model_key": "layer_vggish", "extractor_name": "VGGish",
                                    "dataset": "YouTube-8M", "array": audio}],
                "filename": fn, "shard_size": rows_s, "shard_name": name,
            })
            meta.append({"filename": fn, "id": "vid%09d" % vid, "segment": [10, 20]})
            vid += 1
        with open(os.path.join(feat_dir, name + ".pkl"), "wb") as f:
            pickle.dump(rows_out, f)
        with open(os.path.join(meta_dir, name + ".json"), "w") as f:
            json.dump(meta, f)
    last = "%06d" % (n_shards - 1)
    return os.path.join(feat_dir, "shard-{000000..%s}.pkl" % last)


def overlapping_rows(seed, n, d, comps, spread, noise=0.3):
    """Rows of the BASELINE-shape assign goldens (gen_golden.py `kmeans_big`): mixture components whose centres are only
    `spread` apart per coordinate (centre distance ~ spread*sqrt(2d)) under noise of radius noise*sqrt(d), so that
    neighbouring clusters OVERLAP and a trained clustering leaves rows near the bisector of two centres.  Regenerated
    from the seed by the generator and by the tests (128 / 256 MB of rows are not committed)."""
    rs = np.random.RandomState(seed)
    cen = (spread * rs.randn(comps, d)).astype(np.float32)
    comp = rs.randint(0, comps, n)
    x = np.empty((n, d), np.float32)
    for s in range(0, n, 4096):  # chunked: the float64 draw of the whole matrix would be 0.5 GB at d = 2048
        e = min(n, s + 4096)
        x[s:e] = cen[comp[s:e]] + (noise * rs.randn(e - s, d)).astype(np.float32)
    return x


def bisector_rows(x, centers, idx, i, j, t):
    """x[idx] + t * (centers[j] - centers[i]) evaluated in fp32 element by element (two IEEE roundings per element:
    the product, then the sum) -- the near-tie rows of the `kmeans_big` goldens, rebuilt from the fixture's (idx, i, j, t)."""
    u = centers[j.astype(np.int64)] - centers[i.astype(np.int64)]
    return (x[idx] + (t.astype(np.float32)[:, None] * u).astype(np.float32)).astype(np.float32)
