"""Shard / metadata / csv I/O of the curation path -- the reference's file contract, columnar inside.

Formats (probed from the reference):
  feature shard   {name}.pkl = list of rows {'video_features': [ {model_key, extractor_name, dataset,
                  'array': {'layer_i': float32[d_i]} | [..] | ndarray} ], 'audio_features': [...],
                  'filename', 'shard_size', 'shard_name'}          (feature_extraction save.py:48-74)
  assignment shard same rows with '*_assignments' and np.int64 labels (clustering/code/save.py:48-74)
  run manifest    log_{host}_{pid}_{ts}.json = {hostname,pid,timestamp,time,'shards': [...]}
                  (clustering/code/save.py:9-17) -- groups shards into partitions for the selection
  metadata        {shard}.json = list of {'filename','id','segment',...}
  output.csv      rows shard_name,filename,id,segment appended  (subset_selection/code/save.py:6-44)

The per-row Python dicts only exist at the file boundary; between the files and the GPU the data is
columnar: one float32 [N,d] matrix per (model, layer) view, one int64 [N,D] label matrix.
"""
import csv
import datetime
import itertools
import json
import os
import pickle
import platform
import re
import time
from collections import OrderedDict
from pathlib import Path

import numpy as np


# ------------------------------------------------------------------------------ small helpers
def brace_expand(pattern):
    """'{000..003}' numeric ranges (zero padded) and '{a,b}' lists, nested/multiple groups."""
    pattern = str(pattern)
    m = re.search(r'\{([^{}]*)\}', pattern)
    if not m:
        return [pattern]
    body, out = m.group(1), []
    rng = re.fullmatch(r'(-?\d+)\.\.(-?\d+)', body)
    if rng:
        lo, hi = rng.group(1), rng.group(2)
        width = max(len(lo), len(hi)) if (lo.startswith('0') or hi.startswith('0')) and len(lo) == len(hi) else 0
        step = 1 if int(hi) >= int(lo) else -1
        alts = [str(v).zfill(width) for v in range(int(lo), int(hi) + step, step)]
    else:
        alts = body.split(',')
    for alt in alts:
        out += brace_expand(pattern[:m.start()] + alt + pattern[m.end():])
    return out


def to_brace(names):
    names = list(names)
    if not names:
        return ''
    return names[0] if len(names) == 1 else '{' + ','.join(names) + '}'


def load_pickle(path):
    with open(str(path), 'rb') as f:
        return pickle.load(f)


def dump_pickle(obj, path):
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    with open(str(path), 'wb') as f:
        pickle.dump(obj, f)


def load_json(path):
    with open(str(path), 'r') as f:
        return json.load(f)


def dump_json(obj, path, indent=None):
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    with open(str(path), 'w') as f:
        json.dump(obj, f, indent=indent)


def run_info():
    """clustering/code/utils.py:41-55: identifies one clustering run (and names its manifest)."""
    return {'hostname': platform.uname()[1], 'pid': os.getpid(), 'timestamp': int(time.time()),
            'time': str(datetime.datetime.now())}


def run_id(info):
    return '_'.join(str(info[k]) for k in ('hostname', 'pid', 'timestamp'))


# ------------------------------------------------------------------------------ feature shards
def _layers_of(array):
    """feature 'array' -> ordered [(layer_name, vector)] (clustering data/clustering.py:69-105)."""
    if isinstance(array, dict):
        return list(array.items())
    if isinstance(array, (list, tuple)):
        return [(f'layer_{i}', v) for i, v in enumerate(array)]
    return [('model', array)]


class FeatureTable:
    """Columnar view of a list of feature shards.

    views: OrderedDict[(kind, model_key, layer)] -> float32 [N, d]   (kind = 'audio' | 'video')
    tags:  (kind, model_key) -> (extractor_name, dataset)
    rows:  filename[N], shard_name[N], shard_size[N]; shard_rows: shard -> list of row ids
    """

    def __init__(self):
        self.views = OrderedDict()
        self.tags = OrderedDict()
        self.filename, self.shard_name, self.shard_size = [], [], []
        self.shard_rows = OrderedDict()

    def __len__(self):
        return len(self.filename)


def load_feature_shards(paths, model_order=None, audio_models=()):
    """Read shards in the given order (training order of a single-stream loader: sorted shards,
    rows in file order -- clustering data/clustering.py:153-186 with num_workers=0).
    Corrupt shards are reported and skipped like the reference does (:167-182)."""
    table = FeatureTable()
    cols = OrderedDict()
    for path in paths:
        path = Path(path)
        try:
            rows = load_pickle(path)
        except Exception as exc:  # EOFError and friends
            print(exc)
            print('Exception in shard loading: {}'.format(path.stem))
            continue
        ids = []
        for row in rows:
            ids.append(len(table.filename))
            table.filename.append(row['filename'])
            table.shard_name.append(row.get('shard_name', path.stem))
            table.shard_size.append(row.get('shard_size', len(rows)))
            for kind, key in (('audio', 'audio_features'), ('video', 'video_features')):
                for feat in row.get(key, []):
                    mk = feat['model_key']
                    table.tags.setdefault((kind, mk), (feat.get('extractor_name'), feat.get('dataset')))
                    for layer, vec in _layers_of(feat['array']):
                        cols.setdefault((kind, mk, layer), []).append(np.asarray(vec, dtype=np.float32))
        table.shard_rows[path.stem] = ids
    n = len(table)
    order = sorted(cols, key=lambda v: _view_rank(v, model_order, audio_models))
    for view in order:
        if len(cols[view]) != n:
            raise ValueError(f"view {view} is missing in {n - len(cols[view])} rows")
        table.views[view] = np.stack(cols[view], 0) if n else np.zeros((0, 0), np.float32)
    return table


def _view_rank(view, model_order, audio_models):
    """The order in which the reference builds its KMeans objects (and consumes the RNG):
    args.models order, then layer index (run_clustering.py:32-44)."""
    kind, mk, layer = view
    if model_order and mk in model_order:
        m = model_order.index(mk)
    else:
        m = (0 if kind == 'audio' or mk in audio_models else 1) + (len(model_order) if model_order else 0)
    num = re.findall(r'\d+', layer)
    return (m, mk, int(num[-1]) if num else -1, layer)


def assignment_rows(table, labels, row_ids):
    """labels: {view: int64 [N]} -> the reference's per-row dict schema (clustering save.py:48-74)."""
    out = []
    by_model = OrderedDict()
    for (kind, mk, layer) in labels:
        by_model.setdefault((kind, mk), []).append(layer)
    for r in row_ids:
        row = {'video_assignments': [], 'audio_assignments': []}
        for (kind, mk), layers in by_model.items():
            name, dataset = table.tags[(kind, mk)]
            entry = {'model_key': mk, 'extractor_name': name, 'dataset': dataset,
                     'array': {layer: np.int64(labels[(kind, mk, layer)][r]) for layer in layers}}
            row[f'{kind}_assignments'].append(entry)
        row['filename'] = table.filename[r]
        row['shard_size'] = table.shard_size[r]
        row['shard_name'] = table.shard_name[r]
        out.append(row)
    return out


# --------------------------------------------------------------------------- assignment shards
def load_assignment_shards(paths):
    """-> (assignments int64 [V,D], shard_names[V], filenames[V], clustering_types)  --
    dataloader.format_row / format_assignments / preprocess (subset dataloader.py:17-69):
    clustering_types = sorted (model_key, layer) tuples; only dict/list 'array's are supported by the
    reference (its scalar branch references an undefined name), so is the same here."""
    per_row, shard_names, filenames = [], [], []
    for path in paths:
        for row in load_pickle(path):
            res = {}
            for key in ('audio_assignments', 'video_assignments'):
                for feat in row.get(key, []):
                    for layer, val in _layers_of(feat['array']):
                        if layer == 'model':
                            raise ValueError("scalar assignment arrays are not supported (dataloader.py:27-35)")
                        res[(feat['model_key'], layer)] = int(val)
            per_row.append(res)
            shard_names.append(row['shard_name'])
            filenames.append(row['filename'])
    return rows_to_matrix(per_row) + (shard_names, filenames)


def rows_to_matrix(per_row):
    types = sorted(per_row[0].keys()) if per_row else []
    mat = np.array([[r[t] for t in types] for r in per_row], dtype=np.int64).reshape(len(per_row), len(types))
    return mat, types


def load_partitions(shards_dir):
    """shard -> partition id from the run manifests, newer logs win (subset dataloader.py:72-83)."""
    logs = sorted(Path(shards_dir).glob('log_*.json'), key=lambda p: str(p).split('.')[-2].split('_')[-1])
    parts = {}
    for i, log in enumerate(logs):
        for shard in load_json(log)['shards']:
            parts[shard] = i
    return parts


def load_metas(shard_paths, metas_path):
    """{shard: {file stem: meta row}} (subset dataloader.py:206-255)"""
    metas = {}
    for p in shard_paths:
        mp = Path(metas_path) / '{}.json'.format(Path(p).stem)
        if mp.is_file():
            metas[Path(p).stem] = {Path(r['filename']).stem: r for r in load_json(mp)}
    return metas


def shard_sizes_from_meta(shard_paths, meta_path):
    """{shard: number of clips} from the {shard}.json files; shards without one are dropped from the
    run (clustering data/meta.py:29-57, data/shards.py:32-34)."""
    out = OrderedDict()
    for p in shard_paths:
        stem = Path(p).stem
        mp = Path(meta_path) / f'{stem}.json' if meta_path is not None else Path(p).parent / f'{stem}.json'
        if mp.is_file():
            out[stem] = len(load_json(mp))
    return out


# --------------------------------------------------------------------------------- output.csv
def append_output_csv(data, metas, out_path, name='', sharded_meta=True):
    """subset_selection/code/save.py:6-44: join the selected rows with their metadata and APPEND
    'shard_name,filename,id,segment' lines.  Returns (path, lines written)."""
    out_path = Path(out_path)
    out_path.parent.mkdir(exist_ok=True, parents=True)
    joined = OrderedDict()
    order = []
    for row in data:
        stem = Path(row['filename']).stem
        meta = None
        if sharded_meta:
            meta = metas.get(row['shard_name'], {}).get(stem)
        else:
            meta = metas.get(stem)
        if meta is None:
            meta = {'id': '-1', 'segment': [-1.0, -1.0]}
        joined[stem] = {**row, **meta}
        order.append(stem)
    out_path = out_path.parent / (name + out_path.name)
    count = 0
    with open(out_path, 'a+', newline='') as f:
        writer = csv.writer(f)
        for stem in order:
            r = joined[stem]
            writer.writerow([r[h] for h in ('shard_name', 'filename', 'id', 'segment')])
            count += 1
    return out_path, count


def merge_csvs(ins, out):
    """save.py:88-96"""
    count = 0
    with open(out, 'a+') as out_f:
        for src in sorted(ins):
            with open(src, 'r') as in_f:
                for line in in_f:
                    out_f.write(line)
                    count += 1
    return count


def chunked(items, size):
    it = iter(items)
    while True:
        block = list(itertools.islice(it, size))
        if not block:
            return
        yield block
