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


# ------------------------------------------------------------------ columnar sidecars
# The pkl shard (a list of per-row dicts of small numpy vectors) stays THE contract: it is what the upstream
# feature extractor writes and what the reference reads.  But unpickling + per-row dict walking is the wall-clock
# floor of a run once the kernels are fast (~45 us per row and view), so a shard can carry a columnar twin next to
# it:   <stem>.cols/meta.json + v<i>.npy  (one float32 [rows, d] matrix per view, memory-mapped on load).
# A sidecar is only trusted while the pkl's size and mtime match what meta.json recorded.
SIDECAR_VERSION = 1


def sidecar_mode(mode=None):
    """'off' (never look), 'auto' (use a valid sidecar, never write: default), 'write' (also build missing ones)."""
    mode = (mode or os.environ.get('ACAV_SHARD_SIDECAR', 'auto')).lower()
    if mode not in ('off', 'auto', 'write'):
        raise ValueError("sidecar mode must be off|auto|write, not {!r}".format(mode))
    return mode


def _source_stamp(path):
    st = os.stat(path)
    return {'size': st.st_size, 'mtime_ns': st.st_mtime_ns}


def feature_sidecar_dir(path):
    path = Path(path)
    return path.with_name(path.stem + '.cols')


def _shard_columns_from_rows(rows, stem):
    """per-row dicts -> {'filename','shard_name','shard_size', 'views': OrderedDict[(kind,mk,layer)] -> [n,d], 'tags'}"""
    out = {'filename': [], 'shard_name': [], 'shard_size': [], 'views': OrderedDict(), 'tags': OrderedDict()}
    cols = OrderedDict()
    for row in rows:
        out['filename'].append(row['filename'])
        out['shard_name'].append(row.get('shard_name', stem))
        out['shard_size'].append(row.get('shard_size', len(rows)))
        for kind, key in (('audio', 'audio_features'), ('video', 'video_features')):
            for feat in row.get(key, []):
                mk = feat['model_key']
                out['tags'].setdefault((kind, mk), (feat.get('extractor_name'), feat.get('dataset')))
                for layer, vec in _layers_of(feat['array']):
                    cols.setdefault((kind, mk, layer), []).append(np.asarray(vec, dtype=np.float32))
    n = len(out['filename'])
    for view, vecs in cols.items():
        if len(vecs) != n:
            raise ValueError(f"view {view} is missing in {n - len(vecs)} rows of shard {stem}")
        out['views'][view] = np.stack(vecs, 0) if n else np.zeros((0, 0), np.float32)
    return out


def write_feature_sidecar(path, columns=None):
    """Build <stem>.cols/ for the pkl shard at `path` (atomically: temp dir + rename).  Returns the directory."""
    path = Path(path)
    if columns is None:
        columns = _shard_columns_from_rows(load_pickle(path), path.stem)
    final = feature_sidecar_dir(path)
    tmp = final.with_name(final.name + '.tmp{}'.format(os.getpid()))
    tmp.mkdir(parents=True, exist_ok=True)
    views = []
    for i, (view, mat) in enumerate(columns['views'].items()):
        np.save(tmp / 'v{}.npy'.format(i), np.ascontiguousarray(mat, np.float32))
        views.append(list(view) + [int(mat.shape[1]) if mat.ndim == 2 else 0])
    meta = {'version': SIDECAR_VERSION, 'source': _source_stamp(path), 'rows': len(columns['filename']),
            'filename': list(columns['filename']), 'shard_name': list(columns['shard_name']),
            'shard_size': [int(x) for x in columns['shard_size']], 'views': views,
            'tags': [[k[0], k[1], v[0], v[1]] for k, v in columns['tags'].items()]}
    dump_json(meta, tmp / 'meta.json')
    if final.exists():
        import shutil
        shutil.rmtree(final)
    os.rename(tmp, final)
    return final


def read_feature_sidecar(path):
    """-> columns dict (views memory-mapped) or None when there is no sidecar or it is stale / unreadable."""
    d = feature_sidecar_dir(path)
    try:
        meta = load_json(d / 'meta.json')
        if meta.get('version') != SIDECAR_VERSION or meta.get('source') != _source_stamp(path):
            return None
        # (the shard name of every row as ONE string object, as in an unpickled shard: the assignment files written from
        # these columns are then the same bytes -- pickle memoises by identity -- whichever source the rows came from)
        import sys
        out = {'filename': meta['filename'], 'shard_name': [sys.intern(str(v)) for v in meta['shard_name']],
               'shard_size': meta['shard_size'], 'views': OrderedDict(), 'tags': OrderedDict()}
        for kind, mk, name, dataset in meta['tags']:
            out['tags'][(kind, mk)] = (name, dataset)
        for i, (kind, mk, layer, dim) in enumerate(meta['views']):
            mat = np.load(d / 'v{}.npy'.format(i), mmap_mode='r')
            if mat.dtype != np.float32 or mat.shape != (meta['rows'], dim):
                return None
            out['views'][(kind, mk, layer)] = mat
        return out
    except (OSError, ValueError, KeyError):
        return None


# ---- parallel shard loading (the reference spreads its shard reading over `computation.num_workers` DataLoader workers,
# clustering/code/data/clustering.py:17-66).  Unpickling a shard is ~45 us per row and view of pure Python: one loader thread
# delivers 58 k rows/s against 2-4 M rows/s of GPU training per epoch (profiles/r03_streamed.txt).  Worker PROCESSES read
# the shards and write their rows straight into ONE shared-memory matrix per view -- the table the parent hands to the GPU
# upload -- and send back only the per-row metadata; nothing is pickled, piped or concatenated in the parent.
_POOL = None
_POOL_LOCK = __import__('threading').Lock()


def _pool(workers):
    """The process-wide worker pool, created ONCE (the prefetch thread and the main thread both come here: resizing it
    under a running map() is a race): sized for the larger of this request and the reference's `num_workers` default
    of 40, capped by the cores this rank may use (cores / world: every rank of a node has its own pool)."""
    global _POOL
    with _POOL_LOCK:
        if _POOL is None:
            import atexit
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor
            world = max(1, int(os.environ.get('WORLD_SIZE', '1') or 1))
            cap = max(2, (os.cpu_count() or 2) // world)
            size = max(2, min(max(int(workers), 40), cap))
            ex = ProcessPoolExecutor(max_workers=size, mp_context=mp.get_context('spawn'))  # never fork a process that holds a GPU
            atexit.register(ex.shutdown, wait=False)
            _POOL = (size, ex)
        return _POOL[1]


def _noop(_):
    return os.getpid()


def warm_pool(workers):
    """Start every worker process NOW.  Even `spawn` forks this process for an instant (fork + exec), and a fork of a process
    whose host blocks are registered with the GPU runtime write-protects those pages: every one of 40 forks then makes the
    driver revalidate the registered ranges (measured: 12 s of worker start-up and 18 s of stalled GPU work when the pool
    was first needed at the end of a streamed run, against 1-2 s at its start).  Called before the first block is pinned."""
    if workers and workers > 1:
        ex = _pool(workers)
        for _ in range(_POOL[0]):  # one process per submit while none is idle yet; they finish starting in the background
            ex.submit(_noop, 0)


def _worker_load(job):
    """(path, mode, row offset, expected rows, [(view, d, shm name, total rows)]) -> per-row metadata, or a reason string"""
    from multiprocessing import shared_memory
    path, mode, base, n_expect, views = job
    path = Path(path)
    columns = read_feature_sidecar(path) if mode != 'off' else None
    if columns is None:
        try:
            rows = load_pickle(path)
        except Exception as exc:
            return {'skip': '{}'.format(exc)}  # the parent reports and skips it, as the plain loop does
        columns = _shard_columns_from_rows(rows, path.stem)
        if mode == 'write':
            try:
                write_feature_sidecar(path, columns)
            except OSError:
                pass
    n = len(columns['filename'])
    # an incomplete shard (fewer rows than its metadata says: normal, shard_ok_ratio exists for it) still goes in -- the
    # parent closes the gap; MORE rows than reserved, or another set of views, cannot be placed
    if n > n_expect or {v for v, _, _, _ in views} != {tuple(v) for v in columns['views']}:
        return 'layout'
    for view, d, name, total in views:
        mat = columns['views'][tuple(view)]
        if mat.shape != (n, d):
            return 'layout'
        shm = shared_memory.SharedMemory(name=name)
        try:
            np.ndarray((total, d), np.float32, buffer=shm.buf)[base:base + n] = mat
        finally:
            shm.close()
    return {'filename': list(columns['filename']), 'shard_name': list(columns['shard_name']),
            'shard_size': list(columns['shard_size']), 'tags': list(columns['tags'].items()), 'rows': n}


class _ShmCache:
    """Shared-memory blocks are REUSED from one row group to the next: a fresh block costs a page fault per 4 KB when the
    workers first write it (0.3 s per GB -- more than the unpickling it carries), and a reused block can stay registered
    with the GPU runtime (pinned: the upload runs at the link rate instead of the pageable-copy rate)."""

    def __init__(self, keep=8):
        self.free, self.keep, self.pinned = [], keep, {}

    def acquire(self, size):
        from multiprocessing import shared_memory
        best = None
        for b in self.free:
            if b.size >= size and (best is None or b.size < best.size):
                best = b
        if best is not None:
            self.free.remove(best)
            return best
        b = shared_memory.SharedMemory(create=True, size=max(size + size // 8, 1))  # slack: groups differ by a shard or two
        self._pin(b)
        return b

    def _pin(self, b):
        if os.environ.get('ACAV_PIN_SHM', '1') == '0':
            return
        try:  # only when torch and a GPU are already in the process (the clustering CLI): never imports torch itself
            import sys
            torch = sys.modules.get('torch')
            if torch is None or not torch.cuda.is_available():
                return
            import ctypes
            addr = ctypes.addressof(ctypes.c_char.from_buffer(b.buf))
            if int(torch.cuda.cudart().cudaHostRegister(addr, b.size, 0)) == 0:
                self.pinned[b.name] = addr
        except Exception:
            pass

    def release(self, blocks):
        for b in blocks:
            if len(self.free) < self.keep:
                self.free.append(b)
            else:
                self._drop(b)

    def _drop(self, b):
        try:
            addr = self.pinned.pop(b.name, None)
            if addr is not None:
                import sys
                sys.modules['torch'].cuda.cudart().cudaHostUnregister(addr)
        except Exception:
            pass
        try:
            b.close()
        except Exception:  # BufferError: a view of the block is still alive somewhere -- the name must go all the same
            pass
        try:
            b.unlink()
        except Exception:
            pass

    def clear(self):
        for b in self.free:
            self._drop(b)
        self.free = []


_SHM = _ShmCache()
import atexit as _atexit  # noqa: E402
_atexit.register(_SHM.clear)


def _load_parallel(paths, mode, expect_rows, expect_views, workers):
    """-> FeatureTable whose views live in shared memory, or None (a shard did not have the expected layout: the caller
    reads the group the plain way).  expect_rows: path stem -> rows; expect_views: OrderedDict view -> d."""
    paths = [Path(p) for p in paths]
    try:
        counts = [int(expect_rows[p.stem]) for p in paths]
    except KeyError:
        return None
    total = sum(counts)
    if total == 0:
        return None
    shms = []
    try:
        for view, d in expect_views.items():
            shms.append(_SHM.acquire(max(total * int(d) * 4, 1)))
        views = [(tuple(v), int(d), shm.name, total) for (v, d), shm in zip(expect_views.items(), shms)]
        base, jobs = 0, []
        for p, n in zip(paths, counts):
            jobs.append((str(p), mode, base, n, views))
            base += n
        results = list(_pool(workers).map(_worker_load, jobs, chunksize=max(1, len(jobs) // (workers * 4))))
    except Exception as exc:  # no /dev/shm, worker start-up failure ...: the plain path still works
        print('parallel shard loading unavailable ({}); reading in this process'.format(exc))
        results = None
    if results is None or any(not isinstance(r, dict) for r in results):
        _SHM.release(shms)
        return None
    return _assemble_table(paths, counts, results, expect_views, shms, total)


def _assemble_table(paths, counts, results, expect_views, shms, total):
    """per-shard results (metadata dicts; the rows are already in the shared blocks) -> FeatureTable"""
    import weakref
    table = FeatureTable()
    mats = [np.ndarray((total, int(d)), np.float32, buffer=shm.buf) for (view, d), shm in zip(expect_views.items(), shms)]
    src = dst = 0
    for p, n, r in zip(paths, counts, results):
        if 'skip' in r:  # unreadable: reported and skipped (clustering data/clustering.py:167-182), its slice is closed up
            print(r['skip'])
            print('Exception in shard loading: {}'.format(p.stem))
            src += n
            continue
        have = int(r['rows'])
        if dst != src:  # a shorter / skipped shard before this one: move this shard's rows down (dst < src: no overlap hazard)
            for m in mats:
                m[dst:dst + have] = m[src:src + have]
        table.filename.extend(r['filename'])
        table.shard_name.extend(r['shard_name'])
        table.shard_size.extend(r['shard_size'])
        for key, tag in r['tags']:
            table.tags.setdefault(tuple(key), tuple(tag))
        table.shard_rows[p.stem] = list(range(dst, dst + have))
        dst += have
        src += n
    for (view, d), m in zip(expect_views.items(), mats):
        table.views[view] = m[:dst]

    table._shm = shms
    weakref.finalize(table, _SHM.release, shms)  # back to the cache when the table goes away
    return table


# ---- native shard reading (round 4).  The worker processes above still pay pickle.load: ~10 Python objects per clip and
# view, then np.stack and the copy into the shared block -- 24 ms per 1000-clip shard in each of 40 processes on the GPU box
# (0.5-0.9 M rows/s per pass).  libacav_hip's acav_pkl_* (csrc/acav_shardio.hip) read the file, walk the pickle opcodes once
# and copy every vector straight into its row of the table: 3-4 ms per shard, a whole GROUP of shards per call on the
# library's own threads (no interpreter lock, no spawn, no pipes, no result pickling: Python threads calling a per-shard
# entry point convoyed on the GIL and on the process-wide mm lock of their mappings).  A shard outside the reader's subset is
# read with pickle.load; the table is the same either way (tests/test_shards_io.py).
_KINDS = ('audio', 'video')


def _native_lib():
    if os.environ.get('ACAV_SHARD_NATIVE', '1') == '0':
        return None
    try:
        from . import _lib
        return _lib.load_library()
    except Exception:  # library not built: the worker processes still work
        return None


def _dec(b):
    return None if b is None else b.decode('utf-8', 'surrogatepass')


def _native_views(lib, h, nviews):
    import ctypes as C
    specs, tags = [], OrderedDict()
    for v in range(nviews):
        kind, mk, layer, ex, ds, d = C.c_int(), C.c_char_p(), C.c_char_p(), C.c_char_p(), C.c_char_p(), C.c_int64()
        lib.acav_pkl_shard_view(h, v, C.byref(kind), C.byref(mk), C.byref(layer), C.byref(ex), C.byref(ds), C.byref(d))
        key = (_KINDS[kind.value], _dec(mk.value), _dec(layer.value))
        tags.setdefault(key[:2], (_dec(ex.value), _dec(ds.value)))
        specs.append((v, key, d.value))
    return specs, tags


def _native_meta(lib, h, n, n_names, stem):
    """-> filename, shard_name, shard_size lists of the handle's rows, as _shard_columns_from_rows builds them"""
    import ctypes as C
    if not n:
        return [], [], []
    fn, fl, nm, nl = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
    ids, of_row, ssz = C.c_void_p(), C.c_void_p(), C.c_void_p()
    lib.acav_pkl_shard_meta(h, C.byref(fn), C.byref(fl), C.byref(nm), C.byref(nl), C.byref(ids), C.byref(of_row), C.byref(ssz))
    filename = C.string_at(fn, fl.value).decode('utf-8', 'surrogatepass').split('\n')
    # ONE str object per shard_name object of the pickle (pickle.load gives the rows of a shard the same object; the
    # assignment shards written from this table memoise by identity)
    names = C.string_at(nm, nl.value).decode('utf-8', 'surrogatepass').split('\n') if n_names else []
    by_id = dict(zip(np.frombuffer((C.c_int64 * n_names).from_address(ids.value), np.int64).tolist(), names)) if n_names else {}
    by_id[-1] = stem
    shard_name = [by_id[i] for i in np.frombuffer((C.c_int64 * n).from_address(of_row.value), np.int64).tolist()]
    sizes = np.frombuffer((C.c_int64 * n).from_address(ssz.value), np.int64)
    shard_size = np.where(sizes == np.iinfo(np.int64).min, n, sizes).tolist()
    return filename, shard_name, shard_size


def read_shard_native(lib, path, dest=None, base=0, n_expect=None):
    """One pkl shard through acav_pkl_shard_*: -> columns dict as _shard_columns_from_rows builds it, or None when the
    shard is outside the native reader's subset.  dest: {view: [total, d] matrix} -- the vectors then go straight into
    rows [base, base + n) (and columns['views'] holds those slices); 'layout' when they do not fit."""
    import ctypes as C
    h = C.c_void_p()
    if lib.acav_pkl_shard_open(os.fsencode(str(path)), C.byref(h)) != 0:
        return None
    try:
        rows, nv, nn = C.c_int64(), C.c_int(), C.c_int()
        lib.acav_pkl_shard_info(h, C.byref(rows), C.byref(nv), C.byref(nn))
        n = rows.value
        out = {'views': OrderedDict()}
        specs, out['tags'] = _native_views(lib, h, nv.value)
        if dest is not None:
            if (n_expect is not None and n > n_expect) or {k for _, k, _ in specs} != set(dest) or \
                    any(dest[k].shape[1] != d for _, k, d in specs):
                return 'layout'
        for v, key, d in specs:
            mat = dest[key][base:base + n] if dest is not None else np.empty((n, d), np.float32)
            if n:
                lib.acav_pkl_shard_copy_view(h, v, mat.ctypes.data_as(C.c_void_p), mat.strides[0] // 4)
            out['views'][key] = mat
        out['filename'], out['shard_name'], out['shard_size'] = _native_meta(lib, h, n, nn.value, Path(path).stem)
        return out
    finally:
        lib.acav_pkl_shard_close(h)


def _load_native(paths, expect_rows, expect_views, workers):
    """_load_parallel through acav_pkl_load_group (one call per group, the library's own threads) instead of worker
    processes + pickle.load; same table, same fallbacks.  pkl shards only: sidecars stay with the worker processes."""
    import ctypes as C
    lib = _native_lib()
    if lib is None:
        return None
    paths = [Path(p) for p in paths]
    try:
        counts = [int(expect_rows[p.stem]) for p in paths]
    except KeyError:
        return None
    total, n, nv = sum(counts), len(paths), len(expect_views)
    if total == 0:
        return None
    world = max(1, int(os.environ.get('WORLD_SIZE', '1') or 1))
    threads = max(1, min(max(int(workers), 8), max(2, (os.cpu_count() or 2) // world), 64))
    threads = int(os.environ.get('ACAV_SHARD_THREADS', threads))  # experiments (tools/exp/loader_probe.py)
    shms = [_SHM.acquire(max(total * int(d) * 4, 1)) for d in expect_views.values()]
    mats = [np.ndarray((total, int(d)), np.float32, buffer=shm.buf) for d, shm in zip(expect_views.values(), shms)]
    keys = [tuple(v) for v in expect_views]
    c_paths = (C.c_char_p * n)(*[os.fsencode(str(p)) for p in paths])
    c_base = (C.c_int64 * n)(*np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64).tolist())
    c_exp = (C.c_int64 * n)(*counts)
    c_kind = (C.c_int * nv)(*[_KINDS.index(k[0]) for k in keys])
    c_mk = (C.c_char_p * nv)(*[k[1].encode('utf-8', 'surrogatepass') for k in keys])
    c_layer = (C.c_char_p * nv)(*[k[2].encode('utf-8', 'surrogatepass') for k in keys])
    c_dims = (C.c_int64 * nv)(*[int(d) for d in expect_views.values()])
    c_dest = (C.c_void_p * nv)(*[m.ctypes.data for m in mats])
    handles, status = (C.c_void_p * n)(), (C.c_int * n)()
    import time
    t0 = time.perf_counter()
    rc = lib.acav_pkl_load_group(c_paths, n, c_base, c_exp, nv, c_kind, c_mk, c_layer, c_dims, c_dest, threads, handles, status)
    t1 = time.perf_counter()
    results = []
    try:
        if rc != 0:
            results = None
        for i, p in enumerate(paths if results is not None else []):
            if status[i] == 2:
                results = None
                break
            if status[i] == 0:
                h = C.c_void_p(handles[i])
                rows, nvs, nn = C.c_int64(), C.c_int(), C.c_int()
                lib.acav_pkl_shard_info(h, C.byref(rows), C.byref(nvs), C.byref(nn))
                _, tags = _native_views(lib, h, nvs.value)
                fnm, snm, ssz = _native_meta(lib, h, rows.value, nn.value, p.stem)
                results.append({'filename': fnm, 'shard_name': snm, 'shard_size': ssz, 'tags': list(tags.items()), 'rows': rows.value})
                continue
            try:  # outside the native subset (or unreadable): the general reader
                shard_rows = load_pickle(p)
            except Exception as exc:  # an unreadable FILE is reported and skipped (data/clustering.py:167-182) ...
                results.append({'skip': '{}'.format(exc)})
                continue
            # ... inconsistent CONTENT (a view missing in some rows) is not: the ValueError propagates exactly as it does
            # from the plain loop of load_feature_shards and from the worker processes, whichever loader ran
            columns = _shard_columns_from_rows(shard_rows, p.stem)
            k = len(columns['filename'])
            if k > counts[i] or set(columns['views']) != set(keys) or \
                    any(columns['views'][key].shape != (k, m.shape[1]) for key, m in zip(keys, mats)):
                results = None
                break
            for key, m in zip(keys, mats):
                m[c_base[i]:c_base[i] + k] = columns['views'][key]
            results.append({'filename': columns['filename'], 'shard_name': columns['shard_name'], 'shard_size': columns['shard_size'],
                            'tags': list(columns['tags'].items()), 'rows': k})
    except Exception:  # corrupt content: the blocks go back before the error leaves (the caller does not see them)
        del mats
        _SHM.release(shms)
        raise
    finally:
        for i in range(n):
            if handles[i]:
                lib.acav_pkl_shard_close(C.c_void_p(handles[i]))
    mats = None
    if results is None:
        _SHM.release(shms)
        return None
    table = _assemble_table(paths, counts, results, expect_views, shms, total)
    if os.environ.get('ACAV_SHARD_TIMING'):
        t2 = time.perf_counter()
        print('[acav] native shard group: {} shards, {} rows, {} threads: read + parse + copy {:.1f} ms, row metadata + table {:.1f} ms'.format(
            n, len(table), threads, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
    return table


def load_feature_shards(paths, model_order=None, audio_models=(), sidecar=None, workers=0, expect_rows=None, expect_views=None):
    """Read shards in the given order (training order of a single-stream loader: sorted shards,
    rows in file order -- clustering data/clustering.py:153-186 with num_workers=0).
    Corrupt shards are reported and skipped like the reference does (:167-182).
    sidecar: see sidecar_mode(); a valid columnar twin of a shard replaces its unpickling.
    workers > 1 with expect_rows (stem -> rows, from the metadata) and expect_views (view -> d, in table order, from a
    probe shard): the shards are read by that many worker processes into shared memory (above); any surprise -- a short
    or unreadable shard, another view set -- falls back to the plain loop below, which reports and skips as ever."""
    mode = sidecar_mode(sidecar)
    paths = list(paths)
    if workers and workers > 1 and expect_rows is not None and expect_views:
        table = None
        if mode == 'off' or (mode == 'auto' and not any(feature_sidecar_dir(p).exists() for p in paths)):
            table = _load_native(paths, expect_rows, expect_views, int(workers))  # the library's reader and threads (above)
        if table is None and len(paths) >= 16:  # worker processes + pickle.load / sidecars (start-up ~1 s)
            table = _load_parallel(paths, mode, expect_rows, expect_views, min(int(workers), len(paths)))
        if table is not None:
            return table
    table = FeatureTable()
    parts = OrderedDict()
    lib = _native_lib()
    for path in paths:
        path = Path(path)
        columns = read_feature_sidecar(path) if mode != 'off' else None
        if columns is None:
            columns = read_shard_native(lib, path) if lib is not None else None
            if columns is None:  # outside the native reader's subset (or no library): the general reader
                try:
                    rows = load_pickle(path)
                except Exception as exc:  # EOFError and friends
                    print(exc)
                    print('Exception in shard loading: {}'.format(path.stem))
                    continue
                columns = _shard_columns_from_rows(rows, path.stem)
            if mode == 'write':
                try:
                    write_feature_sidecar(path, columns)
                except OSError as exc:  # read-only data directory: the pkl path still works
                    print('could not write the sidecar of {}: {}'.format(path.stem, exc))
        base = len(table.filename)
        n = len(columns['filename'])
        table.filename.extend(columns['filename'])
        table.shard_name.extend(columns['shard_name'])
        table.shard_size.extend(columns['shard_size'])
        for key, tag in columns['tags'].items():
            table.tags.setdefault(key, tag)
        for view, mat in columns['views'].items():
            parts.setdefault(view, []).append((base, mat))
        table.shard_rows[path.stem] = list(range(base, base + n))
    n = len(table)
    order = sorted(parts, key=lambda v: _view_rank(v, model_order, audio_models))
    for view in order:
        have = sum(m.shape[0] for _, m in parts[view])
        if have != n:
            raise ValueError(f"view {view} is missing in {n - have} rows")
        table.views[view] = np.concatenate([m for _, m in parts[view]], 0) if n else np.zeros((0, 0), np.float32)
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
    row_ids = list(row_ids)
    return _assignment_rows_from_columns(
        [table.filename[r] for r in row_ids], [table.shard_size[r] for r in row_ids], [table.shard_name[r] for r in row_ids],
        list(table.tags.items()), [(view, np.asarray(lab)[row_ids]) for view, lab in labels.items()])


def _assignment_rows_from_columns(filenames, shard_sizes, shard_names, tags, label_columns):
    """the same from plain columns (what a writer process receives): label_columns = [((kind, mk, layer), int64 [n])]"""
    tags = {tuple(k): tuple(v) for k, v in tags}
    by_model = OrderedDict()
    for (kind, mk, layer), col in label_columns:
        by_model.setdefault((kind, mk), []).append((layer, col))
    out = []
    for i in range(len(filenames)):
        row = {'video_assignments': [], 'audio_assignments': []}
        for (kind, mk), layers in by_model.items():
            name, dataset = tags[(kind, mk)]
            entry = {'model_key': mk, 'extractor_name': name, 'dataset': dataset,
                     'array': {layer: np.int64(col[i]) for layer, col in layers}}
            row[f'{kind}_assignments'].append(entry)
        row['filename'] = filenames[i]
        row['shard_size'] = shard_sizes[i]
        row['shard_name'] = shard_names[i]
        out.append(row)
    return out


def _worker_write_assignments(job):
    out_path, filenames, shard_sizes, shard_names, tags, label_columns, sidecar = job
    rows = _assignment_rows_from_columns(filenames, shard_sizes, shard_names, tags, label_columns)
    dump_pickle(rows, out_path)
    if sidecar:
        write_assignment_sidecar(out_path, rows)
    return str(out_path)


class AssignmentWriter:
    """Writes assignment shards ({out}/{shard}.pkl, the reference's per-row dict schema).  Building a million small dicts
    and pickling them is ~10 us per row of pure Python -- more than the GPU spends on the whole clustering -- so with
    workers > 1 the shards are written by the loader's worker processes while the caller goes on labelling."""

    def __init__(self, workers=0):
        self.workers, self.pending = int(workers or 0), []

    def submit(self, table, labels, ids, out_path, sidecar=False):
        ids = list(ids)
        job = (str(out_path), [table.filename[r] for r in ids], [table.shard_size[r] for r in ids],
               [table.shard_name[r] for r in ids], list(table.tags.items()),
               [(view, np.ascontiguousarray(np.asarray(lab)[ids], np.int64)) for view, lab in labels.items()], bool(sidecar))
        if self.workers > 1:
            self.pending.append(_pool(self.workers).submit(_worker_write_assignments, job))
        else:
            _worker_write_assignments(job)

    def finish(self):
        for f in self.pending:
            f.result()
        self.pending = []


# --------------------------------------------------------------------------- assignment shards
def assignment_sidecar_path(path):
    path = Path(path)
    return path.with_name(path.stem + '.assign.npz')


def write_assignment_sidecar(path, rows=None):
    """<stem>.assign.npz next to an assignment pkl: labels int64 [rows, D] + types + names (uncompressed)."""
    path = Path(path)
    if rows is None:
        rows = load_pickle(path)
    per_row, shard_names, filenames = _assignment_rows_to_lists(rows)
    mat, types = rows_to_matrix(per_row)
    stamp = _source_stamp(path)
    tmp = assignment_sidecar_path(path).with_suffix('.tmp{}.npz'.format(os.getpid()))
    np.savez(tmp, labels=mat, types=np.array(json.dumps([list(t) for t in types])),
             shard_name=np.array(shard_names), filename=np.array(filenames),
             source=np.array([SIDECAR_VERSION, stamp['size'], stamp['mtime_ns']], np.int64))
    os.replace(tmp, assignment_sidecar_path(path))
    return assignment_sidecar_path(path)


def read_assignment_sidecar(path):
    try:
        stamp = _source_stamp(path)
        with np.load(assignment_sidecar_path(path), allow_pickle=False) as z:
            if z['source'].tolist() != [SIDECAR_VERSION, stamp['size'], stamp['mtime_ns']]:
                return None
            types = [tuple(t) for t in json.loads(str(z['types']))]
            return z['labels'].astype(np.int64), types, z['shard_name'].tolist(), z['filename'].tolist()
    except (OSError, ValueError, KeyError):
        return None


def _assignment_rows_to_lists(rows):
    per_row, shard_names, filenames = [], [], []
    for row in rows:
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
    return per_row, shard_names, filenames


def _assignment_shards_native(paths):
    """{path index: (labels int64 [rows, D] in sorted-type order, types, shard_names, filenames)} for the shards the library's
    reader covers (acav_pkl_assign_load_group: all of them parsed in one call on the library's threads; 3.4 ms -> 0.1 ms per
    1000-row shard); the others are left to pickle.load."""
    import ctypes as C
    lib = _native_lib()
    paths = list(paths)
    if lib is None or not paths:
        return {}
    n = len(paths)
    world = max(1, int(os.environ.get('WORLD_SIZE', '1') or 1))
    threads = max(1, min(32, max(2, (os.cpu_count() or 2) // world), n))
    c_paths = (C.c_char_p * n)(*[os.fsencode(str(p)) for p in paths])
    handles, status = (C.c_void_p * n)(), (C.c_int * n)()
    out = {}
    if lib.acav_pkl_assign_load_group(c_paths, n, threads, handles, status) != 0:
        return {}
    try:
        for i in range(n):
            if status[i] != 0:
                continue
            h = C.c_void_p(handles[i])
            rows, nv, nn = C.c_int64(), C.c_int(), C.c_int()
            lib.acav_pkl_shard_info(h, C.byref(rows), C.byref(nv), C.byref(nn))
            specs, _ = _native_views(lib, h, nv.value)
            keys = [(k[1], k[2]) for _, k, _ in specs]  # (model key, layer): the reference's clustering type
            types = sorted(keys)
            filename, shard_name, _ = _native_meta(lib, h, rows.value, nn.value, Path(paths[i]).stem)
            lab = C.c_void_p()
            lib.acav_pkl_shard_labels(h, C.byref(lab))
            if rows.value and nv.value:
                mat = np.frombuffer((C.c_int64 * (rows.value * nv.value)).from_address(lab.value), np.int64).reshape(rows.value, nv.value)
                mat = np.ascontiguousarray(mat[:, [keys.index(t) for t in types]])
            else:
                mat = np.zeros((rows.value, nv.value), np.int64)
            out[i] = (mat, types, shard_name, filename)
    finally:
        for i in range(n):
            if handles[i]:
                lib.acav_pkl_shard_close(C.c_void_p(handles[i]))
    return out


def load_assignment_shards(paths, sidecar=None):
    """-> (assignments int64 [V,D], clustering_types, shard_names[V], filenames[V])  --
    dataloader.format_row / format_assignments / preprocess (subset dataloader.py:17-69):
    clustering_types = sorted (model_key, layer) tuples; only dict/list 'array's are supported by the
    reference (its scalar branch references an undefined name), so is the same here.
    A valid <stem>.assign.npz (written by our clustering stage next to each pkl) replaces the per-row parsing."""
    mode = sidecar_mode(sidecar)
    mats, types, shard_names, filenames = [], None, [], []
    paths = list(paths)
    side = {i: read_assignment_sidecar(p) for i, p in enumerate(paths)} if mode != 'off' else {}
    todo = [i for i in range(len(paths)) if side.get(i) is None]
    native = dict(zip(todo, [None] * len(todo)))
    if mode != 'write':  # (writing the twins needs the rows themselves)
        native = {todo[j]: v for j, v in _assignment_shards_native([paths[i] for i in todo]).items()}
    for i, path in enumerate(paths):
        got = side.get(i)
        if got is None:
            got = native.get(i)
        if got is None:
            rows = load_pickle(path)
            per_row, sn, fn = _assignment_rows_to_lists(rows)
            mat, ty = rows_to_matrix(per_row)
            if per_row and any(sorted(r.keys()) != ty for r in per_row):
                raise KeyError("rows of {} do not share one set of clusterings".format(Path(path).stem))
            if mode == 'write':
                try:
                    write_assignment_sidecar(path, rows)
                except OSError as exc:
                    print('could not write the sidecar of {}: {}'.format(Path(path).stem, exc))
        else:
            mat, ty, sn, fn = got
        if len(sn) == 0:
            continue
        if types is None:
            types = ty
        elif ty != types:
            raise KeyError("shard {} holds clusterings {} instead of {}".format(Path(path).stem, ty, types))
        mats.append(mat)
        shard_names.extend(sn)
        filenames.extend(fn)
    types = types or []
    mat = np.concatenate(mats, 0) if mats else np.zeros((0, len(types)), np.int64)
    return mat, types, shard_names, filenames


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


def shard_sizes_from_meta(shard_paths, meta_path, use_cache=False):
    """{shard: number of clips} from the {shard}.json files; shards without one are dropped from the
    run (clustering data/meta.py:29-57, data/shards.py:32-34)."""
    out = OrderedDict()
    cached = {}
    if use_cache and meta_path is not None:
        # the reference trusts <meta dir>/meta_cache.pkl for every shard it lists and reads the json files of the others only
        # (data/meta.py:11-20); parsing 1000 of them was 0.5 s of a 2.3 s streamed run
        try:
            cached = dict(load_pickle(Path(meta_path) / 'meta_cache.pkl'))
        except Exception:
            cached = {}
    for p in shard_paths:
        stem = Path(p).stem
        if stem in cached:
            out[stem] = int(cached[stem])
            continue
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
