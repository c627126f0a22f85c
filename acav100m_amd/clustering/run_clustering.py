"""Clustering stage: train_clusters + assign_clusters (clustering/code/run_clustering.py:25-272),
restructured for a 288 GB-HBM GPU: the feature shards are read once into resident [N,d] matrices
(one per (model, layer) view), every KMeans trains on its matrix with the bulk C-ABI call (no host
synchronisation per step), then one assign sweep per view labels all rows; the reference's per-row
dict / pkl schema only exists at the file boundary (acav100m_amd/shards.py).

Same observable behaviour as the reference:
  * KMeans objects are created in args.models order, layer by layer (RNG order of run_clustering.py:32-44)
  * the training batches are the ones the reference's DataLoader delivers for the configured computation.num_workers
    (default 40: whole batches round-robin over worker shard subsets, data/clustering.py:17-66,212-228; 0: one stream),
    every stream cut / cycled to get_length() samples (parallel/row_plan.py: loader_stream; data.loader_order / loader_tail)
  * per batch, every clustering takes one add() -- the warm-up draws are
    interleaved across clusterings exactly like the reference's per-batch loop (:229-241)
  * lr = 0.1 ** (2 + epoch // 5), drop_last batches; several GPUs: clustering.multi_gpu (config.py) -- `views` keeps the
    one-GPU batch stream and epoch count, `reference` is the reference's own N-GPU stream with ceil(epochs / num_gpus) epochs
  * cache_epoch_{e}_{name} checkpoints, skip of already written shards, log_*.json manifest
"""
import math
import pickle
from collections import OrderedDict
from pathlib import Path

import numpy as np

from .. import shards as io
from ..parallel import world
from .sgd_clustering import KMeans


def _device(args):
    import torch
    return f"cuda:{torch.cuda.current_device()}"


def _cache_path(args, epoch):
    return Path(args.data.output.path) / "cache_epoch_{}_{}".format(epoch, Path(args.data.path).name)


def init_clusterings(args, table):
    """run_clustering.py:32-52 -- one KMeans per view, created in view order (RNG order)."""
    cl = OrderedDict()
    for view, mat in table.views.items():
        cl[view] = KMeans(args, mat.shape[1], args.clustering.ncentroids)
    return _to_device(args, cl)


def _to_device(args, cl):
    dev = _device(args)
    for km in cl.values():
        km.to(dev)
        km.initialize()
    return cl


def load_clusterings(args, table):
    """run_clustering.py:55-107: resume from cache_epoch_{e}_{name} (or a cache trained on a subset of
    these shards) when --clustering.cached_epoch is set."""
    epoch = args.clustering.cached_epoch
    if isinstance(epoch, int):
        path = _cache_path(args, epoch)
        if not path.is_file() and args.clustering.load_cache_from_shard_subset:
            path = _subset_cache(args, epoch)
        if path is not None and path.is_file():
            saved = _read_cache(path)
            if saved is not None:  # a reference-written file does not name the modality: match on (model_key, layer)
                by_name = {k[1:]: v for k, v in saved.items()}
                saved = {v: saved.get(v, by_name.get(v[1:])) for v in table.views}
                saved = saved if all(x is not None for x in saved.values()) else {}
            if saved is not None and set(saved.keys()) >= set(table.views.keys()):
                print("loading from clustering cache: {}".format(path))
                cl = OrderedDict((v, KMeans.load(saved[v])) for v in table.views)
                for km in cl.values():
                    km.args = args
                return _to_device(args, cl), True
            if saved is not None:
                print("clustering cache features does not match with the given models")
        print("no clustering cache found.")
    return init_clusterings(args, table), False


def _subset_cache(args, epoch):
    want = set(io.brace_expand(Path(args.data.path).name))
    best = None
    for p in Path(args.data.output.path).glob("cache_epoch_{}_*".format(epoch)):
        have = set(io.brace_expand(p.name[p.name.find('shard-'):]))
        if have and not (have - want) and (best is None or len(have) > best[0]):
            best = (len(have), p)
    return None if best is None else best[1]


def _attrs_of(obj):
    """attrs dict of one cached clustering: ours (a dict), or a pickled KMeans object (the reference's default
    scheme, run_clustering.py:110-116, readable when its class is importable)"""
    if isinstance(obj, dict):
        dt = dict(obj)
    elif hasattr(obj, 'get_attrs'):
        dt = dict(obj.get_attrs())
    else:
        dt = dict(vars(obj))
    for key in ('centers', 'counts'):
        v = dt[key]
        if hasattr(v, 'detach'):
            v = v.detach().cpu().numpy()
        dt[key] = np.ascontiguousarray(v, np.float32)
    return dt


def _read_cache(path):
    """-> {(kind, model_key, layer): attrs} or None.  The file is the reference's nested
    {model_key: {'layer_i' | 'model': KMeans | attrs}} written by torch.save (run_clustering.py:110-116); the flat
    pickle of round 1 is still accepted.  An unreadable file is reported and treated as absent."""
    import torch
    saved = None
    for reader in (lambda f: torch.load(f, map_location='cpu', weights_only=False), pickle.load):
        try:
            with open(path, 'rb') as f:
                saved = reader(f)
            break
        except Exception as exc:  # zip vs pickle stream, missing classes, truncated file ...
            err = exc
    if saved is None:
        print("clustering cache {} is not loadable: {}".format(path, err))
        return None
    try:
        flat = OrderedDict()
        for key, val in saved.items():
            if isinstance(key, tuple):  # round-1 layout: {(kind, model_key, layer): attrs}
                flat[key] = _attrs_of(val)
                continue
            for layer, obj in val.items():
                dt = _attrs_of(obj)
                flat[(dt.pop('_kind', None), key, layer)] = dt
        return flat
    except Exception as exc:
        print("clustering cache {} has an unknown layout: {}".format(path, exc))
        return None


def save_clusterings(args, epoch, cl):
    """torch.save of {model_key: {layer: attrs}} -- the reference's file and layout (its save_scheme_ver2 form,
    run_clustering.py:110-116: plain attrs instead of pickled objects, so either side can read it)."""
    import torch
    path = _cache_path(args, epoch)
    print("saving clustering cache to: {}".format(path))
    path.parent.mkdir(parents=True, exist_ok=True)
    nested = OrderedDict()
    for (kind, mk, layer), km in cl.items():
        dt = km.get_attrs_plain()
        dt['_kind'] = kind
        nested.setdefault(mk, OrderedDict())[layer] = dt
    torch.save(nested, str(path))


class _RowGroups:
    """The feature shards as a sequence of ROW GROUPS that fit the device budget.

    The reference never holds more than a batch (clustering/code/data/clustering.py:17-66, run_clustering.py:132-177:
    a DataLoader over the shards, re-read every epoch).  Here a group is as many consecutive shards as fit in half the
    budget (the next group is unpickled / memory-mapped by a host thread while the GPU works on the current one);
    when everything fits there is ONE group and its device tensors stay resident across epochs and the assign sweep.
    Nothing holds [N, d] for all N unless N fits."""

    def __init__(self, args, paths, sizes, row_bytes, budget, view_dims=None):
        import os
        self.args = args
        self.model_order = list(args.models or [])
        self.audio_models = tuple(args.model_types.audio or ())
        # the reference's `computation.num_workers` DataLoader workers (default 40) -> shard-reading worker processes
        # (shards.py: straight into shared memory); ACAV_LOAD_WORKERS overrides, 0 / 1 = read in this process
        nw = os.environ.get('ACAV_LOAD_WORKERS')
        nw = int(nw) if nw is not None else int(args.computation.num_workers or 0)
        self.workers = max(0, min(nw, os.cpu_count() or 1))
        self.sizes, self.view_dims = sizes, view_dims
        self.paths, self.row_bytes, self.budget = list(paths), int(row_bytes), int(budget)
        self._pool_warm = False
        total = sum(sizes[p.stem] for p in paths) * row_bytes
        if total <= budget:
            self.groups = [list(paths)]
        else:
            self.groups, cur, cur_bytes = [], [], 0
            for p in paths:
                nbytes = sizes[p.stem] * row_bytes
                if cur and cur_bytes + nbytes > budget // 2:
                    self.groups.append(cur)
                    cur, cur_bytes = [], 0
                cur.append(p)
                cur_bytes += nbytes
            if cur:
                self.groups.append(cur)
            print("streaming {} shards in {} groups (device budget {:.1f} MB, data {:.1f} MB)".format(
                len(paths), len(self.groups), budget / 1e6, total / 1e6))
        self._resident = None
        self.t_wait = self.t_upload = 0.0  # main-thread seconds waiting for the loader / copying host -> device (tools/bench_streamed.py)

    @property
    def streamed(self):
        return len(self.groups) > 1

    def iterate_stream(self, paths, extents, lb, row_bytes, budget, views=None):
        """Training order: -> (table, {view: device tensor [rows, d]}) per ROW GROUP of an epoch's batch stream.

        extents: [(shard index into paths, first row, rows)] in delivery order (row_plan.loader_stream), whole batches of lb
        rows in total.  Resident data: ONE group -- the resident rows gathered into stream order on the device (no gather when
        the stream is the rows in order).  Streamed: the stream is cut at batch boundaries where the shards a group touches
        would exceed half the budget; a group loads exactly the shards it touches (a shard the cut falls into is read by both
        neighbours; with worker streams a group touches one shard per worker at a time), uploads them and gathers."""
        import torch
        dev = _device(self.args)

        def gather(x, idx):
            if x.shape[1] == 0:
                return torch.empty((len(idx), 0))
            if len(idx) and idx[-1] - idx[0] == len(idx) - 1 and bool((np.diff(idx) == 1).all()):
                return x[int(idx[0]):int(idx[-1]) + 1]  # consecutive rows: a view, no copy
            return x[torch.from_numpy(idx).to(x.device)]

        def index_of(table, stems, ext):
            ids = {si: np.asarray(table.shard_rows.get(stems[si]) or (), np.int64) for si in {e[0] for e in ext}}
            for si, f, n in ext:
                if f + n > len(ids[si]):
                    raise RuntimeError("shard {} delivered {} rows, the batch plan (from its metadata) expects at least {}: fix the "
                                       "metadata or drop the shard from the list".format(stems[si], len(ids[si]), f + n))
            return np.concatenate([ids[si][f:f + n] for si, f, n in ext]) if ext else np.empty(0, np.int64)

        stems = [p.stem for p in paths]
        if not self.streamed:
            for _gi, table, rows in self.iterate(views=views):
                idx = index_of(table, stems, extents)
                in_order = len(idx) > 0 and idx[0] == 0 and idx[-1] == len(idx) - 1 and bool((np.diff(idx) == 1).all())
                # another order (worker streams, a wrapped tail): gathered copies of a few GB of batches at a time -- never a
                # second copy of the resident matrices; the SGD chain simply continues from call to call (as between row groups)
                piece = len(idx) if in_order or not len(idx) else max(lb, min(budget // 8, 4 << 30) // max(1, row_bytes) // lb * lb)
                for a in range(0, len(idx), max(1, piece)):
                    sub = idx[a:a + piece]
                    yield table, OrderedDict((v, gather(x, sub)) for v, x in rows.items())
            return
        # ---- cut the stream into groups
        size = [self.sizes[p.stem] * row_bytes for p in paths]
        cuts, cur, cur_sh, cur_bytes, cur_rows = [], [], set(), 0, 0

        def close():
            nonlocal cur, cur_sh, cur_bytes, cur_rows
            keep, head, acc = cur_rows // lb * lb, [], 0
            rest = []
            for si, f, n in cur:
                if acc + n <= keep:
                    head.append((si, f, n))
                elif acc >= keep:
                    rest.append((si, f, n))
                else:
                    head.append((si, f, keep - acc))
                    rest.append((si, f + keep - acc, n - (keep - acc)))
                acc += n
            if head:
                cuts.append(head)
            cur = rest
            cur_sh = {e[0] for e in cur}
            cur_bytes, cur_rows = sum(size[si] for si in cur_sh), sum(e[2] for e in cur)

        for si, f, n in extents:
            if si not in cur_sh and cur_rows >= lb and cur_bytes + size[si] > budget // 2:
                close()
            cur.append((si, f, n))
            if si not in cur_sh:
                cur_sh.add(si)
                cur_bytes += size[si]
            cur_rows += n
        close()
        assert not cur, "the batch stream is not a whole number of batches"

        def load(ext):
            order = list(OrderedDict.fromkeys(e[0] for e in ext))
            return self._load([paths[si] for si in sorted(order)])

        def up(v, m):
            if views is not None and v not in views:
                return torch.empty((m.shape[0], 0))
            import time
            t0 = time.perf_counter()
            t = torch.from_numpy(np.ascontiguousarray(m)).to(dev)
            self.t_upload += time.perf_counter() - t0
            return t

        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=1) as pool:
            pending = pool.submit(load, cuts[0]) if cuts else None
            for k, ext in enumerate(cuts):
                import time
                t0 = time.perf_counter()
                table = pending.result()
                self.t_wait += time.perf_counter() - t0
                if k + 1 < len(cuts):
                    pending = pool.submit(load, cuts[k + 1])  # host work beside the GPU's
                idx = index_of(table, stems, ext)
                out = OrderedDict()
                for v, m in table.views.items():
                    x = up(v, m)
                    out[v] = gather(x, idx)
                    del x
                yield table, out

    def _load(self, group):
        # streamed: every shard is read once per epoch and once more for the assign sweep.  Since the library reads the pkl
        # files itself (acav_pkl_load_group: 2.1 M rows/s per pass against 2.5 M from the columnar twins and 0.8 M through
        # pickle.load, profiles/r04_streamed_native.txt) a streamed run no longer writes twins on its own: ACAV_SHARD_SIDECAR=write
        sidecar = None
        if self.streamed and not self._pool_warm:  # the worker processes (sidecar reading, assignment writing) start before
            self._pool_warm = True                  # the first host block is registered with the GPU runtime (io.warm_pool)
            io.warm_pool(self.workers)
        return io.load_feature_shards(group, model_order=self.model_order, audio_models=self.audio_models, sidecar=sidecar,
                                      workers=self.workers, expect_rows=self.sizes, expect_views=self.view_dims)

    def __iter__(self):
        return self.iterate()

    def iterate(self, views=None, shards=None):
        """-> (index, table, {view: device tensor [rows, d]}).

        views:  the views this rank needs ON THE DEVICE (None: all).  The others come back as zero-width host tensors with
                the right number of rows -- a rank that trains 1 of 10 clusterings uploads 1 of 10 matrices.
        shards: only these shard stems (None: all).  Streamed: a group without any of them is skipped WITHOUT being read,
                and only the wanted shards of a group are unpickled.  Resident: the rows of the wanted shards are gathered
                on the host and uploaded (N / world rows); `table` stays the full table, and the third item carries
                '_rows' -> the table row indices the gathered rows correspond to."""
        import torch
        dev = _device(self.args)

        def up(v, m):
            if views is not None and v not in views:
                return torch.empty((m.shape[0], 0))
            import time
            t0 = time.perf_counter()
            t = torch.from_numpy(np.ascontiguousarray(m)).to(dev)
            self.t_upload += time.perf_counter() - t0
            return t

        if not self.streamed:
            if self._resident is None:
                self._resident = (self._load(self.groups[0]), OrderedDict())
            table, on_dev = self._resident
            if shards is not None and set(shards) < set(table.shard_rows):
                idx = np.array(sorted(i for s in shards for i in (table.shard_rows.get(s) or ())), np.int64)
                # a view that training left on the device is indexed there (no second copy of the matrix through the host)
                idx_dev = torch.from_numpy(idx).to(dev) if on_dev else None
                rows = OrderedDict((v, on_dev[v][idx_dev] if v in on_dev and (views is None or v in views) else up(v, m[idx]))
                                   for v, m in table.views.items())
                rows['_rows'] = idx
                yield 0, table, rows
                return
            rows = OrderedDict()
            for v, m in table.views.items():
                if views is not None and v not in views:
                    rows[v] = torch.empty((m.shape[0], 0))
                    continue
                if v not in on_dev:
                    on_dev[v] = up(v, m)  # stays resident across epochs and the assign sweep
                rows[v] = on_dev[v]
            yield 0, table, rows
            return
        want = None if shards is None else set(shards)
        todo = [(gi, g if want is None else [p for p in g if p.stem in want]) for gi, g in enumerate(self.groups)]
        todo = [(gi, g) for gi, g in todo if g]
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=1) as pool:
            pending = pool.submit(self._load, todo[0][1]) if todo else None
            for k, (gi, _g) in enumerate(todo):
                import time
                t0 = time.perf_counter()
                table = pending.result()
                self.t_wait += time.perf_counter() - t0
                if k + 1 < len(todo):
                    pending = pool.submit(self._load, todo[k + 1][1])  # host work beside the GPU's
                yield gi, table, OrderedDict((v, up(v, m)) for v, m in table.views.items())


def loader_settings(args):
    """(computation.num_workers as the batch ORDER sees it, tail) -- config.py: data.loader_order / data.loader_tail"""
    import os
    mode = str(os.environ.get('ACAV_LOADER_ORDER') or args.data.loader_order or 'reference').lower()
    if mode not in ('reference', 'workers', 'single'):
        raise ValueError("data.loader_order / ACAV_LOADER_ORDER must be 'reference' or 'single', not {!r}".format(mode))
    tail = str(os.environ.get('ACAV_LOADER_TAIL') or args.data.loader_tail or 'wrap').lower()
    if tail not in ('wrap', 'drop'):
        raise ValueError("data.loader_tail / ACAV_LOADER_TAIL must be 'wrap' or 'drop', not {!r}".format(tail))
    return (0 if mode == 'single' else int(args.computation.num_workers or 0)), tail


def _device_budget(args):
    """bytes of feature rows the device may hold at once: data.resident_bytes / ACAV_RESIDENT_BYTES, else 60 % of the
    free HBM (288 GB on MI355X: cfg2/cfg3's 2 x 4.1 GB are resident, cfg5's 2 x 410 GB stream)"""
    import os
    import torch
    v = os.environ.get('ACAV_RESIDENT_BYTES') or args.data.resident_bytes
    if v:
        return int(float(v))
    free, _total = torch.cuda.mem_get_info()
    return int(free * 0.6)


def train_clusters(args, probe, groups, shard_stems=()):
    import torch
    cl, loaded = load_clusterings(args, probe)
    if loaded and not args.clustering.resume_training:
        return cl
    pre = args.clustering.cached_epoch if loaded else 0
    rank, w = world()
    # `views` (default): several GPUs split the CLUSTERINGS, every one of which still sees every batch of every epoch with
    # the one-GPU arithmetic -- the N-GPU run produces the files of the one-GPU run.  (The reference divides the epochs by
    # the number of GPUs, run_clustering.py:146, and has every rank stream every shard with a per-rank batch of
    # batch_size / N: that run is `clustering.multi_gpu=reference`, _train_clusters_planned.)
    epochs = int(args.clustering.epochs)
    b = int(args.data.batch_size)
    print("training sgd kmeans for views: {}".format([v[1:] for v in cl]))
    if w > 1 and multi_gpu_mode(args) != 'views':
        return _train_clusters_planned(args, cl, groups, pre, epochs, b, multi_gpu_mode(args), shard_stems)
    # WHICH rows form the batches: the reference's DataLoader for this computation.num_workers (parallel/row_plan.py), as a
    # stream of (shard, first row, rows) per epoch.  The stream is planned from the shard sizes the metadata states (the
    # reference sizes its loader from the json files too, data/clustering.py:47-52); resident data: from the rows that loaded.
    from ..parallel import make_plan
    paths, sizes, row_bytes, budget = groups.paths, groups.sizes, groups.row_bytes, groups.budget
    meta_rows = [int(sizes[p.stem]) for p in paths]
    shard_rows = meta_rows
    if not groups.streamed:
        (_gi, table, _rows), = list(groups.iterate(views=set()))  # (reads the shards; uploads nothing yet)
        shard_rows = [len(table.shard_rows.get(p.stem) or ()) for p in paths]
    nw_order, tail = loader_settings(args)
    plan = make_plan('views', shard_rows, 1, b, epochs, num_workers=nw_order, meta_rows=meta_rows, tail=tail)
    first_row = np.concatenate([[0], np.cumsum(shard_rows)]).astype(np.int64)
    if plan.steps == 0:
        print("no whole batch of {} rows in {} rows: nothing to train on".format(b, sum(shard_rows)))
    print("loader order: {} -> {} steps of {} rows per epoch".format(plan.loader, plan.steps, b))

    def shard_extents(pe):  # the plan's (row range) extents back in (shard, first, rows) form, cut to the epoch
        out, need = [], pe.steps * b
        for _o, f, n in pe.extents[0]:
            n = min(n, need)
            need -= n
            while n > 0:
                si = int(np.searchsorted(first_row, f, side='right')) - 1
                m = min(n, int(first_row[si + 1]) - f)
                out.append((si, f - int(first_row[si]), m))
                f, n = f + m, n - m
            if need <= 0:
                break
        return out

    for epoch in range(pre, pre + epochs):
        lr = 0.1 ** (2 + epoch // 5)
        for km in cl.values():
            km.lr = lr
        # `for batch in dataloader` (run_clustering.py:235) creates a DataLoader iterator, which draws its
        # 64-bit base seed from the global CPU generator (torch/utils/data/dataloader.py,
        # _BaseDataLoaderIter.__init__: torch.empty((), dtype=int64).random_()) = two mt19937 words per
        # epoch, between the centre initialisation and the first warm-up draw.  Consumed here so that a
        # seeded run reproduces the reference CLI's files bit for bit (tests/test_gpu_cli.py).
        gen = next(iter(cl.values()))._generator
        gen.u32()
        gen.u32()
        # several GPUs: a rank only trains (hence only uploads) its share of the views -- view i -> rank i % world
        own = None if w == 1 else {v for i, v in enumerate(cl) if i % w == rank}
        ext = shard_extents(plan.at_epoch(epoch - pre))
        for table, part in groups.iterate_stream(paths, ext, b, row_bytes, budget, views=own):
            n = next(iter(part.values())).shape[0] if part else 0
            assert n % b == 0
            steps = n // b
            if steps == 0:
                continue
            # warm-up labels, drawn batch by batch across the clusterings like the reference loop (every rank draws
            # all of them from the same stream, so the generators stay in step whoever trains which clustering)
            need = {v: km.warmup_steps(b, steps) for v, km in cl.items()}
            warm = {v: np.empty((need[v], b), np.int64) for v in cl}
            for t in range(max(need.values(), default=0)):
                for v, km in cl.items():
                    if t < need[v]:
                        warm[v][t] = km.draw_warmup(b)
            if w > 1:  # the clusterings are dealt out over the GPUs; states are exchanged once per epoch
                from ..parallel import train_epoch_view_parallel
                train_epoch_view_parallel(cl, part, b, lr, warm, broadcast=False)
            else:  # all clusterings of the batch stream side by side on the GPU (independent SGD chains)
                KMeans.train_epoch_multi(list(cl.values()), [part[v] for v in cl], b, lr=lr, warm_bests=[warm[v] for v in cl])
            for km in cl.values():  # the gathered copy of this group is released before the next one is built
                km.synchronize()
            del part
        if w > 1:  # state identical to the one-GPU run on every rank
            from ..parallel import broadcast_states
            broadcast_states(cl)
        if rank == 0:
            save_clusterings(args, epoch, cl)
    return cl


def multi_gpu_mode(args):
    """clustering.multi_gpu: 'views' (default), 'striped', 'reference' or 'rows' -- see config.py"""
    mode = str(args.clustering.multi_gpu or 'views')
    if mode not in ('views', 'striped', 'reference', 'rows'):
        raise ValueError("clustering.multi_gpu must be 'views', 'striped', 'reference' or 'rows', not {!r}".format(mode))
    return mode


def _train_clusters_planned(args, cl, groups, pre, epochs, b, mode, shard_stems):
    """Multi-GPU training with the rows PARTITIONED over the ranks: `groups` holds THIS rank's shards (rank::world,
    mps/distributed.py:439 -- the node's aggregate HBM holds the rows, nothing streams through one GPU) and a plan
    (parallel/row_plan.py) says which rows form which step's global batch:

      striped     the ONE-GPU batch stream over partitioned rows: one slot, the shards in their global order,
                  batch_size rows per step, `epochs` epochs -- the files of the one-GPU run (and of the default `views` mode,
                  which reads every shard on the training rank instead), the rows resident on the ranks that own them.
      reference   the reference's own N-GPU run (sgd_clustering.py:94-129 under is_distributed): rank q feeds
                  int(batch_size / N) rows per step (data/clustering.py:25) of ITS stream over ALL shards in the rotated
                  order full[q::N] + full[q+1::N] + ... (mps/distributed.py:433-437), ceil(epochs / N) epochs
                  (run_clustering.py:146).  Global batch = batch_size; N * rows / batch_size steps per epoch.
      rows        large batch: every rank feeds batch_size rows of its OWN shards per step (global batch N x batch_size),
                  ceil(epochs / N) epochs -- N x N fewer SGD steps than `reference`; not the reference's run.

    KMeans.train_epoch_plan_multi: the rows of 1 024 steps at a time travel to the rank that runs a clustering's chain
    (clustering v -> rank v % N), no collective on the step path; then the trainers hand out their states.  The state
    equals ONE process fed the same global batches (tests/test_gpu_cli.py); the reference's own all-reduce of per-rank
    deltas differs from that by fp32 re-association in an order NCCL does not promise."""
    import torch
    import torch.distributed as dist
    from ..parallel import make_plan, plan_warmup_labels
    rank, w = world()
    if groups.streamed:
        raise RuntimeError("clustering.multi_gpu={} keeps every rank's rows resident: {} groups do not fit the device budget "
                           "(raise data.resident_bytes, use more GPUs, or clustering.multi_gpu=views)".format(mode, len(groups.groups)))
    (_gi, table, rows), = list(groups.iterate())
    groups_sizes = getattr(groups, 'all_sizes', None)
    # rows every shard REALLY delivered (an unreadable shard was reported and skipped by the loader: 0 rows), from its owner
    mine = {stem: len(table.shard_rows.get(stem) or ()) for stem in shard_stems[rank::w]}
    everyone = [None] * w
    dist.all_gather_object(everyone, mine)
    have = {}
    for part in everyone:
        have.update(part)
    shard_rows = [int(have.get(stem, 0)) for stem in shard_stems]
    nw_order, tail = loader_settings(args)
    meta_rows = [int(groups_sizes.get(stem, 0)) for stem in shard_stems] if groups_sizes else None
    plan = make_plan(mode, shard_rows, w, b, epochs, num_workers=nw_order, meta_rows=meta_rows, tail=tail)
    seg = [sum(shard_rows[r::w]) for r in range(w)]
    # the reference clamps num_gpus to the number of shards (script.py:22,37); a rank without rows (more GPUs than shards,
    # or every shard of a rank unreadable) or an epoch without a single step would leave untrained clusterings behind
    if min(seg) == 0 or plan.steps == 0:
        raise RuntimeError("clustering.multi_gpu={}: rows per rank {} give {} steps of {} rows -- fewer shards than GPUs, or "
                           "unreadable shards; lower computation.num_gpus".format(mode, seg, plan.steps, plan.global_batch))
    print("rank {}: {} local rows; mode {}: {} steps of {} x {} rows per epoch, {} epochs; loader {}".format(
        rank, seg[rank], mode, plan.steps, plan.slots, plan.lb, plan.epochs, plan.loader))
    kms = list(cl.values())
    xs = [rows[v] for v in cl]
    plan0 = plan
    for epoch in range(pre, pre + plan0.epochs):
        plan = plan0.at_epoch(epoch - pre)  # (the in-process loader's stream continues across epochs when it wraps)
        lr = 0.1 ** (2 + epoch // 5)
        for km in kms:
            km.lr = lr
        gen = kms[0]._generator  # the DataLoader iterator's seed draw, as in the one-GPU loop (every rank creates its own)
        gen.u32()
        gen.u32()
        # warm-up labels of this rank's slot, drawn batch by batch across the clusterings like the reference loop
        # (run_clustering.py:170-175 steps every clustering per batch), then exchanged once per clustering
        need = [km.warmup_steps(plan.global_batch, plan.steps) for km in kms]
        local = [np.empty((nd, plan.lb), np.int64) for nd in need]
        for t in range(max(need, default=0)):
            for i, km in enumerate(kms):
                if t < need[i]:
                    local[i][t] = km.draw_warmup(plan.lb)
        warm = [plan_warmup_labels(km, plan, device=xs[i].device, mine=local[i], comm_slot=i) for i, km in enumerate(kms)]
        trainers = KMeans.train_epoch_plan_multi(kms, xs, plan, lr=lr, warm_bests=warm)
        for i, km in enumerate(kms):
            km.broadcast_state_from(trainers[i], comm_slot=i)
        if rank == 0:
            save_clusterings(args, epoch, cl)
    return cl


def assign_clusters(args, groups, cl, shard_names):
    """run_clustering.py:180-272: label every row of this rank's shards, write {out}/{shard}.pkl."""
    out_dir = Path(args.data.output.path)
    prefix = '' if args.clustering.cached_epoch is None else 'epoch_{}_'.format(args.clustering.cached_epoch)
    print("extracting clustering for views: {}".format([v[1:] for v in cl]))
    # this rank's shards that still have to be written (:248-250), decided from the NAMES: a group without any of them
    # is never read, and only their rows are labelled (every rank sweeping every row would repeat the HBM-bound sweep,
    # the unpickling and the upload `world` times over)
    mine = [s for s in shard_names if not (out_dir / (s + '.pkl')).is_file()]
    saved = []
    writer = io.AssignmentWriter(groups.workers if len(mine) >= 16 else 0)
    mine_set = set(mine)
    for gi, table, rows in groups.iterate(shards=mine):
        todo = [s for s in table.shard_rows if s in mine_set]
        if not todo:
            continue
        sel = rows.pop('_rows', None)  # resident table, several ranks: the rows of this rank's shards, gathered
        labels = OrderedDict()
        for v, km in cl.items():
            best, _ = km.calc_best(rows[v], need_mean=False)
            best = best.cpu().numpy()
            if sel is not None:  # back into table row order (rows of other ranks' shards: never read)
                full = np.full(len(table), -1, np.int64)
                full[sel] = best
                best = full
            labels[v] = best
        for shard in todo:
            ids = table.shard_rows.get(shard)
            if not ids:  # unreadable (reported and skipped by the loader) or empty shard: nothing to write
                continue
            size = table.shard_size[ids[0]]
            if len(ids) < round(size * args.data.output.shard_ok_ratio):
                continue  # too incomplete to save (:261-268)
            out_path = out_dir / (prefix + shard + '.pkl')
            # (sidecar: columnar twin for our own subset-selection loader, opt-in: extra files)
            writer.submit(table, labels, ids, out_path, sidecar=io.sidecar_mode() == 'write')
            saved.append(out_path)
    writer.finish()
    order = {s: i for i, s in enumerate(shard_names)}
    return sorted(saved, key=lambda p: order.get(p.name[len(prefix):-4], 0))


def run_clustering(args):
    paths = [Path(p) for p in sorted(io.brace_expand(args.data.path))]
    sizes = io.shard_sizes_from_meta(paths, args.data.meta.path, use_cache=True)
    rank, w = world()
    if args.data.meta.path is not None and rank == 0:  # the reference's meta_cache.pkl in the meta dir: read above, kept up to date
        cache_path = Path(args.data.meta.path) / 'meta_cache.pkl'
        try:
            known = dict(io.load_pickle(cache_path)) if cache_path.is_file() else {}
        except Exception:
            known = {}
        if any(k not in known for k in sizes):
            io.dump_pickle({**known, **dict(sizes)}, cache_path)
    paths = [p for p in paths if p.stem in sizes]
    if not paths:
        print(f"All shards of {args.data.path} processing already done!")
        return []
    print(f"processing {len(paths)} shards")
    # the first shard THAT LOADS tells the views and their widths (the order KMeans objects are created in, and the row
    # size): the loader reports and skips an unreadable shard, and an empty probe would create no clustering at all
    for first in paths:
        probe = io.load_feature_shards([first], model_order=list(args.models or []),
                                       audio_models=tuple(args.model_types.audio or ()))
        if probe.views:
            break
    else:
        print(f"None of the {len(paths)} shards of {args.data.path} could be read")
        return []
    row_bytes = 4 * sum(m.shape[1] for m in probe.views.values())
    view_dims = OrderedDict((v, m.shape[1]) for v, m in probe.views.items())
    partitioned = w > 1 and multi_gpu_mode(args) != 'views'
    if partitioned and w > len(paths):  # the reference clamps num_gpus to the number of shards (script.py:22,37)
        raise RuntimeError("clustering.multi_gpu={}: {} GPUs for {} shards -- a rank without shards holds no rows to feed the "
                           "steps; lower computation.num_gpus".format(multi_gpu_mode(args), w, len(paths)))
    # reference / rows mode: a rank reads, holds and labels its own shards only (rank::world) -- training included
    groups = _RowGroups(args, paths[rank::w] if partitioned else paths, sizes, row_bytes, _device_budget(args), view_dims)
    groups.all_sizes = sizes  # metadata sizes of ALL shards (the loader length of a partitioned run is computed over all of them)
    cl = train_clusters(args, probe, groups, [p.stem for p in paths])
    mine = [p.stem for p in paths][rank::w]  # assign: shards strided over ranks (mps/distributed.py:439)
    return assign_clusters(args, groups, cl, mine)


def store_shards_set(args, saved_paths):
    """clustering/code/save.py:9-17"""
    if len(saved_paths) == 0:
        print("All shards already processed")
        return None
    out_path = saved_paths[0].parent / ('log_' + args.run_id + '.json')
    io.dump_json({**args.run_info, 'shards': [p.stem for p in saved_paths]}, out_path, indent=None)
    return out_path
