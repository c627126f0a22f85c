"""Which rank's rows make up which SGD step -- the batch STREAM of a multi-GPU k-means training epoch as data.

Every rank holds the feature rows of its own shards (a SEGMENT: shards rank::world, mps/distributed.py:439).  A training
step's global batch is put together from `slots` row streams, `lb` rows of each per step, slot-major:

    batch(t) = concat_{q = 0 .. slots-1}  stream_q[t * lb : (t + 1) * lb]

and a stream is a list of EXTENTS (owner rank, first local row, rows) in the order its rows are consumed.  The three
multi-GPU training modes are three plans over the same machinery (the rows of `chunk_steps` future steps travel in bulk to
the one rank that runs a clustering's SGD chain -- KMeans.train_epoch_plan_multi / acav_kmeans_train_plan_multi):

  reference   what /root/reference does with N GPUs: slot q = the reference's rank q, which streams ALL shards in the
              rotated order  full[q::W] + full[q+1::W] + ...  (mps/distributed.py:433-437: node_selection(is_train=True))
              with a per-rank batch of int(batch_size / W) rows (data/clustering.py:25), for ceil(epochs / W) epochs
              (run_clustering.py:146).  The global batch stays batch_size rows; an epoch has W * N / batch_size steps.
  views       the one-GPU run's own stream: ONE slot, the shards in their global order, batch_size rows per step,
  (striped)   `epochs` epochs -- the arithmetic (and the files) of a one-GPU run over all rows, with the rows living on
              the ranks that own them: SURVEY 8(e)'s row-striped partition (bench.py --gpus N; the CLI's
              `clustering.multi_gpu=striped`.  The CLI's default `views` mode reaches the same result by reading every
              shard on the training rank and needs no plan).
  rows        a LARGE-BATCH operating point, not the reference's run: slot q = rank q's own segment only, batch_size rows
              of every rank per step (global batch W * batch_size), ceil(epochs / W) epochs: W * W fewer SGD steps than
              `reference` (64 x at 8 GPUs).

WHICH rows a rank's stream delivers, and in which order, is the reference's DataLoader (loader_stream below): with
`computation.num_workers` > 0 (its default is 40, config.py:29) torch's DataLoader round-robins whole batches over worker
processes, worker w streaming the rank's shards [w::num_workers] (data/clustering.py:212-228), and every worker's stream is
cut / cycled to get_length() samples (mps/distributed.py:444-460, webdataset.ResizedDataset at data/clustering.py:61-65).

Pure host arithmetic (no torch, no GPU): the CPU tests pin it against the shard orders the reference's own
node_selection returns (tests/golden/ddp_stream.npz) and against the batches the reference's own
get_clustering_dataloader delivers (tests/golden/loader_order.npz).
"""
import math
from collections import namedtuple

Piece = namedtuple("Piece", "slot rel owner first rows")  # rows [first, first + rows) of `owner` = stream positions [rel, rel + rows) of `slot`, chunk-relative


class RowPlan:
    def __init__(self, mode, world, slots, lb, extents, steps, epochs, per_epoch=None, loader=None):
        self.mode, self.world, self.slots, self.lb = str(mode), int(world), int(slots), int(lb)
        self.extents = [[(int(o), int(f), int(n)) for o, f, n in ext if n > 0] for ext in extents]  # per slot, stream order
        self.steps, self.epochs = int(steps), int(epochs)
        # per_epoch(e) -> extents of epoch e (counted from the start of THIS run) when they differ from epoch 0's: the
        # in-process loader (num_workers = 0) keeps its source iterator across epochs, so a stream that wraps starts every
        # epoch where the previous one stopped
        self._per_epoch, self.loader = per_epoch, dict(loader or {})
        assert self.slots == len(self.extents) and self.lb > 0 and self.steps >= 0
        self._starts = []
        for ext in self.extents:
            pos, st = 0, []
            for _o, _f, n in ext:
                st.append(pos)
                pos += n
            self._starts.append(st)
            assert pos >= self.steps * self.lb, "a stream is shorter than the epoch"

    def at_epoch(self, e):
        """the plan of epoch e of this run (e = 0, 1, ...): same steps and batch geometry, that epoch's rows"""
        if self._per_epoch is None or e == 0:
            return self
        return RowPlan(self.mode, self.world, self.slots, self.lb, self._per_epoch(int(e)), self.steps, self.epochs,
                       loader=self.loader)

    def order(self, slot=0):
        """the rows of one slot's stream over the epoch as one index array (owner-local rows; one-owner plans: the gather
        that puts resident rows into batch order)"""
        import numpy as np
        need, parts, pos = self.steps * self.lb, [], 0
        for _o, f, n in self.extents[slot]:
            n = min(n, need - pos)
            if n <= 0:
                break
            parts.append(np.arange(f, f + n, dtype=np.int64))
            pos += n
        return np.concatenate(parts) if parts else np.empty(0, np.int64)

    def is_identity(self, rows):
        """one slot streaming rows 0 .. steps * lb - 1 of one owner in order (the single-stream loader on resident rows)"""
        return self.slots == 1 and len(self.extents[0]) == 1 and self.extents[0][0][1] == 0 and self.extents[0][0][2] <= rows

    @property
    def global_batch(self):
        return self.slots * self.lb

    def table(self):
        """the extents as the flat int64 [n_ext, 4] array (slot, owner, first row, rows) of the C ABI"""
        import numpy as np
        rows = [(q, o, f, n) for q, ext in enumerate(self.extents) for o, f, n in ext]
        return np.asarray(rows, np.int64).reshape(-1, 4)

    def pieces(self, t0, t1):
        """the row ranges that make up steps [t0, t1), ordered by (slot, stream position)"""
        import bisect
        out = []
        lo, hi = t0 * self.lb, t1 * self.lb
        for q, ext in enumerate(self.extents):
            st = self._starts[q]
            for i in range(max(0, bisect.bisect_right(st, lo) - 1), len(ext)):
                pos = st[i]
                if pos >= hi:
                    break
                owner, first, n = ext[i]
                a, b = max(lo, pos), min(hi, pos + n)
                if a < b:
                    out.append(Piece(q, a - lo, owner, first + (a - pos), b - a))
        return out

    def rows_of(self, rank, t0, t1):
        return sum(p.rows for p in self.pieces(t0, t1) if p.owner == rank)

    def batch_sources(self, t):
        """[(owner, local row)] of step t's global batch, in batch order (tests)"""
        out = [None] * self.global_batch
        for p in self.pieces(t, t + 1):
            for i in range(p.rows):
                out[p.slot * self.lb + p.rel + i] = (p.owner, p.first + i)
        return out

    def describe(self):
        return {"mode": self.mode, "world": self.world, "global_batch": self.global_batch, "rows_per_slot_and_step": self.lb,
                "slots": self.slots, "sgd_steps_per_epoch": self.steps, "epochs": self.epochs, **({"loader": self.loader} if self.loader else {})}


def segments_of(shard_rows, world):
    """rows every rank holds when the shards (in their global, sorted order) are strided rank::world and kept in that
    order: -> (rows per rank, {shard index: (owner, first local row)})"""
    seg = [0] * world
    where = {}
    for s, n in enumerate(shard_rows):
        r = s % world
        where[s] = (r, seg[r])
        seg[r] += int(n)
    return seg, where


def loader_workers(num_workers, n_shards):
    """-> (DataLoader num_workers, streams): FeatureDataset.num_workers = min(computation.num_workers, shards of a rank)
    (data/clustering.py:118-127; every rank streams ALL shards when training); 0 = the in-process loader, one stream."""
    nw = max(0, min(int(num_workers or 0), int(n_shards)))
    return nw, max(1, nw)


def loader_length(meta_rows, lb, num_workers):
    """samples every worker's stream is cut / cycled to: mps/distributed.py:444-460 get_length(is_train=True) -- the longest
    worker of the UNROTATED shard list, rounded up to whole batches.  meta_rows: the shard sizes the metadata states (the
    reference computes the length from the json files, data/clustering.py:47-52)."""
    _nw, eff = loader_workers(num_workers, len(meta_rows))
    return max((math.ceil(sum(meta_rows[w::eff]) / lb) for w in range(eff)), default=0) * lb


class _Cycled:
    """a worker's source: its shards one after the other, started over when exhausted (ResizedDataset.__iter__)"""

    def __init__(self, shards, shard_rows):
        self.src = [(s, int(shard_rows[s])) for s in shards if int(shard_rows[s]) > 0]
        self.starts, pos = [], 0
        for _s, n in self.src:
            self.starts.append(pos)
            pos += n
        self.total = pos

    def take(self, pos, n):
        """extents (shard, first, rows) of stream positions [pos, pos + n)"""
        import bisect
        if self.total == 0:
            raise ValueError("a loader worker has no rows to stream (all of its shards are empty or unreadable)")
        out = []
        while n > 0:
            p = pos % self.total
            i = bisect.bisect_right(self.starts, p) - 1
            s, rows = self.src[i]
            off = p - self.starts[i]
            m = min(n, rows - off)
            if out and out[-1][0] == s and out[-1][1] + out[-1][2] == off:
                out[-1] = (s, out[-1][1], out[-1][2] + m)
            else:
                out.append((s, off, m))
            pos, n = pos + m, n - m
        return out


def loader_stream(shard_rows, order, num_workers, lb, length, epoch=0, tail="wrap"):
    """The rows ONE rank's DataLoader delivers in one epoch, in delivery order: [(shard, first row, rows)].

    shard_rows  rows every shard really holds, by global shard index; order: the rank's shard order (node_selection)
    num_workers computation.num_workers as configured; clamped like the reference (loader_workers)
    lb, length  per-rank batch and loader_length()
    tail        'wrap' (the reference: every stream is `length` samples, a short source starts over -- and with num_workers = 0
                the source iterator lives on into the next epoch, so epoch e starts at sample e * length of the cycled
                source) | 'drop' (rounds 1-5 of this build: whole batches of the rows that exist, nothing repeated)

    num_workers > 0: worker w streams order[w::nw]; the DataLoader hands out batch j of worker 0, 1, ..., nw - 1, then batch
    j + 1 (tasks go to the workers round-robin and results come back in task order); every worker is a fresh process per
    epoch (its ResizedDataset starts at its source's first sample every time).  Pinned by tests/golden/loader_order.npz."""
    nw, eff = loader_workers(num_workers, len(order))
    workers = [_Cycled(order[w::eff], shard_rows) for w in range(eff)]
    if tail == "drop":
        if eff == 1:
            n = workers[0].total // lb * lb
            return workers[0].take(0, n) if n else []
        rounds = [wk.total // lb for wk in workers]
        out = []
        for j in range(max(rounds, default=0)):  # a worker that runs out of whole batches drops out of the round-robin
            for w, wk in enumerate(workers):
                if j < rounds[w]:
                    out += wk.take(j * lb, lb)
        return out
    if eff == 1:
        start = epoch * length if nw == 0 else 0
        return workers[0].take(start, length)
    out = []
    for j in range(length // lb):
        for wk in workers:
            out += wk.take(j * lb, lb)
    return out


def _merge(ext):
    out = []
    for o, f, n in ext:
        if out and out[-1][0] == o and out[-1][1] + out[-1][2] == f:
            out[-1] = (o, out[-1][1], out[-1][2] + n)
        else:
            out.append((o, f, n))
    return out


def _loader_plan(mode, shard_rows, world, slots, lb, epochs, orders, num_workers, meta_rows, tail, owners):
    """slots streams, stream q = the loader of shard order orders[q]; owners: {shard: (owner rank, first local row)}"""
    meta_rows = list(shard_rows if meta_rows is None else meta_rows)
    length = loader_length(meta_rows, lb, num_workers)
    nw, eff = loader_workers(num_workers, len(shard_rows))

    def extents(epoch):
        out = []
        for order in orders:
            st = loader_stream(shard_rows, order, num_workers, lb, length, epoch, tail)
            out.append(_merge([(owners[s][0], owners[s][1] + f, n) for s, f, n in st]))
        return out

    first = extents(0)
    steps = min((sum(n for _o, _f, n in ext) for ext in first), default=0) // lb
    varies = tail == "wrap" and nw == 0 and length % max(1, sum(int(n) for n in shard_rows)) != 0
    return RowPlan(mode, world, slots, lb, first, steps, epochs, per_epoch=extents if varies else None,
                   loader={"num_workers": nw, "streams_per_rank": eff, "samples_per_stream": length, "tail": tail})


def plan_reference(shard_rows, world, batch_size, epochs, num_workers=0, meta_rows=None, tail="wrap"):
    """The reference's N-GPU batch stream (module docstring).  shard_rows: rows of every shard in global order."""
    world = int(world)
    lb = int(batch_size / world)  # data/clustering.py:25
    if lb < 1:
        raise ValueError("data.batch_size {} is smaller than the number of GPUs {}: the reference's per-rank batch "
                         "int(batch_size / num_gpus) would be empty".format(batch_size, world))
    _seg, where = segments_of(shard_rows, world)
    nsh = len(shard_rows)
    orders = []
    for q in range(world):  # mps/distributed.py:433-437: for i in [q, q + 1, ...] mod W: full_urls[i::W]
        orders.append([s for i in (x % world for x in range(q, q + world)) for s in range(i, nsh, world)])
    return _loader_plan("reference", shard_rows, world, world, lb, math.ceil(epochs / world),  # run_clustering.py:146
                        orders, num_workers, meta_rows, tail, where)


def plan_views(shard_rows, world, batch_size, epochs, num_workers=0, meta_rows=None, tail="wrap"):
    """The one-GPU run's stream over partitioned rows: one slot, the one-GPU loader over the shards in global order."""
    _seg, where = segments_of(shard_rows, world)
    return _loader_plan("views", shard_rows, world, 1, int(batch_size), int(epochs), [list(range(len(shard_rows)))],
                        num_workers, meta_rows, tail, where)


def plan_rows(shard_rows, world, batch_size, epochs, num_workers=0, meta_rows=None, tail="wrap"):
    """Large-batch mode: every rank feeds batch_size of ITS rows per step; the shortest rank decides the step count."""
    seg, _ = segments_of(shard_rows, world)
    lb = int(batch_size)
    return RowPlan("rows", world, world, lb, [[(q, 0, seg[q])] for q in range(world)], min(seg) // lb if seg else 0,
                   math.ceil(epochs / world))


PLANS = {"reference": plan_reference, "views": plan_views, "striped": plan_views, "rows": plan_rows}


def make_plan(mode, shard_rows, world, batch_size, epochs, num_workers=0, meta_rows=None, tail="wrap"):
    """num_workers: computation.num_workers (0 = the single-stream loader); meta_rows: the shard sizes the metadata states
    (None: shard_rows); tail: 'wrap' (the reference's ResizedDataset) | 'drop'."""
    if mode not in PLANS:
        raise ValueError("unknown multi-GPU training mode {!r} (one of {})".format(mode, sorted(PLANS)))
    if tail not in ("wrap", "drop"):
        raise ValueError("loader tail must be 'wrap' or 'drop', not {!r}".format(tail))
    return PLANS[mode](list(shard_rows), int(world), int(batch_size), int(epochs), num_workers, meta_rows, tail)
