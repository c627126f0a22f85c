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

Pure host arithmetic (no torch, no GPU): the CPU tests pin it against the shard orders the reference's own
node_selection returns (tests/golden/ddp_stream.npz).
"""
import math
from collections import namedtuple

Piece = namedtuple("Piece", "slot rel owner first rows")  # rows [first, first + rows) of `owner` = stream positions [rel, rel + rows) of `slot`, chunk-relative


class RowPlan:
    def __init__(self, mode, world, slots, lb, extents, steps, epochs):
        self.mode, self.world, self.slots, self.lb = str(mode), int(world), int(slots), int(lb)
        self.extents = [[(int(o), int(f), int(n)) for o, f, n in ext if n > 0] for ext in extents]  # per slot, stream order
        self.steps, self.epochs = int(steps), int(epochs)
        assert self.slots == len(self.extents) and self.lb > 0 and self.steps >= 0
        for ext in self.extents:
            assert sum(n for _o, _f, n in ext) >= self.steps * self.lb, "a stream is shorter than the epoch"

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
        out = []
        lo, hi = t0 * self.lb, t1 * self.lb
        for q, ext in enumerate(self.extents):
            pos = 0
            for owner, first, n in ext:
                a, b = max(lo, pos), min(hi, pos + n)
                if a < b:
                    out.append(Piece(q, a - lo, owner, first + (a - pos), b - a))
                pos += n
                if pos >= hi:
                    break
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
                "slots": self.slots, "sgd_steps_per_epoch": self.steps, "epochs": self.epochs}


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


def plan_reference(shard_rows, world, batch_size, epochs):
    """The reference's N-GPU batch stream (module docstring).  shard_rows: rows of every shard in global order."""
    world = int(world)
    lb = int(batch_size / world)  # data/clustering.py:25
    if lb < 1:
        raise ValueError("data.batch_size {} is smaller than the number of GPUs {}: the reference's per-rank batch "
                         "int(batch_size / num_gpus) would be empty".format(batch_size, world))
    seg, where = segments_of(shard_rows, world)
    total = sum(seg)
    extents = []
    for q in range(world):  # mps/distributed.py:433-437: for i in [q, q + 1, ...] mod W: full_urls[i::W]
        ext = []
        for i in (x % world for x in range(q, q + world)):
            ext.append((i, 0, seg[i]))  # rank i's segment IS full[i::W] in order
        extents.append(ext)
    # every stream covers all rows; DataLoader(drop_last=True) -- the reference's ResizedDataset rounds the stream UP to a
    # batch multiple by wrapping around (mps/distributed.py:444-460), which cannot be checked offline: we round down
    return RowPlan("reference", world, world, lb, extents, total // lb, math.ceil(epochs / world))  # run_clustering.py:146


def plan_views(shard_rows, world, batch_size, epochs):
    """The one-GPU run's stream over partitioned rows: one slot, shards in global order."""
    seg, where = segments_of(shard_rows, world)
    ext = [(where[s][0], where[s][1], int(n)) for s, n in enumerate(shard_rows)]
    merged = []
    for o, f, n in ext:  # neighbouring extents of one owner that are contiguous become one (world == 1: a single extent)
        if merged and merged[-1][0] == o and merged[-1][1] + merged[-1][2] == f:
            merged[-1] = (o, merged[-1][1], merged[-1][2] + n)
        else:
            merged.append((o, f, n))
    return RowPlan("views", world, 1, int(batch_size), [merged], sum(seg) // int(batch_size), int(epochs))


def plan_rows(shard_rows, world, batch_size, epochs):
    """Large-batch mode: every rank feeds batch_size of ITS rows per step; the shortest rank decides the step count."""
    seg, _ = segments_of(shard_rows, world)
    lb = int(batch_size)
    return RowPlan("rows", world, world, lb, [[(q, 0, seg[q])] for q in range(world)], min(seg) // lb if seg else 0,
                   math.ceil(epochs / world))


PLANS = {"reference": plan_reference, "views": plan_views, "striped": plan_views, "rows": plan_rows}


def make_plan(mode, shard_rows, world, batch_size, epochs):
    if mode not in PLANS:
        raise ValueError("unknown multi-GPU training mode {!r} (one of {})".format(mode, sorted(PLANS)))
    return PLANS[mode](list(shard_rows), int(world), int(batch_size), int(epochs))
