"""Multi-GPU KMeans training -- the host side of the exchanges (torch.distributed: RCCL on the GPUs, gloo in the CPU
tests).  The library's own RCCL path (csrc/acav_comm.hip) does the same inside one C call; this module is what runs when
the group's backend is not RCCL, and what defines the semantics.

Reference per step and clustering (sgd_clustering.py:94-129 under is_distributed): all_gather([batch]) (only to take
len()), all_reduce([counts]), all_reduce([deltas]) = 2 x K*d*4 bytes of ring traffic, every rank labelling its own
int(batch_size / world) rows (data/clustering.py:25).  Here the rows travel instead (b*d*4 bytes) and ONE process applies
the whole global batch with the deterministic in-order update kernel: the state equals a single process fed the same
global batches (the reference's own sum of per-rank deltas differs from that by fp32 re-association only, in an order
NCCL does not promise).  WHICH rows form a step's global batch is a plan (row_plan.py): `reference` = the reference's own
N-GPU stream, `views` = the one-GPU stream, `rows` = a large-batch mode (global batch world x batch_size).

`engine` is anything with calc_best(batch) -> (labels, mean) and apply_update(x, labels, lr): the GPU
KMeans in production; the multi-process CPU tests plug a host stand-in to exercise the
collective plumbing under gloo.
"""
import torch
import torch.distributed as dist

from .collectives import gather_rows_and_labels, world


def _as_tensor(a, like=None):
    if torch.is_tensor(a):
        return a
    t = torch.as_tensor(a)
    return t.to(like.device) if like is not None else t


def distributed_add(engine, batch, lr):
    """One global SGD step.  Returns the mean of the per-rank mean min-distances."""
    rank, w = world()
    batch_t = _as_tensor(batch)
    best, mean = engine.calc_best(batch)
    best_t = _as_tensor(best, like=batch_t)
    xg, bg = gather_rows_and_labels(batch_t, best_t)
    engine.apply_update(xg, bg, lr)
    if w > 1:
        m = torch.tensor([float(mean)], dtype=torch.float64, device=xg.device)
        dist.all_reduce(m)
        mean = float(m.item()) / w
    return mean


def train_epoch_dp(engine, x_local, b_local, lr, chunk_steps=1024, force_collective=False, trainer=None):
    """One training epoch in the LARGE-BATCH mode (row_plan.plan_rows; not the reference's N-GPU run, whose global batch
    stays batch_size -- train_epoch_plan with plan_reference): step t's global batch is the rank-major concatenation of
    every rank's rows [t*b_local, (t+1)*b_local), and every rank applies the same update.

    The SGD chain is latency-bound (each step needs the centres of the previous one), so a per-step collective
    would cost more than the step itself.  But the ROWS of future steps do not depend on the state: they are
    all-gathered in bulk, `chunk_steps` steps at a time (one large collective over xGMI, overlapped with the
    training of the previous chunk), and every rank then runs the identical device-resident epoch over the global
    batches (engine.train_epoch, b = world * b_local) -- no collective on the step path, bit-identical state on all
    ranks, equal to distributed_add() step by step.  Only the warm-up labels (drawn from each rank's own
    generator) travel separately: a few KB, once.

    `engine`: train_epoch(x, b, lr, warm_best=), warmup_steps(b, steps), draw_warmup(b), synchronize().
    force_collective: take the gather path even with one rank (single-GPU test of the plumbing).
    trainer: the ONE rank that runs the chain (the same number on every rank; the others follow with broadcast_state):
    the rows are gathered TO it (dist.gather) instead of all-gathered -- what acav_kmeans_train_dp does under
    ACAV_DP_ROOTED -- and a rank that only sends keeps `count` in step through engine.skip_epoch."""
    rank, w = world()
    x_local = _as_tensor(x_local)
    n_local, d = x_local.shape
    steps = n_local // b_local
    if w == 1 and not force_collective:
        engine.train_epoch(x_local, b_local, lr)
        return
    bg = w * b_local
    # warm-up labels: rank r draws the labels of ITS rows; everybody needs all of them
    need = engine.warmup_steps(bg, steps)
    warm = None
    if need:
        mine = torch.stack([_as_tensor(engine.draw_warmup(b_local)).to(torch.long) for _ in range(need)])
        mine = mine.to(x_local.device).contiguous()
        allw = torch.empty((w * need, b_local), dtype=torch.long, device=x_local.device)  # rank-major concatenation
        dist.all_gather_into_tensor(allw, mine)
        warm = allw.view(w, need, b_local).permute(1, 0, 2).reshape(need, bg).cpu().numpy()  # [step, rank-major rows]

    here = trainer is None or int(trainer) == rank

    def gather(c0, s):
        loc = x_local[c0 * b_local:(c0 + s) * b_local].contiguous()
        if trainer is None:
            g = torch.empty((w * s * b_local, d), dtype=loc.dtype, device=loc.device)  # rank-major concatenation
            dist.all_gather_into_tensor(g, loc)
        else:  # only the trainer receives
            parts = [torch.empty_like(loc) for _ in range(w)] if here else None
            dist.gather(loc, parts, dst=int(trainer))
            if not here:
                return None
            g = torch.cat(parts)
        # [rank, step, row] -> [step, rank, row]: the global batches, contiguous
        return g.view(w, s, b_local, d).permute(1, 0, 2, 3).reshape(s * bg, d).contiguous()

    done_warm = 0
    nxt = gather(0, min(chunk_steps, steps)) if steps else None
    for c0 in range(0, steps, chunk_steps):
        s = min(chunk_steps, steps - c0)
        cur = nxt
        engine.synchronize()  # the chunk trained before `cur` is finished: its buffer may be recycled
        wb = None
        nw = min(max(need - done_warm, 0), s)
        if need:
            wb = warm[done_warm:done_warm + nw]
            done_warm += nw
        if here:
            engine.train_epoch(cur, bg, lr, warm_best=wb if need else None)  # asynchronous on the engine's stream
        if c0 + s < steps:  # gather the next chunk while this one trains
            nxt = gather(c0 + s, min(chunk_steps, steps - c0 - s))
    engine.synchronize()
    if not here:
        engine.skip_epoch(steps * bg)


def plan_warmup_labels(engine, plan, device=None, comm=None, mine=None, comm_slot=None):
    """labels of the warm-up steps of a planned epoch, [need, slots * lb] in batch order (None when there are none).
    slots == world (reference / rows): slot q's rows are labelled by rank q -- calc_best(local batch) draws
    torch.rand(k, local_b) from the rank's own generator (sgd_clustering.py:67-68,111) -- and the labels are exchanged
    once.  One slot (views): the one-GPU run's draws, torch.rand(k, batch_size) per step from this process's generator
    (every rank draws them: generators seeded alike stay in step; the trainer's are used).
    mine: this rank's labels [need, lb] drawn by the caller (several clusterings sharing one generator draw batch by
    batch across the clusterings).  comm_slot: take the library's communicator of that slot when RCCL is the backend."""
    rank, w = world()
    need = engine.warmup_steps(plan.global_batch, plan.steps)
    if not need:
        return None
    import numpy as np
    if plan.slots != w or w == 1:
        if mine is not None and plan.slots == 1:
            return np.ascontiguousarray(mine, np.int64)
        per = [np.asarray(engine.draw_warmup(plan.global_batch)).astype(np.int64) for _ in range(need)]
        return np.ascontiguousarray(np.stack(per))
    if comm is None and comm_slot is not None and device is not None and getattr(device, "type", str(device)[:4]) == "cuda":
        from .rccl_comm import default_comm
        comm = default_comm(comm_slot)
    if mine is None:
        mine = [_as_tensor(engine.draw_warmup(plan.lb)).to(torch.long) for _ in range(need)]
        mine = torch.stack(mine).contiguous()
    else:
        mine = torch.as_tensor(np.ascontiguousarray(mine, np.int64))
        assert tuple(mine.shape) == (need, plan.lb), (tuple(mine.shape), need, plan.lb)
    if comm is not None:  # the library's communicator (RCCL)
        send = mine.to(device)
        recv = torch.empty((w, need, plan.lb), dtype=torch.long, device=device)
        comm.allgather(send, recv)
        comm.synchronize()
    else:
        send = mine.to(device if device is not None else "cpu")
        recv = torch.empty((w * need, plan.lb), dtype=torch.long, device=send.device)
        dist.all_gather_into_tensor(recv, send)
        recv = recv.view(w, need, plan.lb)
    return np.ascontiguousarray(recv.cpu().numpy().transpose(1, 0, 2).reshape(need, w * plan.lb))  # [step, slot-major rows]


def train_epoch_plan(engine, x_local, plan, lr, trainer, chunk_steps=1024, warm=None):
    """One training epoch over the global batches of `plan` (row_plan.RowPlan) with the rows living on their owners: per
    chunk of `chunk_steps` steps every rank packs the rows it owns, they are gathered to rank `trainer` (dist.gather, a
    chunk ahead of the training), which places them into global batches [step][slot][row] and trains
    (engine.train_epoch at the global batch size).  What acav_kmeans_train_plan_multi does inside the library over
    RCCL; this form runs under any torch.distributed backend (gloo: CPU tests, several ranks on one GPU).  A rank that only sends keeps `count` in step (engine.skip_epoch) and takes the
    trainer's state afterwards (broadcast_state).  warm: plan_warmup_labels() (drawn here when None)."""
    rank, w = world()
    x_local = _as_tensor(x_local)
    d = x_local.shape[1]
    here = int(trainer) == rank
    steps, bg, lb = plan.steps, plan.global_batch, plan.lb
    if warm is None:
        warm = plan_warmup_labels(engine, plan, device=x_local.device)
    need = 0 if warm is None else len(warm)
    if steps == 0:
        return

    def gather(t0, t1):
        pcs = plan.pieces(t0, t1)
        by = [sum(p.rows for p in pcs if p.owner == r) for r in range(w)]
        cap = max(by) if by else 0
        send = torch.zeros((max(cap, 1), d), dtype=x_local.dtype, device=x_local.device)  # dist.gather wants equal shapes
        mine = [x_local[p.first:p.first + p.rows] for p in pcs if p.owner == rank]
        if mine:
            send[:by[rank]] = torch.cat(mine)
        if w > 1:
            parts = [torch.empty_like(send) for _ in range(w)] if here else None
            dist.gather(send, parts, dst=int(trainer))
        else:
            parts = [send]
        if not here:
            return None
        out = torch.empty(((t1 - t0) * bg, d), dtype=x_local.dtype, device=x_local.device)
        fill = [0] * w
        for p in pcs:
            rel = torch.arange(p.rel, p.rel + p.rows, device=x_local.device)
            out[(rel // lb) * bg + p.slot * lb + rel % lb] = parts[p.owner][fill[p.owner]:fill[p.owner] + p.rows]
            fill[p.owner] += p.rows
        return out

    done_warm = 0
    nxt = gather(0, min(chunk_steps, steps))
    for c0 in range(0, steps, chunk_steps):
        s = min(chunk_steps, steps - c0)
        cur = nxt
        engine.synchronize()  # the chunk trained before `cur` is finished: its buffer may be recycled
        nw = min(max(need - done_warm, 0), s)
        wb = warm[done_warm:done_warm + nw] if need else None
        done_warm += nw
        if here:
            engine.train_epoch(cur, bg, lr, warm_best=wb)  # asynchronous on the engine's stream
        if c0 + s < steps:  # gather the next chunk while this one trains
            nxt = gather(c0 + s, c0 + s + min(chunk_steps, steps - c0 - s))
    engine.synchronize()
    if not here:
        engine.skip_epoch(steps * bg)


def average_state(centers, counts):
    """KMeans.initialize(): all_reduce(SUM) then * 1/world (mps/distributed.py:139-155)."""
    rank, w = world()
    if w == 1:
        return centers, counts
    buf = torch.cat([centers.reshape(-1), counts.reshape(-1)])
    # Ranks that drew the same initial state (same seed: the normal case here) must keep it bit for bit -- (c + c + c) / 3
    # is not c in fp32, and an N-GPU run is meant to produce the files of the one-GPU run.  Only ranks that really
    # differ are averaged, as the reference does.
    digest = buf.view(torch.int32).to(torch.int64).sum().reshape(1)
    lo, hi = digest.clone(), digest.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if int(lo.item()) == int(hi.item()):
        return centers, counts
    dist.all_reduce(buf)
    buf = buf * (1.0 / w)
    n = centers.numel()
    return buf[:n].reshape(centers.shape), buf[n:].reshape(counts.shape)


def broadcast_state(engine, src):
    """Every rank ends up with rank `src`'s clustering state (centres, usage counts, count, fallback): two small
    broadcasts (K*d*4 + K*4 bytes, 16 bytes).  `engine`: state_arrays() -> (centers f32 [K,d], counts f32 [K], count,
    fallback) and load_state_arrays(centers, counts, count, fallback)."""
    rank, w = world()
    if w == 1:
        return
    centers, counts, count, fallback = engine.state_arrays()
    dev = _collective_device()
    flat = torch.cat([torch.as_tensor(centers).reshape(-1), torch.as_tensor(counts).reshape(-1)]).to(dev)
    ints = torch.tensor([int(count), int(fallback)], dtype=torch.int64, device=dev)
    dist.broadcast(flat, src)
    dist.broadcast(ints, src)
    if rank != src:
        n = centers.size
        host = flat.cpu().numpy()
        engine.load_state_arrays(host[:n].reshape(centers.shape), host[n:], int(ints[0].item()), int(ints[1].item()))


def _collective_device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def broadcast_states(engines):
    """every clustering's state from the rank that trains it (view i -> rank i % world)"""
    rank, w = world()
    for i, v in enumerate(engines):
        broadcast_state(engines[v], i % w)


def train_epoch_view_parallel(engines, rows, b, lr, warm, broadcast=True):
    """One epoch of every clustering with the clusterings dealt out over the ranks (view i -> rank i % world).

    What several GPUs are FOR in this stage: the SGD chain of ONE clustering is a sequence of n/b dependent steps
    of a few microseconds each -- any per-step exchange between GPUs costs more than the step -- but the reference
    trains SEVERAL clusterings over the same rows (audio + visual, 5 + 5 layers in the real pipeline,
    run_clustering.py:32-44) and those chains are independent of each other.  Each rank runs its share of them at
    full single-GPU speed with the single-process arithmetic, then the owners broadcast their state: the result on
    every rank is bit-identical to a one-GPU run, with no collective on any step path.

    engines / rows / warm: same-order mappings view -> engine, resident rows, pre-drawn warm-up labels (drawn by
    every rank from the same stream, so the generators stay in step)."""
    rank, w = world()
    views = list(engines)
    mine = [v for i, v in enumerate(views) if i % w == rank]
    multi = getattr(type(engines[mine[0]]), "train_epoch_multi", None) if mine else None
    if multi is not None:  # this rank's share side by side on its GPU
        multi([engines[v] for v in mine], [rows[v] for v in mine], b, lr=lr, warm_bests=[warm[v] for v in mine])
    else:
        for v in mine:
            engines[v].train_epoch(rows[v], b, lr=lr, warm_best=warm[v])
    for i, v in enumerate(views):
        if i % w == rank:
            engines[v].synchronize()
        else:
            engines[v].skip_epoch(rows[v].shape[0] // b * b)  # keeps `count` (and hence the next warm-up plan) in step
    if broadcast:  # a streamed epoch calls this once per row group and exchanges the states at the end of the epoch
        broadcast_states(engines)
