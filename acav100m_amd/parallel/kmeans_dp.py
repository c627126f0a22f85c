"""Data-parallel KMeans.add() -- the reference's DDP branch (sgd_clustering.py:94-129 with
is_distributed) re-designed for exactness and xGMI.

Reference per step and clustering: all_gather([batch]) (only to take len()), all_reduce([counts]),
all_reduce([deltas]) = 2 x K*d*4 bytes of ring traffic.  Here each rank labels its local rows,
ONE all-gather moves the rows and labels (b*d*4 + b*8 bytes), and every rank applies the same
global-batch update with the deterministic in-order kernel.  The resulting state is bit-identical on
all ranks and equal to a single process fed the rank-major concatenation of the local batches.

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


def average_state(centers, counts):
    """KMeans.initialize(): all_reduce(SUM) then * 1/world (mps/distributed.py:139-155)."""
    rank, w = world()
    if w == 1:
        return centers, counts
    buf = torch.cat([centers.reshape(-1), counts.reshape(-1)])
    dist.all_reduce(buf)
    buf = buf * (1.0 / w)
    n = centers.numel()
    return buf[:n].reshape(centers.shape), buf[n:].reshape(counts.shape)
