import torch
import torch.distributed as dist


def world():
    """(rank, world_size): the process group when there is one, else the launcher environment (the selection
    stage's per-GPU processes never communicate and do not form a group), else (0, 1)."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    import os
    if os.environ.get("ACAV_NO_GROUP") == "1" and "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        return int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    return 0, 1


def gather_rows_and_labels(batch, best, device_index=None):
    """All-gather the rank-local rows [b_local,d] and labels [b_local] into the global batch in
    rank-major order (rank 0's rows first).  Every rank must pass the same b_local."""
    rank, w = world()
    x = batch.detach().contiguous()
    lab = best.detach().to(torch.long).contiguous()
    if x.device != lab.device:
        lab = lab.to(x.device)
    if w == 1:
        return x, lab
    xs = [torch.empty_like(x) for _ in range(w)]
    ls = [torch.empty_like(lab) for _ in range(w)]
    dist.all_gather(xs, x)
    dist.all_gather(ls, lab)
    return torch.cat(xs, 0), torch.cat(ls, 0)


def shard_slice(n_items, rank=None, world_size=None):
    """Strided ownership rank::world of the reference's assign / chunk loops
    (mps/distributed.py:439, chunk.py:26,40-41)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return range(rank, n_items, world_size)
