"""world_size-2 tests (gloo, CPU) of the multi-GPU host logic: the per-step all-gather of rows+labels,
the identical global update on every rank, initialize()'s averaging, and the shard striding.
The compute engine is the oracle stand-in (no GPU here); on the GPU box the same plumbing drives the
HIP KMeans over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Engine:
    """host stand-in with the engine surface of acav100m_amd KMeans (calc_best / apply_update)"""

    def __init__(self, O, d, k, centers, counts, count):
        self.km = O.KMeans(d, k, O.Rng(0), centers=centers)
        self.km.set_state(None, counts, count)

    def calc_best(self, batch):
        best, mean = self.km.calc_best(np.asarray(batch))
        return torch.from_numpy(best), mean

    def apply_update(self, x, best, lr):
        self.km.apply_update(x.numpy(), best.numpy(), lr)

    # bulk surface used by train_epoch_dp
    def warmup_steps(self, b, steps):
        c, cnt, count, fb = self.km.get_state()
        lim = 10 * c.shape[0]
        return 0 if count >= lim else min(int(steps), -(-(lim - count) // int(b)))

    def draw_warmup(self, b):
        return torch.from_numpy(self.rng_labels.pop(0)[:b].copy())

    def train_epoch(self, x, b, lr, warm_best=None):
        x = x.numpy()
        nw = 0 if warm_best is None else len(warm_best)
        for t in range(len(x) // b):
            xb = x[t * b:(t + 1) * b]
            if t < nw:
                self.km.apply_update(xb, np.asarray(warm_best[t], np.int64), lr)
            else:
                self.km.add(xb, lr)

    def synchronize(self):
        pass

    # state exchange surface used by broadcast_state / train_epoch_view_parallel
    def state_arrays(self):
        return self.km.get_state()

    def load_state_arrays(self, centers, counts, count, fallback):
        self.km.set_state(np.ascontiguousarray(centers, np.float32), np.ascontiguousarray(counts, np.float32), int(count))

    def skip_epoch(self, rows):
        c, cnt, count, fb = self.km.get_state()
        self.km.set_state(None, None, count + int(rows))


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import oracle as O
        from acav100m_amd.parallel import average_state, distributed_add, shard_slice, world as wfn
        assert wfn() == (rank, world)
        d, k, b_local, steps = 48, 12, 16, 12
        rs = np.random.RandomState(0)
        cen = rs.randn(k, d).astype(np.float32)
        x = (cen[rs.randint(0, k, steps * world * b_local)] + 0.3 * rs.randn(steps * world * b_local, d)).astype(np.float32)
        # initialize(): ranks start from different centres, end with the average
        c0 = torch.from_numpy((cen + rank).astype(np.float32))
        n0 = torch.full((k,), float(rank))
        c_avg, n_avg = average_state(c0, n0)
        assert torch.allclose(c_avg, torch.from_numpy(cen) + 0.5) and torch.allclose(n_avg, torch.full((k,), 0.5))
        counts = np.full(k, 40, np.float32)
        eng = _Engine(O, d, k, cen, counts, 10 * k + 100)
        for t in range(steps):
            gb = x[t * world * b_local:(t + 1) * world * b_local]          # global batch, rank-major
            distributed_add(eng, torch.from_numpy(gb[rank * b_local:(rank + 1) * b_local]), 0.01)
        c, cnt, count, fb = eng.km.get_state()
        np.savez(os.path.join(tmp, f"rank{rank}.npz"), c=c, cnt=cnt, count=count)
        if rank == 0:  # single-process reference on the concatenated batches
            ref = _Engine(O, d, k, cen, counts, 10 * k + 100)
            for t in range(steps):
                ref.km.add(x[t * world * b_local:(t + 1) * world * b_local])
            rc, rcnt, rcount, _ = ref.km.get_state()
            np.savez(os.path.join(tmp, "single.npz"), c=rc, cnt=rcnt, count=rcount)
        # bulk-gathered epoch (train_epoch_dp) == the per-step path above == single process, warm-up included
        from acav100m_amd.parallel import train_epoch_dp
        steps2, bl = 9, 8
        xl = (cen[rs.randint(0, k, steps2 * bl)] + 0.3 * np.random.RandomState(100 + rank).randn(steps2 * bl, d)).astype(np.float32)
        lab_rs = np.random.RandomState(7 + rank)
        eng2 = _Engine(O, d, k, cen, np.zeros(k, np.float32), 0)       # count 0: the first steps are warm-up
        eng2.rng_labels = [lab_rs.randint(0, k, bl).astype(np.int64) for _ in range(steps2)]
        my_labels = [a.copy() for a in eng2.rng_labels]
        train_epoch_dp(eng2, torch.from_numpy(xl), bl, 0.01, chunk_steps=4)
        c2, cnt2, count2, _ = eng2.km.get_state()
        np.savez(os.path.join(tmp, f"dp_rank{rank}.npz"), c=c2, cnt=cnt2, count=count2, x=xl, lab=np.stack(my_labels))
        # the same epoch with ONE trainer (rank 1): the rows are gathered to it only, the other rank follows by broadcast
        from acav100m_amd.parallel import broadcast_state
        eng3 = _Engine(O, d, k, cen, np.zeros(k, np.float32), 0)
        eng3.rng_labels = [a.copy() for a in my_labels]
        train_epoch_dp(eng3, torch.from_numpy(xl), bl, 0.01, chunk_steps=4, trainer=1)
        if rank == 0:  # a rank that only sends has not touched its state
            c3, cnt3, _, _ = eng3.km.get_state()
            assert np.array_equal(c3, cen) and not cnt3.any()
        broadcast_state(eng3, 1)
        c3, cnt3, count3, _ = eng3.km.get_state()
        assert np.array_equal(c3, c2) and np.array_equal(cnt3, cnt2) and count3 == count2, f"rooted epoch, rank {rank}"
        assert list(shard_slice(7)) == list(range(rank, 7, world))
        # view-parallel epochs (the CLI's multi-GPU mode): 3 clusterings dealt out over 2 ranks, 2 epochs; every rank
        # must end with the state a single process reaches for every clustering
        from acav100m_amd.parallel import train_epoch_view_parallel
        nv, bv, nrows = 3, 8, 96
        rsv = np.random.RandomState(5)
        rows = {v: torch.from_numpy((cen[rsv.randint(0, k, nrows)] + 0.3 * rsv.randn(nrows, d)).astype(np.float32)) for v in range(nv)}
        warm_all = {v: rsv.randint(0, k, (2, nrows // bv, bv)).astype(np.int64) for v in range(nv)}
        engs = {v: _Engine(O, d, k, cen + np.float32(0.01 * v), np.zeros(k, np.float32), 0) for v in range(nv)}
        refs = {v: _Engine(O, d, k, cen + np.float32(0.01 * v), np.zeros(k, np.float32), 0) for v in range(nv)}
        for epoch in range(2):
            need = {v: engs[v].warmup_steps(bv, nrows // bv) for v in range(nv)}
            warm = {v: warm_all[v][epoch][:need[v]] for v in range(nv)}
            train_epoch_view_parallel(engs, rows, bv, 0.01, warm)
            for v in range(nv):
                assert refs[v].warmup_steps(bv, nrows // bv) == need[v]
                refs[v].train_epoch(rows[v], bv, 0.01, warm_best=warm[v])
        for v in range(nv):
            a, b2 = engs[v].km.get_state(), refs[v].km.get_state()
            assert np.array_equal(a[0], b2[0]) and np.array_equal(a[1], b2[1]) and a[2] == b2[2], f"view {v} rank {rank}"
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_distributed_add_two_ranks(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1, s = (np.load(tmp_path / f) for f in ("rank0.npz", "rank1.npz", "single.npz"))
    assert np.array_equal(r0["c"], r1["c"]) and np.array_equal(r0["cnt"], r1["cnt"])      # ranks identical
    assert np.array_equal(r0["c"], s["c"]) and np.array_equal(r0["cnt"], s["cnt"])        # == single process
    # train_epoch_dp: ranks identical, and equal to one process fed the rank-major global batches
    from oracle import oracle as O
    d0, d1 = (np.load(tmp_path / f) for f in ("dp_rank0.npz", "dp_rank1.npz"))
    assert np.array_equal(d0["c"], d1["c"]) and np.array_equal(d0["cnt"], d1["cnt"]) and d0["count"] == d1["count"]
    k, dd = d0["c"].shape
    rs = np.random.RandomState(0)
    cen = rs.randn(k, dd).astype(np.float32)
    ref = O.KMeans(dd, k, O.Rng(0), centers=cen)
    ref.set_state(None, np.zeros(k, np.float32), 0)
    steps2, bl = d0["lab"].shape
    for t in range(steps2):
        xb = np.concatenate([d0["x"][t * bl:(t + 1) * bl], d1["x"][t * bl:(t + 1) * bl]])
        c, cnt, count, _ = ref.get_state()
        if count < 10 * k:
            ref.apply_update(xb, np.concatenate([d0["lab"][t], d1["lab"][t]]), 0.01)
        else:
            ref.add(xb, 0.01)
    rc, rcnt, rcount, _ = ref.get_state()
    assert np.array_equal(d0["c"], rc) and np.array_equal(d0["cnt"], rcnt) and int(d0["count"]) == rcount
    assert int(r0["count"]) == int(s["count"])
