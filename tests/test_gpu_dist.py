"""GPU tests of the multi-GPU code path on ONE GPU: a 1-rank RCCL process group (nccl init, the bulk row exchange to the
trainer rank + warm-up labels through the library's communicator, the state hand-out, initialize()) -- with world = 1 the
result must equal the plain epoch -- and two ranks sharing the GPU under gloo, including bench.py --gpus 2 launched through
torch.distributed.run as the driver launches it.  (The per-step all-gather of rows and labels of round 2 survives only in
distributed_add(), the KMeans.add() form.)"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_distributed_add_world1_rccl():
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import acav100m_amd
    from acav100m_amd.clustering import KMeans
    from oracle import oracle as O
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        class NS:
            pass
        args = NS()
        args.computation = NS()
        args.computation.device = "cuda"
        args.computation.num_gpus = 2  # forces the distributed branch (is_distributed)
        d, k, b, steps = 64, 12, 32, 20
        rs = np.random.RandomState(0)
        x = rs.randn(steps * b, d).astype(np.float32)
        acav100m_amd.manual_seed(4)
        km = KMeans(args, d, k).to("cuda:0")
        km.initialize()
        assert km.is_distributed
        ref = O.KMeans(d, k, O.Rng(4))
        xt = torch.from_numpy(x).cuda()
        for t in range(steps):
            m = km.add(xt[t * b:(t + 1) * b])
            m_ref = ref.add(x[t * b:(t + 1) * b])
            assert abs(m - m_ref) <= 1e-5 * abs(m_ref) + 1e-30
        assert np.array_equal(km.centers.numpy(), ref.centers)
        assert np.array_equal(km.counts.numpy(), ref.counts) and km.count == ref.count
        # the bulk-gathered epoch (what bench.py --gpus N runs): all_gather_into_tensor of row chunks + warm-up
        # labels over RCCL, then the device-resident epoch over the global batches; with one rank == plain epoch
        from acav100m_amd.parallel import train_epoch_dp
        d2, k2, b2, steps2 = 256, 40, 32, 100
        x2 = (rs.randn(k2, d2)[rs.randint(0, k2, steps2 * b2)] * 3 + rs.randn(steps2 * b2, d2)).astype(np.float32)
        acav100m_amd.manual_seed(9)
        km2 = KMeans(args, d2, k2).to("cuda:0")
        train_epoch_dp(km2, torch.from_numpy(x2).cuda(), b2, 0.01, chunk_steps=16, force_collective=True)
        ref2 = O.KMeans(d2, k2, O.Rng(9))
        ref2.train_epoch(x2, b2, 0.01)
        assert np.array_equal(km2.centers.numpy(), ref2.centers)
        assert np.array_equal(km2.counts.numpy(), ref2.counts) and km2.count == ref2.count
    finally:
        dist.destroy_process_group()


def _two_rank_worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)  # both ranks share cuda:0; gloo moves the bytes
    try:
        import acav100m_amd
        from acav100m_amd.clustering import KMeans

        class NS:
            pass
        args = NS()
        args.computation = NS()
        args.computation.device = "cuda"
        args.computation.num_gpus = world
        d, k, b, steps = 256, 24, 16, 40
        rs = np.random.RandomState(100 + rank)
        x = (np.random.RandomState(5).randn(k, d)[rs.randint(0, k, steps * b)] * 3 + rs.randn(steps * b, d)).astype(np.float32)
        acav100m_amd.manual_seed(11 + rank)           # every rank draws ITS warm-up labels from its own stream
        km = KMeans(args, d, k).to("cuda:0")
        km.centers = np.random.RandomState(5).randn(k, d).astype(np.float32) * 1e-5   # same start on all ranks
        km.initialize()
        km.train_epoch_distributed(torch.from_numpy(x).cuda(), b, lr=0.01, chunk_steps=16)
        np.savez(os.path.join(tmp, f"r{rank}.npz"), c=km.centers.numpy(), n=km.counts.numpy(), count=km.count, x=x)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_bulk_gathered_epoch(tmp_path):
    """The multi-GPU epoch end to end with TWO processes and real GPU engines (both on cuda:0, gloo carrying the
    collectives): identical state on both ranks, equal to one process fed the rank-major global batches -- warm-up
    labels of each rank's own generator included."""
    import socket
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert np.array_equal(r0["c"], r1["c"]) and np.array_equal(r0["n"], r1["n"]) and r0["count"] == r1["count"]
    from oracle import oracle as O
    d, k, b, steps = 256, 24, 16, 40
    ref = O.KMeans(d, k, O.Rng(0), centers=np.random.RandomState(5).randn(k, d).astype(np.float32) * 1e-5)
    g0, g1 = O.Rng(11), O.Rng(12)
    # the product draws torch.rand(k, d) * 1e-5 at construction from each rank's generator before the warm-up labels
    g0.rand(k * d), g1.rand(k * d)
    for t in range(steps):
        xb = np.concatenate([r0["x"][t * b:(t + 1) * b], r1["x"][t * b:(t + 1) * b]])
        c, cnt, count, _ = ref.get_state()
        if count < 10 * k:
            lab = np.concatenate([g0.rand(k, b).argmin(0), g1.rand(k, b).argmin(0)]).astype(np.int64)  # sgd_clustering.py:67-68,78
            ref.apply_update(xb, lab, 0.01)
        else:
            ref.add(xb, 0.01)
    rc, rcnt, rcount, _ = ref.get_state()
    assert np.array_equal(r0["c"], rc) and np.array_equal(r0["n"], rcnt) and int(r0["count"]) == rcount


def test_comm_c_abi_world1():
    """The C-ABI communicator (acav_comm_*, RCCL resolved at run time) with a world of one on the real GPU: init from a
    unique id, all-reduce / all-gather / broadcast are identities, acav_kmeans_allreduce_init leaves the state bit for
    bit, and acav_kmeans_train_dp (bulk row exchange + interleave kernel + device-resident epoch, 3 chunks, warm-up
    inside) == the plain epoch == the oracle."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ctypes as C
    import acav100m_amd
    from acav100m_amd import _lib
    from acav100m_amd.clustering import KMeans
    from acav100m_amd.parallel.rccl_comm import Comm
    from oracle import oracle as O
    comm = Comm(0, 1, Comm.unique_id(), 0)
    t = torch.arange(1000, dtype=torch.float32, device="cuda")
    want = t.clone()
    comm.allreduce_(t)
    out = torch.empty_like(t)
    comm.allgather(t, out)
    comm.broadcast_(t, 0)
    comm.synchronize()
    assert torch.equal(t, want) and torch.equal(out, want)
    d, k, b, steps = 256, 40, 32, 40
    rs = np.random.RandomState(1)
    x = (rs.randn(k, d)[rs.randint(0, k, steps * b)] * 3 + rs.randn(steps * b, d)).astype(np.float32)
    acav100m_amd.manual_seed(9)
    km = KMeans(None, d, k).to("cuda:0")
    before = km.centers.numpy().copy()
    _lib.check(_lib._lib.acav_kmeans_allreduce_init(km._h, comm._h))
    assert np.array_equal(km.centers.numpy(), before)
    ref = O.KMeans(d, k, O.Rng(9))
    xt = torch.from_numpy(x).cuda()
    for epoch in range(2):
        km.train_epoch_comm(comm, xt, b, 0.01, chunk_steps=16)
        ref.train_epoch(x, b, 0.01)
        _lib.check(_lib._lib.acav_kmeans_broadcast_state(km._h, comm._h, 0))  # the root keeps its state
        assert np.array_equal(km.centers.numpy(), ref.centers), f"epoch {epoch}"
        assert np.array_equal(km.counts.numpy(), ref.counts) and km.count == ref.count
    # the multi-GPU bench's calls: a communicator per clustering, nothing waits until everything is enqueued, then the
    # trainer hands out its state -- with one rank the result is one more plain epoch
    km.train_epoch_distributed(xt, b, lr=0.01, train_here=True, comm_slot=1, wait=False)
    km.broadcast_state_from(0, comm_slot=1)
    ref.train_epoch(x, b, 0.01)
    assert np.array_equal(km.centers.numpy(), ref.centers) and km.count == ref.count
    # the rows SENT to the one trainer (ACAV_DP_ROOTED: grouped ncclSend / ncclRecv instead of the all-gather)
    km.train_epoch_comm(comm, xt, b, 0.01, chunk_steps=16, train_here=True, trainer=0)
    ref.train_epoch(x, b, 0.01)
    assert np.array_equal(km.centers.numpy(), ref.centers) and km.count == ref.count
    # a rank that only takes part in the row exchange (train_here = 0) leaves its state alone
    before = (km.centers.numpy().copy(), km.count)
    km.train_epoch_comm(comm, xt, b, 0.01, chunk_steps=16, train_here=False)
    assert np.array_equal(km.centers.numpy(), before[0]) and km.count == before[1]
    # several clusterings chunk by chunk (acav_kmeans_train_dp_multi, what bench.py --gpus N calls): with one rank == one
    # plain epoch each; a clustering dealt to another rank (trainer 1 in a world of one: nobody) only feeds its exchange
    d2 = 128
    x2 = (rs.randn(k, d2)[rs.randint(0, k, steps * b)] * 3 + rs.randn(steps * b, d2)).astype(np.float32)
    x2t = torch.from_numpy(x2).cuda()
    acav100m_amd.manual_seed(11)
    ka, kb = KMeans(None, d, k).to("cuda:0"), KMeans(None, d2, k).to("cuda:0")
    ra, rb = O.KMeans(d, k, O.Rng(11)), None
    rb = O.KMeans(d2, k, ra.rng)  # the two clusterings share the generator, as ka / kb share the library's
    for epoch in range(2):
        # chunks of 8 steps: the 13 warm-up steps of the first epoch span two of them
        tr = KMeans.train_epoch_distributed_multi([ka, kb], [xt, x2t], b, lr=0.01, chunk_steps=8)
        assert tr == [0, 0]
        for v, kmv in enumerate((ka, kb)):
            kmv.broadcast_state_from(tr[v], comm_slot=v)
        ra.train_epoch(x, b, 0.01)
        rb.train_epoch(x2, b, 0.01)
        assert np.array_equal(ka.centers.numpy(), ra.centers) and ka.count == ra.count, f"epoch {epoch}"
        assert np.array_equal(kb.centers.numpy(), rb.centers) and kb.count == rb.count, f"epoch {epoch}"
    before = (kb.centers.numpy().copy(), kb.count, ka.count)
    KMeans.train_epoch_distributed_multi([ka, kb], [xt, x2t], b, lr=0.01, chunk_steps=16, trainers=[0, 1])
    assert np.array_equal(kb.centers.numpy(), before[0]) and kb.count == before[1] and ka.count == before[2] + steps * b


def test_plan_driven_epoch_c_abi_world1():
    """acav_kmeans_train_plan_multi (the row exchange of clustering.multi_gpu=reference / bench.py --gpus N inside the
    library: pack kernel, grouped send / receive on the clustering's communicator, placement kernel, chunk pipeline,
    device-resident chain) on the one GPU of this box: a communicator of ONE rank, but plans with SEVERAL slots whose
    extents all name rank 0 -- rotated streams over four ragged "segments" of the local rows, as plan_reference lays them
    out, and the one-slot shard-order stream of plan_views.  Two clusterings per call (communicator per clustering),
    chunks of 7 steps (the warm-up spans chunks; pieces are cut by chunk AND extent boundaries).  The state must equal the
    oracle fed the plan's global batches row for row."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import acav100m_amd
    from acav100m_amd.clustering import KMeans
    from acav100m_amd.parallel.rccl_comm import default_comm
    from acav100m_amd.parallel.row_plan import RowPlan
    from oracle import oracle as O
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if default_comm(0) is None:
        pytest.skip("RCCL not available")
    rs = np.random.RandomState(2)
    k, b = 24, 32
    seg = [200, 136, 168, 152]  # four ragged segments laid out one after the other in this rank's rows
    first = np.concatenate([[0], np.cumsum(seg)[:-1]])
    n = sum(seg)
    dims = (256, 128)
    xs = [(rs.randn(k, d)[rs.randint(0, k, n)] * 3 + rs.randn(n, d)).astype(np.float32) for d in dims]
    xt = [torch.from_numpy(x).cuda() for x in xs]
    plans = []
    lb = b // 4
    rot = [[(0, int(first[i % 4]), seg[i % 4]) for i in range(q, q + 4)] for q in range(4)]
    plans.append(RowPlan("reference-like", 1, 4, lb, rot, n // lb, 1))
    plans.append(RowPlan("views-like", 1, 1, b, [[(0, int(first[i]), seg[i]) for i in (2, 0, 3, 1)]], n // b, 1))
    plans.append(RowPlan("rows-like", 1, 2, b, [[(0, 0, 336)], [(0, 336, 320)]], 320 // b, 1))
    for plan in plans:
        acav100m_amd.manual_seed(21)
        kms = [KMeans(None, d, k).to("cuda:0") for d in dims]
        rng = O.Rng(21)
        refs = [O.KMeans(d, k, rng) for d in dims]
        for epoch in range(2):
            # one rank: the labels of a warm-up step are argmin torch.rand(k, global batch), clustering by clustering
            tr = KMeans.train_epoch_plan_multi(kms, xt, plan, lr=0.01, chunk_steps=7)
            assert tr == [0, 0]
            for v, km in enumerate(kms):
                km.broadcast_state_from(0, comm_slot=v)
            for ref, x in zip(refs, xs):
                need = min(plan.steps, max(0, -(-(10 * k - ref.count) // plan.global_batch)))
                warm = [np.argmin(rng.rand(k, plan.global_batch), axis=0) for _ in range(need)]
                for t in range(plan.steps):
                    xb = np.stack([x[row] for _owner, row in plan.batch_sources(t)])
                    if t < need:
                        ref.apply_update(xb, warm[t].astype(np.int64), 0.01)
                    else:
                        ref.add(xb, 0.01)
            for km, ref in zip(kms, refs):
                assert np.array_equal(km.centers.numpy(), ref.centers), (plan.mode, epoch)
                assert np.array_equal(km.counts.numpy(), ref.counts) and km.count == ref.count, (plan.mode, epoch)
        # a clustering whose trainer is another rank (nobody in a world of one) only feeds its exchange
        before = (kms[1].centers.numpy().copy(), kms[1].count, kms[0].count)
        KMeans.train_epoch_plan_multi(kms, xt, plan, lr=0.01, chunk_steps=16, trainers=[0, 1])
        assert np.array_equal(kms[1].centers.numpy(), before[0]) and kms[1].count == before[1] + plan.steps * plan.global_batch
        assert kms[0].count == before[2] + plan.steps * plan.global_batch


def test_bench_two_ranks_on_one_gpu():
    """bench.py's N > 1 path end to end, launched the way the driver launches it (torch.distributed.run, one process per
    rank) with two ranks sharing the one GPU of this box (ACAV_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device):
    plan-driven training epochs with the bulk row exchange (every --multi-gpu mode), the state hand-out, per-rank assign
    and selection, the MAX-over-ranks timing, --verify (state hashes agree across the ranks, rank 0's labels == the
    oracle's on 16 384 rows) and ONE JSON line from rank 0."""
    import json
    import socket
    import subprocess
    import sys
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    rows = 65536
    want = {"views": (32, 2 * rows // 32, 2), "reference": (32, 2 * 2 * rows // 32, 1), "rows": (64, rows // 32, 1)}
    for mode, (gb, steps, epochs) in want.items():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ, ACAV_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
               "--rows", str(rows), "--no-cpu-baseline", "--no-variants", "--verify", "--multi-gpu", mode]
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["steps"] == 1 and out["scaling"] == "weak" and out["unit"] == "clips/s"
        cfg = out["config"]
        assert cfg["multi_gpu_mode"] == mode and cfg["rows_per_gpu"] == rows
        assert (cfg["global_batch"], cfg["sgd_steps_per_epoch"], cfg["train_epochs"]) == (gb, steps, epochs), cfg
        assert mode in cfg["workload"]
        assert out["value"] > 0 and abs(out["value"] - 2 * rows / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
        assert 0 < out["roofline"]["frac"] < 1 and out["roofline"]["rows"] == rows
        v = out["verify"]
        assert v["ranks_agree"] and v["oracle_sample_rows"] == 16384 and v["oracle_labels_equal"] == [True, True], v


def test_bench_verify_one_gpu():
    """bench.py --verify on one GPU (the driver's N = 1 launch shape, small): the verdict object is in the line"""
    import json
    import subprocess
    import sys
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--rows", "65536", "--no-cpu-baseline",
           "--no-variants", "--verify"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["config"]["multi_gpu_mode"] is None and out["config"]["global_batch"] == 32
    assert out["verify"]["ranks_agree"] and out["verify"]["oracle_labels_equal"] == [True, True]
