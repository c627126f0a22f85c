"""GPU test of the multi-GPU code path with a 1-rank RCCL process group: the driver runs bench.py with
--gpus N on an 8-GPU node that this session cannot reach, so at least the collective plumbing (nccl init,
all_gather of rows + labels on device tensors, apply_update on the gathered batch, initialize()) is
executed on a real GPU here.  With world = 1 the result must equal the plain add()."""
import os

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
