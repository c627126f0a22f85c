"""How large is the UNION of candidate centres over the undecided rows that share a filter minimum k1?  (VERDICT r4 item 5 proposes
one dense exact tile per k1 group over that union and assumes ~32 centres.)  Data and centres as bench.py's assign_hard_variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import acav100m_amd
acav100m_amd.configure_runtime(quiet=True)
import numpy as np, torch
from acav100m_amd.clustering import KMeans
dev = "cuda:0"
n, d, k, b = 1_000_000, 1024, 256, 32
for spread in (0.06, 0.05):
    gen = torch.Generator(device=dev).manual_seed(7)
    cen = spread * torch.randn(k, d, device=dev, generator=gen)
    comp = torch.randint(0, k, (n,), device=dev, generator=gen)
    x = torch.empty(n, d, device=dev)
    for s in range(0, n, 65536):
        e = min(n, s + 65536)
        x[s:e] = cen[comp[s:e]] + 0.3 * torch.randn(e - s, d, device=dev, generator=gen)
    acav100m_amd.manual_seed(3)
    km = KMeans(None, d, k).to(dev)
    km.train_epoch(x[:262144], b, lr=0.01)
    lab = km.calc_best(x, need_mean=False)[0]
    _, _, und = km.filter_stats()
    rows, pairs, full = km.recheck_stats()
    c = km.centers.to(dev)
    m = 200_000
    xs = x[:m]
    dist = (xs * xs).sum(1, keepdim=True) - 2 * xs @ c.T + (c * c).sum(1)[None]
    dmin, k1 = dist.min(1)
    gap = dist - dmin[:, None]
    # window calibrated so that the share of rows with >= 2 candidates equals the filter's undecided share
    target = und / n
    second = gap.kthvalue(2, dim=1).values
    w = torch.quantile(second, target).item() if 0 < target < 1 else 0.0
    cand = gap <= w
    undecided = cand.sum(1) >= 2
    ppr = cand[undecided].sum(1).float().mean().item()
    union = torch.zeros(k, k, dtype=torch.bool, device=dev)
    union.index_put_((k1[undecided].repeat_interleave(cand[undecided].sum(1)), cand[undecided].nonzero()[:, 1]), torch.tensor(True, device=dev), accumulate=False)
    us = union.sum(1).float()
    rows_per_group = torch.bincount(k1[undecided], minlength=k).float()
    print(f"spread {spread}: filter undecided {und} of {n} ({100 * und / n:.1f} %), {pairs / max(rows, 1):.2f} pairs per row; probe window {w:.3f}: "
          f"{ppr:.2f} pairs per undecided row; union of candidates per k1 group over {m} rows: mean {us.mean():.0f} median {us.median():.0f} "
          f"max {us.max():.0f} of {k} centres; undecided rows per group mean {rows_per_group.mean():.0f}", flush=True)
