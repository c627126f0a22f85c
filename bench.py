#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: clips/sec curated by the k-means hot path.

Workload (config.workload): BASELINE.json configs[1] -- 1M clips, 1024-d features, K=256,
k-means only (update + assign), one GPU.  One bench "step" = one k-means pass over the resident
feature matrix exactly as the reference runs it: one training epoch of KMeans.add at the
reference's batch size b=32 (floor(N/32) sequentially dependent SGD steps,
run_clustering.py:229-241) followed by one assign sweep (KMeans.calc_best over all N rows,
run_clustering.py:290-296).  value = N * n_gpus * steps / time, inputs resident in HBM.

    python bench.py                      # 1 GPU, defaults
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

N > 1 (weak scaling): every rank holds its own N-row shard; assign is embarrassingly parallel;
training uses the distributed add() (per step: local labels for 32 local rows, one RCCL
all-gather of rows+labels, identical global update on every rank -- i.e. the reference run with
--data.batch_size=32*W).

The JSON line also carries
  roofline      for the assign sweep = k_assign_bf16 (bf16-MFMA filter over all rows) + k_assign_f32 (exact
                fp32 re-check of the rows whose top-2 gap is below the proven error bound; labels are
                bit-identical to the all-exact path): algorithmic bytes N*d*4 + N*8 per sweep over the
                sweep duration measured with HIP events on the library's stream; bound = HBM.
  cpu_baseline  the oracle (oracle/libacav_oracle.so, a C port of the reference algorithm) timed on
                this box's host cores on a bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # f32-input MFMA = fp32 vector peak


def synth_features(torch, n, d, k, seed, device):
    """SURVEY 8(d) generator: K Gaussian components, centre ~ N(0,1)^d, row = centre + 0.3 N(0,1)."""
    gen = torch.Generator(device=device).manual_seed(seed)
    cen = torch.randn(k, d, device=device, generator=gen)
    comp = torch.randint(0, k, (n,), device=device, generator=gen)
    x = torch.empty(n, d, device=device, dtype=torch.float32)
    step = 65536
    for s in range(0, n, step):
        e = min(n, s + step)
        x[s:e] = cen[comp[s:e]] + 0.3 * torch.randn(e - s, d, device=device, generator=gen)
    return x


def cpu_baseline(n, d, k, b, seed, budget_s=20.0):
    """Oracle on the host cores, bounded sample: SGD steps (after the warm-up) + an assign slice.
    add() is row-parallel over only b=32 rows, so it runs on min(cores, 16) threads (more threads only
    add fork/join overhead); the assign slice uses every core.  Best of 3 repetitions each."""
    from oracle import oracle as O
    rs = np.random.RandomState(seed)
    cen = rs.randn(k, d).astype(np.float32)
    n_steps, n_assign_rows = 48, 16384
    xs = (cen[rs.randint(0, k, n_steps * b + n_assign_rows)] +
          0.3 * rs.randn(n_steps * b + n_assign_rows, d)).astype(np.float32)
    cores = os.cpu_count() or 1
    train_threads = min(cores, 16)
    t_train, t_assign = float("inf"), float("inf")
    for rep in range(3):
        km = O.KMeans(d, k, O.Rng(seed), centers=(cen + 0.1 * rs.randn(k, d)).astype(np.float32))
        km.set_state(None, np.full(k, 50.0, np.float32), 10 * k + 12800)
        O.set_threads(train_threads)
        t0 = time.perf_counter()
        for t in range(n_steps):
            km.add(xs[t * b:(t + 1) * b])
        t_train = min(t_train, (time.perf_counter() - t0) / (n_steps * b))   # s per clip
        O.set_threads(cores)
        t0 = time.perf_counter()
        km.calc_best(xs[n_steps * b:])
        t_assign = min(t_assign, (time.perf_counter() - t0) / n_assign_rows)  # s per clip
    return {
        "value": 1.0 / (t_train + t_assign), "unit": "clips/s", "cores": cores, "kind": "port",
        "sample": f"best of 3: {n_steps} add() steps of b={b} on {train_threads} threads + calc_best over "
                  f"{n_assign_rows} rows on {cores} threads, d={d}, K={k} (oracle C port, OpenMP over rows); "
                  f"per-clip times summed and inverted",
        "train_clips_per_s": 1.0 / t_train, "assign_clips_per_s": 1.0 / t_assign,
    }


def mi_stage(seed, v=100_000, c=256, oracle_iters=100):
    """Informational (outside the timed region): one chunk of the greedy MI selection -- BASELINE configs[2]'s
    second stage at the reference's chunk granularity -- on the GPU, and the oracle's loop on the host."""
    import itertools
    import acav100m_amd
    from acav100m_amd.subset_selection import get_measure
    rs = np.random.RandomState(seed)
    comp = rs.randint(0, c, v)
    a = np.stack([np.where(rs.rand(v) < 0.5, comp, rs.randint(0, c, v)) for _ in range(2)], 1).astype(np.int64)
    a[0] = c - 1
    pairs = list(itertools.combinations(range(2), 2))
    cand = [int(i) for i in rs.permutation(v)]
    subset = round(0.2 * v)
    acav100m_amd.manual_seed(seed)
    import contextlib
    import io
    m = get_measure("batch_mi")(a, ncentroids=c, batch_size=20, selection_size=4, device="cuda", keep_unselected=True)
    m.init(pairs, cand[1:])
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        S, _, _, _ = m.run_greedy(subset, cand[:1], None)
        dt = time.perf_counter() - t0
    iters = (subset + 3) // 4
    out = {"workload": f"one chunk: V={v}, D=2, C={c}, select 20% (B=20, k=4)", "iterations": iters,
           "us_per_iteration": dt / iters * 1e6, "selected_clips_per_s": len(S) / dt, "curated_clips_per_s": v / dt,
           "permutation_stream_GBs": sum(16 * (v - 1 - 4 * t) for t in range(iters)) / dt / 1e9}
    # the same chunk size with 8 chunks in lockstep (computation.concurrent_chunks): aggregate per chunk-iteration
    try:
        from acav100m_amd.rng import Generator
        from acav100m_amd.subset_selection.measures.batch import EfficientBatchMI
        ms = []
        for i in range(8):
            mi = get_measure("batch_mi")(a, ncentroids=c, batch_size=20, selection_size=4, device="cuda",
                                         keep_unselected=True, generator=Generator(seed + 1 + i))
            mi.init(pairs, cand[1:])
            ms.append(mi)
        with contextlib.redirect_stdout(io.StringIO()):
            t0 = time.perf_counter()
            EfficientBatchMI.run_greedy_multi(ms, [subset] * 8, [cand[:1]] * 8)
            dt8 = time.perf_counter() - t0
        out["lockstep_8_chunks"] = {"us_per_chunk_iteration": dt8 / (8 * iters) * 1e6, "curated_clips_per_s": 8 * v / dt8}
        del ms
    except Exception as exc:
        out["lockstep_8_chunks"] = {"error": str(exc)}
    try:
        from oracle import oracle as O
        O.set_threads(1)
        t0 = time.perf_counter()
        O.BatchMI(a, c, pairs).run_greedy(cand[1:], cand[:1], subset, 20, 4, O.Rng(seed), max_iters=oracle_iters)
        out["cpu_port_us_per_iteration"] = (time.perf_counter() - t0) / oracle_iters * 1e6
    except Exception as exc:  # the oracle is optional here
        out["cpu_port_us_per_iteration"] = None
        out["cpu_port_error"] = str(exc)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=1024)
    ap.add_argument("--k", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("ACAV_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("ACAV_BENCH_BACKEND", "nccl")  # "gloo": several ranks on ONE GPU (plumbing check)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import acav100m_amd
    from acav100m_amd import _lib
    from acav100m_amd.clustering import KMeans
    lib = acav100m_amd.load_library()

    n, d, k, b = args.n, args.d, args.k, args.batch
    x = synth_features(torch, n, d, k, 1234 + rank, dev)
    torch.cuda.synchronize()
    acav100m_amd.manual_seed(0)

    class _A:  # the reference's args.computation view
        pass
    cargs = _A()
    cargs.computation = _A()
    cargs.computation.device = "cuda"
    cargs.computation.num_gpus = world
    km = KMeans(cargs, d, k).to(dev)
    km.initialize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    labels = torch.empty(n, dtype=torch.long, device=dev)
    assign_ms = []

    def one_pass(timed):
        # --- update: one epoch of add() at b=32
        if world == 1:
            km.train_epoch(x, b, lr=0.01)
        else:  # reference DDP semantics: global batch = world * b rows per step; rows all-gathered in bulk
            km.train_epoch_distributed(x, b, lr=0.01)
        # --- assign sweep, timed with HIP events on the library's own stream
        _lib.check(lib.acav_kmeans_timer_begin(km._h))
        _lib.check(lib.acav_kmeans_assign(km._h, _lib.ptr(x), n, _lib.ptr(labels), None))
        ms = C.c_float(0)
        _lib.check(lib.acav_kmeans_timer_end(km._h, C.byref(ms)))
        if timed:
            assign_ms.append(ms.value)

    for _ in range(args.warmup):
        one_pass(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = n * world * args.steps / elapsed
        a_ms = float(np.mean(assign_ms))
        bytes_per_launch = n * d * 4 + n * 8
        flops_per_launch = 2.0 * n * k * d
        gbs = bytes_per_launch / (a_ms * 1e-3) / 1e9
        tfs = flops_per_launch / (a_ms * 1e-3) / 1e12
        traffic, traffic_src = None, None
        for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
            if name.endswith("_pmc_assign.json"):  # summary of the separate rocprofv3 --pmc passes of this command
                pm = json.load(open(os.path.join(ROOT, "profiles", name)))
                if (pm.get("rows"), pm.get("d"), pm.get("K")) == (n, d, k) and pm.get("kernel", "").startswith("k_assign_bf16"):
                    traffic, traffic_src = pm["traffic_bytes_per_launch"], "profiles/" + name
                    break
        fl, frows, frecheck = km.filter_stats()
        out = {
            "metric": "clips/sec curated (k-means update epoch at b=32 + assign sweep)",
            "value": value, "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: {n} clips x {d}-d, K={k}, k-means only "
                                   f"(1 training epoch at b={b} = {n // b} SGD steps + 1 assign sweep per step)",
                       "global_batch": b * world, "rows_per_gpu": n},
            "roofline": {"kernel": "k_assign_bf16 (+ k_assign_f32 exact re-check pass)", "bound": "hbm",
                         "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "launch_ms": a_ms, "algorithmic_bytes": bytes_per_launch,
                         "algorithmic_flops": flops_per_launch, "effective_TFLOPs": tfs,
                         "rows_rechecked_exact": frecheck, "rows": frows},
            # the kernel that dominates the wall clock is NOT throughput-bound: n/b dependent SGD steps (each needs
            # the centres the previous one wrote); its bytes against the HBM roof are reported for completeness
            "train_kernel": {"kernel": "k_train_persistent" if world == 1 else "k_step_dist_dma + k_step_update (global batch %d)" % (b * world),
                             "bound": "latency (dependent chain of %d steps)" % (n // b),
                             "us_per_step": (ms_per_step - a_ms) * 1e3 / (n // b),
                             "algorithmic_bytes_per_epoch": n * d * 4 * world,
                             "achieved_GBs": n * d * 4 * world / ((ms_per_step - a_ms) * 1e-3) / 1e9,
                             "frac_of_hbm_peak": n * d * 4 * world / ((ms_per_step - a_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "stages": {"assign_ms": a_ms, "assign_clips_per_s": n / (a_ms * 1e-3),
                       "train_epoch_ms": ms_per_step - a_ms,
                       "train_clips_per_s": n / ((ms_per_step - a_ms) * 1e-3),
                       "train_us_per_sgd_step": (ms_per_step - a_ms) * 1e3 / (n // b)},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n, d, k, b, 1234)
            out["mi_stage"] = mi_stage(1234)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
