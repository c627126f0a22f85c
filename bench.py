#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: clips/sec curated = N / (t_kmeans_train + t_kmeans_assign + t_MI).

Workload (config.workload): BASELINE.json configs[2] -- 1M clips, two 1024-d views (audio + visual), K=256,
select 20 % by greedy batch-MI, one GPU, end to end.  One bench "step" is one pass of the whole hot path over the
resident synthetic features, exactly as the reference's two CLIs run it back to back:

  k-means   per view: KMeans(args, 1024, 256), `clustering.epochs` = 2 epochs of KMeans.add at the reference's batch
            size b=32 (2 x 31 250 sequentially dependent SGD steps, run_clustering.py:132-177) and one
            KMeans.calc_best sweep over all rows (run_clustering.py:180-272)
  hand-off  labels -> assignments [V, 2] int64 (host, the assignment-shard contract of dataloader.py:17-69)
  MI        run_greedy._run_greedy (run_greedy.py:9-54): C = max+1, python shuffle of the candidates, start index,
            EfficientBatchMI(B=20, k=4, keep_unselected) greedy selection of round(0.2 V) = 200 000 clips in ONE
            chunk -- the reference's default (`chunk_size: None`, subset_selection/code/config.py) -- 50 000
            iterations, each a full torch.randperm of the ~10^6 remaining candidates (batch.py:29-32).

value = N * n_gpus * steps / time with the features resident in HBM when the timed region starts.

    python bench.py                      # 1 GPU, defaults
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
           --master-port P bench.py --gpus N --steps K --warmup W

Other workloads (`--workload`, never the default: the driver's line stays on configs[2]) run the PER-GPU SLICE of the two
8-GPU configurations of BASELINE.json on however many GPUs are given (1 here; the driver may launch the same flag with
--gpus 8: every rank then holds its own slice):
  cfg4   configs[3]: 10M clips / 8 = 1.25M clips per GPU, visual 2048-d + audio 128-d, K = 1024
  cfg5   configs[4]: 100M clips / 8 = 12.5M clips per GPU, two 1024-d views (102 GB resident), K = 1024, selection in
         chunks of 100 shards = 100k clips, 10 chunks in lockstep (computation.concurrent_chunks)
Their `roofline` is quoted on the K = 1024 filter against the 16-bit (half / bf16: 2.5 PFLOP/s dense) MFMA roof that binds there (intensity K/2 = 512 flop/B).
  real10 the reference's REAL pipeline shape: ten clusterings over the 5 + 5 layer outputs of VGGish (64 / 128 / 256 / 512 / 128,
         models/vggish.py:20) and SlowFast (88 / 352 / 704 / 1408 / 2304, models/slowfast.py:31), K = 32 (config.py:41; --k 256),
         `combination` pairing P = 45, one chunk of 1M clips; `roofline` on the widest view's filter (2304-d, HBM-bound)

N > 1 (weak scaling: every rank holds its own 1M-clip partition, N million clips in all).  Assign is local and the MI
selection runs per rank on its own partition -- the reference's chunked mode with one chunk per GPU (chunk.py:21-53) -- no
exchange.  k-means TRAINING is one global clustering per view, and its SGD chain is sequential by construction (a step
needs the centres of the step before): what N GPUs do there is a choice, `--multi-gpu` (acav100m_amd/parallel/row_plan.py):
  views      (default) the ONE-GPU arithmetic over all N million rows: batch 32, 2 epochs, shards in global order -- the
             result (and the files of the CLI) of a one-GPU run over the union; the two views' chains run on two
             different ranks, the rows travel to them in bulk (no collective on the step path).  Comparable with the
             N = 1 line row for row; the chain grows with N, so this stage does NOT weak-scale (about 0.4 s x N).
  reference  the reference's own N-GPU run: per-rank batch int(32 / N) of a rotated stream over ALL shards
             (data/clustering.py:25, mps/distributed.py:433-437), ceil(2 / N) epochs (run_clustering.py:146): global batch
             32, N * (N million) / 32 steps per epoch -- N = 8: 2 M steps against the 62.5 k of N = 1.
  rows       large batch, NOT the reference's run: 32 rows of every rank per step (global batch 32 N), ceil(2 / N) epochs:
             N x N fewer steps than `reference`; a different operating point (what earlier rounds timed).
`config` names the mode, the global batch and the SGD steps per epoch.  `--verify` (any N): after the timed region every
rank hashes every clustering's state (all ranks must agree) and rank 0 re-labels 16 384 of its rows with the oracle.

The JSON line also carries
  roofline      k_assign_f16_rw, the HBM-bound kernel of the path: algorithmic bytes N*d*4 + N*8 per launch over the
                kernel's duration measured with HIP events on the library's stream around that launch alone.  `frac` / `achieved`
                are the SUSTAINED figures (160 launches back to back, the median of the last third: the socket's power cap has
                throttled the shader clock by then); `*_timed_region` = the launches inside the timed pass (device at rest)
                (`sweep_*` keys: the whole calc_best sweep incl. centre preparation and the exact re-check pass)
  roofline_mi   the candidate-permutation stream of the greedy loop (16 L bytes per iteration) vs HBM
  variants      the same pipeline with the selection chunked (chunk_size = 100 shards = 100k clips, 10 chunks in
                lockstep, computation.concurrent_chunks) -- a different, equally legal reference configuration
  cpu_baseline  the oracle (oracle/libacav_oracle.so, a C port of the reference algorithm incl. its dense
                [B,P,C,C] scoring and per-iteration randperm) timed on this box's host cores on a bounded sample.
"""
import argparse
import contextlib
import ctypes as C
import io
import itertools
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_COPY_GBS = 6290.0  # same guide: 6.29 TB/s measured with a float4 copy (79 % of the spec) -- context, not the roof
HBM_MEASURED_READ_GBS = 7100.0  # profiles/r03_stream_bench.txt: read-only coalesced nt stream on this chip (0.888 of the spec)
FILTER_DMA_SKELETON_MS = 0.75   # same file: the filter's two LDS-DMA streams (rows nt + 16-bit centres from L2) with no compute, 1M x 1024
EPOCHS = 2                 # clustering/code/config.py: clustering.epochs
RATIO, BATCH_B, SELECT_K = 0.2, 20, 4   # subset_selection/code/config.py: subset.ratio, batch.*


MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak

# per-GPU slices of BASELINE.json's configs (rows per GPU, view widths, K, selection chunk size in clips / lockstep width)
WORKLOADS = {
    "cfg3": dict(name="BASELINE configs[2]", n=1_000_000, dims=(1024, 1024), k=256, chunk=None, width=1),
    "cfg4": dict(name="BASELINE configs[3], per-GPU slice (10M clips / 8)", n=1_250_000, dims=(2048, 128), k=1024, chunk=None, width=1),
    "cfg5": dict(name="BASELINE configs[4], per-GPU slice (100M clips / 8)", n=12_500_000, dims=(1024, 1024), k=1024,
                 chunk=100_000, width=10),
    # the reference's REAL pipeline: ten clusterings over the 5 + 5 layer outputs of its two extractors (clustering/code/
    # models/vggish.py:20: 64 / 128 / 256 / 512 / 128; models/slowfast.py:31: 88 / 352 / 704 / 1408 / 2304), K = 32
    # (clustering/code/config.py:41; --k 256 for the larger point), `combination` pairing over D = 10 -> P = 45, one chunk
    "real10": dict(name="the reference's real 5 + 5 layer pipeline (VGGish + SlowFast layer widths)", n=1_000_000,
                   dims=(64, 128, 256, 512, 128, 88, 352, 704, 1408, 2304), k=32, chunk=None, width=1, audio_views=5),
}


def synth_views(torch, n, d, k, seed, device, views=2, rho=0.5, part=0):
    """SURVEY 8(d) generator: K Gaussian components per view, component centres ~ N(0,1)^d, row = centre + 0.3 N(0,1);
    the views share the component id with probability rho, else draw their own (gives the MI selection a signal).
    The component centres depend on `seed` alone, the rows on (seed, part): the partitions of a multi-GPU run are draws
    from ONE mixture (their union is what a one-GPU run over all rows would cluster)."""
    gen_c = torch.Generator(device=device).manual_seed(seed)
    gen = torch.Generator(device=device).manual_seed(seed + 1 + 1000003 * part)
    dims = [d] * views if isinstance(d, int) else list(d)
    cens = [torch.randn(k, dv, device=device, generator=gen_c) for dv in dims]
    shared = torch.randint(0, k, (n,), device=device, generator=gen)
    out = []
    for d, cen in zip(dims, cens):
        own = torch.randint(0, k, (n,), device=device, generator=gen)
        comp = torch.where(torch.rand(n, device=device, generator=gen) < rho, shared, own)
        x = torch.empty(n, d, device=device, dtype=torch.float32)
        step = 65536
        for s in range(0, n, step):
            e = min(n, s + step)
            x[s:e] = cen[comp[s:e]] + 0.3 * torch.randn(e - s, d, device=device, generator=gen)
        out.append(x)
    return out


class _NS(dict):
    __getattr__ = dict.get


def select_args(seed=0):
    return _NS(batch=_NS(batch_size=BATCH_B, selection_size=SELECT_K, keep_unselected=True),
               computation=_NS(device="cuda", random_seed=seed), log_every=1000, log_times=10,
               node_rank=None, parent_pid=None)


def cpu_baseline(n, dims, k, b, seed, chunk=None):
    """Oracle on the host cores, bounded sample of the same pipeline: SGD steps (after the warm-up) + an assign slice
    per view width, and greedy iterations at V = n (or the chunk size) with the reference's dense scoring.  add() is
    row-parallel over only b=32 rows: min(cores, 16) threads; assign uses every core; the greedy loop is serial.
    Best of 3 for k-means."""
    from oracle import oracle as O
    rs = np.random.RandomState(seed)
    views = len(dims)
    n_steps, n_assign_rows = 256, 65536
    # dense [B,P,C,C] scoring: work per iteration ~ P C^2 (16x at C = 1024, 45x at the real pipeline's P = 45); ~15 s of CPU work in all
    pairs_n = views * (views - 1) // 2
    mi_iters = max(300 if pairs_n == 1 else 20, min(1500, int(1500 * 65536 / (max(k, 64) ** 2 * pairs_n))))
    cores = os.cpu_count() or 1
    train_threads = min(cores, 16)
    per_d = {}
    for d in sorted(set(dims)):
        cen = rs.randn(k, d).astype(np.float32)
        xs = (cen[rs.randint(0, k, n_steps * b + n_assign_rows)] +
              0.3 * rs.randn(n_steps * b + n_assign_rows, d)).astype(np.float32)
        t_train, t_assign = float("inf"), float("inf")
        for rep in range(3):
            km = O.KMeans(d, k, O.Rng(seed), centers=(cen + 0.1 * rs.randn(k, d)).astype(np.float32))
            km.set_state(None, np.full(k, 50.0, np.float32), 10 * k + 12800)
            O.set_threads(train_threads)
            t0 = time.perf_counter()
            for t in range(n_steps):
                km.add(xs[t * b:(t + 1) * b])
            t_train = min(t_train, (time.perf_counter() - t0) / (n_steps * b))   # s per clip and epoch
            O.set_threads(cores)
            t0 = time.perf_counter()
            km.calc_best(xs[n_steps * b:])
            t_assign = min(t_assign, (time.perf_counter() - t0) / n_assign_rows)  # s per clip
        per_d[d] = (t_train, t_assign)
    v = min(n, chunk) if chunk else n   # clips of one selection
    comp = rs.randint(0, k, v)
    a = np.stack([np.where(rs.rand(v) < 0.5, comp, rs.randint(0, k, v)) for _ in range(views)], 1).astype(np.int64)
    a[0] = k - 1
    pairs = list(itertools.combinations(range(views), 2))
    cand = [int(i) for i in rs.permutation(v)]
    O.set_threads(1)
    t0 = time.perf_counter()
    O.BatchMI(a, k, pairs).run_greedy(cand[1:], cand[:1], round(RATIO * v), BATCH_B, SELECT_K, O.Rng(seed), dense=True,
                                      max_iters=mi_iters)
    t_iter = (time.perf_counter() - t0) / mi_iters
    iters = -(-round(RATIO * v) // SELECT_K) * (n // v)   # iterations of all selections over the n clips
    t_clip = sum(EPOCHS * per_d[d][0] + per_d[d][1] for d in dims) + iters * t_iter / n
    return {
        "value": 1.0 / t_clip, "unit": "clips/s", "cores": cores, "kind": "port",
        "sample": f"k-means, best of 3 per view width {sorted(set(dims))}: {n_steps} add() steps of b={b} on {train_threads} "
                  f"threads + calc_best over {n_assign_rows} rows on {cores} threads (K={k}), scaled to {views} views x "
                  f"({EPOCHS} epochs + 1 sweep); MI: {mi_iters} greedy iterations at V={v} (dense [B,P,C,C] scoring + full "
                  f"randperm per iteration, 1 thread) scaled to {iters} iterations; oracle C port, per-clip times summed "
                  f"and inverted",
        "train_clips_per_s_per_epoch": {str(d): 1.0 / per_d[d][0] for d in per_d},
        "assign_clips_per_s": {str(d): 1.0 / per_d[d][1] for d in per_d},
        "mi_ms_per_iteration": t_iter * 1e3,
        # how this port relates to the TRUE reference (its Python, imported in the build container:
        # 8 cores, torch 2.10 CPU) on the rows of BASELINE.md section 2 -- the port is 2.4-12.6x FASTER than the code it
        # restates (no per-op dispatch, no unique()), so the GPU / reference ratio is larger than value / this value
        "calibration_vs_true_reference": {
            "where": "build container, 8 cores; reference = /root/reference imported (BASELINE.md section 2), port = oracle/acav_oracle.c",
            "kmeans_train_10k_x512_K64_624_steps_s": {"reference": 1.21, "port_1_thread": 0.096},
            "kmeans_assign_10k_x512_K64_s": {"reference": 0.04, "port_1_thread": 0.032, "port_8_threads": 0.0059},
            "mi_greedy_V10k_C64_500_iters_s": {"reference": 1.81, "port_dense_1_thread": 0.338},
            "mi_greedy_V100k_C64_5000_iters_s": {"reference": 14.0, "port_dense_1_thread": 5.81},
            "cfg1_like_end_to_end_s": {"reference": 4.3, "port": 0.59},
        },
    }


def select_chunked(a, types, chunk, width, seed0=1):
    """The reference's chunked selection (chunk.py:21-53): every chunk of `chunk` clips selects 20 % of its own clips,
    `width` chunks in lockstep on this GPU (computation.concurrent_chunks).  -> list of per-chunk (S, GAIN) in chunk order,
    S in chunk-local ids."""
    from acav100m_amd.rng import Generator
    from acav100m_amd.subset_selection.measures.batch import EfficientBatchMI
    from acav100m_amd.subset_selection.run_greedy import _prepare
    from concurrent.futures import ThreadPoolExecutor
    sargs = select_args()
    n = a.shape[0]
    out = []

    def prepare_group(g0):  # host work of a group (candidate shuffle, tables, device handles): under the previous group's greedy loop
        return [_prepare(sargs, a[c0:c0 + chunk], types, None, RATIO, "batch_mi", "combination", True, False,
                         generator=Generator(seed0 + g0 // chunk + i))
                for i, c0 in enumerate(range(g0, min(n, g0 + chunk * width), chunk))]

    groups = list(range(0, n, chunk * width))
    with contextlib.redirect_stdout(io.StringIO()), ThreadPoolExecutor(1) as pool:
        nxt = pool.submit(prepare_group, groups[0])
        for gi, g0 in enumerate(groups):
            prepared = nxt.result()
            if gi + 1 < len(groups):
                nxt = pool.submit(prepare_group, groups[gi + 1])
            res = EfficientBatchMI.run_greedy_multi([p[0] for p in prepared], [p[2] for p in prepared],
                                                    [p[1] for p in prepared])
            out.extend((r[0], r[1]) for r in res)
            # the group's handles go HERE, on the launching thread, before the next group's set-up: their device blocks are parked
            # (no hipFree since the pool holds a group's ~1 000 blocks) and the next set-up finds them -- released from the helper
            # thread they raced with that set-up, and a set-up that misses the pool is a storm of hipMalloc under the next loop
            # (tools/exp/NOTES_r06.md section 10); the product's own runner (subset_selection/run.py) drops them the same way
            prepared.clear()
            del prepared
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg3")  # cfg3 = the metric's configuration (the driver's line)
    ap.add_argument("--n", "--rows", dest="n", type=int, default=None)  # --rows: torchrun's own parser trips over "--n"
    ap.add_argument("--d", type=int, default=None)   # overrides (tests): one width for both views
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--multi-gpu", dest="multi_gpu", choices=("views", "reference", "rows"), default="views")  # N > 1: see the docstring
    ap.add_argument("--verify", action="store_true")  # cross-rank state hashes + an oracle sample on rank 0, outside the timed region
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    args = ap.parse_args()
    # the contract is ONE JSON line on stdout: whatever the libraries print there (RCCL's version banner at communicator
    # creation, for one) goes to stderr instead -- fd 1 is parked and handed back for the result line only
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import acav100m_amd
    acav100m_amd.configure_runtime()  # GPU_MAX_HW_QUEUES, before the HIP runtime initialises
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

    import random
    import acav100m_amd
    from acav100m_amd import _lib
    from acav100m_amd.clustering import KMeans
    from acav100m_amd.subset_selection.run_greedy import _run_greedy
    lib = acav100m_amd.load_library()

    wl = WORKLOADS[args.workload]
    n = args.n if args.n is not None else wl["n"]
    dims = (args.d, args.d) if args.d is not None else tuple(wl["dims"])
    k = args.k if args.k is not None else wl["k"]
    chunk = wl["chunk"] if wl["chunk"] and wl["chunk"] < n else None
    b, nviews = args.batch, len(dims)
    d = max(dims)  # the view the roofline is quoted on
    xs = synth_views(torch, n, dims, k, 1234, dev, part=rank)
    torch.cuda.synchronize()

    cargs = _NS(computation=_NS(device="cuda", num_gpus=world))
    sargs = select_args()
    n_audio = wl.get("audio_views", 1)  # clustering types as dataloader.py:17-69 builds them: sorted (model_key, layer)
    types = [("audio_model", f"layer_{i}") for i in range(n_audio)] + [("visual_model", f"layer_{i}") for i in range(nviews - n_audio)]
    npairs = nviews * (nviews - 1) // 2  # `combination` pairing (pairing.py:5-41)
    subset = round(RATIO * n) if chunk is None else sum(round(RATIO * min(chunk, n - c0)) for c0 in range(0, n, chunk))
    iters = -(-subset // SELECT_K) if chunk is None else sum(-(-round(RATIO * min(chunk, n - c0)) // SELECT_K) for c0 in range(0, n, chunk))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    plan = None
    if world > 1:  # every rank's n rows = shards of 1000 clips, dealt rank::world like the CLI's shards
        from acav100m_amd.parallel import make_plan
        per_rank = [min(1000, n - s0) for s0 in range(0, n, 1000)]
        plan = make_plan(args.multi_gpu, [rows for rows in per_rank for _ in range(world)], world, b, EPOCHS)
    labels = [torch.empty(n, dtype=torch.long, device=dev) for _ in range(nviews)]
    stage = {"train": [], "assign": [], "handoff": [], "mi": []}
    filt_ms, sweep_ms, filt_stats = [], [], []
    last = {}

    def one_pass(timed):
        last.pop("kms", None)  # the previous pass's handles (and their streams) go before this pass creates its own -- see back_to_back_sweeps
        acav100m_amd.manual_seed(0)
        random.seed(0)
        kms = [KMeans(cargs, dv, k).to(dev) for dv in dims]
        for km in kms:
            km.initialize()
        t0 = time.perf_counter()
        for epoch in range(EPOCHS if plan is None else plan.epochs):
            lr = 0.1 ** (2 + epoch // 5)
            if world == 1:  # the two views' SGD chains are independent: side by side on the GPU (run_clustering does the same)
                KMeans.train_epoch_multi(kms, xs, b, lr=lr)
            else:
                # the rows of 1 024 steps at a time travel to the rank that runs a view's chain (view v -> rank v % world),
                # chunk by chunk across the views (acav_kmeans_train_plan_multi: one communicator per view, every rank feeds
                # every exchange before it blocks in its own chain): the views train on different ranks at the same time,
                # then the trainers hand out their states.  WHICH rows form a step's batch: `plan` (--multi-gpu)
                trainers = KMeans.train_epoch_plan_multi(kms, xs, plan, lr=lr)
                for v, km in enumerate(kms):
                    km.broadcast_state_from(trainers[v], comm_slot=v)
        for km in kms:
            km.synchronize()
        t1 = time.perf_counter()
        last["train_stats"] = [list(km.train_stats()) for km in kms]
        last["kms"] = kms
        for vi, (km, x, lab) in enumerate(zip(kms, xs, labels)):  # assign sweep, timed with HIP events on the library's own stream
            _lib.check(lib.acav_kmeans_timer_begin(km._h))
            _lib.check(lib.acav_kmeans_assign(km._h, _lib.ptr(x), n, _lib.ptr(lab), None))
            ms = C.c_float(0)
            _lib.check(lib.acav_kmeans_timer_end(km._h, C.byref(ms)))
            fm = C.c_float(0)
            _lib.check(lib.acav_kmeans_filter_time(km._h, C.byref(fm)))
            if timed:
                sweep_ms.append((vi, ms.value))
                filt_ms.append((vi, fm.value))
                filt_stats.append(km.filter_stats())
        t2 = time.perf_counter()
        a = torch.stack(labels, 1).cpu().numpy()  # the k-means -> MI hand-off (assignment shards in the CLI flow)
        t3 = time.perf_counter()
        if chunk is None:
            with contextlib.redirect_stdout(io.StringIO()):
                S, GAIN, _ = _run_greedy(sargs, a, types, None, RATIO, "batch_mi", "combination", True, False)
            t4 = time.perf_counter()
            assert len(S) == subset and len(set(S)) == subset
        else:  # the reference's chunked mode: every chunk selects 20 % of its own clips, `width` chunks in lockstep
            res = select_chunked(a, types, chunk, wl["width"])
            t4 = time.perf_counter()
            S = [c0 * chunk + i for c0, (Sc, _) in enumerate(res) for i in Sc]
            assert len(S) == subset and len(set(S)) == subset
        last["S"], last["a"] = S, a
        if args.verify:
            last["states"] = [km.state_arrays() for km in kms]
        if timed:
            stage["train"].append(t1 - t0)
            stage["assign"].append(t2 - t1)
            stage["handoff"].append(t3 - t2)
            stage["mi"].append(t4 - t3)
        del kms

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

    verify = None
    if args.verify:
        verify = run_verify(torch, dist, world, rank, last, xs, labels, dims, k)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = n * world * args.steps / elapsed
        st = {key: float(np.mean(v)) for key, v in stage.items()}
        vq = int(np.argmax(dims))  # the view the roofline is quoted on: the widest (cfg4: the 2048-d visual view)
        same = len(set(dims)) == 1  # views of one shape: the figures are the mean over all their sweeps
        f_ms = float(np.mean([t for vi, t in filt_ms if same or vi == vq]))
        s_ms = float(np.mean([t for vi, t in sweep_ms if same or vi == vq]))
        per_view = [{"d": dv, "filter_ms": float(np.mean([t for vi, t in filt_ms if vi == i])),
                     "sweep_ms": float(np.mean([t for vi, t in sweep_ms if vi == i]))} for i, dv in enumerate(dims)]
        bytes_per_launch = n * d * 4 + n * 8
        flops_per_launch = 2.0 * n * k * d
        gbs = bytes_per_launch / (f_ms * 1e-3) / 1e9
        tflops = flops_per_launch / (f_ms * 1e-3) / 1e12
        mfma_bound = k > 256  # intensity K/2 flop/B against the bf16 ridge of ~314 flop/B
        traffic_profile = None
        pdir = os.path.join(ROOT, "profiles")
        for name in sorted(os.listdir(pdir), reverse=True) if os.path.isdir(pdir) else []:
            if name.endswith(".json") and "_pmc_assign" in name:  # summaries of the separate rocprofv3 --pmc passes (tools/summarize_pmc.py)
                pm = json.load(open(os.path.join(pdir, name)))
                if (pm.get("d"), pm.get("K")) == (d, k) and pm.get("kernel", "").startswith("k_assign_"):
                    scale = n / pm["rows"]  # per-row traffic of the same kernel shape, scaled to this launch's rows
                    traffic_profile = {"file": "profiles/" + name, "rows_in_profile": pm["rows"],
                                       "bytes_per_launch": pm["traffic_bytes_per_launch"] * scale}
                    break
        steps_per_epoch, train_epochs = (n // b, EPOCHS) if plan is None else (plan.steps, plan.epochs)
        train_steps = train_epochs * nviews * steps_per_epoch
        if chunk is None:
            perm_bytes = sum(16 * (n - 1 - SELECT_K * t) for t in range(iters))
        else:
            perm_bytes = sum(sum(16 * (min(chunk, n - c0) - 1 - SELECT_K * t)
                                 for t in range(-(-round(RATIO * min(chunk, n - c0)) // SELECT_K))) for c0 in range(0, n, chunk))
        views_txt = " + ".join(f"{dv}-d" for dv in dims)
        sel_txt = (f"ONE-chunk greedy batch-MI selection of {subset} clips (B={BATCH_B}, k={SELECT_K}, {iters} iterations; the "
                   f"reference's default chunk_size=None)" if chunk is None else
                   f"chunked greedy batch-MI selection (chunk_size = {chunk} clips = {chunk // 1000} shards, {-(-n // chunk)} chunks, "
                   f"{wl['width']} in lockstep; every chunk selects 20 % of its clips: {subset} clips, {iters} iterations in all, "
                   f"B={BATCH_B}, k={SELECT_K})")
        roof = {"kernel": "k_assign_f16_rw" + (" + k_assign_merge (K > 256: one workgroup per (row tile, centre group) pair)" if k > 256 else ""),
                "view": f"{d}-d"}
        if mfma_bound:
            roof.update({"bound": "mfma", "achieved": tflops, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tflops / MFMA_BF16_PEAK_TFLOPS, "hbm_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS})
        else:
            roof.update({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "frac_of_measured_copy_rate": gbs / HBM_MEASURED_COPY_GBS,
                         "frac_of_measured_read_rate": gbs / HBM_MEASURED_READ_GBS,
                         "dma_only_skeleton_ms_1M_x_1024": FILTER_DMA_SKELETON_MS})
        roof.update({
            # HBM bytes per launch from the rocprofv3 --pmc passes of this kernel on this shape (FETCH_SIZE x 2 per the MI355X
            # guide + WRITE_SIZE; separate runs, tools/collect_profiles.sh): the committed summary named below -- a counter
            # pass cannot run inside the timed region
            "traffic": traffic_profile["bytes_per_launch"] if traffic_profile else None,
            "traffic_from_committed_profile": traffic_profile,
            "launch_ms": f_ms, "algorithmic_bytes": bytes_per_launch, "algorithmic_flops": flops_per_launch,
            "effective_TFLOPs": tflops, "sweep_ms": s_ms, "sweep_GBs": bytes_per_launch / (s_ms * 1e-3) / 1e9,
            "sweep_frac": bytes_per_launch / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "sweep_over_launch": s_ms / f_ms,
            "rows_rechecked_exact": [int(f[2]) for f in filt_stats[-nviews:]], "rows": n, "per_view": per_view})
        out = {
            "metric": f"clips/sec curated (k-means train + assign, {nviews} views) + MI greedy selection of 20 %",
            "value": value, "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{wl['name']}: {n} clips{' per GPU' if world > 1 else ''} x {nviews} views ({views_txt}), K={k}: per view "
                                   f"{train_epochs} training epochs of {steps_per_epoch} SGD steps at a global batch of "
                                   f"{b if plan is None else plan.global_batch} rows"
                                   + (f" (multi-GPU training mode '{plan.mode}': see bench.py's docstring)" if plan is not None else "")
                                   + f" + 1 assign sweep, then {sel_txt}, end to end on 1 GPU per partition",
                       "global_batch": b if plan is None else plan.global_batch, "sgd_steps_per_epoch": steps_per_epoch,
                       "train_epochs": train_epochs, "multi_gpu_mode": None if plan is None else plan.mode,
                       "rows_per_gpu": n, "views": nviews, "view_dims": list(dims), "epochs": EPOCHS,
                       "select": subset, "mi_chunks": 1 if chunk is None else -(-n // chunk), "mi_pairs": npairs,
                       **({} if world == 1 else {"weak_scaling_projection": scaling_projection(plan.mode, world, st)})},
            "stages": {"train_s": st["train"], "assign_s": st["assign"], "handoff_s": st["handoff"], "mi_s": st["mi"],
                       "train_us_per_sgd_step": st["train"] * 1e6 / train_steps,
                       "assign_sweep_ms": s_ms, "mi_us_per_iteration": st["mi"] * 1e6 / iters,
                       "train_clips_per_s": n / st["train"], "assign_clips_per_s": n / st["assign"],
                       "mi_clips_per_s": n / st["mi"]},
            "roofline": roof,
            "roofline_mi": {"kernel": "k_mt_generate_lanes + k_fy_part + k_fy_tile + k_fy_resolve + k_fy_gather_select (one greedy iteration)",
                            "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                            "algorithmic_bytes": perm_bytes, "achieved": perm_bytes / st["mi"] / 1e9,
                            "frac": perm_bytes / st["mi"] / 1e9 / HBM_PEAK_GBS,
                            "note": "16 L bytes per iteration (int64 candidate permutation read + write, SURVEY 8(d)) over "
                                    "the whole selection incl. its host part (python shuffle, table set-up)"},
            "train_kernel": {"kernel": "persistent epoch kernels (the views' launches side by side: acav_kmeans_train_multi)" if world == 1 else
                             "persistent epoch kernels per 1 024-step chunk at the trainer rank of each view (global batch %d; per-step "
                             "launches k_step_dist_dma + k_step_update beyond batch 32)" % plan.global_batch,
                             "bound": "latency (dependent chain of %d steps per epoch and view)" % steps_per_epoch,
                             "us_per_step": st["train"] * 1e6 / train_steps,
                             "persistent_launches_and_fallbacks": last.get("train_stats")},
        }
        if not args.no_variants and world == 1 and chunk is None and n >= 200_000:
            out["variants"] = {"chunked_lockstep": chunked_variant(last["a"], types, n, st)}
            if not mfma_bound:  # (an HBM figure: the K = 256 shapes)
                bb = back_to_back_sweeps(lib, last["kms"][vq], xs[vq], labels[vq], n, bytes_per_launch)
                out["roofline"]["back_to_back"] = bb
                if "frac_settled" in bb:
                    # the HEADLINE fraction is the sustained one: what a sweep over more than a few million rows
                    # sees once the power controller has settled; the launch inside the timed pass (after a latency-bound training
                    # stage: the device at rest) is the `*_timed_region` figure
                    rf = out["roofline"]
                    rf["frac_timed_region"], rf["achieved_timed_region"], rf["launch_ms_timed_region"] = rf["frac"], rf["achieved"], rf["launch_ms"]
                    rf["frac"], rf["launch_ms"] = bb["frac_settled"], bb["launch_ms_settled"]
                    rf["achieved"] = bytes_per_launch / (bb["launch_ms_settled"] * 1e-3) / 1e9
                    rf["frac_of_measured_copy_rate"] = rf["achieved"] / HBM_MEASURED_COPY_GBS
                    rf["frac_of_measured_read_rate"] = rf["achieved"] / HBM_MEASURED_READ_GBS
                    rf["what_caps_it"] = ("socket power: back to back the filter pulls ~1.35 kW of the 1.4 kW cap (PVIOL 95-100 %), the shader "
                                          "clock falls 2.4 -> 1.55-1.65 GHz within ~50 ms; without the MFMAs (timing-only ablation) 0.93 kW, no "
                                          "throttling, 0.685-0.69: profiles/r06_filter_sustained.txt")
        if not args.no_variants and world == 1 and k <= 256:
            del xs
            torch.cuda.empty_cache()
            out["variants"] = dict(out.get("variants", {}), assign_hard_data=assign_hard_variant(torch, lib, n, d, k, b, dev))
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n, dims, k, b, 1234, chunk)
        if verify is not None:
            out["verify"] = verify
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


def scaling_projection(mode, world, st):
    """What the N-GPU line should be read against.  Assign and the MI selection are per partition (flat in N);
    the SGD chain of a view is ONE sequential chain whatever N is (a step needs the centres of the step before -- the reference's
    own DDP run takes the same number of global steps): `views` walks all N n rows at the one-GPU batch, so the training stage
    grows ~ N and the whole job cannot weak-scale.  From THIS run's stage times: the one-GPU line would spend about train / N in
    training (views / reference: the same per-step cost) -> efficiency = (train / N + rest) / (train + rest)."""
    rest = st["assign"] + st["handoff"] + st["mi"]
    steps_ratio = {"views": world, "reference": world, "rows": 1.0 / world}[mode]  # SGD steps per epoch-set vs the one-GPU run
    t1 = st["train"] / steps_ratio + rest
    return {"mode": mode, "train_steps_vs_one_gpu": steps_ratio, "estimated_one_gpu_pass_s": t1,
            "expected_weak_scaling_efficiency": t1 / (st["train"] + rest),
            "why": "one sequential SGD chain per view: its length is set by the global row count (views / reference) -- only assign "
                   "and the selection scale with N; `rows` (global batch 32 N) is a different operating point, not the reference's run"}


def run_verify(torch, dist, world, rank, last, xs, labels, dims, k, sample=16384):
    """--verify, outside the timed region: (i) every rank hashes every clustering's final state (centres, usage counts,
    count) -- after the trainers' hand-out all ranks must hold the same bytes; (ii) rank 0 re-labels the first `sample`
    rows of its partition with the ORACLE (oracle/, the checker) from that state and compares with the labels the assign
    sweep wrote; (iii) the selection's size / uniqueness was asserted in the pass itself.  Returns the verdict (rank 0)
    and raises on every rank when something differs, so that a first multi-GPU contact ends in a verdict, not a hang."""
    import hashlib
    digests = []
    for centers, counts, count, fallback in last["states"]:
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(centers, np.float32).tobytes())
        h.update(np.ascontiguousarray(counts, np.float32).tobytes())
        h.update(str((int(count), int(fallback))).encode())
        digests.append(h.hexdigest())
    everyone = [digests]
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, digests)
    same = all(d == everyone[0] for d in everyone)
    verdict = {"state_sha256": everyone[0], "ranks_agree": bool(same), "oracle_sample_rows": 0, "oracle_labels_equal": None}
    ok = same
    if rank == 0:
        from oracle import oracle as O
        m = min(sample, xs[0].shape[0])
        equal = []
        for (centers, counts, count, _fb), x, lab, d in zip(last["states"], xs, labels, dims):
            ref = O.KMeans(d, k, O.Rng(0), centers=np.ascontiguousarray(centers, np.float32))
            ref.set_state(None, np.ascontiguousarray(counts, np.float32), int(count))
            want = ref.calc_best(x[:m].cpu().numpy())[0]
            equal.append(bool(np.array_equal(want, lab[:m].cpu().numpy())))
        verdict.update(oracle_sample_rows=int(m), oracle_labels_equal=equal)
        ok = ok and all(equal)
    if world > 1:
        flag = [ok]
        dist.broadcast_object_list(flag, src=0)
        ok = bool(flag[0]) and same
    if not ok:
        raise SystemExit("bench.py --verify FAILED on rank {}: {}".format(rank, verdict))
    return verdict


def back_to_back_sweeps(lib, km, x, lab, n, bytes_per_launch, reps=160):
    """The roofline kernel under SUSTAINED load: `reps` sweeps of the timed workload's widest view with nothing between them (~0.13 s of
    load).  The timed pass launches the filter after a latency-bound training stage, i.e. on a device at rest; back to back the same
    launch runs into the socket's POWER cap: ~1.35 of 1.4 kW, PVIOL 95-100 %, the shader clock drops from 2.4 to ~1.1 GHz within
    ~5 ms (the controller overshoots: launches of 1.0-1.1 ms) and settles at 1.55-1.65 GHz after ~50 ms (0.80-0.82 ms per launch;
    tools/filter_sustained.py, profiles/r06_filter_sustained.txt -- the 12-sweep window of round 5 sat inside the overshoot and read
    0.555).  Settled = the median of the last third.  Measured once after the timed region."""
    try:
        from acav100m_amd import _lib
        ms = []
        for _ in range(reps):
            _lib.check(lib.acav_kmeans_assign(km._h, _lib.ptr(x), n, _lib.ptr(lab), None))
            km.synchronize()
            fm = C.c_float(0)
            _lib.check(lib.acav_kmeans_filter_time(km._h, C.byref(fm)))
            ms.append(fm.value)
        settled = float(np.median(ms[2 * reps // 3:]))
        return {"sweeps": reps, "launch_ms_first": ms[0], "launch_ms_settled": settled, "launch_ms_worst": float(max(ms)),
                "frac_settled": bytes_per_launch / (settled * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "frac_worst_of_the_transient": bytes_per_launch / (max(ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "launch_ms_every_8th": [round(v, 4) for v in ms[::8]]}
    except Exception as exc:  # informational leg: never takes the driver line down
        return {"error": repr(exc)}


def assign_hard_variant(torch, lib, n, d, k, b, dev):
    """The assign sweep on data the half-precision filter CANNOT decide everywhere (the timed workload's clusters are well separated:
    0 rows undecided): centre spread 0.06 / 0.05 against the 0.3 noise, centres out of real training (tools/recheck_table.py's
    rows).  Undecided rows go through the emission pass + the exact evaluation of their candidate centres (k_assign_cand);
    labels are bit-identical to the exact sweep either way.  Measured once, outside the timed region."""
    try:
        import acav100m_amd
        from acav100m_amd import _lib
        from acav100m_amd.clustering import KMeans
        rows = []
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
            lab = torch.empty(n, dtype=torch.long, device=dev)
            best = (1e9, 1e9)
            for rep in range(4):
                _lib.check(lib.acav_kmeans_timer_begin(km._h))
                _lib.check(lib.acav_kmeans_assign(km._h, _lib.ptr(x), n, _lib.ptr(lab), None))
                ms, fm = C.c_float(0), C.c_float(0)
                _lib.check(lib.acav_kmeans_timer_end(km._h, C.byref(ms)))
                _lib.check(lib.acav_kmeans_filter_time(km._h, C.byref(fm)))
                if rep:
                    best = min(best, (ms.value, fm.value))
            _, _, undecided = km.filter_stats()
            cand_rows, cand_pairs, full_rows = km.recheck_stats()
            rows.append({"centre_spread": spread, "rows": n, "rows_undecided_by_the_filter": int(undecided),
                         "settled_by_candidates_rows": int(cand_rows), "candidate_pairs": int(cand_pairs),
                         "full_exact_sweep_rows": int(full_rows), "filter_ms": best[1], "sweep_ms": best[0],
                         "sweep_frac_of_hbm": (n * d * 4 + n * 8) / (best[0] * 1e-3) / 1e9 / HBM_PEAK_GBS})
            del x, km, lab
        return rows
    except Exception as exc:  # informational leg: never takes the driver line down
        return {"error": repr(exc)}


def chunked_variant(a, types, n, st, chunk=100_000, width=10):
    """The same selection chunked: chunk_size = 100 shards of 1000 clips (SURVEY 8(d)'s suggestion for cfg5), every
    chunk selects 20 % of its clips (chunk.py:21-53), `width` chunks in lockstep on one GPU
    (computation.concurrent_chunks).  Measured once, outside the driver's timed region; combined with the timed
    k-means stages of this run."""
    try:
        import random
        random.seed(0)
        t0 = time.perf_counter()
        total = sum(len(r[0]) for r in select_chunked(a, types, chunk, width))
        t_mi = time.perf_counter() - t0
        t_all = st["train"] + st["assign"] + st["handoff"] + t_mi
        return {"workload": f"{n // chunk} chunks of {chunk} clips, {width} in lockstep, each selects 20 %",
                "mi_s": t_mi, "selected": total, "clips_per_s": n / t_all,
                "note": "k-means stages as timed above; this selection measured once after the timed region"}
    except Exception as exc:  # informational leg: never takes the driver line down
        return {"error": repr(exc)}


if __name__ == "__main__":
    main()
