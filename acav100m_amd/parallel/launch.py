"""One process per GPU, the way the reference's CLIs start them.

The reference spawns its own workers: `torch.multiprocessing.spawn(nprocs=num_gpus)` from
clustering/code/script.py:52-65 and subset_selection/code/chunk.py:28,53, so `bash run.sh` on an 8-GPU node uses
all eight GPUs without any launcher.  Here the CLI re-executes itself once per GPU with the torchrun environment
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT); started under torchrun it finds that environment
already there and just joins.  Children bind to their GPU before anything touches the device.
"""
import os
import socket
import subprocess
import sys


def env_world():
    """(rank, local_rank, world) from the launcher environment, or None outside one"""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        rank = int(os.environ["RANK"])
        return rank, int(os.environ.get("LOCAL_RANK", rank)), int(os.environ["WORLD_SIZE"])
    return None


def bind_device():
    """torch.cuda.set_device(LOCAL_RANK) for a launched process; no-op otherwise or without a GPU"""
    ew = env_world()
    if ew is None:
        return None
    import torch
    n = torch.cuda.device_count()
    if n > 0:
        torch.cuda.set_device(ew[1] % n)
    return ew


def init_process_group(backend="nccl"):
    """Join the process group described by the environment (no-op for a single process).  `backend` falls back
    to gloo when the ranks share a GPU or there is none (tests)."""
    ew = bind_device()
    if ew is None or ew[2] <= 1:
        return ew
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return ew
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    backend = os.environ.get("ACAV_DIST_BACKEND", backend)
    if backend == "nccl" and torch.cuda.device_count() >= ew[2]:
        dist.init_process_group("nccl", rank=ew[0], world_size=ew[2],
                                device_id=torch.device("cuda", ew[1] % torch.cuda.device_count()))
    else:
        dist.init_process_group("gloo", rank=ew[0], world_size=ew[2])
    return ew


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_per_gpu(module, argv, nprocs, extra_env=None):
    """Run `python -m module argv...` once per rank 0..nprocs-1 and wait for all of them.  Returns the list of
    exit codes; raises SystemExit with the first non-zero one (the reference's spawn re-raises a worker failure)."""
    port = _free_port()
    procs = []
    for rank in range(nprocs):
        env = dict(os.environ)
        env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(nprocs),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "ACAV_PARENT_PID": str(os.getpid())})
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, "-m", module] + list(argv), env=env))
    # a rank that dies would leave the others waiting in a collective for ever: watch them all, and take the rest down
    # with the first failure (torch.multiprocessing.spawn does the same for the reference)
    import time
    codes = [None] * nprocs
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        failed = [c for c in codes if c not in (None, 0)]
        if failed:
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            raise SystemExit(failed[0])
        time.sleep(0.05)
    return codes
