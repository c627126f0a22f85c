"""`python cli.py run|reduce_csvs --shards_path=<brace glob .pkl> --meta_path=<dir> --out_path=<file|dir>`
-- the entry points of subset_selection/code/cli.py:17-100 + args.py:11-34 on the MI355X hot path."""
import datetime
import os
import sys
import time
from pathlib import Path

from ..config import SUBSET_DEFAULTS, merge, parse_cli
from .run import compare_measures, merge_all_csvs, reduce_all_pkls, run_chunks, run_single


def get_args(**kwargs):
    args = merge(SUBSET_DEFAULTS, {k: v for k, v in kwargs.items() if k not in ('out_path', 'shards_path', 'meta_path')})
    import torch
    args.computation.device = 'cuda' if (args.computation.use_gpu and torch.cuda.device_count() > 0) else 'cpu'
    if args.computation.device == 'cpu':
        print("no gpu available: the acav100m_amd measures have no CPU path and will refuse to run")
    if args.computation.num_gpus is None:
        args.computation.num_gpus = sys.maxsize
    args.computation.num_gpus = min(args.computation.num_gpus, torch.cuda.device_count())
    return args


def prepare(**kwargs):
    """cli.py:18-43"""
    args = get_args(**kwargs)
    if 'out_path' in kwargs:
        args.data.output.path = Path(kwargs['out_path']).resolve()
    opath = Path(args.data.output.path)
    if opath.stem == opath.name:  # potential dir
        opath = opath / 'output.csv'
    opath.parent.mkdir(parents=True, exist_ok=True)
    args.data.output.path = opath
    if 'shards_path' in kwargs:
        args.data.path = Path(kwargs['shards_path']).resolve()
    if 'meta_path' in kwargs:
        args.data.meta.path = Path(kwargs['meta_path']).resolve()
    mpath = args.data.meta.path
    if mpath is None:
        mpath = Path(args.data.path).parent
    if not mpath.is_dir() and mpath.parent.is_dir():
        mpath = mpath.parent
    args.data.meta.path = mpath
    return args


def run(args):
    """cli.py:90-100"""
    if args.measure_name == 'contrastive':
        from .run_contrastive import run_chunks_contrastive, run_single_contrastive
        return run_single_contrastive(args) if args.chunk_size is None else run_chunks_contrastive(args)
    return run_single(args) if args.chunk_size is None else run_chunks(args)


class Cli:
    def run(self, **kwargs):
        start = time.time()
        out = run(prepare(**kwargs))
        print('done. total time elasped: {}'.format(datetime.timedelta(seconds=time.time() - start)))
        return out

    def reduce_csvs(self, **kwargs):
        start = time.time()
        out = merge_all_csvs(prepare(**kwargs))
        print('done. total time elasped: {}'.format(datetime.timedelta(seconds=time.time() - start)))
        return out

    def reduce_pkls(self, **kwargs):
        start = time.time()
        out = reduce_all_pkls(prepare(**kwargs))
        print('done. total time elasped: {}'.format(datetime.timedelta(seconds=time.time() - start)))
        return out

    def reduce(self, **kwargs):
        """cli.py:69-78: csv caches or pickle caches, whichever the run wrote"""
        args = prepare(**kwargs)
        return merge_all_csvs(args) if (args.save_cache_as_csvs or args.save_cache_as_csvs is None) else reduce_all_pkls(args)

    def compare_measures(self, **kwargs):
        out = compare_measures(prepare(**kwargs))
        print('done')
        return out

    def merge_contrastive(self, **kwargs):
        from .run_contrastive import merge_contrastive
        return merge_contrastive(prepare(**kwargs))


def main(argv=None):
    """Chunked runs use one process per GPU like the reference (chunk.py:28,53: spawn(nprocs=num_gpus)): started
    plainly on a multi-GPU node with a chunk_size, `run` re-executes itself once per GPU; every process takes its
    block of chunks and never talks to the others (no process group).  Under torchrun the environment is already
    there."""
    argv = sys.argv[1:] if argv is None else list(argv)
    command, kwargs = parse_cli(argv)
    from .. import configure_runtime
    configure_runtime()  # hardware queues for side-by-side clusterings: before the first device call of the process
    from ..parallel import launch
    if launch.env_world() is None:
        if command == 'run' and kwargs.get('chunk_size') is not None:
            import torch
            want = kwargs.get('computation.num_gpus')
            have = torch.cuda.device_count()
            if os.environ.get('ACAV_OVERSUBSCRIBE') == '1':  # tests: several processes share a GPU
                have = max(have, int(want or have))
            want = have if want is None else min(int(want), have)
            if want > 1:
                launch.spawn_per_gpu('acav100m_amd.subset_selection.cli', argv, want, {'ACAV_NO_GROUP': '1'})
                return None
        elif command == 'run' and _contrastive_ddp(kwargs):
            # run_contrastive.py:118-152: the unchunked contrastive run trains and scores with one worker per GPU
            import torch
            want = kwargs.get('computation.num_gpus')
            have = torch.cuda.device_count()
            if os.environ.get('ACAV_OVERSUBSCRIBE') == '1':
                have = max(have, int(want or have))
            want = have if want is None else min(int(want), have)
            if want > 1:
                launch.spawn_per_gpu('acav100m_amd.subset_selection.cli', argv, want)
                return None
    elif command == 'run' and kwargs.get('chunk_size') is None and _contrastive_ddp(kwargs):
        launch.init_process_group(kwargs.get('computation.dist_backend', 'nccl'))
    else:
        os.environ.setdefault('ACAV_NO_GROUP', '1')
        launch.bind_device()
    _seed_from_env()
    return getattr(Cli(), command)(**kwargs)


def _contrastive_ddp(kwargs):
    use = kwargs.get('computation.use_distributed')
    use = SUBSET_DEFAULTS['computation']['use_distributed'] if use is None else use
    return kwargs.get('measure_name') == 'contrastive' and kwargs.get('chunk_size') is None and str(use).lower() not in ('false', '0')


def _seed_from_env():
    """ACAV_SEED=<int>: seed the torch-stream generator and Python's `random` in this process (the reference's CLIs
    never seed -- `computation.random_seed` is unused there; every process of a spawned run gets the same seed)."""
    import os
    seed = os.environ.get('ACAV_SEED')
    if seed is not None:
        import random
        from ..rng import manual_seed
        manual_seed(int(seed))
        random.seed(int(seed))


if __name__ == '__main__':
    main()
