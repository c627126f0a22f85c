"""`python cli.py cluster --feature_path=<brace glob .pkl> --out_path=<dir> --meta_path=<dir> [--a.b=v]`
-- the entry point of clustering/code/cli.py:12-27 + args.py:11-24 on the MI355X hot path."""
import sys

from .. import shards as io
from ..config import CLUSTERING_DEFAULTS, merge, parse_cli
from .run_clustering import run_clustering, store_shards_set


def get_args(**kwargs):
    if 'out_path' in kwargs:
        kwargs['data.output.path'] = kwargs.pop('out_path')
    if 'feature_path' in kwargs:
        kwargs['shards_path'] = kwargs.pop('feature_path')
    if 'shards_path' in kwargs:
        kwargs['data.path'] = kwargs.pop('shards_path')
    if 'meta_path' in kwargs:
        kwargs['data.meta.path'] = kwargs.pop('meta_path')
    args = merge(CLUSTERING_DEFAULTS, kwargs)
    if args.computation.num_gpus is None:
        import torch
        args.computation.num_gpus = torch.cuda.device_count()
    from ..parallel import world
    args.computation.num_gpus = max(1, min(args.computation.num_gpus, world()[1]))
    args.run_info = io.run_info()
    args.run_id = io.run_id(args.run_info)
    return args


class Cli:
    def run(self, **kwargs):
        return self.cluster(**kwargs)

    def cluster(self, **kwargs):
        args = get_args(**kwargs)
        args.data.output.path.mkdir(parents=True, exist_ok=True)
        saved = run_clustering(args)
        store_shards_set(args, saved)
        print('done')
        return saved


def main(argv=None):
    """One process per GPU like the reference (script.py:52-65): started plainly with computation.num_gpus > 1
    (default: every visible GPU) the CLI re-executes itself once per GPU; started by torchrun -- or as one of those
    children -- it binds to its GPU (LOCAL_RANK) and joins the process group before anything else."""
    argv = sys.argv[1:] if argv is None else list(argv)
    command, kwargs = parse_cli(argv)
    from .. import configure_runtime
    configure_runtime()  # hardware queues for side-by-side clusterings: before the first device call of the process
    from ..parallel import launch
    if launch.env_world() is None:
        want = kwargs.get('computation.num_gpus')
        if want is None:
            import torch
            want = torch.cuda.device_count()
        if int(want) > 1:
            launch.spawn_per_gpu('acav100m_amd.clustering.cli', argv, int(want))
            return None
    else:
        launch.init_process_group(kwargs.get('computation.dist_backend', 'nccl'))
    _seed_from_env()
    return getattr(Cli(), command)(**kwargs)


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
