"""CLI argument handling shared by the two stages.

Behavioural contract of the reference (clustering/code/args.py:11-83 + config.py, and
subset_selection/code/args.py:11-89 + config.py): keyword arguments with dotted names override a
nested default dict (`--a.b.c=v`), unknown keys are created, every `path` entry becomes an absolute
pathlib.Path, and missing attributes read as None (munch.DefaultMunch(None)).
"""
import ast
import copy
from pathlib import Path


class Namespace(dict):
    """dict with attribute access; missing keys read as None (DefaultMunch(None) semantics)."""

    def __getattr__(self, key):
        if key.startswith("__"):
            raise AttributeError(key)
        return self.get(key)

    def __setattr__(self, key, value):
        self[key] = value

    def __deepcopy__(self, memo):
        return Namespace({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(tree):
    return Namespace({k: _wrap(v) if isinstance(v, dict) else v for k, v in tree.items()})


def _resolve_paths(tree):
    for key, val in tree.items():
        if isinstance(val, dict):
            _resolve_paths(val)
        elif val is not None and (key == 'path' or key.endswith('_file') or key.endswith('_dir')):
            tree[key] = Path(val).resolve()
    return tree


def merge(defaults, overrides):
    """defaults (nested dict) <- {'a.b.c': v}; returns a Namespace tree with resolved paths."""
    tree = copy.deepcopy(defaults)
    for dotted, value in overrides.items():
        node = tree
        parts = dotted.split('.')
        for part in parts[:-1]:
            if not isinstance(node.get(part), dict):
                node[part] = {}
            node = node[part]
        node[parts[-1]] = value
    return _wrap(_resolve_paths(tree))


def parse_cli(argv):
    """`<command> --key=value --flag ...` -> (command, kwargs); values go through literal_eval like
    python-fire does (numbers, booleans, None, lists), everything else stays a string."""
    if not argv:
        raise SystemExit("usage: cli.py <command> [--key=value ...]")
    command, kwargs, i = argv[0], {}, 1
    while i < len(argv):
        tok = argv[i]
        if not tok.startswith('--'):
            raise SystemExit(f"unexpected argument {tok!r}")
        if '=' in tok:
            key, raw = tok[2:].split('=', 1)
        elif i + 1 < len(argv) and not argv[i + 1].startswith('--'):
            key, raw = tok[2:], argv[i + 1]
            i += 1
        else:
            key, raw = tok[2:], 'True'
        try:
            val = ast.literal_eval(raw)
        except (ValueError, SyntaxError):
            val = raw
        kwargs[key] = val
        i += 1
    return command, kwargs


# clustering/code/config.py:1-58 (only the keys the hot path reads; the rest is accepted and ignored)
CLUSTERING_DEFAULTS = {
    'models': ['layer_vggish', 'layer_slow_fast'],
    'model_types': {'audio': ['vggish', 'layer_vggish'], 'visual': ['slow_fast', 'layer_slowfast']},
    'data': {
        'path': 'data',
        'batch_size': 32,
        'resident_bytes': None,  # ours: device budget for feature rows; beyond it the shards stream in groups
        # ours: WHICH rows form the training batches (acav100m_amd/parallel/row_plan.py: loader_stream).
        # 'reference' (default): the batches the reference's DataLoader delivers for this computation.num_workers -- with
        #   num_workers > 0 (its default: 40) whole batches round-robin over worker streams, worker w reading shards [w::num_workers]
        #   (data/clustering.py:17-66,212-228); num_workers = 0: one stream over the shards in order.
        # 'single': the single-stream order whatever num_workers says (rounds 1-5 of this build).  ACAV_LOADER_ORDER overrides.
        'loader_order': 'reference',
        # 'wrap' (default): every stream is get_length() samples (mps/distributed.py:444-460), a short one starts over -- and the
        #   in-process loader (num_workers = 0) continues where it stopped in the next epoch -- webdataset.ResizedDataset as the
        #   reference wraps its dataset (data/clustering.py:61-65).  'drop': whole batches of the rows that exist.  ACAV_LOADER_TAIL overrides.
        'loader_tail': 'wrap',
        'meta': {'path': None},
        'output': {'path': 'output', 'shard_ok_ratio': 0.99},
    },
    'computation': {
        'random_seed': 0,
        'device': 'cuda',
        'num_workers': 40,
        'num_gpus': None,
        'dist_backend': 'nccl',
        'dist_init_method': 'tcp://localhost:9999',
    },
    'clustering': {
        'ncentroids': 32,
        'epochs': 2,
        'cached_epoch': None,
        'resume_training': False,
        'load_cache_from_shard_subset': True,
        # ours: what several GPUs do in training (acav100m_amd/parallel/row_plan.py).
        # 'views' (default): the clusterings are dealt out over the GPUs, every GPU trains its share over ALL rows with the
        #   one-GPU batch stream and epoch count: the N-GPU run writes the files of the one-GPU run.
        # 'striped': the same one-GPU batch stream and epoch count (hence the same files) with the ROWS partitioned: every GPU holds
        #   the rows of its own shards (rank::N) and they travel in bulk to the GPU that runs a clustering's chain.
        # 'reference': the reference's own N-GPU run -- every GPU holds the rows of its own shards (rank::N), rank q feeds
        #   int(batch_size / N) rows per step (data/clustering.py:25) of its rotated stream over ALL shards
        #   (mps/distributed.py:433-437), epochs = ceil(epochs / N) (run_clustering.py:146): the global batch stays
        #   batch_size, N * rows / batch_size steps per epoch.
        # 'rows': a LARGE-BATCH operating point, not the reference's run: batch_size rows of every GPU's own shards per step
        #   (global batch N x batch_size), ceil(epochs / N) epochs -- N x N fewer SGD steps than 'reference' (64 x at N = 8).
        'multi_gpu': 'views',
    },
    'debug': False,
}

# subset_selection/code/config.py:1-53
SUBSET_DEFAULTS = {
    'data': {'path': 'data', 'output': {'path': 'output.csv'}, 'meta': {'path': None}},
    'computation': {
        'random_seed': 0,
        'num_workers': 40,
        'use_gpu': True,
        'num_gpus': None,
        'dist_backend': 'nccl',
        'dist_init_method': 'tcp://localhost:9967',
        'use_distributed': True,  # contrastive: one worker per GPU, gradients averaged (run_contrastive.py:56-60,118-168)
        'load_async': False,
        'concurrent_chunks': 1,  # ours: > 1 selects from that many chunks in lockstep on one GPU (own RNG stream each)
    },
    'subset': {'ratio': 0.2, 'size': None},
    'clustering': {'pairing': 'combination'},
    'batch': {'batch_size': 20, 'selection_size': 4, 'keep_unselected': True},
    'contrastive': {'num_epochs': 3, 'num_warmup_steps': 1, 'base_lr': 2e-4, 'train_batch_size': 128, 'test_batch_size': 128,
                    'cached_epoch': None, 'train_from_cached': False},
    'measure_name': 'batch_mi',
    'shuffle_candidates': True,
    'chunk_size': None,
    'save_cache_as_csvs': True,
    'log_every': 1000,
    'log_times': 10,
    'verbose': True,
    'debug': False,
}
