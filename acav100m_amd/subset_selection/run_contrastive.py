"""The contrastive baseline's drivers (subset_selection/code/run_contrastive.py:16-110,236-273, merge_contrastive.py):
train the audio-visual InfoNCE model on the feature shards, score every clip, write the per-process inference cache;
`merge_contrastive` sorts the caches by score and writes output.csv.  Host orchestration around
measures/contrastive.Contrastive; single process per run like the reference's non-distributed path."""
import os
import shutil
from pathlib import Path

import numpy as np

from .. import shards as io
from .measures.contrastive import Contrastive

VIDEO_KEY, AUDIO_KEY, LAYER = ('SLOWFAST_8x8_R50', 'kinetics-400'), ('VGGish', 'YouTube-8M'), 'layer_4'


def _penultimate_views(table):
    """the two feature matrices the reference model reads (contrastive.py:111-112): batch['SLOWFAST_8x8_R50/kinetics-400']
    ['layer_4'] and batch['VGGish/YouTube-8M']['layer_4']; other extractors: the last layer of the video / audio model"""
    def pick(kind, want):
        views = [(v, m) for v, m in table.views.items() if v[0] == kind]
        assert views, "no {} features in the shards".format(kind)
        for v, m in views:
            if table.tags.get((v[0], v[1])) == want and v[2] == LAYER:
                return m
        return sorted(views, key=lambda vm: vm[0][2])[-1][1]
    return pick('video', VIDEO_KEY), pick('audio', AUDIO_KEY)


def feature_batches(table, batch_size):
    """-> (visual [n, vis], audio [n, aud], offsets, rows): consecutive batches of `batch_size` clips in shard order with
    duplicate clips (same file stem) dropped inside a batch, first occurrence kept (contrastive.py:103-115)."""
    visual, audio = _penultimate_views(table)
    keep, offsets = [], [0]
    n = len(table)
    for s in range(0, n, batch_size):
        seen = set()
        for r in range(s, min(n, s + batch_size)):
            stem = Path(table.filename[r]).stem
            if stem not in seen:
                seen.add(stem)
                keep.append(r)
        offsets.append(len(keep))
    idx = np.asarray(keep, np.int64)
    rows = [{'id': Path(table.filename[r]).stem, 'filename': table.filename[r], 'shard_name': table.shard_name[r]} for r in keep]
    return np.ascontiguousarray(visual[idx]), np.ascontiguousarray(audio[idx]), np.asarray(offsets, np.int64), rows


def copy_measure(measure, prev_measure):
    """run_contrastive.py:63-69: ONLY the two weight matrices (and the epoch) travel into the new object -- the biases stay
    the freshly initialised ones of `measure`"""
    sd, prev = measure.state_dict(), prev_measure.state_dict()
    for k in ('visual_linear.weight', 'audio_linear.weight'):
        sd[k] = prev[k]
    measure.load_state_dict(sd)
    measure.epoch = prev_measure.epoch
    return measure


def _distributed(args):
    """run_contrastive.py:56-60,88-93: `computation.use_distributed` (the reference's default) sends training and inference
    through one worker per GPU.  Here the workers ARE the processes of the run (one per GPU, cli.main / torchrun): with a
    single process the flag changes nothing but which process draws the biases the reference's spawned child draws from its
    own, unseeded generator -- not reproducible there either."""
    import torch.distributed as dist
    return (bool(getattr(args.computation, 'use_distributed', False)) and dist.is_available() and dist.is_initialized()
            and dist.get_world_size() > 1)


def _new_measure(args, sizes):
    cfg = args.contrastive
    m = Contrastive(cfg.num_epochs, args.computation.device, cfg.base_lr, cfg.num_warmup_steps, distributed=_distributed(args),
                    sizes=sizes)
    if m.distributed:
        from ..parallel.rccl_comm import default_comm
        comm = default_comm()  # None under gloo: the gradients take the torch.distributed route
        if comm is not None:
            m.set_comm(comm)
    return m


def _train(args, paths, batches, sizes, measure=None):
    """run_contrastive.py:72-86: a NEW Contrastive (its own nn.Linear draws from the global generator), the weights of
    the cached model copied in when there is one, then the training loop"""
    prev = measure
    measure = _new_measure(args, sizes)
    if prev is not None:
        measure = copy_measure(measure, prev)
    print("training contrastive loss" + (" with distributed" if measure.distributed else ""))
    measure.train(args, paths, batches, args.log_every, args.verbose)
    if measure.distributed:  # _train_distributed (:156-168): the master leaves the trained model for the parent to load
        from ..parallel import world
        if world()[0] == 0:
            import torch
            cache_dir = Path(args.data.output.path).parent / 'caches'
            cache_dir.mkdir(parents=True, exist_ok=True)
            torch.save({'base_lr': measure.base_lr, 'model': {k: torch.from_numpy(v) for k, v in measure.state_dict().items()}},
                       cache_dir / "contrastive_trained_model_cache_{}.pkl".format(args.parent_pid))
    return measure


def _run(args, paths):
    """run_contrastive.py:16-51,97-116 object for object: the reference builds one Contrastive in _run (first pair of
    nn.Linear draws), trains a SECOND one in _train (second pair of draws) and scores with a THIRD one in _infer whose
    weights are copied from the trained model while its biases are its own fresh draws (copy_measure, :63-69) -- a seeded
    run ranks the clips with exactly those parameters.  (The shuffled DataLoader order of the reference is not
    reproduced: batches are taken in shard order.)"""
    cfg = args.contrastive
    table = io.load_feature_shards([Path(p) for p in paths])
    visual, audio, offsets, rows = feature_batches(table, int(cfg.train_batch_size))
    sizes = (visual.shape[1], audio.shape[1])
    batches = (visual, audio, offsets)
    measure = _new_measure(args, sizes)
    if cfg.cached_epoch is not None:
        cache_path = measure.get_cache_path_load(args, paths, cfg.cached_epoch)
        if cache_path is not None and Path(cache_path).is_file():
            if args.verbose:
                print("cache file found: {}".format(Path(cache_path).stem))
                print("loading from cached file")
            measure.load_cache(args, paths, cfg.cached_epoch)
            if cfg.train_from_cached:
                if args.verbose:
                    print("training from cached file")
                measure = _train(args, paths, batches, sizes, measure)
        else:
            if args.verbose:
                print("no cache file found")
                print("training from scratch")
            measure = _train(args, paths, batches, sizes)
    else:
        measure = _train(args, paths, batches, sizes)
    scorer = copy_measure(_new_measure(args, sizes), measure)  # _infer (:97-103)
    if scorer.distributed:
        from ..parallel import world
        args.node_rank = world()[0]  # the inference cache of this process (contrastive.py:243-246: du.get_rank())
        print("inferring with distributed")
    print("(node {}) running inference".format(args.node_rank or 0))
    tv, ta, toff, trows = feature_batches(table, int(cfg.test_batch_size))
    metas = io.load_metas([Path(p) for p in paths], args.data.meta.path)
    # (distributed: `ids` index the rows THIS worker scored -- the third value -- not the full list)
    scores, ids, scored_rows = scorer.infer(args, (tv, ta, toff), trows, metas, args.subset.size, args.verbose)
    print("(node {}) done inference".format(args.node_rank or 0))
    return measure, scores, [scored_rows[i] for i in ids]


def run_single_contrastive(args):
    """run_contrastive.py:236-247: the whole dataset is scored (subset.size = None, ratio = 1); the selection is made
    afterwards by `merge_contrastive` from the inference caches"""
    args.parent_pid = str(args.parent_pid or os.environ.get('ACAV_PARENT_PID') or os.getpid())  # one name for all workers
    args.node_rank = 0 if args.node_rank is None else args.node_rank
    args.chunk_num = 0 if args.chunk_num is None else args.chunk_num
    paths = [p for p in sorted(io.brace_expand(args.data.path)) if Path(p).is_file()]
    args.subset.size = None
    args.subset.ratio = 1.0
    out = _run(args, paths)
    print("done")
    return out


def run_chunks_contrastive(args):
    """chunk_contrastive.py:17-49,115-131 for THIS process's rank (one process per GPU): the shards are cut into chunks of
    `chunk_size`, every rank takes a contiguous block of them (utils.split_chunks) and runs the whole contrastive flow --
    train a model on the chunk, score the chunk's clips -- chunk after chunk; every chunk appends its scores to the rank's
    inference cache, `merge_contrastive` ranks all of them afterwards.
    Deviation, on purpose: the reference cannot finish this mode -- its run_chunk unpacks the `None` that run_contrastive._run
    returns (chunk_contrastive.py:146-148 vs run_contrastive.py:45-51: the `return` sits inside a string literal) and raises
    after the first chunk of every rank.  Here every chunk is processed; the files written per chunk are the reference's."""
    import copy
    import math
    from ..parallel import world
    args.parent_pid = str(args.parent_pid or os.environ.get('ACAV_PARENT_PID') or os.getpid())
    paths = [p for p in sorted(io.brace_expand(args.data.path)) if Path(p).is_file()]
    chunks = list(enumerate(io.chunked(paths, int(args.chunk_size))))
    num_chunks = len(chunks)
    rank, w = world()
    gpus = max(1, min(w, num_chunks))
    if w > num_chunks:
        print("num_gpus ({}) exceeds num_chunks ({})".format(w, num_chunks))
        print("thresholding num_gpus to be equal to num_chunks")
    chunk_args = copy.deepcopy(args)
    if isinstance(chunk_args.subset.size, int):
        chunk_args.subset.size = math.ceil(chunk_args.subset.size / num_chunks)
    print("running {} chunks in {} gpus".format(num_chunks, gpus))
    per = math.ceil(num_chunks / gpus)
    mine = chunks[rank * per:(rank + 1) * per] if rank < gpus else []
    chunk_args.node_rank = rank
    done = []
    for i, (num, chunk) in enumerate(mine):
        print("running chunk {}".format(num))
        one = copy.deepcopy(chunk_args)
        one.chunk_num = i
        one.computation.use_distributed = False  # a chunk is one process's job from training to scoring
        _run(one, chunk)
        done.append(num)
    print("done")
    return done


def merge_contrastive(args):
    """merge_contrastive.py:107-131 without the shell: concatenate the inference caches of the run with the most files,
    sort by score descending (then by the rest of the line, duplicates of the sort key dropped: `sort -t , -u -k 1,1gr
    -k 2`), drop the score column, drop repeated lines, write output.csv."""
    out_path = Path(args.data.output.path)
    cache_dir = out_path.parent / 'caches'
    groups = {}
    for p in cache_dir.glob("{}_contrastive_inferred_cache_*_*.csv".format(out_path.stem)):
        groups.setdefault('_'.join(p.stem.split('_')[:-1]), []).append(p)
    assert groups, "no inference caches under {}".format(cache_dir)
    most = max(len(v) for v in groups.values())
    name, paths = [(k, v) for k, v in groups.items() if len(v) == most][0]
    print("loading from cache {}".format(name))
    print("{} total cache files".format(len(paths)))
    lines = []
    for p in sorted(paths):
        with open(p) as f:
            lines.extend(line.rstrip('\n') for line in f if line.strip())
    merged = cache_dir / ('merged_' + out_path.name)
    merged.write_text(''.join(line + '\n' for line in lines))

    def key(line):
        score, rest = line.split(',', 1)
        return (-float(score), rest)
    ordered, seen = [], set()
    for line in sorted(lines, key=key):
        k = key(line)
        if k not in seen:
            seen.add(k)
            ordered.append(line)
    (cache_dir / ('sorted_' + out_path.name)).write_text(''.join(line + '\n' for line in ordered))
    unique, seen = [], set()
    for line in ordered:
        rest = line.split(',', 1)[1]
        if rest not in seen:
            seen.add(rest)
            unique.append(rest)
    final = cache_dir / ('unique_' + out_path.name)
    final.write_text(''.join(line + '\n' for line in unique))
    shutil.copy(final, out_path)
    print("done")
    return out_path, len(unique)
