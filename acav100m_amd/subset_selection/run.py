"""Selection stage drivers: run_single / run_partition (subset_selection/code/run.py:6-48) and the
chunked mode (chunk.py:21-53,95-131) -- host orchestration around the GPU measure.

Chunks are independent (no collective anywhere on this path): with several GPUs each process takes a
contiguous block of ceil(num_chunks / num_gpus) chunks (utils.split_chunks) and writes per-chunk CSVs
to caches/, merged later by `cli.py reduce_csvs`.
"""
import copy
import math
import os
from collections import defaultdict
from pathlib import Path

import numpy as np

from .. import shards as io
from ..parallel import world
from .run_greedy import _prepare, run_greedy


def load_data(shard_paths, metas_path, verbose=False):
    """dataloader.load_data (dataloader.py:152-203): shards grouped into partitions by the clustering
    run manifests (shards without a log form partition -1), plus the metadata."""
    if not isinstance(shard_paths, list):
        shard_paths = io.brace_expand(shard_paths)
    paths = sorted(Path(p) for p in shard_paths if Path(p).is_file())
    if not paths:
        return {}, {}
    parts = io.load_partitions(paths[0].parent)
    grouped = defaultdict(list)
    for p in paths:
        grouped[parts.get(p.stem, -1)].append(p)
    if verbose:
        print("num_shards: {} (clustering_partitions: {})".format(len(paths), {k: len(v) for k, v in grouped.items()}))
    return dict(grouped), io.load_metas(paths, metas_path)


def run_partition(args, shard_paths):
    assignments, clustering_types, shard_names, filenames = io.load_assignment_shards(shard_paths)
    return run_greedy(args, assignments, shard_names, filenames, clustering_types, args.subset.size,
                      args.subset.ratio, measure_name=args.measure_name, cluster_pairing=args.clustering.pairing,
                      shuffle_candidates=args.shuffle_candidates, verbose=args.verbose)


def _load(args, path):
    """chunk.load_and_preprocess (chunk.py:156-175): everything of a chunk that does not need the GPU."""
    partitions, metas = load_data(path, args.data.meta.path, args.verbose)
    data = [(k, io.load_assignment_shards(partitions[k])) for k in sorted(partitions)]
    return data, metas


def _select(args, data):
    results = []
    for k, (assignments, clustering_types, shard_names, filenames) in data:
        print('running partition {}/{}'.format(k, len(data)))
        results.append(run_greedy(args, assignments, shard_names, filenames, clustering_types, args.subset.size,
                                  args.subset.ratio, measure_name=args.measure_name,
                                  cluster_pairing=args.clustering.pairing, shuffle_candidates=args.shuffle_candidates,
                                  verbose=args.verbose))
    return results


def _run(args, path):
    data, metas = _load(args, path)
    return _select(args, data), metas


def run_single(args):
    results, metas = _run(args, args.data.path)
    counts, out_path = 0, None
    for samples in results:
        out_path, count = io.append_output_csv(samples, metas, args.data.output.path)
        counts += count
    if out_path is None:
        print("No files saved")
    if args.verbose:
        print("Saved Results: added {} lines to {}".format(counts, out_path))
    return out_path, counts


def run_chunks(args):
    """chunk.py:21-53 + run_chunks_node :115-131 for THIS process's rank (one process per GPU)."""
    args.parent_pid = str(args.parent_pid or os.environ.get('ACAV_PARENT_PID') or os.getpid())  # chunk.py:22
    paths = [p for p in sorted(io.brace_expand(args.data.path)) if Path(p).is_file()]
    chunks = list(enumerate(io.chunked(paths, int(args.chunk_size))))
    num_chunks = len(chunks)
    rank, w = world()
    gpus = max(1, min(w, num_chunks))
    chunk_args = copy.deepcopy(args)
    if isinstance(chunk_args.subset.size, int):
        chunk_args.subset.size = math.ceil(chunk_args.subset.size / num_chunks)
    per = math.ceil(num_chunks / gpus)
    mine = chunks[rank * per:(rank + 1) * per] if rank < gpus else []
    print("running {} chunks in {} gpus".format(num_chunks, gpus))
    chunk_args.node_rank = rank
    width = int(chunk_args.computation.concurrent_chunks or 1)
    if width > 1:
        return _run_chunks_lockstep(args, chunk_args, mine, rank, width)
    written = []
    # computation.load_async (chunk.py:119-120,197-226): the next chunk's shards are read and parsed by a host
    # thread while the GPU selects from the current one.  Chunks are still consumed in order, so the cache files
    # are the same as in the synchronous mode.
    pool = None
    if chunk_args.computation.load_async and len(mine) > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=1)
    pending = pool.submit(_load, chunk_args, mine[0][1]) if pool else None
    for i, (num, chunk) in enumerate(mine):
        print("running chunk {}".format(num))
        if pool:
            data, metas = pending.result()
            pending = pool.submit(_load, chunk_args, mine[i + 1][1]) if i + 1 < len(mine) else None
            results = _select(chunk_args, data)
        else:
            results, metas = _run(chunk_args, chunk)
        res = results[0] if results else []  # a chunk is assumed to be a single partition (chunk.py:152)
        name = "cache_{}_{}_{}".format(chunk_args.parent_pid, rank, i)
        written.append(_save_chunk(args, name, res, metas))
    if pool:
        pool.shutdown()
    return written


def _save_chunk(args, name, res, metas):
    """chunk.py:125-131: the chunk's selection as caches/{name}_output.csv, or -- save_cache_as_csvs=False -- as the
    pickle caches/{name}.pkl = {'res', 'metas'} that `cli.py reduce_pkls` turns into the csv later"""
    cache_dir = Path(args.data.output.path).parent / 'caches'
    if args.save_cache_as_csvs or args.save_cache_as_csvs is None:
        out_path, _ = io.append_output_csv(res, metas, cache_dir / Path(args.data.output.path).name, name + '_')
        return out_path
    cache_dir.mkdir(parents=True, exist_ok=True)
    out_path = cache_dir / (name + '.pkl')
    io.dump_pickle({'res': res, 'metas': metas}, out_path)
    return out_path


def reduce_all_pkls(args):
    """chunk.py:56-92 (reduce_all_pkls / reduce_single_cache): caches/cache_*_*.pkl -> per-cache csv -> output.csv"""
    cache_dir = Path(args.data.output.path).parent / 'caches'
    groups = defaultdict(list)
    for p in cache_dir.glob('cache_*_*.pkl'):
        groups['_'.join(p.stem.split('_')[:2])].append(p)
    total = 0
    for key in sorted(groups):
        print('processing cache set {}'.format(key))
        outs = []
        for p in sorted(groups[key]):
            print("loading cache ({})".format(p.stem))
            cache = io.load_pickle(p)
            res = cache['res']
            if isinstance(args.subset.size, int):
                res = res[:args.subset.size]
            print("saving cache ({}), subset size: {}".format(p.stem, len(res)))
            out_path, _ = io.append_output_csv(res, cache['metas'], cache_dir / Path(args.data.output.path).name, p.stem + '_')
            outs.append(out_path)
        print("merging csvs")
        total += io.merge_csvs(sorted(outs), args.data.output.path)
    if args.verbose:
        print("Saved Results: added {} lines to {}".format(total, args.data.output.path))
    return total


def compare_measures(args):
    """tests.py:10-46: run several measures on every partition and report how far their selections agree (the
    reference's version stops in a debugger; this one prints and returns the figures)."""
    from .run_greedy import _run_greedy
    names = args.measure_names or ['mem_mi', 'mi']
    partitions, _ = load_data(args.data.path, args.data.meta.path, args.verbose)
    report = []
    for k in sorted(partitions):
        assignments, clustering_types, _, _ = io.load_assignment_shards(partitions[k])
        runs = {}
        for name in names:
            runs[name] = _run_greedy(args, assignments, clustering_types, args.subset.size, args.subset.ratio, measure_name=name,
                                     cluster_pairing=args.clustering.pairing, shuffle_candidates=False, verbose=False)
        keys = list(runs)
        for a in range(len(keys)):
            for b in range(a + 1, len(keys)):
                (sa, ga, _), (sb, gb, _) = runs[keys[a]], runs[keys[b]]
                same = float(np.mean([int(x == y) for x, y in zip(sa, sb)])) if sa and sb else float('nan')
                gd = float(np.mean([abs(x - y) for x, y in zip(ga, gb)])) if ga and gb else float('nan')
                print(keys[a], 'vs.', keys[b])
                print('S equivalence: ', same)
                print('GAIN diff mean: ', gd)
                report.append((k, keys[a], keys[b], same, gd))
    return report


def _run_chunks_lockstep(args, chunk_args, mine, rank, width):
    """computation.concurrent_chunks = width > 1: `width` chunks at a time share ONE set of kernel launches per
    greedy iteration (acav_mi_run_greedy_multi) -- the loop of a single chunk is a chain of small dependent kernels
    and leaves most of the GPU idle.  The reference runs a process's chunks one after the other on one RNG
    stream (chunk.py:115-131); chunks in flight together need a stream each: chunk number `num` draws from
    Generator(computation.random_seed + 1 + num), whichever rank or group it lands in."""
    from ..rng import Generator
    from .measures.batch import EfficientBatchMI
    assert chunk_args.measure_name == 'batch_mi', "lockstep chunks are implemented for the batch_mi measure"
    from concurrent.futures import ThreadPoolExecutor
    base_seed = int(chunk_args.computation.random_seed or 0)
    written = []

    def prepare_group(group):
        """Host side of a group: shard loading, candidate shuffle, tables, device handles.  Runs on a helper thread under the
        PREVIOUS group's greedy loop (the library call releases the GIL; per-chunk generators: nothing is shared)."""
        prepared = []
        for num, chunk in group:
            print("loading chunk {}".format(num))
            data, metas = _load(chunk_args, chunk)
            if not data:
                prepared.append(None)
                continue
            k, (assignments, clustering_types, shard_names, filenames) = data[0]  # single partition (chunk.py:152)
            measure, start, subset = _prepare(chunk_args, assignments, clustering_types, chunk_args.subset.size,
                                              chunk_args.subset.ratio, chunk_args.measure_name,
                                              chunk_args.clustering.pairing, chunk_args.shuffle_candidates,
                                              chunk_args.verbose, generator=Generator(base_seed + 1 + num))
            prepared.append((measure, start, subset, shard_names, filenames, metas))
        return prepared

    groups = [mine[g0:g0 + width] for g0 in range(0, len(mine), width)]
    pool = ThreadPoolExecutor(1)
    nxt = pool.submit(prepare_group, groups[0]) if groups else None
    for gi, group in enumerate(groups):
        g0 = gi * width
        prepared = nxt.result()
        nxt = pool.submit(prepare_group, groups[gi + 1]) if gi + 1 < len(groups) else None
        live = [p for p in prepared if p is not None]
        print("running chunks {} in lockstep".format([num for num, _ in group]))
        results = EfficientBatchMI.run_greedy_multi([p[0] for p in live], [p[2] for p in live], [p[1] for p in live],
                                                    verbose=chunk_args.verbose) if live else []
        ri = 0
        for j, (num, chunk) in enumerate(group):
            i = g0 + j
            res, metas = [], {}
            if prepared[j] is not None:
                S = sorted(results[ri][0])
                ri += 1
                _, _, _, shard_names, filenames, metas = prepared[j]
                res = [{'filename': filenames[s], 'shard_name': shard_names[s]} for s in S]
            name = "cache_{}_{}_{}".format(chunk_args.parent_pid, rank, i)
            cache_out = Path(args.data.output.path).parent / 'caches' / Path(args.data.output.path).name
            out_path, _ = io.append_output_csv(res, metas, cache_out, name + '_')
            written.append(out_path)
    pool.shutdown()
    return written


def merge_all_csvs(args):
    """save.merge_all_csvs (save.py:106-121): caches/cache_{pid}_{rank}_{i}_{name} -> output.csv"""
    cache_dir = Path(args.data.output.path).parent / 'caches'
    name = Path(args.data.output.path).name
    groups = defaultdict(list)
    for p in cache_dir.glob('cache_*_*_{}'.format(name)):
        groups['_'.join(p.stem.split('_')[:2])].append(p)
    total = 0
    for key in sorted(groups):
        print('processing cache set {}'.format(key))
        counts = io.merge_csvs(sorted(groups[key]), args.data.output.path)
        total += counts
        if args.verbose:
            print("Saved Results: added {} lines to {}".format(counts, args.data.output.path))
    return total
