"""run_greedy -- subset_selection/code/run_greedy.py:9-74 on top of the GPU measure."""
import random

from .measures import get_measure
from .pairing import get_cluster_pairing


def _prepare(args, assignments, clustering_types, subset_size, subset_ratio, measure_name='mi',
             cluster_pairing='combination', shuffle_candidates=True, verbose=False, generator=None):
    """Everything of _run_greedy (run_greedy.py:9-48) up to the measure's run_greedy call:
    -> (measure, start_indices, subset_size)."""
    ncentroids = int(assignments.max()) + 1  # C = max label + 1, not K (run_greedy.py:20)
    dataset_size = assignments.shape[0]
    if subset_size is None:
        subset_size = round(subset_ratio * dataset_size)
    if verbose:
        print("extracting {} samples from {} total datapoints".format(subset_size, dataset_size))
    clustering_combinations = get_cluster_pairing(clustering_types, cluster_pairing)

    batch_size = min(args.batch.batch_size, dataset_size - 1)
    selection_size = min(args.batch.selection_size, batch_size)

    extra = {} if generator is None else {'generator': generator}
    measure = get_measure(measure_name)(assignments, ncentroids=ncentroids, batch_size=batch_size,
                                        selection_size=selection_size, device=args.computation.device,
                                        keep_unselected=args.batch.keep_unselected, **extra)

    candidates = list(range(dataset_size))
    if shuffle_candidates:
        print("shuffling candidates")
        random.shuffle(candidates)  # Python's RNG, as the reference (run_greedy.py:40)

    # start with singleton: it seeds the tables but is never part of S (batch.py:205-206)
    start_indices = [candidates[0]]
    candidates = candidates[1:]

    measure.init(clustering_combinations, candidates)
    return measure, start_indices, subset_size


def _run_greedy(args, assignments, clustering_types, subset_size, subset_ratio, measure_name='mi',
                cluster_pairing='combination', shuffle_candidates=True, verbose=False):
    measure, start_indices, subset_size = _prepare(args, assignments, clustering_types, subset_size, subset_ratio,
                                                   measure_name, cluster_pairing, shuffle_candidates, verbose)
    S, GAIN, timelapse, LOOKUPS = measure.run_greedy(
        subset_size, start_indices, None, verbose=verbose, log_every=args.log_every, log_times=args.log_times,
        node_rank=args.node_rank, pid=args.parent_pid)
    return S, GAIN, timelapse


def run_greedy(args, assignments, shard_names, filenames, clustering_types, subset_size, subset_ratio,
               measure_name='mi', cluster_pairing='combination', shuffle_candidates=True, verbose=False):
    S, GAIN, timelapse = _run_greedy(args, assignments, clustering_types, subset_size, subset_ratio,
                                     measure_name, cluster_pairing, shuffle_candidates, verbose)
    S = sorted(list(S))
    return [{'filename': filenames[s], 'shard_name': shard_names[s]} for s in S]
