"""Driver of one selection: the public functions of subset_selection/code/run_greedy.py (`_run_greedy`, `run_greedy`,
same positional arguments) on top of the GPU measures.

Plan of a run (reference run_greedy.py:9-74):
  C            = max label + 1                      (not K: labels never used by a clustering are not counted)
  subset size  = given, or round(ratio * V)
  B, k         = batch.batch_size clamped to V - 1, batch.selection_size clamped to B
  candidates   = 0..V-1, shuffled with Python's `random` when asked; the first one becomes the start index
  measure      = get_measure(name)(...); measure.init(pairs, candidates); measure.run_greedy(...)
`_prepare` builds everything up to the measure call so that several chunks can be prepared first and then selected
in lockstep (run.py).
"""
import numpy as np

from .measures import get_measure
from .pairing import get_cluster_pairing


class _Plan:
    """sizes of one selection, derived once from the assignment matrix and the `batch` options"""

    def __init__(self, args, assignments, subset_size, subset_ratio):
        self.rows = int(assignments.shape[0])
        self.ncentroids = int(assignments.max()) + 1
        self.subset = round(subset_ratio * self.rows) if subset_size is None else subset_size
        self.batch = min(args.batch.batch_size, self.rows - 1)
        self.select = min(args.batch.selection_size, self.batch)

    def candidate_order(self, shuffle):
        if shuffle:
            print("shuffling candidates")
            from ..rng import python_shuffled_range
            return python_shuffled_range(self.rows)  # random.shuffle: Python's generator, not torch's (run_greedy.py:40)
        return np.arange(self.rows, dtype=np.int64)


def _prepare(args, assignments, clustering_types, subset_size, subset_ratio, measure_name='mi',
             cluster_pairing='combination', shuffle_candidates=True, verbose=False, generator=None):
    """-> (measure ready to run, start_indices, subset_size)"""
    plan = _Plan(args, assignments, subset_size, subset_ratio)
    if verbose:
        print("extracting {} samples from {} total datapoints".format(plan.subset, plan.rows))
    options = dict(ncentroids=plan.ncentroids, batch_size=plan.batch, selection_size=plan.select,
                   device=args.computation.device, keep_unselected=args.batch.keep_unselected)
    if generator is not None:
        options['generator'] = generator
    measure = get_measure(measure_name)(assignments, **options)
    order = plan.candidate_order(shuffle_candidates)
    head, rest = order[:1], order[1:]  # a singleton start: it seeds the tables, it is never selected (batch.py:205-206)
    measure.init(get_cluster_pairing(clustering_types, cluster_pairing), rest)
    return measure, head, plan.subset


def _run_greedy(args, assignments, clustering_types, subset_size, subset_ratio, measure_name='mi',
                cluster_pairing='combination', shuffle_candidates=True, verbose=False):
    measure, head, subset = _prepare(args, assignments, clustering_types, subset_size, subset_ratio, measure_name,
                                     cluster_pairing, shuffle_candidates, verbose)
    picked, gains, seconds, _lookups = measure.run_greedy(subset, head, None, verbose=verbose,
                                                         log_every=args.log_every, log_times=args.log_times,
                                                         node_rank=args.node_rank, pid=args.parent_pid)
    return picked, gains, seconds


def run_greedy(args, assignments, shard_names, filenames, clustering_types, subset_size, subset_ratio,
               measure_name='mi', cluster_pairing='combination', shuffle_candidates=True, verbose=False):
    """-> rows {'filename', 'shard_name'} of the selected clips, ordered by clip index (run_greedy.py:72)"""
    picked, _, _ = _run_greedy(args, assignments, clustering_types, subset_size, subset_ratio, measure_name,
                               cluster_pairing, shuffle_candidates, verbose)
    return [dict(filename=filenames[i], shard_name=shard_names[i]) for i in sorted(picked)]
