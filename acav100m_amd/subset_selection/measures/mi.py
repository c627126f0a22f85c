"""EfficientMI / EfficientMemMI -- the reference's exact-greedy measures 'mi' and 'mem_mi'
(subset_selection/code/measures/mi.py:14-207, 284-412; registry measures/__init__.py:5-14).

Every iteration scores ALL remaining candidates against the current contingency tables, commits the first
maximum and removes it (mi.py:76-114).  The reference evaluates this densely -- `mi` on [W,P,C,C] fp32 tensors,
`mem_mi` through running fp32 n-log-n sums -- and its two measures already disagree with each other on near-ties
(tests/golden/gen_golden.py: S equivalence 12-66 %).  Both are the same function of the integer tables; here that
function is the canonical float64 closed form of libacav_hip.so (one kernel launch per iteration, all candidates
scored in parallel on the GPU).  Same constructor / init / run_greedy surface as the reference, so
run_greedy._run_greedy drives it unchanged.  No CPU path.
"""
import ctypes as C
import time

import numpy as np

from ... import _lib
from .batch import EfficientBatchMI


class EfficientMI(EfficientBatchMI):
    """ this implementation requires the users to use the same ncentroids for all clusterings """

    def __init__(self, assignments, measure_type='mutual_info', average_method='arithmetic', ncentroids=20,
                 device='cuda', **kwargs):
        kwargs.pop('batch_size', None)
        kwargs.pop('selection_size', None)
        kwargs.pop('keep_unselected', None)
        super().__init__(assignments, measure_type=measure_type, average_method=average_method,
                         ncentroids=ncentroids, batch_size=1, selection_size=1, device=device, **kwargs)

    # calc_measure (mi.py:108-114) for callers that step the greedy themselves
    def calc_measure(self):
        S, G = self._run(2, 0, None, False)
        return G[0], S[0]

    def _run(self, subset_size, ns, forced_pos, record_trace):
        cand = self.candidate_ids
        L = len(cand)
        niters = max(0, min(int(subset_size) - 1 - int(ns), L))
        S = np.empty(niters + 1, np.int64)
        G = np.empty(niters + 1, np.float64)
        fp = None if forced_pos is None else np.ascontiguousarray(forced_pos, np.int64)
        tr_sc = np.empty((niters, L), np.float64) if record_trace else None
        tr_am = np.empty(niters + 1, np.int64) if record_trace else None
        nsel = C.c_int64(0)
        _lib.check(_lib._lib.acav_mi_run_exact(self._h, _lib.ptr(cand), L, int(ns), int(subset_size), _lib.ptr(S),
                                               _lib.ptr(G), C.byref(nsel), _lib.ptr(fp), _lib.ptr(tr_sc),
                                               _lib.ptr(tr_am)))
        n = nsel.value
        if record_trace:
            self.trace = dict(scores=tr_sc[:n], argmax=tr_am[:n].copy())
        # the candidate list of the reference shrinks as it goes (remove_idx_all, mi.py:104-106)
        picked = set(S[:n].tolist())
        if picked:
            self.candidate_ids = np.ascontiguousarray([c for c in cand.tolist() if c not in picked], np.int64)
        return S[:n].tolist(), G[:n].tolist()

    def run_greedy(self, subset_size, start_indices, intermediate_target=None, verbose=False, log_every=1,
                   log_times=None, node_rank=None, pid=None, record_trace=False, forced_pos=None):
        """mi.py:150-192: returns (S, GAIN, timelapse, LOOKUPS) with S = start_indices + the picks.
        forced_pos: ORIGINAL positions (indices into the candidate list given to init) to commit instead of the
        argmax -- replays a recorded run."""
        start = list(start_indices)
        t0 = time.time()
        S, GAIN = self._run(subset_size, len(start), forced_pos, record_trace)
        elapsed = time.time() - t0
        n = len(S)
        if verbose:
            msg = "(LEN: {}, MEASURE: {})".format(len(start) + n, GAIN[-1] if GAIN else float('nan'))
            if node_rank is not None:
                msg = 'Node: {}, '.format(node_rank) + msg
            print(msg)
            print("Time Consumed: {} seconds".format(elapsed))
        return (start + S, GAIN, [elapsed / max(n, 1)] * n, [0] * n)


class EfficientMemMI(EfficientMI):
    """mi.py:284-412: the memory-lean formulation of the same greedy; identical here."""


class EfficientAMI(EfficientMI):
    """adjusted MI (mi.py:212-259; 'ami' in measures/__init__.py:5-14): the exact greedy on
    (MI - EMI) / max(mean entropy - EMI, eps), EMI being the reference's own one-term-per-cell expression (calc_EMI).
    Same kernel as `mi` with the adjusted score (acav_mi_set_measure); float64 over integer counts, within 4e-7 relative of
    the reference's fp32 scores (tests/golden/mi_ami_*.npz).  average_method: 'arithmetic' (the reference default) only."""

    def init(self, clustering_combinations, candidates):
        assert self.average_method == 'arithmetic', "ami: only the reference's default average_method is built"
        super().init(clustering_combinations, candidates)
        _lib.check(_lib._lib.acav_mi_set_measure(self._h, 1))


class EfficientNMI(EfficientAMI):
    """normalised MI (mi.py:262-271): the exact greedy on 2 MI / max(mean entropy, eps).  The reference defines the class but
    its registry (measures/__init__.py:5-14) does not name it; here it is reachable as 'nmi'.  Same kernel, score 2
    (acav_mi_set_measure); float64 over integer counts, pinned on the reference class's own run (tests/golden/mi_nmi_*.npz)."""

    def init(self, clustering_combinations, candidates):
        assert self.average_method == 'arithmetic', "nmi: only the reference's default average_method is built"
        EfficientMI.init(self, clustering_combinations, candidates)
        _lib.check(_lib._lib.acav_mi_set_measure(self._h, 2))


class ConstantMeasure(EfficientMI):
    """mi.py:274-281: every candidate scores 1, so the exact greedy takes the first remaining candidate every iteration and
    every gain is 1.0 -- the reference's "no measure" control ('constant' here; not in the reference's registry either)."""

    def init(self, clustering_combinations, candidates):
        super().init(clustering_combinations, candidates)
        _lib.check(_lib._lib.acav_mi_set_measure(self._h, 3))
