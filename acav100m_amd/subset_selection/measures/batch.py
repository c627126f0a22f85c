"""EfficientBatchMI -- drop-in for subset_selection/code/measures/batch.py:10-260 (on top of
EfficientMI, measures/mi.py:14-148).

Same constructor / init / run_greedy surface as the reference so run_greedy._run_greedy
(run_greedy.py:31-53) drives it unchanged.  The contingency tables, the per-iteration
permutation of the candidate list (torch.randperm semantics on the same MT19937 stream), the
delta-MI scoring, top-k, cache update and re-queue all run on the GPU inside libacav_hip.so; see
acav100m_amd/csrc/acav_mi.hip.  No CPU path.
"""
import ctypes as C
import math
import time

import numpy as np

from ... import _lib
from ...rng import default_generator


def _device_index(device):
    s = str(device)
    if s == "cpu":
        raise _lib.AcavError("acav100m_amd EfficientBatchMI has no CPU path: use device='cuda'")
    if ":" in s:
        return int(s.split(":")[1])
    try:
        import torch
        return torch.cuda.current_device()
    except Exception:
        return 0


class EfficientBatchMI:
    """ this implementation requires the users to use the same ncentroids for all clusterings """

    def __init__(self, assignments, measure_type='mutual_info', average_method='arithmetic',
                 ncentroids=20, batch_size=1, selection_size=1, device='cpu', keep_unselected=False,
                 generator=None, **kwargs):
        self.average_method = average_method.lower()
        self.ncentroids = int(ncentroids)
        self.assignments = np.ascontiguousarray(assignments, dtype=np.int64)  # V x D
        self.eps = np.finfo('float64').eps
        self.B = batch_size
        self.k = selection_size
        self.device = device
        self.keep_unselected = keep_unselected
        self._generator = generator if generator is not None else default_generator
        self._h = None
        self.trace = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and _lib._lib is not None:  # module globals die first at interpreter exit
            _lib._lib.acav_mi_destroy(h)

    # ------------------------------------------------------------------ init (mi.py:27-39)
    def init(self, clustering_combinations, candidates):
        self.combinations = clustering_combinations
        self.init_cache()
        self.init_candidates(candidates)

    def init_cache(self):
        lib = _lib.load_library()
        if self._h is not None:
            lib.acav_mi_destroy(self._h)
            self._h = None
        # the reference only ever uses columns 0 and 1 of a combination (pair_ids[:, 0], pair_ids[:, 1], batch.py:47-48):
        # bipartite pairing over three or more model groups yields longer tuples
        combos = [tuple(c) for c in self.combinations]
        if not combos or any(len(c) < 2 for c in combos):
            raise ValueError("every clustering combination needs at least two clustering indices, got {}".format(combos[:4]))
        pairs = np.ascontiguousarray([c[:2] for c in combos], dtype=np.int32)
        self._npairs = len(pairs)
        v, d = self.assignments.shape
        h = C.c_void_p()
        _lib.check(lib.acav_mi_create(C.byref(h), _device_index(self.device), _lib.ptr(self.assignments), v, d,
                                      self.ncentroids, _lib.ptr(pairs), len(pairs), None))
        self._h = h

    def init_candidates(self, candidates):
        self.candidate_ids = np.ascontiguousarray(candidates, dtype=np.int64)

    @property
    def cache(self):
        """{'N','a','b','n'} integer contingency tables (the reference holds them as fp32 + eps)."""
        p, c = self._npairs, self.ncentroids
        N = np.empty((p, c, c), np.int32)
        a = np.empty((p, c), np.int32)
        b = np.empty((p, c), np.int32)
        n = C.c_int64(0)
        _lib.check(_lib._lib.acav_mi_get_counts(self._h, _lib.ptr(N), _lib.ptr(a), _lib.ptr(b), C.byref(n)))
        return {'N': N, 'a': a, 'b': b, 'n': n.value}

    def add_samples(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        _lib.check(_lib._lib.acav_mi_add_samples(self._h, _lib.ptr(ids), len(ids)))

    def score_batch(self, ids):
        """scores.mean(-1) of operate_block for the given candidate ids (batch.py:123-130,144)."""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        out = np.empty(len(ids), np.float64)
        _lib.check(_lib._lib.acav_mi_score_batch(self._h, _lib.ptr(ids), len(ids), _lib.ptr(out)))
        return out

    def get_batch_ranges(self):
        return [[0, self.B]]  # batch.py:56-87: the O(1) scoring never needs GPU-memory chunking

    def modify_k(self, subset_size):
        """batch.py:173-188"""
        D = self.assignments.shape[0]
        S, K, B = subset_size, self.k, self.B
        term = B * S / D
        if K < term and not self.keep_unselected:
            print("k={} is too small to get {} samples from {} datapoints with batch_size {}".format(K, S, D, B))
            K = math.ceil(term)
            print("resizing k to {}".format(K))
        return K

    # ---------------------------------------------------------- run_greedy (batch.py:195-260)
    def run_greedy(self, subset_size, start_indices, intermediate_target=None, verbose=False, log_every=1,
                   log_times=None, node_rank=None, pid=None, record_trace=False, forced_pos=None,
                   max_iters=-1):
        print('using {} blocks per iter for gpu computation'.format(len(self.get_batch_ranges())))
        self.k = self.modify_k(subset_size)
        B, k = int(self.B), int(self.k)
        start = np.ascontiguousarray(start_indices, dtype=np.int64)
        cand = self.candidate_ids
        niters = math.ceil(subset_size / k) if max_iters < 0 else min(math.ceil(subset_size / k), int(max_iters))
        S = np.empty(niters * k + k, np.int64)
        G = np.empty(niters * k + k, np.float64)
        tr_ids = np.empty((niters, B), np.int64) if record_trace else None
        tr_sc = np.empty((niters, B), np.float64) if record_trace else None
        tr_pos = np.empty((niters, k), np.int32) if record_trace else None
        fp = None if forced_pos is None else np.ascontiguousarray(forced_pos, dtype=np.int32)
        nsel, nit = C.c_int64(0), C.c_int64(0)
        greedy_start_time = time.time()
        _lib.check(_lib._lib.acav_mi_run_greedy(
            self._h, _lib.ptr(cand), len(cand), _lib.ptr(start), len(start), int(subset_size), B, k,
            int(bool(self.keep_unselected)), self._generator.handle, _lib.ptr(S), _lib.ptr(G), C.byref(nsel),
            C.byref(nit), _lib.ptr(tr_ids), _lib.ptr(tr_sc), _lib.ptr(tr_pos), _lib.ptr(fp), int(max_iters)))
        elapsed = time.time() - greedy_start_time
        nit, nsel = nit.value, nsel.value
        if record_trace:
            self.trace = dict(ids=tr_ids[:nit], scores=tr_sc[:nit], pos=tr_pos[:nit])
        S_list = S[:nsel].tolist()  # S = S[:subset_size]  (batch.py:258)
        GAIN = G[:nit * k].tolist()
        timelapse = [elapsed / max(nit, 1)] * nit
        LOOKUPS = [1] * nit
        if verbose:
            msg = "(LEN: {}/{}, MEASURE: {})".format(len(S_list), subset_size,
                                                     float(np.mean(GAIN[-k:])) if GAIN else float('nan'))
            if node_rank is not None:
                msg = 'Node: {}, '.format(node_rank) + msg
            print(msg)
        print("Time Consumed: {} seconds".format(elapsed))
        return (S_list, GAIN, timelapse, LOOKUPS)


    # ------------------------------------------------------------ several chunks in lockstep
    @staticmethod
    def run_greedy_multi(measures, subset_sizes, start_indices_list, verbose=False):
        """run_greedy for several independent measures (one per chunk, chunk.py:21-53) with ONE set of kernel
        launches per iteration (acav_mi_run_greedy_multi).  Every measure needs its own generator and must share
        batch_size / selection_size / keep_unselected; element i of the result is what
        measures[i].run_greedy(subset_sizes[i], start_indices_list[i]) returns."""
        n = len(measures)
        assert n > 0 and len(subset_sizes) == n and len(start_indices_list) == n
        m0 = measures[0]
        for m, sub in zip(measures, subset_sizes):
            m.k = m.modify_k(sub)
        B, k = int(m0.B), int(m0.k)
        assert all(int(m.B) == B and int(m.k) == k and bool(m.keep_unselected) == bool(m0.keep_unselected)
                   for m in measures), "chunks run in lockstep must share batch_size / selection_size / keep_unselected"
        assert len({id(m._generator) for m in measures}) == n, "every chunk needs its own generator"
        cands = [m.candidate_ids for m in measures]
        starts = [np.ascontiguousarray(s, dtype=np.int64) for s in start_indices_list]
        niters = [math.ceil(int(s) / k) for s in subset_sizes]
        S = [np.empty(it * k + k, np.int64) for it in niters]
        G = [np.empty(it * k + k, np.float64) for it in niters]

        def parr(ptrs):
            return (C.c_void_p * n)(*[p.value if isinstance(p, C.c_void_p) else p for p in ptrs])

        h_arr = parr([m._h for m in measures])
        c_arr = parr([_lib.ptr(c) for c in cands])
        s_arr = parr([_lib.ptr(s) if len(s) else None for s in starts])
        r_arr = parr([m._generator.handle for m in measures])
        S_arr = parr([_lib.ptr(a) for a in S])
        G_arr = parr([_lib.ptr(a) for a in G])
        L = np.array([len(c) for c in cands], np.int64)
        ns = np.array([len(s) for s in starts], np.int32)
        sub = np.array([int(s) for s in subset_sizes], np.int64)
        nsel = np.zeros(n, np.int64)
        nit = np.zeros(n, np.int64)
        t0 = time.time()
        _lib.check(_lib._lib.acav_mi_run_greedy_multi(h_arr, n, c_arr, _lib.ptr(L), s_arr, _lib.ptr(ns), _lib.ptr(sub), B, k,
                                                      int(bool(m0.keep_unselected)), r_arr, S_arr, G_arr, _lib.ptr(nsel),
                                                      _lib.ptr(nit)))
        elapsed = time.time() - t0
        if verbose:
            print("Time Consumed: {} seconds for {} chunks in lockstep".format(elapsed, n))
        out = []
        for i in range(n):
            it = int(nit[i])
            out.append((S[i][:int(nsel[i])].tolist(), G[i][:it * k].tolist(), [elapsed / max(it, 1)] * it, [1] * it))
        return out
