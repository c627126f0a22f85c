"""ctypes view of oracle/libacav_oracle.so -- the CPU restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py.  Nothing under acav100m_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libacav_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "acav_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp, i64, i32, f32, f64, u32 = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_double, C.c_uint32
        sig = {
            "orc_rng_create": (vp, [u32]),
            "orc_rng_seed": (None, [vp, u32]),
            "orc_rng_destroy": (None, [vp]),
            "orc_rng_u32": (u32, [vp]),
            "orc_rng_rand_f32": (None, [vp, vp, i64]),
            "orc_rng_randperm": (None, [vp, i64, vp]),
            "orc_rng_get_state": (None, [vp, vp, vp]),
            "orc_rng_set_state": (None, [vp, vp, i32]),
            "orc_sumsq": (f32, [vp, i32]),
            "orc_norm2": (f32, [vp, i32]),
            "orc_dot": (f32, [vp, vp, i32]),
            "orc_kmeans_create": (vp, [i32, i32, vp, vp]),
            "orc_kmeans_destroy": (None, [vp]),
            "orc_kmeans_get_state": (None, [vp, vp, vp, vp, vp]),
            "orc_kmeans_set_state": (None, [vp, vp, vp, i64, i64]),
            "orc_kmeans_set_hyper": (None, [vp, i32, f64, f64]),
            "orc_kmeans_threshold": (f32, [vp]),
            "orc_kmeans_calc_best": (None, [vp, vp, i64, vp, vp, vp]),
            "orc_kmeans_apply_update": (None, [vp, vp, i64, vp, f64]),
            "orc_kmeans_add": (f32, [vp, vp, i64, f64, vp, vp]),
            "orc_kmeans_train_epoch": (None, [vp, vp, i64, i64, f64, vp]),
            "orc_num_threads": (i32, []),
            "orc_set_threads": (None, [i32]),
            "orc_mi_create": (vp, [vp, i64, i32, i32, vp, i32]),
            "orc_mi_destroy": (None, [vp]),
            "orc_mi_add_samples": (None, [vp, vp, i64]),
            "orc_mi_scores_dense": (None, [vp, vp, i32, vp]),
            "orc_mi_scores_canon": (None, [vp, vp, i32, vp]),
            "orc_topk_desc": (None, [vp, i32, i32, vp]),
            "orc_mi_run_greedy": (i64, [vp, vp, i64, vp, i32, i64, i32, i32, i32, vp, i32, vp, vp, i64, vp, vp, vp, vp]),
            "orc_mi_run_exact": (i64, [vp, vp, i64, vp, i32, i64, vp, vp, vp, vp, vp]),
            "orc_mi_get_counts": (None, [vp, vp, vp, vp, vp]),
            "orc_mi_set_measure": (None, [vp, i32]),
            "orc_mi_scores_ami": (None, [vp, vp, i32, vp]),
            "orc_canon_exp": (f64, [f64]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class Rng:
    """torch's CPU mt19937 generator (manual_seed / rand / randperm)."""

    def __init__(self, seed=0):
        self.h = lib().orc_rng_create(int(seed) & 0xFFFFFFFF)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:  # module globals die first at interpreter exit
            lib().orc_rng_destroy(self.h)
            self.h = None

    def seed(self, s):
        lib().orc_rng_seed(self.h, int(s) & 0xFFFFFFFF)

    def u32(self):
        return int(lib().orc_rng_u32(self.h))

    def rand(self, *shape):
        out = np.empty(shape, np.float32)
        lib().orc_rng_rand_f32(self.h, _p(out), out.size)
        return out

    def randperm(self, n):
        out = np.empty(n, np.int64)
        lib().orc_rng_randperm(self.h, n, _p(out))
        return out

    def get_state(self):
        mt = np.empty(624, np.uint32)
        idx = C.c_int(0)
        lib().orc_rng_get_state(self.h, _p(mt), C.byref(idx))
        return mt, idx.value

    def set_state(self, mt, idx):
        mt = np.ascontiguousarray(mt, np.uint32)
        lib().orc_rng_set_state(self.h, _p(mt), int(idx))


def sumsq(v):
    v = _f32(v)
    return float(lib().orc_sumsq(_p(v), v.size))


def norm2(v):
    v = _f32(v)
    return float(lib().orc_norm2(_p(v), v.size))


def dot(c, x):
    c, x = _f32(c), _f32(x)
    return float(lib().orc_dot(_p(c), _p(x), c.size))


class KMeans:
    """Restatement of clustering/code/sgd_clustering.py:10-129 (non-distributed branch)."""

    def __init__(self, d, k, rng, centers=None, lr=1e-2):
        self.k, self.d, self.rng, self.lr = k, d, rng, lr
        c0 = None if centers is None else _f32(centers)
        self.h = lib().orc_kmeans_create(k, d, _p(c0), rng.h)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:  # module globals die first at interpreter exit
            lib().orc_kmeans_destroy(self.h)
            self.h = None

    def get_state(self):
        centers = np.empty((self.k, self.d), np.float32)
        counts = np.empty(self.k, np.float32)
        count, fb = C.c_int64(0), C.c_int64(0)
        lib().orc_kmeans_get_state(self.h, _p(centers), _p(counts), C.byref(count), C.byref(fb))
        return centers, counts, count.value, fb.value

    def set_state(self, centers, counts, count, fallback=0):
        c = None if centers is None else _f32(centers)
        n = None if counts is None else _f32(counts)
        lib().orc_kmeans_set_state(self.h, _p(c), _p(n), int(count), int(fallback))

    @property
    def centers(self):
        return self.get_state()[0]

    @property
    def counts(self):
        return self.get_state()[1]

    @property
    def count(self):
        return self.get_state()[2]

    @property
    def fallback(self):
        return self.get_state()[3]

    def threshold(self):
        return float(lib().orc_kmeans_threshold(self.h))

    def calc_best(self, x):
        x = _f32(x)
        best = np.empty(len(x), np.int64)
        mean = C.c_float(0)
        lib().orc_kmeans_calc_best(self.h, _p(x), len(x), _p(best), C.byref(mean), self.rng.h)
        return best, mean.value

    def apply_update(self, x, best, lr=None):
        x, best = _f32(x), _i64(best)
        lib().orc_kmeans_apply_update(self.h, _p(x), len(x), _p(best), float(self.lr if lr is None else lr))

    def add(self, x, return_best=False):
        x = _f32(x)
        best = np.empty(len(x), np.int64)
        mean = lib().orc_kmeans_add(self.h, _p(x), len(x), float(self.lr), self.rng.h, _p(best))
        return (mean, best) if return_best else mean

    def train_epoch(self, x, b, lr=None):
        x = _f32(x)
        lib().orc_kmeans_train_epoch(self.h, _p(x), len(x), b, float(self.lr if lr is None else lr), self.rng.h)


class BatchMI:
    """Restatement of EfficientBatchMI (measures/batch.py) + EfficientMI tables (measures/mi.py)."""

    def __init__(self, assignments, ncentroids, pairs):
        a = _i64(assignments)
        self.V, self.D = a.shape
        self.C = int(ncentroids)
        self.pairs = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        self.P = len(self.pairs)
        self.h = lib().orc_mi_create(_p(a), self.V, self.D, self.C, _p(self.pairs), self.P)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:  # module globals die first at interpreter exit
            lib().orc_mi_destroy(self.h)
            self.h = None

    def add_samples(self, ids):
        ids = _i64(ids)
        lib().orc_mi_add_samples(self.h, _p(ids), len(ids))

    def scores_dense(self, ids):
        ids = _i64(ids)
        out = np.empty((len(ids), self.P), np.float32)
        lib().orc_mi_scores_dense(self.h, _p(ids), len(ids), _p(out))
        return out

    def scores_canon(self, ids):
        ids = _i64(ids)
        out = np.empty(len(ids), np.float64)
        lib().orc_mi_scores_canon(self.h, _p(ids), len(ids), _p(out))
        return out

    def set_measure(self, name):
        """exact greedy scores: 'mi' / 'mem_mi' (calc_MI), 'ami' (calc_AMI, mi.py:212-259), 'nmi' (calc_NMI, :262-271) or
        'constant' (ConstantMeasure, :274-281)"""
        lib().orc_mi_set_measure(self.h, {"ami": 1, "nmi": 2, "constant": 3}.get(name, 0))

    def scores_ami(self, ids):
        ids = _i64(ids)
        out = np.empty(len(ids), np.float64)
        lib().orc_mi_scores_ami(self.h, _p(ids), len(ids), _p(out))
        return out

    def counts(self):
        Nc = np.empty((self.P, self.C, self.C), np.int32)
        ac = np.empty((self.P, self.C), np.int32)
        bc = np.empty((self.P, self.C), np.int32)
        nc = C.c_int64(0)
        lib().orc_mi_get_counts(self.h, _p(Nc), _p(ac), _p(bc), C.byref(nc))
        return Nc, ac, bc, nc.value

    def run_greedy(self, candidates, start, subset, B, k, rng, keep_unselected=True, dense=False,
                   max_iters=-1, trace=False, forced_pos=None):
        cand, start = _i64(candidates), _i64(start)
        cap = int(subset) + 2 * k + 8
        S = np.empty(cap, np.int64)
        G = np.empty(cap, np.float64)
        nit_cap = (int(subset) + k - 1) // k + 1
        t_ids = np.empty((nit_cap, B), np.int64) if trace else None
        t_sc = np.empty((nit_cap, B), np.float64) if trace else None
        t_pos = np.empty((nit_cap, k), np.int32) if trace else None
        fp = None if forced_pos is None else np.ascontiguousarray(forced_pos, np.int32)
        if fp is not None:
            max_iters = len(fp) if max_iters < 0 else min(max_iters, len(fp))
        nit = lib().orc_mi_run_greedy(self.h, _p(cand), len(cand), _p(start), len(start), int(subset), B, k,
                                      int(keep_unselected), rng.h, int(dense), _p(S), _p(G), int(max_iters),
                                      _p(t_ids), _p(t_sc), _p(t_pos), _p(fp))
        if nit < 0:
            raise RuntimeError("fewer candidates than batch_size: the reference raises here (batch.py:147-149)")
        n = min(nit * k, int(subset))
        res = dict(S=S[:n].copy(), GAIN=G[:nit * k].copy(), iters=nit)  # S is cut (batch.py:258), GAIN is not
        if trace:
            res.update(ids=t_ids[:nit], scores=t_sc[:nit], pos=t_pos[:nit])
        return res


    def run_exact(self, candidates, start, subset, forced_idx=None, trace=False):
        """EfficientMI / EfficientMemMI exact greedy (mi.py:150-192): S holds the picks AFTER the start indices."""
        cand, start = _i64(candidates), _i64(start)
        niter = max(int(subset) - 1 - len(start), 0)
        S = np.empty(niter + 1, np.int64)
        G = np.empty(niter + 1, np.float64)
        fi = None if forced_idx is None else _i64(forced_idx)
        if fi is not None:
            assert len(fi) >= min(niter, len(cand))
        t_sc = np.empty((niter, len(cand)), np.float64) if trace else None
        t_am = np.empty(niter + 1, np.int64) if trace else None
        n = lib().orc_mi_run_exact(self.h, _p(cand), len(cand), _p(start), len(start), int(subset), _p(S), _p(G),
                                   _p(fi), _p(t_sc), _p(t_am))
        res = dict(S=S[:n].copy(), GAIN=G[:n].copy(), iters=int(n))
        if trace:
            res.update(scores=t_sc[:n], argmax=t_am[:n].copy())
        return res


def topk_desc(scores, k):
    s = np.ascontiguousarray(scores, np.float64)
    pos = np.empty(k, np.int32)
    lib().orc_topk_desc(_p(s), len(s), k, _p(pos))
    return pos


def num_threads():
    return int(lib().orc_num_threads())


def set_threads(n):
    lib().orc_set_threads(int(n))
