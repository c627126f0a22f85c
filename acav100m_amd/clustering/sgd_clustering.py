"""KMeans -- drop-in for the reference's clustering/code/sgd_clustering.py:10-129.

Same constructor, methods and public attributes (centers, counts, count, lr, fallback,
initial_rounds, reinit, sequential, args), so run_clustering.init_clusterings
(run_clustering.py:32-52), process_batch._train_batch / _extract_batch (process_batch.py:6-17,
37-56) and the checkpoint code (run_clustering.py:87-116) work unchanged.  All arithmetic runs in
libacav_hip.so on the MI355X; torch tensors are containers only.  There is no CPU path: the
object must be moved to a GPU with .to('cuda') before calc_best/add.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..rng import default_generator


def _as_f32_2d(batch, width=None):
    """-> (object keeping the memory alive, void*, rows, is_torch_cuda); width: the d the library will stride by"""
    shape = tuple(batch.shape) if hasattr(batch, "shape") else np.shape(batch)
    if len(shape) != 2 or (width is not None and shape[1] != width):
        raise ValueError("expected a [rows, {}] feature matrix, got shape {}".format(width if width is not None else "d", shape))
    if hasattr(batch, "data_ptr"):  # torch tensor (cpu or cuda)
        import torch
        t = batch.detach()
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(torch.float32).contiguous()
        if t.is_cuda:
            # the library runs on its own stream: whatever torch still has queued for this tensor must land first
            torch.cuda.current_stream(t.device).synchronize()
        return t, C.c_void_p(t.data_ptr()), t.shape[0], t.is_cuda
    a = np.ascontiguousarray(batch, dtype=np.float32)
    return a, a.ctypes.data_as(C.c_void_p), a.shape[0], False


def _device_index(device):
    s = str(device)
    if s == "cpu":
        raise _lib.AcavError("acav100m_amd.KMeans has no CPU path: move it to a GPU (device='cuda')")
    if not s.startswith("cuda"):
        raise ValueError(f"unsupported device {device!r}")
    if ":" in s:
        return int(s.split(":")[1])
    try:
        import torch
        return torch.cuda.current_device()
    except Exception:
        return 0


class KMeans:
    """KMeans by Gradient Descent (reference docstring: sgd_clustering.py:11-17)."""

    def __init__(self, args=None, d=None, k=None, lr=1e-2, initial_rounds=10, reinit=(.7, 5.0),
                 saved_dt=None, generator=None):
        self._h = None
        self._device = None
        self._generator = generator if generator is not None else default_generator
        if saved_dt is not None:
            self.load_from_saves(saved_dt)
        else:
            self.args = args
            # torch.rand(k, d) * 1e-5 from the CPU generator (sgd_clustering.py:24)
            self._centers0 = self._generator.rand(k, d) * np.float32(1e-5)
            self._counts0 = np.zeros(k, np.float32)
            self._count0 = 0
            self._fallback0 = 0
            self.lr = lr
            self.initial_rounds = initial_rounds
            self.reinit = reinit
            self.sequential = False

    # ------------------------------------------------------------------ handle management
    def __del__(self):
        self._release()

    def _release(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and _lib._lib is not None:  # module globals die first at interpreter exit
            _lib._lib.acav_kmeans_destroy(h)

    def _require_handle(self):
        if self._h is None:
            raise _lib.AcavError("KMeans is not on a GPU yet: call .to('cuda') (there is no CPU fallback)")
        return self._h

    def to(self, device):
        """sgd_clustering.py:59-61.  Creates the device-side state on `device` (cuda only)."""
        idx = _device_index(device)
        centers, counts, count, fallback = self._host_state()
        self._release()
        lib = _lib.load_library()
        h = C.c_void_p()
        k, d = centers.shape
        _lib.check(lib.acav_kmeans_create(C.byref(h), idx, k, d, _lib.ptr(centers), None))
        self._h, self._device = h, idx
        p, r = self.reinit
        _lib.check(lib.acav_kmeans_set_hyper(h, int(self.initial_rounds), float(p), float(r)))
        _lib.check(lib.acav_kmeans_set_state(h, None, _lib.ptr(counts), int(count), int(fallback)))
        return self

    def _host_state(self):
        if self._h is None:
            return self._centers0, self._counts0, self._count0, self._fallback0
        k, d = self._shape
        centers = np.empty((k, d), np.float32)
        counts = np.empty(k, np.float32)
        count, fb = C.c_int64(0), C.c_int64(0)
        _lib.check(_lib._lib.acav_kmeans_get_state(self._h, _lib.ptr(centers), _lib.ptr(counts), C.byref(count),
                                                   C.byref(fb)))
        return centers, counts, count.value, fb.value

    def _scalars(self):
        """(count, fallback) without moving the centres: `count` is host state of the handle, `fallback` one
        16-byte read -- add() / calc_best() / warmup_steps() look at them every step"""
        if self._h is None:
            return self._count0, self._fallback0
        count, fb = C.c_int64(0), C.c_int64(0)
        _lib.check(_lib._lib.acav_kmeans_get_state(self._h, None, None, C.byref(count), C.byref(fb)))
        return count.value, fb.value

    @property
    def _shape(self):
        return self._centers0.shape

    # ------------------------------------------------------------- reference attributes
    @property
    def centers(self):
        import torch
        return torch.from_numpy(self._host_state()[0])

    @centers.setter
    def centers(self, value):
        v = np.ascontiguousarray(value.detach().cpu().numpy() if hasattr(value, "detach") else value, np.float32)
        if self._h is None:
            self._centers0 = v
        else:
            c = self._scalars()
            _lib.check(_lib._lib.acav_kmeans_set_state(self._h, _lib.ptr(v), None, c[0], c[1]))

    @property
    def counts(self):
        import torch
        return torch.from_numpy(self._host_state()[1])

    @counts.setter
    def counts(self, value):
        v = np.ascontiguousarray(value.detach().cpu().numpy() if hasattr(value, "detach") else value, np.float32)
        if self._h is None:
            self._counts0 = v
        else:
            c = self._scalars()
            _lib.check(_lib._lib.acav_kmeans_set_state(self._h, None, _lib.ptr(v), c[0], c[1]))

    @property
    def count(self):
        if self._h is None:
            return self._count0
        count = C.c_int64(0)  # host-side integer of the handle: no stream synchronisation
        _lib.check(_lib._lib.acav_kmeans_get_state(self._h, None, None, C.byref(count), None))
        return count.value

    @count.setter
    def count(self, value):
        if self._h is None:
            self._count0 = int(value)
        else:
            _lib.check(_lib._lib.acav_kmeans_set_state(self._h, None, None, int(value), self._scalars()[1]))

    @property
    def fallback(self):
        return self._scalars()[1]

    @fallback.setter
    def fallback(self, value):
        if self._h is None:
            self._fallback0 = int(value)
        else:
            _lib.check(_lib._lib.acav_kmeans_set_state(self._h, None, None, self._scalars()[0], int(value)))

    # ------------------------------------------------------------------- save / load
    def get_attrs(self):
        """sgd_clustering.py:34-46"""
        centers, counts, count, fallback = self._host_state()
        return {
            'args': self.args,
            'count': count,
            'lr': self.lr,
            'initial_rounds': self.initial_rounds,
            'reinit': self.reinit,
            'fallback': fallback,
            'sequential': self.sequential,
            'centers': centers,
            'counts': counts,
        }

    def load_from_saves(self, dt):
        """sgd_clustering.py:48-52"""
        self.args = dt.get('args')
        self.lr = dt.get('lr', 1e-2)
        self.initial_rounds = dt.get('initial_rounds', 10)
        self.reinit = tuple(dt.get('reinit', (.7, 5.0)))
        self.sequential = dt.get('sequential', False)
        self._centers0 = np.ascontiguousarray(dt['centers'], np.float32)
        self._counts0 = np.ascontiguousarray(dt['counts'], np.float32)
        self._count0 = int(dt.get('count', 0))
        self._fallback0 = int(dt.get('fallback', 0))

    @classmethod
    def load(cls, dt):
        return cls(saved_dt=dt)

    def __getstate__(self):  # torch.save(dict of KMeans) -- run_clustering.py:110-116
        st = self.get_attrs()
        st['_device'] = self._device
        return st

    def __setstate__(self, st):
        self._h = None
        self._device = None
        self._generator = default_generator
        self.load_from_saves(st)
        if st.get('_device') is not None and _lib.device_count() > 0:
            self.to(f"cuda:{st['_device']}")

    # ---------------------------------------------------------------------- compute
    def calc_best(self, batch, need_mean=True):
        """sgd_clustering.py:63-79 -> (best LongTensor[b], mean of the minima).
        need_mean=False (bulk assign): the library may take its HBM-bound half-precision-filter + exact-re-check path
        (bit-identical labels); the second return value is then None."""
        import torch
        h = self._require_handle()
        k = self._shape[0]
        keep, xp, b, on_gpu = _as_f32_2d(batch, self._shape[1])
        if self.count < self.initial_rounds * k:
            best, mean = self._generator.warmup_best(k, b)
            out = torch.from_numpy(best)
            return (out.cuda(self._device) if on_gpu else out), mean
        labels = torch.empty(b, dtype=torch.long, device=(batch.device if on_gpu else "cpu"))
        if not need_mean:
            _lib.check(_lib._lib.acav_kmeans_assign(h, xp, b, _lib.ptr(labels), None))
            _lib.check(_lib._lib.acav_kmeans_sync(h))
            return labels, None
        mean = C.c_float(0)
        _lib.check(_lib._lib.acav_kmeans_assign(h, xp, b, _lib.ptr(labels), C.byref(mean)))
        return labels, mean.value

    def train_stats(self):
        """(bulk training calls that ran as one persistent launch, launches that gave up and were re-run per step)"""
        a, b = C.c_int64(0), C.c_int64(0)
        _lib.check(_lib._lib.acav_kmeans_train_stats(self._require_handle(), C.byref(a), C.byref(b)))
        return a.value, b.value

    def filter_stats(self):
        """(filter launches, rows of the last one, rows that needed the exact re-check)"""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _lib.check(_lib._lib.acav_kmeans_filter_stats(self._require_handle(), C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def recheck_stats(self):
        """(rows settled by their candidate centres, (row, centre) pairs evaluated for them, rows that took the full exact sweep)
        of the last filter sweep"""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _lib.check(_lib._lib.acav_kmeans_recheck_stats(self._require_handle(), C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    @property
    def is_distributed(self):
        """sgd_clustering.py:81-86"""
        return (
            self.args is not None and
            self.args.computation.device == 'cuda' and
            self.args.computation.num_gpus > 1
        )

    def initialize(self):
        """sgd_clustering.py:88-92: average centers/counts over ranks so every rank starts equal."""
        if self.is_distributed:
            import torch
            from ..parallel import average_state
            centers, counts, count, fb = self._host_state()
            dev = f"cuda:{self._device}"
            c, n = average_state(torch.from_numpy(centers).to(dev), torch.from_numpy(counts).to(dev))
            _lib.check(_lib._lib.acav_kmeans_set_state(self._h, _lib.ptr(c.contiguous()), _lib.ptr(n.contiguous()),
                                                       count, fb))

    def add(self, batch):
        """sgd_clustering.py:94-129 (fast parallel update).  Returns the mean min-distance."""
        h = self._require_handle()
        if self.sequential:
            raise NotImplementedError("the reference's slow `sequential` branch is never enabled (sequential=False)")
        k = self._shape[0]
        lr = self.lr(self.count) if callable(self.lr) else self.lr
        keep, xp, b, on_gpu = _as_f32_2d(batch, self._shape[1])
        if self.is_distributed:
            return self._add_distributed(batch, lr)
        mean = C.c_float(0)
        if self.count < self.initial_rounds * k:
            best, m = self._generator.warmup_best(k, b)
            _lib.check(_lib._lib.acav_kmeans_step(h, xp, b, float(lr), _lib.ptr(best), None))
            return m
        _lib.check(_lib._lib.acav_kmeans_step(h, xp, b, float(lr), None, C.byref(mean)))
        return mean.value

    def _add_distributed(self, batch, lr):
        """Multi-GPU add(): see acav100m_amd/parallel/kmeans_dp.py."""
        from ..parallel import distributed_add
        return distributed_add(self, batch, lr)

    def apply_update(self, x, best, lr):
        """The update half of add() on an already-labelled global batch (sgd_clustering.py:113-128)."""
        keep, xp, b, _ = _as_f32_2d(x, self._shape[1])
        if hasattr(best, "data_ptr"):
            import torch
            lab = best.detach().to(torch.long).contiguous()
        else:
            lab = np.ascontiguousarray(best, np.int64)
        _lib.check(_lib._lib.acav_kmeans_apply_update(self._require_handle(), xp, b, _lib.ptr(lab), float(lr)))

    # ----------------------------------------------------------- bulk (device-resident) API
    def warmup_steps(self, batch_size, steps):
        """number of the next `steps` add() calls that still fall in the warm-up (count < initial_rounds*k)"""
        lim, cnt = self.initial_rounds * self._shape[0], self.count
        return 0 if cnt >= lim else min(int(steps), -(-(lim - cnt) // int(batch_size)))

    def draw_warmup(self, batch_size):
        """labels of one warm-up step: argmin_k torch.rand(k, b) (sgd_clustering.py:67-68,78)"""
        return self._generator.warmup_best(self._shape[0], int(batch_size))[0]

    def train_epoch(self, x, batch_size, lr=None, warm_best=None):
        """The train-loop body of run_clustering.py:229-241 for this clustering over a resident
        feature matrix x [n,d]: floor(n/b) add() steps with no host sync in between.
        warm_best [need,b]: pre-drawn warm-up labels (when several clusterings share the RNG stream
        and must consume it batch by batch); drawn here otherwise."""
        h = self._require_handle()
        lr = self.lr if lr is None else lr
        keep, xp, n, _ = _as_f32_2d(x, self._shape[1])
        need = self.warmup_steps(batch_size, n // batch_size)
        if warm_best is None:
            warm = np.empty((need, batch_size), np.int64)
            for t in range(need):
                warm[t] = self.draw_warmup(batch_size)
        else:
            warm = np.ascontiguousarray(warm_best, np.int64)
            assert warm.shape == (need, batch_size), (warm.shape, need, batch_size)
        _lib.check(_lib._lib.acav_kmeans_train(h, xp, n, int(batch_size), float(lr),
                                               _lib.ptr(warm) if need else None, need))

    @staticmethod
    def train_epoch_multi(clusterings, xs, batch_size, lr=None, warm_bests=None):
        """train_epoch for several clusterings of one device at once (acav_kmeans_train_multi): the reference steps
        every clustering per batch (run_clustering.py:229-241); their SGD chains are independent, so the persistent
        kernels run side by side -- same results as one train_epoch per clustering.  warm_bests[i] as in train_epoch
        (None: drawn here, clustering by clustering -- pass them when the clusterings share a generator and the
        reference's batch-by-batch interleaving of the draws matters)."""
        kms = list(clusterings)
        if not kms:
            return
        keep, ptrs, ns, warms, needs = [], [], [], [], []
        for i, (km, x) in enumerate(zip(kms, xs)):
            km._require_handle()
            k, xp, n, _ = _as_f32_2d(x, km._shape[1])
            need = km.warmup_steps(batch_size, n // batch_size)
            if warm_bests is None or warm_bests[i] is None:
                w = np.empty((need, batch_size), np.int64)
                for t in range(need):
                    w[t] = km.draw_warmup(batch_size)
            else:
                w = np.ascontiguousarray(warm_bests[i], np.int64)
                assert w.shape == (need, batch_size), (w.shape, need, batch_size)
            keep.append((k, w))
            ptrs.append(xp.value)
            ns.append(n)
            warms.append(w.ctypes.data if need else None)
            needs.append(need)
        cnt = len(kms)
        h_arr = (C.c_void_p * cnt)(*[km._h.value for km in kms])
        x_arr = (C.c_void_p * cnt)(*ptrs)
        w_arr = (C.c_void_p * cnt)(*warms)
        n_arr = np.asarray(ns, np.int64)
        nw_arr = np.asarray(needs, np.int64)
        lr = kms[0].lr if lr is None else lr
        _lib.check(_lib._lib.acav_kmeans_train_multi(h_arr, cnt, x_arr, _lib.ptr(n_arr), int(batch_size), float(lr), w_arr,
                                                     _lib.ptr(nw_arr)))

    def train_epoch_distributed(self, x_local, batch_size, lr=None, chunk_steps=1024, train_here=True, comm_slot=0, wait=True,
                                trainer=None):
        """One epoch in the large-batch mode (global batch = world * batch_size rows, rank-major: every rank feeds
        batch_size of its own rows per step -- not the reference's N-GPU run, see train_epoch_plan_multi) without a
        collective per step: acav100m_amd/parallel/kmeans_dp.py:train_epoch_dp."""
        lr = self.lr if lr is None else lr
        from ..parallel.rccl_comm import default_comm
        comm = default_comm(comm_slot) if hasattr(x_local, "is_cuda") and x_local.is_cuda else None
        if comm is None:  # gloo / host tensors (CPU tests): the same schedule through torch.distributed
            from ..parallel import train_epoch_dp
            return train_epoch_dp(self, x_local, int(batch_size), lr, chunk_steps=chunk_steps, trainer=trainer)
        self.train_epoch_comm(comm, x_local, int(batch_size), lr, chunk_steps, train_here, wait, trainer)

    @staticmethod
    def train_epoch_distributed_multi(clusterings, xs_local, batch_size, lr=None, chunk_steps=1024, trainers=None):
        """train_epoch_distributed for several clusterings over the same local rows, chunk by chunk across all of them
        (acav_kmeans_train_dp_multi): every rank feeds every clustering's row exchange (communicator slot = position in
        the list), the replicated chain of clustering v runs on rank trainers[v] only (default v % world) -- different
        ranks train different clusterings at the same time.  Follow with broadcast_state_from(trainers[v], comm_slot=v).
        Without RCCL (gloo / host tensors: CPU tests) the clusterings go one after the other through torch.distributed."""
        kms = list(clusterings)
        if not kms:
            return []
        from ..parallel.collectives import world as _world
        from ..parallel.rccl_comm import default_comm
        rank, w = _world()
        trainers = [v % w for v in range(len(kms))] if trainers is None else [int(t) for t in trainers]
        lr = kms[0].lr if lr is None else lr
        on_gpu = all(hasattr(x, "is_cuda") and x.is_cuda for x in xs_local)
        comms = [default_comm(v) for v in range(len(kms))] if on_gpu else [None] * len(kms)
        if any(c is None for c in comms):
            from ..parallel import train_epoch_dp
            for v, (km, x) in enumerate(zip(kms, xs_local)):
                train_epoch_dp(km, x, int(batch_size), lr, chunk_steps=chunk_steps, trainer=trainers[v])
            return trainers
        import torch
        keep, ptrs, warms, needs = [], [], [], []
        n_local = None
        for v, (km, x, comm) in enumerate(zip(kms, xs_local, comms)):
            k, xp, n, _ = _as_f32_2d(x, km._shape[1])
            assert n_local is None or n == n_local, "the clusterings of one call share the local rows"
            n_local = n
            need = km.warmup_steps(w * int(batch_size), n // int(batch_size))
            warm = None
            if need:  # drawn per rank (every rank: the shared generator must advance alike), exchanged once
                mine = np.stack([km.draw_warmup(int(batch_size)) for _ in range(need)]).astype(np.int64)
                if w > 1:
                    send = torch.from_numpy(mine).to(x.device)
                    recv = torch.empty((w, need, int(batch_size)), dtype=torch.long, device=x.device)
                    comm.allgather(send, recv)
                    comm.synchronize()
                    warm = np.ascontiguousarray(recv.cpu().numpy().transpose(1, 0, 2).reshape(need, w * int(batch_size)))
                else:
                    warm = np.ascontiguousarray(mine)
            keep.append((k, warm))
            ptrs.append(xp.value)
            warms.append(warm.ctypes.data if need else None)
            needs.append(need)
        cnt = len(kms)
        h_arr = (C.c_void_p * cnt)(*[km._require_handle().value for km in kms])
        c_arr = (C.c_void_p * cnt)(*[c._h.value for c in comms])
        x_arr = (C.c_void_p * cnt)(*ptrs)
        w_arr = (C.c_void_p * cnt)(*warms)
        nw_arr = np.asarray(needs, np.int64)
        # ACAV_DP_TRAIN | ACAV_DP_ROOTED | root << 8: the rows of clustering v are sent to rank trainers[v] only
        here = np.asarray([(1 if t == rank else 0) | 4 | (t << 8) for t in trainers], np.int32)
        _lib.check(_lib._lib.acav_kmeans_train_dp_multi(h_arr, c_arr, cnt, x_arr, int(n_local), int(batch_size), float(lr), w_arr,
                                                        _lib.ptr(nw_arr), int(chunk_steps), _lib.ptr(here)))
        return trainers

    @staticmethod
    def train_epoch_plan_multi(clusterings, xs_local, plan, lr=None, chunk_steps=1024, trainers=None, warm_bests=None):
        """One epoch of several clusterings over the global batches of `plan` (parallel/row_plan.py: `reference` = the
        reference's own N-GPU batch stream, `views` = the one-GPU stream over partitioned rows, `rows` = large batch) with
        the rows living on the ranks that own them: acav_kmeans_train_plan_multi.  Every rank feeds every clustering's row
        exchange (communicator slot = position in the list); the chain of clustering v runs on rank trainers[v] only
        (default v % world): different ranks train different clusterings at the same time.  Follow with
        broadcast_state_from(trainers[v], comm_slot=v).  warm_bests[v]: [need, global batch] warm-up labels (None: drawn
        and exchanged here, clustering by clustering -- parallel.plan_warmup_labels).  Without RCCL (gloo / host tensors:
        CPU tests, several ranks on one GPU) the clusterings go one after the other through torch.distributed
        (parallel.train_epoch_plan)."""
        kms = list(clusterings)
        if not kms:
            return []
        from ..parallel.collectives import world as _world
        from ..parallel.kmeans_dp import plan_warmup_labels, train_epoch_plan
        from ..parallel.rccl_comm import default_comm
        rank, w = _world()
        trainers = [v % w for v in range(len(kms))] if trainers is None else [int(t) for t in trainers]
        lr = kms[0].lr if lr is None else lr
        on_gpu = all(hasattr(x, "is_cuda") and x.is_cuda for x in xs_local)
        comms = [default_comm(v) for v in range(len(kms))] if on_gpu else [None] * len(kms)
        warms = []
        for v, (km, x) in enumerate(zip(kms, xs_local)):  # drawn in clustering order on every rank: the shared generator advances alike
            wb = None if warm_bests is None else warm_bests[v]
            if wb is None:
                wb = plan_warmup_labels(km, plan, device=getattr(x, "device", None), comm=comms[v] if w > 1 else None)
            warms.append(None if wb is None else np.ascontiguousarray(wb, np.int64))
        if any(c is None for c in comms):
            for v, (km, x) in enumerate(zip(kms, xs_local)):
                train_epoch_plan(km, x, plan, lr, trainers[v], chunk_steps=chunk_steps, warm=warms[v])
            return trainers
        keep, ptrs = [], []
        n_local = None
        for km, x in zip(kms, xs_local):
            k, xp, n, _ = _as_f32_2d(x, km._shape[1])
            assert n_local is None or n == n_local, "the clusterings of one call share the local rows"
            n_local = n
            keep.append(k)
            ptrs.append(xp.value)
        cnt = len(kms)
        ext = np.ascontiguousarray(plan.table(), np.int64)
        needs = np.asarray([0 if wm is None else len(wm) for wm in warms], np.int64)
        for v, wm in enumerate(warms):
            assert wm is None or wm.shape == (needs[v], plan.global_batch), (wm.shape, needs[v], plan.global_batch)
        h_arr = (C.c_void_p * cnt)(*[km._require_handle().value for km in kms])
        c_arr = (C.c_void_p * cnt)(*[c._h.value for c in comms])
        x_arr = (C.c_void_p * cnt)(*ptrs)
        w_arr = (C.c_void_p * cnt)(*[None if wm is None else wm.ctypes.data for wm in warms])
        here = np.asarray([(1 if t == rank else 0) | 4 | (t << 8) for t in trainers], np.int32)  # ACAV_DP_TRAIN | ACAV_DP_ROOTED | root << 8
        _lib.check(_lib._lib.acav_kmeans_train_plan_multi(h_arr, c_arr, cnt, x_arr, int(n_local), int(plan.slots), int(plan.lb),
                                                          _lib.ptr(ext), int(ext.shape[0]), int(plan.steps), float(lr), w_arr,
                                                          _lib.ptr(needs), int(chunk_steps), _lib.ptr(here)))
        for v, km in enumerate(kms):  # a rank that only fed the exchange keeps `count` in step until the state arrives
            if trainers[v] != rank:
                km.skip_epoch(plan.steps * plan.global_batch)
        return trainers

    def broadcast_state_from(self, root, comm_slot=0):
        """every rank's state <- rank `root`'s (RCCL through the C ABI; torch.distributed under gloo)"""
        from ..parallel.rccl_comm import default_comm
        comm = default_comm(comm_slot)
        if comm is None:
            from ..parallel import broadcast_state
            return broadcast_state(self, int(root))
        _lib.check(_lib._lib.acav_kmeans_broadcast_state(self._require_handle(), comm._h, int(root)))

    def train_epoch_comm(self, comm, x_local, b_local, lr, chunk_steps=1024, train_here=True, wait=True, trainer=None):
        """acav_kmeans_train_dp: the DDP epoch with the bulk row exchange through RCCL inside the library -- no torch
        op between the collective and the SGD chain.  Only the warm-up labels (drawn per rank) are exchanged here, once.
        trainer: the ONE rank that runs the chain (every rank passes the same number, train_here == (rank == trainer)):
        the rows are sent to it instead of all-gathered."""
        import torch
        keep, xp, n_local, on_gpu = _as_f32_2d(x_local, self._shape[1])
        w = comm.world
        steps = n_local // b_local
        need = self.warmup_steps(w * b_local, steps)
        warm = None
        if need:
            mine = np.stack([self.draw_warmup(b_local) for _ in range(need)]).astype(np.int64)  # [need, b_local]
            if w > 1:
                dev = x_local.device
                send = torch.from_numpy(mine).to(dev)
                recv = torch.empty((w, need, b_local), dtype=torch.long, device=dev)
                comm.allgather(send, recv)
                comm.synchronize()
                warm = np.ascontiguousarray(recv.cpu().numpy().transpose(1, 0, 2).reshape(need, w * b_local))
            else:
                warm = np.ascontiguousarray(mine)
        _lib.check(_lib._lib.acav_kmeans_train_dp(self._require_handle(), comm._h, xp, n_local, int(b_local), float(lr),
                                                  _lib.ptr(warm) if need else None, need, int(chunk_steps),
                                                  (1 if train_here else 0) | (0 if wait else 2) |
                                                  (0 if trainer is None else 4 | (int(trainer) << 8))))  # ACAV_DP_ROOTED

    # ------------------------------------------------- state exchange (parallel/kmeans_dp.py:broadcast_state)
    def state_arrays(self):
        return self._host_state()

    def load_state_arrays(self, centers, counts, count, fallback):
        c = np.ascontiguousarray(centers, np.float32)
        n = np.ascontiguousarray(counts, np.float32)
        if self._h is None:
            self._centers0, self._counts0, self._count0, self._fallback0 = c, n, int(count), int(fallback)
        else:
            _lib.check(_lib._lib.acav_kmeans_set_state(self._h, _lib.ptr(c), _lib.ptr(n), int(count), int(fallback)))

    def skip_epoch(self, rows):
        """another rank trains this clustering over `rows` rows; its state arrives by broadcast at the end of the epoch.
        `count` advances here all the same: the warm-up plan of the next row group (and with it the number of draws from
        the shared generator) must be the owner's"""
        self.count = self.count + int(rows)

    def get_attrs_plain(self):
        """get_attrs() without the args object (what the checkpoint files hold)"""
        dt = self.get_attrs()
        dt['args'] = None
        return dt

    def synchronize(self):
        _lib.check(_lib._lib.acav_kmeans_sync(self._require_handle()))
