"""torch's default CPU generator, as consumed by the reference on this path.

The reference draws torch.rand(k,d)*1e-5 (sgd_clustering.py:24), torch.rand(k,b)
(sgd_clustering.py:68) and torch.randperm(L) (measures/batch.py:31) from torch's global CPU
mt19937.  Generator reproduces that stream bit for bit through the C ABI (acav_rng_*), so a run
seeded with manual_seed(s) here equals a reference run seeded with torch.manual_seed(s).
An unseeded torch process starts from default seed 67280421310721; so does default_generator.
"""
import ctypes as C

import numpy as np

from . import _lib

TORCH_DEFAULT_SEED = 67280421310721


class Generator:
    def __init__(self, seed=TORCH_DEFAULT_SEED):
        lib = _lib.load_library()
        h = C.c_void_p()
        _lib.check(lib.acav_rng_create(C.byref(h), int(seed) & 0xFFFFFFFF))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and _lib._lib is not None:  # module globals die first at interpreter exit
            _lib._lib.acav_rng_destroy(h)

    @property
    def handle(self):
        return self._h

    def manual_seed(self, seed):
        """torch.manual_seed: mt19937 is initialised from the low 32 bits of the seed."""
        _lib.check(_lib._lib.acav_rng_seed(self._h, int(seed) & 0xFFFFFFFF))
        return self

    def u32(self):
        v = C.c_uint32(0)
        _lib.check(_lib._lib.acav_rng_u32(self._h, C.byref(v)))
        return v.value

    def jump(self, n):
        """skip n draws (GF(2) jump-ahead): same state as n calls of u32()"""
        _lib.check(_lib._lib.acav_rng_jump(self._h, int(n)))
        return self

    def rand(self, *shape):
        """torch.rand(*shape) (float32, CPU) as a numpy array."""
        out = np.empty(shape, np.float32)
        _lib.check(_lib._lib.acav_rng_rand_f32(self._h, _lib.ptr(out), out.size))
        return out

    def randperm(self, n):
        out = np.empty(int(n), np.int64)
        _lib.check(_lib._lib.acav_rng_randperm(self._h, int(n), _lib.ptr(out)))
        return out

    def warmup_best(self, k, b):
        """argmin over k of torch.rand(k, b) per column + mean of the minima (sgd_clustering.py:67-68,78-79)."""
        best = np.empty(int(b), np.int64)
        mean = C.c_float(0)
        _lib.check(_lib._lib.acav_rng_warmup_best(self._h, int(k), int(b), _lib.ptr(best), C.byref(mean)))
        return best, mean.value

    def get_state(self):
        mt = np.empty(624, np.uint32)
        idx = C.c_int(0)
        _lib.check(_lib._lib.acav_rng_get_state(self._h, _lib.ptr(mt), C.byref(idx)))
        return mt, idx.value

    def set_state(self, mt, idx):
        mt = np.ascontiguousarray(mt, np.uint32)
        _lib.check(_lib._lib.acav_rng_set_state(self._h, _lib.ptr(mt), int(idx)))


class _Lazy:
    """default_generator is created on first use so importing the package never needs the .so."""
    _gen = None

    def _get(self):
        if _Lazy._gen is None:
            _Lazy._gen = Generator()
        return _Lazy._gen

    def __getattr__(self, name):
        return getattr(self._get(), name)


default_generator = _Lazy()


def manual_seed(seed):
    default_generator.manual_seed(seed)


def python_shuffled_range(n):
    """`order = list(range(n)); random.shuffle(order)` -- same result, same effect on Python's global generator -- as
    an int64 array, computed in the library (the interpreter needs 0.6 s per million elements)."""
    import random
    version, internal, gauss = random.getstate()
    gen = Generator(0)
    gen.set_state(np.array(internal[:624], dtype=np.uint32), int(internal[624]))
    out = np.empty(int(n), np.int64)
    _lib.check(_lib._lib.acav_rng_py_shuffle(gen.handle, int(n), _lib.ptr(out)))
    mt, idx = gen.get_state()
    random.setstate((version, tuple(int(v) for v in mt) + (int(idx),), gauss))
    return out
