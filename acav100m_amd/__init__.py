"""acav100m_amd -- MI355X-native implementation of ACAV100M's curation hot path.

Only the hot path of SURVEY.md section 8 lives here: the SGD k-means of clustering/code and the greedy
batch-MI subset selection of subset_selection/code, behind the reference's own class
interfaces (KMeans, EfficientBatchMI / get_measure), computed by hand-written HIP kernels in
libacav_hip.so (C ABI: include/acav_hip.h).
"""
import os as _os
import sys as _sys
import warnings as _warnings

from ._lib import AcavError, LIB_PATH, device_count, load_library  # noqa: F401
from .rng import Generator, default_generator, manual_seed  # noqa: F401


def runtime_initialised():
    """True when some HIP runtime call already ran in this process as far as this package can tell: one of its own
    handles / device queries, or torch's lazy CUDA(=HIP) initialisation."""
    from . import _lib
    if _lib.device_touched():
        return True
    torch = _sys.modules.get("torch")
    try:
        return bool(torch is not None and torch.cuda.is_initialized())
    except Exception:
        return False


def configure_runtime(hw_queues=16, quiet=False):
    """Process-wide HIP runtime settings the hot path wants -- an EXPLICIT call (the CLIs, bench.py and the test suite
    make it; importing the package changes nothing in the host application's environment).

    `GPU_MAX_HW_QUEUES`: the reference trains every clustering of a batch stream per batch (ten in its real pipeline:
    5 + 5 layers); here each is a persistent launch on its own stream, and launches only overlap across HARDWARE queues.
    The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES queues (default 4): with ten clusterings the fifth
    launch waits for one of the first four to END (29.6 us per step of the ten instead of 14.9,
    tools/bench_train_real10.py).  The runtime reads the variable ONCE, when it initialises (the first device call of
    the process, torch's included): a call after that cannot take effect and says so.  An explicit setting of the
    caller wins.  Interaction: with more queues the lockstep MI chunks keep ONE generator stream per group
    (`ACAV_MI_SHARE_GEN`, default 1) -- one stream per chunk would spread ten generators over ten queues (7.4 vs 3.7 us
    per chunk-iteration).

    Returns the value in force for runtimes initialised from now on (str), or None when the runtime was already up."""
    if runtime_initialised():
        cur = _os.environ.get("GPU_MAX_HW_QUEUES")
        if not quiet and (cur is None or int(cur) < int(hw_queues)):
            _warnings.warn(
                "acav100m_amd.configure_runtime(): the HIP runtime of this process is already initialised "
                f"(GPU_MAX_HW_QUEUES={cur or 'default 4'}); clusterings beyond that many train in rounds, not side "
                "by side.  Call configure_runtime() before the first device call (torch.cuda included).",
                RuntimeWarning, stacklevel=2)
        return None
    return _os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(hw_queues)))



def trim_device_cache():
    """Return every parked device block of the library to the driver (`acav_trim_device_cache`, include/acav_hip.h); bytes freed."""
    import ctypes as _C
    from ._lib import check as _check, load_library as _load
    freed = _C.c_int64(0)
    _check(_load().acav_trim_device_cache(_C.byref(freed)))
    return int(freed.value)


__all__ = ["AcavError", "LIB_PATH", "device_count", "load_library", "Generator", "default_generator",
           "manual_seed", "configure_runtime", "runtime_initialised", "trim_device_cache"]
