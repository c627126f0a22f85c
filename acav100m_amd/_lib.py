"""ctypes binding of libacav_hip.so (the C ABI declared in include/acav_hip.h).

The product path has no CPU fallback: if the library is missing, or no HIP device is visible
when a handle is created, the caller gets a loud RuntimeError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ACAV_LIB_PATH: another build of the same library (experiment builds with -D knobs, tools/exp/) -- still the HIP path
LIB_PATH = os.environ.get("ACAV_LIB_PATH") or os.path.join(_HERE, "libacav_hip.so")

vp, i64, i32, f32, f64, u32 = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_double, C.c_uint32
pp = C.POINTER(C.c_void_p)

# name -> argtypes (every function returns int except acav_last_error); mirrors include/acav_hip.h
SIGNATURES = {
    "acav_version": [],
    "acav_device_count": [C.POINTER(i32)],
    "acav_device_info": [i32, C.c_char_p, i32, C.POINTER(i32), C.POINTER(i64)],
    "acav_trim_device_cache": [C.POINTER(i64)],
    "acav_rng_create": [pp, u32],
    "acav_rng_destroy": [vp],
    "acav_rng_seed": [vp, u32],
    "acav_rng_u32": [vp, C.POINTER(u32)],
    "acav_rng_jump": [vp, i64],
    "acav_rng_py_shuffle": [vp, i64, vp],
    "acav_rng_rand_f32": [vp, vp, i64],
    "acav_rng_randperm": [vp, i64, vp],
    "acav_rng_get_state": [vp, vp, C.POINTER(i32)],
    "acav_rng_set_state": [vp, vp, i32],
    "acav_rng_warmup_best": [vp, i32, i64, vp, C.POINTER(f32)],
    "acav_kmeans_create": [pp, i32, i32, i32, vp, vp],
    "acav_kmeans_destroy": [vp],
    "acav_kmeans_get_state": [vp, vp, vp, C.POINTER(i64), C.POINTER(i64)],
    "acav_kmeans_set_state": [vp, vp, vp, i64, i64],
    "acav_kmeans_set_hyper": [vp, i32, f64, f64],
    "acav_kmeans_assign": [vp, vp, i64, vp, C.POINTER(f32)],
    "acav_kmeans_step": [vp, vp, i64, f64, vp, C.POINTER(f32)],
    "acav_kmeans_train": [vp, vp, i64, i64, f64, vp, i64],
    "acav_kmeans_train_multi": [vp, i32, vp, vp, i64, f64, vp, vp],
    "acav_kmeans_apply_update": [vp, vp, i64, vp, f64],
    "acav_kmeans_sync": [vp],
    "acav_kmeans_shape": [vp, C.POINTER(i32), C.POINTER(i32)],
    "acav_kmeans_stream": [vp, pp],
    "acav_comm_unique_id": [vp],
    "acav_comm_init": [pp, i32, i32, i32, vp, vp],
    "acav_comm_destroy": [vp],
    "acav_comm_info": [vp, C.POINTER(i32), C.POINTER(i32)],
    "acav_comm_sync": [vp],
    "acav_comm_allreduce_f32": [vp, vp, i64],
    "acav_comm_allgather": [vp, vp, vp, i64],
    "acav_comm_broadcast": [vp, vp, i64, i32],
    "acav_kmeans_allreduce_init": [vp, vp],
    "acav_kmeans_train_dp": [vp, vp, vp, i64, i64, f64, vp, i64, i64, i32],
    "acav_kmeans_train_dp_multi": [vp, vp, i32, vp, i64, i64, f64, vp, vp, i64, vp],
    "acav_kmeans_train_plan_multi": [vp, vp, i32, vp, i64, i32, i64, vp, i64, i64, f64, vp, vp, i64, vp],
    "acav_kmeans_broadcast_state": [vp, vp, i32],
    "acav_kmeans_timer_begin": [vp],
    "acav_kmeans_timer_end": [vp, C.POINTER(f32)],
    "acav_kmeans_stats": [vp, C.POINTER(i64), C.POINTER(i64)],
    "acav_kmeans_train_stats": [vp, C.POINTER(i64), C.POINTER(i64)],
    "acav_kmeans_filter_stats": [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)],
    "acav_kmeans_recheck_stats": [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)],
    "acav_kmeans_filter_time": [vp, C.POINTER(f32)],
    "acav_contrastive_create": [pp, i32, i32, i32, i32, vp, vp],
    "acav_contrastive_destroy": [vp],
    "acav_contrastive_get_params": [vp, vp, C.POINTER(i64)],
    "acav_contrastive_set_params": [vp, vp],
    "acav_contrastive_train": [vp, vp, vp, vp, i64, f64, vp, vp],
    "acav_contrastive_infer": [vp, vp, vp, i64, vp],
    "acav_contrastive_set_comm": [vp, vp],
    "acav_contrastive_backward": [vp, vp, vp, i64, vp, vp],
    "acav_contrastive_get_grads": [vp, vp],
    "acav_contrastive_set_grads": [vp, vp],
    "acav_contrastive_step": [vp, f64],
    "acav_mi_create": [pp, i32, vp, i64, i32, i32, vp, i32, vp],
    "acav_mi_destroy": [vp],
    "acav_mi_add_samples": [vp, vp, i64],
    "acav_mi_score_batch": [vp, vp, i32, vp],
    "acav_mi_run_greedy": [vp, vp, i64, vp, i32, i64, i32, i32, i32, vp, vp, vp, C.POINTER(i64),
                           C.POINTER(i64), vp, vp, vp, vp, i64],
    "acav_mi_run_greedy_multi": [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp],
    "acav_mi_run_exact": [vp, vp, i64, i32, i64, vp, vp, C.POINTER(i64), vp, vp, vp],
    "acav_mi_set_measure": [vp, i32],
    "acav_mi_get_counts": [vp, vp, vp, vp, C.POINTER(i64)],
    "acav_mi_sync": [vp],
    "acav_mi_timer_begin": [vp],
    "acav_mi_timer_end": [vp, C.POINTER(f32)],
    "acav_pkl_shard_open": [C.c_char_p, pp],
    "acav_pkl_shard_close": [vp],
    "acav_pkl_shard_info": [vp, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)],
    "acav_pkl_shard_view": [vp, i32, C.POINTER(i32), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                            C.POINTER(C.c_char_p), C.POINTER(i64)],
    "acav_pkl_shard_copy_view": [vp, i32, vp, i64],
    "acav_pkl_shard_meta": [vp, pp, C.POINTER(i64), pp, C.POINTER(i64), pp, pp, pp],
    "acav_pkl_load_group": [vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp],
    "acav_pkl_assign_load_group": [vp, i32, i32, vp, vp],
    "acav_pkl_shard_labels": [vp, pp],
}

_lib = None
# entry points that are necessarily the first device call of a handle's life (everything else needs a handle):
# calling one marks the HIP runtime as initialised for acav100m_amd.configure_runtime()
_FIRST_DEVICE_CALLS = ("acav_device_count", "acav_device_info", "acav_kmeans_create", "acav_mi_create",
                       "acav_contrastive_create", "acav_comm_init")
_device_touched = False


def device_touched():
    return _device_touched


def _marking(fn):
    def call(*args):
        global _device_touched
        _device_touched = True
        return fn(*args)
    call.argtypes, call.restype, call.__name__ = fn.argtypes, fn.restype, fn.__name__
    return call


class AcavError(RuntimeError):
    pass


def load_library():
    """Load libacav_hip.so.  torch (when installed) is imported first so that the process ends up
    with ONE HIP runtime: torch's bundled libamdhip64.so.7 has the same SONAME as ROCm's."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AcavError(
            f"{LIB_PATH} is missing: run `python __graft_entry__.py` (hipcc --offload-arch=gfx950) first. "
            "There is no CPU fallback.")
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.acav_last_error.restype = C.c_char_p
    lib.acav_last_error.argtypes = []
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library drift
        fn.restype = i32
        fn.argtypes = args
    for name in _FIRST_DEVICE_CALLS:
        setattr(lib, name, _marking(getattr(lib, name)))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = _lib.acav_last_error().decode("utf-8", "replace") if _lib is not None else ""
        if rc == -1:
            raise ValueError(f"acav: {msg}")
        if rc == -5:
            raise RuntimeError(f"acav: {msg}")
        if rc == -7:
            raise TimeoutError(f"acav: {msg}")
        raise AcavError(f"acav error {rc}: {msg}")


def device_count():
    n = i32(0)
    check(load_library().acav_device_count(C.byref(n)))
    return n.value


def ptr(obj):
    """void* of a numpy array / torch tensor (host or device) / None."""
    if obj is None:
        return None
    if hasattr(obj, "data_ptr"):
        return C.c_void_p(obj.data_ptr())
    return obj.ctypes.data_as(C.c_void_p)
