"""ctypes binding of liblocrec.so (include/locrec.h).

There is deliberately no fallback: if the HIP library is missing, or no GPU is
usable, every operator raises -- the product path never routes through a CPU
implementation.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LOCREC_LIB_PATH: a development build (e.g. one made with DEBUG_SWITCHES=1) instead of the in-tree library
LIB_PATH = os.environ.get("LOCREC_LIB_PATH") or os.path.join(_HERE, "liblocrec.so")

OK, E_INVALID_ARG, E_NOT_FOUND, E_DEVICE, E_OOM, E_ARITHMETIC = 0, 1, 2, 3, 4, 5
MEM_HOST, MEM_DEVICE = 0, 1
KNN_BATCH_MAX_K = 1024


class IllegalArgumentException(ValueError):
    """What the reference throws for a failed require() or an unknown id
    (KnnRecommender.scala:17-20,83; StochasticRecommender.scala:33-34,70)."""


class LocrecRuntimeError(RuntimeError):
    """Device / allocation failures (the JNI shim maps these to RuntimeException)."""


_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)

# name -> (argtypes); every function returns int32 unless listed in _RESTYPE
SIGNATURES = {
    "locrec_last_error": [],
    "locrec_version": [],
    "locrec_device_count": [_i32p],
    "locrec_set_device": [C.c_int32],
    "locrec_device_allocations": [_i64p],
    "locrec_device_bytes_in_use": [_i64p],
    "locrec_cache_acquire": [C.c_int32, C.c_char_p, C.POINTER(C.c_void_p)],
    "locrec_cache_publish": [C.c_int32, C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)],
    "locrec_cache_release": [C.c_int32, C.c_void_p],
    "locrec_cache_set_limits": [C.c_int64, C.c_int64],
    "locrec_cache_clear": [],
    "locrec_cache_stats": [_i64p, _i64p, _i64p, _i64p, _i64p],
    "locrec_knn_create": [C.c_int64, _i64p, _i64p, _i32p, _f64p, C.c_int32, _i64p, _i32p, _f64p, C.c_int32,
                          _i64p, _i64p, _i64p, C.POINTER(C.c_void_p)],
    "locrec_knn_create_from_device": [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)],
    "locrec_knn_destroy": [C.c_void_p],
    "locrec_knn_info": [C.c_void_p, _i64p, _i64p, _i32p],
    "locrec_knn_vector_lengths": [C.c_void_p, _f64p, _f64p],
    "locrec_knn_cosine_similarity": [C.c_void_p, C.c_int64, C.c_int64, _f64p, _f64p],
    "locrec_knn_scan_plan": [C.c_void_p, _i32p, _i32p, _i32p, _i32p],
    "locrec_knn_batch_scan_bytes": [C.c_void_p, _i64p],
    "locrec_knn_ht_image_info": [C.c_void_p, _i64p, _i64p, _i64p],
    "locrec_knn_query": [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_int64, _i64p, _f64p, _i64p],
    "locrec_knn_recommend": [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_int64, _i64p, _f64p, _i64p],
    "locrec_knn_recommend_batch": [C.c_void_p, C.c_int64, _i64p, C.c_double, C.c_double, C.c_int64,
                                   _i64p, _i64p, _f64p, _i64p],
    "locrec_knn_recommend_range_async": [C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_int64],
    "locrec_knn_fetch_recommend": [C.c_void_p, C.c_int64, _i64p, _i64p, _f64p, _i64p],
    "locrec_knn_query_shard": [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_int64, C.c_int32, C.c_int32,
                               _i64p, _f64p, _i64p],
    "locrec_knn_recommend_neighbours": [C.c_void_p, C.c_int64, _i64p, _f64p, _i64p, _f64p, _i64p],
    "locrec_knn_query_batch": [C.c_void_p, C.c_int64, _i64p, C.c_double, C.c_double, C.c_int64, _i64p, _f64p, _i64p],
    "locrec_knn_all_pairs_topk": [C.c_void_p, C.c_double, C.c_double, C.c_int64, _i64p, _f64p, _i64p],
    "locrec_knn_row_person_ids": [C.c_void_p, C.c_int64, C.c_int64, _i64p],
    "locrec_knn_topk_range_async": [C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_int64],
    "locrec_knn_fetch_topk": [C.c_void_p, C.c_int64, C.c_int64, _i64p, _f64p, _i64p],
    "locrec_knn_replayed_intervals": [C.c_void_p, _i64p],
    "locrec_knn_set_stream": [C.c_void_p, C.c_void_p],
    "locrec_knn_synchronize": [C.c_void_p],
    "locrec_knn_profile_enable": [C.c_void_p, C.c_int32],
    "locrec_knn_profile_read": [C.c_void_p, _f64p, _i64p],
    "locrec_sg_create": [C.c_int64, _i64p, _i64p, _f64p, C.POINTER(C.c_void_p)],
    "locrec_sg_destroy": [C.c_void_p],
    "locrec_sg_info": [C.c_void_p, _i64p, _i64p, _i64p],
    "locrec_sg_device_bytes": [C.c_void_p, _i64p],
    "locrec_sg_weight_dictionary": [C.c_void_p, _i32p],
    "locrec_sg_recommend": [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_int64, _i64p, _f64p, _i64p, _i64p, _i32p],
    "locrec_sg_iterate_async": [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_int64],
    "locrec_sg_sweeps_async": [C.c_void_p, C.c_int64, C.c_double, C.c_int64],
    "locrec_sg_fetch": [C.c_void_p, _i64p, _f64p, _i64p, _i64p, _i32p],
    "locrec_sg_create_sharded": [C.c_int64, _i64p, _i64p, _f64p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)],
    "locrec_sg_create_target_sharded": [C.c_int64, _i64p, _i64p, _f64p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)],
    "locrec_sg_live_count": [C.c_void_p, _i64p],
    "locrec_sg_shard_begin": [C.c_void_p, C.c_int64],
    "locrec_sg_shard_sigma": [C.c_void_p, C.c_void_p],
    "locrec_sg_shard_apply": [C.c_void_p, C.c_void_p, C.c_double],
    "locrec_sg_shard_d2": [C.c_void_p, _f64p],
    "locrec_sg_shard_finish": [C.c_void_p, C.c_int64, C.c_int32],
    "locrec_sg_group_create": [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p)],
    "locrec_sg_group_destroy": [C.c_void_p],
    "locrec_sg_group_sweeps_async": [C.c_void_p, _i64p, C.c_double, C.c_int64],
    "locrec_sg_group_iterate_async": [C.c_void_p, _i64p, C.c_double, C.c_double, C.c_int64],
    "locrec_sg_group_synchronize": [C.c_void_p],
    "locrec_set_devices": [C.c_int32, _i32p],
    "locrec_knn_replicas_create": [C.c_int32, _i32p, C.c_int64, _i64p, _i64p, _i32p, _f64p, C.c_int32, _i64p, _i32p, _f64p, C.c_int32,
                                   _i64p, _i64p, _i64p, C.POINTER(C.c_void_p)],
    "locrec_knn_replicas_destroy": [C.c_void_p],
    "locrec_knn_replicas_info": [C.c_void_p, _i32p, C.POINTER(C.c_void_p)],
    "locrec_knn_replicas_recommend_batch": [C.c_void_p, C.c_int64, _i64p, C.c_double, C.c_double, C.c_int64, _i64p, _i64p, _f64p, _i64p],
    "locrec_knn_replicas_query_batch": [C.c_void_p, C.c_int64, _i64p, C.c_double, C.c_double, C.c_int64, _i64p, _f64p, _i64p],
    "locrec_sg_sharded_create": [C.c_int32, _i32p, C.c_int64, _i64p, _i64p, _f64p, C.c_int32, C.POINTER(C.c_void_p)],
    "locrec_sg_sharded_destroy": [C.c_void_p],
    "locrec_sg_sharded_info": [C.c_void_p, _i32p, _i32p, _i64p, _i64p],
    "locrec_sg_sharded_recommend": [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_int64, _i64p, _f64p, _i64p, _i64p, _i32p],
    "locrec_sg_sharded_iterate_async": [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_int64],
    "locrec_sg_sharded_sweeps_async": [C.c_void_p, C.c_int64, C.c_double, C.c_int64],
    "locrec_sg_sharded_fetch": [C.c_void_p, C.c_int32, _i64p, _f64p, _i64p, _i64p, _i32p],
    "locrec_sg_set_stream": [C.c_void_p, C.c_void_p],
    "locrec_sg_synchronize": [C.c_void_p],
    "locrec_sg_profile_enable": [C.c_void_p, C.c_int32],
    "locrec_sg_profile_read": [C.c_void_p, _f64p, _i64p],
    # the producers (prep.hip): array arguments are void* because they may be device pointers
    "locrec_calc_ratings": [C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, _i64p],
    "locrec_calc_rating_vectors": [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, _i64p, _i64p, _i64p],
    "locrec_build_balanced_edges": [C.c_int32, _f64p, _i64p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                    C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    "locrec_calc_place_visits": [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int64, C.c_double, C.c_int32,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _i64p],
    "locrec_distance_meters": [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p],
    "locrec_rank_recommendations": [C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                    C.c_int32, C.c_void_p, C.c_void_p, _i64p],
}
_RESTYPE = {"locrec_last_error": C.c_char_p, "locrec_version": C.c_char_p, "locrec_knn_replicas_destroy": None,
            "locrec_sg_sharded_destroy": None, "locrec_sg_group_destroy": None}

_lib = None


def lib():
    """The loaded library; raises if it has not been built (see __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LocrecRuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)")
        try:
            # torch ships its own libamdhip64 with the same SONAME; import it first so that this
            # process ends up with ONE HIP runtime whichever order the caller imports things in
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the header and the library disagree
            fn.argtypes = argtypes
            fn.restype = _RESTYPE.get(name, C.c_int32)
        _lib = handle
    return _lib


def check(status):
    if status == OK:
        return
    msg = lib().locrec_last_error().decode("utf-8", "replace")
    if status in (E_INVALID_ARG, E_NOT_FOUND):
        raise IllegalArgumentException(msg)
    if status == E_OOM:
        raise MemoryError(msg)
    if status == E_ARITHMETIC:
        raise ArithmeticError(msg)   # java.lang.ArithmeticException (RatingVectorsBuilder.scala:36-41)
    raise LocrecRuntimeError(msg)


def require_current_device(tensors):
    """The library's kernels run on the CURRENT HIP device and take raw pointers: a tensor that lives on another GPU
    would be a cross-device access from those kernels (a memory fault without peer access), and outputs would be
    allocated where the kernels do not run.  IllegalArgumentException instead."""
    import torch
    cur = torch.cuda.current_device()
    for t in tensors:
        if t is not None and t.device.index != cur:
            raise IllegalArgumentException(
                f"requirement failed: device arrays must live on the current device cuda:{cur}, got {t.device} "
                f"(select it with torch.cuda.set_device / locrec_set_device first)")


def device_allocations():
    """hipMalloc calls the library has made in this process so far (locrec_device_allocations)."""
    n = C.c_int64()
    check(lib().locrec_device_allocations(C.byref(n)))
    return n.value


def device_bytes_in_use():
    """Bytes of device memory the library holds right now (locrec_device_bytes_in_use)."""
    n = C.c_int64()
    check(lib().locrec_device_bytes_in_use(C.byref(n)))
    return n.value


CACHE_KNN, CACHE_SG = 0, 1


def cache_stats():
    """{entries, entry_bytes, hits, misses, evictions} of the process-wide handle cache."""
    v = [C.c_int64() for _ in range(5)]
    check(lib().locrec_cache_stats(*[C.byref(x) for x in v]))
    return dict(zip(("entries", "entry_bytes", "hits", "misses", "evictions"), (x.value for x in v)))


def ptr(a, ctype):
    """Pointer to a C-contiguous numpy array of the matching dtype (or NULL for None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "array must be contiguous"
    return a.ctypes.data_as(C.POINTER(ctype))


def as_i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def as_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
