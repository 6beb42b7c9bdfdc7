"""ctypes binding of liblocrec_parquet.so (include/locrec_parquet.h): the reference's Parquet sets -> device handles by
native code (Arrow C++), the path a JVM takes through JNI instead of `spark.read.parquet(...).collect()`.
mains.py holds the numpy / pyarrow reader of the same files - the second restatement these are checked against."""
import ctypes as C
import os

import numpy as np

from . import _lib as L

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblocrec_parquet.so")


class _KnnArrays(C.Structure):
    _fields_ = [("n", C.c_int64), ("person_ids", C.POINTER(C.c_int64)), ("p_rowptr", C.POINTER(C.c_int64)),
                ("p_idx", C.POINTER(C.c_int32)), ("p_val", C.POINTER(C.c_double)), ("p_dim", C.c_int32),
                ("c_rowptr", C.POINTER(C.c_int64)), ("c_idx", C.POINTER(C.c_int32)), ("c_val", C.POINTER(C.c_double)),
                ("c_dim", C.c_int32), ("r_rowptr", C.POINTER(C.c_int64)), ("r_place", C.POINTER(C.c_int64)),
                ("r_rating", C.POINTER(C.c_int64))]


class _Edges(C.Structure):
    _fields_ = [("n_edges", C.c_int64), ("source_ids", C.POINTER(C.c_int64)), ("target_ids", C.POINTER(C.c_int64)),
                ("balanced_weights", C.POINTER(C.c_double))]


SIGNATURES = {
    "locrec_parquet_last_error": ([], C.c_char_p),
    "locrec_parquet_read_knn": ([C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.POINTER(_KnnArrays))], C.c_int32),
    "locrec_parquet_free_knn": ([C.POINTER(_KnnArrays)], None),
    "locrec_knn_create_from_parquet": ([C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)], C.c_int32),
    "locrec_parquet_read_edges": ([C.c_char_p, C.POINTER(C.POINTER(_Edges))], C.c_int32),
    "locrec_parquet_free_edges": ([C.POINTER(_Edges)], None),
    "locrec_sg_create_from_parquet": ([C.c_char_p, C.POINTER(C.c_void_p)], C.c_int32),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise L.LocrecRuntimeError(f"{LIB_PATH} is missing: `make -C locations-recommender_amd/csrc parquet` "
                                       "(needs the Arrow C++ headers and libraries of the pyarrow wheel)")
        L.lib()  # liblocrec.so (and the HIP runtime torch ships) first
        h = C.CDLL(LIB_PATH)
        for name, (argtypes, restype) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.argtypes, fn.restype = argtypes, restype
        _lib = h
    return _lib


def _check(status):
    if status == L.OK:
        return
    msg = lib().locrec_parquet_last_error().decode("utf-8", "replace")
    if status in (L.E_INVALID_ARG, L.E_NOT_FOUND):
        raise L.IllegalArgumentException(msg)
    if status == L.E_OOM:
        raise MemoryError(msg)
    raise L.LocrecRuntimeError(msg)


def _np(ptr, n, dtype):
    return np.ctypeslib.as_array(ptr, shape=(max(int(n), 1),))[:int(n)].astype(dtype, copy=True) if n else np.empty(0, dtype)


def read_knn(place_rating_vectors, category_rating_vectors, place_ratings):
    """The arrays of locrec_knn_create as numpy copies (dict with the keys the tests' datasets use)."""
    a = C.POINTER(_KnnArrays)()
    _check(lib().locrec_parquet_read_knn(os.fsencode(place_rating_vectors), os.fsencode(category_rating_vectors),
                                         os.fsencode(place_ratings), C.byref(a)))
    try:
        s = a.contents
        n = s.n
        pe = s.p_rowptr[n] if n else 0
        ce = s.c_rowptr[n] if n else 0
        re_ = s.r_rowptr[n] if n else 0
        return {"person_ids": _np(s.person_ids, n, np.int64), "p_rowptr": _np(s.p_rowptr, n + 1, np.int64),
                "p_idx": _np(s.p_idx, pe, np.int32), "p_val": _np(s.p_val, pe, np.float64), "p_dim": int(s.p_dim),
                "c_rowptr": _np(s.c_rowptr, n + 1, np.int64), "c_idx": _np(s.c_idx, ce, np.int32),
                "c_val": _np(s.c_val, ce, np.float64), "c_dim": int(s.c_dim), "r_rowptr": _np(s.r_rowptr, n + 1, np.int64),
                "r_place": _np(s.r_place, re_, np.int64), "r_rating": _np(s.r_rating, re_, np.int64)}
    finally:
        lib().locrec_parquet_free_knn(a)


def read_edges(stochastic_graph):
    e = C.POINTER(_Edges)()
    _check(lib().locrec_parquet_read_edges(os.fsencode(stochastic_graph), C.byref(e)))
    try:
        s = e.contents
        return _np(s.source_ids, s.n_edges, np.int64), _np(s.target_ids, s.n_edges, np.int64), _np(s.balanced_weights, s.n_edges, np.float64)
    finally:
        lib().locrec_parquet_free_edges(e)


def knn_index(place_rating_vectors, category_rating_vectors, place_ratings):
    """locrec_knn_create_from_parquet -> KnnIndex (no Python touches the data)."""
    from .knn import KnnIndex
    h = C.c_void_p()
    _check(lib().locrec_knn_create_from_parquet(os.fsencode(place_rating_vectors), os.fsencode(category_rating_vectors),
                                                os.fsencode(place_ratings), C.byref(h)))
    ix = KnnIndex.__new__(KnnIndex)
    ix._h, ix.person_ids = h, None
    return ix


def sg_graph(stochastic_graph):
    from .stochastic import SgGraph
    h = C.c_void_p()
    _check(lib().locrec_sg_create_from_parquet(os.fsencode(stochastic_graph), C.byref(h)))
    g = SgGraph.__new__(SgGraph)
    g._h = h
    return g
