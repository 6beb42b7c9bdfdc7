"""ctypes binding of oracle/liblocrec_oracle.so -- TEST INFRASTRUCTURE (the checker).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ORACLE_DIR, "liblocrec_oracle.so")

OK, E_INVALID_ARG, E_NOT_FOUND = 0, 1, 2
_i64p, _i32p, _f64p = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)
_lib = None


class OracleIllegalArgument(ValueError):
    pass


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ORACLE_DIR, "locrec_oracle.c")
        if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
            subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)
        h = C.CDLL(SO)
        h.oracle_vector_length.restype = C.c_double
        h.oracle_vector_length.argtypes = [_f64p, C.c_int64]
        h.oracle_cosine.restype = C.c_double
        h.oracle_cosine.argtypes = [C.c_int64, _i32p, _f64p, C.c_int64, _i32p, _f64p]
        h.oracle_sparse_dot.restype = C.c_double
        h.oracle_sparse_dot.argtypes = [C.c_int64, _i32p, _f64p, C.c_int64, _i32p, _f64p]
        csr = [C.c_int64, _i64p, _i64p, _i32p, _f64p, _i64p, _i32p, _f64p]
        h.oracle_knn_similar.restype = C.c_int32
        h.oracle_knn_similar.argtypes = csr + [C.c_int64, C.c_double, C.c_double, C.c_int64, _i64p, _f64p, _i64p]
        h.oracle_knn_recommend.restype = C.c_int32
        h.oracle_knn_recommend.argtypes = csr + [_i64p, _i64p, _i64p, C.c_int64, C.c_double, C.c_double, C.c_int64,
                                                 _i64p, _f64p, _i64p]
        h.oracle_knn_similar_batch.restype = C.c_int32
        h.oracle_knn_similar_batch.argtypes = csr + [C.c_int64, _i64p, C.c_double, C.c_double, C.c_int64,
                                                     _i64p, _f64p, _i64p, C.c_int32]
        h.oracle_sg_recommend.restype = C.c_int32
        h.oracle_sg_recommend.argtypes = [C.c_int64, _i64p, _i64p, _f64p, C.c_int64, C.c_double, C.c_double, C.c_int64,
                                          _i64p, _f64p, _i64p, _i64p, _i32p]
        h.oracle_sg_sweeps_csr.restype = C.c_double
        h.oracle_sg_sweeps_csr.argtypes = [C.c_int64, _i64p, _i32p, _f64p, C.c_int64, C.c_double, C.c_int64, _f64p,
                                           C.c_int32]
        h.oracle_calc_ratings.restype = C.c_int32
        h.oracle_calc_ratings.argtypes = [C.c_int64, _i64p, _i64p, C.c_int64, _i64p, _i64p, _i64p, _i64p]
        h.oracle_calc_rating_vectors.restype = C.c_int32
        h.oracle_calc_rating_vectors.argtypes = [C.c_int64, _i64p, _i64p, _i64p, _i64p, _i64p, _i32p, _f64p, _i64p, _i64p, _i64p]
        pp = C.POINTER(C.c_void_p)
        h.oracle_balanced_edges.restype = C.c_int32
        h.oracle_balanced_edges.argtypes = [C.c_int32, _f64p, _i64p, pp, pp, pp, _i64p, _i64p, _f64p]
        h.oracle_distance_meters.restype = C.c_double
        h.oracle_distance_meters.argtypes = [C.c_double] * 4
        h.oracle_place_visits.restype = C.c_int32
        h.oracle_place_visits.argtypes = [C.c_int64, _i64p, _f64p, _f64p, _i64p, C.c_int64, _f64p, _f64p, _i64p,
                                          C.c_int64, C.c_double, C.c_int64, _i64p, _i64p, _i64p]
        h.oracle_rank_recommendations.restype = C.c_int32
        h.oracle_rank_recommendations.argtypes = [C.c_int64, _i64p, _f64p, C.c_int64, _i64p, _i64p, C.c_int64, C.c_int64,
                                                  _i64p, _f64p, _i64p]
        _lib = h
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _check(st):
    if st != OK:
        raise OracleIllegalArgument(f"oracle status {st}")


def vector_length(values):
    v = np.ascontiguousarray(values, np.float64)
    return lib().oracle_vector_length(_p(v, C.c_double), len(v))


def cosine(i1, v1, i2, v2):
    i1, i2 = np.ascontiguousarray(i1, np.int32), np.ascontiguousarray(i2, np.int32)
    v1, v2 = np.ascontiguousarray(v1, np.float64), np.ascontiguousarray(v2, np.float64)
    return lib().oracle_cosine(len(i1), _p(i1, C.c_int32), _p(v1, C.c_double), len(i2), _p(i2, C.c_int32),
                               _p(v2, C.c_double))


def _csr_args(d):
    a = dict(ids=np.ascontiguousarray(d["person_ids"], np.int64),
             prp=np.ascontiguousarray(d["p_rowptr"], np.int64), pidx=np.ascontiguousarray(d["p_idx"], np.int32),
             pval=np.ascontiguousarray(d["p_val"], np.float64),
             crp=np.ascontiguousarray(d["c_rowptr"], np.int64), cidx=np.ascontiguousarray(d["c_idx"], np.int32),
             cval=np.ascontiguousarray(d["c_val"], np.float64))
    args = [len(a["ids"]), _p(a["ids"], C.c_int64), _p(a["prp"], C.c_int64), _p(a["pidx"], C.c_int32),
            _p(a["pval"], C.c_double), _p(a["crp"], C.c_int64), _p(a["cidx"], C.c_int32), _p(a["cval"], C.c_double)]
    return a, args


def knn_similar(d, person_id, pw, cw, k):
    keep, args = _csr_args(d)
    n = len(keep["ids"])
    ids = np.empty(max(n, 1), np.int64)
    sims = np.empty(max(n, 1), np.float64)
    cnt = C.c_int64(max(n, 1))
    _check(lib().oracle_knn_similar(*args, int(person_id), float(pw), float(cw), int(k), _p(ids, C.c_int64),
                                    _p(sims, C.c_double), C.byref(cnt)))
    return ids[:cnt.value], sims[:cnt.value]


def ratings_of(d):
    """placeRatings CSR: explicit if present, else the place vectors (what the builder writes)."""
    if "r_rowptr" in d:
        return (np.ascontiguousarray(d["r_rowptr"], np.int64), np.ascontiguousarray(d["r_place"], np.int64),
                np.ascontiguousarray(d["r_rating"], np.int64))
    return (np.ascontiguousarray(d["p_rowptr"], np.int64), np.ascontiguousarray(d["p_idx"], np.int64),
            np.ascontiguousarray(d["p_val"], np.int64))


def knn_recommend(d, person_id, pw, cw, k):
    keep, args = _csr_args(d)
    rrp, rpl, rra = ratings_of(d)
    cap = max(1, len(rpl))
    places = np.empty(cap, np.int64)
    est = np.empty(cap, np.float64)
    cnt = C.c_int64(cap)
    _check(lib().oracle_knn_recommend(*args, _p(rrp, C.c_int64), _p(rpl, C.c_int64), _p(rra, C.c_int64),
                                      int(person_id), float(pw), float(cw), int(k), _p(places, C.c_int64),
                                      _p(est, C.c_double), C.byref(cnt)))
    return places[:cnt.value], est[:cnt.value]


def knn_similar_batch(d, qrows, pw, cw, k, nthreads=1):
    keep, args = _csr_args(d)
    q = np.ascontiguousarray(qrows, np.int64)
    ids = np.empty((len(q), k), np.int64)
    sims = np.empty((len(q), k), np.float64)
    cnt = np.empty(len(q), np.int64)
    _check(lib().oracle_knn_similar_batch(*args, len(q), _p(q, C.c_int64), float(pw), float(cw), int(k),
                                          _p(ids, C.c_int64), _p(sims, C.c_double), _p(cnt, C.c_int64), int(nthreads)))
    return ids, sims, cnt


def sg_recommend(src, dst, w, vertex_id, alpha, epsilon, max_iterations):
    s, t = np.ascontiguousarray(src, np.int64), np.ascontiguousarray(dst, np.int64)
    ww = np.ascontiguousarray(w, np.float64)
    cap = 2 * len(s) + 1
    ids = np.empty(cap, np.int64)
    probs = np.empty(cap, np.float64)
    cnt, it, conv = C.c_int64(cap), C.c_int64(), C.c_int32()
    _check(lib().oracle_sg_recommend(len(s), _p(s, C.c_int64), _p(t, C.c_int64), _p(ww, C.c_double), int(vertex_id),
                                     float(alpha), float(epsilon), int(max_iterations), _p(ids, C.c_int64),
                                     _p(probs, C.c_double), C.byref(cnt), C.byref(it), C.byref(conv)))
    return ids[:cnt.value], probs[:cnt.value], it.value, bool(conv.value)


def sg_sweeps_csr(src, dst, w, vertex_id, alpha, sweeps, nthreads=1):
    """`sweeps` applications of calcNextX (StochasticRecommender.scala:108-128), multi-threaded over
    targets.  -> (sorted vertex ids, x, seconds spent in the sweeps alone)."""
    import time
    s, t = np.ascontiguousarray(src, np.int64), np.ascontiguousarray(dst, np.int64)
    ww = np.ascontiguousarray(w, np.float64)
    vid, inv = np.unique(np.concatenate([s, t]), return_inverse=True)
    cs, ct = inv[:len(s)], inv[len(s):]
    order = np.argsort(ct, kind="stable")           # CSR by target, in-row order = edge-list order
    rowptr = np.zeros(len(vid) + 1, np.int64)
    np.cumsum(np.bincount(ct, minlength=len(vid)), out=rowptr[1:])
    col = np.ascontiguousarray(cs[order], np.int32)
    wv = np.ascontiguousarray(ww[order])
    target = int(np.searchsorted(vid, vertex_id))
    if target >= len(vid) or vid[target] != vertex_id:
        raise OracleIllegalArgument(f"No such vertex in the graph: {vertex_id}")
    x = np.full(len(vid), 1.0 / len(vid))
    t0 = time.perf_counter()
    lib().oracle_sg_sweeps_csr(len(vid), _p(rowptr, C.c_int64), _p(col, C.c_int32), _p(wv, C.c_double), target,
                               float(alpha), int(sweeps), _p(x, C.c_double), int(nthreads))
    return vid, x, time.perf_counter() - t0


E_ARITHMETIC = 5


def calc_ratings(person_ids, entity_ids, top_n):
    p, e = np.ascontiguousarray(person_ids, np.int64), np.ascontiguousarray(entity_ids, np.int64)
    n = len(p)
    op, oe, orr = (np.empty(max(n, 1), np.int64) for _ in range(3))
    cnt = C.c_int64()
    _check(lib().oracle_calc_ratings(n, _p(p, C.c_int64), _p(e, C.c_int64), int(top_n), _p(op, C.c_int64), _p(oe, C.c_int64),
                                     _p(orr, C.c_int64), C.byref(cnt)))
    return op[:cnt.value], oe[:cnt.value], orr[:cnt.value]


def calc_rating_vectors(person_ids, entity_ids, ratings):
    p, e = np.ascontiguousarray(person_ids, np.int64), np.ascontiguousarray(entity_ids, np.int64)
    r = np.ascontiguousarray(ratings, np.int64)
    n = len(p)
    ids, ptr = np.empty(max(n, 1), np.int64), np.empty(n + 1, np.int64)
    idx, val = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.float64)
    npers, nnz, size = C.c_int64(), C.c_int64(), C.c_int64()
    st = lib().oracle_calc_rating_vectors(n, _p(p, C.c_int64), _p(e, C.c_int64), _p(r, C.c_int64), _p(ids, C.c_int64),
                                          _p(ptr, C.c_int64), _p(idx, C.c_int32), _p(val, C.c_double), C.byref(npers),
                                          C.byref(nnz), C.byref(size))
    if st == E_ARITHMETIC:
        raise ArithmeticError(f"Index out of Int range: {size.value}")
    _check(st)
    return ids[:npers.value], ptr[:npers.value + 1], idx[:nnz.value], val[:nnz.value], int(size.value)


def balanced_edges(betas, fams):
    nf = len(fams)
    cols = [[np.ascontiguousarray(f[0], np.int64), np.ascontiguousarray(f[1], np.int64), np.ascontiguousarray(f[2], np.float64)]
            for f in fams]
    counts = np.array([len(c[0]) for c in cols], np.int64)
    b = np.ascontiguousarray(betas, np.float64)
    arr = [(C.c_void_p * max(nf, 1))(*[c[k].ctypes.data for c in cols]) for k in range(3)]
    total = int(counts.sum())
    os_, ot, ow = np.empty(max(total, 1), np.int64), np.empty(max(total, 1), np.int64), np.empty(max(total, 1), np.float64)
    _check(lib().oracle_balanced_edges(nf, _p(b, C.c_double), _p(counts, C.c_int64), arr[0], arr[1], arr[2],
                                       _p(os_, C.c_int64), _p(ot, C.c_int64), _p(ow, C.c_double)))
    return os_[:total], ot[:total], ow[:total]


def distance_meters(lat1, lon1, lat2, lon2):
    return lib().oracle_distance_meters(float(lat1), float(lon1), float(lat2), float(lon2))


def place_visits(visits, places, visits_from, max_meters=100.0):
    """-> (visit rows, place rows) of the matches, ordered by (visit, place); raises on an invalid Location
    with .row = ("visit" | "place", index)."""
    vt = np.ascontiguousarray(visits["timestamp"], np.int64)
    vlat, vlon = np.ascontiguousarray(visits["latitude"], np.float64), np.ascontiguousarray(visits["longitude"], np.float64)
    vr = np.ascontiguousarray(visits["region_id"], np.int64)
    plat, plon = np.ascontiguousarray(places["latitude"], np.float64), np.ascontiguousarray(places["longitude"], np.float64)
    pr = np.ascontiguousarray(places["region_id"], np.int64)
    cap = 1 << 16
    while True:
        ov, op = np.empty(cap, np.int64), np.empty(cap, np.int64)
        cnt = C.c_int64()
        st = lib().oracle_place_visits(len(vt), _p(vt, C.c_int64), _p(vlat, C.c_double), _p(vlon, C.c_double), _p(vr, C.c_int64),
                                       len(plat), _p(plat, C.c_double), _p(plon, C.c_double), _p(pr, C.c_int64),
                                       int(visits_from), float(max_meters), cap, _p(ov, C.c_int64), _p(op, C.c_int64),
                                       C.byref(cnt))
        if st != OK:
            err = OracleIllegalArgument(f"invalid Location, code {cnt.value}")
            k = -cnt.value - 1
            err.row = ("visit", k) if k < len(vt) else ("place", k - len(vt))
            raise err
        if cnt.value <= cap:
            return ov[:cnt.value], op[:cnt.value]
        cap = cnt.value


def rank_recommendations(ids, scores, place_ids, place_region_ids, target_region_id, max_recommendations):
    i, sc = np.ascontiguousarray(ids, np.int64), np.ascontiguousarray(scores, np.float64)
    pi, pr = np.ascontiguousarray(place_ids, np.int64), np.ascontiguousarray(place_region_ids, np.int64)
    oi, osc = np.empty(max(len(i), 1), np.int64), np.empty(max(len(i), 1), np.float64)
    cnt = C.c_int64()
    _check(lib().oracle_rank_recommendations(len(i), _p(i, C.c_int64), _p(sc, C.c_double), len(pi), _p(pi, C.c_int64),
                                             _p(pr, C.c_int64), int(target_region_id), int(max_recommendations),
                                             _p(oi, C.c_int64), _p(osc, C.c_double), C.byref(cnt)))
    return oi[:cnt.value], osc[:cnt.value]
