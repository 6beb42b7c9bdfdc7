"""The producers of the two hot paths' inputs, computed by liblocrec.so's kernels (csrc/prep.hip,
SURVEY.md 8f rows f-2 and f-4) behind the names of the reference's builder objects:

    calc_ratings                 RatingsBuilder.calcRatings               knn/RatingsBuilder.scala:32-48
    calc_rating_vectors          RatingVectorsBuilder.calcRatingVectors   knn/RatingVectorsBuilder.scala:10-25,52-84
    build_with_balanced_weights  StochasticGraphBuilder.buildWithBalancedWeights
                                                                          stochastic/StochasticGraphBuilder.scala:8-28
    calc_place_visits            PlaceVisits.calcPlaceVisits              PlaceVisits.scala:11-46
    distance_meters              Location.distanceMeters                  Location.scala:30-38
    rank_recommendations         printRecommendations of both mains       knn/KnnRecommenderMain.scala:90-101

Every function takes numpy arrays (host in, host out) or torch CUDA tensors (device in, device out:
nothing passes through the host, and the outputs of calc_rating_vectors go straight into
KnnIndex.from_device).  There is no CPU fallback: without the HIP library / a GPU they raise.
"""
import ctypes as C

import numpy as np

from . import _lib as L

DISTANCE_ACCURACY_METERS = 100.0   # PlaceVisits.scala:127
VISITED_PLACES_TOP_N = 100         # RatingsBuilder.scala:9
VISITED_CATEGORIES_TOP_N = 10      # RatingsBuilder.scala:10


def _is_tensor(a):
    return type(a).__module__.startswith("torch") and hasattr(a, "data_ptr")


class _Cols:
    """Columns of one call, all host (numpy) or all device (torch CUDA tensors)."""

    def __init__(self, *arrays):
        self.device = any(_is_tensor(a) for a in arrays)
        if self.device:
            import torch
            self.torch = torch
            assert all(_is_tensor(a) and a.is_cuda for a in arrays), "all columns must be CUDA tensors (or all numpy)"
            self.dev = arrays[0].device
            L.require_current_device(arrays)
            torch.cuda.current_stream(self.dev).synchronize()   # the library works on its own stream
        self.mem = L.MEM_DEVICE if self.device else L.MEM_HOST
        self._keep = []

    def col(self, a, np_dtype):
        if self.device:
            want = getattr(self.torch, np.dtype(np_dtype).name)
            a = a.to(want).contiguous()
            self._keep.append(a)
            return C.c_void_p(a.data_ptr()) if a.numel() else None
        a = np.ascontiguousarray(a, np_dtype)
        self._keep.append(a)
        return C.c_void_p(a.ctypes.data) if a.size else None

    def out(self, n, np_dtype):
        n = max(int(n), 1)
        if self.device:
            a = self.torch.empty(n, dtype=getattr(self.torch, np.dtype(np_dtype).name), device=self.dev)
            return a, C.c_void_p(a.data_ptr())
        a = np.empty(n, np_dtype)
        return a, C.c_void_p(a.ctypes.data)


def calc_ratings(person_ids, entity_ids, top_n):
    """RatingsBuilder.calcRatings: visits -> (person_id, entity_id, rating = number of visits), keeping
    per person the entities whose SQL rank() by rating descending is <= top_n (ties share a rank: a tie
    straddling top_n is kept whole).  Rows ordered by (person, entity)."""
    c = _Cols(person_ids, entity_ids)
    n = len(person_ids)
    assert len(entity_ids) == n
    p, e = c.col(person_ids, np.int64), c.col(entity_ids, np.int64)
    (op, opp), (oe, oep), (orr, orp) = c.out(n, np.int64), c.out(n, np.int64), c.out(n, np.int64)
    cnt = C.c_int64()
    L.check(L.lib().locrec_calc_ratings(n, p, e, int(top_n), c.mem, opp, oep, orp, C.byref(cnt)))
    m = cnt.value
    return op[:m], oe[:m], orr[:m]


def calc_rating_vectors(person_ids, entity_ids, ratings):
    """RatingVectorsBuilder.calcRatingVectors: one SparseVector per person as CSR.
    -> person_ids (ascending), rowptr, indices (int32, ascending per person), values (float64), size."""
    c = _Cols(person_ids, entity_ids, ratings)
    n = len(person_ids)
    assert len(entity_ids) == n and len(ratings) == n
    p, e, r = c.col(person_ids, np.int64), c.col(entity_ids, np.int64), c.col(ratings, np.int64)
    (oid, oidp), (optr, optrp) = c.out(n, np.int64), c.out(n + 1, np.int64)
    (oidx, oidxp), (oval, ovalp) = c.out(n, np.int32), c.out(n, np.float64)
    npers, nnz, size = C.c_int64(), C.c_int64(), C.c_int64()
    L.check(L.lib().locrec_calc_rating_vectors(n, p, e, r, c.mem, oidp, optrp, oidxp, ovalp, C.byref(npers), C.byref(nnz),
                                               C.byref(size)))
    return oid[:npers.value], optr[:npers.value + 1], oidx[:nnz.value], oval[:nnz.value], int(size.value)


def build_with_balanced_weights(betas, all_edges):
    """StochasticGraphBuilder.buildWithBalancedWeights: every family's weight times its beta, families
    concatenated in the given order.  all_edges: (source_id, target_id, weight) triples or mappings with
    those keys.  -> (source_id, target_id, balanced_weight)."""
    fams = []
    for e in all_edges:
        fams.append((e["source_id"], e["target_id"], e["weight"]) if hasattr(e, "keys") or hasattr(e, "columns") else tuple(e))
    if len(betas) != len(fams) or not fams:
        raise L.IllegalArgumentException("one beta per edge family is required")
    c = _Cols(*[a for f in fams for a in f])
    nf = len(fams)
    counts = (C.c_int64 * nf)(*[len(f[0]) for f in fams])
    b = (C.c_double * nf)(*[float(x) for x in betas])
    src, dst, w = (C.c_void_p * nf)(), (C.c_void_p * nf)(), (C.c_void_p * nf)()
    for i, f in enumerate(fams):
        assert len(f[1]) == len(f[0]) and len(f[2]) == len(f[0])
        src[i], dst[i], w[i] = c.col(f[0], np.int64), c.col(f[1], np.int64), c.col(f[2], np.float64)
    total = sum(counts)
    (os_, osp), (ot, otp), (ow, owp) = c.out(total, np.int64), c.out(total, np.int64), c.out(total, np.float64)
    L.check(L.lib().locrec_build_balanced_edges(nf, b, counts, src, dst, w, c.mem, osp, otp, owp))
    return os_[:total], ot[:total], ow[:total]


def calc_place_visits(visits, places, visits_from, max_meters=DISTANCE_ACCURACY_METERS):
    """PlaceVisits.calcPlaceVisits.  visits: mapping with person_id, timestamp (int64), latitude,
    longitude, region_id; places: mapping with id, latitude, longitude, region_id, category_id;
    visits_from: the timestamp calcVisitsFromTimestamp yields (PlaceVisits.scala:48-58).
    -> dict(person_id, timestamp, place_id, region_id, category_id), ordered by (visit row, place row)."""
    vcols = [visits[k] for k in ("person_id", "timestamp", "latitude", "longitude", "region_id")]
    pcols = [places[k] for k in ("id", "latitude", "longitude", "region_id", "category_id")]
    c = _Cols(*vcols, *pcols)
    nv, npl = len(vcols[0]), len(pcols[0])
    va = [c.col(vcols[0], np.int64), c.col(vcols[1], np.int64), c.col(vcols[2], np.float64), c.col(vcols[3], np.float64),
          c.col(vcols[4], np.int64)]
    pa = [c.col(pcols[0], np.int64), c.col(pcols[1], np.float64), c.col(pcols[2], np.float64), c.col(pcols[3], np.int64),
          c.col(pcols[4], np.int64)]
    cnt = C.c_int64(0)   # first call: count only
    L.check(L.lib().locrec_calc_place_visits(nv, *va, npl, *pa, int(visits_from), float(max_meters), c.mem,
                                             None, None, None, None, None, C.byref(cnt)))
    m = cnt.value
    outs = [c.out(m, np.int64) for _ in range(5)]
    cnt = C.c_int64(m)
    if m:
        L.check(L.lib().locrec_calc_place_visits(nv, *va, npl, *pa, int(visits_from), float(max_meters), c.mem,
                                                 *[o[1] for o in outs], C.byref(cnt)))
    names = ("person_id", "timestamp", "place_id", "region_id", "category_id")
    return {k: o[0][:m] for k, o in zip(names, outs)}


def distance_meters(lat1, lon1, lat2, lon2):
    """Location.distanceMeters of n pairs, by the device code the join uses (NaN for an invalid Location)."""
    c = _Cols(lat1, lon1, lat2, lon2)
    n = len(lat1)
    a = [c.col(x, np.float64) for x in (lat1, lon1, lat2, lon2)]
    out, outp = c.out(n, np.float64)
    L.check(L.lib().locrec_distance_meters(n, *a, c.mem, outp))
    return out[:n]


def rank_recommendations(ids, scores, place_ids, place_region_ids, target_region_id, max_recommendations):
    """printRecommendations of both mains (KnnRecommenderMain.scala:90-101, StochasticRecommenderMain.scala:
    64-75): the target region's places JOIN the recommendations ON id, ORDER BY score DESC, LIMIT n.
    Ties: id ascending (Spark leaves them undefined).  -> (ids, scores)."""
    c = _Cols(ids, scores, place_ids, place_region_ids)
    n, npl = len(ids), len(place_ids)
    assert len(scores) == n and len(place_region_ids) == npl
    a = [c.col(ids, np.int64), c.col(scores, np.float64)]
    p = [c.col(place_ids, np.int64), c.col(place_region_ids, np.int64)]
    cap = max(0, min(n, int(max_recommendations)))
    (oi, oip), (osc, oscp) = c.out(cap, np.int64), c.out(cap, np.float64)
    cnt = C.c_int64()
    L.check(L.lib().locrec_rank_recommendations(n, a[0], a[1], npl, p[0], p[1], int(target_region_id), int(max_recommendations),
                                                c.mem, oip, oscp, C.byref(cnt)))
    return oi[:cnt.value], osc[:cnt.value]


def knn_index_from_visits(person_ids, place_ids, category_ids, places_top_n=VISITED_PLACES_TOP_N,
                          categories_top_n=VISITED_CATEGORIES_TOP_N):
    """RatingVectorsBuilderMain's pipeline (RatingVectorsBuilderMain.scala:38-73) on the device: place
    visits (person_id, place_id, category_id) -> place / category ratings -> rating vectors -> KnnIndex,
    with placeRatings = the place ratings.  With CUDA tensors nothing passes through the host."""
    from .knn import KnnIndex
    pp, pe, pr = calc_ratings(person_ids, place_ids, places_top_n)
    cp, ce, cr = calc_ratings(person_ids, category_ids, categories_top_n)
    ids, p_ptr, p_idx, p_val, p_dim = calc_rating_vectors(pp, pe, pr)
    ids_c, c_ptr, c_idx, c_val, c_dim = calc_rating_vectors(cp, ce, cr)
    # both sets come from the same visits, so they name the same persons (every visit has a place and a category)
    same = len(ids) == len(ids_c) and bool((ids == ids_c).all())
    if not same:
        raise L.IllegalArgumentException("place and category visits name different persons")
    if _is_tensor(ids):
        return KnnIndex.from_device(ids, p_ptr.contiguous(), p_idx.contiguous(), p_val.contiguous(), p_dim,
                                    c_ptr.contiguous(), c_idx.contiguous(), c_val.contiguous(), c_dim,
                                    p_ptr.contiguous(), pe.contiguous(), pr.contiguous())
    return KnnIndex(ids, p_ptr, p_idx, p_val, p_dim, c_ptr, c_idx, c_val, c_dim, p_ptr, pe, pr)
