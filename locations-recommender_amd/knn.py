"""Host-side mirror of knn/KnnRecommender.scala:9-25 over the C ABI.

Same constructor arguments, method name, column names and error behaviour as
the Scala class; the three DataFrames are pandas frames with the reference's
schemas (RatingVectorsBuilder.scala:81-82, RatingsBuilder.scala:38-47):
  placeRatingVectors / categoryRatingVectors: (person_id: long, rating_vector: SparseVector)
  placeRatings:                               (person_id: long, place_id: long, rating: long)
"""
import ctypes as C

import numpy as np

from . import _cache
from . import _lib as L


class SparseVector:
    """org.apache.spark.ml.linalg.SparseVector(size, indices, values)."""

    __slots__ = ("size", "indices", "values")

    def __init__(self, size, indices, values):
        self.size = int(size)
        self.indices = np.asarray(indices, dtype=np.int32)
        self.values = np.asarray(values, dtype=np.float64)
        if self.indices.shape != self.values.shape:
            raise L.IllegalArgumentException("requirement failed: Sparse vectors require that the dimension of the "
                                             "indices match the dimension of the values.")


def _vectors_to_csr(person_ids, frame):
    """Collect a (person_id, rating_vector) frame to CSR rows aligned with person_ids."""
    by_id = {int(p): v for p, v in zip(frame["person_id"], frame["rating_vector"])}
    sizes = {v.size for v in by_id.values()}
    if len(sizes) > 1:
        raise L.IllegalArgumentException(f"rating vectors of different sizes: {sorted(sizes)}")
    dim = sizes.pop() if sizes else 1
    rowptr = np.zeros(len(person_ids) + 1, dtype=np.int64)
    idx, val = [], []
    for i, p in enumerate(person_ids):
        v = by_id.get(int(p))
        if v is not None:
            idx.append(v.indices)
            val.append(v.values)
            rowptr[i + 1] = rowptr[i] + len(v.indices)
        else:
            rowptr[i + 1] = rowptr[i]
    idx = np.concatenate(idx) if idx else np.zeros(0, np.int32)
    val = np.concatenate(val) if val else np.zeros(0, np.float64)
    return rowptr, L.as_i32(idx), L.as_f64(val), dim


class KnnIndex:
    """Owner of a locrec_knn_index handle (CSR in, device-resident afterwards)."""

    def __init__(self, person_ids, p_rowptr, p_idx, p_val, p_dim, c_rowptr, c_idx, c_val, c_dim,
                 r_rowptr=None, r_place=None, r_rating=None):
        lib = L.lib()
        self._h = C.c_void_p()
        self.person_ids = L.as_i64(person_ids)
        a = [L.as_i64(p_rowptr), L.as_i32(p_idx), L.as_f64(p_val), L.as_i64(c_rowptr), L.as_i32(c_idx), L.as_f64(c_val)]
        r = [None, None, None] if r_rowptr is None else [L.as_i64(r_rowptr), L.as_i64(r_place), L.as_i64(r_rating)]
        L.check(lib.locrec_knn_create(
            len(self.person_ids), L.ptr(self.person_ids, C.c_int64),
            L.ptr(a[0], C.c_int64), L.ptr(a[1], C.c_int32), L.ptr(a[2], C.c_double), int(p_dim),
            L.ptr(a[3], C.c_int64), L.ptr(a[4], C.c_int32), L.ptr(a[5], C.c_double), int(c_dim),
            L.ptr(r[0], C.c_int64), L.ptr(r[1], C.c_int64), L.ptr(r[2], C.c_int64), C.byref(self._h)))

    @classmethod
    def from_device(cls, person_ids, p_rowptr, p_idx, p_val, p_dim, c_rowptr, c_idx, c_val, c_dim,
                    r_rowptr=None, r_place=None, r_rating=None):
        """The same index from arrays that already live in DEVICE memory (torch tensors on the current
        GPU: int64 ids / row pointers / rating columns, int32 indices, float64 values): nothing passes
        through the host, the index is built by kernels (locrec_knn_create_from_device)."""
        import torch

        def dev(t, dtype):
            if t is None:
                return None
            assert t.is_cuda and t.dtype == dtype and t.is_contiguous(), "device tensor of the ABI's dtype expected"
            return t

        t = [dev(person_ids, torch.int64), dev(p_rowptr, torch.int64), dev(p_idx, torch.int32), dev(p_val, torch.float64),
             dev(c_rowptr, torch.int64), dev(c_idx, torch.int32), dev(c_val, torch.float64),
             dev(r_rowptr, torch.int64), dev(r_place, torch.int64), dev(r_rating, torch.int64)]
        L.require_current_device(t)
        torch.cuda.current_stream().synchronize()  # the library reads the arrays on its own stream
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        ptr = [C.c_void_p(x.data_ptr()) if x is not None and x.numel() > 0 else None for x in t]
        L.check(L.lib().locrec_knn_create_from_device(
            int(t[0].numel()), ptr[0], ptr[1], ptr[2], ptr[3], int(p_dim), ptr[4], ptr[5], ptr[6], int(c_dim),
            ptr[7], ptr[8], ptr[9], C.byref(self._h)))
        self.person_ids = t[0].cpu().numpy()
        return self

    @classmethod
    def through_cache(cls, key, build):
        """The process-wide cached index for `key` (include/locrec.h, "Handle cache"); build() -> KnnIndex is
        called on a miss only and its handle becomes the cache's.  close() / garbage collection of the
        returned object drop a REFERENCE; the device index stays for the next constructor with this key."""
        h = _cache.acquire(L.CACHE_KNN, key)
        if h is None:
            before = L.device_bytes_in_use()
            built = build()
            h = _cache.publish(L.CACHE_KNN, key, built._h, L.device_bytes_in_use() - before)
            built._h = None          # owned by the cache now (or destroyed by it, if the key appeared meanwhile)
        self = cls.__new__(cls)
        self._h, self._cached, self.person_ids = h, True, None
        self.lock = _cache.handle_lock(h)
        return self

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_cached", False):
                _cache.release(L.CACHE_KNN, self._h)
            else:
                L.lib().locrec_knn_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def n(self):
        if self.person_ids is None:  # a cache hit never saw the input arrays
            if getattr(self, "_n", None) is None:
                self._n = self.info()["n"]
            return self._n
        return len(self.person_ids)

    def info(self):
        n, b, p = C.c_int64(), C.c_int64(), C.c_int32()
        L.check(L.lib().locrec_knn_info(self._h, C.byref(n), C.byref(b), C.byref(p)))
        bb = C.c_int64()
        L.check(L.lib().locrec_knn_batch_scan_bytes(self._h, C.byref(bb)))
        return {"n": n.value, "scan_bytes": b.value, "batch_scan_bytes": bb.value, "packed": bool(p.value), "mode": p.value}

    def ht_image_info(self):
        """{head_words, tail_postings, wide_rows} of the head / tail image (locrec_knn_ht_image_info)."""
        v = [C.c_int64() for _ in range(3)]
        L.check(L.lib().locrec_knn_ht_image_info(self._h, *[C.byref(x) for x in v]))
        return {"head_words": v[0].value, "tail_postings": v[1].value, "wide_rows": v[2].value}

    def scan_plan(self):
        """Plan of the last batched scan: kernel (1 = knn_scan, 2 = knn_scan_ht, 3 = knn_scan in head/tail mode), mode, query tile, waves."""
        v = [C.c_int32() for _ in range(4)]
        L.check(L.lib().locrec_knn_scan_plan(self._h, *[C.byref(x) for x in v]))
        return {"kernel": v[0].value, "mode": v[1].value, "query_tile": v[2].value, "waves": v[3].value}

    def scan_kernel_name(self):
        p = self.scan_plan()
        return {1: "knn_scan", 2: "knn_scan_ht", 3: "knn_scan"}.get(p["kernel"], "none") + f"<mode {p['mode']}, QT {p['query_tile']}, {p['waves']} waves>"

    def query_tile(self):
        return self.scan_plan()["query_tile"]

    def vector_lengths(self):
        lp, lc = np.empty(self.n, np.float64), np.empty(self.n, np.float64)
        L.check(L.lib().locrec_knn_vector_lengths(self._h, L.ptr(lp, C.c_double), L.ptr(lc, C.c_double)))
        return lp, lc

    def cosine_similarity(self, person_a, person_b):
        """Distance.cosineSimilarity (Distance.scala:7-9) of the two persons' place and category vectors, any sign."""
        p, c = C.c_double(), C.c_double()
        L.check(L.lib().locrec_knn_cosine_similarity(self._h, int(person_a), int(person_b), C.byref(p), C.byref(c)))
        return p.value, c.value

    def query(self, person_id, pw, cw, k):
        cap = int(max(1, min(k, max(1, self.n))))
        ids = np.empty(cap, np.int64)
        sims = np.empty(cap, np.float64)
        cnt = C.c_int64(cap)
        L.check(L.lib().locrec_knn_query(self._h, int(person_id), float(pw), float(cw), int(k),
                                         L.ptr(ids, C.c_int64), L.ptr(sims, C.c_double), C.byref(cnt)))
        m = min(cnt.value, cap)
        return ids[:m], sims[:m]

    def _out_buffers(self, capacity):
        """Reusable output arrays (grown on demand): a request must not pay for two fresh
        half-megabyte allocations - and their page faults - every time."""
        buf = getattr(self, "_rec_buf", None)
        if buf is None or len(buf[0]) < capacity:
            buf = (np.empty(capacity, np.int64), np.empty(capacity, np.float64))
            self._rec_buf = buf
        return buf

    def recommend(self, person_id, pw, cw, k, capacity=1 << 12):
        while True:
            places, est = self._out_buffers(capacity)
            cnt = C.c_int64(len(places))
            L.check(L.lib().locrec_knn_recommend(self._h, int(person_id), float(pw), float(cw), int(k),
                                                 L.ptr(places, C.c_int64), L.ptr(est, C.c_double), C.byref(cnt)))
            if cnt.value <= len(places):
                return places[:cnt.value].copy(), est[:cnt.value].copy()
            capacity = cnt.value

    def recommend_batch(self, person_ids, pw, cw, k):
        """makeRecommendations for many persons: (offsets[nq + 1], place_ids, estimated_ratings);
        rows of person i are offsets[i]:offsets[i + 1], ordered by place id."""
        q = L.as_i64(person_ids)
        off = np.zeros(len(q) + 1, np.int64)
        cap = C.c_int64(0)
        places, est = np.empty(0, np.int64), np.empty(0, np.float64)
        for _ in range(2):  # first call sizes the result, second fills it
            L.check(L.lib().locrec_knn_recommend_batch(self._h, len(q), L.ptr(q, C.c_int64), float(pw), float(cw), int(k),
                                                       L.ptr(off, C.c_int64), L.ptr(places, C.c_int64),
                                                       L.ptr(est, C.c_double), C.byref(cap)))
            if cap.value <= len(places):
                break
            places, est = np.empty(cap.value, np.int64), np.empty(cap.value, np.float64)
            cap = C.c_int64(len(places))
        return off, places[:off[-1]], est[:off[-1]]

    def recommend_range_async(self, first, nq, pw, cw, k):
        L.check(L.lib().locrec_knn_recommend_range_async(self._h, int(first), int(nq), float(pw), float(cw), int(k)))

    def fetch_recommend(self, nq):
        off = np.zeros(nq + 1, np.int64)
        cap = C.c_int64(0)
        L.check(L.lib().locrec_knn_fetch_recommend(self._h, int(nq), L.ptr(off, C.c_int64), None, None, C.byref(cap)))
        places, est = np.empty(max(1, cap.value), np.int64), np.empty(max(1, cap.value), np.float64)
        cap = C.c_int64(len(places))
        L.check(L.lib().locrec_knn_fetch_recommend(self._h, int(nq), L.ptr(off, C.c_int64), L.ptr(places, C.c_int64),
                                                   L.ptr(est, C.c_double), C.byref(cap)))
        return off, places[:off[-1]], est[:off[-1]]

    def query_shard(self, person_id, pw, cw, k, shard_index, shard_count):
        """Local top-K of candidate shard shard_index of shard_count (include/locrec.h)."""
        cap = int(max(1, min(k, max(1, self.n))))
        ids = np.empty(cap, np.int64)
        sims = np.empty(cap, np.float64)
        cnt = C.c_int64(cap)
        L.check(L.lib().locrec_knn_query_shard(self._h, int(person_id), float(pw), float(cw), int(k),
                                               int(shard_index), int(shard_count),
                                               L.ptr(ids, C.c_int64), L.ptr(sims, C.c_double), C.byref(cnt)))
        m = min(cnt.value, cap)
        return ids[:m], sims[:m]

    def recommend_neighbours(self, neighbour_ids, similarities, capacity=1 << 12):
        """makeRecommendations0 (KnnRecommender.scala:51-70) for a given list of similar persons."""
        nb, sm = L.as_i64(neighbour_ids), L.as_f64(similarities)
        if len(nb) != len(sm):
            raise L.IllegalArgumentException("neighbour columns of different lengths")
        while True:
            places, est = self._out_buffers(capacity)
            capacity = len(places)
            cnt = C.c_int64(capacity)
            L.check(L.lib().locrec_knn_recommend_neighbours(self._h, len(nb), L.ptr(nb, C.c_int64), L.ptr(sm, C.c_double),
                                                            L.ptr(places, C.c_int64), L.ptr(est, C.c_double), C.byref(cnt)))
            if cnt.value <= capacity:
                return places[:cnt.value].copy(), est[:cnt.value].copy()
            capacity = cnt.value

    def query_batch(self, person_ids, pw, cw, k):
        q = L.as_i64(person_ids)
        kk = max(int(k), 0)  # a non-positive K is rejected by the library, with the reference's message
        ids = np.empty((len(q), kk), np.int64)
        sims = np.empty((len(q), kk), np.float64)
        cnt = np.empty(len(q), np.int64)
        L.check(L.lib().locrec_knn_query_batch(self._h, len(q), L.ptr(q, C.c_int64), float(pw), float(cw), int(k),
                                               L.ptr(ids, C.c_int64), L.ptr(sims, C.c_double), L.ptr(cnt, C.c_int64)))
        return ids, sims, cnt

    def all_pairs_topk(self, pw, cw, k):
        kk = max(int(k), 0)
        ids = np.empty((self.n, kk), np.int64)
        sims = np.empty((self.n, kk), np.float64)
        cnt = np.empty(self.n, np.int64)
        L.check(L.lib().locrec_knn_all_pairs_topk(self._h, float(pw), float(cw), int(k),
                                                  L.ptr(ids, C.c_int64), L.ptr(sims, C.c_double), L.ptr(cnt, C.c_int64)))
        return ids, sims, cnt

    # device-resident form (bench.py)
    def topk_range_async(self, first, nq, pw, cw, k):
        L.check(L.lib().locrec_knn_topk_range_async(self._h, int(first), int(nq), float(pw), float(cw), int(k)))

    def row_person_ids(self, first, nq):
        out = np.empty(nq, np.int64)
        L.check(L.lib().locrec_knn_row_person_ids(self._h, int(first), int(nq), L.ptr(out, C.c_int64)))
        return out

    def fetch_topk(self, nq, k):
        ids = np.empty((nq, k), np.int64)
        sims = np.empty((nq, k), np.float64)
        cnt = np.empty(nq, np.int64)
        L.check(L.lib().locrec_knn_fetch_topk(self._h, int(nq), int(k), L.ptr(ids, C.c_int64),
                                              L.ptr(sims, C.c_double), L.ptr(cnt, C.c_int64)))
        return ids, sims, cnt

    def replayed_intervals(self):
        """Flush intervals of the batched scan that overran a survivor queue and were replayed with
        synchronous insertion inside the kernel (statistics; results are never affected)."""
        v = C.c_int64()
        L.check(L.lib().locrec_knn_replayed_intervals(self._h, C.byref(v)))
        return v.value

    def set_stream(self, hip_stream):
        L.check(L.lib().locrec_knn_set_stream(self._h, C.c_void_p(hip_stream)))

    def synchronize(self):
        L.check(L.lib().locrec_knn_synchronize(self._h))

    def profile_enable(self, on=True):
        L.check(L.lib().locrec_knn_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        ms, n = C.c_double(), C.c_int64()
        L.check(L.lib().locrec_knn_profile_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


class KnnRecommender:
    """new KnnRecommender(placeRatingVectors, categoryRatingVectors, placeRatings,
    placeWeight, categoryWeight, kNearest).makeRecommendations(personId)"""

    def __init__(self, placeRatingVectors, categoryRatingVectors, placeRatings,
                 placeWeight, categoryWeight, kNearest):
        # require()s, KnnRecommender.scala:17-20 (same messages; checked again in the library)
        if not (placeWeight > 0 and placeWeight < 1.0):
            raise L.IllegalArgumentException(f"requirement failed: Place weight must be in the interval (0; 1): {placeWeight}")
        if not (categoryWeight > 0 and categoryWeight < 1.0):
            raise L.IllegalArgumentException(f"requirement failed: Category weight must be in the interval (0; 1): {categoryWeight}")
        if not (placeWeight + categoryWeight == 1.0):
            raise L.IllegalArgumentException(
                f"requirement failed: Sum of weights must be 1.0: place: {placeWeight}, category: {categoryWeight}")
        if not (kNearest > 0):
            raise L.IllegalArgumentException("requirement failed: K nearest must be positive")
        self.placeWeight, self.categoryWeight, self.kNearest = float(placeWeight), float(categoryWeight), int(kNearest)
        _cache.require_gpu_backend("KnnRecommender")
        # what the three frames ARE - not the weights or K, which every request passes to the library
        key = "|".join((_cache.frame_key(placeRatingVectors, ("person_id", "rating_vector")),
                        _cache.frame_key(categoryRatingVectors, ("person_id", "rating_vector")),
                        _cache.frame_key(placeRatings, ("person_id", "place_id", "rating"))))
        self._index = KnnIndex.through_cache(
            key, lambda: self._collect(placeRatingVectors, categoryRatingVectors, placeRatings))

    @staticmethod
    def _collect(placeRatingVectors, categoryRatingVectors, placeRatings):
        """The three frames collected to CSR and built into a device index (a cache miss only)."""
        pids = sorted(set(int(p) for p in placeRatingVectors["person_id"]) |
                      set(int(p) for p in categoryRatingVectors["person_id"]) |
                      set(int(p) for p in placeRatings["person_id"]))
        prp, pidx, pval, pdim = _vectors_to_csr(pids, placeRatingVectors)
        crp, cidx, cval, cdim = _vectors_to_csr(pids, categoryRatingVectors)
        pos = {p: i for i, p in enumerate(pids)}
        rows = np.fromiter((pos[int(p)] for p in placeRatings["person_id"]), dtype=np.int64,
                           count=len(placeRatings["person_id"]))
        order = np.argsort(rows, kind="stable")
        rrp = np.zeros(len(pids) + 1, np.int64)
        np.add.at(rrp, rows + 1, 1)
        rrp = np.cumsum(rrp)
        rplace = L.as_i64(np.asarray(placeRatings["place_id"])[order])
        rrating = L.as_i64(np.asarray(placeRatings["rating"])[order])
        return KnnIndex(pids, prp, pidx, pval, pdim, crp, cidx, cval, cdim, rrp, rplace, rrating)

    def close(self):
        """Drops this object's reference; the device index stays cached for the next constructor."""
        self._index.close()

    def findSimilarPersons(self, personId):
        """(person_id, similarity) of the kNearest most similar persons (KnnRecommender.scala:27-49)."""
        import pandas as pd
        with self._index.lock:
            ids, sims = self._index.query(personId, self.placeWeight, self.categoryWeight, self.kNearest)
        return pd.DataFrame({"person_id": ids, "similarity": sims})

    def makeRecommendationsBatch(self, personIds):
        """Additive: makeRecommendations for many persons in one device pass ->
        (person_id, place_id, estimated_rating)."""
        import pandas as pd
        ids = L.as_i64(personIds)
        with self._index.lock:
            off, places, est = self._index.recommend_batch(ids, self.placeWeight, self.categoryWeight, self.kNearest)
        return pd.DataFrame({"person_id": np.repeat(ids, np.diff(off)), "place_id": places, "estimated_rating": est})

    def makeRecommendations(self, personId):
        """(place_id, estimated_rating) (KnnRecommender.scala:22-25,51-70)."""
        import pandas as pd
        with self._index.lock:
            places, est = self._index.recommend(personId, self.placeWeight, self.categoryWeight, self.kNearest)
        return pd.DataFrame({"place_id": places, "estimated_rating": est})
