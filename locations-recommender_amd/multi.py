"""Several GPUs of one node from ONE process (include/locrec.h "Several devices in one process", csrc/multi.hip):
what the Scala host - one JVM - binds through JNI; bench.py's one-process-per-GPU forms live in shard.py.

    KnnReplicas   every device holds the candidate set (set up by a tile-wise block all-gather over xGMI), the
                  queries of a batch are sharded over the devices
    SgSharded     one stochastic graph with its rows sharded over the devices, sigma exchanged per sweep by a
                  kernel that reads the peers' buffers (all-reduce in device order, or all-gather of owned rows)
A device may be listed more than once: logical shards on one GPU (how a one-GPU box rehearses the path)."""
import ctypes as C

import numpy as np

from . import _lib as L


def set_devices(device_ids):
    """locrec_set_devices: the default device list of the multi-device entry points ([] forgets it)."""
    ids = L.as_i32(device_ids)
    L.check(L.lib().locrec_set_devices(len(ids), L.ptr(ids, C.c_int32) if len(ids) else None))


def _dev_args(device_ids):
    if device_ids is None:
        return 0, None, None
    ids = L.as_i32(device_ids)
    return len(ids), L.ptr(ids, C.c_int32), ids


class KnnReplicas:
    def __init__(self, device_ids, person_ids, p_rowptr, p_idx, p_val, p_dim, c_rowptr, c_idx, c_val, c_dim,
                 r_rowptr=None, r_place=None, r_rating=None):
        self._h = C.c_void_p()
        nd, dp, _keep = _dev_args(device_ids)
        ids = L.as_i64(person_ids)
        a = [L.as_i64(p_rowptr), L.as_i32(p_idx), L.as_f64(p_val), L.as_i64(c_rowptr), L.as_i32(c_idx), L.as_f64(c_val)]
        r = [None, None, None] if r_rowptr is None else [L.as_i64(r_rowptr), L.as_i64(r_place), L.as_i64(r_rating)]
        L.check(L.lib().locrec_knn_replicas_create(
            nd, dp, len(ids), L.ptr(ids, C.c_int64), L.ptr(a[0], C.c_int64), L.ptr(a[1], C.c_int32), L.ptr(a[2], C.c_double),
            int(p_dim), L.ptr(a[3], C.c_int64), L.ptr(a[4], C.c_int32), L.ptr(a[5], C.c_double), int(c_dim),
            L.ptr(r[0], C.c_int64), L.ptr(r[1], C.c_int64), L.ptr(r[2], C.c_int64), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            L.lib().locrec_knn_replicas_destroy(self._h)
            self._h = None

    __del__ = close

    def devices(self):
        n = C.c_int32()
        L.check(L.lib().locrec_knn_replicas_info(self._h, C.byref(n), None))
        return n.value

    def recommend_batch(self, person_ids, pw, cw, k):
        q = L.as_i64(person_ids)
        off = np.zeros(len(q) + 1, np.int64)
        cap = C.c_int64(0)
        places, est = np.empty(0, np.int64), np.empty(0, np.float64)
        for _ in range(2):
            L.check(L.lib().locrec_knn_replicas_recommend_batch(self._h, len(q), L.ptr(q, C.c_int64), float(pw), float(cw), int(k),
                                                                L.ptr(off, C.c_int64), L.ptr(places, C.c_int64),
                                                                L.ptr(est, C.c_double), C.byref(cap)))
            if cap.value <= len(places):
                break
            places, est = np.empty(cap.value, np.int64), np.empty(cap.value, np.float64)
            cap = C.c_int64(len(places))
        return off, places[:off[-1]], est[:off[-1]]

    def query_batch(self, person_ids, pw, cw, k):
        q = L.as_i64(person_ids)
        kk = max(int(k), 0)
        ids, sims, cnt = np.empty((len(q), kk), np.int64), np.empty((len(q), kk), np.float64), np.empty(len(q), np.int64)
        L.check(L.lib().locrec_knn_replicas_query_batch(self._h, len(q), L.ptr(q, C.c_int64), float(pw), float(cw), int(k),
                                                        L.ptr(ids, C.c_int64), L.ptr(sims, C.c_double), L.ptr(cnt, C.c_int64)))
        return ids, sims, cnt


class SgSharded:
    def __init__(self, device_ids, source_ids, target_ids, balanced_weights, by_target=False):
        self._h = C.c_void_p()
        nd, dp, _keep = _dev_args(device_ids)
        s, t, w = L.as_i64(source_ids), L.as_i64(target_ids), L.as_f64(balanced_weights)
        if not (len(s) == len(t) == len(w)):
            raise L.IllegalArgumentException("edge columns of different lengths")
        L.check(L.lib().locrec_sg_sharded_create(nd, dp, len(s), L.ptr(s, C.c_int64), L.ptr(t, C.c_int64), L.ptr(w, C.c_double),
                                                 1 if by_target else 0, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            L.lib().locrec_sg_sharded_destroy(self._h)
            self._h = None

    __del__ = close

    def info(self):
        d, p, e, v = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
        L.check(L.lib().locrec_sg_sharded_info(self._h, C.byref(d), C.byref(p), C.byref(e), C.byref(v)))
        return {"devices": d.value, "peer_access": bool(p.value), "exchanged_entries": e.value, "vertices": v.value}

    def iterate_async(self, vertex_id, alpha, epsilon, max_iterations):
        L.check(L.lib().locrec_sg_sharded_iterate_async(self._h, int(vertex_id), float(alpha), float(epsilon), int(max_iterations)))

    def sweeps_async(self, vertex_id, alpha, sweeps):
        L.check(L.lib().locrec_sg_sharded_sweeps_async(self._h, int(vertex_id), float(alpha), int(sweeps)))

    def fetch(self, which=0):
        cap = max(1, self.info()["vertices"])
        ids, probs = np.empty(cap, np.int64), np.empty(cap, np.float64)
        cnt, it, conv = C.c_int64(cap), C.c_int64(), C.c_int32()
        L.check(L.lib().locrec_sg_sharded_fetch(self._h, int(which), L.ptr(ids, C.c_int64), L.ptr(probs, C.c_double), C.byref(cnt),
                                                C.byref(it), C.byref(conv)))
        return ids[:cnt.value], probs[:cnt.value], it.value, bool(conv.value)

    def recommend(self, vertex_id, alpha, epsilon, max_iterations):
        self.iterate_async(vertex_id, alpha, epsilon, max_iterations)
        return self.fetch()
