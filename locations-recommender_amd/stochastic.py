"""Host-side mirror of stochastic/StochasticRecommender.scala:28-34,66-71 over the C ABI."""
import ctypes as C

import numpy as np

from . import _cache
from . import _lib as L

ALPHA = 0.15  # StochasticRecommender.scala:38


class SgGraph:
    """Owner of a locrec_sg_graph handle."""

    def __init__(self, source_ids, target_ids, balanced_weights, shard_index=0, shard_count=1, by_target=False):
        """shard_count > 1: this handle keeps the edges of the source vertices ("rows of P") that
        fall into shard shard_index of the SAME global edge list; iterate it with ShardedSgRecommender."""
        self._h = C.c_void_p()
        s, t, w = L.as_i64(source_ids), L.as_i64(target_ids), L.as_f64(balanced_weights)
        if not (len(s) == len(t) == len(w)):
            raise L.IllegalArgumentException("edge columns of different lengths")
        if shard_count == 1 and shard_index == 0:
            L.check(L.lib().locrec_sg_create(len(s), L.ptr(s, C.c_int64), L.ptr(t, C.c_int64),
                                             L.ptr(w, C.c_double), C.byref(self._h)))
        else:
            create = L.lib().locrec_sg_create_target_sharded if by_target else L.lib().locrec_sg_create_sharded
            L.check(create(len(s), L.ptr(s, C.c_int64), L.ptr(t, C.c_int64), L.ptr(w, C.c_double),
                           int(shard_index), int(shard_count), C.byref(self._h)))

    @classmethod
    def through_cache(cls, key, build):
        """The process-wide cached graph for `key` (include/locrec.h, "Handle cache"); build() -> SgGraph runs
        on a miss only.  close() / garbage collection drop a reference, the device graph stays."""
        h = _cache.acquire(L.CACHE_SG, key)
        if h is None:
            before = L.device_bytes_in_use()
            built = build()
            h = _cache.publish(L.CACHE_SG, key, built._h, L.device_bytes_in_use() - before)
            built._h = None
        self = cls.__new__(cls)
        self._h, self._cached = h, True
        self.lock = _cache.handle_lock(h)
        return self

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_cached", False):
                _cache.release(L.CACHE_SG, self._h)
            else:
                L.lib().locrec_sg_destroy(self._h)
            self._h = None

    __del__ = close

    def info(self):
        v, e, b = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.lib().locrec_sg_info(self._h, C.byref(v), C.byref(e), C.byref(b)))
        db = C.c_int64()
        L.check(L.lib().locrec_sg_device_bytes(self._h, C.byref(db)))
        nd = C.c_int32()
        L.check(L.lib().locrec_sg_weight_dictionary(self._h, C.byref(nd)))
        return {"vertices": v.value, "edges": e.value, "sweep_bytes": b.value, "device_sweep_bytes": db.value,
                "weight_dictionary": nd.value}

    def recommend(self, vertex_id, alpha, epsilon, max_iterations):
        self.iterate_async(vertex_id, alpha, epsilon, max_iterations)
        return self.fetch()

    def iterate_async(self, vertex_id, alpha, epsilon, max_iterations):
        L.check(L.lib().locrec_sg_iterate_async(self._h, int(vertex_id), float(alpha), float(epsilon),
                                                int(max_iterations)))

    def sweeps_async(self, vertex_id, alpha, sweeps):
        L.check(L.lib().locrec_sg_sweeps_async(self._h, int(vertex_id), float(alpha), int(sweeps)))

    def fetch(self):
        cap = max(1, self.info()["vertices"])
        ids = np.empty(cap, np.int64)
        probs = np.empty(cap, np.float64)
        cnt, it, conv = C.c_int64(cap), C.c_int64(), C.c_int32()
        L.check(L.lib().locrec_sg_fetch(self._h, L.ptr(ids, C.c_int64), L.ptr(probs, C.c_double), C.byref(cnt),
                                        C.byref(it), C.byref(conv)))
        return ids[:cnt.value], probs[:cnt.value], it.value, bool(conv.value)

    # ---- row-sharded iteration (include/locrec.h, "Row-sharded form") ----
    def live_count(self):
        n = C.c_int64()
        L.check(L.lib().locrec_sg_live_count(self._h, C.byref(n)))
        return n.value

    def shard_begin(self, vertex_id):
        L.check(L.lib().locrec_sg_shard_begin(self._h, int(vertex_id)))

    def shard_sigma(self, sigma_device_ptr):
        L.check(L.lib().locrec_sg_shard_sigma(self._h, C.c_void_p(sigma_device_ptr)))

    def shard_apply(self, sigma_device_ptr, alpha):
        L.check(L.lib().locrec_sg_shard_apply(self._h, C.c_void_p(sigma_device_ptr), float(alpha)))

    def shard_d2(self):
        d2 = C.c_double()
        L.check(L.lib().locrec_sg_shard_d2(self._h, C.byref(d2)))
        return d2.value

    def shard_finish(self, iterations, converged):
        L.check(L.lib().locrec_sg_shard_finish(self._h, int(iterations), 1 if converged else 0))

    def set_stream(self, hip_stream):
        L.check(L.lib().locrec_sg_set_stream(self._h, C.c_void_p(hip_stream)))

    def synchronize(self):
        L.check(L.lib().locrec_sg_synchronize(self._h))

    def profile_enable(self, on=True):
        L.check(L.lib().locrec_sg_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        ms, n = C.c_double(), C.c_int64()
        L.check(L.lib().locrec_sg_profile_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


class SgGroup:
    """Independent graphs iterated together (locrec_sg_group_*): one sweep and one combine launch per
    round for all of them.  The graphs stay owned by the caller and are read with SgGraph.fetch()."""

    def __init__(self, graphs):
        self.graphs = list(graphs)
        arr = (C.c_void_p * len(self.graphs))(*[g._h for g in self.graphs])
        self._h = C.c_void_p()
        L.check(L.lib().locrec_sg_group_create(arr, len(self.graphs), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            L.lib().locrec_sg_group_destroy(self._h)
            self._h = None

    __del__ = close

    def sweeps_async(self, vertex_ids, alpha, sweeps):
        v = L.as_i64(vertex_ids)
        assert len(v) == len(self.graphs)
        L.check(L.lib().locrec_sg_group_sweeps_async(self._h, L.ptr(v, C.c_int64), float(alpha), int(sweeps)))

    def iterate_async(self, vertex_ids, alpha, epsilon, max_iterations):
        """makeRecommendations' iteration for every graph (each stops at its own isConverged)."""
        v = L.as_i64(vertex_ids)
        assert len(v) == len(self.graphs)
        L.check(L.lib().locrec_sg_group_iterate_async(self._h, L.ptr(v, C.c_int64), float(alpha), float(epsilon),
                                                      int(max_iterations)))

    def synchronize(self):
        L.check(L.lib().locrec_sg_group_synchronize(self._h))


class StochasticRecommender:
    """new StochasticRecommender(stochasticEdges, epsilon, maxIterations).makeRecommendations(vertexId)

    stochasticEdges: frame with columns source_id, target_id, balanced_weight
    (StochasticGraphBuilder.scala:12-16); ids may be ints of any width."""

    def __init__(self, stochasticEdges, epsilon, maxIterations, quiet=False):
        if not (epsilon >= 0):
            raise L.IllegalArgumentException("requirement failed: epsilon must be non-negative")
        if not (maxIterations >= 0):
            raise L.IllegalArgumentException("requirement failed: max iterations number must be non-negative")
        self.epsilon, self.maxIterations, self.quiet = float(epsilon), int(maxIterations), quiet
        _cache.require_gpu_backend("StochasticRecommender")
        # the edge list alone identifies the graph; epsilon and maxIterations are per-request arguments
        key = _cache.frame_key(stochasticEdges, ("source_id", "target_id", "balanced_weight"))
        self._graph = SgGraph.through_cache(key, lambda: SgGraph(
            np.asarray(stochasticEdges["source_id"]), np.asarray(stochasticEdges["target_id"]),
            np.asarray(stochasticEdges["balanced_weight"])))

    def close(self):
        """Drops this object's reference; the device graph stays cached for the next constructor."""
        self._graph.close()

    def makeRecommendations(self, vertexId):
        import pandas as pd
        with self._graph.lock:
            ids, probs, iterations, converged = self._graph.recommend(vertexId, ALPHA, self.epsilon, self.maxIterations)
        if not self.quiet:  # the two println()s of step(), StochasticRecommender.scala:94,100
            if converged:
                print(f"Converged in {iterations} iterations")
            else:
                print(f"Number of iterations {iterations} reached the maximum {self.maxIterations}")
        return pd.DataFrame({"id": ids, "probability": probs})
