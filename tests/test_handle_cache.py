"""The drop-in behind the UNCHANGED mains (VERDICT r02 "What's missing" 1, "Next round" 3): the reference builds a
new recommender from freshly read DataFrames for every stdin request (KnnRecommenderMain.scala:53-67,
StochasticRecommenderMain.scala:53-62) and closes nothing, so the device handle has to live in a process-wide
cache (include/locrec.h "Handle cache", csrc/cache.hip) keyed by what the frames ARE.

CPU part: the cache's bookkeeping entry points, the keys and the LOCREC_BACKEND switch (no device needed).
GPU part: 100 constructors on the same frames = one create, flat allocation count and device bytes; LRU eviction
by bytes; references keep an evicted handle alive; results identical to the oracle through the cache."""
import ctypes as C
import gc
import os

import numpy as np
import pandas as pd
import pytest


@pytest.fixture
def cache(pkg):
    from locations_recommender_amd import _cache
    return _cache


def test_cache_entry_points_without_a_device(pkg, cache):
    from locations_recommender_amd import _lib as L
    lib = pkg.lib()
    h = C.c_void_p(1)
    assert lib.locrec_cache_acquire(L.CACHE_KNN, b"no-such-key", C.byref(h)) == 0 and not h.value
    assert lib.locrec_cache_acquire(7, b"k", C.byref(h)) == L.E_INVALID_ARG
    assert lib.locrec_cache_acquire(L.CACHE_SG, None, C.byref(h)) == L.E_INVALID_ARG
    assert lib.locrec_cache_publish(L.CACHE_SG, b"k", None, 0, C.byref(h)) == L.E_INVALID_ARG
    assert lib.locrec_cache_release(L.CACHE_KNN, None) == 0          # closing twice is harmless
    s = L.cache_stats()
    assert s["misses"] >= 1 and s["entries"] >= 0
    assert L.device_bytes_in_use() >= 0
    cache.set_limits(-1, -1)


def test_frame_keys(pkg, cache, tmp_path):
    a = pd.DataFrame({"source_id": [1, 2], "target_id": [2, 1], "balanced_weight": [1.0, 1.0]})
    b = pd.DataFrame({"source_id": [1, 2], "target_id": [2, 1], "balanced_weight": [1.0, 1.0]})
    c = pd.DataFrame({"source_id": [1, 2], "target_id": [2, 1], "balanced_weight": [1.0, 0.5]})
    cols = ("source_id", "target_id", "balanced_weight")
    ka = cache.frame_key(a, cols)
    assert ka == cache.frame_key(a, cols) == cache.frame_key(b, cols), "equal content must share a handle"
    assert ka != cache.frame_key(c, cols)
    v1 = pd.DataFrame({"person_id": [1], "rating_vector": [pkg.SparseVector(5, [1, 3], [1.0, 2.0])]})
    v2 = pd.DataFrame({"person_id": [1], "rating_vector": [pkg.SparseVector(5, [1, 3], [1.0, 3.0])]})
    assert cache.frame_key(v1, ("person_id", "rating_vector")) != cache.frame_key(v2, ("person_id", "rating_vector"))
    # files: name + size + modification time (Spark: df.inputFiles)
    p = tmp_path / "f.parquet"
    p.write_bytes(b"x" * 10)
    k1 = cache.files_key([str(p)])
    assert k1 == cache.files_key([str(p)])
    os.utime(p, ns=(1, 1))
    assert cache.files_key([str(p)]) != k1, "a rewritten file must miss"
    d = tmp_path / "dir"
    d.mkdir()
    (d / "part-0").write_bytes(b"a")
    (d / "_SUCCESS").write_text("")
    kd = cache.files_key([str(d)])
    (d / "part-1").write_bytes(b"b")
    assert cache.files_key([str(d)]) != kd
    a.attrs["inputFiles"] = [str(p)]
    assert cache.frame_key(a, cols).startswith("files:")


def test_backend_switch(pkg, monkeypatch):
    """LOCREC_BACKEND=spark selects the reference's Spark implementation,
    which only the Scala side has: the Python mirror refuses instead of computing on the CPU."""
    edges = pd.DataFrame({"source_id": [1], "target_id": [2], "balanced_weight": [1.0]})
    monkeypatch.setenv("LOCREC_BACKEND", "spark")
    with pytest.raises(pkg.LocrecRuntimeError, match="LOCREC_BACKEND=spark"):
        pkg.StochasticRecommender(edges, 0.1, 10)
    monkeypatch.setenv("LOCREC_BACKEND", "cuda")
    with pytest.raises(pkg.IllegalArgumentException, match="LOCREC_BACKEND"):
        pkg.StochasticRecommender(edges, 0.1, 10)
    monkeypatch.setenv("LOCREC_BACKEND", "gpu")
    from locations_recommender_amd import _cache
    assert _cache.backend() == "gpu"


# ---------------------------------------------------------------------------------------------------------
def knn_frames(pkg, d):
    def vecs(rowptr, idx, val, dim):
        return pd.DataFrame({"person_id": d["person_ids"],
                             "rating_vector": [pkg.SparseVector(dim, idx[rowptr[i]:rowptr[i + 1]], val[rowptr[i]:rowptr[i + 1]])
                                               for i in range(len(d["person_ids"]))]})
    rows = np.repeat(np.arange(len(d["person_ids"])), np.diff(d["p_rowptr"]))
    ratings = pd.DataFrame({"person_id": d["person_ids"][rows], "place_id": d["p_idx"].astype(np.int64),
                            "rating": d["p_val"].astype(np.int64)})
    return (vecs(d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"]), vecs(d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"]),
            ratings)


@pytest.mark.gpu
def test_hundred_constructors_one_create(pkg, oracle, cache):
    """KnnRecommenderMain's loop: 100 x `new KnnRecommender(frames...).makeRecommendations(id)` on the same frames
    = ONE locrec_knn_create; allocation count and device bytes flat after the first; weights and K may change
    from request to request (they are not part of the key); nothing is ever closed by the caller."""
    from locations_recommender_amd import _lib as L, synth
    cache.clear()
    d = synth.knn_dataset(3_000, 400, seed=41)
    d["r_rowptr"], d["r_place"], d["r_rating"] = d["p_rowptr"], d["p_idx"].astype(np.int64), d["p_val"].astype(np.int64)
    pv, cv, pr = knn_frames(pkg, d)
    s0 = L.cache_stats()
    allocs = bytes_ = None
    for i in range(100):
        pid = int(d["person_ids"][(i * 37) % 3_000])
        pw = 0.5 if i % 2 == 0 else 0.25
        k = 50 if i % 3 else 20
        rec = pkg.KnnRecommender(pv, cv, pr, pw, 1.0 - pw, k)          # never closed, as in the reference's main
        got = rec.makeRecommendations(pid)
        if i in (0, 1, 2, 50, 99):
            oplaces, oest = oracle.knn_recommend(d, pid, pw, 1.0 - pw, k)
            assert np.array_equal(got["place_id"].to_numpy(), oplaces)
            np.testing.assert_allclose(got["estimated_rating"].to_numpy(), oest, rtol=1e-6, atol=0)
        if i == 2:       # the work buffers of both K have settled
            allocs, bytes_ = L.device_allocations(), L.device_bytes_in_use()
    del rec
    gc.collect()
    s1 = L.cache_stats()
    assert s1["misses"] - s0["misses"] == 1 and s1["hits"] - s0["hits"] == 99, (s0, s1)
    assert s1["entries"] - s0["entries"] == 1
    assert L.device_allocations() == allocs, "a warm request through the constructor allocated device memory"
    assert L.device_bytes_in_use() == bytes_
    # an equal-content copy of the frames hits too (content key), a changed frame misses
    pv2, cv2, pr2 = knn_frames(pkg, d)
    pkg.KnnRecommender(pv2, cv2, pr2, 0.5, 0.5, 10).makeRecommendations(int(d["person_ids"][0]))
    assert L.cache_stats()["misses"] == s1["misses"]
    pr3 = pr2.copy()
    pr3.loc[0, "rating"] += 1
    pkg.KnnRecommender(pv2, cv2, pr3, 0.5, 0.5, 10)
    assert L.cache_stats()["misses"] == s1["misses"] + 1
    cache.clear()
    gc.collect()


@pytest.mark.gpu
def test_sg_constructor_through_the_cache_and_lru(pkg, oracle, cache):
    """StochasticRecommenderMain's loop over several region pairs with a byte budget that holds two graphs:
    least-recently-used unreferenced graphs are destroyed, a referenced one survives its eviction until released."""
    from locations_recommender_amd import _lib as L, synth
    cache.clear()
    gc.collect()
    graphs = [synth.sg_dataset(n_persons=4_000, n_places=400, seed=60 + i) for i in range(4)]
    frames = [pd.DataFrame({"source_id": g["source_id"], "target_id": g["target_id"], "balanced_weight": g["balanced_weight"]})
              for g in graphs]
    base = L.device_bytes_in_use()
    r0 = pkg.StochasticRecommender(frames[0], 0.01, 20, quiet=True)
    one = L.device_bytes_in_use() - base
    assert one > 0
    r0.close()
    assert L.device_bytes_in_use() - base == one, "close() must only drop a reference"
    cache.set_limits(base + int(2.5 * one), -1)
    ev0 = L.cache_stats()["evictions"]
    for rounds in range(3):
        for i, (g, f) in enumerate(zip(graphs, frames)):
            v = int(g["first_person"]) + rounds
            rec = pkg.StochasticRecommender(f, 0.01, 20, quiet=True)
            got = rec.makeRecommendations(v)
            oi, op, _, _ = oracle.sg_recommend(g["source_id"], g["target_id"], g["balanced_weight"], v, 0.15, 0.01, 20)
            assert np.array_equal(got["id"].to_numpy(), oi)
            np.testing.assert_allclose(got["probability"].to_numpy(), op, rtol=1e-6, atol=0)
            del rec
            gc.collect()
            assert L.device_bytes_in_use() - base <= 2.5 * one + 1, "the byte budget does not hold"
    st = L.cache_stats()
    assert st["evictions"] - ev0 >= 8 and st["entries"] <= 2
    # a referenced handle outlives clear(): still usable, destroyed by its last release
    keep = pkg.StochasticRecommender(frames[1], 0.01, 20, quiet=True)
    cache.clear()
    assert len(keep.makeRecommendations(int(graphs[1]["first_person"]))) > 0
    held = L.device_bytes_in_use()
    keep.close()
    assert L.device_bytes_in_use() < held
    cache.set_limits(64 << 30, 64)
    cache.clear()


@pytest.mark.gpu
def test_main_loop_from_files(pkg, oracle, cache, tmp_path):
    """mains.knn_make_recommendations / sg_make_recommendations = the unchanged mains' per-request bodies
    (KnnRecommenderMain.scala:53-67): the files are read once per region pair; rewriting a file misses."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    from locations_recommender_amd import _lib as L, mains, synth
    from test_mains import write_vectors
    cache.clear()
    d = synth.knn_dataset(1_500, 300, seed=6)
    base = tmp_path
    write_vectors(base / "place_rating_vectors_region0_region2", d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"])
    write_vectors(base / "category_rating_vectors_region0_region2", d["person_ids"], d["c_rowptr"], d["c_idx"], d["c_val"],
                  d["c_dim"], shuffle_seed=1)
    rows = np.repeat(np.arange(1_500), np.diff(d["p_rowptr"]))
    d["r_rowptr"], d["r_place"], d["r_rating"] = d["p_rowptr"], d["p_idx"].astype(np.int64), d["p_val"].astype(np.int64)
    pq.write_table(pa.table({"person_id": d["person_ids"][rows], "place_id": d["r_place"], "rating": d["r_rating"]}),
                   base / "place_ratings_region0_region2")
    m0 = L.cache_stats()["misses"]
    for i in range(20):
        pid = int(d["person_ids"][i * 50])
        places, est = mains.knn_make_recommendations(str(base), [2, 0], pid, 0.5, 0.5, 50)
        oplaces, oest = oracle.knn_recommend(d, pid, 0.5, 0.5, 50)
        assert np.array_equal(places, oplaces)
        np.testing.assert_allclose(est, oest, rtol=1e-6, atol=0)
    assert L.cache_stats()["misses"] - m0 == 1
    with pytest.raises(pkg.IllegalArgumentException, match="No such person: 77"):
        mains.knn_make_recommendations(str(base), [0, 2], 77, 0.5, 0.5, 50)
    assert L.cache_stats()["misses"] - m0 == 1
    os.utime(base / "place_ratings_region0_region2", ns=(5, 5))      # "the builder ran again"
    mains.knn_make_recommendations(str(base), [0, 2], int(d["person_ids"][3]), 0.5, 0.5, 50)
    assert L.cache_stats()["misses"] - m0 == 2
    g = synth.sg_dataset(n_persons=1_500, n_places=300, seed=8)
    pq.write_table(pa.table({"source_id": g["source_id"], "target_id": g["target_id"],
                             "balanced_weight": g["balanced_weight"]}), base / "stochastic_graph_region0_region2")
    m1 = L.cache_stats()["misses"]
    for i in range(10):
        v = int(g["first_person"]) + i
        ids, probs, it, conv = mains.sg_make_recommendations(str(base), [0, 2], v, 0.01, 20)
        oi, op, oit, oconv = oracle.sg_recommend(g["source_id"], g["target_id"], g["balanced_weight"], v, 0.15, 0.01, 20)
        assert np.array_equal(ids, oi) and (it, conv) == (oit, oconv)
        np.testing.assert_allclose(probs, op, rtol=1e-6, atol=0)
    assert L.cache_stats()["misses"] - m1 == 1
    cache.clear()
