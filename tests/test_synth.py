"""The synthetic generator is counter-based: shards and re-runs are bit-identical."""
import numpy as np


def test_knn_dataset_is_deterministic_and_shardable(pkg):
    from locations_recommender_amd import synth
    full = synth.knn_dataset(4000, 1000, seed=0x5EED0002)
    again = synth.knn_dataset(4000, 1000, seed=0x5EED0002)
    for k in ("p_rowptr", "p_idx", "p_val", "c_rowptr", "c_idx", "c_val", "person_ids"):
        assert np.array_equal(full[k], again[k])
    shard = synth.knn_dataset(4000, 1000, seed=0x5EED0002, first_row=1000, rows=500)
    lo, hi = full["p_rowptr"][1000], full["p_rowptr"][1500]
    assert np.array_equal(shard["p_idx"], full["p_idx"][lo:hi])
    assert np.array_equal(shard["p_val"], full["p_val"][lo:hi])
    assert np.array_equal(shard["person_ids"], full["person_ids"][1000:1500])
    # SparseVector invariants: strictly ascending indices inside a row, integer counts >= 1
    rp, idx = full["p_rowptr"], full["p_idx"]
    inner = np.ones(len(idx), bool)
    inner[rp[:-1][rp[:-1] < len(idx)]] = False
    assert np.all(np.diff(idx.astype(np.int64))[inner[1:]] > 0)
    assert full["p_val"].min() >= 1 and np.all(full["p_val"] == np.floor(full["p_val"]))
    assert np.all(np.diff(rp) >= 1) and np.all(np.diff(full["c_rowptr"]) >= 1)
    nnz = np.diff(rp)
    assert 15 < nnz.mean() < 30 and nnz.max() <= 100


def test_sg_dataset_is_stochastic(pkg):
    from locations_recommender_amd import synth
    g = synth.sg_dataset(n_persons=3000, n_places=300, seed=9)
    src, w = g["source_id"], g["balanced_weight"]
    order = np.argsort(src, kind="stable")
    s, ws = src[order], w[order]
    starts = np.flatnonzero(np.r_[True, s[1:] != s[:-1]])
    sums = np.add.reduceat(ws, starts)
    np.testing.assert_allclose(sums, 1.0, rtol=0, atol=1e-12)   # StochasticGraphBuilderTest.scala:52-60
    assert g["target_id"].max() < g["first_person"]              # no edge targets a person (H5)


def test_parallel_generation_is_chunking_independent(pkg):
    from locations_recommender_amd import synth
    whole = synth.knn_dataset(5000, 800, seed=99)
    par = synth.knn_dataset_parallel(5000, 800, seed=99, workers=2, chunk=1300)
    for k, v in whole.items():
        assert np.array_equal(np.asarray(v), np.asarray(par[k])), k
