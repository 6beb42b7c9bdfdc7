"""Queries that are too long for an LDS panel: RatingsBuilder's rank()-with-ties
top-N keeps a tie whole (RatingsBuilder.scala:42-47), so a person who visited thousands of places
once each keeps them all.  Such a query is served by the dense-query scan (knn_scan_dense) and, in
batches, by the sort path into its slot of the result arrays - same results as the oracle, no
length limit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def dataset_with_long_rows(seed, n=4300, p_dim=9000, integer=True, long_rows=((7, 4200), (1234, 9000), (4000, 5000))):
    rng = np.random.default_rng(seed)
    lens = dict(long_rows)
    prp, pidx, pval, crp, cidx, cval = [0], [], [], [0], [], []
    w = 1.0 / np.arange(1, p_dim + 1)
    w /= w.sum()
    for i in range(n):
        kp = lens.get(i, int(rng.integers(1, 30)))
        ip = np.sort(rng.choice(p_dim, size=kp, replace=False, p=None if kp > 100 else w))
        vp = (np.ones(kp) if kp > 100 else rng.integers(1, 6, size=kp)).astype(np.float64)   # the long rows: ties at 1
        if i == 1234:
            vp = rng.integers(1, 3, size=kp).astype(np.float64)
        kc = int(rng.integers(1, 7))
        ic = np.sort(rng.choice(20, size=kc, replace=False))
        vc = rng.integers(1, 20, size=kc).astype(np.float64)
        if not integer:
            vp = vp + rng.random(kp) * 0.5
            vc = vc * 0.37
        pidx.append(ip), pval.append(vp), prp.append(prp[-1] + kp)
        cidx.append(ic), cval.append(vc), crp.append(crp[-1] + kc)
    ids = rng.permutation(n).astype(np.int64) * 5 + 2040
    d = {"person_ids": ids, "p_rowptr": np.array(prp, np.int64), "p_idx": np.concatenate(pidx).astype(np.int32),
         "p_val": np.concatenate(pval), "p_dim": p_dim, "c_rowptr": np.array(crp, np.int64),
         "c_idx": np.concatenate(cidx).astype(np.int32), "c_val": np.concatenate(cval), "c_dim": 20}
    if integer:
        d["r_rowptr"], d["r_place"] = d["p_rowptr"], d["p_idx"].astype(np.int64) + 40
        d["r_rating"] = d["p_val"].astype(np.int64)
    return d, [r for r, _ in long_rows]


def make_index(pkg, d):
    extra = (d["r_rowptr"], d["r_place"], d["r_rating"]) if "r_rowptr" in d else ()
    return pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                        d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"], *extra)


@pytest.mark.parametrize("integer", [True, False])
def test_long_query_single_requests(pkg, oracle, integer):
    d, long_rows = dataset_with_long_rows(3, integer=integer)
    ix = make_index(pkg, d)
    for row in long_rows + [0, 99]:
        pid = int(d["person_ids"][row])
        for k in (1, 50, 700):
            ids, sims = ix.query(pid, 0.5, 0.5, k)
            oids, osims = oracle.knn_similar(d, pid, 0.5, 0.5, k)
            assert np.array_equal(ids, oids) and np.array_equal(sims, osims), (row, k)
        if integer:
            places, est = ix.recommend(pid, 0.25, 0.75, 50)
            oplaces, oest = oracle.knn_recommend(d, pid, 0.25, 0.75, 50)
            assert np.array_equal(places, oplaces), row
            np.testing.assert_allclose(est, oest, rtol=RTOL, atol=0)
    # the shipped --k-nearest 2000000: every positive-similarity person
    pid = int(d["person_ids"][long_rows[1]])
    ids, sims = ix.query(pid, 0.5, 0.5, 2_000_000)
    oids, osims = oracle.knn_similar(d, pid, 0.5, 0.5, 2_000_000)
    assert np.array_equal(ids, oids) and np.array_equal(sims, osims)
    ix.close()


def test_long_queries_inside_batches(pkg, oracle):
    d, long_rows = dataset_with_long_rows(5)
    ix = make_index(pkg, d)
    rng = np.random.default_rng(0)
    rows = np.unique(np.r_[rng.choice(len(d["person_ids"]), 300, replace=False), long_rows])
    ids, sims, cnt = ix.query_batch(d["person_ids"][rows], 0.5, 0.5, 20)
    oids, osims, ocnt = oracle.knn_similar_batch(d, rows, 0.5, 0.5, 20, nthreads=8)
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sims, osims)
    # only long queries; and a single-query batch
    ids, sims, cnt = ix.query_batch(d["person_ids"][long_rows], 0.5, 0.5, 7)
    oids, osims, ocnt = oracle.knn_similar_batch(d, np.array(long_rows), 0.5, 0.5, 7)
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sims, osims)
    ids, sims, cnt = ix.query_batch(d["person_ids"][long_rows[:1]], 0.5, 0.5, 7)
    assert np.array_equal(ids[0], oids[0]) and np.array_equal(sims[0], osims[0])
    # makeRecommendations for a batch that contains them
    sel = d["person_ids"][rows[:40].tolist() + long_rows]
    off, places, est = ix.recommend_batch(sel, 0.5, 0.5, 20)
    for j, pid in enumerate(sel):
        oplaces, oest = oracle.knn_recommend(d, int(pid), 0.5, 0.5, 20)
        assert np.array_equal(places[off[j]:off[j + 1]], oplaces), j
        np.testing.assert_allclose(est[off[j]:off[j + 1]], oest, rtol=RTOL, atol=0)
    ix.close()


def test_all_pairs_with_long_rows(pkg, oracle):
    d, long_rows = dataset_with_long_rows(8, n=4200, p_dim=6000, long_rows=((10, 4500), (4100, 6000)))
    ix = make_index(pkg, d)
    ids, sims, cnt = ix.all_pairs_topk(0.5, 0.5, 10)
    rows = np.unique(np.r_[np.arange(0, 4200, 37), long_rows])
    oids, osims, ocnt = oracle.knn_similar_batch(d, rows, 0.5, 0.5, 10, nthreads=8)
    assert np.array_equal(cnt[rows], ocnt) and np.array_equal(ids[rows], oids) and np.array_equal(sims[rows], osims)
    ix.close()


def test_small_index_with_a_long_row(pkg, oracle):
    """Fewer than 64 candidate slices: a single request normally takes the tiled path, which cannot hold the row."""
    d, long_rows = dataset_with_long_rows(2, n=700, p_dim=7000, long_rows=((5, 6000),))
    ix = make_index(pkg, d)
    for row in (5, 6):
        pid = int(d["person_ids"][row])
        ids, sims = ix.query(pid, 0.5, 0.5, 30)
        oids, osims = oracle.knn_similar(d, pid, 0.5, 0.5, 30)
        assert np.array_equal(ids, oids) and np.array_equal(sims, osims)
    ix.close()


def test_dense_query_scan_agrees_on_ordinary_data(pkg, oracle, monkeypatch):
    """LOCREC_KNN_FORCE_DENSE_QUERY sends EVERY single request through knn_scan_dense."""
    from locations_recommender_amd import synth
    monkeypatch.setenv("LOCREC_KNN_FORCE_DENSE_QUERY", "1")
    for integer, negative in ((True, False), (False, False), (False, True)):
        d = synth.small_knn_dataset(n=5000, p_dim=800, seed=11, integer=integer, negative=negative)
        ix = make_index(pkg, d)
        for row in range(0, 5000, 331):
            pid = int(d["person_ids"][row])
            ids, sims = ix.query(pid, 0.3, 0.7, 25)
            oids, osims = oracle.knn_similar(d, pid, 0.3, 0.7, 25)
            assert np.array_equal(ids, oids) and np.array_equal(sims, osims), (integer, negative, row)
        ix.close()
