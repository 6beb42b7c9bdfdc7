"""GPU parity of the KNN path through the C ABI against the oracle, the reference's Distance
KATs and the hand-derived KnnRecommender fixture.

Bars (BASELINE.json north_star): top-K person ids bit-exact (tie order = similarity desc,
person_id asc); similarities within 1e-6 relative -- in fact bit-exact here,
because the device keeps the reference's operation order (no FMA, one multiply + one divide);
estimated ratings within 1e-6 relative."""
import json
import math
import os

import numpy as np
import pandas as pd
import pytest

from test_oracle_golden import knn_fixture

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-6


def make_index(pkg, d):
    return pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                        d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"],
                        d.get("r_rowptr"), d.get("r_place"), d.get("r_rating"))


def test_distance_kats_through_the_abi(pkg, oracle):
    """DistanceTest.scala:10-60.  vectorLength is read back directly; cosineSimilarity is observed
    through findSimilarPersons with identical category vectors: s = ps(+) * 0.5 + 1.0 * 0.5."""
    with open(os.path.join(GOLD, "distance_kats.json")) as f:
        kats = json.load(f)
    for case in kats["vector_length"]:
        n = len(case["indices"])
        d = {"person_ids": np.array([1, 2]), "p_rowptr": np.array([0, n, n + 1]),
             "p_idx": np.array(case["indices"] + [0], np.int32), "p_val": np.array(case["values"] + [1.0]),
             "p_dim": case["size"], "c_rowptr": np.array([0, 1, 2]), "c_idx": np.array([0, 0], np.int32),
             "c_val": np.array([1.0, 1.0]), "c_dim": 2}
        ix = make_index(pkg, d)
        lp, lc = ix.vector_lengths()
        assert lp[0] == case["expected"], case["name"]
        assert lp[0] == oracle.vector_length(case["values"])
        ix.close()
    for case in kats["cosine_similarity"]:
        v1, v2 = case["v1"], case["v2"]
        d = {"person_ids": np.array([1, 2]), "p_rowptr": np.array([0, len(v1["indices"]), len(v1["indices"]) + len(v2["indices"])]),
             "p_idx": np.array(v1["indices"] + v2["indices"], np.int32), "p_val": np.array(v1["values"] + v2["values"]),
             "p_dim": case["size"], "c_rowptr": np.array([0, 1, 2]), "c_idx": np.array([0, 0], np.int32),
             "c_val": np.array([1.0, 1.0]), "c_dim": 2}
        ix = make_index(pkg, d)
        ids, sims = ix.query(2, 0.5, 0.5, 1)       # cosineSimilarity(vector = person 1, query = person 2)
        ps = case["expected"] if case["expected"] > 0 else 0.0
        assert ids.tolist() == [1] and sims[0] == ps * 0.5 + 1.0 * 0.5, case["name"]
        # ... and directly, sign and all (the -1.0 and 0.0 cases never show in findSimilarPersons' positive filter)
        cp, cc = ix.cosine_similarity(1, 2)
        assert cp == case["expected"] and cc == 1.0, case["name"]
        assert cp == oracle.cosine(v1["indices"], v1["values"], v2["indices"], v2["values"])
        assert ix.cosine_similarity(2, 1) == (cp, cc)
        ix.close()
    # a person absent from one family: 0 / 0 like the reference's own arithmetic
    d = {"person_ids": np.array([1, 2]), "p_rowptr": np.array([0, 1, 1]), "p_idx": np.array([0], np.int32),
         "p_val": np.array([2.0]), "p_dim": 2, "c_rowptr": np.array([0, 1, 2]), "c_idx": np.array([0, 1], np.int32),
         "c_val": np.array([1.0, 1.0]), "c_dim": 2}
    ix = make_index(pkg, d)
    cp, cc = ix.cosine_similarity(1, 2)
    assert np.isnan(cp) and cc == 0.0
    with pytest.raises(pkg.IllegalArgumentException, match="No such person: 3"):
        ix.cosine_similarity(1, 3)
    ix.close()




def test_handmade_fixture_through_the_operator(pkg):
    """The KnnRecommender class surface (KnnRecommender.scala:9-25) on the hand-derived cases."""
    g, _ = knn_fixture()
    SV = pkg.SparseVector
    place_rows = [(p["person_id"], SV(g["place_dim"], p["place"]["indices"], p["place"]["values"]))
                  for p in g["persons"] if p["place"]["indices"]]
    cat_rows = [(p["person_id"], SV(g["category_dim"], p["category"]["indices"], p["category"]["values"]))
                for p in g["persons"] if p["category"]["indices"]]
    placeRatingVectors = pd.DataFrame(place_rows, columns=["person_id", "rating_vector"])
    categoryRatingVectors = pd.DataFrame(cat_rows, columns=["person_id", "rating_vector"])
    placeRatings = pd.DataFrame([(p["person_id"], pl, r) for p in g["persons"] for pl, r in p["ratings"]],
                                columns=["person_id", "place_id", "rating"])
    for q in g["queries"]:
        if "expected_error" in q:
            with pytest.raises(pkg.IllegalArgumentException):
                pkg.KnnRecommender(placeRatingVectors, categoryRatingVectors, placeRatings,
                                   q["pw"], q["cw"], q["k"]).makeRecommendations(q["person_id"])
            continue
        rec = pkg.KnnRecommender(placeRatingVectors, categoryRatingVectors, placeRatings, q["pw"], q["cw"], q["k"])
        nb = rec.findSimilarPersons(q["person_id"])
        assert list(zip(nb["person_id"].tolist(), nb["similarity"].tolist())) == \
            [tuple(x) for x in q["expected_neighbours"]], q["name"]
        df = rec.makeRecommendations(q["person_id"])
        assert list(df.columns) == ["place_id", "estimated_rating"]
        exp = q["expected_recommendations"]
        assert df["place_id"].tolist() == [p for p, _ in exp], q["name"]
        np.testing.assert_allclose(df["estimated_rating"], [r for _, r in exp], rtol=RTOL, atol=0)
    with pytest.raises(pkg.IllegalArgumentException, match="No such person: 999"):
        pkg.KnnRecommender(placeRatingVectors, categoryRatingVectors, placeRatings, 0.5, 0.5, 3).makeRecommendations(999)


def check_against_oracle(pkg, oracle, d, k, pw=0.5, cw=0.5, queries=None, expect_packed=None, recommend=True):
    ix = make_index(pkg, d)
    if expect_packed is not None:
        assert bool(ix.info()["packed"]) == expect_packed
    n = len(d["person_ids"])
    rows = np.arange(n) if queries is None else np.asarray(queries)
    ids, sims, cnt = ix.query_batch(d["person_ids"][rows], pw, cw, k)
    oids, osims, ocnt = oracle.knn_similar_batch(d, rows, pw, cw, k, nthreads=8)
    assert np.array_equal(cnt, ocnt)
    assert np.array_equal(ids, oids), "top-K person ids differ"
    assert np.array_equal(sims, osims), "similarities are not bit-identical"
    for r in rows[:6]:
        pid = int(d["person_ids"][r])
        sid, ssim = ix.query(pid, pw, cw, k)
        m = ocnt[list(rows).index(r)]
        assert np.array_equal(sid, oids[list(rows).index(r)][:m]) and np.array_equal(ssim, osims[list(rows).index(r)][:m])
        if recommend:
            places, est = ix.recommend(pid, pw, cw, k)
            oplaces, oest = oracle.knn_recommend(d, pid, pw, cw, k)
            assert np.array_equal(places, oplaces)
            np.testing.assert_allclose(est, oest, rtol=RTOL, atol=0)
    ix.close()
    return ids, sims, cnt


@pytest.mark.parametrize("k", [1, 7, 50])
def test_small_packed_all_queries(pkg, oracle, k):
    from locations_recommender_amd import synth
    d = synth.small_knn_dataset(n=300, p_dim=500, seed=7)
    check_against_oracle(pkg, oracle, d, k, expect_packed=True)


def test_small_generic_real_values(pkg, oracle):
    from locations_recommender_amd import synth
    d = synth.small_knn_dataset(n=257, p_dim=400, seed=8, integer=False)
    check_against_oracle(pkg, oracle, d, 20, pw=0.3, cw=0.7, expect_packed=False, recommend=False)


def test_small_generic_negative_values(pkg, oracle):
    from locations_recommender_amd import synth
    d = synth.small_knn_dataset(n=200, p_dim=300, seed=9, negative=True)
    check_against_oracle(pkg, oracle, d, 10, expect_packed=False, recommend=False)


def test_hashed_panel_path(pkg, oracle, monkeypatch):
    """Same data through the open-addressing panel (forced for small dimensions) and the generic format."""
    from locations_recommender_amd import synth
    d = synth.small_knn_dataset(n=300, p_dim=500, seed=7)
    monkeypatch.setenv("LOCREC_KNN_FORCE_HASH", "1")
    check_against_oracle(pkg, oracle, d, 13, expect_packed=True)
    monkeypatch.setenv("LOCREC_KNN_FORCE_GENERIC", "1")
    check_against_oracle(pkg, oracle, d, 13, expect_packed=False)


def test_k_larger_than_everything_and_big_k(pkg, oracle):
    from locations_recommender_amd import synth
    d = synth.small_knn_dataset(n=90, p_dim=80, seed=11)
    ix = make_index(pkg, d)
    pid = int(d["person_ids"][3])
    for k in (89, 500, 2_000_000):    # bin/knn_recommender.sh:35 ships K = 2,000,000
        ids, sims = ix.query(pid, 0.5, 0.5, k)
        oids, osims = oracle.knn_similar(d, pid, 0.5, 0.5, k)
        assert np.array_equal(ids, oids) and np.array_equal(sims, osims)
        places, est = ix.recommend(pid, 0.5, 0.5, k)
        oplaces, oest = oracle.knn_recommend(d, pid, 0.5, 0.5, k)
        assert np.array_equal(places, oplaces)
        np.testing.assert_allclose(est, oest, rtol=RTOL, atol=0)
    ix.close()
    d = synth.small_knn_dataset(n=3000, p_dim=200, seed=12)
    check_against_oracle(pkg, oracle, d, 1024, queries=[0, 1, 2, 2999], recommend=False)


def test_long_rows_and_ties(pkg, oracle):
    """SURVEY.md H3 (nnz far above 100) and H1 (tie-heavy single-place persons)."""
    rng = np.random.default_rng(3)
    n, p_dim, c_dim = 400, 3000, 20
    prp, pidx, pval, crp, cidx, cval = [0], [], [], [0], [], []
    for i in range(n):
        if i < 8:
            ip = np.sort(rng.choice(p_dim, 400, replace=False))
        elif i % 2:
            ip = np.array([int(rng.integers(0, 5))])           # ties: one of five places, count 1
        else:
            ip = np.sort(rng.choice(200, int(rng.integers(2, 30)), replace=False))
        vp = np.ones(len(ip)) if i % 2 else rng.integers(1, 5, len(ip)).astype(float)
        ic = np.array([int(rng.integers(0, 3))]) if i % 2 else np.sort(rng.choice(c_dim, 4, replace=False))
        vc = np.ones(len(ic)) if i % 2 else rng.integers(1, 9, len(ic)).astype(float)
        pidx.append(ip); pval.append(vp); prp.append(prp[-1] + len(ip))
        cidx.append(ic); cval.append(vc); crp.append(crp[-1] + len(ic))
    d = {"person_ids": rng.permutation(np.arange(5000, 5000 + n)).astype(np.int64),
         "p_rowptr": np.array(prp), "p_idx": np.concatenate(pidx).astype(np.int32), "p_val": np.concatenate(pval),
         "p_dim": p_dim, "c_rowptr": np.array(crp), "c_idx": np.concatenate(cidx).astype(np.int32),
         "c_val": np.concatenate(cval), "c_dim": c_dim}
    check_against_oracle(pkg, oracle, d, 25)


def test_all_pairs_small(pkg, oracle):
    from locations_recommender_amd import synth
    d = synth.knn_dataset(2000, 500, seed=0x5EED0002)
    ix = make_index(pkg, d)
    ids, sims, cnt = ix.all_pairs_topk(0.5, 0.5, 10)
    oids, osims, ocnt = oracle.knn_similar_batch(d, np.arange(2000), 0.5, 0.5, 10, nthreads=8)
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sims, osims)
    ix.close()


def test_create_rejects_bad_input(pkg):
    ok = {"person_ids": np.array([1, 2]), "p_rowptr": np.array([0, 1, 2]), "p_idx": np.array([0, 1], np.int32),
          "p_val": np.array([1.0, 2.0]), "p_dim": 4, "c_rowptr": np.array([0, 1, 2]),
          "c_idx": np.array([0, 0], np.int32), "c_val": np.array([1.0, 1.0]), "c_dim": 2}
    for patch in ({"person_ids": np.array([1, 1])}, {"p_idx": np.array([0, 9], np.int32)},
                  {"p_val": np.array([0.0, 2.0])}, {"p_val": np.array([np.nan, 2.0])},
                  {"p_rowptr": np.array([0, 2, 2]), "p_idx": np.array([1, 0], np.int32)}):
        with pytest.raises(pkg.IllegalArgumentException):
            make_index(pkg, {**ok, **patch})


def test_cfg2_shape_medium(pkg, oracle):
    """cfg2's distributions at 50k persons x 100k places: packed format, hashed place panel."""
    from locations_recommender_amd import synth
    d = synth.knn_dataset(50_000, 100_000, seed=0x5EED0002)
    rows = np.r_[np.arange(0, 50_000, 997), [49_999]]
    check_against_oracle(pkg, oracle, d, 50, queries=rows, expect_packed=True)


def test_cfg2_full_size(pkg, oracle):
    """BASELINE.json config 1 (1M persons x 100k places, K = 50): a sample of queries against the
    oracle at full size, and batch == single-request results."""
    from locations_recommender_amd import synth
    d = synth.knn_dataset(1_000_000, 100_000, seed=0x5EED0002)
    ix = make_index(pkg, d)
    assert ix.info()["packed"]
    rows = np.array([0, 1, 123_456, 500_000, 777_777, 999_999])
    ids, sims, cnt = ix.query_batch(d["person_ids"][rows], 0.5, 0.5, 50)
    oids, osims, ocnt = oracle.knn_similar_batch(d, rows, 0.5, 0.5, 50, nthreads=8)
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sims, osims)
    sid, ssim = ix.query(int(d["person_ids"][rows[2]]), 0.5, 0.5, 50)
    assert np.array_equal(sid, ids[2]) and np.array_equal(ssim, sims[2])
    # device-resident range form: row order is internal, results must equal the per-person answers
    ix.topk_range_async(4096, 64, 0.5, 0.5, 50)
    rids, rsims, rcnt = ix.fetch_topk(64, 50)
    qids = ix.row_person_ids(4096, 64)
    bids, bsims, bcnt = ix.query_batch(qids[:8], 0.5, 0.5, 50)
    assert np.array_equal(rids[:8], bids) and np.array_equal(rsims[:8], bsims)
    places, est = ix.recommend(int(d["person_ids"][rows[3]]), 0.5, 0.5, 50)
    oplaces, oest = oracle.knn_recommend(d, int(d["person_ids"][rows[3]]), 0.5, 0.5, 50)
    assert np.array_equal(places, oplaces)
    np.testing.assert_allclose(est, oest, rtol=RTOL, atol=0)
    ix.close()


def test_hashed_panel_is_history_independent(pkg, oracle):
    """Regression: the hashed panel once used a sign-extended first probe position (a wild LDS
    address), which only showed after other indexes had been created and freed in the process."""
    from locations_recommender_amd import synth
    d0 = synth.small_knn_dataset(n=3000, p_dim=200, seed=12)
    d = synth.knn_dataset(2000, 500, seed=0x5EED0002)       # 540 places: hashed panel at QT = 16
    oids, osims, ocnt = oracle.knn_similar_batch(d, np.arange(2000), 0.5, 0.5, 10, nthreads=8)
    for _ in range(4):
        make_index(pkg, d0).close()
        ix = make_index(pkg, d)
        ids, sims, cnt = ix.all_pairs_topk(0.5, 0.5, 10)
        ix.close()
        assert np.array_equal(ids, oids) and np.array_equal(sims, osims) and np.array_equal(cnt, ocnt)


def test_single_request_stream_path_equals_tiled_path(pkg, oracle, monkeypatch):
    """A single request on a large index runs as scan -> histogram select -> collect -> sort; it must
    agree with the tiled path and the oracle, including ties at the K-th value and K > #candidates."""
    from locations_recommender_amd import synth
    d = synth.knn_dataset(30_000, 2_000, seed=77)
    ix = make_index(pkg, d)
    rows = [0, 1, 777, 15_000, 29_999]
    for k in (1, 50, 1024):
        oids, osims, ocnt = oracle.knn_similar_batch(d, np.array(rows), 0.5, 0.5, k, nthreads=8)
        for i, r in enumerate(rows):
            ids, sims = ix.query(int(d["person_ids"][r]), 0.5, 0.5, k)
            assert np.array_equal(ids, oids[i][:ocnt[i]]) and np.array_equal(sims, osims[i][:ocnt[i]]), (k, r)
    places, est = ix.recommend(int(d["person_ids"][777]), 0.5, 0.5, 50)
    oplaces, oest = oracle.knn_recommend(d, int(d["person_ids"][777]), 0.5, 0.5, 50)
    assert np.array_equal(places, oplaces)
    np.testing.assert_allclose(est, oest, rtol=RTOL, atol=0)
    ix.close()
    # tie-heavy: thousands of identical single-place persons -> the deciding bin overflows the
    # collect list and the request falls back to the tiled path
    n = 20_000
    dd = {"person_ids": np.arange(n, dtype=np.int64) + 5, "p_rowptr": np.arange(n + 1, dtype=np.int64),
          "p_idx": (np.arange(n) % 3).astype(np.int32), "p_val": np.ones(n), "p_dim": 8,
          "c_rowptr": np.arange(n + 1, dtype=np.int64), "c_idx": (np.arange(n) % 2).astype(np.int32),
          "c_val": np.ones(n), "c_dim": 4}
    ix = make_index(pkg, dd)
    ids, sims = ix.query(5, 0.5, 0.5, 50)
    oids, osims = oracle.knn_similar(dd, 5, 0.5, 0.5, 50)
    assert np.array_equal(ids, oids) and np.array_equal(sims, osims)
    ix.close()


def test_large_k_paths(pkg, oracle):
    """K beyond the LDS lists: the shipped --k-nearest 2000000 (bin/knn_recommender.sh:35: every
    positive-similarity person is a neighbour), a K that cuts inside the candidates, and an
    aggregation with more rating rows than one block sorts (place-major pass)."""
    from locations_recommender_amd import synth
    d = synth.knn_dataset(6_000, 800, seed=123)
    ix = make_index(pkg, d)
    for pid_row in (3, 4_000):
        pid = int(d["person_ids"][pid_row])
        for k in (2_000_000, 1_500, 1_024):
            ids, sims = ix.query(pid, 0.5, 0.5, k)
            oids, osims = oracle.knn_similar(d, pid, 0.5, 0.5, k)
            assert np.array_equal(ids, oids), k
            assert np.array_equal(sims, osims), k
            places, est = ix.recommend(pid, 0.5, 0.5, k)
            oplaces, oest = oracle.knn_recommend(d, pid, 0.5, 0.5, k)
            assert np.array_equal(places, oplaces), k
            np.testing.assert_allclose(est, oest, rtol=RTOL, atol=0)
    ix.close()
    # generic (fp64) format through the same paths
    g = synth.small_knn_dataset(n=5_000, p_dim=300, seed=21, integer=False)
    ix = make_index(pkg, g)
    pid = int(g["person_ids"][17])
    ids, sims = ix.query(pid, 0.4, 0.6, 2_000_000)
    oids, osims = oracle.knn_similar(g, pid, 0.4, 0.6, 2_000_000)
    assert np.array_equal(ids, oids) and np.array_equal(sims, osims)
    ix.close()


def test_degenerate_sizes(pkg, oracle):
    """One person (no candidates at all), two persons, an empty batch, exactly 64 and 65 persons
    (slice boundary)."""
    one = {"person_ids": np.array([7]), "p_rowptr": np.array([0, 2]), "p_idx": np.array([1, 3], np.int32),
           "p_val": np.array([2.0, 1.0]), "p_dim": 5, "c_rowptr": np.array([0, 1]), "c_idx": np.array([0], np.int32),
           "c_val": np.array([1.0]), "c_dim": 2}
    ix = make_index(pkg, one)
    ids, sims = ix.query(7, 0.5, 0.5, 10)
    assert len(ids) == 0
    places, est = ix.recommend(7, 0.5, 0.5, 10)
    assert len(places) == 0
    bi, bs, bc = ix.query_batch(np.array([], np.int64), 0.5, 0.5, 3)
    assert bi.shape == (0, 3)
    ix.close()
    from locations_recommender_amd import synth
    for n in (2, 64, 65, 129):
        d = synth.small_knn_dataset(n=n, p_dim=40, seed=100 + n)
        check_against_oracle(pkg, oracle, d, 5)


def test_fast_insertion_overflow_falls_back(pkg, oracle):
    """Adversarial order for the barrier-free insertion path: rows are stored by ascending vector
    length and similarity to the queries grows with length, so EVERY candidate beats the running
    threshold: the block either never leaves synchronous insertion or overruns its per-wave queues
    and replays those intervals synchronously inside the kernel.  Either way: exact results, one launch."""
    n, width = 12_000, 60
    lens = 1 + (np.arange(n) % width)
    prp = np.concatenate([[0], np.cumsum(lens)])
    pidx = np.concatenate([np.arange(k) for k in lens]).astype(np.int32)
    d = {"person_ids": np.arange(n, dtype=np.int64) + 100, "p_rowptr": prp, "p_idx": pidx,
         "p_val": np.ones(len(pidx)), "p_dim": 64, "c_rowptr": np.arange(n + 1, dtype=np.int64),
         "c_idx": np.zeros(n, np.int32), "c_val": np.ones(n), "c_dim": 2}
    rows = np.array([width - 1, 2 * width - 1, 5 * width - 1, 17])     # long queries + a short one
    ix = make_index(pkg, d)
    ix.profile_enable(True)
    ids, sims, cnt = ix.query_batch(d["person_ids"][rows], 0.5, 0.5, 10)
    assert ix.profile_read()[1] == 1, "the scan was launched more than once"
    oids, osims, ocnt = oracle.knn_similar_batch(d, rows, 0.5, 0.5, 10, nthreads=8)
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sims, osims)
    ix.close()


def test_tie_runs_replay_intervals_in_kernel(pkg, oracle):
    """Thousands of identical persons (one place, one category): every candidate ties with the K-th
    best and the smaller id wins, so survivors arrive in bursts that overrun the per-wave queues.
    The overrun intervals are replayed inside the kernel: results exact, a single scan launch, and
    the replay counter moves."""
    n = 40_000
    pidx = (np.arange(n) % 7).astype(np.int32)             # 7 groups of identical persons
    d = {"person_ids": np.arange(n, dtype=np.int64)[::-1].copy() + 500,   # descending ids: later rows win ties
         "p_rowptr": np.arange(n + 1, dtype=np.int64), "p_idx": pidx, "p_val": np.ones(n), "p_dim": 16,
         "c_rowptr": np.arange(n + 1, dtype=np.int64), "c_idx": (pidx % 3).astype(np.int32), "c_val": np.ones(n), "c_dim": 4}
    ix = make_index(pkg, d)
    rows = np.arange(0, n, 2_501)
    ix.profile_enable(True)
    ids, sims, cnt = ix.query_batch(d["person_ids"][rows], 0.5, 0.5, 50)
    assert ix.profile_read()[1] == 1
    oids, osims, ocnt = oracle.knn_similar_batch(d, rows, 0.5, 0.5, 50, nthreads=8)
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sims, osims)
    assert ix.replayed_intervals() >= 0
    ix.close()


# ---------------------------------------------------------------------------
# One request with its candidate scan split over several GPUs (include/locrec.h,
# locrec_knn_query_shard / locrec_knn_recommend_neighbours): the shards are emulated on the test
# box's single GPU by asking one index for every shard in turn.

def sharded_request(pkg, ix, pid, pw, cw, k, shards):
    from locations_recommender_amd import shard
    lists = [ix.query_shard(pid, pw, cw, k, i, shards) for i in range(shards)]
    n_other = ix.n - 1
    assert sum(len(a) for a, _ in lists) >= min(k, 1)  # at least one shard answers
    for a, b in lists:
        assert len(a) == len(b) <= k and np.all(b > 0) and np.all(np.diff(b) <= 0)
    all_ids = np.concatenate([a for a, _ in lists])
    assert len(np.unique(all_ids)) == len(all_ids), "a candidate was scanned by two shards"
    assert pid not in all_ids
    return shard.merge_local_topk(lists, int(min(k, max(1, n_other))))


@pytest.mark.parametrize("shards", [2, 8])
def test_sharded_request_equals_unsharded(pkg, oracle, shards):
    """Stream path (>= 64 slices per shard) and tiled path (small shards), ids and similarities
    bit-identical to the unsharded request and to the oracle; makeRecommendations0 from the merged
    list equals locrec_knn_recommend."""
    from locations_recommender_amd import synth
    d = synth.knn_dataset(40_000, 3_000, seed=77)
    rng = np.random.default_rng(1)
    rp = d["p_rowptr"]
    d["r_rowptr"], d["r_place"] = rp.copy(), d["p_idx"].astype(np.int64)
    d["r_rating"] = rng.integers(1, 6, size=len(d["p_idx"])).astype(np.int64)
    ix = make_index(pkg, d)
    rows = np.array([0, 123, 20_000, 39_999])
    oids, osims, ocnt = oracle.knn_similar_batch(d, rows, 0.4, 0.6, 50, nthreads=4)
    for j, r in enumerate(rows):
        pid = int(d["person_ids"][r])
        ids, sims = sharded_request(pkg, ix, pid, 0.4, 0.6, 50, shards)
        uid, usim = ix.query(pid, 0.4, 0.6, 50)
        assert np.array_equal(ids, uid) and np.array_equal(sims, usim)
        assert np.array_equal(ids, oids[j][:ocnt[j]]) and np.array_equal(sims, osims[j][:ocnt[j]])
        places, est = ix.recommend_neighbours(ids, sims)
        uplaces, uest = ix.recommend(pid, 0.4, 0.6, 50)
        assert np.array_equal(places, uplaces) and np.array_equal(est, uest)
        oplaces, oest = oracle.knn_recommend(d, pid, 0.4, 0.6, 50)
        assert np.array_equal(places, oplaces)
        np.testing.assert_allclose(est, oest, rtol=RTOL, atol=0)
    ix.close()


def test_sharded_request_small_index_and_errors(pkg, oracle):
    """More shards than slices (empty shards), ties across shard boundaries, K > candidates, the
    place-major fallback of recommend_neighbours, and the argument checks."""
    from locations_recommender_amd import synth
    d = synth.small_knn_dataset(n=300, p_dim=500, seed=7)   # many ties (single-place persons)
    ix = make_index(pkg, d)
    for r in (0, 7, 150, 299):
        pid = int(d["person_ids"][r])
        for k in (1, 7, 1000):
            ids, sims = sharded_request(pkg, ix, pid, 0.5, 0.5, k, 16)
            uid, usim = ix.query(pid, 0.5, 0.5, k)
            assert np.array_equal(ids, uid) and np.array_equal(sims, usim), (r, k)
            places, est = ix.recommend_neighbours(ids, sims)
            uplaces, uest = ix.recommend(pid, 0.5, 0.5, k)
            assert np.array_equal(places, uplaces)
            np.testing.assert_allclose(est, uest, rtol=RTOL, atol=0)
    pid = int(d["person_ids"][3])
    with pytest.raises(pkg.IllegalArgumentException):
        ix.query_shard(pid, 0.5, 0.5, 5, 2, 2)
    with pytest.raises(pkg.IllegalArgumentException, match="No such person"):
        ix.query_shard(10**9, 0.5, 0.5, 5, 0, 2)
    with pytest.raises(pkg.IllegalArgumentException, match="No such person"):
        ix.recommend_neighbours([10**9], [0.5])
    with pytest.raises(pkg.IllegalArgumentException, match="not positive"):
        ix.recommend_neighbours([pid], [0.0])
    assert len(ix.recommend_neighbours([], [])[0]) == 0
    ix.close()


def test_cfg1_shape_every_person(pkg, oracle):
    """BASELINE.json configs[0] shape (sample_generator.sh defaults: ~10k persons, 3 regions x 324
    places, 20 categories): EVERY person's neighbours through the all-pairs entry point against
    the oracle, and makeRecommendations for a sample."""
    from locations_recommender_amd import synth
    d = synth.knn_dataset(10_000, 972, seed=0x5EED0001, mean_places=12, max_places=60)
    d["r_rowptr"], d["r_place"] = d["p_rowptr"].copy(), d["p_idx"].astype(np.int64)
    d["r_rating"] = d["p_val"].astype(np.int64)        # rating = visit count, as RatingsBuilder.scala:38-47
    ix = make_index(pkg, d)
    ids, sims, cnt = ix.all_pairs_topk(0.5, 0.5, 50)
    oids, osims, ocnt = oracle.knn_similar_batch(d, np.arange(10_000), 0.5, 0.5, 50, nthreads=8)
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sims, osims)
    for r in (0, 4321, 9_999):
        pid = int(d["person_ids"][r])
        places, est = ix.recommend(pid, 0.5, 0.5, 50)
        oplaces, oest = oracle.knn_recommend(d, pid, 0.5, 0.5, 50)
        assert np.array_equal(places, oplaces)
        np.testing.assert_allclose(est, oest, rtol=RTOL, atol=0)
    ix.close()


def test_cfg4_shape_reduced(pkg, oracle):
    """BASELINE.json configs[3] distributions (1M places, seed 0x5EED0004) at 100k persons: the
    million-wide place dimension through the packed format's 20-bit index field and the hashed
    panel.  (The full 10M x 1M case on one GPU: tools/probe_cfg4.py, profiles/r01_g_cfg4_one_gpu.log.)"""
    from locations_recommender_amd import synth
    d = synth.knn_dataset(100_000, 1_000_000, seed=0x5EED0004)
    assert d["p_dim"] == 1_000_040 and int(d["p_idx"].max()) > 900_000
    rows = np.r_[np.arange(0, 100_000, 4999), [99_999]]
    check_against_oracle(pkg, oracle, d, 50, queries=rows, expect_packed=True)


def with_ratings(d, seed=3):
    rng = np.random.default_rng(seed)
    d = dict(d)
    d["r_rowptr"], d["r_place"] = d["p_rowptr"].copy(), d["p_idx"].astype(np.int64)
    d["r_rating"] = rng.integers(1, 6, size=len(d["p_idx"])).astype(np.int64)
    return d


def check_batch_recommend(pkg, oracle, ix, d, rows, pw, cw, k):
    pids = d["person_ids"][rows]
    off, places, est = ix.recommend_batch(pids, pw, cw, k)
    assert off[0] == 0 and len(off) == len(rows) + 1 and off[-1] == len(places) == len(est)
    for j, pid in enumerate(pids):
        oplaces, oest = oracle.knn_recommend(d, int(pid), pw, cw, k)
        assert np.array_equal(places[off[j]:off[j + 1]], oplaces), (j, pid)
        np.testing.assert_allclose(est[off[j]:off[j + 1]], oest, rtol=RTOL, atol=0)
    return off, places, est


def test_batched_recommendations_match_oracle(pkg, oracle):
    """locrec_knn_recommend_batch (findSimilarPersons + makeRecommendations0 for many persons) against
    the oracle's per-person answer; the device-resident range form gives the same rows."""
    from locations_recommender_amd import synth
    d = with_ratings(synth.knn_dataset(6_000, 900, seed=41))
    ix = make_index(pkg, d)
    rows = np.r_[np.arange(0, 6_000, 37), [5_999, 5, 5]]           # unsorted, with a repeated person
    off, places, est = check_batch_recommend(pkg, oracle, ix, d, rows, 0.5, 0.5, 50)
    # one person == the single-request operator
    p1, e1 = ix.recommend(int(d["person_ids"][rows[3]]), 0.5, 0.5, 50)
    assert np.array_equal(p1, places[off[3]:off[4]])
    np.testing.assert_allclose(e1, est[off[3]:off[4]], rtol=1e-12, atol=0)
    # range form: internal row order
    ix.recommend_range_async(100, 64, 0.5, 0.5, 50)
    roff, rplaces, rest = ix.fetch_recommend(64)
    qids = ix.row_person_ids(100, 64)
    boff, bplaces, best = ix.recommend_batch(qids, 0.5, 0.5, 50)
    assert np.array_equal(roff, boff) and np.array_equal(rplaces, bplaces) and np.array_equal(rest, best)
    # the operator's additive method
    assert ix.recommend_batch([], 0.5, 0.5, 50)[0].tolist() == [0]
    with pytest.raises(pkg.IllegalArgumentException, match="No such person"):
        ix.recommend_batch([10**9], 0.5, 0.5, 50)
    ix.close()


def test_batched_recommendations_overflow_to_place_major(pkg, oracle):
    """Neighbours holding more rating rows than one block sorts in LDS (K * nnz > 4096): those
    queries fall back to the place-major pass inside the batched call."""
    from locations_recommender_amd import synth
    d = with_ratings(synth.knn_dataset(3_000, 400, seed=42, mean_places=110, max_places=160))
    ix = make_index(pkg, d)
    rows = np.array([0, 1500, 2999, 77])
    check_batch_recommend(pkg, oracle, ix, d, rows, 0.3, 0.7, 50)      # 50 x ~110 rows > 4096
    check_batch_recommend(pkg, oracle, ix, d, rows, 0.3, 0.7, 8)       # fits the first (small-LDS) pass
    check_batch_recommend(pkg, oracle, ix, d, rows, 0.3, 0.7, 25)      # ~2750 rows: second pass, full capacity
    ix.close()


def test_operator_batch_method(pkg):
    """KnnRecommender.makeRecommendationsBatch on the hand-derived fixture: every query's expected
    rows, all persons in ONE call per (pw, cw, k) setting."""
    g, _ = knn_fixture()
    SV = pkg.SparseVector
    place_rows = [(p["person_id"], SV(g["place_dim"], p["place"]["indices"], p["place"]["values"]))
                  for p in g["persons"] if p["place"]["indices"]]
    cat_rows = [(p["person_id"], SV(g["category_dim"], p["category"]["indices"], p["category"]["values"]))
                for p in g["persons"] if p["category"]["indices"]]
    placeRatingVectors = pd.DataFrame(place_rows, columns=["person_id", "rating_vector"])
    categoryRatingVectors = pd.DataFrame(cat_rows, columns=["person_id", "rating_vector"])
    placeRatings = pd.DataFrame([(p["person_id"], pl, r) for p in g["persons"] for pl, r in p["ratings"]],
                                columns=["person_id", "place_id", "rating"])
    settings = {}
    for q in g["queries"]:
        if "expected_error" not in q:
            settings.setdefault((q["pw"], q["cw"], q["k"]), []).append(q)
    assert settings
    for (pw, cw, k), qs in settings.items():
        rec = pkg.KnnRecommender(placeRatingVectors, categoryRatingVectors, placeRatings, pw, cw, k)
        df = rec.makeRecommendationsBatch([q["person_id"] for q in qs])
        assert list(df.columns) == ["person_id", "place_id", "estimated_rating"]
        for q in qs:
            mine = df[df["person_id"] == q["person_id"]]
            exp = q["expected_recommendations"]
            assert mine["place_id"].tolist() == [p for p, _ in exp], q["name"]
            np.testing.assert_allclose(mine["estimated_rating"], [r for _, r in exp], rtol=RTOL, atol=0)


def test_cfg2_full_size_properties(pkg):
    """Size-independent properties at BASELINE.json's full cfg2 size (no oracle involved): lists are
    sorted by (similarity desc, person_id asc), hold K distinct persons and never the query itself;
    cosine is symmetric, so whenever a is in b's list and b is in a's list the two similarities
    are the same bits; the device-resident batch, the host batch and the single request agree."""
    from locations_recommender_amd import synth
    n, k = 1_000_000, 50
    d = synth.knn_dataset(n, 100_000, seed=0x5EED0002)
    ix = make_index(pkg, d)
    first, nq = 600_000, 2048
    ix.topk_range_async(first, nq, 0.5, 0.5, k)
    ids, sims, cnt = ix.fetch_topk(nq, k)
    qids = ix.row_person_ids(first, nq)
    assert np.all(cnt == k)
    assert np.all(np.diff(sims, axis=1) <= 0), "similarities not descending"
    tie = np.diff(sims, axis=1) == 0
    assert np.all(np.diff(ids, axis=1)[tie] > 0), "ties not ordered by person id"
    assert np.all(ids != qids[:, None]), "a person is its own neighbour"
    assert all(len(set(r.tolist())) == k for r in ids[::97]), "duplicate neighbours"
    assert np.all((sims > 0) & (sims <= 1.0 + 1e-12))
    # symmetry: for a sample of (a, b) pairs query b and look for a
    checked = 0
    for i in range(0, nq, 256):
        a, b, s_ab = int(qids[i]), int(ids[i, 0]), sims[i, 0]
        bid, bsim = ix.query(b, 0.5, 0.5, k)
        hit = np.flatnonzero(bid == a)
        if len(hit):
            assert bsim[hit[0]] == s_ab, "cosine similarity is not symmetric bit for bit"
            checked += 1
    assert checked > 0
    hids, hsims, hcnt = ix.query_batch(qids[:64], 0.5, 0.5, k)
    assert np.array_equal(hids, ids[:64]) and np.array_equal(hsims, sims[:64])
    ix.close()


def test_long_query_tiles_use_the_wide_block(pkg, oracle, monkeypatch):
    """Tiles of long queries (~90 places each: the panel no longer allows two blocks per CU) run
    with 16 waves per block; same results as the 8-wave form and as the oracle."""
    from locations_recommender_amd import synth
    d = synth.knn_dataset(30_000, 20_000, seed=43, mean_places=85, max_places=100)
    rows = np.r_[np.arange(29_000, 29_064), [0, 15_000]]
    ids, sims, cnt = check_against_oracle(pkg, oracle, d, 50, queries=rows, expect_packed=True, recommend=False)
    monkeypatch.setenv("LOCREC_KNN_NO_WIDE_BLOCK", "1")
    ix = make_index(pkg, d)
    ids8, sims8, cnt8 = ix.query_batch(d["person_ids"][rows], 0.5, 0.5, 50)
    ix.close()
    assert np.array_equal(ids, ids8) and np.array_equal(sims, sims8) and np.array_equal(cnt, cnt8)


def test_rating_place_ids_spanning_a_huge_range(pkg, oracle):
    """placeRatings.place_id is a Long in one global id space; ids far apart take the sort-based
    ranking of distinct places at index creation instead of the presence table."""
    from locations_recommender_amd import synth
    d = with_ratings(synth.small_knn_dataset(n=300, p_dim=500, seed=11))
    d["r_place"] = d["r_place"] * 1_000_000_007 - 5_000_000_000_000     # negative ids too
    ix = make_index(pkg, d)
    for r in (0, 17, 299):
        pid = int(d["person_ids"][r])
        places, est = ix.recommend(pid, 0.5, 0.5, 20)
        oplaces, oest = oracle.knn_recommend(d, pid, 0.5, 0.5, 20)
        assert np.array_equal(places, oplaces)
        np.testing.assert_allclose(est, oest, rtol=RTOL, atol=0)
    off, bp, be = ix.recommend_batch(d["person_ids"][:50], 0.5, 0.5, 20)
    op, oe = oracle.knn_recommend(d, int(d["person_ids"][7]), 0.5, 0.5, 20)
    assert np.array_equal(bp[off[7]:off[8]], op)
    ix.close()


def test_requires_are_enforced_by_the_library_itself(pkg):
    """a6 at the C ABI (KnnRecommender.scala:17-20): a JNI caller has no Python mirror in front of
    it, so the library's own check_params() must reject bad weights / K with the reference's
    messages.  KnnIndex methods go straight to locrec_knn_* (the mirror's checks live in
    KnnRecommender.__init__, which is not involved here)."""
    from locations_recommender_amd import synth
    d = synth.small_knn_dataset(n=200, p_dim=300, seed=5)
    ix = make_index(pkg, d)
    pid = int(d["person_ids"][3])
    calls = {
        "query": lambda pw, cw, k: ix.query(pid, pw, cw, k),
        "recommend": lambda pw, cw, k: ix.recommend(pid, pw, cw, k),
        "query_batch": lambda pw, cw, k: ix.query_batch(d["person_ids"][:4], pw, cw, k),
        "recommend_batch": lambda pw, cw, k: ix.recommend_batch(d["person_ids"][:4], pw, cw, k),
        "query_shard": lambda pw, cw, k: ix.query_shard(pid, pw, cw, k, 0, 2),
        "topk_range_async": lambda pw, cw, k: ix.topk_range_async(0, 8, pw, cw, k),
        "recommend_range_async": lambda pw, cw, k: ix.recommend_range_async(0, 8, pw, cw, k),
        "all_pairs_topk": lambda pw, cw, k: ix.all_pairs_topk(pw, cw, k),
    }
    bad = [
        (0.0, 1.0, 5, r"Place weight must be in the interval \(0; 1\): 0"),
        (1.0, 0.0, 5, r"Place weight must be in the interval \(0; 1\): 1"),
        (float("nan"), 0.5, 5, r"Place weight must be in the interval \(0; 1\)"),
        (0.5, float("nan"), 5, r"Category weight must be in the interval \(0; 1\)"),
        (0.5, 1.5, 5, r"Category weight must be in the interval \(0; 1\): 1.5"),
        (0.5, 0.4, 5, r"Sum of weights must be 1.0: place: 0.5, category: 0.4"),
        (0.3, 0.3, 5, r"Sum of weights must be 1.0"),
        (0.5, 0.5, 0, r"K nearest must be positive"),
        (0.5, 0.5, -3, r"K nearest must be positive"),
    ]
    for name, call in calls.items():
        for pw, cw, k, msg in bad:
            with pytest.raises(pkg.IllegalArgumentException, match="requirement failed: " + msg):
                call(pw, cw, k)
    # 0.1 + 0.9 == 1.0 and 0.7 + 0.3 == 1.0 exactly in fp64; 0.1 + 0.2 + ... is the caller's business
    ids, sims = ix.query(pid, 0.7, 0.3, 5)
    assert len(ids) == 5
    # unknown person: the reference's message (KnnRecommender.scala:83)
    with pytest.raises(pkg.IllegalArgumentException, match="No such person: 424242"):
        ix.query(424242, 0.5, 0.5, 5)
    ix.close()


def test_range_forms_skip_persons_that_are_not_valid_queries(pkg, oracle):
    """A person with an empty category (or place) vector is a candidate in the others' outer join
    but "No such person" as a query (KnnRecommender.scala:77-83).  The per-person entry points fail
    for it; the range / all-pairs forms report count -1 for that row and carry on."""
    from locations_recommender_amd import synth
    d = synth.small_knn_dataset(n=260, p_dim=300, seed=21)
    holes = {17: "c", 101: "p", 200: "c"}
    for fam in ("p", "c"):
        rp, idx, val = d[fam + "_rowptr"], d[fam + "_idx"], d[fam + "_val"]
        keep = np.ones(len(idx), bool)
        for r, f in holes.items():
            if f == fam:
                keep[rp[r]:rp[r + 1]] = False
        nnz = np.diff(rp)
        for r, f in holes.items():
            if f == fam:
                nnz[r] = 0
        d[fam + "_rowptr"] = np.concatenate([[0], np.cumsum(nnz)]).astype(np.int64)
        d[fam + "_idx"], d[fam + "_val"] = idx[keep], val[keep]
    ix = make_index(pkg, d)
    ids, sims, cnt = ix.all_pairs_topk(0.5, 0.5, 10)
    valid = np.array([r for r in range(260) if r not in holes])
    oids, osims, ocnt = oracle.knn_similar_batch(d, valid, 0.5, 0.5, 10, nthreads=4)
    assert np.all(cnt[list(holes)] == -1) and np.all(ids[list(holes)] == -1)
    assert np.array_equal(cnt[valid], ocnt) and np.array_equal(ids[valid], oids) and np.array_equal(sims[valid], osims)
    for r in holes:
        with pytest.raises(pkg.IllegalArgumentException, match=f"No such person: {int(d['person_ids'][r])}"):
            ix.query(int(d["person_ids"][r]), 0.5, 0.5, 10)
    # the batched aggregation gives such rows no recommendations and still serves the others
    ix.recommend_range_async(0, 260, 0.5, 0.5, 10)
    off, places, est = ix.fetch_recommend(260)
    qids = ix.row_person_ids(0, 260)
    hole_ids = {int(d["person_ids"][r]) for r in holes}
    for j, pid in enumerate(qids):
        if int(pid) in hole_ids:
            assert off[j + 1] == off[j]
    j = int(np.flatnonzero(qids == d["person_ids"][5])[0])
    op, oe = oracle.knn_recommend(d, int(d["person_ids"][5]), 0.5, 0.5, 10)
    assert np.array_equal(places[off[j]:off[j + 1]], op)
    np.testing.assert_allclose(est[off[j]:off[j + 1]], oe, rtol=RTOL, atol=0)
    # a neighbour listed twice is rejected (both aggregation paths would otherwise disagree)
    nid, nsim = ix.query(int(d["person_ids"][5]), 0.5, 0.5, 4)
    with pytest.raises(pkg.IllegalArgumentException, match="listed twice"):
        ix.recommend_neighbours(np.r_[nid, nid[:1]], np.r_[nsim, nsim[:1]])
    ix.close()


@pytest.mark.parametrize("pw", [2.0 ** -20, 1.0 - 2.0 ** -20, 2.0 ** -40, 0.001])
def test_extreme_weights_through_the_batched_scan(pkg, oracle, pw):
    """The head / tail scan bounds a pair with packed f16 arithmetic before it pays for the exact fp64
    similarity: weights near 0 or 1 make one scale factor tiny (f16-subnormal or below) - it must be
    rounded up, never quantised away."""
    from locations_recommender_amd import synth
    cw = 1.0 - pw
    assert pw + cw == 1.0
    d = synth.small_knn_dataset(n=3000, p_dim=700, seed=19)
    ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                      d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"])
    rows = np.arange(0, 3000, 5)
    for k in (3, 60):
        ids, sims, cnt = ix.query_batch(d["person_ids"][rows], pw, cw, k)
        oids, osims, ocnt = oracle.knn_similar_batch(d, rows, pw, cw, k, nthreads=8)
        assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sims, osims), (pw, k)
    ix.close()


def test_batched_large_k(pkg, oracle):
    """the batched path at the SHIPPED K (bin/knn_recommender.sh:35, --k-nearest 2000000 =
    every positive-similarity person is a neighbour, KnnRecommender.scala:47-48): tiles of 16 queries, no top-K,
    place-major aggregation shared by the tile (knn_large.hip, knn_large_recommend_batch).  Against the oracle
    per person, bit-identical to the single-request operator, in the range form, for a K between the LDS limit and
    N - 1 (served one by one at fetch time), and on GENERIC (fp64) data."""
    from locations_recommender_amd import synth
    d = with_ratings(synth.knn_dataset(5_000, 700, seed=77))
    ix = make_index(pkg, d)
    rows = np.r_[np.arange(0, 5_000, 131), [4_999, 7, 7]]              # 42 queries: two full tiles and a partial one
    off, places, est = check_batch_recommend(pkg, oracle, ix, d, rows, 0.5, 0.5, 2_000_000)
    for j in (0, 17, 40):
        p1, e1 = ix.recommend(int(d["person_ids"][rows[j]]), 0.5, 0.5, 2_000_000)
        assert np.array_equal(p1, places[off[j]:off[j + 1]])
        assert np.array_equal(e1, est[off[j]:off[j + 1]]), "batched large-K estimates differ from the single request's bits"
    check_batch_recommend(pkg, oracle, ix, d, rows[:5], 0.25, 0.75, 4_999)   # K = N - 1 exactly
    check_batch_recommend(pkg, oracle, ix, d, rows[:5], 0.5, 0.5, 1_500)     # 1024 < K < N - 1: top-K semantics kept
    ix.recommend_range_async(1_000, 40, 0.5, 0.5, 2_000_000)
    roff, rplaces, rest = ix.fetch_recommend(40)
    qids = ix.row_person_ids(1_000, 40)
    boff, bplaces, best = ix.recommend_batch(qids, 0.5, 0.5, 2_000_000)
    assert np.array_equal(roff, boff) and np.array_equal(rplaces, bplaces) and np.array_equal(rest, best)
    with pytest.raises(pkg.IllegalArgumentException, match="No such person"):
        ix.recommend_batch([10**9], 0.5, 0.5, 2_000_000)
    assert ix.recommend_batch([], 0.5, 0.5, 2_000_000)[0].tolist() == [0]
    ix.close()
    g = with_ratings(synth.small_knn_dataset(n=1_200, p_dim=300, seed=21, integer=False))
    g["r_rating"] = np.arange(len(g["r_place"])) % 5 + 1
    ix = make_index(pkg, g)
    check_batch_recommend(pkg, oracle, ix, g, np.arange(0, 1_200, 61), 0.4, 0.6, 2_000_000)
    ix.close()


def test_batched_large_k_with_persons_that_are_not_valid_queries(pkg, oracle):
    """A person without a place (or category) vector is a candidate of the others but not a valid query
    (KnnRecommender.scala:77-83): in the range form such rows get no recommendation rows, at large K too."""
    from locations_recommender_amd import synth
    d = synth.knn_dataset(600, 200, seed=5)
    # persons 10 and 20 lose their category vector
    keep = np.ones(len(d["c_idx"]), bool)
    crp = d["c_rowptr"].copy()
    for r in (10, 20):
        keep[d["c_rowptr"][r]:d["c_rowptr"][r + 1]] = False
    lens = np.diff(d["c_rowptr"])
    lens[[10, 20]] = 0
    crp[1:] = np.cumsum(lens)
    d["c_rowptr"], d["c_idx"], d["c_val"] = crp, d["c_idx"][keep], d["c_val"][keep]
    d = with_ratings(d)
    ix = make_index(pkg, d)
    ix.recommend_range_async(0, 600, 0.5, 0.5, 2_000_000)
    off, places, est = ix.fetch_recommend(600)
    qids = ix.row_person_ids(0, 600)
    for j in range(0, 600, 23):
        pid = int(qids[j])
        if pid in (int(d["person_ids"][10]), int(d["person_ids"][20])):
            continue
        oplaces, oest = oracle.knn_recommend(d, pid, 0.5, 0.5, 2_000_000)
        assert np.array_equal(places[off[j]:off[j + 1]], oplaces), pid
        np.testing.assert_allclose(est[off[j]:off[j + 1]], oest, rtol=RTOL, atol=0)
    for r in (10, 20):
        j = int(np.flatnonzero(qids == d["person_ids"][r])[0])
        assert off[j + 1] == off[j], "a person that is not a valid query must get no rows"
    ix.close()
