"""The CPU oracle against the reference's own known answers (and the hand-derived KNN cases).

These pin the oracle: DistanceTest.scala:10-60 and StochasticRecommenderTest.scala:11-21,39-94
are reproduced bit for bit; KnnRecommender has no reference test ("parity unpinned"), so the
hand-made fixture (tests/golden/make_golden.py) stands in."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def test_vector_length_kats(oracle):
    for case in load("distance_kats.json")["vector_length"]:
        assert oracle.vector_length(case["values"]) == case["expected"], case["name"]


def test_cosine_kats(oracle):
    for case in load("distance_kats.json")["cosine_similarity"]:
        got = oracle.cosine(case["v1"]["indices"], case["v1"]["values"], case["v2"]["indices"], case["v2"]["values"])
        assert got == case["expected"], case["name"]  # exact equality, as DistanceTest does


def sg_edges(g):
    e = np.array(g["edges"], dtype=np.float64)
    return e[:, 0].astype(np.int64), e[:, 1].astype(np.int64), e[:, 2].copy()


def test_sg_kats(oracle):
    g = load("sg_kats.json")
    src, dst, w = sg_edges(g)
    for case in g["cases"]:
        if "expected_error" in case:
            with pytest.raises(oracle.OracleIllegalArgument):
                oracle.sg_recommend(src, dst, w, case["vertex_id"], g["alpha"], case["epsilon"], case["max_iterations"])
            continue
        ids, probs, it, conv = oracle.sg_recommend(src, dst, w, case["vertex_id"], g["alpha"], case["epsilon"],
                                                   case["max_iterations"])
        got = sorted(zip(ids.tolist(), probs.tolist()), key=lambda t: -t[1])  # .sortBy(-_._2)
        assert got == [tuple(x) for x in case["expected_sorted_by_probability_desc"]], case["name"]
    # the converge case stops with the reference's counter at 3
    ids, probs, it, conv = oracle.sg_recommend(src, dst, w, 1, 0.15, 0.05, 1000)
    assert (it, conv) == (3, True)
    ids, probs, it, conv = oracle.sg_recommend(src, dst, w, 1, 0.15, 0.01, 1)
    assert (it, conv) == (1, False)


def knn_fixture():
    g = load("knn_handmade.json")
    ids, prp, pidx, pval, crp, cidx, cval, rrp, rpl, rra = [], [0], [], [], [0], [], [], [0], [], []
    for p in g["persons"]:
        ids.append(p["person_id"])
        pidx += p["place"]["indices"]; pval += p["place"]["values"]; prp.append(len(pidx))
        cidx += p["category"]["indices"]; cval += p["category"]["values"]; crp.append(len(cidx))
        for place, rating in p["ratings"]:
            rpl.append(place); rra.append(rating)
        rrp.append(len(rpl))
    d = {"person_ids": np.array(ids, np.int64),
         "p_rowptr": np.array(prp, np.int64), "p_idx": np.array(pidx, np.int32), "p_val": np.array(pval, np.float64),
         "p_dim": g["place_dim"],
         "c_rowptr": np.array(crp, np.int64), "c_idx": np.array(cidx, np.int32), "c_val": np.array(cval, np.float64),
         "c_dim": g["category_dim"],
         "r_rowptr": np.array(rrp, np.int64), "r_place": np.array(rpl, np.int64), "r_rating": np.array(rra, np.int64)}
    return g, d


def test_knn_handmade(oracle):
    g, d = knn_fixture()
    for q in g["queries"]:
        if "expected_error" in q:
            with pytest.raises(oracle.OracleIllegalArgument):
                oracle.knn_similar(d, q["person_id"], q["pw"], q["cw"], q["k"])
            with pytest.raises(oracle.OracleIllegalArgument):
                oracle.knn_recommend(d, q["person_id"], q["pw"], q["cw"], q["k"])
            continue
        ids, sims = oracle.knn_similar(d, q["person_id"], q["pw"], q["cw"], q["k"])
        assert list(zip(ids.tolist(), sims.tolist())) == [tuple(x) for x in q["expected_neighbours"]], q["name"]
        places, est = oracle.knn_recommend(d, q["person_id"], q["pw"], q["cw"], q["k"])
        exp = q["expected_recommendations"]
        assert places.tolist() == [p for p, _ in exp], q["name"]
        np.testing.assert_allclose(est, [r for _, r in exp], rtol=1e-12, atol=0, err_msg=q["name"])


def test_batch_oracle_matches_single(oracle, pkg):
    from locations_recommender_amd import synth
    d = synth.small_knn_dataset(n=120, p_dim=90, seed=5)
    rows = np.array([0, 5, 17, 63, 119])
    ids, sims, cnt = oracle.knn_similar_batch(d, rows, 0.5, 0.5, 7, nthreads=2)
    for i, r in enumerate(rows):
        sid, ssim = oracle.knn_similar(d, int(d["person_ids"][r]), 0.5, 0.5, 7)
        assert cnt[i] == len(sid)
        assert np.array_equal(ids[i, :cnt[i]], sid) and np.array_equal(sims[i, :cnt[i]], ssim)
