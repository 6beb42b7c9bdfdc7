"""CPU checks of the oracle's f-2 / f-4 restatements (oracle/locrec_oracle.c): against the reference's
own vectors where it has some (LocationTest.scala, StochasticGraphBuilderTest.scala), against
hand-derived cases, and against the independent numpy host mirrors of mains.py."""
import json
import os

import numpy as np
import pytest

import prep_cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture
def mains(pkg):
    from locations_recommender_amd import mains
    return mains


def test_location_kats(oracle):
    """LocationTest.scala:8-27."""
    k = json.load(open(os.path.join(GOLDEN, "location_kats.json")))
    s = k["same_location"]
    assert oracle.distance_meters(s["lat"], s["lon"], s["lat"], s["lon"]) == s["expected"]
    t = k["two_distinct"]
    assert abs(oracle.distance_meters(t["lat1"], t["lon1"], t["lat2"], t["lon2"]) - t["expected"]) <= t["tolerance"]
    c = k["commutative"]
    assert oracle.distance_meters(c["lat1"], c["lon1"], c["lat2"], c["lon2"]) == \
        oracle.distance_meters(c["lat2"], c["lon2"], c["lat1"], c["lon1"])


def test_calc_ratings_hand_derived(oracle):
    """rank() leaves gaps after ties: counts 3,2,2,1 rank 1,2,2,4."""
    persons = [7] * 8 + [9] * 2
    places = [10, 10, 10, 11, 11, 12, 12, 13, 20, 21]
    for top_n, want in [(1, [10]), (2, [10, 11, 12]), (3, [10, 11, 12]), (4, [10, 11, 12, 13])]:
        p, e, r = oracle.calc_ratings(persons, places, top_n)
        assert e[p == 7].tolist() == want
        assert e[p == 9].tolist() == [20, 21]                      # both rank 1
    p, e, r = oracle.calc_ratings(persons, places, 4)
    assert r[p == 7].tolist() == [3, 2, 2, 1]
    assert oracle.calc_ratings([], [], 5)[0].size == 0
    assert oracle.calc_ratings(persons, places, 0)[0].size == 0


@pytest.mark.parametrize("seed", range(6))
def test_calc_ratings_two_restatements_agree(oracle, mains, seed):
    p, e = prep_cases.visits_case(seed, 3000, negative_ids=seed % 2 == 1)
    for top_n in (1, 3, 10, 1000):
        a = oracle.calc_ratings(p, e, top_n)
        b = mains.calc_ratings(p, e, top_n)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_calc_rating_vectors(oracle, mains):
    ids, ptr, idx, val, size = oracle.calc_rating_vectors([5, 5, 3, 5], [7, 2, 9, 4], [1, 3, 2, 6])
    assert ids.tolist() == [3, 5] and ptr.tolist() == [0, 1, 4] and idx.tolist() == [9, 2, 4, 7]
    assert val.tolist() == [2.0, 3.0, 6.0, 1.0] and size == 10
    # an index already in the person's TreeSet is not replaced (RatingVectorsBuilder.scala:43-50,68-72)
    ids, ptr, idx, val, size = oracle.calc_rating_vectors([1, 1, 1], [4, 4, 2], [10, 20, 30])
    assert idx.tolist() == [2, 4] and val.tolist() == [30.0, 10.0]
    with pytest.raises(ArithmeticError):
        oracle.calc_rating_vectors([1], [2 ** 31], [1])              # checkedCast (:36-41)
    with pytest.raises(oracle.OracleIllegalArgument):
        oracle.calc_rating_vectors([1], [-3], [1])                   # SparseVector's own require
    for seed in range(4):
        p, e, r = prep_cases.ratings_case(seed, 2000)
        a = oracle.calc_rating_vectors(p, e, r)
        b = mains.calc_rating_vectors(p, e, r)
        for x, y in zip(a[:4], b[:4]):
            assert np.array_equal(x, y)
        assert a[4] == b[4]


def test_balanced_edges_golden(oracle):
    """StochasticGraphBuilderTest.scala: every source's balanced weights sum to exactly 1.0."""
    g = json.load(open(os.path.join(GOLDEN, "graph_builder_kat.json")))
    fams = [(np.array([e[0] for e in f["edges"]]), np.array([e[1] for e in f["edges"]]), np.array([e[2] for e in f["edges"]]))
            for f in g["families"]]
    s, t, w = oracle.balanced_edges(g["betas"], fams)
    assert len(s) == sum(len(f["edges"]) for f in g["families"])
    for src in np.unique(s):
        assert w[s == src].sum() == 1.0
    at = 0
    for beta, f in zip(g["betas"], fams):
        assert np.array_equal(w[at:at + len(f[0])], f[2] * beta) and np.array_equal(s[at:at + len(f[0])], f[0])
        at += len(f[0])


def test_place_visits_hand_derived(oracle):
    """Three places on one meridian 0 m / 90 m / 220 m from a visit: the first two are visited."""
    step = 1.0 / prep_cases.METERS_PER_DEGREE
    places = {"latitude": np.array([10.0, 10.0 + 90 * step, 10.0 + 220 * step, 10.0]),
              "longitude": np.array([20.0, 20.0, 20.0, 20.0]), "region_id": np.array([1, 1, 1, 2])}
    visits = {"timestamp": np.array([100, 100, 50, 100]), "latitude": np.array([10.0, 10.0 + 150 * step, 10.0, 10.0]),
              "longitude": np.array([20.0, 20.0, 20.0, 20.0]), "region_id": np.array([1, 1, 1, 3])}
    v, p = oracle.place_visits(visits, places, visits_from=100)
    # visit 0: places 0, 1; visit 1 (150 m north): places 1 (60 m), 2 (70 m); visit 2 too old; visit 3: region without places
    assert list(zip(v.tolist(), p.tolist())) == [(0, 0), (0, 1), (1, 1), (1, 2)]
    v, p = oracle.place_visits(visits, places, visits_from=0)
    assert (2, 0) in list(zip(v.tolist(), p.tolist()))
    # Location's require()s fail the job - but only for rows that take part in a joined pair
    bad = dict(visits, latitude=np.array([10.0, 10.0, 10.0, 91.0]))
    oracle.place_visits(bad, places, visits_from=0)                  # region 3 has no places: never constructed
    bad = dict(visits, latitude=np.array([10.0, np.nan, 10.0, 10.0]))
    with pytest.raises(oracle.OracleIllegalArgument) as ei:
        oracle.place_visits(bad, places, visits_from=0)
    assert ei.value.row == ("visit", 1)
    badp = dict(places, longitude=np.array([20.0, 20.0, 20.0, 181.0]))
    oracle.place_visits(visits, badp, visits_from=0)                 # region 2 has no visits
    badp = dict(places, longitude=np.array([20.0, -180.5, 20.0, 20.0]))
    with pytest.raises(oracle.OracleIllegalArgument) as ei:
        oracle.place_visits(visits, badp, visits_from=0)
    assert ei.value.row == ("place", 1)


def test_join_cases_have_both_outcomes(oracle):
    for where in ("moscow", "antimeridian", "north_pole"):
        visits, places, visits_from = prep_cases.join_case(3, 120, 400, where)
        v, p = oracle.place_visits(visits, places, visits_from)
        assert 50 < len(v) < 400 * 120
        d = [oracle.distance_meters(visits["latitude"][a], visits["longitude"][a], places["latitude"][b], places["longitude"][b])
             for a, b in zip(v[:50], p[:50])]
        assert max(d) <= 100.0


def test_rank_recommendations_two_restatements_agree(oracle, mains):
    """f-3: the oracle's insertion-sort restatement against the numpy mirror of mains.py."""
    rng = np.random.default_rng(3)
    place_ids = 40 + np.arange(300)
    regions = rng.integers(0, 3, 300)
    ids = np.r_[rng.choice(place_ids, 150, replace=False), 5000 + np.arange(40)]      # places and non-places (persons)
    scores = np.round(rng.random(len(ids)), 2)                                          # ties on purpose
    for target in (0, 1, 2, 7):
        for limit in (0, 1, 10, 1000):
            a = oracle.rank_recommendations(ids, scores, place_ids, regions, target, limit)
            b = mains.rank_recommendations(ids, scores, place_ids, regions, target, limit)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (target, limit)
    a = oracle.rank_recommendations([42, 41, 43, 9], [0.5, 0.5, 0.9, 1.0], [41, 42, 43, 44], [1, 1, 1, 2], 1, 10)
    assert a[0].tolist() == [43, 41, 42] and a[1].tolist() == [0.9, 0.5, 0.5]             # 9 is no place; ties by id
