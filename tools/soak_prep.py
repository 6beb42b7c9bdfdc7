#!/usr/bin/env python3
"""Randomised soak of the producers (csrc/prep.hip) against the oracle: SOAK_SEEDS seeds (dev tool)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
import oracle_binding as ob  # noqa: E402
import prep_cases  # noqa: E402

pkg = graft.load_package()
prep = pkg.prep
bad = 0
for seed in range(int(os.environ.get("SOAK_SEEDS", "200"))):
    rng = np.random.default_rng(70_000 + seed)
    n = int(rng.choice([1, 2, 63, 64, 65, 500, 4000, 30000]))
    p, e = prep_cases.visits_case(seed, n, persons=int(rng.integers(1, 60)), entities=int(rng.integers(1, 80)),
                                  negative_ids=bool(rng.integers(0, 2)))
    top_n = int(rng.choice([1, 2, 3, 10, 100]))
    a, b = prep.calc_ratings(p, e, top_n), ob.calc_ratings(p, e, top_n)
    ok = all(np.array_equal(x, y) for x, y in zip(a, b))
    pp, ee, rr = prep_cases.ratings_case(seed, n, persons=int(rng.integers(1, 40)), entities=int(rng.integers(1, 300)))
    a, b = prep.calc_rating_vectors(pp, ee, rr), ob.calc_rating_vectors(pp, ee, rr)
    ok = ok and all(np.array_equal(x, y) for x, y in zip(a[:4], b[:4])) and a[4] == b[4]
    nplaces = int(rng.integers(1, 400))
    place_ids = rng.permutation(40 + np.arange(nplaces)).astype(np.int64)
    regions = rng.integers(0, 3, nplaces).astype(np.int64)
    ids = rng.integers(30, 40 + nplaces + 20, int(rng.integers(1, 500))).astype(np.int64)
    ids = np.unique(ids)
    scores = np.round(rng.random(len(ids)), 2)
    lim = int(rng.choice([0, 1, 5, 1000]))
    a, b = prep.rank_recommendations(ids, scores, place_ids, regions, 1, lim), ob.rank_recommendations(ids, scores, place_ids, regions, 1, lim)
    ok = ok and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    where = ["moscow", "equator", "antimeridian", "antimeridian_west", "north_pole", "south_pole"][seed % 6]
    visits, places, vf = prep_cases.join_case(seed, int(rng.integers(1, 300)), int(rng.integers(1, 1500)), where,
                                              lat_span=float(rng.choice([0.002, 0.01, 0.05])))
    radius = float(rng.choice([30.0, 100.0, 100.0, 400.0]))
    wv, wp = ob.place_visits(visits, places, vf, radius)
    got = prep.calc_place_visits(visits, places, vf, radius)
    same = len(got["place_id"]) == len(wp) and np.array_equal(got["place_id"], places["id"][wp]) and \
        np.array_equal(got["person_id"], visits["person_id"][wv]) and np.array_equal(got["timestamp"], visits["timestamp"][wv])
    if not same:   # only pairs within 1e-6 m of the radius may differ
        g = set(zip(got["person_id"].tolist(), got["timestamp"].tolist(), got["place_id"].tolist()))
        w = set(zip(visits["person_id"][wv].tolist(), visits["timestamp"][wv].tolist(), places["id"][wp].tolist()))
        row = {int(i): r for r, i in enumerate(places["id"])}
        for person, ts, place in g ^ w:
            rows = np.flatnonzero((visits["person_id"] == person) & (visits["timestamp"] == ts))
            d = min(abs(ob.distance_meters(visits["latitude"][i], visits["longitude"][i], places["latitude"][row[place]],
                                           places["longitude"][row[place]]) - radius) for i in rows)
            if d >= 1e-6:
                ok = False
    if not ok:
        bad += 1
        print("MISMATCH at seed", seed, flush=True)
print("soak done, mismatches:", bad, flush=True)
sys.exit(1 if bad else 0)
