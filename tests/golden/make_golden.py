#!/usr/bin/env python3
"""Writes the golden fixtures under tests/golden/.

Provenance (paths relative to the reference checkout, which is NOT needed to
run this script -- the reference's known answers are transcribed as data):

* distance_kats.json  -- the 8 known-answer tests of
  recommender/src/test/scala/com/github/tashoyan/recommender/knn/DistanceTest.scala:10-60
  (inputs and expected values; `1 / math.sqrt(2)` is evaluated here in IEEE
  double, exactly as the JVM does).
* sg_kats.json        -- graph, parameters and expected rows of
  recommender/src/test/scala/com/github/tashoyan/recommender/stochastic/StochasticRecommenderTest.scala:11-21,39-94.
* knn_handmade.json   -- the reference has NO test of KnnRecommender, so these
  cases are hand-built and their expected values are computed below by an
  independent dict-based restatement of KnnRecommender.scala:27-70,76-96
  (derived by restatement, NOT by running Spark: "parity unpinned" by the
  reference).  Tie order (similarity desc, person_id asc) is this project's
  definition.

Run: python tests/golden/make_golden.py
"""
import json
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def distance_kats():
    return {
        "source": "knn/DistanceTest.scala:10-60",
        "vector_length": [
            {"name": "all vector components are zero", "size": 2, "indices": [], "values": [], "expected": 0.0},
            {"name": "one vector component is non-zero", "size": 2, "indices": [0], "values": [1.0], "expected": 1.0},
            {"name": "all vector components are non-zero", "size": 2, "indices": [0, 1], "values": [3.0, 4.0], "expected": 5.0},
            {"name": "all vector components are negative", "size": 2, "indices": [0, 1], "values": [-3.0, -4.0], "expected": 5.0},
        ],
        "cosine_similarity": [
            {"name": "collinear vectors, same direction", "size": 2,
             "v1": {"indices": [0], "values": [2.0]}, "v2": {"indices": [0], "values": [3.0]}, "expected": 1.0},
            {"name": "collinear vectors, opposite direction", "size": 2,
             "v1": {"indices": [0], "values": [2.0]}, "v2": {"indices": [0], "values": [-3.0]}, "expected": -1.0},
            {"name": "orthogonal vectors", "size": 2,
             "v1": {"indices": [0], "values": [2.0]}, "v2": {"indices": [1], "values": [3.0]}, "expected": 0.0},
            {"name": "vectors at 45 degrees", "size": 2,
             "v1": {"indices": [0], "values": [2.0]}, "v2": {"indices": [0, 1], "values": [1.0, 1.0]},
             "expected": 1 / math.sqrt(2)},
        ],
    }


def sg_kats():
    edges = [
        [1, 2, 0.4], [1, 3, 0.24], [1, 5, 0.36],
        [2, 4, 0.3], [2, 3, 0.7],
        [3, 5, 1.0],
        [4, 2, 0.3], [4, 5, 0.7],
        [5, 3, 1.0],
    ]
    return {
        "source": "stochastic/StochasticRecommenderTest.scala:11-21,39-94",
        "alpha": 0.15,
        "edges": edges,
        "cases": [
            {"name": "1 iteration", "vertex_id": 1, "epsilon": 0.01, "max_iterations": 1,
             "expected_sorted_by_probability_desc": [[5, 0.3502], [3, 0.3298], [2, 0.11900000000000001], [4, 0.051]]},
            {"name": "converge", "vertex_id": 1, "epsilon": 0.05, "max_iterations": 1000,
             "expected_sorted_by_probability_desc": [[3, 0.408242766375], [5, 0.3716171248749999],
                                                     [2, 0.055161925125], [4, 0.014978183624999999]]},
            {"name": "non-existing vertex", "vertex_id": 100, "epsilon": 0.05, "max_iterations": 1000,
             "expected_error": "IllegalArgumentException"},
        ],
    }


# ---------------------------------------------------------------------------
# independent restatement for the hand-made KNN cases (dicts, no merges)

def _norm(vec):
    s = 0.0
    for v in vec.values():
        s = s + v * v
    return math.sqrt(s)


def _cos(cand, query):
    d = 0.0
    for i in sorted(cand):
        if i in query:
            d = d + cand[i] * query[i]
    return d / (_norm(cand) * _norm(query))


def knn_expected(persons, ratings, person_id, pw, cw, k):
    """persons: {id: (place_dict, cat_dict)}; ratings: {id: {place: rating}}"""
    if person_id not in persons or not persons[person_id][0] or not persons[person_id][1]:
        return None
    qp, qc = persons[person_id]
    sims = []
    for pid, (pp, pc) in persons.items():
        if pid == person_id:
            continue
        ps = cs = None
        if pp:
            s = _cos(pp, qp)
            if s > 0:
                ps = s
        if pc:
            s = _cos(pc, qc)
            if s > 0:
                cs = s
        if ps is None and cs is None:
            continue
        sims.append((pid, (ps or 0.0) * pw + (cs or 0.0) * cw))
    sims.sort(key=lambda t: (-t[1], t[0]))
    nbs = sims[:k]
    num, den = {}, {}
    for pid, s in nbs:
        for place, r in ratings[pid].items():
            num[place] = num.get(place, 0.0) + float(r) * s
            den[place] = den.get(place, 0.0) + s
    recs = [[p, num[p] / den[p]] for p in sorted(num)]
    return [[p, s] for p, s in nbs], recs


def knn_handmade():
    D_P, D_C = 12, 4
    # id: (place vector, category vector)
    persons = {
        10: ({1: 2.0, 3: 1.0, 5: 4.0}, {0: 3.0, 1: 1.0}),          # the usual query
        11: ({1: 2.0, 3: 1.0, 5: 4.0}, {0: 3.0, 1: 1.0}),          # identical to 10 -> s = 1.0
        12: ({1: 1.0, 7: 5.0}, {0: 1.0, 2: 2.0}),                  # partial overlap
        13: ({8: 3.0, 9: 1.0}, {2: 5.0, 3: 1.0}),                  # disjoint in both -> no candidate
        14: ({8: 2.0}, {0: 2.0}),                                  # similar by category only
        15: ({1: -2.0, 5: -1.0, 6: 2.0}, {1: 4.0}),                # negative place sim, positive category
        16: ({3: 7.0}, {1: 1.0, 3: 2.0}),                          # tie partner A
        17: ({3: 2.0}, {1: 3.0, 3: 6.0}),                          # tie partner B (same direction as 16)
        18: ({5: 1.0, 10: 1.0}, {}),                               # absent from the category family
        9: ({3: 1.0}, {1: 2.0, 3: 4.0}),                           # third member of the tie, lowest id
    }
    ratings = {pid: {pl: int(abs(v)) for pl, v in pp.items()} for pid, (pp, _) in persons.items()}
    queries = [
        {"name": "k larger than the candidate count", "person_id": 10, "pw": 0.5, "cw": 0.5, "k": 100},
        {"name": "k cuts inside an exact tie (16 and 17 tie bit-for-bit; id asc wins)", "person_id": 10, "pw": 0.5, "cw": 0.5, "k": 5},
        {"name": "k cuts inside a three-way exact tie (10, 11, 13)", "person_id": 14, "pw": 0.5, "cw": 0.5, "k": 2},
        {"name": "k = 1", "person_id": 10, "pw": 0.5, "cw": 0.5, "k": 1},
        {"name": "uneven weights", "person_id": 10, "pw": 0.25, "cw": 0.75, "k": 5},
        {"name": "query similar to others by category only", "person_id": 14, "pw": 0.5, "cw": 0.5, "k": 3},
        {"name": "query with negative place values", "person_id": 15, "pw": 0.5, "cw": 0.5, "k": 10},
        {"name": "query absent from the category family", "person_id": 18, "pw": 0.5, "cw": 0.5, "k": 3,
         "expected_error": "IllegalArgumentException"},
        {"name": "unknown person", "person_id": 999, "pw": 0.5, "cw": 0.5, "k": 3,
         "expected_error": "IllegalArgumentException"},
        {"name": "weights do not sum to 1", "person_id": 10, "pw": 0.5, "cw": 0.4, "k": 3,
         "expected_error": "IllegalArgumentException"},
        {"name": "zero k", "person_id": 10, "pw": 0.5, "cw": 0.5, "k": 0,
         "expected_error": "IllegalArgumentException"},
        {"name": "weight out of (0,1)", "person_id": 10, "pw": 1.0, "cw": 0.0, "k": 3,
         "expected_error": "IllegalArgumentException"},
    ]
    for q in queries:
        if "expected_error" in q:
            continue
        nbs, recs = knn_expected(persons, ratings, q["person_id"], q["pw"], q["cw"], q["k"])
        q["expected_neighbours"] = nbs
        q["expected_recommendations"] = recs
    ids = sorted(persons)
    return {
        "source": "hand-derived; expected values by an independent restatement of "
                  "knn/KnnRecommender.scala:27-70,76-96 in make_golden.py, not by running Spark",
        "tie_order": "similarity desc, person_id asc (project definition, SURVEY.md H1)",
        "place_dim": D_P,
        "category_dim": D_C,
        "persons": [
            {"person_id": pid,
             "place": {"indices": sorted(persons[pid][0]), "values": [persons[pid][0][i] for i in sorted(persons[pid][0])]},
             "category": {"indices": sorted(persons[pid][1]), "values": [persons[pid][1][i] for i in sorted(persons[pid][1])]},
             "ratings": [[pl, ratings[pid][pl]] for pl in sorted(ratings[pid])]}
            for pid in ids
        ],
        "queries": queries,
    }


def main():
    for name, fn in (("distance_kats.json", distance_kats), ("sg_kats.json", sg_kats),
                     ("knn_handmade.json", knn_handmade)):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(fn(), f, indent=1)
            f.write("\n")
        print("wrote", name)


if __name__ == "__main__":
    main()
