"""f-1 / f-3: the Parquet readers against files written here in Spark's layout
(VectorUDT struct; "parity unpinned" against a real Spark file, see mains.py), the file naming of
DataUtils.scala:52-58, and the final ranking of the mains."""
import os

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest


@pytest.fixture
def mains(pkg):
    from locations_recommender_amd import mains
    return mains


VECTOR = pa.struct([("type", pa.int8()), ("size", pa.int32()), ("indices", pa.list_(pa.int32())),
                    ("values", pa.list_(pa.float64()))])


def write_vectors(path, ids, rowptr, idx, val, dim, shuffle_seed=0, parts=2):
    """A Spark-style output directory (part files, _SUCCESS), rows in arbitrary order."""
    path.mkdir()
    order = np.random.default_rng(shuffle_seed).permutation(len(ids))
    rows = [{"type": 0, "size": dim, "indices": idx[rowptr[r]:rowptr[r + 1]].tolist(),
             "values": val[rowptr[r]:rowptr[r + 1]].tolist()} for r in order]
    t = pa.table({"person_id": pa.array(np.asarray(ids)[order], pa.int64()), "rating_vector": pa.array(rows, VECTOR)})
    step = max(1, len(ids) // parts)
    for i, first in enumerate(range(0, len(ids), step)):
        pq.write_table(t.slice(first, step), path / f"part-{i:05d}.snappy.parquet")
    (path / "_SUCCESS").write_text("")


def test_file_names(mains):
    assert mains.generate_file_name([2, 0, 2], "/d", "place_ratings") == "/d/place_ratings_region0_region2"
    assert mains.generate_file_name([7], "/d", "stochastic_graph") == "/d/stochastic_graph_region7"


def test_vector_reader_round_trip(mains, tmp_path, pkg):
    from locations_recommender_amd import synth
    d = synth.knn_dataset(500, 300, seed=5)
    write_vectors(tmp_path / "v", d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"])
    ids, rowptr, idx, val, dim = mains.load_rating_vectors(str(tmp_path / "v"))
    assert dim == d["p_dim"] and np.array_equal(ids, d["person_ids"]) and np.array_equal(rowptr, d["p_rowptr"])
    assert np.array_equal(idx, d["p_idx"]) and np.array_equal(val, d["p_val"]) and idx.dtype == np.int32


def knn_files(tmp_path, d, regions="region0_region2", shuffle=True):
    """The three Parquet sets of a region pair in Spark's layout; the ratings in shuffled row order."""
    base = tmp_path
    write_vectors(base / f"place_rating_vectors_{regions}", d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"])
    write_vectors(base / f"category_rating_vectors_{regions}", d["person_ids"], d["c_rowptr"], d["c_idx"], d["c_val"],
                  d["c_dim"], shuffle_seed=1)
    rows = np.repeat(np.arange(len(d["person_ids"])), np.diff(d["p_rowptr"]))
    perm = np.random.default_rng(2).permutation(len(rows)) if shuffle else np.arange(len(rows))
    pq.write_table(pa.table({"person_id": d["person_ids"][rows][perm], "place_id": d["p_idx"].astype(np.int64)[perm],
                             "rating": d["p_val"].astype(np.int64)[perm]}), base / f"place_ratings_{regions}")
    return [str(base / f"{p}_{regions}") for p in ("place_rating_vectors", "category_rating_vectors", "place_ratings")]


def test_native_parquet_loader_equals_the_numpy_reader(mains, tmp_path, pkg):
    """liblocrec_parquet.so (Arrow C++; include/locrec_parquet.h) against mains.py's pyarrow / numpy reader on the same
    files: part files, shuffled rows, a person missing from one family, int32 ids, and the error cases.  Both follow the
    published VectorUDT layout ("parity unpinned" against a Spark-written file: none exists in the reference tree)."""
    from locations_recommender_amd import parquet, synth
    d = synth.knn_dataset(700, 300, seed=5)
    # person 3 has no category vector, person 5 no place vector and no ratings
    keepc = np.ones(700, bool)
    keepc[3] = False
    files = knn_files(tmp_path, d)
    pv, cv, pr = files

    def rewrite_without(path, drop_pid):
        import shutil
        t = pq.read_table(path)
        shutil.rmtree(path) if os.path.isdir(path) else os.remove(path)
        keep = np.asarray(t["person_id"].to_numpy()) != drop_pid
        pq.write_table(t.filter(pa.array(keep)), path)
    import os
    rewrite_without(cv, int(d["person_ids"][3]))
    rewrite_without(pv, int(d["person_ids"][5]))
    rewrite_without(pr, int(d["person_ids"][5]))
    got = parquet.read_knn(pv, cv, pr)
    pp, prp, pidx, pval, pdim = mains.load_rating_vectors(pv)
    cp, crp, cidx, cval, cdim = mains.load_rating_vectors(cv)
    rp, rplace, rrating = mains.load_place_ratings(pr)
    ids = np.union1d(np.union1d(pp, cp), rp)
    assert np.array_equal(got["person_ids"], ids) and got["p_dim"] == pdim and got["c_dim"] == cdim
    assert np.array_equal(got["p_rowptr"], mains._align(ids, pp, prp, pidx, pval)[0])
    assert np.array_equal(got["c_rowptr"], mains._align(ids, cp, crp, cidx, cval)[0])
    assert np.array_equal(got["p_idx"], pidx) and np.array_equal(got["p_val"], pval)
    assert np.array_equal(got["c_idx"], cidx) and np.array_equal(got["c_val"], cval) and got["p_idx"].dtype == np.int32
    rows = np.searchsorted(ids, rp)
    order = np.argsort(rows, kind="stable")
    assert np.array_equal(got["r_place"], rplace[order]) and np.array_equal(got["r_rating"], rrating[order])
    assert np.array_equal(np.diff(got["r_rowptr"]), np.bincount(rows, minlength=len(ids)))
    # edges: int32 target ids are widened, file order kept
    pq.write_table(pa.table({"source_id": pa.array([5, 6, 5], pa.int64()), "target_id": pa.array([1, 2, 3], pa.int32()),
                             "balanced_weight": [0.5, 1.0, 0.5]}), tmp_path / "stochastic_graph_region1")
    s, t, w = parquet.read_edges(str(tmp_path / "stochastic_graph_region1"))
    assert s.tolist() == [5, 6, 5] and t.tolist() == [1, 2, 3] and w.tolist() == [0.5, 1.0, 0.5]
    # errors carry a message and the reference-side exception type
    with pytest.raises(pkg.IllegalArgumentException, match="no Parquet file"):
        parquet.read_edges(str(tmp_path / "nothing_here"))
    with pytest.raises(pkg.IllegalArgumentException, match="has no column source_id"):
        parquet.read_edges(pr)
    rows2 = [{"type": 0, "size": 5, "indices": [1], "values": [1.0]}, {"type": 0, "size": 6, "indices": [2], "values": [1.0]}]
    pq.write_table(pa.table({"person_id": pa.array([1, 2], pa.int64()), "rating_vector": pa.array(rows2, VECTOR)}), tmp_path / "bad.parquet")
    with pytest.raises(pkg.IllegalArgumentException, match="different sizes"):
        parquet.read_knn(str(tmp_path / "bad.parquet"), cv, pr)
    dense = [{"type": 1, "size": 2, "indices": None, "values": [1.0, 2.0]}]
    pq.write_table(pa.table({"person_id": pa.array([1], pa.int64()), "rating_vector": pa.array(dense, VECTOR)}), tmp_path / "dense.parquet")
    with pytest.raises(pkg.IllegalArgumentException, match="dense rating vectors|null entry"):
        parquet.read_knn(str(tmp_path / "dense.parquet"), cv, pr)


@pytest.mark.gpu
def test_native_parquet_to_device(mains, tmp_path, pkg, oracle):
    """locrec_knn_create_from_parquet / locrec_sg_create_from_parquet: files -> device handles with no Python on the
    data path; requests against the oracle."""
    from locations_recommender_amd import parquet, synth
    d = synth.knn_dataset(2_000, 300, seed=6)
    pv, cv, pr = knn_files(tmp_path, d)
    d["r_rowptr"], d["r_place"], d["r_rating"] = d["p_rowptr"], d["p_idx"].astype(np.int64), d["p_val"].astype(np.int64)
    ix = parquet.knn_index(pv, cv, pr)
    for row in (0, 321, 1_999):
        pid = int(d["person_ids"][row])
        places, est = ix.recommend(pid, 0.5, 0.5, 50)
        oplaces, oest = oracle.knn_recommend(d, pid, 0.5, 0.5, 50)
        assert np.array_equal(places, oplaces)
        np.testing.assert_allclose(est, oest, rtol=1e-6, atol=0)
    ix.close()
    g = synth.sg_dataset(n_persons=1_500, n_places=300, seed=8)
    pq.write_table(pa.table({"source_id": g["source_id"], "target_id": g["target_id"], "balanced_weight": g["balanced_weight"]}),
                   tmp_path / "stochastic_graph_region0_region2")
    sg = parquet.sg_graph(str(tmp_path / "stochastic_graph_region0_region2"))
    v = int(g["first_person"]) + 3
    ids, probs, it, conv = sg.recommend(v, 0.15, 0.01, 20)
    oi, op, oit, oconv = oracle.sg_recommend(g["source_id"], g["target_id"], g["balanced_weight"], v, 0.15, 0.01, 20)
    assert np.array_equal(ids, oi) and (it, conv) == (oit, oconv)
    np.testing.assert_allclose(probs, op, rtol=1e-6, atol=0)
    sg.close()


def test_vector_reader_rejects_mixed_sizes(mains, tmp_path):
    rows = [{"type": 0, "size": 5, "indices": [1], "values": [1.0]}, {"type": 0, "size": 6, "indices": [2], "values": [1.0]}]
    pq.write_table(pa.table({"person_id": pa.array([1, 2], pa.int64()), "rating_vector": pa.array(rows, VECTOR)}),
                   tmp_path / "bad.parquet")
    with pytest.raises(Exception, match="different sizes"):
        mains.load_rating_vectors(str(tmp_path / "bad.parquet"))


def test_edges_ratings_places_readers(mains, tmp_path):
    pq.write_table(pa.table({"source_id": pa.array([5, 6], pa.int64()), "target_id": pa.array([1, 2], pa.int32()),
                             "balanced_weight": [0.5, 1.0]}), tmp_path / "stochastic_graph_region1")
    s, t, w = mains.load_stochastic_graph(mains.generate_file_name([1], str(tmp_path), "stochastic_graph"))
    assert s.dtype == t.dtype == np.int64 and t.tolist() == [1, 2] and w.tolist() == [0.5, 1.0]
    pq.write_table(pa.table({"person_id": [9, 9], "place_id": [40, 41], "rating": [3, 1]}), tmp_path / "pr.parquet")
    assert [a.tolist() for a in mains.load_place_ratings(str(tmp_path / "pr.parquet"))] == [[9, 9], [40, 41], [3, 1]]
    pq.write_table(pa.table({"id": [40, 41], "latitude": [0.0, 1.0], "longitude": [0.0, 1.0],
                             "region_id": pa.array([0, 1], pa.int32())}), tmp_path / "places_sample")
    pid, reg = mains.load_places(str(tmp_path))
    assert pid.tolist() == [40, 41] and reg.tolist() == [0, 1] and reg.dtype == np.int64


def test_final_ranking(mains):
    """KnnRecommenderMain.scala:96-101: only places of the target region survive the join; order by
    score desc; limit.  Ties by id ascending (the project's rule)."""
    place_ids, regions = [40, 41, 42, 43, 44], [0, 1, 1, 1, 0]
    ids = [44, 43, 42, 41, 2040, 7]                 # a person id and a category id among the SG rows
    scores = [0.9, 0.2, 0.5, 0.5, 0.99, 0.8]
    got = mains.rank_recommendations(ids, scores, place_ids, regions, 1, 10)
    assert got[0].tolist() == [41, 42, 43] and got[1].tolist() == [0.5, 0.5, 0.2]
    got = mains.rank_recommendations(ids, scores, place_ids, regions, 1, 2)
    assert got[0].tolist() == [41, 42]
    assert len(mains.rank_recommendations(ids, scores, place_ids, regions, 9, 5)[0]) == 0


@pytest.mark.gpu
def test_parquet_to_device_end_to_end(mains, tmp_path, pkg, oracle):
    """f-1 + hot path + f-3: files -> KnnIndex / SgGraph on the device -> the mains' top-N."""
    from locations_recommender_amd import synth
    d = synth.knn_dataset(2_000, 300, seed=6)
    regions = [0, 2]
    base = tmp_path
    write_vectors(base / "place_rating_vectors_region0_region2", d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"])
    write_vectors(base / "category_rating_vectors_region0_region2", d["person_ids"], d["c_rowptr"], d["c_idx"], d["c_val"],
                  d["c_dim"], shuffle_seed=1)
    rows = np.repeat(np.arange(2_000), np.diff(d["p_rowptr"]))
    d["r_rowptr"], d["r_place"], d["r_rating"] = d["p_rowptr"], d["p_idx"].astype(np.int64), d["p_val"].astype(np.int64)
    perm = np.random.default_rng(2).permutation(len(rows))
    pq.write_table(pa.table({"person_id": d["person_ids"][rows][perm], "place_id": d["r_place"][perm],
                             "rating": d["r_rating"][perm]}), base / "place_ratings_region0_region2")
    ix = mains.knn_index_from_parquet(str(base), [2, 0])
    pid = int(d["person_ids"][321])
    places, est = ix.recommend(pid, 0.5, 0.5, 50)
    oplaces, oest = oracle.knn_recommend(d, pid, 0.5, 0.5, 50)
    assert np.array_equal(places, oplaces)
    np.testing.assert_allclose(est, oest, rtol=1e-6, atol=0)
    place_ids = np.arange(40, 340)
    top_ids, top = mains.rank_recommendations(places, est, place_ids, place_ids % 3, 2, 10)
    assert len(top_ids) == 10 and np.all(top_ids % 3 == 2) and np.all(np.diff(top) <= 0)
    ix.close()
    g = synth.sg_dataset(n_persons=1_500, n_places=300, seed=8)
    pq.write_table(pa.table({"source_id": g["source_id"], "target_id": g["target_id"],
                             "balanced_weight": g["balanced_weight"]}), base / "stochastic_graph_region0_region2")
    sg = mains.sg_graph_from_parquet(str(base), regions)
    v = int(g["first_person"]) + 3
    ids, probs, it, conv = sg.recommend(v, 0.15, 0.01, 20)
    oi, op, oit, oconv = oracle.sg_recommend(g["source_id"], g["target_id"], g["balanced_weight"], v, 0.15, 0.01, 20)
    assert np.array_equal(ids, oi) and (it, conv) == (oit, oconv)
    top_ids, top = mains.rank_recommendations(ids, probs, place_ids, place_ids % 3, 0, 10)
    want = sorted(((p, i) for i, p in zip(oi, op) if 40 <= i < 340 and i % 3 == 0), key=lambda t: (-t[0], t[1]))[:10]
    assert top_ids.tolist() == [i for _, i in want]
    sg.close()


def graph_builder_kat():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graph_builder_kat.json")) as f:
        return json.load(f)


def test_balanced_weights_reference_kat(mains):
    """StochasticGraphBuilderTest.scala:11-66 as the reference writes it: for every source the
    balanced weights sum to EXACTLY 1.0; plus the order of the union."""
    g = graph_builder_kat()
    fams = [tuple(np.array(c) for c in zip(*f["edges"])) for f in g["families"]]
    s, t, w = mains.build_with_balanced_weights(g["betas"], fams)
    assert s.tolist() == [1, 1, 1, 3, 5, 2, 4, 2, 4] and t.tolist() == [2, 3, 5, 5, 3, 4, 2, 3, 5]
    for source in np.unique(s):
        total = 0.0
        for x in w[s == source]:          # Spark's sum: left to right over the partition
            total += x
        assert total == g["expected_sum_per_source"], f"For source {source} sum must be 1.0"
    with pytest.raises(Exception):
        mains.build_with_balanced_weights(g["betas"][:2], fams)


@pytest.mark.gpu
def test_builder_output_feeds_the_device_path(mains, pkg, oracle):
    g = graph_builder_kat()
    fams = [tuple(np.array(c) for c in zip(*f["edges"])) for f in g["families"]]
    s, t, w = mains.build_with_balanced_weights(g["betas"], fams)
    sg = pkg.SgGraph(s, t, w)
    ids, probs, it, conv = sg.recommend(1, 0.15, 1e-6, 200)
    oi, op, oit, oconv = oracle.sg_recommend(s, t, w, 1, 0.15, 1e-6, 200)
    assert np.array_equal(ids, oi) and (it, conv) == (oit, oconv)
    np.testing.assert_allclose(probs, op, rtol=1e-6, atol=0)
    assert abs(probs.sum() + 0.0 - (1.0 - 0.0)) < 1.0  # a distribution over the other vertices (person 1 excluded)
    sg.close()


def test_calc_ratings_rank_with_ties(mains):
    """RatingsBuilder.scala:32-48 on a hand-made visit log (no reference test exists: unpinned).
    person 1: place 10 x3, 11 x2, 12 x2, 13 x1 -> ranks 1, 2, 2, 4; top 2 keeps 10, 11, 12 (the tie
    straddles the limit and rank() keeps it whole); top 3 the same; top 4 adds 13."""
    persons = [1, 1, 1, 1, 1, 1, 1, 1, 2, 2]
    places = [10, 11, 10, 12, 11, 10, 12, 13, 10, 10]
    for top_n, want in ((1, [10]), (2, [10, 11, 12]), (3, [10, 11, 12]), (4, [10, 11, 12, 13])):
        p, e, r = mains.calc_ratings(persons, places, top_n)
        assert e[p == 1].tolist() == want, top_n
        assert r[p == 1].tolist() == [3, 2, 2, 1][:len(want)]
        assert e[p == 2].tolist() == [10] and r[p == 2].tolist() == [2]
    # brute force on random data
    rng = np.random.default_rng(0)
    persons, places = rng.integers(0, 30, 3000), rng.integers(0, 12, 3000)
    p, e, r = mains.calc_ratings(persons, places, 4)
    for person in range(30):
        ents, cnts = np.unique(places[persons == person], return_counts=True)
        keep = [(int(a), int(c)) for a, c in zip(ents, cnts) if 1 + np.sum(cnts > c) <= 4]
        assert list(zip(e[p == person].tolist(), r[p == person].tolist())) == keep, person


def test_calc_rating_vectors(mains):
    ids, rowptr, idx, val, size = mains.calc_rating_vectors([5, 5, 3, 5], [7, 2, 9, 4], [1, 3, 2, 6])
    assert ids.tolist() == [3, 5] and rowptr.tolist() == [0, 1, 4] and size == 10
    assert idx.tolist() == [9, 2, 4, 7] and val.tolist() == [2.0, 3.0, 6.0, 1.0] and idx.dtype == np.int32
    with pytest.raises(ArithmeticError, match="Index out of Int range"):
        mains.calc_rating_vectors([1], [2**31], [1])
