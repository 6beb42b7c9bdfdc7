"""The index built on the device (csrc/knn_build.hip: locrec_knn_create / locrec_knn_create_from_device)
against the host build it replaced (LOCREC_KNN_HOST_BUILD=1) and against the oracle: identical
neighbours, similarities and recommendations for every stored format, with and without explicit
ratings, from host arrays and from device tensors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def make(pkg, d):
    return pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                        d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"],
                        d.get("r_rowptr"), d.get("r_place"), d.get("r_rating"))


def results(ix, d, rows, k):
    ids, sims, cnt = ix.query_batch(d["person_ids"][rows], 0.5, 0.5, k)
    off, places, est = ix.recommend_batch(d["person_ids"][rows[:24]], 0.5, 0.5, k)
    one = ix.query(int(d["person_ids"][rows[1]]), 0.5, 0.5, k)
    big = ix.query(int(d["person_ids"][rows[2]]), 0.5, 0.5, 2_000_000)
    lp, lc = ix.vector_lengths()
    return ids, sims, cnt, off, places, est, one[0], one[1], big[0], big[1], lp, lc


@pytest.mark.parametrize("variant", ["pack16", "ratings", "pack32", "generic", "no_ht", "tiny_head"])
def test_device_build_equals_host_build(pkg, oracle, monkeypatch, variant):
    from locations_recommender_amd import synth
    rng = np.random.default_rng(5)
    d = synth.knn_dataset(4_000, 3_000, seed=0x5EED0077)
    if variant == "ratings":
        d["r_rowptr"], d["r_place"] = d["p_rowptr"].copy(), d["p_idx"].astype(np.int64) * 7 - 1000
        d["r_rating"] = rng.integers(1, 6, size=len(d["p_idx"])).astype(np.int64)
    if variant == "pack32":
        monkeypatch.setenv("LOCREC_KNN_NO_PACK16", "1")
    if variant == "generic":
        d["p_val"] = d["p_val"] + 0.25
    if variant == "no_ht":
        monkeypatch.setenv("LOCREC_KNN_NO_HT", "1")
    if variant == "tiny_head":
        monkeypatch.setenv("LOCREC_KNN_HT_H", "8")
    rows = np.r_[np.arange(0, 4_000, 41), [3_999]]
    dev = make(pkg, d)
    got = results(dev, d, rows, 20)
    info = dev.info()
    dev.close()
    monkeypatch.setenv("LOCREC_KNN_HOST_BUILD", "1")
    host = make(pkg, d)
    want = results(host, d, rows, 20)
    assert host.info() == info
    host.close()
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    oi, os_, oc = oracle.knn_similar_batch(d, rows, 0.5, 0.5, 20, nthreads=8)
    assert np.array_equal(got[0], oi) and np.array_equal(got[1], os_) and np.array_equal(got[2], oc)


def test_create_from_device_tensors(pkg, oracle):
    """locrec_knn_create_from_device: the arrays never leave the GPU."""
    import torch
    from locations_recommender_amd import synth
    d = synth.knn_dataset(20_000, 50_000, seed=0x5EED0078)
    d["r_rowptr"], d["r_place"] = d["p_rowptr"], d["p_idx"].astype(np.int64)
    d["r_rating"] = 1 + d["r_place"] % 5
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items() if isinstance(v, np.ndarray)}
    for with_ratings in (True, False):
        r = (t["r_rowptr"], t["r_place"], t["r_rating"]) if with_ratings else (None, None, None)
        ix = pkg.KnnIndex.from_device(t["person_ids"], t["p_rowptr"], t["p_idx"], t["p_val"], d["p_dim"],
                                      t["c_rowptr"], t["c_idx"], t["c_val"], d["c_dim"], *r)
        rows = np.arange(0, 20_000, 997)
        ids, sims, cnt = ix.query_batch(d["person_ids"][rows], 0.5, 0.5, 50)
        oi, os_, oc = oracle.knn_similar_batch(d, rows, 0.5, 0.5, 50, nthreads=8)
        assert np.array_equal(ids, oi) and np.array_equal(sims, os_) and np.array_equal(cnt, oc)
        dd = d if with_ratings else {k: v for k, v in d.items() if not k.startswith("r_")}
        pid = int(d["person_ids"][rows[3]])
        places, est = ix.recommend(pid, 0.5, 0.5, 50)
        op, oe = oracle.knn_recommend(dd, pid, 0.5, 0.5, 50)
        assert np.array_equal(places, op)
        np.testing.assert_allclose(est, oe, rtol=RTOL, atol=0)
        ix.close()


def test_device_build_rejects_bad_input(pkg):
    """The device-side validation reports the same conditions, with the person's id where the host build gave it."""
    ok = {"person_ids": np.array([11, 12, 13]), "p_rowptr": np.array([0, 2, 3, 4]), "p_idx": np.array([0, 3, 1, 2], np.int32),
          "p_val": np.array([1.0, 2.0, 1.0, 5.0]), "p_dim": 4, "c_rowptr": np.array([0, 1, 2, 3]),
          "c_idx": np.array([0, 1, 0], np.int32), "c_val": np.array([1.0, 1.0, 2.0]), "c_dim": 2}
    make(pkg, ok).close()
    cases = [({"person_ids": np.array([11, 12, 11])}, "duplicate person_id 11"),
             ({"p_idx": np.array([0, 9, 1, 2], np.int32)}, r"place index 9 out of range \[0,4\)"),
             ({"p_idx": np.array([3, 0, 1, 2], np.int32)}, "place indices of person 11 not strictly ascending"),
             ({"p_val": np.array([1.0, 2.0, 0.0, 5.0])}, "place vector of person 12 has zero norm"),
             ({"c_val": np.array([1.0, np.inf, 2.0])}, "category value is not finite"),
             ({"p_rowptr": np.array([0, 3, 2, 4])}, "place rowptr not monotone at 1"),
             ({"c_rowptr": np.array([1, 1, 2, 3])}, "category rowptr must start at 0")]
    for patch, msg in cases:
        with pytest.raises(pkg.IllegalArgumentException, match=msg):
            make(pkg, {**ok, **patch})
