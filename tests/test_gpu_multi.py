"""Several devices in ONE process (include/locrec.h, csrc/multi.hip): the multi-GPU forms a JVM can
reach.  The test box has one GPU, so the device lists below name it several times - logical shards on one device: the
whole protocol (tile-wise set-up gather with peer copies, per-device streams, event ordering, the exchange kernel that
reads the other shards' buffers, query sharding over replicas) runs, with the same device on both ends of every copy.
Results must be those of the single-device entry points: bit-identical for the replicas and the all-gather form."""
import json
import os

import numpy as np
import pytest

from test_gpu_knn import RTOL, make_index, with_ratings

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_knn_replicas_equal_one_device(pkg, oracle, devices):
    from locations_recommender_amd import synth
    d = with_ratings(synth.knn_dataset(6_000, 900, seed=41))
    one = make_index(pkg, d)
    rep = pkg.KnnReplicas(devices, d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"], d["c_rowptr"],
                          d["c_idx"], d["c_val"], d["c_dim"], d["r_rowptr"], d["r_place"], d["r_rating"])
    assert rep.devices() == len(devices)
    rows = np.r_[np.arange(0, 6_000, 53), [5_999, 5, 5]]
    pids = d["person_ids"][rows]
    off, places, est = rep.recommend_batch(pids, 0.5, 0.5, 50)
    o1, p1, e1 = one.recommend_batch(pids, 0.5, 0.5, 50)
    assert np.array_equal(off, o1) and np.array_equal(places, p1) and np.array_equal(est, e1)
    for j in (0, 57, len(rows) - 1):
        oplaces, oest = oracle.knn_recommend(d, int(pids[j]), 0.5, 0.5, 50)
        assert np.array_equal(places[off[j]:off[j + 1]], oplaces)
        np.testing.assert_allclose(est[off[j]:off[j + 1]], oest, rtol=RTOL, atol=0)
    ids, sims, cnt = rep.query_batch(pids, 0.3, 0.7, 20)
    i1, s1, c1 = one.query_batch(pids, 0.3, 0.7, 20)
    assert np.array_equal(ids, i1) and np.array_equal(sims, s1) and np.array_equal(cnt, c1)
    # the shipped K through the replicas (batched large-K path), fewer queries than devices, and the error path
    off, places, est = rep.recommend_batch(pids[:2], 0.5, 0.5, 2_000_000)
    o1, p1, e1 = one.recommend_batch(pids[:2], 0.5, 0.5, 2_000_000)
    assert np.array_equal(off, o1) and np.array_equal(places, p1) and np.array_equal(est, e1)
    assert rep.recommend_batch([], 0.5, 0.5, 50)[0].tolist() == [0]
    with pytest.raises(pkg.IllegalArgumentException, match="No such person"):
        rep.recommend_batch([int(pids[0]), 10**9, int(pids[1])], 0.5, 0.5, 50)
    with pytest.raises(pkg.IllegalArgumentException, match="Sum of weights"):
        rep.query_batch(pids, 0.5, 0.4, 20)
    rep.close()
    one.close()


def test_device_list(pkg):
    from locations_recommender_amd import multi
    with pytest.raises(pkg.IllegalArgumentException, match="not one of the"):
        multi.set_devices([0, 99])
    multi.set_devices([0, 0])
    g = pkg.SgSharded(None, [1, 2, 3], [2, 3, 1], [1.0, 1.0, 1.0])      # the list given to locrec_set_devices
    assert g.info()["devices"] == 2
    g.close()
    multi.set_devices([])
    g = pkg.SgSharded(None, [1, 2, 3], [2, 3, 1], [1.0, 1.0, 1.0])      # no list: the current device
    assert g.info()["devices"] == 1
    ids, probs, it, conv = g.recommend(1, 0.15, 0.0, 2)
    assert ids.tolist() == [2, 3] and it == 2
    g.close()


@pytest.mark.parametrize("by_target", [False, True], ids=["all_reduce", "all_gather"])
@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0, 0]])
def test_sg_sharded_in_the_library(pkg, oracle, devices, by_target):
    """The reference's known answers (StochasticRecommenderTest.scala:39-94), a random graph against the oracle and the
    unsharded handle, fixed-sweep runs, and the error paths."""
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "sg_kats.json")))
    e = np.array(kat["edges"], dtype=object)
    src, dst, w = e[:, 0].astype(np.int64), e[:, 1].astype(np.int64), e[:, 2].astype(np.float64)
    g = pkg.SgSharded(devices, src, dst, w, by_target=by_target)
    for case in kat["cases"]:
        if "expected_error" in case:
            with pytest.raises(pkg.IllegalArgumentException, match="No such vertex in the graph: 100"):
                g.recommend(case["vertex_id"], 0.15, case["epsilon"], case["max_iterations"])
            continue
        ids, probs, it, conv = g.recommend(case["vertex_id"], 0.15, case["epsilon"], case["max_iterations"])
        want = sorted(case["expected_sorted_by_probability_desc"], key=lambda t: t[0])
        assert ids.tolist() == [t[0] for t in want], case["name"]
        if by_target:
            assert probs.tolist() == [t[1] for t in want], case["name"]
        else:
            np.testing.assert_allclose(probs, [t[1] for t in want], rtol=1e-12, atol=0)
    with pytest.raises(pkg.IllegalArgumentException, match="epsilon"):
        g.recommend(1, 0.15, -1.0, 3)
    g.close()
    from locations_recommender_amd import synth
    gr = synth.sg_dataset(n_persons=6_000, n_places=900, seed=29)
    src, dst, w = gr["source_id"], gr["target_id"], gr["balanced_weight"]
    whole = pkg.SgGraph(src, dst, w)
    g = pkg.SgSharded(devices, src, dst, w, by_target=by_target)
    assert g.info()["devices"] == len(devices) and g.info()["exchanged_entries"] == whole.info()["vertices"] - 6_000
    for vertex, eps, max_it in ((int(gr["first_person"]) + 9, 1e-5, 400), (41, 0.0, 6), (int(gr["first_person"]), 0.01, 20)):
        got = g.recommend(vertex, 0.15, eps, max_it)
        oi, op, oit, oconv = oracle.sg_recommend(src, dst, w, vertex, 0.15, eps, max_it)
        assert np.array_equal(got[0], oi) and got[2:] == (oit, oconv)
        np.testing.assert_allclose(got[1], op, rtol=RTOL, atol=0)
        want = whole.recommend(vertex, 0.15, eps, max_it)
        if by_target:
            assert np.array_equal(got[1], want[1]), "the all-gather form differs from the single-GPU x"
        for which in range(len(devices)):           # every device holds the same x
            other = g.fetch(which)
            assert np.array_equal(other[1], got[1]) and other[2:] == got[2:]
    g.sweeps_async(int(gr["first_person"]), 0.15, 25)
    ids, probs, it, conv = g.fetch()
    whole.sweeps_async(int(gr["first_person"]), 0.15, 25)
    wi, wp, wit, wconv = whole.fetch()
    assert (it, conv) == (25, False) and np.array_equal(ids, wi)
    if by_target:
        assert np.array_equal(probs, wp)
    else:
        np.testing.assert_allclose(probs, wp, rtol=1e-9, atol=0)
    g.close()
    whole.close()


def test_sg_sharded_cfg3_eight_logical_shards(pkg, oracle):
    """configs[4]'s literal form at configs[2]'s full size through the in-library driver: 8 logical shards, all-gather
    form, 100 sweeps - bit-identical to the unsharded handle, 1e-6 against the oracle."""
    from locations_recommender_amd import synth
    gr = synth.sg_dataset()
    src, dst, w = gr["source_id"], gr["target_id"], gr["balanced_weight"]
    v = int(gr["first_person"])
    g = pkg.SgSharded([0] * 8, src, dst, w, by_target=True)
    g.sweeps_async(v, 0.15, 100)
    ids, probs, it, conv = g.fetch()
    whole = pkg.SgGraph(src, dst, w)
    whole.sweeps_async(v, 0.15, 100)
    wi, wp, _, _ = whole.fetch()
    assert np.array_equal(ids, wi) and np.array_equal(probs, wp) and (it, conv) == (100, False)
    oi, op, _, _ = oracle.sg_recommend(src, dst, w, v, 0.15, 0.0, 100)
    assert np.array_equal(ids, oi)
    np.testing.assert_allclose(probs, op, rtol=RTOL, atol=0)
    g.close()
    whole.close()
