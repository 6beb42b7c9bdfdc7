"""BASELINE.json's configurations at FULL size against the oracle, in exactly the regimes bench.py
times ("next round" item 1):

  configs[1]  the benchmark's own step - recommend_range_async of 16,384 queries at 1M x 100k -
              sampled queries bit-exact (ids, similarities) and 1e-6 (ratings) against the oracle
  configs[3]  10M persons x 1M places on one GPU (what each of the 8 ranks holds), sampled queries
  configs[4]  the per-GPU share of the 64-graph job: 8 full-size graphs iterated concurrently,
              each against oracle_sg_recommend at 100 sweeps
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-6
THREADS = max(1, min(32, len(os.sched_getaffinity(0))))


def bench_knn_input(n, places, seed):
    """bench.py's KNN input: synth data + placeRatings derived from the place index."""
    from locations_recommender_amd import synth
    d = synth.knn_dataset(n, places, seed) if n <= 2_000_000 else \
        synth.knn_dataset_parallel(n, places, seed, workers=min(16, THREADS))
    d["r_rowptr"] = d["p_rowptr"]
    d["r_place"] = d["p_idx"].astype(np.int64)
    d["r_rating"] = 1 + d["r_place"] % 5
    return d


def check_sampled_queries(ix, d, oracle, first, nq, k, sample, ids, sims, cnt, rec=None):
    """Rows [first, first + nq) were the queries; compare `sample` of them with the oracle."""
    qids = ix.row_person_ids(first, nq)
    # person ids are contiguous in the synthetic sets: input row = id - first id
    qrows = (qids[sample] - int(d["person_ids"][0])).astype(np.int64)
    assert np.array_equal(d["person_ids"][qrows], qids[sample])
    oi, os_, oc = oracle.knn_similar_batch(d, qrows, 0.5, 0.5, k, nthreads=THREADS)
    for j, s in enumerate(sample):
        c = int(cnt[s])
        assert c == int(oc[j]), (s, c, int(oc[j]))
        assert np.array_equal(ids[s, :c], oi[j, :c]), f"neighbour ids of batch query {s} differ from the oracle"
        assert np.array_equal(sims[s, :c], os_[j, :c]), f"similarities of batch query {s} differ bit-wise"
    if rec is not None:
        off, places, est = rec
        for s in sample[:: max(1, len(sample) // 6)]:
            op, oe = oracle.knn_recommend(d, int(qids[s]), 0.5, 0.5, k)
            assert np.array_equal(places[off[s]:off[s + 1]], op), s
            np.testing.assert_allclose(est[off[s]:off[s + 1]], oe, rtol=RTOL, atol=0)


def test_cfg2_benchmarked_step_matches_oracle(pkg, oracle):
    """bench.py's step: recommend_range_async(first, 16384, 0.5, 0.5, 50) at configs[1] - 1,024 tiles
    x candidate chunks, long barrier-free phases, in-kernel interval replays - on the batches the
    benchmark visits first (golden-ratio stride over the length quantiles) and on the shortest rows
    (tie-heavy: the intervals that overrun a survivor queue and are replayed)."""
    from locations_recommender_amd import shard
    n, places, k, batch = 1_000_000, 100_000, 50, 16_384
    d = bench_knn_input(n, places, 0x5EED0002)
    ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                      d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"], d["r_rowptr"], d["r_place"], d["r_rating"])
    nb = n // batch
    sample = np.unique(np.r_[np.linspace(0, batch - 1, 14).astype(np.int64), [1, batch - 2]])
    assert len(sample) >= 16
    replays0 = ix.replayed_intervals()
    for step_no, b in enumerate((shard.query_batch_of(1, 0, 1, nb), 0, nb - 1)):
        ix.profile_enable(True)
        allocs0 = pkg._lib.device_allocations()
        ix.recommend_range_async(b * batch, batch, 0.5, 0.5, k)
        ix.synchronize()
        # work buffers settle with the first batch (the hits buffer is sized for the widest window of rows):
        # a later, larger batch must not free and allocate - a device-wide synchronisation - inside a step
        assert step_no == 0 or pkg._lib.device_allocations() == allocs0, "a steady-state step allocated device memory"
        _, launches = ix.profile_read()
        ids, sims, cnt = ix.fetch_topk(batch, k)
        rec = ix.fetch_recommend(batch)
        _, extra = ix.profile_read()
        ix.profile_enable(False)
        assert launches == 1 and extra == 0, "reading the batch back launched another scan"
        assert np.all(cnt == k)
        check_sampled_queries(ix, d, oracle, b * batch, batch, k, sample, ids, sims, cnt, rec)
    print("flush intervals replayed in kernel over the three batches:", ix.replayed_intervals() - replays0)
    ix.close()


def test_cfg2_batched_path_at_the_shipped_k(pkg, oracle):
    """VERDICT r02 item 5's condition: makeRecommendationsBatch at the SHIPPED K (bin/knn_recommender.sh:35,
    --k-nearest 2000000 >= N: every positive-similarity person is a neighbour, KnnRecommender.scala:47-48) at
    configs[1]'s size, in the form bench.py's knn_large_k_batched leg times (recommend_range_async of 64 queries =
    four tiles of 16): sampled queries against the oracle (places ==, estimates 1e-6), and the batch form
    (recommend_batch, person ids) identical to it."""
    n, places, k, nq = 1_000_000, 100_000, 2_000_000, 64
    d = bench_knn_input(n, places, 0x5EED0002)
    ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                      d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"], d["r_rowptr"], d["r_place"], d["r_rating"])
    first = 333_333
    ix.recommend_range_async(first, nq, 0.5, 0.5, k)
    off, rplaces, est = ix.fetch_recommend(nq)
    qids = ix.row_person_ids(first, nq)
    assert off[-1] > 0 and np.all(np.diff(off) > 0)
    for s in (0, 15, 16, 41, 63):
        op, oe = oracle.knn_recommend(d, int(qids[s]), 0.5, 0.5, k)
        assert np.array_equal(rplaces[off[s]:off[s + 1]], op), s
        np.testing.assert_allclose(est[off[s]:off[s + 1]], oe, rtol=RTOL, atol=0)
    boff, bplaces, best = ix.recommend_batch(qids[:20], 0.5, 0.5, k)
    assert np.array_equal(boff, off[:21]) and np.array_equal(bplaces, rplaces[:off[20]]) and np.array_equal(best, est[:off[20]])
    ix.close()


@pytest.mark.parametrize("k", [1, 7, 600, 1024])
def test_cfg2_seeded_scan_other_k(pkg, oracle, k, monkeypatch):
    """The threshold-seeding pass at full size for other K: K = 1 (the seed is the 2nd largest lane maximum),
    K + 1 above the 512 lane maxima (no seed), and the largest batched K; also with the pass switched off."""
    n, places, nq = 1_000_000, 100_000, 512
    d = bench_knn_input(n, places, 0x5EED0002)
    sample = np.linspace(0, nq - 1, 8).astype(np.int64)
    got = []
    for no_seed in (False, True):
        if no_seed:
            monkeypatch.setenv("LOCREC_KNN_NO_SEED", "1")
        ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                          d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"])
        first = 700_000
        ix.topk_range_async(first, nq, 0.5, 0.5, k)
        ids, sims, cnt = ix.fetch_topk(nq, k)
        got.append((ids, sims, cnt))
        if not no_seed:
            check_sampled_queries(ix, d, oracle, first, nq, k, sample, ids, sims, cnt)
        ix.close()
    assert all(np.array_equal(a, b) for a, b in zip(*got)), "seeded and unseeded scans differ"


def test_cfg4_full_size_one_gpu(pkg, oracle):
    """configs[3] as one rank sees it: the FULL 10M x 1M candidate set on one GPU, one 16,384-query
    batch (device-resident form) and the single-request operator, sampled against the oracle."""
    n, places, k, batch = 10_000_000, 1_000_000, 50, 16_384
    d = bench_knn_input(n, places, 0x5EED0004)
    ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                      d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"])
    assert ix.info()["packed"]
    first = 7 * batch * 11
    ix.topk_range_async(first, batch, 0.5, 0.5, k)
    ids, sims, cnt = ix.fetch_topk(batch, k)
    sample = np.linspace(0, batch - 1, 16).astype(np.int64)
    check_sampled_queries(ix, d, oracle, first, batch, k, sample, ids, sims, cnt)
    pid = int(d["person_ids"][n // 2])
    a, b = ix.query(pid, 0.5, 0.5, k)
    o1, o2, o3 = oracle.knn_similar_batch(d, np.array([n // 2], np.int64), 0.5, 0.5, k, nthreads=1)
    assert np.array_equal(a, o1[0, :int(o3[0])]) and np.array_equal(b, o2[0, :int(o3[0])])
    ix.close()


def test_cfg5_per_gpu_share_full_size(pkg, oracle):
    """configs[4]'s per-GPU share: 8 full-size graphs (seeds 0x5EED0500 + g, ~4.8M edges each)
    resident together and iterated CONCURRENTLY on their own streams for 100 sweeps, each against
    oracle_sg_recommend (epsilon = 0, 100 iterations)."""
    import torch
    from locations_recommender_amd import synth
    graphs, handles, streams, targets = [], [], [], []
    for i in range(8):
        g = synth.sg_dataset(seed=0x5EED0500 + i)
        h = pkg.SgGraph(g["source_id"], g["target_id"], g["balanced_weight"])
        st = torch.cuda.Stream()
        h.set_stream(st.cuda_stream)
        graphs.append(g), handles.append(h), streams.append(st), targets.append(int(g["first_person"]) + i)
    for h, v in zip(handles, targets):
        h.sweeps_async(v, 0.15, 100)              # all enqueued before any is read back
    for g, h, v in zip(graphs, handles, targets):
        ids, probs, it, conv = h.fetch()
        oi, op, oit, oconv = oracle.sg_recommend(g["source_id"], g["target_id"], g["balanced_weight"], v, 0.15, 0.0, 100)
        assert np.array_equal(ids, oi) and (it, conv) == (100, False)
        np.testing.assert_allclose(probs, op, rtol=RTOL, atol=0)
    for h in handles:
        h.close()


@pytest.mark.parametrize("by_target", [False, True], ids=["rows_of_P_all_reduce", "rows_of_Pt_all_gather"])
def test_cfg3_row_sharded_full_size(pkg, oracle, by_target):
    """configs[4] in its literal wording - "SpMV rows sharded 8 x MI355X with all-reduce of x each iteration" - at
    configs[2]'s FULL size (E ~ 4.8 M, seed 0x5EED0003: uint16 columns at their boundary, ~19 k pieces, the long
    category rows cut over all shards), 8 shards emulated on the one GPU in both forms
    (locrec_sg_create_sharded = rows of P + all-reduce, locrec_sg_create_target_sharded = rows of P^T +
    all-gather): 100 fixed sweeps and the shipped epsilon = 0.01, against oracle_sg_recommend
    (StochasticRecommender.scala:108-141) within 1e-6; the all-gather form bit-identical to the unsharded handle."""
    from locations_recommender_amd import synth
    from test_gpu_sg import sharded_recommend
    g = synth.sg_dataset()
    src, dst, w = g["source_id"], g["target_id"], g["balanced_weight"]
    assert 4_500_000 < len(src) < 5_500_000
    v = int(g["first_person"])
    whole = pkg.SgGraph(src, dst, w)
    # 100 sweeps, never stopping early (what sweeps_async / bench.py's sg leg runs)
    ids, probs, it, conv = sharded_recommend(pkg, src, dst, w, 8, v, 0.15, 0.0, 100, by_target=by_target,
                                             fixed_sweeps=True)
    oi, op, _, _ = oracle.sg_recommend(src, dst, w, v, 0.15, 0.0, 100)
    assert (it, conv) == (100, False) and np.array_equal(ids, oi)
    np.testing.assert_allclose(probs, op, rtol=RTOL, atol=0)
    whole.sweeps_async(v, 0.15, 100)
    wi, wp, wit, wconv = whole.fetch()
    assert np.array_equal(ids, wi) and (wit, wconv) == (100, False)
    if by_target:
        assert np.array_equal(probs, wp), "the all-gather form differs from the single-GPU x at full size"
    else:
        np.testing.assert_allclose(probs, wp, rtol=1e-9, atol=0)
    # the shipped parameters (bin/stochastic_recommender.sh): early exit at the reference's counter
    for vertex in (v, v + 12_345, int(dst[0])):
        ids, probs, it, conv = sharded_recommend(pkg, src, dst, w, 8, vertex, 0.15, 0.01, 20, by_target=by_target)
        oi, op, oit, oconv = oracle.sg_recommend(src, dst, w, vertex, 0.15, 0.01, 20)
        assert (it, conv) == (oit, oconv) and np.array_equal(ids, oi)
        np.testing.assert_allclose(probs, op, rtol=RTOL, atol=0)
        if by_target:
            wi, wp, wit, wconv = whole.recommend(vertex, 0.15, 0.01, 20)
            assert np.array_equal(ids, wi) and np.array_equal(probs, wp) and (it, conv) == (wit, wconv)
    whole.close()


def test_cfg2_candidate_sharded_request_full_size(pkg, oracle):
    """SURVEY 8e "KNN single request, candidates sharded" at configs[1]'s full size (1 M x 100 k): the request's
    scan cut into 8 candidate shards (locrec_knn_query_shard, emulated on the one GPU), local lists merged by
    (similarity desc, id asc) - bit-identical to the unsharded request and to the oracle
    (KnnRecommender.scala:27-49); makeRecommendations0 from the merged list equals locrec_knn_recommend."""
    from test_gpu_knn import sharded_request
    n, places, k = 1_000_000, 100_000, 50
    d = bench_knn_input(n, places, 0x5EED0002)
    ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                      d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"], d["r_rowptr"], d["r_place"], d["r_rating"])
    rows = np.array([0, 17, 333_333, 500_000, 777_777, n - 1], np.int64)
    oids, osims, ocnt = oracle.knn_similar_batch(d, rows, 0.5, 0.5, k, nthreads=THREADS)
    for j, r in enumerate(rows):
        pid = int(d["person_ids"][r])
        for shards in (8, 2):
            ids, sims = sharded_request(pkg, ix, pid, 0.5, 0.5, k, shards)
            assert np.array_equal(ids, oids[j][:ocnt[j]]) and np.array_equal(sims, osims[j][:ocnt[j]]), (r, shards)
        uid, usim = ix.query(pid, 0.5, 0.5, k)
        assert np.array_equal(ids, uid) and np.array_equal(sims, usim)
        places_, est = ix.recommend_neighbours(ids, sims)
        uplaces, uest = ix.recommend(pid, 0.5, 0.5, k)
        assert np.array_equal(places_, uplaces) and np.array_equal(est, uest)
    ix.close()
