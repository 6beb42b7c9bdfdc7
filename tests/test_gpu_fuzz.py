"""Randomised parity sweep of the KNN path against the oracle: many small indexes of random shape
(dimension, row lengths, value ranges, K, weights), under the tuning switches that select the
different kernel paths (head / tail form at several head widths vs the row scan, hashed vs direct panel, PACK16 / PACK32 /
GENERIC, popularity split on and off, tile width, barrier-free insertion on and off).  Deterministic seeds; a failure prints its case."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SWITCHES = [
    {},                                                             # head / tail form wherever it is legal
    {"LOCREC_KNN_HT_H": "8"},                                       # almost everything in the inverted tail
    {"LOCREC_KNN_HT_H": "64", "LOCREC_KNN_NO_FAST": "1"},
    {"LOCREC_KNN_HT_H": "2048", "LOCREC_KNN_FLUSH": "2", "LOCREC_KNN_ENTER": "512"},
    {"LOCREC_KNN_NO_HT": "1"},                                      # the row scan (MODE 1 / 2) from here on
    {"LOCREC_KNN_NO_HT": "1", "LOCREC_KNN_FORCE_HASH": "1"},
    {"LOCREC_KNN_NO_HT": "1", "LOCREC_KNN_FORCE_HASH": "1", "LOCREC_KNN_NO_POP": "1"},
    {"LOCREC_KNN_NO_PACK16": "1", "LOCREC_KNN_FORCE_HASH": "1"},
    {"LOCREC_KNN_FORCE_GENERIC": "1"},
    {"LOCREC_KNN_NO_HT": "1", "LOCREC_KNN_QT": "8", "LOCREC_KNN_FORCE_HASH": "1"},
    {"LOCREC_KNN_NO_HT": "1", "LOCREC_KNN_NO_FAST": "1"},
    {"LOCREC_KNN_NO_HT": "1", "LOCREC_KNN_NO_SINGLE": "1", "LOCREC_KNN_FORCE_HASH": "1", "LOCREC_KNN_POP_H": "64"},
    {"LOCREC_KNN_NO_HT": "1", "LOCREC_KNN_FLUSH": "2", "LOCREC_KNN_ENTER": "512"},
    {"LOCREC_KNN_NO_SINGLE": "1", "LOCREC_KNN_HT_H": "16"},         # single requests through the tiled head / tail form
    {"LOCREC_KNN_HT_V1": "1", "LOCREC_KNN_HT_H": "32"},             # first form of the head / tail scan (knn_scan MODE 3)
    {"LOCREC_KNN_HT_W": "12", "LOCREC_KNN_BLOCKS": "4096"},         # 12-wave blocks, many candidate chunks
    {"LOCREC_KNN_HT_W": "6", "LOCREC_KNN_HT_H": "4"},
    {"LOCREC_KNN_SEED_MIN_SLICES": "1"},                             # threshold-seeding pass even on these small sets (it samples every slice)
    {"LOCREC_KNN_SEED_MIN_SLICES": "1", "LOCREC_KNN_HT_H": "16", "LOCREC_KNN_BLOCKS": "4096"},
    {"LOCREC_KNN_NO_SEED": "1"},
    {"LOCREC_KNN_NO_DIRECT8": "1"},                                  # single requests through the hashed panel (knn_scan1<1>)
    {"LOCREC_KNN_NO_DIRECT8": "1", "LOCREC_KNN_FORCE_HASH": "1", "LOCREC_KNN_NO_HT": "1"},
    {"LOCREC_KNN_NO_PACK": "1"},                                     # small results read back with one copy per array
    {"LOCREC_KNN_NO_ROW_FALLBACK": "1"},                             # a few wide rows demote the whole index (round 2's behaviour)
    {"LOCREC_KNN_NO_TILE_SPECIAL": "1", "LOCREC_KNN_HT_H": "32"},    # wide / too-long queries of a batch one by one
    {"LOCREC_KNN_NO_TILE_SPECIAL": "1", "LOCREC_KNN_NO_SINGLE": "1"},
]
ALL_KEYS = sorted({k for sw in SWITCHES for k in sw})


def random_dataset(rng):
    n = int(rng.choice([65, 200, 1000, 3000]))
    p_dim = int(rng.choice([12, 300, 5000, 40_000]))
    c_dim = int(rng.choice([3, 20]))
    max_p = int(min(p_dim, rng.choice([3, 20, 70])))
    zipf = rng.random() < 0.6
    weights = 1.0 / np.arange(1, p_dim + 1) if zipf else np.ones(p_dim)
    weights /= weights.sum()
    vmax = int(rng.choice([1, 9, 300, -9]))   # -9: counts up to 9 and a FEW rows with a count of 300 - 700 (per-row format fallback)
    few_wide = vmax < 0
    vmax = abs(vmax)
    prp, pidx, pval, crp, cidx, cval = [0], [], [], [0], [], []
    for _ in range(n):
        kp = int(rng.integers(1, max_p + 1))
        ip = np.sort(rng.choice(p_dim, size=kp, replace=False, p=weights))
        kc = int(rng.integers(1, min(c_dim, 6) + 1))
        ic = np.sort(rng.choice(c_dim, size=kc, replace=False))
        pidx.append(ip), pval.append(rng.integers(1, vmax + 1, size=kp).astype(np.float64))
        cidx.append(ic), cval.append(rng.integers(1, vmax + 1, size=kc).astype(np.float64))
        prp.append(prp[-1] + kp), crp.append(crp[-1] + kc)
    if few_wide:
        for r in rng.choice(n, size=max(1, n // 100), replace=False):
            fam = pval if rng.random() < 0.7 else cval
            fam[r][int(rng.integers(0, len(fam[r])))] = float(rng.integers(300, 700))
    ids = rng.permutation(n).astype(np.int64) * 3 + 1000
    d = {"person_ids": ids, "p_rowptr": np.array(prp, np.int64), "p_idx": np.concatenate(pidx).astype(np.int32),
         "p_val": np.concatenate(pval), "p_dim": p_dim, "c_rowptr": np.array(crp, np.int64),
         "c_idx": np.concatenate(cidx).astype(np.int32), "c_val": np.concatenate(cval), "c_dim": c_dim}
    d["r_rowptr"], d["r_place"] = d["p_rowptr"].copy(), d["p_idx"].astype(np.int64) + 40
    d["r_rating"] = rng.integers(1, 6, size=len(d["p_idx"])).astype(np.int64)
    return d


@pytest.mark.parametrize("seed", range(int(os.environ.get("LOCREC_FUZZ_SEEDS", "104"))))
def test_random_index_matches_oracle(pkg, oracle, monkeypatch, seed):
    rng = np.random.default_rng(1000 + seed)
    d = random_dataset(rng)
    n = len(d["person_ids"])
    sw = SWITCHES[seed % len(SWITCHES)]
    for key in ALL_KEYS:
        monkeypatch.delenv(key, raising=False)
    for key, val in sw.items():
        monkeypatch.setenv(key, val)
    k = int(rng.choice([1, 5, 50, 200]))
    pw = float(rng.choice([0.5, 0.25, 0.9]))
    cw = 1.0 - pw
    case = f"seed {seed} switches {sw} n {n} p_dim {d['p_dim']} k {k} pw {pw}"
    ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                      d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"], d["r_rowptr"], d["r_place"], d["r_rating"])
    rows = np.arange(n) if n <= 1000 else rng.choice(n, 700, replace=False)
    ids, sims, cnt = ix.query_batch(d["person_ids"][rows], pw, cw, k)
    oids, osims, ocnt = oracle.knn_similar_batch(d, rows, pw, cw, k, nthreads=8)
    assert np.array_equal(cnt, ocnt), case
    assert np.array_equal(ids, oids), case
    assert np.array_equal(sims, osims), case
    for r in rows[:4]:
        pid = int(d["person_ids"][r])
        j = list(rows).index(r)
        sid, ssim = ix.query(pid, pw, cw, k)
        assert np.array_equal(sid, oids[j][:ocnt[j]]) and np.array_equal(ssim, osims[j][:ocnt[j]]), case
        places, est = ix.recommend(pid, pw, cw, k)
        oplaces, oest = oracle.knn_recommend(d, pid, pw, cw, k)
        assert np.array_equal(places, oplaces), case
        np.testing.assert_allclose(est, oest, rtol=1e-6, atol=0, err_msg=case)
    off, bplaces, best = ix.recommend_batch(d["person_ids"][rows[:40]], pw, cw, k)
    for j, r in enumerate(rows[:40]):
        oplaces, oest = oracle.knn_recommend(d, int(d["person_ids"][r]), pw, cw, k)
        assert np.array_equal(bplaces[off[j]:off[j + 1]], oplaces), case
        np.testing.assert_allclose(best[off[j]:off[j + 1]], oest, rtol=1e-6, atol=0, err_msg=case)
    ix.close()


SG_SWITCHES = [
    {},
    {"LOCREC_SG_NO_COL16": "1"},
    {"LOCREC_SG_PPW": "4"},
    {"LOCREC_SG_GS": "2"},
    {"LOCREC_SG_PERSIST": "1"},
    {"LOCREC_SG_PPW": "2", "LOCREC_SG_NO_COL16": "1"},
    {"LOCREC_SG_NO_GRAPH": "1"},                                    # every iteration launched on its own (no hipGraph replay)
    {"LOCREC_SG_NO_GRAPH": "1", "LOCREC_SG_PPW": "8"},
    {"LOCREC_SG_NO_PACK": "1"},                                     # convergence polls and read-back through device-to-host copies
    {"LOCREC_SG_NO_DENSE_IDS": "1"},                                # vertex ids ranked by sorting, not through the id table
    {"LOCREC_SG_NO_DICT": "1"},                                     # fp64 weights streamed (the default is the dictionary form)
    {"LOCREC_SG_DICT_THREADS": "256", "LOCREC_SG_DICT_PPW": "4"},
    {"LOCREC_SG_DICT_THREADS": "512", "LOCREC_SG_DICT_PPW": "1", "LOCREC_SG_NO_COL16": "1", "LOCREC_SG_NO_GRAPH": "1"},
    {"LOCREC_SG_FUSED": "1"},                                       # the fused iteration (experiment): sums in another order, 1e-15
    {"LOCREC_SG_FUSED": "1", "LOCREC_SG_FUSED_ONE_STREAM": "1", "LOCREC_SG_NO_GRAPH": "1"},
]
SG_KEYS = sorted({k for sw in SG_SWITCHES for k in sw})


def random_graph(rng):
    """Random stochastic graph: a sparse id space, a few hub targets, vertices that are only sources
    or only targets (dangling), self loops and repeated (source, target) pairs; out-weights of every
    source sum to 1 up to rounding."""
    nv = int(rng.choice([5, 60, 700, 5000]))
    ids = np.sort(rng.choice(10 * nv + 50, size=nv, replace=False)).astype(np.int64)
    n_src = max(1, int(nv * rng.uniform(0.3, 1.0)))
    sources = rng.choice(nv, size=n_src, replace=False)
    hubs = rng.choice(nv, size=max(1, nv // 50), replace=False)
    src, dst, w = [], [], []
    for s in sources:
        deg = int(rng.integers(1, min(nv, 40) + 1))
        t = np.where(rng.random(deg) < 0.4, rng.choice(hubs, size=deg), rng.integers(0, nv, size=deg))
        cnt = rng.integers(1, 10, size=deg).astype(np.float64)
        src.append(np.full(deg, s)), dst.append(t), w.append(cnt / cnt.sum())
    src, dst, w = np.concatenate(src), np.concatenate(dst), np.concatenate(w)
    perm = rng.permutation(len(src)) if rng.random() < 0.5 else np.arange(len(src))
    return ids[src[perm]], ids[dst[perm]], w[perm], ids


@pytest.mark.parametrize("seed", range(int(os.environ.get("LOCREC_FUZZ_SEEDS_B", "45"))))
def test_random_graph_matches_oracle(pkg, oracle, monkeypatch, seed):
    rng = np.random.default_rng(5000 + seed)
    src, dst, w, ids = random_graph(rng)
    sw = SG_SWITCHES[seed % len(SG_SWITCHES)]
    for key in SG_KEYS:
        monkeypatch.delenv(key, raising=False)
    for key, val in sw.items():
        monkeypatch.setenv(key, val)
    case = f"seed {seed} switches {sw} V {len(ids)} E {len(src)}"
    sg = pkg.SgGraph(src, dst, w)
    present = np.unique(np.concatenate([src, dst]))
    for _ in range(3):
        v = int(rng.choice(present))
        eps = float(rng.choice([0.0, 1e-3, 0.05]))
        max_it = int(rng.choice([0, 1, 7, 60]))
        got = sg.recommend(v, 0.15, eps, max_it)
        want = oracle.sg_recommend(src, dst, w, v, 0.15, eps, max_it)
        assert np.array_equal(got[0], want[0]), case
        if eps == 0.0 and (got[3] or want[3]):
            # epsilon = 0 "converges" only at an exact fp64 fixed point (x' == x bit for bit).  Rows of in-degree > 4
            # are summed in another order than the oracle's edge-list order, so one side may reach that point a
            # sweep earlier or later (seed 32 of the extended sweep: 53 vs 54) - or settle while the other keeps
            # flipping a last bit until maxIterations (seed 60: 31 vs 60).  Unpinned by the reference
            # (StochasticRecommenderTest.scala:8 TODO); the probabilities below still agree to 1e-6.
            assert got[2] <= max_it and want[2] <= max_it, (case, got[2:], want[2:])
        else:
            assert got[2:] == want[2:], (case, got[2:], want[2:])
        np.testing.assert_allclose(got[1], want[1], rtol=1e-6, atol=0, err_msg=case)
    absent = int(ids.max()) + 7
    with pytest.raises(pkg.IllegalArgumentException, match="No such vertex"):
        sg.recommend(absent, 0.15, 0.01, 5)
    sg.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("LOCREC_FUZZ_SEEDS_C", "12"))))
def test_random_sharded_forms_match_unsharded(pkg, oracle, monkeypatch, seed):
    """The two multi-GPU forms emulated on one GPU, on random inputs: a KNN request with its
    candidate scan cut into 2..9 shards (bit-identical to the unsharded request), and an SG graph
    with its rows of P cut into 2..5 shards (same ids and iteration count, probabilities to 1e-9)."""
    from test_gpu_knn import sharded_request
    from test_gpu_sg import sharded_recommend
    for key in ALL_KEYS + SG_KEYS:
        monkeypatch.delenv(key, raising=False)
    rng = np.random.default_rng(9000 + seed)
    d = random_dataset(rng)
    ix = pkg.KnnIndex(d["person_ids"], d["p_rowptr"], d["p_idx"], d["p_val"], d["p_dim"],
                      d["c_rowptr"], d["c_idx"], d["c_val"], d["c_dim"], d["r_rowptr"], d["r_place"], d["r_rating"])
    shards = int(rng.integers(2, 10))
    k = int(rng.choice([1, 10, 50]))
    for r in rng.choice(len(d["person_ids"]), 3, replace=False):
        pid = int(d["person_ids"][r])
        ids, sims = sharded_request(pkg, ix, pid, 0.5, 0.5, k, shards)
        uid, usim = ix.query(pid, 0.5, 0.5, k)
        assert np.array_equal(ids, uid) and np.array_equal(sims, usim), (seed, shards, k, pid)
        places, est = ix.recommend_neighbours(ids, sims)
        uplaces, uest = ix.recommend(pid, 0.5, 0.5, k)
        assert np.array_equal(places, uplaces) and np.array_equal(est, uest), (seed, shards, k, pid)
    ix.close()
    src, dst, w, ids_all = random_graph(rng)
    gshards = int(rng.integers(2, 6))
    v = int(rng.choice(np.unique(np.concatenate([src, dst]))))
    eps, max_it = float(rng.choice([0.0, 1e-3])), int(rng.choice([1, 9, 40]))
    got = sharded_recommend(pkg, src, dst, w, gshards, v, 0.15, eps, max_it)
    want = oracle.sg_recommend(src, dst, w, v, 0.15, eps, max_it)
    assert np.array_equal(got[0], want[0]), (seed, gshards, v)
    if not (eps == 0.0 and (got[3] or want[3])):  # (epsilon = 0: see test_random_graph_matches_oracle)
        assert got[2:] == want[2:], (seed, gshards, v, eps, max_it, got[2:], want[2:])
    np.testing.assert_allclose(got[1], want[1], rtol=1e-9, atol=0)
    # rows of P^T sharded (all-gather form): bit-identical to the single-GPU handle
    whole = pkg.SgGraph(src, dst, w)
    one = whole.recommend(v, 0.15, eps, max_it)
    whole.close()
    tgt = sharded_recommend(pkg, src, dst, w, gshards, v, 0.15, eps, max_it, by_target=True)
    assert np.array_equal(tgt[0], one[0]) and np.array_equal(tgt[1], one[1]) and tgt[2:] == one[2:], (seed, gshards, v)
