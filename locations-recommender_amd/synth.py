"""Deterministic synthetic inputs of the shapes SURVEY.md 8(d) / BASELINE.json name.

Counter-based: every random number is splitmix64(seed, row, slot), so any rank
(and the CPU baseline) rebuilds bit-identical data without exchanging it.
Nothing here touches the GPU or the oracle.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def u01(seed, stream, row, slot):
    """Uniform [0,1) doubles from (seed, stream, row, slot)."""
    with np.errstate(over="ignore"):
        k = splitmix64(np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1342543DE82EF95)))
        k = splitmix64(k ^ np.asarray(row, dtype=np.uint64))
        k = splitmix64(k ^ (np.asarray(slot, dtype=np.uint64) * np.uint64(0xA24BAED4963EE407)))
    return (k >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _poisson_inv(u, lam, kmax):
    k = np.arange(kmax + 1)
    logp = -lam + k * np.log(lam) - np.cumsum(np.concatenate(([0.0], np.log(np.arange(1, kmax + 1)))))
    cdf = np.cumsum(np.exp(logp))
    return np.searchsorted(cdf, u, side="right")


def _geometric_count(u, cap):
    """1 + Geometric(1/2), capped."""
    g = np.floor(np.log1p(-u) / np.log(0.5)).astype(np.int64)
    return np.minimum(1 + g, cap)


def _ragged(seed, stream, rows0, nnz, draw):
    """For each row r (global index rows0 + r) draw nnz[r] items with draw(u) and return
    (rowptr, sorted-unique items per row, slot-uniforms for values)."""
    n = len(nnz)
    row = np.repeat(np.arange(n, dtype=np.int64), nnz)
    start = np.cumsum(nnz) - nnz
    slot = np.arange(len(row), dtype=np.int64) - np.repeat(start, nnz)
    u = u01(seed, stream, row + rows0, slot)
    item = draw(u).astype(np.int64)
    key = np.unique(row * np.int64(1 << 32) + item)  # sorts by (row, item) and drops duplicates
    row_u = key >> 32
    item_u = key & np.int64(0xFFFFFFFF)
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, row_u + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr, row_u, item_u


def knn_dataset(n_persons, n_places, seed, n_categories=20, mean_places=24, max_places=100,
                mean_categories=4, max_categories=10, value_cap=365, first_row=0, rows=None):
    """cfg2/cfg4-shaped KNN input: Zipf(1.0) places, Poisson row lengths, integer
    counts 1 + Geometric(1/2) capped at value_cap.  Place ids start at 40 like the reference's
    sample generator (SampleGeneratorMain.scala:36-37), person ids follow the places.
    rows/first_row select a shard [first_row, first_row + rows) with identical content."""
    rows = n_persons - first_row if rows is None else rows
    p_dim = 40 + n_places
    ridx = np.arange(first_row, first_row + rows, dtype=np.int64)
    nnz_p = np.clip(1 + _poisson_inv(u01(seed, 1, ridx, 0), mean_places, 4 * max_places), 1, max_places)
    nnz_c = np.clip(1 + _poisson_inv(u01(seed, 2, ridx, 0), mean_categories, 8 * max_categories), 1, max_categories)
    zipf_cdf = np.cumsum(1.0 / np.arange(1, n_places + 1))
    zipf_cdf /= zipf_cdf[-1]
    prp, prow, pitem = _ragged(seed, 3, first_row, nnz_p,
                               lambda u: 40 + np.minimum(np.searchsorted(zipf_cdf, u, side="right"), n_places - 1))
    crp, crow, citem = _ragged(seed, 4, first_row, nnz_c,
                               lambda u: np.minimum((u * n_categories).astype(np.int64), n_categories - 1))
    pval = _geometric_count(u01(seed, 5, prow + first_row, pitem), value_cap).astype(np.float64)
    cval = _geometric_count(u01(seed, 6, crow + first_row, citem), value_cap).astype(np.float64)
    person_ids = p_dim + ridx
    return {
        "person_ids": person_ids,
        "p_rowptr": prp, "p_idx": pitem.astype(np.int32), "p_val": pval, "p_dim": p_dim,
        "c_rowptr": crp, "c_idx": citem.astype(np.int32), "c_val": cval, "c_dim": n_categories,
    }


def sg_dataset(n_persons=280_000, n_places=10_000, n_categories=20, seed=0x5EED0003):
    """cfg3-shaped stochastic graph: person->place (beta 0.5), person->category
    (beta 0.5), place->place (beta 1), category->place (beta 1); every source's out-weights sum
    to 1 up to rounding; no edge targets a person.  Ids share one space:
    categories 0.., places 40.., persons after the places (SampleGeneratorMain.scala:36-37,54).
    Edge order = the four families concatenated (StochasticGraphBuilder.scala:8-28)."""
    place0, person0 = 40, 40 + n_places
    pr = np.arange(n_persons, dtype=np.int64)
    zipf_cdf = np.cumsum(1.0 / np.arange(1, n_places + 1))
    zipf_cdf /= zipf_cdf[-1]

    def family(stream, n_src, nnz, draw, beta, src0):
        rp, row, item = _ragged(seed, stream, 0, nnz, draw)
        cnt = _geometric_count(u01(seed, stream + 100, row, item), 365).astype(np.float64)
        tot = np.zeros(n_src)
        np.add.at(tot, row, cnt)
        w = (cnt / tot[row]) * beta  # weight = count/total (PersonLikesPlace.scala:81), then * beta
        return src0 + row, item, w

    nnz_pp = np.clip(_poisson_inv(u01(seed, 11, pr, 0), 12.6, 64), 1, 64)
    s1, t1, w1 = family(12, n_persons, nnz_pp,
                        lambda u: place0 + np.minimum(np.searchsorted(zipf_cdf, u, side="right"), n_places - 1),
                        0.5, person0)
    nnz_pc = np.clip(_poisson_inv(u01(seed, 13, pr, 0), 5.5, 32), 1, 10)
    s2, t2, w2 = family(14, n_persons, nnz_pc,
                        lambda u: np.minimum((u * n_categories).astype(np.int64), n_categories - 1), 0.5, person0)
    lr = np.arange(n_places, dtype=np.int64)
    nnz_ll = 1 + np.minimum((u01(seed, 15, lr, 0) * 50).astype(np.int64), 49)
    s3, t3, w3 = family(16, n_places, nnz_ll,
                        lambda u: place0 + np.minimum((u * n_places).astype(np.int64), n_places - 1), 1.0, place0)
    nnz_cl = np.full(n_categories, min(100, n_places), np.int64)
    s4, t4, w4 = family(18, n_categories, nnz_cl,
                        lambda u: place0 + np.minimum((u * n_places).astype(np.int64), n_places - 1), 1.0, 0)
    return {
        "source_id": np.concatenate([s1, s2, s3, s4]),
        "target_id": np.concatenate([t1, t2, t3, t4]),
        "balanced_weight": np.concatenate([w1, w2, w3, w4]),
        "first_person": person0,
    }


def small_knn_dataset(n=300, p_dim=500, c_dim=20, seed=7, integer=True, negative=False):
    """Small ragged KNN input for parity tests (ties on purpose: many single-place persons)."""
    rng = np.random.default_rng(seed)
    prp, pidx, pval, crp, cidx, cval = [0], [], [], [0], [], []
    for i in range(n):
        kp = 1 if i % 7 == 0 else int(rng.integers(1, 40))
        kc = int(rng.integers(1, min(10, c_dim) + 1))
        ip = np.sort(rng.choice(min(p_dim, 60) if i % 3 else p_dim, size=min(kp, min(p_dim, 60)), replace=False))
        ic = np.sort(rng.choice(c_dim, size=kc, replace=False))
        vp = rng.integers(1, 9, size=len(ip)).astype(np.float64)
        vc = rng.integers(1, 30, size=len(ic)).astype(np.float64)
        if not integer:
            vp = vp + rng.random(len(ip))
            vc = vc * 0.37
        if negative:
            vp = vp * rng.choice([-1.0, 1.0], size=len(ip), p=[0.2, 0.8])
        pidx.append(ip); pval.append(vp); prp.append(prp[-1] + len(ip))
        cidx.append(ic); cval.append(vc); crp.append(crp[-1] + len(ic))
    ids = rng.permutation(np.arange(1000, 1000 + n)).astype(np.int64)
    return {
        "person_ids": ids,
        "p_rowptr": np.array(prp, np.int64), "p_idx": np.concatenate(pidx).astype(np.int32),
        "p_val": np.concatenate(pval), "p_dim": p_dim,
        "c_rowptr": np.array(crp, np.int64), "c_idx": np.concatenate(cidx).astype(np.int32),
        "c_val": np.concatenate(cval), "c_dim": c_dim,
    }


def _knn_chunk(args):
    n_persons, n_places, seed, first, rows, kw = args
    return knn_dataset(n_persons, n_places, seed, first_row=first, rows=rows, **kw)


def concat_knn_parts(parts):
    """Concatenate row-range shards of one KNN data set (bit-identical to generating it whole)."""
    out = {"p_dim": parts[0]["p_dim"], "c_dim": parts[0]["c_dim"],
           "person_ids": np.concatenate([p["person_ids"] for p in parts])}
    for fam in ("p", "c"):
        out[fam + "_idx"] = np.concatenate([p[fam + "_idx"] for p in parts])
        out[fam + "_val"] = np.concatenate([p[fam + "_val"] for p in parts])
        ptrs, base = [np.zeros(1, np.int64)], 0
        for p in parts:
            ptrs.append(p[fam + "_rowptr"][1:] + base)
            base += int(p[fam + "_rowptr"][-1])
        out[fam + "_rowptr"] = np.concatenate(ptrs)
    return out


def knn_dataset_parallel(n_persons, n_places, seed, workers=8, chunk=250_000, **kw):
    """knn_dataset() generated in row chunks by a process pool: the generator is counter-based, so
    the result does not depend on the chunking (tests/test_synth.py)."""
    import multiprocessing as mp
    jobs = [(n_persons, n_places, seed, f, min(chunk, n_persons - f), kw) for f in range(0, n_persons, chunk)]
    if workers <= 1 or len(jobs) == 1:
        return concat_knn_parts([_knn_chunk(j) for j in jobs])
    with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
        return concat_knn_parts(pool.map(_knn_chunk, jobs))
