"""The callers either side of the hot path, host side:

* f-1  the on-disk inputs -> device handles: the Parquet sets the reference's builders write
       (RatingVectorsBuilderMain.scala:67-73, StochasticGraphBuilderMain.scala:68-73) and its mains
       read back per request (KnnRecommenderMain.scala:69-88, StochasticRecommenderMain.scala:78-84),
       named by DataUtils.scala:34-58.  Here they are read ONCE with pyarrow into the CSR / edge
       arrays the C ABI takes, and the handle stays on the device between requests.
* f-3  the final ranking of the mains (KnnRecommenderMain.scala:90-107,
       StochasticRecommenderMain.scala:64-83): places of the TARGET region joined with the
       recommendations, ordered by score descending, limited to maxRecommendations.

Spark stores ml.linalg.SparseVector through VectorUDT as
struct<type: tinyint, size: int, indices: array<int>, values: array<double>> (type 0 = sparse,
1 = dense).  That layout is Spark's, not the reference's, and no Spark-written file exists in
/root/reference: the reader follows the published layout and is "parity unpinned" against a real
file (tests write files of this layout with pyarrow)."""
import os

import numpy as np

from . import _lib as L


def generate_file_name(region_ids, dir_path, file_prefix):
    """DataUtils.scala:52-58: <dir>/<prefix>_region<a>_region<b>, ids sorted and distinct."""
    regs = sorted(set(int(r) for r in region_ids))
    return f"{dir_path}/{file_prefix}_" + "_".join(f"region{r}" for r in regs)


def _read(path, columns=None):
    import pyarrow.parquet as pq
    return pq.read_table(path, columns=columns)  # a Spark output directory or a single file


def _list_column(col):
    """offsets (int64, n + 1) and flat values of a list<...> column, as numpy."""
    import pyarrow as pa
    arr = col.combine_chunks() if isinstance(col, pa.ChunkedArray) else col
    if arr.null_count:
        raise L.IllegalArgumentException("null entry in a vector column")
    offs = np.asarray(arr.offsets.to_numpy(zero_copy_only=False), dtype=np.int64)
    vals = arr.values.to_numpy(zero_copy_only=False)
    return offs - offs[0], vals[offs[0]:offs[-1]]


def load_rating_vectors(path, vector_column="rating_vector"):
    """(person_id: long, rating_vector: VectorUDT) -> person_ids, rowptr, indices, values, dim,
    rows sorted by person_id."""
    import pyarrow as pa
    t = _read(path, ["person_id", vector_column])
    pid = np.asarray(t["person_id"].to_numpy(), dtype=np.int64)
    st = t[vector_column].combine_chunks()
    if not pa.types.is_struct(st.type):
        raise L.IllegalArgumentException(f"column {vector_column} is not a VectorUDT struct")
    typ = np.asarray(st.field("type").to_numpy(zero_copy_only=False), dtype=np.int64)
    voff, vals = _list_column(st.field("values"))
    if not np.all(typ == 0):
        # RatingVectorsBuilder.scala:74-77 only ever builds SparseVector; a dense vector here means the
        # file does not come from the reference's builder
        raise L.IllegalArgumentException("dense rating vectors are not produced by the reference's builder")
    size = np.asarray(st.field("size").to_numpy(zero_copy_only=False), dtype=np.int64)
    ioff, idx = _list_column(st.field("indices"))
    if not np.array_equal(ioff, voff):
        raise L.IllegalArgumentException("indices and values of a sparse vector differ in length")
    if len(pid) == 0:
        return pid, np.zeros(1, np.int64), np.empty(0, np.int32), np.empty(0, np.float64), 0
    if size.min() != size.max():
        raise L.IllegalArgumentException("rating vectors of different sizes in one file")
    order = np.argsort(pid, kind="stable")
    lens = np.diff(ioff)[order]
    rowptr = np.zeros(len(pid) + 1, np.int64)
    np.cumsum(lens, out=rowptr[1:])
    gather = np.concatenate([np.arange(ioff[r], ioff[r + 1]) for r in order]) if len(order) else np.empty(0, np.int64)
    return pid[order], rowptr, np.asarray(idx, np.int64)[gather].astype(np.int32), \
        np.asarray(vals, np.float64)[gather], int(size[0])


def load_place_ratings(path):
    """(person_id, place_id, rating: long) (RatingsBuilder.scala:38-47) as three int64 arrays."""
    t = _read(path, ["person_id", "place_id", "rating"])
    return tuple(np.asarray(t[c].to_numpy(), dtype=np.int64) for c in ("person_id", "place_id", "rating"))


def load_stochastic_graph(path):
    """(source_id, target_id, balanced_weight) (StochasticGraphBuilder.scala:12-16); ids of any
    integer width are widened to int64 (StochasticGraphBuilderTest.scala:20-23,56)."""
    t = _read(path, ["source_id", "target_id", "balanced_weight"])
    return (np.asarray(t["source_id"].to_numpy(), dtype=np.int64), np.asarray(t["target_id"].to_numpy(), dtype=np.int64),
            np.asarray(t["balanced_weight"].to_numpy(), dtype=np.float64))


def load_places(data_dir):
    """DataUtils.loadPlaces (:17-23): places_sample with region_id cast to long -> (id, region_id)."""
    t = _read(os.path.join(data_dir, "places_sample"), ["id", "region_id"])
    return np.asarray(t["id"].to_numpy(), dtype=np.int64), np.asarray(t["region_id"].to_numpy(), dtype=np.int64)


def _align(all_ids, ids, rowptr, idx, val):
    """Re-index one family's rows onto the union of person ids (absent persons get empty rows)."""
    pos = np.searchsorted(all_ids, ids)
    lens = np.zeros(len(all_ids), np.int64)
    lens[pos] = np.diff(rowptr)
    out = np.zeros(len(all_ids) + 1, np.int64)
    np.cumsum(lens, out=out[1:])
    return out, idx, val  # rows keep their relative order (both id lists are sorted), so idx / val are unchanged


def knn_index_from_parquet(data_dir, region_ids):
    """KnnRecommenderMain.makeRecommendations' three loads (:53-57) -> one device-resident KnnIndex."""
    from .knn import KnnIndex
    pp, prp, pidx, pval, pdim = load_rating_vectors(generate_file_name(region_ids, data_dir, "place_rating_vectors"))
    cp, crp, cidx, cval, cdim = load_rating_vectors(generate_file_name(region_ids, data_dir, "category_rating_vectors"))
    rp, rplace, rrating = load_place_ratings(generate_file_name(region_ids, data_dir, "place_ratings"))
    ids = np.union1d(np.union1d(pp, cp), rp)
    prp, pidx, pval = _align(ids, pp, prp, pidx, pval)
    crp, cidx, cval = _align(ids, cp, crp, cidx, cval)
    rows = np.searchsorted(ids, rp)
    order = np.argsort(rows, kind="stable")
    rrp = np.zeros(len(ids) + 1, np.int64)
    np.cumsum(np.bincount(rows, minlength=len(ids)), out=rrp[1:])
    return KnnIndex(ids, prp, pidx, pval, pdim, crp, cidx, cval, cdim, rrp, rplace[order], rrating[order])


def sg_graph_from_parquet(data_dir, region_ids):
    """StochasticRecommenderMain.loadStochasticGraph (:78-84) -> one device-resident SgGraph."""
    from .stochastic import SgGraph
    return SgGraph(*load_stochastic_graph(generate_file_name(region_ids, data_dir, "stochastic_graph")))


def read_parquet_frame(path, columns=None):
    """spark.read.parquet(path) for the host mirror: a pandas frame that remembers its input files
    (`attrs["inputFiles"]`, the stand-in for Spark's df.inputFiles the handle cache keys on)."""
    df = _read(path, columns).to_pandas()
    df.attrs["inputFiles"] = [path]
    return df


def knn_make_recommendations(data_dir, region_ids, person_id, place_weight, category_weight, k_nearest):
    """KnnRecommenderMain.makeRecommendations (KnnRecommenderMain.scala:53-67) as the unchanged main runs it for
    EVERY request - name the three Parquet sets of the region pair, construct a recommender, ask it - with the
    device index taken from the process-wide handle cache (keyed by the files' names, sizes and modification
    times): only the first request for a region pair reads the files and builds the index.
    -> (place_id, estimated_rating) arrays."""
    from . import _cache
    from .knn import KnnIndex
    _cache.require_gpu_backend("knn_make_recommendations")
    key = _cache.files_key([generate_file_name(region_ids, data_dir, f)
                            for f in ("place_rating_vectors", "category_rating_vectors", "place_ratings")])
    ix = KnnIndex.through_cache(key, lambda: knn_index_from_parquet(data_dir, region_ids))
    try:
        with ix.lock:
            return ix.recommend(person_id, place_weight, category_weight, k_nearest)
    finally:
        ix.close()  # drops the reference only


def sg_make_recommendations(data_dir, region_ids, vertex_id, epsilon, max_iterations, alpha=0.15):
    """StochasticRecommenderMain.makeRecommendations (StochasticRecommenderMain.scala:53-62), graph from the cache.
    -> (id, probability, iterations, converged)."""
    from . import _cache
    from .stochastic import SgGraph
    _cache.require_gpu_backend("sg_make_recommendations")
    key = _cache.files_key([generate_file_name(region_ids, data_dir, "stochastic_graph")])
    g = SgGraph.through_cache(key, lambda: sg_graph_from_parquet(data_dir, region_ids))
    try:
        with g.lock:
            return g.recommend(vertex_id, alpha, epsilon, max_iterations)
    finally:
        g.close()


def rank_recommendations(ids, scores, place_ids, place_region_ids, target_region_id, max_recommendations):
    """printRecommendations of both mains: places.where(region_id == target) JOIN recommendations
    ON id, ORDER BY score DESC, LIMIT maxRecommendations.  Rows whose id is not a place of the target
    region (persons, categories, places elsewhere) drop out in the join.  Ties: Spark leaves the
    order undefined; here (score desc, id asc)."""
    ids, scores = np.asarray(ids, np.int64), np.asarray(scores, np.float64)
    allowed = np.unique(np.asarray(place_ids, np.int64)[np.asarray(place_region_ids, np.int64) == int(target_region_id)])
    keep = np.isin(ids, allowed)
    ids, scores = ids[keep], scores[keep]
    order = np.lexsort((ids, -scores))[:max(0, int(max_recommendations))]
    return ids[order], scores[order]


def build_with_balanced_weights(betas, all_edges):
    """StochasticGraphBuilder.buildWithBalancedWeights (StochasticGraphBuilder.scala:8-28), the
    producer of the SG path's input: every family's `weight` times its beta,
    families concatenated in the given order (`union` keeps it) - the edge-list order the device
    layout preserves inside each row.  all_edges: sequence of (source_id, target_id, weight) column
    triples or mappings with those keys.  -> (source_id int64, target_id int64, balanced_weight)."""
    if len(betas) != len(all_edges) or not all_edges:
        raise L.IllegalArgumentException("one beta per edge family is required")
    src, dst, w = [], [], []
    for beta, e in zip(betas, all_edges):
        s, t, wt = (e["source_id"], e["target_id"], e["weight"]) if hasattr(e, "keys") or hasattr(e, "columns") else e
        src.append(np.asarray(s, np.int64))
        dst.append(np.asarray(t, np.int64))
        w.append(np.asarray(wt, np.float64) * float(beta))   # col("weight") * beta
    return np.concatenate(src), np.concatenate(dst), np.concatenate(w)


def calc_ratings(person_ids, entity_ids, top_n):
    """RatingsBuilder.calcRatings (RatingsBuilder.scala:32-48): visits -> (person_id, entity_id,
    rating = number of visits), keeping per person the entities whose rank() by rating descending
    is <= top_n.  rank() leaves gaps after ties: an entity's rank is 1 + the number
    of the person's entities with a STRICTLY larger count, so a tie straddling top_n is kept whole.
    No reference test covers it ("parity unpinned").  Rows come back ordered by (person, entity)."""
    p, e = np.asarray(person_ids, np.int64), np.asarray(entity_ids, np.int64)
    if len(p) == 0:
        return p, e, np.empty(0, np.int64)
    order = np.lexsort((e, p))
    p, e = p[order], e[order]
    first = np.r_[True, (p[1:] != p[:-1]) | (e[1:] != e[:-1])]
    gp, ge = p[first], e[first]
    cnt = np.diff(np.r_[np.flatnonzero(first), len(p)])          # count("*") per (person, entity)
    # rank within the person by count descending, ties sharing the smallest position
    o2 = np.lexsort((-cnt, gp))
    sp, sc = gp[o2], cnt[o2]
    pstart = np.r_[True, sp[1:] != sp[:-1]]
    pos = np.arange(len(sp)) - np.maximum.accumulate(np.where(pstart, np.arange(len(sp)), 0))
    newval = pstart | np.r_[True, sc[1:] != sc[:-1]]
    # positions of equal counts inherit the first position of their run
    run_first = np.maximum.accumulate(np.where(newval, np.arange(len(sp)), 0))
    rank = 1 + pos[run_first]
    keep = np.zeros(len(gp), bool)
    keep[o2] = rank <= int(top_n)
    return gp[keep], ge[keep], cnt[keep]


def calc_rating_vectors(person_ids, entity_ids, ratings):
    """RatingVectorsBuilder.calcRatingVectors (:10-25, 52-84): one SparseVector per person, size =
    max entity id + 1 (an id beyond Int range is an ArithmeticException, :36-41), indices ascending,
    values = rating.toDouble (:69).  -> person_ids, rowptr, indices(int32), values(float64), size."""
    p, e, r = np.asarray(person_ids, np.int64), np.asarray(entity_ids, np.int64), np.asarray(ratings, np.int64)
    if len(p) == 0:
        return p, np.zeros(1, np.int64), np.empty(0, np.int32), np.empty(0, np.float64), 0
    max_id = int(e.max())
    if max_id > 2**31 - 1 or int(e.min()) < 0:
        raise ArithmeticError(f"Index out of Int range: {max_id if max_id > 2**31 - 1 else int(e.min())}")
    order = np.lexsort((e, p))
    p, e, r = p[order], e[order], r[order]
    dup = np.r_[False, (p[1:] == p[:-1]) & (e[1:] == e[:-1])]   # TreeSet ordered by index: first one wins
    p, e, r = p[~dup], e[~dup], r[~dup]
    ids, counts = np.unique(p, return_counts=True)
    rowptr = np.zeros(len(ids) + 1, np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return ids, rowptr, e.astype(np.int32), r.astype(np.float64), max_id + 1
