/*
 * locrec.h -- C ABI of liblocrec.so, the MI355X (gfx950) implementation of the
 * two hot paths of tashoyan/locations-recommender.
 *
 * The reference has NO native boundary (it is 100 % Scala on Spark); the
 * drop-in boundary is the public surface of two Scala classes, and this header
 * is the FFI a JNI shim under those classes binds (INTEGRATION.md shows the
 * shim).  Reference paths below are relative to
 *   recommender/src/main/scala/com/github/tashoyan/recommender/
 *
 *   knn/KnnRecommender.scala:9-25        class KnnRecommender(placeRatingVectors,
 *                                        categoryRatingVectors, placeRatings,
 *                                        placeWeight, categoryWeight, kNearest)
 *                                        .makeRecommendations(personId)
 *   stochastic/StochasticRecommender.scala:28-34,66-71
 *                                        class StochasticRecommender(stochasticEdges,
 *                                        epsilon, maxIterations)
 *                                        .makeRecommendations(vertexId)
 *
 * Conventions
 *   - every function returns an int32 status; LOCREC_OK == 0.  The message of
 *     the last failure on the calling thread is locrec_last_error().
 *   - LOCREC_E_INVALID_ARG and LOCREC_E_NOT_FOUND correspond to the
 *     IllegalArgumentException the reference throws (KnnRecommender.scala:17-20,83;
 *     StochasticRecommender.scala:33-34,70); the message text is the reference's.
 *   - handles are opaque; inputs are caller-owned host arrays that are COPIED to
 *     the device at create time (the library never retains a caller pointer);
 *     outputs go to caller-allocated buffers with an in/out element count: on
 *     entry the capacity, on return the number of rows the result HAS (which
 *     may exceed the capacity; only min(capacity, count) rows are written).
 *   - one handle = one device + one HIP stream; calls on one handle must be
 *     serialised by the caller, distinct handles are independent.
 *   - plain C types only: pointers, sizes, doubles.  No C++/torch types.
 */
#ifndef LOCREC_H
#define LOCREC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LOCREC_OK 0
#define LOCREC_E_INVALID_ARG 1 /* IllegalArgumentException: require() failed            */
#define LOCREC_E_NOT_FOUND 2   /* IllegalArgumentException: "No such person / vertex"    */
#define LOCREC_E_DEVICE 3      /* HIP runtime failure, no GPU, kernel image missing      */
#define LOCREC_E_OOM 4         /* host or device allocation failed                        */
#define LOCREC_E_ARITHMETIC 5  /* ArithmeticException: "Index out of Int range: $l"
                                  (knn/RatingVectorsBuilder.scala:36-41)                  */

/* Where the arrays of the locrec_calc_* / locrec_build_* functions live. */
#define LOCREC_MEM_HOST 0      /* caller-owned host arrays, copied in and out            */
#define LOCREC_MEM_DEVICE 1    /* arrays of the CURRENT HIP device, inputs and outputs   */

/* Thread-local message of the last non-OK status returned on this thread. */
const char *locrec_last_error(void);
/* Library version string, e.g. "locrec 0.1 (gfx950)". */
const char *locrec_version(void);

/* Number of visible HIP devices / select the device new handles are created on. */
int32_t locrec_device_count(int32_t *out_count);
int32_t locrec_set_device(int32_t ordinal);
/* Device allocations (hipMalloc) this process has made through the library so far.  Diagnostic: the difference
 * across a stretch of steady-state calls should be 0 - work buffers live with their handle and only ever grow. */
int32_t locrec_device_allocations(int64_t *out_count);

/* Bytes of device memory the library holds right now (every handle's index / graph and work buffers). */
int32_t locrec_device_bytes_in_use(int64_t *out_bytes);

/* ===================================================================== */
/* Handle cache: the "handle persists across requests" half of the drop-in.
 *
 * The reference's mains construct a NEW recommender from freshly read DataFrames for every request
 * (knn/KnnRecommenderMain.scala:53-67, stochastic/StochasticRecommenderMain.scala:53-62) and never close
 * anything.  Behind those unchanged mains the device index must outlive the object that built it: the host
 * class derives a KEY from what its DataFrames are (input files + sizes + modification times; the weights,
 * K, epsilon and maxIterations are per-request arguments of the calls below and NOT part of the key), and
 *     locrec_cache_acquire(kind, key, &h)         h != NULL: a hit, reference taken - skip collect / create
 *     locrec_*_create(...)  +  locrec_cache_publish(kind, key, created, bytes, &h)
 *                                                 on a miss: the cache takes OWNERSHIP of `created` (if the key
 *                                                 appeared meanwhile the newcomer is destroyed and the cached
 *                                                 handle returned); reference taken
 *     locrec_cache_release(kind, h)               when the host object is closed or collected (close(),
 *                                                 java.lang.ref.Cleaner, __del__): drops the reference; the
 *                                                 handle STAYS cached.  A handle that was never published is
 *                                                 destroyed, so hosts release every handle the same way.
 * Unreferenced entries are destroyed least-recently-used first while the library's live device bytes
 * (locrec_device_bytes_in_use) exceed the byte limit or the entry count its limit (defaults 64 GiB / 64
 * entries; LOCREC_CACHE_BYTES / LOCREC_CACHE_ENTRIES or locrec_cache_set_limits, -1 = keep).  A handle is
 * still one device + one stream: users of one cached handle serialise their calls (the host classes lock).
 */
#define LOCREC_CACHE_KNN 0
#define LOCREC_CACHE_SG 1
int32_t locrec_cache_acquire(int32_t kind, const char *key, void **out_handle);
int32_t locrec_cache_publish(int32_t kind, const char *key, void *handle, int64_t device_bytes, void **out_handle);
int32_t locrec_cache_release(int32_t kind, void *handle);
int32_t locrec_cache_set_limits(int64_t max_device_bytes, int64_t max_entries);
int32_t locrec_cache_clear(void); /* drop every entry; referenced ones are destroyed by their last release */
int32_t locrec_cache_stats(int64_t *out_entries, int64_t *out_entry_bytes, int64_t *out_hits,
                           int64_t *out_misses, int64_t *out_evictions);

/* ===================================================================== */
/* KNN: knn/KnnRecommender.scala, knn/Distance.scala                     */

typedef struct locrec_knn_index locrec_knn_index;

/*
 * Replaces the three DataFrames of the KnnRecommender constructor
 * (KnnRecommender.scala:9-16), collected to CSR:
 *   person_ids[n]                      person_id column (distinct)
 *   p_rowptr[n+1], p_idx[], p_val[]    placeRatingVectors: per person the
 *                                      SparseVector(size = p_dim, indices
 *                                      ascending, values)   (RatingVectorsBuilder.scala:74-77)
 *   c_rowptr[n+1], c_idx[], c_val[]    categoryRatingVectors, size = c_dim
 *   r_rowptr[n+1], r_place[], r_rating[]  placeRatings (person_id, place_id,
 *                                      rating: Long) grouped by person; pass
 *                                      r_rowptr == NULL to use the place vectors
 *                                      themselves (what RatingVectorsBuilderMain
 *                                      .scala:41-73 writes: the same data in COO).
 * A person with an empty row in a family is absent from that family's frame.
 * Rows whose stored values have zero norm, non-finite values, unsorted or
 * out-of-range indices are rejected with LOCREC_E_INVALID_ARG.
 * Norms (Distance.scala:11-16) are computed ONCE here, on the device.
 */
int32_t locrec_knn_create(
    int64_t n, const int64_t *person_ids,
    const int64_t *p_rowptr, const int32_t *p_idx, const double *p_val, int32_t p_dim,
    const int64_t *c_rowptr, const int32_t *c_idx, const double *c_val, int32_t c_dim,
    const int64_t *r_rowptr, const int64_t *r_place, const int64_t *r_rating,
    locrec_knn_index **out_index);

/*
 * The same with every array already in DEVICE memory of the current device (row pointers included):
 * nothing passes through the host, the whole index is built by kernels and device sorts
 * (csrc/knn_build.hip).  This is what a device-side rating-vector builder or an
 * RCCL all-gather of per-rank shards (8 e) hands over.  The arrays are only read; the caller keeps them.
 */
int32_t locrec_knn_create_from_device(
    int64_t n, const int64_t *person_ids,
    const int64_t *p_rowptr, const int32_t *p_idx, const double *p_val, int32_t p_dim,
    const int64_t *c_rowptr, const int32_t *c_idx, const double *c_val, int32_t c_dim,
    const int64_t *r_rowptr, const int64_t *r_place, const int64_t *r_rating,
    locrec_knn_index **out_index);

int32_t locrec_knn_destroy(locrec_knn_index *index);

/* Number of persons; bytes the candidate rows occupy in HBM in the layout the
 * scan kernel streams (the "algorithmic bytes" of one query-vs-all pass).    */
int32_t locrec_knn_info(const locrec_knn_index *index, int64_t *out_n,
                        int64_t *out_scan_bytes, int32_t *out_packed);

/* Bytes of the image the BATCHED scan streams per query tile: the head / tail form (SELL rows of the head
 * elements + the inverted tail, knn_ht.h) where the index has it, else the same as locrec_knn_info's. */
int32_t locrec_knn_batch_scan_bytes(const locrec_knn_index *index, int64_t *out_scan_bytes);

/* Measurement only: sizes of the head / tail image (knn_ht.h) - 32-bit element words of the SELL head rows of both
 * families incl. padding (every word costs one tile 8 v_pk_mad_u16 and one address add: the exact multiply-add count
 * of a launch is tiles x words / 8 wave instructions, which the roofline derivation in profiles/ uses), postings of the
 * inverted tail, and the number of rows kept out of the image by the per-row format fallback.  Zeros without the image. */
int32_t locrec_knn_ht_image_info(const locrec_knn_index *index, int64_t *out_head_words, int64_t *out_tail_postings,
                                 int64_t *out_wide_rows);

/*
 * Plan of the last batched scan enqueued on this handle (measurement only; bench.py prints it):
 * kernel 1 = knn_scan (row scan over a query panel in LDS), 2 = knn_scan_ht (head / tail form: dense
 * head panel in LDS + inverted tail, knn_ht.h), 3 = knn_scan in its head / tail mode (A/B partner of 2),
 * 0 = none yet; mode 0 / 1 / 2 = GENERIC (fp64) / PACK32 / PACK16, 3 = head / tail (u16 dots); the
 * query tile (queries sharing one read of a candidate row) and the waves per block.
 */
int32_t locrec_knn_scan_plan(const locrec_knn_index *index, int32_t *out_kernel, int32_t *out_mode,
                             int32_t *out_query_tile, int32_t *out_waves_per_block);

/*
 * Distance.vectorLength (knn/Distance.scala:11-16) of every person's two vectors, as
 * computed on the device at create time; arrays of n in the order of person_ids[].
 * An absent (empty) vector has length 0.0 (DistanceTest.scala:10-14).
 */
int32_t locrec_knn_vector_lengths(locrec_knn_index *index, double *out_place_lengths,
                                  double *out_category_lengths);

/*
 * Distance.cosineSimilarity (knn/Distance.scala:7-9) of two persons' place vectors and of their category vectors,
 * computed on the device from the index's own fp64 rows and lengths: (v1 dot v2) / (len(v1) * len(v2)), any sign
 * (findSimilarPersons only ever shows the positive ones, KnnRecommender.scala:91-93), NaN where a vector is empty.
 * DistanceTest.scala:16-60 through the ABI.  LOCREC_E_NOT_FOUND for an unknown id.
 */
int32_t locrec_knn_cosine_similarity(locrec_knn_index *index, int64_t person_a, int64_t person_b,
                                     double *out_place_cosine, double *out_category_cosine);

/*
 * findSimilarPersons (KnnRecommender.scala:27-49): the K nearest persons of
 * person_id, ordered by (similarity desc, person_id asc).
 * place_weight/category_weight/k_nearest are validated exactly as the
 * constructor does (:17-20).  There is no limit on the length of the person's
 * vectors (a row up to 2^21 - 1 non-zeros per family is accepted at create
 * time) nor on k_nearest: a vector too long for an on-chip panel takes a
 * slower global-memory scan with identical results; this holds for every
 * query / recommend function below, batched ones included.
 */
int32_t locrec_knn_query(
    locrec_knn_index *index, int64_t person_id,
    double place_weight, double category_weight, int64_t k_nearest,
    int64_t *out_person_ids, double *out_similarities, int64_t *inout_count);

/*
 * makeRecommendations (KnnRecommender.scala:22-25,51-70):
 * rows (place_id, estimated_rating), ordered by place_id ascending (the
 * reference leaves the order undefined).
 */
int32_t locrec_knn_recommend(
    locrec_knn_index *index, int64_t person_id,
    double place_weight, double category_weight, int64_t k_nearest,
    int64_t *out_place_ids, double *out_estimated_ratings, int64_t *inout_count);

/*
 * Batched makeRecommendations (KnnRecommender.scala:22-25 for many persons; the additive
 * "makeRecommendationsBatch" of SURVEY.md 8b): findSimilarPersons AND makeRecommendations0 on the
 * device for every listed person.  Rows of person i are [out_offsets[i], out_offsets[i+1]) of
 * out_place_ids / out_estimated_ratings, ordered by place id; out_offsets has nq + 1 entries.
 * *inout_capacity: in = room in the two output arrays, out = rows needed; when the room is too
 * small only out_offsets is filled and the call is repeated with larger arrays.
 * k_nearest <= LOCREC_KNN_BATCH_MAX_K.
 */
int32_t locrec_knn_recommend_batch(
    locrec_knn_index *index, int64_t nq, const int64_t *person_ids,
    double place_weight, double category_weight, int64_t k_nearest,
    int64_t *out_offsets, int64_t *out_place_ids, double *out_estimated_ratings, int64_t *inout_capacity);
/* device-resident form over the internal rows [first, first + nq) (see locrec_knn_topk_range_async) */
int32_t locrec_knn_recommend_range_async(
    locrec_knn_index *index, int64_t first, int64_t nq,
    double place_weight, double category_weight, int64_t k_nearest);
int32_t locrec_knn_fetch_recommend(
    locrec_knn_index *index, int64_t nq,
    int64_t *out_offsets, int64_t *out_place_ids, double *out_estimated_ratings, int64_t *inout_capacity);

/*
 * One request with its candidate scan split over several GPUs (SURVEY.md 8e "KNN single request,
 * latency mode").  Every GPU holds the whole index (132 MB at cfg2, 1.3 GB at cfg4: nothing next to
 * 288 GB) and scans candidate shard shard_index of shard_count, a contiguous range of the
 * length-sorted rows holding ~1/shard_count of the stored elements:
 *     locrec_knn_query_shard          findSimilarPersons (:27-49) over that shard -> local top-K
 *     all-gather of the local lists   by the caller (K * 16 bytes per GPU), merge by
 *                                     (similarity desc, person_id asc) -> the K nearest
 *     locrec_knn_recommend_neighbours makeRecommendations0 (:51-70) for that list, on any one GPU
 * k_nearest is limited to LOCREC_KNN_BATCH_MAX_K here.  The union over all shards of the local
 * lists contains the unsharded answer, so the merged result is identical to locrec_knn_query's.
 * locrec_knn_recommend_neighbours rejects a neighbour id that is listed twice (LOCREC_E_INVALID_ARG).
 */
int32_t locrec_knn_query_shard(
    locrec_knn_index *index, int64_t person_id,
    double place_weight, double category_weight, int64_t k_nearest,
    int32_t shard_index, int32_t shard_count,
    int64_t *out_person_ids, double *out_similarities, int64_t *inout_count);
int32_t locrec_knn_recommend_neighbours(
    locrec_knn_index *index, int64_t n_neighbours,
    const int64_t *neighbour_person_ids, const double *similarities,
    int64_t *out_place_ids, double *out_estimated_ratings, int64_t *inout_count);

/*
 * Batched findSimilarPersons: the additive "all-pairs" surface.
 * out_person_ids / out_similarities are [nq * k_nearest] row-major, padded with
 * id -1 / similarity 0.0; out_counts[nq] is the number of valid entries.
 * Requires k_nearest <= LOCREC_KNN_BATCH_MAX_K.
 */
#define LOCREC_KNN_BATCH_MAX_K 1024
int32_t locrec_knn_query_batch(
    locrec_knn_index *index, int64_t nq, const int64_t *person_ids,
    double place_weight, double category_weight, int64_t k_nearest,
    int64_t *out_person_ids, double *out_similarities, int64_t *out_counts);

/* Every person as the query, in the order of person_ids[] given at create.  A person whose place
 * or category vector is empty is a candidate of the others' outer join (KnnRecommender.scala:39-40)
 * but not a valid query (:77-83 throws "No such person" for it): its out_counts entry is -1 and its
 * row is padded; the call does not fail. */
int32_t locrec_knn_all_pairs_topk(
    locrec_knn_index *index,
    double place_weight, double category_weight, int64_t k_nearest,
    int64_t *out_person_ids, double *out_similarities, int64_t *out_counts);

/*
 * Device-resident form (used by bench.py; no host buffers inside the timed
 * region).  The index keeps the persons in an internal ROW order (ascending
 * vector length, so a tile of consecutive rows holds queries of similar size);
 * the queries of this call are the persons at rows [first_row, first_row + nq).
 * Kernels are enqueued on the handle's stream and the result stays in HBM until
 * locrec_knn_fetch_topk().  locrec_knn_row_person_ids() names the persons.
 * Rows that are not valid queries (an empty place or category vector) get count -1, as above;
 * the batched aggregation gives them zero recommendation rows.
 */
int32_t locrec_knn_row_person_ids(locrec_knn_index *index, int64_t first_row, int64_t nq,
                                  int64_t *out_person_ids);
int32_t locrec_knn_topk_range_async(
    locrec_knn_index *index, int64_t first_row, int64_t nq,
    double place_weight, double category_weight, int64_t k_nearest);
int32_t locrec_knn_fetch_topk(
    locrec_knn_index *index, int64_t nq, int64_t k_nearest,
    int64_t *out_person_ids, double *out_similarities, int64_t *out_counts);

/* Use an externally created hipStream_t (passed as void*) for this handle. */
/*
 * Statistics of the batched scan's barrier-free insertion path: how many flush intervals (4 slices
 * of one block), since the index was created, overran a wave's survivor queue and were replayed
 * with synchronous insertion inside the kernel.  Results are never affected and nothing is redone
 * on the host; a few hundred per million intervals is normal (runs of tied candidates).
 */
int32_t locrec_knn_replayed_intervals(locrec_knn_index *index, int64_t *out_blocks);

int32_t locrec_knn_set_stream(locrec_knn_index *index, void *hip_stream);
int32_t locrec_knn_synchronize(locrec_knn_index *index);
/*
 * HIP-event profile of the dominant (scan) kernel: when enabled every scan
 * launch is bracketed by events on the handle's stream; read returns the
 * summed duration and launch count since the last reset and resets them.
 */
int32_t locrec_knn_profile_enable(locrec_knn_index *index, int32_t on);
int32_t locrec_knn_profile_read(locrec_knn_index *index, double *out_scan_ms, int64_t *out_launches);

/* ===================================================================== */
/* SG: stochastic/StochasticRecommender.scala                            */

typedef struct locrec_sg_graph locrec_sg_graph;

/*
 * Replaces the stochasticEdges DataFrame (source_id, target_id,
 * balanced_weight) of the constructor (StochasticRecommender.scala:28-31).
 * Builds vertexes = distinct(source U target) (:42-49) and the device layout.
 */
int32_t locrec_sg_create(
    int64_t n_edges, const int64_t *source_ids, const int64_t *target_ids,
    const double *balanced_weights, locrec_sg_graph **out_graph);

int32_t locrec_sg_destroy(locrec_sg_graph *graph);

/* vertexCount (:51); edges; bytes one sweep x -> x' streams (algorithmic). */
int32_t locrec_sg_info(const locrec_sg_graph *graph, int64_t *out_vertices,
                       int64_t *out_edges, int64_t *out_sweep_bytes);

/* Bytes one sweep x -> x' moves in the DEVICE layout (uint16 / int32 columns + fp64 weights of every
 * slot incl. padding, piece descriptors, partial sums written and re-read, x and x'): the figure
 * bench.py prices the SG kernels against (locrec_sg_info's is SURVEY 8d's reference-width model). */
int32_t locrec_sg_device_bytes(const locrec_sg_graph *graph, int64_t *out_sweep_bytes);

/* Distinct balanced_weight values (by bit pattern, +0.0 of the padding slots included) when the handle streams a
 * uint16 index per edge and looks the fp64 value up in a table (at most 8192 of them: the weights are count / total x
 * beta, few distinct values); 0 when it streams the fp64 weights themselves.  Same results either way, bit for bit. */
int32_t locrec_sg_weight_dictionary(const locrec_sg_graph *graph, int32_t *out_entries);

/*
 * makeRecommendations (StochasticRecommender.scala:66-141).
 * alpha is 0.15 in the reference (:38) and is a parameter here.
 * Rows (id, probability) with id != vertex_id and probability > 0 (:84-88),
 * ordered by id ascending (the reference leaves the order undefined).
 * out_iterations: the 0-based counter the reference prints (:94,100);
 * out_converged: 1 = "Converged in N iterations", 0 = maximum reached.
 */
int32_t locrec_sg_recommend(
    locrec_sg_graph *graph, int64_t vertex_id,
    double alpha, double epsilon, int64_t max_iterations,
    int64_t *out_ids, double *out_probabilities, int64_t *inout_count,
    int64_t *out_iterations, int32_t *out_converged);

/* Device-resident form (bench.py): enqueue the iteration, read back later. */
int32_t locrec_sg_iterate_async(
    locrec_sg_graph *graph, int64_t vertex_id,
    double alpha, double epsilon, int64_t max_iterations);
/*
 * Exactly `sweeps` applications of calcNextX (:108-128) with the fused convergence sum
 * still computed but never acted on -- the fixed-work form the benchmark times (with
 * epsilon = 0 the reference's loop stops as soon as x reaches an exact fp64 fixed point,
 * which is data dependent).  locrec_sg_fetch() then reports converged = 0.
 */
int32_t locrec_sg_sweeps_async(locrec_sg_graph *graph, int64_t vertex_id, double alpha, int64_t sweeps);
int32_t locrec_sg_fetch(
    locrec_sg_graph *graph,
    int64_t *out_ids, double *out_probabilities, int64_t *inout_count,
    int64_t *out_iterations, int32_t *out_converged);

/*
 * Row-sharded form (one graph over several GPUs; BASELINE.json configs[4]: "SpMV rows sharded ...
 * with RCCL all-reduce of x each iteration").  Every rank creates a handle from the SAME edge list
 * with its own shard_index: the handle keeps the edges whose source vertex falls into the shard
 * ("rows of P"), while the vertex set and the layout of x are global and identical on every rank.
 * One iteration of calcNextX (:108-128) is then
 *     locrec_sg_shard_sigma(g, sigma)      this shard's partial sigma over all live vertices
 *     all-reduce(sum) of sigma[0 .. live)  by the caller, on the handle's stream (RCCL / torch.distributed)
 *     locrec_sg_shard_apply(g, sigma, alpha)
 * and the caller runs step()'s loop (:92-106), reading isConverged's sum with locrec_sg_shard_d2.
 * sigma is a caller-owned DEVICE buffer of locrec_sg_live_count() doubles.  Only the live entries of x
 * are exchanged: a vertex without inbound edges never needs communication.
 */
int32_t locrec_sg_create_sharded(
    int64_t n_edges, const int64_t *source_ids, const int64_t *target_ids, const double *balanced_weights,
    int32_t shard_index, int32_t shard_count, locrec_sg_graph **out_graph);
/*
 * The same protocol with the rows of P^T (targets) sharded: live row l - position l of sigma -
 * belongs to shard l % shard_count, which keeps ALL inbound edges of its rows.  locrec_sg_shard_sigma
 * then yields the complete sigma[l] for the owned l (summed in the single-GPU kernel's fixed order) and
 * 0 elsewhere, so the exchange is an all-GATHER of the owned entries (half the bytes of the all-reduce)
 * and the result is bit-identical to the unsharded one.  begin / sigma / apply / d2 / finish as above.
 */
int32_t locrec_sg_create_target_sharded(
    int64_t n_edges, const int64_t *source_ids, const int64_t *target_ids, const double *balanced_weights,
    int32_t shard_index, int32_t shard_count, locrec_sg_graph **out_graph);
int32_t locrec_sg_live_count(const locrec_sg_graph *graph, int64_t *out_live);
int32_t locrec_sg_shard_begin(locrec_sg_graph *graph, int64_t vertex_id);
int32_t locrec_sg_shard_sigma(locrec_sg_graph *graph, double *sigma_device);
int32_t locrec_sg_shard_apply(locrec_sg_graph *graph, const double *sigma_device, double alpha);
int32_t locrec_sg_shard_d2(locrec_sg_graph *graph, double *out_diff_squared);
int32_t locrec_sg_shard_finish(locrec_sg_graph *graph, int64_t iterations, int32_t converged);

/*
 * A group of independent graphs iterated together (BASELINE.json configs[4]: many graphs per GPU, e.g. one
 * per region pair, PlaceVisits.scala:63-67): one sweep kernel and one combine kernel per round for ALL graphs
 * of the group instead of two launches per graph, on the group's own stream.  The graphs stay usable on their
 * own; after locrec_sg_group_sweeps_async every graph holds its result as after locrec_sg_sweeps_async
 * (same kernels' bodies, bit-identical) and is read with locrec_sg_fetch or awaited with
 * locrec_sg_group_synchronize.  The group does not own the graphs: destroy it before them.
 */
typedef struct locrec_sg_group locrec_sg_group;
int32_t locrec_sg_group_create(locrec_sg_graph *const *graphs, int32_t n_graphs, locrec_sg_group **out_group);
void locrec_sg_group_destroy(locrec_sg_group *group);
int32_t locrec_sg_group_sweeps_async(locrec_sg_group *group, const int64_t *vertex_ids, double alpha, int64_t sweeps);
/* makeRecommendations' iteration (StochasticRecommender.scala:92-106) for every graph of the group: up to
 * max_iterations rounds, each graph stopping at its own isConverged (:130-141); read the results - ids,
 * probabilities, the iteration counter the reference prints, converged or not - with locrec_sg_fetch. */
int32_t locrec_sg_group_iterate_async(locrec_sg_group *group, const int64_t *vertex_ids, double alpha, double epsilon,
                                      int64_t max_iterations);
int32_t locrec_sg_group_synchronize(locrec_sg_group *group);

int32_t locrec_sg_set_stream(locrec_sg_graph *graph, void *hip_stream);
int32_t locrec_sg_synchronize(locrec_sg_graph *graph);
int32_t locrec_sg_profile_enable(locrec_sg_graph *graph, int32_t on);
/* Summed duration of the sweep (SpMV) kernel and its launch count; resets. */
int32_t locrec_sg_profile_read(locrec_sg_graph *graph, double *out_sweep_ms, int64_t *out_launches);

/* ===================================================================== */
/* Several devices in one process (SURVEY.md 8b "locrec_set_devices", 8e).
 *
 * The Scala host the reference prescribes is ONE JVM: it cannot start a process per GPU.  These entry points are the
 * multi-GPU forms of both paths inside the library - per-device streams, events and peer access, threads only around
 * calls that block - so that one `new KnnRecommender(...)` can use every GPU of the node (csrc/multi.hip).
 * n_devices = 0 (and device_ids NULL) takes the list given to locrec_set_devices, or the current device when there
 * is none.  A device may be listed more than once (logical shards on one GPU: how the one-GPU test box rehearses it).
 */
int32_t locrec_set_devices(int32_t n_devices, const int32_t *device_ids); /* n_devices = 0 forgets the list */

/*
 * KNN, persons (queries) sharded / candidates on every device (BASELINE.json configs[3]).  Create = locrec_knn_create's
 * arguments; set-up is the block all-gather of the CSR arrays in tiles: device r uploads tile r of every array over
 * PCIe, every other device pulls it over xGMI (hipMemcpyPeerAsync) as soon as it has landed, while later tiles are
 * still uploading; then every device builds its index from device arrays (locrec_knn_create_from_device), all at once.
 * The batch calls hand replica r the r-th contiguous share of the queries and run the replicas concurrently; results
 * are those of the single-device calls of the same names, bit for bit (same kernels, same per-query order).
 */
typedef struct locrec_knn_replicas locrec_knn_replicas;
int32_t locrec_knn_replicas_create(
    int32_t n_devices, const int32_t *device_ids, int64_t n, const int64_t *person_ids,
    const int64_t *p_rowptr, const int32_t *p_idx, const double *p_val, int32_t p_dim,
    const int64_t *c_rowptr, const int32_t *c_idx, const double *c_val, int32_t c_dim,
    const int64_t *r_rowptr, const int64_t *r_place, const int64_t *r_rating, locrec_knn_replicas **out_replicas);
void locrec_knn_replicas_destroy(locrec_knn_replicas *replicas);
/* number of replicas; the first replica's index (owned by the handle): single requests go to it */
int32_t locrec_knn_replicas_info(const locrec_knn_replicas *replicas, int32_t *out_devices, locrec_knn_index **out_first);
int32_t locrec_knn_replicas_recommend_batch(
    locrec_knn_replicas *replicas, int64_t nq, const int64_t *person_ids,
    double place_weight, double category_weight, int64_t k_nearest,
    int64_t *out_offsets, int64_t *out_place_ids, double *out_estimated_ratings, int64_t *inout_capacity);
int32_t locrec_knn_replicas_query_batch(
    locrec_knn_replicas *replicas, int64_t nq, const int64_t *person_ids,
    double place_weight, double category_weight, int64_t k_nearest,
    int64_t *out_person_ids, double *out_similarities, int64_t *out_counts);

/*
 * SG, one graph with its rows sharded over the devices (BASELINE.json configs[4]).  by_target = 0: rows of P (sources)
 * sharded, the T live entries of sigma all-REDUCED per sweep - the additions run in device order on every device, so
 * the result is deterministic (to rounding of the unsharded one); by_target = 1: rows of P^T sharded, owned entries
 * all-GATHERED, bit-identical to one device.  The exchange is a kernel reading the peers' sigma buffers directly
 * (peer access; staged peer copies without it), ordered by events: no host synchronisation in a fixed-sweep run,
 * one 8-byte read per sweep (isConverged, StochasticRecommender.scala:99) otherwise.  Results as locrec_sg_recommend.
 */
typedef struct locrec_sg_sharded locrec_sg_sharded;
int32_t locrec_sg_sharded_create(int32_t n_devices, const int32_t *device_ids, int64_t n_edges, const int64_t *source_ids,
                                 const int64_t *target_ids, const double *balanced_weights, int32_t by_target,
                                 locrec_sg_sharded **out_sharded);
void locrec_sg_sharded_destroy(locrec_sg_sharded *sharded);
int32_t locrec_sg_sharded_info(const locrec_sg_sharded *sharded, int32_t *out_devices, int32_t *out_peer_access,
                               int64_t *out_exchanged_entries, int64_t *out_vertices);
int32_t locrec_sg_sharded_recommend(locrec_sg_sharded *sharded, int64_t vertex_id, double alpha, double epsilon,
                                    int64_t max_iterations, int64_t *out_ids, double *out_probabilities,
                                    int64_t *inout_count, int64_t *out_iterations, int32_t *out_converged);
int32_t locrec_sg_sharded_iterate_async(locrec_sg_sharded *sharded, int64_t vertex_id, double alpha, double epsilon,
                                        int64_t max_iterations);
int32_t locrec_sg_sharded_sweeps_async(locrec_sg_sharded *sharded, int64_t vertex_id, double alpha, int64_t sweeps);
/* the result as device `which` of the handle holds it (all hold the same x) */
int32_t locrec_sg_sharded_fetch(locrec_sg_sharded *sharded, int32_t which, int64_t *out_ids, double *out_probabilities,
                                int64_t *inout_count, int64_t *out_iterations, int32_t *out_converged);

/* ===================================================================== */
/* The producers of the two paths' inputs (f-2, f-4).       */
/* Stateless; they run on the current device (locrec_set_device) and      */
/* return when the outputs are complete.  `mem` says where ALL arrays of  */
/* a call live (LOCREC_MEM_HOST / LOCREC_MEM_DEVICE); counts always come  */
/* back through host pointers.                                            */

/*
 * RatingsBuilder.calcRatings (knn/RatingsBuilder.scala:32-48): visits (person_id, entity_id) ->
 * rows (person_id, entity_id, rating = count("*")) whose SQL rank() by rating descending within
 * the person is <= top_n (ties share a rank, so a tie straddling top_n is kept whole).  Rows
 * come back ordered by (person_id, entity_id) - the reference leaves the order undefined.
 * The three outputs need room for n rows; *out_count = rows written.
 */
int32_t locrec_calc_ratings(int64_t n, const int64_t *person_ids, const int64_t *entity_ids, int64_t top_n,
                            int32_t mem, int64_t *out_person_ids, int64_t *out_entity_ids,
                            int64_t *out_ratings, int64_t *out_count);

/*
 * RatingVectorsBuilder.calcRatingVectors (knn/RatingVectorsBuilder.scala:10-25,52-84): ratings
 * (person_id, entity_id, rating: Long) -> one SparseVector per person as CSR: out_person_ids
 * ascending, indices ascending within a person, an index that repeats keeps its FIRST rating in
 * input order (the TreeSet of :43-50 does not replace), values = rating.toDouble (:69),
 * *out_size = max entity id + 1 (:27-34).  An id outside Int range -> LOCREC_E_ARITHMETIC with
 * the reference's message; a negative id or max id == Int.MaxValue -> LOCREC_E_INVALID_ARG (the
 * SparseVector constructor's require()s).  out_person_ids / out_idx / out_val need n entries,
 * out_rowptr n + 1.  The outputs are exactly locrec_knn_create[_from_device]'s vector inputs.
 */
int32_t locrec_calc_rating_vectors(int64_t n, const int64_t *person_ids, const int64_t *entity_ids,
                                   const int64_t *ratings, int32_t mem, int64_t *out_person_ids,
                                   int64_t *out_rowptr, int32_t *out_idx, double *out_val,
                                   int64_t *out_npersons, int64_t *out_nnz, int64_t *out_size);

/*
 * StochasticGraphBuilder.buildWithBalancedWeights (stochastic/StochasticGraphBuilder.scala:8-28):
 * family f's `weight` times betas[f], families concatenated in the given order - the
 * (source_id, target_id, balanced_weight) edge list locrec_sg_create takes.  betas[] / counts[]
 * and the three arrays of per-family pointers are always host memory; `mem` is about the
 * pointed-to columns and the outputs (sum of counts entries each).  n_families <= 0 fails as
 * betas.head of an empty Seq does.
 */
int32_t locrec_build_balanced_edges(int32_t n_families, const double *betas, const int64_t *counts,
                                    const int64_t *const *source_ids, const int64_t *const *target_ids,
                                    const double *const *weights, int32_t mem, int64_t *out_source_ids,
                                    int64_t *out_target_ids, double *out_balanced_weights);

/*
 * PlaceVisits.calcPlaceVisits (PlaceVisits.scala:11-46): location visits with timestamp >=
 * visits_from (:24) joined with the places of the same region_id (:31) and kept where
 * Location.distanceMeters (Location.scala:30-38, haversine on a 6371 km sphere) <= max_meters
 * (100 in the reference, PlaceVisits.scala:127).  visits_from is calcVisitsFromTimestamp's result
 * (:48-58), computed by the caller in its session time zone; timestamps are opaque int64.
 * Output columns as :40-46 (person_id, timestamp, place_id, region_id, category_id), ordered by
 * (visit row, place row) - the reference leaves the order undefined.  *inout_count: capacity in,
 * number of matches out (may exceed the capacity; call with 0 to size the buffers).
 * A latitude / longitude outside its range (or NaN) in a row that takes part in the join fails as
 * Location's require does: LOCREC_E_INVALID_ARG with the reference's message, *inout_count =
 * -(1 + visit row) or -(1 + n_visits + place row).
 */
int32_t locrec_calc_place_visits(int64_t n_visits, const int64_t *v_person_ids, const int64_t *v_timestamps,
                                 const double *v_latitudes, const double *v_longitudes,
                                 const int64_t *v_region_ids, int64_t n_places, const int64_t *p_ids,
                                 const double *p_latitudes, const double *p_longitudes,
                                 const int64_t *p_region_ids, const int64_t *p_category_ids,
                                 int64_t visits_from, double max_meters, int32_t mem,
                                 int64_t *out_person_ids, int64_t *out_timestamps, int64_t *out_place_ids,
                                 int64_t *out_region_ids, int64_t *out_category_ids, int64_t *inout_count);

/* Location.distanceMeters (Location.scala:30-38) of n pairs with the join's own device code
 * (LocationTest.scala:8-27 runs against it); NaN for a pair with an out-of-range coordinate. */
int32_t locrec_distance_meters(int64_t n, const double *lat1, const double *lon1, const double *lat2,
                               const double *lon2, int32_t mem, double *out_meters);

/*
 * printRecommendations of both mains (knn/KnnRecommenderMain.scala:90-101,
 * stochastic/StochasticRecommenderMain.scala:64-75):
 * places.where(region_id === target_region_id) JOIN recommendations ON id, ORDER BY score DESC,
 * LIMIT max_recommendations.  Rows whose id is not a place of the target region (persons,
 * categories, places elsewhere) drop out in the join; a place listed twice counts once; ties
 * (Spark: undefined) are ordered by id ascending; NaN sorts above every number, as in Spark.
 * Outputs need room for min(n, max_recommendations) rows; *out_count = rows written.
 */
int32_t locrec_rank_recommendations(int64_t n, const int64_t *ids, const double *scores, int64_t n_places,
                                    const int64_t *place_ids, const int64_t *place_region_ids,
                                    int64_t target_region_id, int64_t max_recommendations, int32_t mem,
                                    int64_t *out_ids, double *out_scores, int64_t *out_count);

#ifdef __cplusplus
}
#endif
#endif /* LOCREC_H */
