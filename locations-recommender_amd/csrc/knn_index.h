// knn_index.h -- the KNN index handle, shared by knn.hip (scan / top-K / aggregation kernels) and
// knn_large.hip (K beyond the LDS lists: device radix sort + place-major aggregation).
#pragma once

#include <algorithm>

#include "common.h"

namespace locrec {

namespace cfg {  // constants shared by the scans (knn.hip, knn_ht.h) and the index build (knn_build.hip)
constexpr int kDirectMaxBytes = 16384;  // a family's panel is direct-indexed up to this size
constexpr int kPopTable = 4096;         // renumbered indices below this get a direct u16 slot table (row scan)
constexpr int kHtQt = 16;               // query tile of the head / tail form
constexpr int kHtHead = 512;            // its default head width
// rows a plane of the place panel has room for (knn_ht.h): knn_scan_ht's plane stride is the constant kHtHead * 16
__host__ __device__ constexpr int ht_plane_rows(int h) { return h > kHtHead ? h : kHtHead; }
constexpr int kHtCatRows = 64;          // rows reserved for the category panel (c_dim <= 64)
constexpr int kHtNP = 4;                // place groups knn_scan_ht always loads per slice (images are padded for it)
constexpr int kHtColdBytes = 512;       // device buffer of the launch's HtCold (>= sizeof(HtCold), asserted in knn.hip)
}  // namespace cfg

struct DevFamily {
    DevBuf<uint32_t> sell;
    DevBuf<double> sell_val;
    DevBuf<int64_t> sell_off;
    DevBuf<int32_t> sell_w;
    DevBuf<double> norm;
    DevBuf<float> inorm32;
    DevBuf<int64_t> csr_ptr;
    DevBuf<int32_t> csr_idx;
    DevBuf<double> csr_val;
    int32_t dim = 0, vbits = 0;
    int64_t scan_bytes = 0;
    std::vector<int32_t> nnz;   // host copy, per row
    // popularity split (PACKED place family): indices are renumbered by descending frequency, rows of
    // a slice are sorted by their count of popular (< pop_h) indices, and sell_split[slice] is the
    // number of leading dwordx4 element groups in which EVERY lane holds popular indices only
    DevBuf<int32_t> sell_split;
    int32_t pop_h = 0;
};


// Head / tail form of the place family (knn_ht.h): head elements (index < h after the popularity
// renumbering) in SELL-64 rows, tail elements inverted into one posting list per place.
struct HtIndex {
    bool ready = false;
    int32_t h = 0;        // head dimensions
    int32_t qt = 16;      // query tile the element format was built for (row bytes = 2 * qt)
    int32_t waves = 8;    // waves per block of knn_scan_ht
    DevBuf<uint32_t> p_sell, c_sell;      // value << 16 | index * row bytes
    DevBuf<int64_t> p_off, c_off;
    DevBuf<int32_t> p_w, c_w;
    DevBuf<int64_t> post_ptr;             // [p_dim - h + 1]
    DevBuf<uint32_t> post;                // row << 8 | value, rows ascending inside a place
    DevBuf<uint4> desc;                   // per slice: HtSliceDesc (knn_ht.h)
    DevBuf<uint32_t> rid;                 // ix->rid padded to whole slices
    DevBuf<uint32_t> ss;                  // per row: sum of squares of the place vector | of the category vector << 16
    DevBuf<unsigned char> cold;           // HtCold of the launch in flight
    DevBuf<double> seed;                  // [nq] threshold seeds of the launch (knn_scan_ht), or unused
    bool v1 = false;                      // LOCREC_KNN_HT_V1: the first form (knn_scan MODE 3) instead of knn_scan_ht
    DevBuf<int64_t> tail_hits;            // per row: postings its tail places hold in total
    std::vector<int64_t> tail_hits_ps;    // host prefix sums [n + 1]
    std::vector<int32_t> tail_nnz;        // host, per row
    int64_t scan_bytes = 0;
    // per-batch workspaces (grow-only)
    DevBuf<uint32_t> hits, off;
    DevBuf<int64_t> tile_base;
    DevBuf<int32_t> err;
};

}  // namespace locrec

using locrec::DevBuf;
using locrec::DevFamily;
using locrec::KernelProfile;

struct locrec_knn_index {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t n = 0;
    int32_t nslices = 0;
    // slice range [cand_slice0, cand_slice1) the scans cover: everything, except while a candidate
    // shard is being queried (locrec_knn_query_shard)
    int32_t cand_slice0 = 0, cand_slice1 = 0;
    std::vector<int64_t> slice_cost;  // prefix sums of the slices' stored elements (lazy; shard boundaries)
    bool packed = false;
    bool pack16 = false;  // every dot < 65536: packed 16-bit multiply-add is exact
    bool force_hash = false;
    bool no_wide_block = false;  // LOCREC_KNN_NO_WIDE_BLOCK: never run a long-query tile with 16 waves per block
    bool no_dense_hash = false;  // LOCREC_KNN_NO_DENSE_HASH: never trade hash sparsity for a wider tile
    int qt_max = 16;  // LOCREC_KNN_QT caps the query tile (tuning / tests)
    int waves16 = 8;  // LOCREC_KNN_WAVES: waves per block of the PACK16 kernels (4 or 8)
    DevFamily fp, fc;
    locrec::HtIndex ht;
    bool no_ht = false;  // LOCREC_KNN_NO_HT: never use the head / tail form
    DevBuf<uint32_t> rid;
    DevBuf<int64_t> ids_by_rank;
    DevBuf<int32_t> row_of_rid;
    DevBuf<int64_t> r_ptr, r_place;
    DevBuf<double> r_rating;
    int64_t max_r_nnz = 0;
    // ratings transposed (place-major), for aggregation over many neighbours
    DevBuf<int32_t> r_pidx;       // per rating row: index of its place in cplace_ids
    DevBuf<int64_t> cplace_dev;   // cplace_ids on the device
    DevBuf<int64_t> cp_ptr;       // [n_cplaces + 1]
    DevBuf<int32_t> cp_row;
    DevBuf<double> cp_rating;
    std::vector<int64_t> cplace_ids;  // sorted distinct place ids
    // large-K workspaces
    DevBuf<uint64_t> lk_keys, lk_keys_out;
    DevBuf<uint32_t> lk_vals, lk_vals_out;
    DevBuf<unsigned char> lk_temp;
    DevBuf<double> lk_w, lk_ws, lk_ss;
    DevBuf<int64_t> lk_out_place, lk_out_n;  // rated places in ascending order, their number (lk_finish_*)
    DevBuf<double> lk_out_est;
    DevBuf<int32_t> lk_tile_cnt;             // rated places per tile of kFinishTile
    // segment table of the place-major ratings (knn_large.hip, lazy): raters of a place in runs of <= 4096
    DevBuf<int64_t> lk_seg_begin, lk_seg_end;
    DevBuf<int32_t> lk_place_seg0;
    DevBuf<double> lk_seg_ws, lk_seg_ss;
    int32_t lk_nsegs = -1;
    // batched large K (knn_large.hip, knn_large_recommend_batch): a tile of queries' similarities against all
    // candidates, dense query tables, per-tile aggregation workspaces, and the batch's compacted result
    DevBuf<double> lkb_S, lkb_qd_p, lkb_qd_c, lkb_seg_ws, lkb_seg_ss, lkb_ws, lkb_ss, lkb_est;
    DevBuf<int32_t> lkb_cand, lkb_tile_cnt;
    DevBuf<int64_t> lkb_place;
    std::vector<int64_t> lkb_off;   // [nq + 1] offsets of the resident result, in processing order
    bool have_lkb = false;          // a batched large-K result is resident (locrec_knn_fetch_recommend)
    bool lkb_deferred = false;      // ... or 1024 < K < N - 1: the queries are served one by one at fetch time
    // per-row format fallback (knn_build.hip): rows that by themselves break the head / tail form's legality (a count of
    // 256 or more, a sum of squares of 65,536 or more) are all padding in the packed images; the side kernels of knn.hip
    // score them from the plain CSR.  Empty when the index has none (or too many: then the whole index is demoted).
    std::vector<int32_t> wide_rows;     // ascending
    std::vector<unsigned char> is_wide; // per row (empty = no wide rows)
    DevBuf<int32_t> wide_rows_dev;
    // the wide rows' two vectors once more, lane-major (SELL-64 over the list of wide rows: element j of wide row w at
    // side_off[w / 64] + j * 64 + w % 64, (index, integer value), index -1 = padding): knn_side_topk reads them coalesced
    DevBuf<int2> side_p, side_c;
    DevBuf<int32_t> side_off_p, side_off_c, side_w_p, side_w_c;
    bool row_is_wide(int32_t r) const { return !is_wide.empty() && is_wide[(size_t)r] != 0; }
    std::vector<int64_t> ids_row;       // person id of each row
    std::vector<int32_t> row_of_input;  // create-time position -> row
    // person id -> row: binary search over the ids in ascending order (ids_sorted[k] lives at row row_by_rank[k])
    std::vector<int64_t> ids_sorted;
    std::vector<int32_t> row_by_rank;
    int32_t row_of_person(int64_t id) const
    {
        const auto it = std::lower_bound(ids_sorted.begin(), ids_sorted.end(), id);
        return it == ids_sorted.end() || *it != id ? -1 : row_by_rank[(size_t)(it - ids_sorted.begin())];
    }
    // workspaces (grow-only)
    DevBuf<int32_t> qrows;
    DevBuf<int32_t> qrows_patch;  // a batch's query rows with the too-long ones replaced (enqueue_topk)
    std::vector<int32_t> qrows_host;  // host image of qrows (list forms): the head / tail pre-pass is sized from it
    DevBuf<double> part_s;
    DevBuf<uint32_t> part_rid;
    DevBuf<int32_t> part_cnt;
    DevBuf<double> part2_s;      // second-level lists of a two-level merge
    DevBuf<uint32_t> part2_rid;
    DevBuf<int32_t> part2_cnt;
    DevBuf<int64_t> out_ids, out_cnt;
    DevBuf<double> out_sims;
    DevBuf<int32_t> out_rows;
    DevBuf<double> qd_p, qd_c;    // knn_scan_dense: the long query's vectors as dense arrays (all zero between requests)
    bool force_dense_query = false;
    bool no_direct8 = false, direct8_attr = false;  // knn_scan1_direct8 (LOCREC_KNN_NO_DIRECT8)
    bool no_seed = false;         // LOCREC_KNN_NO_SEED
    // read by knn_read_env like the ones above (the only place of the KNN sources that looks at the environment)
    bool force_generic = false;   // LOCREC_KNN_FORCE_GENERIC
    bool no_pack16 = false;       // LOCREC_KNN_NO_PACK16
    bool no_pop = false;          // LOCREC_KNN_NO_POP
    bool no_row_fallback = false; // LOCREC_KNN_NO_ROW_FALLBACK
    int32_t env_ht_h = 0;         // LOCREC_KNN_HT_H (0 = the default head width)
    int32_t env_pop_h = 0;        // LOCREC_KNN_POP_H
    int32_t env_blocks = 0;       // LOCREC_KNN_BLOCKS (tuning: target block count of a batched scan)
    int32_t env_flush = 0;        // LOCREC_KNN_FLUSH (tuning: drain interval of the barrier-free mode; a power of two <= 64)
    int32_t env_enter = -1;       // LOCREC_KNN_ENTER (tuning: its entry threshold)
    bool no_tile_special = false; // LOCREC_KNN_NO_TILE_SPECIAL: a batch's wide / too-long queries one by one (dense scan + full sort)
    int32_t seed_sample_slices = 1024;  // candidate slices the seeding pass samples per tile (kHtSeedSampleSlices)
    int32_t seed_min_slices = 4096;  // candidate slices from which a batched scan gets a threshold-seeding pass
    int64_t dense_query_scans = 0;
    DevBuf<double> S1;            // single-request path: similarity of every row
    DevBuf<uint32_t> hist1;
    DevBuf<int32_t> sel1;         // b*, above, total, list_n, overflow
    DevBuf<int32_t> tile_ovf;     // overflow flags of a tile of special queries (enqueue_topk)
    // ... and its per-column workspaces: 16 histograms, selection records, collect lists, result slots
    DevBuf<uint32_t> tile_hist, tile_list_r;
    DevBuf<int32_t> tile_sel;
    DevBuf<double> tile_list_s;
    DevBuf<int64_t> tile_slots;
    DevBuf<double> list1_s;
    DevBuf<uint32_t> list1_r;
    bool no_single = false;       // LOCREC_KNN_NO_SINGLE: always use the tiled path (tests)
    bool final1_attr = false;
    bool hist1_dirty = true;  // the single-request histogram / counters need a memset before the next scan
    // pinned host staging for the read-back of small results (one request): copies into pinned
    // memory are truly asynchronous, so a request pays for ONE synchronisation instead of one
    // blocking pageable copy per field
    unsigned char *h_stage = nullptr;
    unsigned char *h_stage_dev = nullptr;  // the same buffer as the device addresses it (knn_pack_host writes into it)
    bool no_pack = false;                  // LOCREC_KNN_NO_PACK: read small results back with one copy per array
    bool single_direct = false;            // the pending single request's result is already in h_stage (knn_final1)
    static constexpr size_t kStageBytes = 160 * 1024;
    ~locrec_knn_index()
    {
        if (h_stage) (void)hipHostFree(h_stage);
        // also reached by every early `return fail(...)` of locrec_knn_create (unique_ptr)
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
    bool no_fast = false;         // LOCREC_KNN_NO_FAST: synchronous insertion in every slice
    bool last_scan_fast = false;
    DevBuf<int32_t> scan_overflow;  // queue overflows of the last tiled scan (fast path)
    struct TiledRequest {
        const int32_t *qrows_dev;
        int32_t qrow0;
        int64_t nq;
        int max_p, max_c;
        double pw, cw;
        int64_t k;
        bool mark_absent;
    } last_tiled{};
    bool single_pending = false;  // a single-request result whose overflow flag has not been read yet
    int32_t single_qrow = 0;
    double single_pw = 0, single_cw = 0;
    DevBuf<int64_t> agg_place, agg_n;
    DevBuf<double> agg_est;
    DevBuf<int32_t> agg_overflow;
    // batched makeRecommendations0: offsets of the compacted rows and the compacted rows themselves
    DevBuf<int64_t> agg_off, agg_dense_place;
    DevBuf<double> agg_dense_est;
    int agg_M = 0;              // per-query capacity of agg_place / agg_est of the last batched aggregation
    bool have_agg = false;      // a batched aggregation is resident (locrec_knn_fetch_recommend)
    int32_t agg_first = 0;      // its first internal row (range form), or -1 when rows came from agg_rows
    std::vector<int32_t> agg_rows;  // query rows of the batch form, in processing order
    double agg_pw = 0, agg_cw = 0;
    KernelProfile prof;
    // plan of the last batched scan (locrec_knn_scan_plan): kernel 1 = knn_scan (row scan over a hashed /
    // direct query panel), 2 = knn_scan_ht (dense head panel + inverted tail), 3 = knn_scan MODE 3; mode 0/1/2 = GENERIC/PACK32/PACK16
    int last_plan_kernel = 0, last_plan_mode = 0, last_plan_qt = 0, last_plan_waves = 0;
    int64_t last_nq = 0, last_k = 0;
    bool have_result = false;
};

namespace locrec {

// knn.hip: the LOCREC_KNN_* tuning switches of a new handle
void knn_read_env(locrec_knn_index *ix);

// knn_build.hip: the index built on the device from DEVICE arrays (locrec_knn_create_from_device)
int32_t knn_build_device(int64_t n, const int64_t *ids, const int64_t *p_ptr, const int32_t *p_idx, const double *p_val,
                         int32_t p_dim, const int64_t *c_ptr, const int32_t *c_idx, const double *c_val, int32_t c_dim,
                         const int64_t *r_ptr, const int64_t *r_place, const int64_t *r_rating, locrec_knn_index **out);

// knn.hip: similarity of every row against the person at row qrow -> ix->S1 (0 = not a candidate)
// and the 65536-bin histogram ix->hist1; enqueued on the handle's stream.
int32_t knn_enqueue_dense(locrec_knn_index *ix, int32_t qrow, double pw, double cw);
int32_t knn_large_topk_device(locrec_knn_index *ix, int32_t qrow, double pw, double cw, int64_t k, int64_t slot);
// knn.hip: the handle's pinned staging buffer (h_stage / h_stage_dev), allocated at first use; NULL if that fails
unsigned char *knn_stage(locrec_knn_index *ix);

// knn_large.hip
int32_t knn_large_topk(locrec_knn_index *ix, int32_t qrow, double pw, double cw, int64_t k,
                       int64_t *out_ids, double *out_sims, int64_t *inout_count);
int32_t knn_large_recommend(locrec_knn_index *ix, int32_t qrow, double pw, double cw, int64_t k,
                            int64_t *out_places, double *out_ratings, int64_t *inout_count);
int32_t knn_large_aggregate(locrec_knn_index *ix, const double *w_host, int64_t *out_places, double *out_ratings,
                            int64_t *inout_count);
// makeRecommendations for many persons at K >= N - 1 (every positive-similarity person is a neighbour): tiles of 16
// queries, no top-K, results resident in ix->lkb_place / lkb_est / lkb_off
int32_t knn_large_recommend_batch(locrec_knn_index *ix, const int32_t *rows, int64_t nq, double pw, double cw);
// the queries of a batch that cannot ride a packed tile (wide rows, rows too long for any LDS tile), 16 at a time:
// similarities of the tile against every row (plain CSR, dense fp64 query tables), one contiguous column per query;
// column t's histogram into hist1
int32_t knn_large_scan_tile(locrec_knn_index *ix, const int32_t *rows, int nt, double pw, double cw);
int32_t knn_large_tile_hists(locrec_knn_index *ix, int nt, uint32_t *hist, const double **cols);

}  // namespace locrec
