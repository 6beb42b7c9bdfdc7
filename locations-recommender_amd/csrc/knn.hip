// knn.hip -- KNN hot path on gfx950: query(s)-vs-all sparse cosine over place and
// category rating vectors, weighted combine, per-query top-K, similarity-weighted
// rating aggregation.
//
// Replaces (paths relative to recommender/src/main/scala/com/github/tashoyan/recommender/):
//   knn/Distance.scala:7-16                    vectorLength, cosineSimilarity
//   knn/KnnRecommender.scala:76-96             findSimilarPersons0 (scan + "> 0" filter)
//   knn/KnnRecommender.scala:27-49             outer join / fill 0 / weights / orderBy.limit(K)
//   knn/KnnRecommender.scala:51-70             makeRecommendations0 (aggregation)
// plus Spark 3.1.2 BLAS.dot(sparse,sparse) (third party, called at Distance.scala:8).
//
// Data layout in HBM (built once in locrec_knn_create; nothing is re-read from the host):
//   * persons are permuted into ROW order = ascending (nnz_place, nnz_category); 64
//     consecutive rows form a SLICE = one wave, lane = candidate row (SELL-64 with a
//     global length sort, so a slice is padded by < 4 elements per row on average).
//   * PACKED format (chosen when every value is an integer count that fits, the
//     shipped data: RatingVectorsBuilder.scala:69 `rating.toDouble`):
//       one uint32 per non-zero = index << vbits | value, stored
//       [slice][j/4][lane][4]  -> one dwordx4 per lane, 1 KiB per wave instruction.
//     Integer products and sums are exact, so the dot is the reference's double
//     bit for bit in any order; 4 B/nnz instead of the reference's 12 B.
//   * GENERIC format (any finite doubles, e.g. DistanceTest's negative values):
//       int32 index [slice][j][lane] + fp64 value [slice][j][lane]; each lane adds
//       its products in ascending index order -- exactly BLAS.dot's order.
//   * per row: fp64 norms of both families (computed once, on the device), rid =
//     rank of the person id among all ids (top-K tie-break = person_id asc, H1).
//   * a plain CSR copy in row order (query staging, aggregation) and the ratings CSR.
//
// Scan kernel (the dominant kernel; HBM-streaming over candidates):
//   grid = (candidate chunks, query tiles); block = 4 waves sharing one tile of QT
//   queries.  The tile's queries are expanded once per block into an LDS PANEL:
//   a slot map (direct for small dimensions, open-addressing hash otherwise) from
//   index -> panel row, and panel[row][q] = value of query q at that index (0 if
//   absent; a miss maps to an all-zero row, so the inner loop has no branch).
//   Each lane walks its candidate row once and does QT multiply-adds per non-zero:
//   candidate bytes are read once per TILE, not once per query.
//   The epilogue forms ps = dot/(|c|*|q|) (one multiply, one divide, no FMA), keeps
//   "> 0" (KnnRecommender.scala:91), combines ps*pw + cs*cw (:43-45) and feeds a
//   per-query LDS top-K (threshold filter + block-wide bitonic compaction).
//   Per-chunk lists are merged by knn_merge.

#include <algorithm>
#include <chrono>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <new>
#include <numeric>
#include <type_traits>
#include <unordered_map>

#include "knn_index.h"

namespace {

using namespace locrec;

constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr int kMergeCap = 8192;      // entries one merge block sorts in LDS
constexpr int kAggCap = 4096;        // rating rows one aggregation block sorts in LDS (28 B each)
constexpr int kDirectMaxBytes = cfg::kDirectMaxBytes;
constexpr int kLdsSoftLimit = 64 * 1024;
constexpr int kLdsHardLimit = 160 * 1024;

struct Family {          // one of {place, category}, device pointers
    const uint32_t *sell;     // PACKED: packed elements; GENERIC: int32 indices
    const double *sell_val;   // GENERIC only
    const int64_t *sell_off;  // [nslices] element offset of the slice
    const int32_t *sell_w;    // [nslices] slice width (PACKED: multiple of 4)
    const double *norm;       // [nrows] Distance.vectorLength
    const float *inorm32;     // [nrows] (float)(1/norm), 0 for an absent vector (prefilter only)
    const int64_t *csr_ptr;   // [nrows+1]
    const int32_t *csr_idx;
    const double *csr_val;
    int32_t vbits;            // PACKED: low bits that hold the value
    int32_t direct;           // panel row = index (no hash)
    int32_t hlog2;            // hash capacity = 1 << hlog2
    int32_t rows_cap;         // panel rows (the last one is the all-zero row when hashed)
    int32_t off_hash;         // LDS byte offsets
    int32_t off_panel;
    const int32_t *sell_split;  // [nslices] leading element groups that are popular in every lane, or nullptr
    int32_t pop_h;              // indices < pop_h also have a direct u16 slot table in LDS (0: none)
    int32_t off_pop;
};

// MODE 3 (head / tail form, knn_ht.h): scan-side parameters
struct HtParams {
    const uint32_t *hits;       // [sum over tiles] lane << 21 | q << 16 | product, sorted by slice inside a tile
    const uint32_t *off;        // [ntiles][off_stride] offset of a slice's hits inside its tile's region
    const int64_t *tile_base;   // [ntiles + 1] first hit of the tile
    int32_t off_stride;         // slices of the scanned range + 1
    int32_t off_tail;           // LDS: W wave-private tail accumulators of 64 rows x QT u16 (swizzled)
    int32_t h;                  // head dimensions of the place panel
    int32_t c_rows;             // rows of the category panel (= c_dim)
};

struct ScanParams {
    Family fp, fc;
    const uint32_t *rid;      // [nrows] rank of the row's person id
    const int32_t *qrows;     // [nq] query rows, or nullptr: rows qrow0 .. qrow0+nq-1
    int32_t qrow0;
    int32_t nq;
    int32_t nrows, nslices;   // nslices = END of the scanned slice range (exclusive)
    int32_t slice0;           // first slice of the range (0 unless a candidate shard is scanned)
    int32_t flush_mask;       // barrier-free mode: queues are drained every flush_mask + 1 slices (pow2)
    int32_t enter_threads;    // ... and entered when <= this many threads of an 8-wave block hold a survivor
    int32_t slices_per_chunk, nchunks;
    int32_t K, S;             // S = pow2 LDS list size per query
    double pw, cw;
    double *part_s;           // [nq][nchunks][K]
    uint32_t *part_rid;
    int32_t *part_cnt;        // [nq][nchunks]
    int32_t off_cand_s, off_cand_rid, off_misc;
    int32_t lds_bytes;        // dynamic LDS size of this launch
    int32_t poison;           // debug: fill LDS with a pattern first (LOCREC_DEBUG_POISON)
    int32_t fast;             // after warm-up, survivors go to per-wave queues (no barrier per slice)
    int32_t off_queue;        // LDS: W queues of kQueueCap entries (s fp64, rid u32, q u32) + W counters
    int32_t *overflow;        // incremented when a queue overflowed: the host reruns without `fast`
    HtParams ht;              // MODE 3 (head / tail form, knn_ht.h)
};

#include "knn_device.h"

#include "knn_ht.h"

#include "knn_rowscan.h"

#include "knn_single.h"

#include "knn_merge.h"

#include "knn_side.h"

#include "knn_aggregate.h"

int pow2ceil(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}
int ceil_log2i(int64_t v)
{
    int l = 0;
    while (((int64_t)1 << l) < v) ++l;
    return l;
}

// host-side image of one family in row order
struct HostFamily {
    std::vector<int64_t> ptr;   // [n+1]
    std::vector<int32_t> idx;
    std::vector<double> val;
    int32_t dim = 0;
    int32_t vbits = 0;
    int32_t max_nnz = 0;
};

}  // namespace


namespace {

int64_t scan_bytes_total(const locrec_knn_index *ix)
{
    // what one query-vs-all pass always streams: both families' element arrays, slice
    // tables and per-row norms (+ the rid in the generic format, which has no prefilter)
    return ix->fp.scan_bytes + ix->fc.scan_bytes + (ix->packed ? 0 : ix->n * 4);
}

int32_t build_family_device(locrec_knn_index *ix, const HostFamily &h, DevFamily &d, bool packed)
{
    const int64_t n = ix->n;
    const int32_t nslices = ix->nslices;
    hipStream_t s = ix->stream;
    d.dim = h.dim;
    d.vbits = h.vbits;
    d.nnz.resize((size_t)n);
    for (int64_t r = 0; r < n; ++r) d.nnz[r] = (int32_t)(h.ptr[r + 1] - h.ptr[r]);
    std::vector<int64_t> off((size_t)nslices + 1, 0);
    std::vector<int32_t> wv((size_t)nslices, 0);
    for (int32_t sl = 0; sl < nslices; ++sl) {
        int w = 0;
        for (int64_t r = (int64_t)sl * 64; r < std::min<int64_t>(n, (int64_t)sl * 64 + 64); ++r)
            w = std::max(w, d.nnz[r]);
        if (packed) w = (w + 3) & ~3;
        wv[sl] = w;
        off[sl + 1] = off[sl] + (int64_t)w * 64;
    }
    const int64_t total = off[nslices];
    std::vector<uint32_t> sell((size_t)total, 0u);
    std::vector<double> sval;
    if (!packed) sval.assign((size_t)total, 0.0);
    for (int32_t sl = 0; sl < nslices; ++sl) {
        for (int lane = 0; lane < 64; ++lane) {
            const int64_t r = (int64_t)sl * 64 + lane;
            if (r >= n) break;
            const int64_t b = h.ptr[r];
            for (int j = 0; j < d.nnz[r]; ++j) {
                if (packed) {
                    const uint32_t e = ((uint32_t)h.idx[b + j] << h.vbits) | (uint32_t)h.val[b + j];
                    sell[off[sl] + (int64_t)(j >> 2) * 256 + lane * 4 + (j & 3)] = e;
                } else {
                    sell[off[sl] + (int64_t)j * 64 + lane] = (uint32_t)h.idx[b + j];
                    sval[off[sl] + (int64_t)j * 64 + lane] = h.val[b + j];
                }
            }
        }
    }
    LOCREC_TRY(d.sell.upload(sell, s));
    if (!packed) LOCREC_TRY(d.sell_val.upload(sval, s));
    LOCREC_TRY(d.sell_off.upload(off.data(), (size_t)nslices, s));
    LOCREC_TRY(d.sell_w.upload(wv, s));
    LOCREC_TRY(d.csr_ptr.upload(h.ptr, s));
    LOCREC_TRY(d.csr_idx.upload(h.idx, s));
    LOCREC_TRY(d.csr_val.upload(h.val, s));
    LOCREC_TRY(d.norm.alloc((size_t)n));
    LOCREC_TRY(d.inorm32.alloc((size_t)n));
    if (n > 0)
        hipLaunchKernelGGL(knn_norms, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d.csr_ptr.p,
                           d.csr_val.p, (int32_t)n, d.norm.p, d.inorm32.p);
    // bytes a scan always reads: the elements, the slice tables, one 4-byte norm per row
    // (the fp64 norm and the rid are touched only by the few survivors of the prefilter)
    d.scan_bytes = total * (packed ? 4 : 12) + (int64_t)nslices * 12 + n * (packed ? 4 : 8);
    LOCREC_HIP_TRY(hipStreamSynchronize(s));  // host vectors go out of scope
    return LOCREC_OK;
}

// Head / tail image (knn_ht.h).  hq: place family with the dimensions renumbered by popularity, rows
// in row order, indices ascending; hc: category family.  H: head dimensions; qt: query tile the
// element format is built for (panel row = 2 * qt bytes).
int32_t build_ht(locrec_knn_index *ix, const HostFamily &hq, const HostFamily &hc, int32_t H, int qt)
{
    const int64_t n = ix->n;
    const int32_t nslices = ix->nslices;
    hipStream_t s = ix->stream;
    locrec::HtIndex &ht = ix->ht;
    const int rsh = 4;  // byte offset of a 16-byte plane row (knn_ht.h)
    std::vector<int32_t> nh((size_t)n);
    ht.tail_nnz.assign((size_t)n, 0);
    for (int64_t r = 0; r < n; ++r) {
        const int32_t *b = hq.idx.data() + hq.ptr[r], *e = hq.idx.data() + hq.ptr[r + 1];
        nh[r] = (int32_t)(std::lower_bound(b, e, H) - b);
        ht.tail_nnz[r] = (int32_t)(e - b) - nh[r];
    }
    std::vector<int64_t> h_off[2];
    std::vector<int32_t> h_w[2];
    int which = 0;
    auto sell_of = [&](const HostFamily &h, const std::vector<int32_t> *limit, DevBuf<uint32_t> &d_sell,
                       DevBuf<int64_t> &d_off, DevBuf<int32_t> &d_w, int64_t &elements) -> int32_t {
        std::vector<int64_t> &off = h_off[which];
        std::vector<int32_t> &wv = h_w[which];
        ++which;
        off.assign((size_t)nslices + 1, 0);
        wv.assign((size_t)nslices, 0);
        auto len_of = [&](int64_t r) { return limit ? (*limit)[r] : (int32_t)(h.ptr[r + 1] - h.ptr[r]); };
        for (int32_t sl = 0; sl < nslices; ++sl) {
            int w = 0;
            for (int64_t r = (int64_t)sl * 64; r < std::min<int64_t>(n, (int64_t)sl * 64 + 64); ++r) w = std::max(w, len_of(r));
            w = (w + 3) & ~3;
            wv[sl] = w;
            off[sl + 1] = off[sl] + (int64_t)w * 64;
        }
        // padding: value 0 x panel row 0 = nothing; knn_scan_ht always loads kHtNP (kHtNC) groups of a slice,
        // so the image is followed by that many zero groups
        std::vector<uint32_t> sell((size_t)off[nslices] + (size_t)kHtNP * 256, 0u);
        for (int64_t r = 0; r < n; ++r) {
            const int64_t base = off[r >> 6], lane = r & 63, b = h.ptr[r];
            const int len = len_of(r);
            for (int j = 0; j < len; ++j)
                sell[base + (int64_t)(j >> 2) * 256 + lane * 4 + (j & 3)] =
                    ((uint32_t)h.val[b + j] << 16) | ((uint32_t)h.idx[b + j] << rsh);
        }
        elements = off[nslices];
        LOCREC_TRY(d_sell.upload(sell, s));
        LOCREC_TRY(d_off.upload(off.data(), (size_t)nslices, s));
        LOCREC_TRY(d_w.upload(wv, s));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
        return LOCREC_OK;
    };
    int64_t pe = 0, ce = 0;
    LOCREC_TRY(sell_of(hq, &nh, ht.p_sell, ht.p_off, ht.p_w, pe));
    LOCREC_TRY(sell_of(hc, nullptr, ht.c_sell, ht.c_off, ht.c_w, ce));
    if (pe / 4 >= ((int64_t)1 << 32) || ce / 4 >= ((int64_t)1 << 32)) return LOCREC_OK;  // descriptor offsets are 32-bit: no head / tail form
    {
        std::vector<uint4> desc((size_t)nslices);
        for (int32_t sl = 0; sl < nslices; ++sl)
            desc[sl] = make_uint4((uint32_t)(h_off[0][sl] / 4), (uint32_t)(h_off[1][sl] / 4),
                                  (uint32_t)(h_w[0][sl] / 4) | ((uint32_t)(h_w[1][sl] / 4) << 16), 0u);
        LOCREC_TRY(ht.desc.upload(desc, s));
        LOCREC_TRY(ht.cold.alloc(sizeof(HtCold)));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
    }
    // postings of the tail places: rows ascending inside a place
    const int64_t nt = std::max<int64_t>(0, (int64_t)hq.dim - H);
    std::vector<int64_t> pptr((size_t)nt + 1, 0);
    for (int64_t r = 0; r < n; ++r)
        for (int64_t e = hq.ptr[r] + nh[r]; e < hq.ptr[r + 1]; ++e) ++pptr[hq.idx[e] - H + 1];
    for (int64_t t = 0; t < nt; ++t) pptr[t + 1] += pptr[t];
    std::vector<uint32_t> post((size_t)pptr[nt]);
    {
        std::vector<int64_t> cur(pptr.begin(), pptr.end() - 1);
        for (int64_t r = 0; r < n; ++r)
            for (int64_t e = hq.ptr[r] + nh[r]; e < hq.ptr[r + 1]; ++e)
                post[(size_t)cur[hq.idx[e] - H]++] = ((uint32_t)r << 8) | (uint32_t)hq.val[e];
    }
    std::vector<int64_t> th((size_t)n, 0);
    ht.tail_hits_ps.assign((size_t)n + 1, 0);
    for (int64_t r = 0; r < n; ++r) {
        int64_t t = 0;
        for (int64_t e = hq.ptr[r] + nh[r]; e < hq.ptr[r + 1]; ++e) t += pptr[hq.idx[e] - H + 1] - pptr[hq.idx[e] - H];
        th[r] = t;
        ht.tail_hits_ps[r + 1] = ht.tail_hits_ps[r] + t;
    }
    {
        // sums of squares of both vectors of a row (exact integers < 65536 under PACK16): the scan derives
        // the f32 inverse norms of its bound and - for a survivor - the exact fp64 norms from them
        std::vector<uint32_t> ss((size_t)nslices * 64, 0u);  // whole slices: the scan loads them unconditionally
        for (int64_t r = 0; r < n; ++r) {
            double sp = 0, sc = 0;
            for (int64_t e = hq.ptr[r]; e < hq.ptr[r + 1]; ++e) sp += hq.val[e] * hq.val[e];
            for (int64_t e = hc.ptr[r]; e < hc.ptr[r + 1]; ++e) sc += hc.val[e] * hc.val[e];
            ss[r] = (uint32_t)sp | ((uint32_t)sc << 16);
        }
        LOCREC_TRY(ht.ss.upload(ss, s));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
    }
    LOCREC_TRY(ht.post_ptr.upload(pptr, s));
    LOCREC_TRY(ht.post.upload(post, s));
    LOCREC_TRY(ht.tail_hits.upload(th, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    // bytes one query-vs-all pass reads in this layout: head and category elements, every posting
    // once per query that holds its place is NOT per-candidate work - count the stored tail once
    ht.scan_bytes = (pe + ce) * 4 + (int64_t)post.size() * 4 + (int64_t)nslices * 24 + n * 8;
    ht.h = H;
    ht.qt = qt;
    ht.ready = true;
    return LOCREC_OK;
}

// KnnRecommender.scala:17-20
int32_t check_params(double pw, double cw, int64_t k)
{
    if (!(pw > 0 && pw < 1.0))
        return fail(LOCREC_E_INVALID_ARG, "requirement failed: Place weight must be in the interval (0; 1): %g", pw);
    if (!(cw > 0 && cw < 1.0))
        return fail(LOCREC_E_INVALID_ARG, "requirement failed: Category weight must be in the interval (0; 1): %g", cw);
    if (!(pw + cw == 1.0))
        return fail(LOCREC_E_INVALID_ARG, "requirement failed: Sum of weights must be 1.0: place: %g, category: %g", pw, cw);
    if (!(k > 0)) return fail(LOCREC_E_INVALID_ARG, "requirement failed: K nearest must be positive");
    return LOCREC_OK;
}

struct Plan {
    int qt = 0;
    int mode = 0;   // 0 GENERIC, 1 PACK32, 2 PACK16
    int waves = 4;  // waves per block
    int S = 0;
    size_t lds = 0;
    Family fp{}, fc{};
    int off_cand_s = 0, off_cand_rid = 0, off_misc = 0, off_queue = 0;
    int off_tail = 0;  // MODE 3
};

bool plan_family(const locrec_knn_index *ix, const DevFamily &d, int qt, int max_nnz, size_t elt,
                 Family &f, size_t &cursor, bool dense_hash = false)
{
    f.sell = d.sell.p;
    f.sell_val = d.sell_val.p;
    f.sell_off = d.sell_off.p;
    f.sell_w = d.sell_w.p;
    f.norm = d.norm.p;
    f.inorm32 = d.inorm32.p;
    f.csr_ptr = d.csr_ptr.p;
    f.csr_idx = d.csr_idx.p;
    f.csr_val = d.csr_val.p;
    f.vbits = d.vbits;
    f.sell_split = nullptr;
    f.pop_h = 0;
    f.off_pop = 0;
    const bool direct = !ix->force_hash && (size_t)d.dim * qt * elt <= (size_t)kDirectMaxBytes;
    f.direct = direct ? 1 : 0;
    if (direct) {
        f.hlog2 = 0;
        f.rows_cap = d.dim;
        f.off_hash = (int32_t)cursor;
    } else {
        const int keys = std::max(1, qt * max_nnz);
        f.rows_cap = keys + 1;
        f.off_hash = (int32_t)cursor;
        if (elt != 8) {
            if (f.rows_cap > 4096) return false;  // 12-bit panel row in the 32-bit hash entry
            // buckets of four entries, <= 1/2 key per bucket on average: a full bucket (the only
            // case that walks) has probability ~1e-4.  dense_hash (tiles of long queries, whose
            // sparse table would push the tile out of LDS): <= 2 keys per bucket, ~5 % full buckets.
            f.hlog2 = std::max(2, ceil_log2i(dense_hash ? ((int64_t)keys + 1) / 2 : 2 * (int64_t)keys));
            cursor += ((size_t)1 << f.hlog2) * 4 * sizeof(uint32_t);
        } else {
            f.hlog2 = std::max(4, ceil_log2i(2 * (int64_t)keys));
            cursor += ((size_t)1 << f.hlog2) * sizeof(uint2);
        }
    }
    cursor = (cursor + 15) & ~(size_t)15;
    f.off_panel = (int32_t)cursor;
    cursor += (size_t)f.rows_cap * qt * elt;
    cursor = (cursor + 15) & ~(size_t)15;
    if (!direct && elt != 8 && d.pop_h > 0 && d.sell_split.p) {  // popularity split (see DevFamily)
        f.sell_split = d.sell_split.p;
        f.pop_h = d.pop_h;
        f.off_pop = (int32_t)cursor;
        cursor += (size_t)d.pop_h * sizeof(unsigned short);
        cursor = (cursor + 15) & ~(size_t)15;
    }
    return true;
}

// Largest query tile whose LDS footprint fits; false if even QT = 1 does not.
// Candidates, best first: PACK16 tiles (two pairs per instruction, 8 waves share the tile so the
// LDS cost per wave halves), then PACK32, else GENERIC.
bool make_plan(const locrec_knn_index *ix, int64_t nq, int max_nnz_p, int max_nnz_c, int K, Plan &pl)
{
    const int S = std::max(64, pow2ceil(2 * K));
    struct Cand { int mode, qt, waves; bool dense; };
    std::vector<Cand> cands;
    // no point in a tile wider than the request
    const int qt_need = nq >= 32 ? 32 : pow2ceil((int)std::max<int64_t>(1, nq));
    if (ix->packed) {
        // a wider tile with a denser hash beats a narrower tile with a sparse one (measured on the
        // longest-row batch of cfg2: QT 16 dense vs QT 8 sparse)
        if (ix->pack16)
            for (int qt : {32, 16, 8})
                if (qt <= qt_need && qt <= ix->qt_max) {
                    cands.push_back({2, qt, ix->waves16, false});
                    if (!ix->no_dense_hash) cands.push_back({2, qt, ix->waves16, true});
                }
        for (int qt : {16, 8, 4, 2, 1})
            if (qt <= qt_need && qt <= ix->qt_max) cands.push_back({1, qt, 4, false});
    } else {
        for (int qt : {8, 4, 2, 1})
            if (qt <= qt_need && qt <= ix->qt_max) cands.push_back({0, qt, 4, false});
    }
    for (int pass = 0; pass < 2; ++pass) {
        const size_t limit = pass == 0 ? kLdsSoftLimit : kLdsHardLimit;
        for (const Cand &c : cands) {
            const size_t elt = c.mode == 0 ? 8 : (c.mode == 1 ? 4 : 2);
            Plan p;
            p.qt = c.qt;
            p.mode = c.mode;
            p.waves = c.waves;
            p.S = S;
            size_t cur = 0;
            if (!plan_family(ix, ix->fp, c.qt, max_nnz_p, elt, p.fp, cur, c.dense)) continue;
            if (!plan_family(ix, ix->fc, c.qt, max_nnz_c, elt, p.fc, cur, c.dense)) continue;
            p.off_cand_s = (int)cur;
            cur += (size_t)c.qt * S * sizeof(double);
            p.off_cand_rid = (int)cur;
            cur += (size_t)c.qt * S * sizeof(uint32_t);
            cur = (cur + 15) & ~(size_t)15;
            p.off_misc = (int)cur;
            cur += (size_t)c.qt * (3 * sizeof(double) + 6 * sizeof(int32_t)) + 16;
            cur = (cur + 15) & ~(size_t)15;
            p.off_queue = (int)cur;
            cur += (size_t)c.waves * kQueueCap * 16 + (size_t)c.waves * 4 + 16;
            p.lds = cur;
            // an 8-wave block may use twice the soft limit: the LDS per wave is what matters
            if (cur <= limit * (pass == 0 && c.waves == 8 ? 2 : 1) && cur <= (size_t)kLdsHardLimit) {
                // A tile too big for two blocks per CU (long queries) would leave the CU with 8 waves:
                // let 16 waves share the one panel instead (same code, twice the queues).
                if (c.mode == 2 && c.qt == 16 && c.waves == 8 && cur > (size_t)kLdsHardLimit / 2 && !ix->no_wide_block) {
                    const size_t cur16 = (size_t)p.off_queue + (size_t)16 * kQueueCap * 16 + (size_t)16 * 4 + 16;
                    if (cur16 <= (size_t)kLdsHardLimit) {
                        p.waves = 16;
                        p.lds = cur16;
                    }
                }
                pl = p;
                return true;
            }
        }
    }
    return false;
}

// MODE 3 (knn_ht.h): [category panel][place head panel][tail accumulators][lists][misc][queues]
bool make_plan_ht(const locrec_knn_index *ix, int K, Plan &pl)
{
    const locrec::HtIndex &ht = ix->ht;
    Plan p;
    p.mode = 3;
    p.qt = ht.qt;
    p.waves = ht.waves;
    p.S = std::max(64, pow2ceil(2 * K));
    auto fam = [&](const DevFamily &d, const DevBuf<uint32_t> &sell, const DevBuf<int64_t> &off, const DevBuf<int32_t> &w,
                   Family &f) {
        f = Family{};
        f.sell = sell.p;
        f.sell_off = off.p;
        f.sell_w = w.p;
        f.norm = d.norm.p;
        f.inorm32 = d.inorm32.p;
        f.csr_ptr = d.csr_ptr.p;
        f.csr_idx = d.csr_idx.p;
        f.csr_val = d.csr_val.p;
        f.direct = 1;
    };
    fam(ix->fp, ht.p_sell, ht.p_off, ht.p_w, p.fp);
    fam(ix->fc, ht.c_sell, ht.c_off, ht.c_w, p.fc);
    p.fp.rows_cap = ht.h;
    p.fc.rows_cap = ix->fc.dim;
    size_t cur = (size_t)kHtCatRows * p.qt * 2 + (size_t)cfg::ht_plane_rows(ht.h) * p.qt * 2;
    cur = (cur + 15) & ~(size_t)15;
    p.off_tail = (int)cur;
    cur += (size_t)p.waves * 64 * p.qt * 2;
    p.off_cand_s = (int)cur;
    cur += (size_t)p.qt * p.S * sizeof(double);
    p.off_cand_rid = (int)cur;
    cur += (size_t)p.qt * p.S * sizeof(uint32_t);
    cur = (cur + 15) & ~(size_t)15;
    p.off_misc = (int)cur;
    cur += (size_t)p.qt * (3 * sizeof(double) + 8 * sizeof(int32_t)) + 16;  // (incl. knn_scan_ht's packed f16 scale factors and -tau32)
    cur = (cur + 15) & ~(size_t)15;
    p.off_queue = (int)cur;
    cur += (size_t)p.waves * kQueueCap * 16 + (size_t)p.waves * 4 + 16;
    p.lds = cur;
    if (cur > (size_t)kLdsHardLimit) return false;
    pl = p;
    return true;
}

// Hits of a batch (knn_ht.h): tile bases from the per-row totals, then one block per tile.
int32_t enqueue_ht_prepass(locrec_knn_index *ix, const int32_t *qrows_dev, int32_t qrow0, int64_t nq, int qt,
                           int64_t total_hits)
{
    hipStream_t s = ix->stream;
    locrec::HtIndex &ht = ix->ht;
    const int ntiles = (int)((nq + qt - 1) / qt);
    const int32_t off_stride = ix->cand_slice1 - ix->cand_slice0 + 1;
    // (the hits of a batch vary widely at cfg2 - 283, 330, 381, ... 427 MB over the first eight batches: grow by
    // half, so that the buffer settles after the first batch instead of being freed and allocated again - a
    // device-wide synchronisation and a 400 MB hipMalloc - at every batch that is larger than all before it)
    // For a range of rows the largest batch of this size is known from the host's prefix sums: allocate for THAT once.
    if ((size_t)total_hits + 64 > ht.hits.n) {
        int64_t want = total_hits + total_hits / 3;
        const int64_t rows = (int64_t)ht.tail_hits_ps.size() - 1;
        if (rows >= nq && nq > 0) {
            int64_t widest = 0;
            for (int64_t i = 0; i + nq <= rows; ++i) widest = std::max(widest, ht.tail_hits_ps[(size_t)(i + nq)] - ht.tail_hits_ps[(size_t)i]);
            // (a list of rows is usually a window with a few rows replaced - the stand-ins of wide or too-long queries -:
            // the widest window plus a little covers the next such batches too)
            want = qrows_dev ? std::max(widest + widest / 16, want) : std::max(widest, total_hits);
        }
        LOCREC_TRY(ht.hits.reserve((size_t)want + 64));
    }
    LOCREC_TRY(ht.off.reserve((size_t)ntiles * off_stride));
    LOCREC_TRY(ht.tile_base.reserve((size_t)ntiles + 1));
    if (!ht.err.p) {
        LOCREC_TRY(ht.err.alloc(1));
        LOCREC_HIP_TRY(hipMemsetAsync(ht.err.p, 0, sizeof(int32_t), s));
    }
    hipLaunchKernelGGL(ht_tile_bases, dim3(1), dim3(1024), 0, s, ht.tail_hits.p, qrows_dev, qrow0, (int32_t)nq, qt, ntiles,
                       ht.tile_base.p);
    HtPreParams PP{};
    PP.csr_ptr = ix->fp.csr_ptr.p;
    PP.csr_idx = ix->fp.csr_idx.p;
    PP.csr_val = ix->fp.csr_val.p;
    PP.post_ptr = ht.post_ptr.p;
    PP.post = ht.post.p;
    PP.qrows = qrows_dev;
    PP.qrow0 = qrow0;
    PP.nq = (int32_t)nq;
    PP.h = ht.h;
    PP.slice0 = ix->cand_slice0;
    PP.nslices = ix->cand_slice1;
    PP.tile_base = ht.tile_base.p;
    PP.hits = ht.hits.p;
    PP.off = ht.off.p;
    PP.off_stride = off_stride;
    PP.error = ht.err.p;
    if (qt == 32)
        hipLaunchKernelGGL(ht_build_hits<32>, dim3((unsigned)ntiles), dim3(kHtPreThreads), 0, s, PP);
    else
        hipLaunchKernelGGL(ht_build_hits<16>, dim3((unsigned)ntiles), dim3(kHtPreThreads), 0, s, PP);
    LOCREC_HIP_TRY(hipGetLastError());
    return LOCREC_OK;
}

// knn_scan_ht (knn_ht.h): the rare-path parameters travel through a small device buffer
// seed_pass: the threshold-seeding launch in front of the scan proper (knn_scan_ht<.., SEED = true>): every
// stride-th slice of the range, one chunk per tile, histograms instead of lists; it leaves ht.seed for the
// scan proper, which is launched with use_seed.
int32_t launch_scan_ht(locrec_knn_index *ix, const Plan &pl, const ScanParams &P, dim3 grid, hipStream_t s,
                       bool seed_pass = false, bool use_seed = false, int32_t seed_stride = 1)
{
    locrec::HtIndex &ht = ix->ht;
    HtCold c{};
    size_t lds = pl.lds;
    int32_t slices_per_chunk = P.slices_per_chunk;
    if (seed_pass) {
        LOCREC_TRY(ht.seed.reserve((size_t)P.nq));
        c.seed_out = ht.seed.p;
        c.stride = seed_stride;
        slices_per_chunk = P.nslices - P.slice0;  // one chunk: the whole range, sampled
        grid = dim3(1, grid.y);
    } else {
        c.seed = use_seed ? ht.seed.p : nullptr;
        c.stride = 1;
    }
    c.norm_p = ix->fp.norm.p;
    c.norm_c = ix->fc.norm.p;
    c.rid = P.rid;
    c.pcsr_ptr = ix->fp.csr_ptr.p;
    c.pcsr_idx = ix->fp.csr_idx.p;
    c.pcsr_val = ix->fp.csr_val.p;
    c.ccsr_ptr = ix->fc.csr_ptr.p;
    c.ccsr_idx = ix->fc.csr_idx.p;
    c.ccsr_val = ix->fc.csr_val.p;
    c.part_s = P.part_s;
    c.part_rid = P.part_rid;
    c.part_cnt = P.part_cnt;
    c.overflow = P.overflow;
    c.qrows = P.qrows;
    c.pw = P.pw;
    c.cw = P.cw;
    c.qrow0 = P.qrow0;
    c.nq = P.nq;
    c.nrows = P.nrows;
    c.K = P.K;
    c.S = P.S;
    c.h = ht.h;
    c.c_rows = ix->fc.dim;
    c.nchunks = P.nchunks;
    c.off_tail = pl.off_tail;
    c.off_cand_s = pl.off_cand_s;
    c.off_cand_rid = pl.off_cand_rid;
    c.off_misc = pl.off_misc;
    c.off_queue = pl.off_queue;
    if (seed_pass) {  // no lists, no queues: [panels][tail accumulators][misc][histograms]
        c.off_misc = pl.off_cand_s;
        c.off_hist = pl.off_cand_s + 1024;
        lds = (size_t)c.off_hist + (size_t)pl.qt * pl.waves * 64 * sizeof(uint32_t);  // the lane maxima, [QT][W * 64]
    }
    c.flush_mask = P.flush_mask;
    c.enter_threads = P.enter_threads;
    c.fast = P.fast;
    if (const char *e = debug_env("LOCREC_DEBUG_HT")) c.dbg = std::atoi(e);
#ifdef LOCREC_DEBUG_SWITCHES
    if (debug_env("LOCREC_DEBUG_HT_CLOCKS")) {
        static unsigned long long *dbg_dev = nullptr;
        if (!dbg_dev) {
            LOCREC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dbg_dev), 16 * sizeof(unsigned long long)));
            LOCREC_HIP_TRY(hipMemset(dbg_dev, 0, 16 * sizeof(unsigned long long)));
        }
        unsigned long long prev[16];
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
        LOCREC_HIP_TRY(hipMemcpy(prev, dbg_dev, sizeof prev, hipMemcpyDeviceToHost));
        fprintf(stderr, "[locrec ht clocks of the previous launch, summed over waves] hot %llu resolve %llu sync %llu drain %llu | slices with a passing pair %llu, resolve q-iterations %llu, sync iterations %llu, sync rounds %llu\n",
                prev[0], prev[1], prev[2], prev[3], prev[4], prev[5], prev[6], prev[7]);
        fprintf(stderr, "[locrec ht work by wave index] %llu %llu %llu %llu %llu %llu %llu %llu\n", prev[8], prev[9], prev[10], prev[11],
                prev[12], prev[13], prev[14], prev[15]);
        LOCREC_HIP_TRY(hipMemset(dbg_dev, 0, 16 * sizeof(unsigned long long)));
        c.dbg_out = dbg_dev;
    }
#endif
    LOCREC_HIP_TRY(hipMemcpyAsync(ht.cold.p, &c, sizeof c, hipMemcpyHostToDevice, s));
    auto kern = seed_pass ? knn_scan_ht<16, 8, true>
                          : pl.waves == 6 ? knn_scan_ht<16, 6> : pl.waves == 12 ? knn_scan_ht<16, 12> : knn_scan_ht<16, 8>;
    if (lds > 64 * 1024)
        LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#define LOCREC_HT_ARGS                                                                                               \
    reinterpret_cast<const u32x4 *>(ht.p_sell.p), reinterpret_cast<const u32x4 *>(ht.c_sell.p),                         \
        reinterpret_cast<const HtSliceDesc *>(ht.desc.p), ht.ss.p, ht.rid.p, ht.hits.p, ht.off.p, ht.tile_base.p,       \
        (int32_t)(ix->cand_slice1 - ix->cand_slice0 + 1), P.slice0, P.nslices, slices_per_chunk,                        \
        reinterpret_cast<const HtCold *>(ht.cold.p)
    if (seed_pass)  // (the seeding pass is part of the step but not of the scan kernel's own duration)
        hipLaunchKernelGGL(kern, grid, dim3(pl.waves * 64), lds, s, LOCREC_HT_ARGS);
    else
        LOCREC_LAUNCH_PROFILED(ix->prof, kern, grid, dim3(pl.waves * 64), lds, s, LOCREC_HT_ARGS);
#undef LOCREC_HT_ARGS
    return LOCREC_OK;
}

template <int MODE, int QT, int W>
int32_t launch_scan_t(KernelProfile &prof, const ScanParams &P, dim3 grid, size_t lds, hipStream_t s)
{
    auto kern = knn_scan<MODE, QT, W>;
    if (lds > 64 * 1024)
        LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    LOCREC_LAUNCH_PROFILED(prof, kern, grid, dim3(W * 64), lds, s, P);
    return LOCREC_OK;
}

int32_t launch_scan(KernelProfile &prof, const Plan &pl, const ScanParams &P, dim3 grid, hipStream_t s)
{
#define LOCREC_CASE(M, Q, W) \
    if (pl.mode == M && pl.qt == Q && pl.waves == W) return launch_scan_t<M, Q, W>(prof, P, grid, pl.lds, s);
    LOCREC_CASE(3, 16, 8)
    LOCREC_CASE(2, 16, 16)
    LOCREC_CASE(2, 32, 8) LOCREC_CASE(2, 16, 8) LOCREC_CASE(2, 8, 8)
    LOCREC_CASE(2, 32, 4) LOCREC_CASE(2, 16, 4) LOCREC_CASE(2, 8, 4)
    LOCREC_CASE(1, 16, 4) LOCREC_CASE(1, 8, 4) LOCREC_CASE(1, 4, 4) LOCREC_CASE(1, 2, 4) LOCREC_CASE(1, 1, 4)
    LOCREC_CASE(0, 8, 4) LOCREC_CASE(0, 4, 4) LOCREC_CASE(0, 2, 4) LOCREC_CASE(0, 1, 4)
#undef LOCREC_CASE
    return fail(LOCREC_E_INVALID_ARG, "internal: no scan kernel for mode %d tile %d", pl.mode, pl.qt);
}

// Launch knn_scan1 for the person at row qrow: S1 and the histogram.  *fits = false when the
// query's panel does not fit the LDS budget (the caller then takes the tiled path).
// ---- a query that is too long for an LDS panel (rank()-with-ties top-N can emit rows of any
// length, e.g. every place visited once, RatingsBuilder.scala:42-47).  The query's two vectors are
// scattered into dense arrays in GLOBAL memory (p_dim / c_dim doubles, zero outside the query) and every
// candidate walks its plain CSR row: sum += value * dense[index], left to right in ascending index order -
// Spark's BLAS.dot order, and a product with an absent query entry adds +-0.0, i.e. nothing - so the
// dot is bit for bit the reference's in every format.  One thread per candidate; a rare path.
struct ScanDenseParams {
    const int64_t *p_ptr, *c_ptr;
    const int32_t *p_idx, *c_idx;
    const double *p_val, *c_val;
    const double *norm_p, *norm_c;
    const double *qd_p, *qd_c;
    int32_t qrow, row0, row1;
    double pw, cw;
    double *S;
    uint32_t *hist;
};

__global__ void knn_dense_query_fill(const int64_t *ptr, const int32_t *idx, const double *val, int32_t qrow, double *dense,
                                     int32_t set)
{
    const int64_t e = ptr[qrow] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < ptr[qrow + 1]) dense[idx[e]] = set ? val[e] : 0.0;
}

__global__ __launch_bounds__(256) void knn_scan_dense(const ScanDenseParams P)
{
    __shared__ uint32_t s_hist[kHistBins];
    for (int i = threadIdx.x; i < kHistBins; i += blockDim.x) s_hist[i] = 0u;
    __syncthreads();
    const int row = P.row0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (row < P.row1) {
        double dp = 0.0, dc = 0.0;
        for (int64_t e = P.p_ptr[row]; e < P.p_ptr[row + 1]; ++e) {
            const double t = P.qd_p[P.p_idx[e]] * P.p_val[e];  // x(kx) * y(ky), x = the query (Distance.scala:8)
            dp = dp + t;
        }
        for (int64_t e = P.c_ptr[row]; e < P.c_ptr[row + 1]; ++e) {
            const double t = P.qd_c[P.c_idx[e]] * P.c_val[e];
            dc = dc + t;
        }
        double sx = 0.0;
        bool have = false;
        if (row != P.qrow)  // person_id =!= personId (KnnRecommender.scala:89)
            have = exact_similarity(dp, dc, P.norm_p[row], P.norm_c[row], P.norm_p[P.qrow], P.norm_c[P.qrow], P.pw, P.cw, sx);
        if (!have) sx = 0.0;
        P.S[row] = sx;
        if (have && P.hist) atomicAdd(&s_hist[sim_bin(sx)], 1u);
    }
    __syncthreads();
    if (P.hist)
        for (int i = threadIdx.x; i < kHistBins; i += blockDim.x) {
            const uint32_t h = s_hist[i];
            if (h) atomicAdd(&P.hist[i], h);
        }
}

SideCsr side_csr_of(const locrec_knn_index *ix)
{
    SideCsr C{};
    C.p_ptr = ix->fp.csr_ptr.p; C.p_idx = ix->fp.csr_idx.p; C.p_val = ix->fp.csr_val.p;
    C.c_ptr = ix->fc.csr_ptr.p; C.c_idx = ix->fc.csr_idx.p; C.c_val = ix->fc.csr_val.p;
    C.norm_p = ix->fp.norm.p; C.norm_c = ix->fc.norm.p;
    C.inorm_p = ix->fp.inorm32.p; C.inorm_c = ix->fc.inorm32.p;
    C.side_p = ix->side_p.p; C.side_c = ix->side_c.p;
    C.side_off_p = ix->side_off_p.p; C.side_off_c = ix->side_off_c.p;
    C.side_w_p = ix->side_w_p.p; C.side_w_c = ix->side_w_c.p;
    return C;
}

int32_t enqueue_dense_query_scan(locrec_knn_index *ix, int32_t qrow, double pw, double cw)
{
    hipStream_t s = ix->stream;
    if (!ix->qd_p.p) {
        LOCREC_TRY(ix->qd_p.alloc((size_t)std::max(1, ix->fp.dim)));
        LOCREC_TRY(ix->qd_c.alloc((size_t)std::max(1, ix->fc.dim)));
        LOCREC_HIP_TRY(hipMemsetAsync(ix->qd_p.p, 0, ix->qd_p.bytes(), s));
        LOCREC_HIP_TRY(hipMemsetAsync(ix->qd_c.p, 0, ix->qd_c.bytes(), s));
    }
    const int np = std::max(1, (ix->fp.nnz[qrow] + 255) / 256), nc = std::max(1, (ix->fc.nnz[qrow] + 255) / 256);
    hipLaunchKernelGGL(knn_dense_query_fill, dim3(np), dim3(256), 0, s, ix->fp.csr_ptr.p, ix->fp.csr_idx.p, ix->fp.csr_val.p,
                       qrow, ix->qd_p.p, 1);
    hipLaunchKernelGGL(knn_dense_query_fill, dim3(nc), dim3(256), 0, s, ix->fc.csr_ptr.p, ix->fc.csr_idx.p, ix->fc.csr_val.p,
                       qrow, ix->qd_c.p, 1);
    ScanDenseParams P{};
    P.p_ptr = ix->fp.csr_ptr.p; P.p_idx = ix->fp.csr_idx.p; P.p_val = ix->fp.csr_val.p;
    P.c_ptr = ix->fc.csr_ptr.p; P.c_idx = ix->fc.csr_idx.p; P.c_val = ix->fc.csr_val.p;
    P.norm_p = ix->fp.norm.p; P.norm_c = ix->fc.norm.p;
    P.qd_p = ix->qd_p.p; P.qd_c = ix->qd_c.p;
    P.qrow = qrow;
    P.row0 = ix->cand_slice0 * 64;
    P.row1 = (int32_t)std::min<int64_t>(ix->n, (int64_t)ix->cand_slice1 * 64);
    P.pw = pw; P.cw = cw;
    P.S = ix->S1.p;
    P.hist = debug_env("LOCREC_DEBUG_NOHIST") ? nullptr : ix->hist1.p;
    const int rows = std::max(0, P.row1 - P.row0);
    if (rows > 0) LOCREC_LAUNCH_PROFILED(ix->prof, knn_scan_dense, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, P);
    // the dense arrays go back to all zero for the next long query
    hipLaunchKernelGGL(knn_dense_query_fill, dim3(np), dim3(256), 0, s, ix->fp.csr_ptr.p, ix->fp.csr_idx.p, ix->fp.csr_val.p,
                       qrow, ix->qd_p.p, 0);
    hipLaunchKernelGGL(knn_dense_query_fill, dim3(nc), dim3(256), 0, s, ix->fc.csr_ptr.p, ix->fc.csr_idx.p, ix->fc.csr_val.p,
                       qrow, ix->qd_c.p, 0);
    LOCREC_HIP_TRY(hipGetLastError());
    ix->dense_query_scans += 1;
    return LOCREC_OK;
}

int32_t enqueue_dense_impl(locrec_knn_index *ix, int32_t qrow, double pw, double cw, bool *fits)
{
    *fits = false;
    hipStream_t s = ix->stream;
    const int mode = ix->packed ? 1 : 0;
    const size_t elt = mode ? 4 : 8;
    Family fp{}, fc{};
    size_t cur = 0;
    // a query whose panel does not fit the LDS (or the 12-bit row field of the packed hash) takes the
    // dense-query scan in global memory instead: no query is too long for a request
    const bool panel_fits = plan_family(ix, ix->fp, 1, ix->fp.nnz[qrow], elt, fp, cur) &&
                            plan_family(ix, ix->fc, 1, ix->fc.nnz[qrow], elt, fc, cur) &&
                            cur <= (size_t)kLdsHardLimit - 1024 - kHistBins * 4 && !ix->force_dense_query &&
                            !ix->row_is_wide(qrow);  // (a wide row's values do not fit the packed panels: plain CSR walk)
    LOCREC_TRY(ix->S1.reserve((size_t)ix->n));
    LOCREC_TRY(ix->hist1.reserve(kHistBins));
    LOCREC_TRY(ix->sel1.reserve(8));
    if (ix->hist1_dirty) {
        // first use, or the previous scan was not followed by knn_select1 (which leaves the
        // histogram and the collect counter clean): large-K requests, failed requests
        LOCREC_HIP_TRY(hipMemsetAsync(ix->hist1.p, 0, kHistBins * sizeof(uint32_t), s));
        LOCREC_HIP_TRY(hipMemsetAsync(ix->sel1.p, 0, 8 * sizeof(int32_t), s));
    }
    ix->hist1_dirty = true;  // until knn_select1 has been enqueued behind this scan
    if (!panel_fits) {
        LOCREC_TRY(enqueue_dense_query_scan(ix, qrow, pw, cw));
        *fits = true;
        return LOCREC_OK;
    }
    // byte tables instead of the hashed panel when the data allows (knn_scan1_direct8)
    const size_t d8_bytes = (((size_t)ix->fp.dim + 15) & ~(size_t)15) + (((size_t)ix->fc.dim + 15) & ~(size_t)15);
    const bool direct8 = mode == 1 && ix->pack16 && d8_bytes <= (size_t)kDirect8MaxBytes && !ix->no_direct8;
    if (direct8) {
        fp.rows_cap = ix->fp.dim;
        fc.rows_cap = ix->fc.dim;
    }
    Scan1Params P{};
    P.ss = direct8 && ix->ht.ready && ix->ht.ss.p ? ix->ht.ss.p : nullptr;
    P.fp = fp;
    P.fc = fc;
    P.qrow = qrow;
    P.nrows = (int32_t)ix->n;
    P.slice0 = ix->cand_slice0;
    P.nslices = ix->cand_slice1;
    P.pw = pw;
    P.cw = cw;
    P.S = ix->S1.p;
    P.hist = debug_env("LOCREC_DEBUG_NOHIST") ? nullptr : ix->hist1.p;
    int blocks = std::max(1, std::min(256, (ix->cand_slice1 - ix->cand_slice0 + kScan1Waves - 1) / kScan1Waves));
    if (const char *e = debug_env("LOCREC_DEBUG_SCAN1_BLOCKS")) blocks = std::max(1, std::atoi(e));
    if (direct8) {
        if (!ix->direct8_attr) {
            LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_scan1_direct8),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, kDirect8MaxBytes));
            ix->direct8_attr = true;
        }
        LOCREC_LAUNCH_PROFILED(ix->prof, knn_scan1_direct8, dim3(blocks), dim3(kScan1Waves * 64), d8_bytes, s, P);
    } else if (mode) {
        if (cur > 64 * 1024)
            LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_scan1<1>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)cur));
        LOCREC_LAUNCH_PROFILED(ix->prof, knn_scan1<1>, dim3(blocks), dim3(kScan1Waves * 64), cur, s, P);
    } else {
        if (cur > 64 * 1024)
            LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_scan1<0>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)cur));
        LOCREC_LAUNCH_PROFILED(ix->prof, knn_scan1<0>, dim3(blocks), dim3(kScan1Waves * 64), cur, s, P);
    }
    if (!ix->wide_rows.empty()) {
        // the packed images hold the index's wide rows as padding: their similarities come from the plain CSR
        const int32_t nw = (int32_t)ix->wide_rows.size();
        hipLaunchKernelGGL(knn_side_scan1, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, side_csr_of(ix), ix->wide_rows_dev.p,
                           nw, qrow, ix->cand_slice0 * 64, (int32_t)std::min<int64_t>(ix->n, (int64_t)ix->cand_slice1 * 64), pw, cw,
                           ix->S1.p, P.hist);
    }
    LOCREC_HIP_TRY(hipGetLastError());
    *fits = true;
    return LOCREC_OK;
}

unsigned char *stage_of(locrec_knn_index *ix);

// One request as a stream: scan -> histogram select -> collect -> sort (see knn_scan1).
// Returns LOCREC_OK with *used = false when the request must take the tiled path instead.
int32_t enqueue_single(locrec_knn_index *ix, int32_t qrow, double pw, double cw, int64_t k, bool *used)
{
    *used = false;
    hipStream_t s = ix->stream;
    const int K = (int)k;
    bool fits = false;
    LOCREC_TRY(enqueue_dense_impl(ix, qrow, pw, cw, &fits));
    if (!fits) return LOCREC_OK;
    LOCREC_TRY(ix->list1_s.reserve(kCollectCap));
    LOCREC_TRY(ix->list1_r.reserve(kCollectCap));
    LOCREC_TRY(ix->out_ids.reserve((size_t)K));
    LOCREC_TRY(ix->out_sims.reserve((size_t)K));
    LOCREC_TRY(ix->out_rows.reserve((size_t)K));
    LOCREC_TRY(ix->out_cnt.reserve(1));
    const int32_t row0 = ix->cand_slice0 * 64;
    const int32_t row1 = (int32_t)std::min<int64_t>(ix->n, (int64_t)ix->cand_slice1 * 64);
    const bool nohist = debug_env("LOCREC_DEBUG_NOHIST") != nullptr;
    // the result also lands in the pinned staging buffer, in fetch_topk's layout, when it fits (knn_final1)
    unsigned char *host = nullptr;
    if (16 + (size_t)K * 16 + 8 <= locrec_knn_index::kStageBytes && stage_of(ix) && ix->h_stage_dev && !ix->no_pack)
        host = ix->h_stage_dev;
    hipLaunchKernelGGL(knn_select1, dim3(1), dim3(1024), 0, s, ix->hist1.p, K, ix->sel1.p);
    hipLaunchKernelGGL(knn_collect1, dim3((unsigned)std::max(1, (row1 - row0 + 255) / 256)), dim3(256), 0, s, ix->S1.p,
                       ix->rid.p, row0, row1, ix->sel1.p, ix->list1_s.p, ix->list1_r.p, ix->sel1.p + 3, (int64_t)0);
    ix->hist1_dirty = nohist;  // knn_select1 cleans up behind itself
    const size_t flds = (size_t)kCollectCap * 12;
    if (!ix->final1_attr) {
        LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_final1),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
        ix->final1_attr = true;
    }
    hipLaunchKernelGGL(knn_final1, dim3(1), dim3(256), flds, s, ix->list1_s.p, ix->list1_r.p, ix->sel1.p + 3, K,
                       ix->ids_by_rank.p, ix->row_of_rid.p, ix->out_ids.p, ix->out_sims.p, ix->out_rows.p,
                       ix->out_cnt.p, ix->sel1.p + 4, host, static_cast<const int64_t *>(nullptr));
    LOCREC_HIP_TRY(hipGetLastError());
    ix->single_direct = host != nullptr;
    // the (rare) overflow of the collect list is checked when the result is read back
    // (resolve_single_overflow); knn_final1 reports zero neighbours in that case
    ix->single_pending = true;
    ix->single_qrow = qrow;
    ix->single_pw = pw;
    ix->single_cw = cw;
    ix->last_nq = 1;
    ix->last_k = k;
    ix->have_result = true;
    *used = true;
    return LOCREC_OK;
}

// Enqueue scan + merge for nq queries given as device rows (qrows_dev) or a row range.
// Candidate slices from which ONE request takes the stream path (scan -> select -> collect -> sort).  It used to be
// 64; a region-sized index (the reference builds one recommender per region: ~3 k persons = ~50 slices) then fell
// to the tiled path: 0.092 ms per query at 2,000 persons against ~0.05 ms on the stream path.
constexpr int kSingleMinSlices = 4;

int32_t enqueue_topk(locrec_knn_index *ix, const int32_t *qrows_dev, int32_t qrow0, int64_t nq,
                     int max_nnz_p, int max_nnz_c, double pw, double cw, int64_t k, bool mark_absent = false)
{
    hipStream_t s = ix->stream;
    const int K = (int)k;
    ix->single_pending = false;
    ix->single_direct = false;
    ix->last_scan_fast = false;
    ix->have_agg = false;  // the neighbour lists a resident batched aggregation was built from are overwritten
    ix->have_lkb = false;
    const int range_slices = ix->cand_slice1 - ix->cand_slice0;
    if (nq == 1 && !ix->no_single && range_slices >= kSingleMinSlices && !mark_absent) {
        int32_t qrow = qrow0;
        if (qrows_dev) LOCREC_HIP_TRY(hipMemcpy(&qrow, qrows_dev, sizeof(int32_t), hipMemcpyDeviceToHost));
        bool used = false;
        LOCREC_TRY(enqueue_single(ix, qrow, pw, cw, k, &used));
        if (used) return LOCREC_OK;
    }
    Plan pl;
    // head / tail form (knn_ht.h) when the index has it and this batch fits its pre-pass: every tile's
    // (query, tail place) pairs fit kHtMaxEntries and its hits a 32-bit offset
    bool use_ht = false;
    int64_t ht_total = 0;
    if (ix->ht.ready && !ix->no_ht && (!qrows_dev || (int64_t)ix->qrows_host.size() == nq)) {
        const int qt = ix->ht.qt;
        bool ok = true;
        for (int64_t t0 = 0; ok && t0 < nq; t0 += qt) {
            int64_t ent = 0, hits = 0;
            for (int64_t i = t0; i < std::min(nq, t0 + qt); ++i) {
                const int32_t r = qrows_dev ? ix->qrows_host[(size_t)i] : qrow0 + (int32_t)i;
                ent += ix->ht.tail_nnz[(size_t)r];
                hits += ix->ht.tail_hits_ps[(size_t)r + 1] - ix->ht.tail_hits_ps[(size_t)r];
            }
            ok = ent <= kHtMaxEntries && hits < ((int64_t)1 << 32);
            ht_total += hits;
        }
        use_ht = ok && ix->ht.desc.p && make_plan_ht(ix, K, pl);
    }
    // a WIDE row as the query (per-row format fallback): its values do not fit the packed panels, so it is served like a
    // query that is too long for any tile - the dense CSR scan + sort into its slot - while the rest runs tiled
    bool wide_query = false;
    if (!ix->wide_rows.empty()) {
        if (qrows_dev && (int64_t)ix->qrows_host.size() != nq) {
            ix->qrows_host.resize((size_t)nq);
            LOCREC_HIP_TRY(hipMemcpy(ix->qrows_host.data(), qrows_dev, (size_t)nq * sizeof(int32_t), hipMemcpyDeviceToHost));
        }
        for (int64_t i = 0; i < nq && !wide_query; ++i)
            wide_query = ix->row_is_wide(qrows_dev ? ix->qrows_host[(size_t)i] : qrow0 + (int32_t)i);
    }
    if (wide_query || (!use_ht && !make_plan(ix, nq, max_nnz_p, max_nnz_c, K, pl))) {
        // Some query of the batch is too long for an LDS tile even on its own (rank()-with-ties can emit rows of
        // any length).  Those queries are served by the dense-query scan + sort path (knn_scan_dense,
        // knn_large.hip) into their slots of the result arrays; the rest of the batch runs tiled with a
        // stand-in row in their places.
        std::vector<int32_t> rows((size_t)nq);
        if (qrows_dev) {
            if ((int64_t)ix->qrows_host.size() == nq) rows = ix->qrows_host;
            else LOCREC_HIP_TRY(hipMemcpy(rows.data(), qrows_dev, (size_t)nq * sizeof(int32_t), hipMemcpyDeviceToHost));
        } else {
            for (int64_t i = 0; i < nq; ++i) rows[(size_t)i] = qrow0 + (int32_t)i;
        }
        std::vector<int64_t> longq;
        int32_t stand_in = -1;
        int mp = 1, mc = 1;
        for (int64_t i = 0; i < nq; ++i) {
            const int32_t r = rows[(size_t)i];
            Plan one;
            if (ix->row_is_wide(r) || !make_plan(ix, 1, ix->fp.nnz[r], ix->fc.nnz[r], K, one)) {
                longq.push_back(i);
            } else {
                if (stand_in < 0) stand_in = r;
                mp = std::max(mp, ix->fp.nnz[r]);
                mc = std::max(mc, ix->fc.nnz[r]);
            }
        }
        if (longq.empty())
            return fail(LOCREC_E_INVALID_ARG, "query tile does not fit in LDS (k=%d, nnz=%d/%d)", K, max_nnz_p, max_nnz_c);
        if (stand_in >= 0 && nq > 1) {
            for (int64_t i : longq) rows[(size_t)i] = stand_in;
            LOCREC_TRY(ix->qrows_patch.reserve((size_t)nq));
            LOCREC_HIP_TRY(hipMemcpyAsync(ix->qrows_patch.p, rows.data(), (size_t)nq * sizeof(int32_t), hipMemcpyHostToDevice, s));
            LOCREC_HIP_TRY(hipStreamSynchronize(s));  // (rows is a local)
            std::vector<int32_t> saved;
            saved.swap(ix->qrows_host);
            ix->qrows_host = rows;
            const int32_t st = enqueue_topk(ix, ix->qrows_patch.p, 0, nq, mp, mc, pw, cw, k, mark_absent);
            ix->qrows_host.swap(saved);
            LOCREC_TRY(st);
        } else {
            LOCREC_TRY(ix->out_ids.reserve((size_t)nq * K));
            LOCREC_TRY(ix->out_sims.reserve((size_t)nq * K));
            LOCREC_TRY(ix->out_rows.reserve((size_t)nq * K));
            LOCREC_TRY(ix->out_cnt.reserve((size_t)nq));
            ix->last_scan_fast = false;
        }
        const std::vector<int32_t> *orig = qrows_dev && (int64_t)ix->qrows_host.size() == nq ? &ix->qrows_host : nullptr;
        // The special queries, 16 at a time: one pass of every candidate's plain CSR row against dense tables of the
        // tile's queries (knn_large.hip: exact for every row in every format) gives S[row][16]; each column then takes
        // the single request's selection (histogram -> deciding bin -> collect -> sort) straight into its slot.  One
        // by one through the dense scan + full radix sort this was ~1 ms per query - 16 wide queries doubled a cfg2 step.
        const bool whole_range = ix->cand_slice0 == 0 && ix->cand_slice1 == ix->nslices;
        std::vector<int32_t> srow(longq.size());
        for (size_t j = 0; j < longq.size(); ++j) {
            const int64_t i = longq[j];
            int32_t r = orig ? (*orig)[(size_t)i] : qrow0 + (int32_t)i;
            if (qrows_dev && !orig) LOCREC_HIP_TRY(hipMemcpy(&r, qrows_dev + i, sizeof(int32_t), hipMemcpyDeviceToHost));
            srow[j] = r;
        }
        if (whole_range && K <= kCollectCap / 2 && !ix->no_tile_special) {
            // per-column workspaces of a tile; knn_select1 leaves histograms and counters clean behind itself, the first
            // use cleans them here
            if (!ix->tile_hist.p) {
                LOCREC_TRY(ix->tile_hist.alloc((size_t)16 * kHistBins));
                LOCREC_TRY(ix->tile_sel.alloc(16 * 8));
                LOCREC_TRY(ix->tile_list_s.alloc((size_t)16 * kCollectCap));
                LOCREC_TRY(ix->tile_list_r.alloc((size_t)16 * kCollectCap));
                LOCREC_TRY(ix->tile_slots.alloc(16));
                LOCREC_TRY(ix->tile_ovf.alloc(16));
                LOCREC_HIP_TRY(hipMemsetAsync(ix->tile_hist.p, 0, ix->tile_hist.bytes(), s));
                LOCREC_HIP_TRY(hipMemsetAsync(ix->tile_sel.p, 0, ix->tile_sel.bytes(), s));
            }
            const size_t flds = (size_t)kCollectCap * 12;
            if (!ix->final1_attr) {
                LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_final1),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
                ix->final1_attr = true;
            }
            const int32_t nrows = (int32_t)ix->n;
            for (size_t j0 = 0; j0 < longq.size(); j0 += 16) {
                const int nt = (int)std::min<size_t>(16, longq.size() - j0);
                LOCREC_TRY(knn_large_scan_tile(ix, srow.data() + j0, nt, pw, cw));
                // all columns of the tile at once: histograms, deciding bins, collect lists, sorted lists into the slots
                // (four launches per tile; one set per column was 64 launches and 0.6 ms of a cfg2 step with 16 wide queries)
                int64_t slots[16] = {0};
                for (int t = 0; t < nt; ++t) slots[t] = longq[j0 + (size_t)t];
                LOCREC_HIP_TRY(hipMemcpyAsync(ix->tile_slots.p, slots, (size_t)nt * sizeof(int64_t), hipMemcpyHostToDevice, s));
                const double *cols = nullptr;
                LOCREC_TRY(knn_large_tile_hists(ix, nt, ix->tile_hist.p, &cols));
                hipLaunchKernelGGL(knn_select1, dim3((unsigned)nt), dim3(1024), 0, s, ix->tile_hist.p, K, ix->tile_sel.p);
                hipLaunchKernelGGL(knn_collect1, dim3((unsigned)std::max(1, (nrows + 255) / 256), (unsigned)nt), dim3(256), 0, s, cols,
                                   ix->rid.p, 0, nrows, ix->tile_sel.p, ix->tile_list_s.p, ix->tile_list_r.p, ix->tile_sel.p + 3,
                                   (int64_t)nrows);
                hipLaunchKernelGGL(knn_final1, dim3((unsigned)nt), dim3(256), flds, s, ix->tile_list_s.p, ix->tile_list_r.p,
                                   ix->tile_sel.p + 3, K, ix->ids_by_rank.p, ix->row_of_rid.p, ix->out_ids.p, ix->out_sims.p,
                                   ix->out_rows.p, ix->out_cnt.p, ix->tile_ovf.p, static_cast<unsigned char *>(nullptr),
                                   ix->tile_slots.p);
                LOCREC_HIP_TRY(hipGetLastError());
                // a deciding bin with more entries than the collect list holds (a pathological tie mass): that slot
                // takes the full sort instead
                int32_t ovf[16] = {0};
                LOCREC_HIP_TRY(hipMemcpyAsync(ovf, ix->tile_ovf.p, (size_t)nt * sizeof(int32_t), hipMemcpyDeviceToHost, s));
                LOCREC_HIP_TRY(hipStreamSynchronize(s));
                for (int t = 0; t < nt; ++t)
                    if (ovf[t]) LOCREC_TRY(knn_large_topk_device(ix, srow[j0 + (size_t)t], pw, cw, k, longq[j0 + (size_t)t]));
            }
        } else {
            for (size_t j = 0; j < longq.size(); ++j) LOCREC_TRY(knn_large_topk_device(ix, srow[j], pw, cw, k, longq[j]));
        }
        ix->single_pending = false;
        ix->last_nq = nq;
        ix->last_k = k;
        ix->have_result = true;
        return LOCREC_OK;
    }
    const int ntiles = (int)((nq + pl.qt - 1) / pl.qt);
    // enough blocks to fill the chip; when that needs more chunks than one merge block can sort
    // (a single request), the merge runs in two levels
    const int max_chunks = std::max(1, kMergeCap / K);
    // target number of blocks.  The row scan was tuned at 2048 (4 rounds of 512 resident blocks).  The head /
    // tail scan does better with ONE chunk per tile at the bench batch (1024 blocks = exactly two rounds):
    // a chunk more per tile means another K*ln(N/K) list insertions and another list to merge
    // (cfg2: 23.2 -> 19.0 ms per 16,384-query batch).
    int want = std::max(1, ((use_ht ? 1024 : 2048) + ntiles - 1) / ntiles);
    if (ix->env_blocks > 0) want = std::max(1, (ix->env_blocks + ntiles - 1) / ntiles);  // tuning
    int nchunks = std::min(std::min(want, max_chunks * max_chunks), std::max(1, range_slices / 8));
    int spc = (range_slices + nchunks - 1) / nchunks;
    spc = std::max(pl.waves, (spc + pl.waves - 1) / pl.waves * pl.waves);
    nchunks = std::max(1, (range_slices + spc - 1) / spc);
    const int ngroups = nchunks > max_chunks ? (nchunks + max_chunks - 1) / max_chunks : 0;

    LOCREC_TRY(ix->part_s.reserve((size_t)nq * nchunks * K));
    LOCREC_TRY(ix->part_rid.reserve((size_t)nq * nchunks * K));
    LOCREC_TRY(ix->part_cnt.reserve((size_t)nq * nchunks));
    if (ngroups) {
        LOCREC_TRY(ix->part2_s.reserve((size_t)nq * ngroups * K));
        LOCREC_TRY(ix->part2_rid.reserve((size_t)nq * ngroups * K));
        LOCREC_TRY(ix->part2_cnt.reserve((size_t)nq * ngroups));
    }
    LOCREC_TRY(ix->out_ids.reserve((size_t)nq * K));
    LOCREC_TRY(ix->out_sims.reserve((size_t)nq * K));
    LOCREC_TRY(ix->out_rows.reserve((size_t)nq * K));
    LOCREC_TRY(ix->out_cnt.reserve((size_t)nq));

    ScanParams P{};
    P.fp = pl.fp;
    P.fc = pl.fc;
    P.rid = ix->rid.p;
    P.qrows = qrows_dev;
    P.qrow0 = qrow0;
    P.nq = (int32_t)nq;
    P.nrows = (int32_t)ix->n;
    P.slice0 = ix->cand_slice0;
    P.nslices = ix->cand_slice1;
    P.slices_per_chunk = spc;
    P.nchunks = nchunks;
    P.K = K;
    P.S = pl.S;
    P.pw = pw;
    P.cw = cw;
    P.part_s = ix->part_s.p;
    P.part_rid = ix->part_rid.p;
    P.part_cnt = ix->part_cnt.p;
    P.off_cand_s = pl.off_cand_s;
    P.off_cand_rid = pl.off_cand_rid;
    P.off_misc = pl.off_misc;
    P.off_queue = pl.off_queue;
    if (!ix->scan_overflow.p) {
        LOCREC_TRY(ix->scan_overflow.alloc(2));  // [0] this launch, [1] cumulative
        LOCREC_HIP_TRY(hipMemsetAsync(ix->scan_overflow.p, 0, 2 * sizeof(int32_t), s));
    }
    LOCREC_HIP_TRY(hipMemsetAsync(ix->scan_overflow.p, 0, sizeof(int32_t), s));
    P.overflow = ix->scan_overflow.p;
    P.fast = (pl.mode != 0 && !ix->no_fast) ? 1 : 0;
    P.flush_mask = kFlushEvery - 1;
    P.enter_threads = kEnterFastThreads;
    if (ix->env_flush > 0) P.flush_mask = ix->env_flush - 1;      // tuning
    if (ix->env_enter >= 0) P.enter_threads = ix->env_enter;
    ix->last_scan_fast = P.fast != 0;
    P.lds_bytes = (int32_t)pl.lds;
    P.poison = (debug_env("LOCREC_DEBUG_POISON") ? 1 : 0) | (debug_env("LOCREC_DEBUG_NOFILTER") ? 2 : 0) |
               (debug_env("LOCREC_DEBUG_NOEPILOGUE") ? 4 : 0);
    if (P.poison) {
        (void)hipMemsetAsync(ix->part_s.p, 0xA5, ix->part_s.bytes(), s);
        (void)hipMemsetAsync(ix->part_rid.p, 0xA5, ix->part_rid.bytes(), s);
        (void)hipMemsetAsync(ix->part_cnt.p, 0xA5, ix->part_cnt.bytes(), s);
    }

    if (use_ht) {
        LOCREC_TRY(enqueue_ht_prepass(ix, qrows_dev, qrow0, nq, pl.qt, ht_total));
        P.ht.hits = ix->ht.hits.p;
        P.ht.off = ix->ht.off.p;
        P.ht.tile_base = ix->ht.tile_base.p;
        P.ht.off_stride = ix->cand_slice1 - ix->cand_slice0 + 1;
        P.ht.off_tail = pl.off_tail;
        P.ht.h = ix->ht.h;
        P.ht.c_rows = ix->fc.dim;
    }
    const bool dedicated = use_ht && !ix->ht.v1 && pl.qt == 16 && ix->ht.h <= cfg::kHtHead;  // (its plane stride is a constant)
    // threshold seeding (knn_ht.h, SEED): worth a launch of its own when the scan is long - ~256 sampled slices
    // per tile cost 1 - 2 % of a cfg2 scan and remove its cold start
    const bool seeded = dedicated && pl.waves == 8 && range_slices >= ix->seed_min_slices && !ix->no_seed;
    if (seeded)
        LOCREC_TRY(launch_scan_ht(ix, pl, P, dim3(1u, (unsigned)ntiles), s, true, false,
                                  std::max(1, range_slices / ix->seed_sample_slices)));
    if (dedicated)
        LOCREC_TRY(launch_scan_ht(ix, pl, P, dim3((unsigned)nchunks, (unsigned)ntiles), s, false, seeded));
    else
        LOCREC_TRY(launch_scan(ix->prof, pl, P, dim3((unsigned)nchunks, (unsigned)ntiles), s));
    ix->last_plan_kernel = !use_ht ? 1 : dedicated ? 2 : 3;
    ix->last_plan_mode = pl.mode;
    ix->last_plan_qt = pl.qt;
    ix->last_plan_waves = pl.waves;
    const double *fs = ix->part_s.p;
    const uint32_t *fr = ix->part_rid.p;
    const int32_t *fc = ix->part_cnt.p;
    int flists = nchunks;
    if (ngroups) {
        const int M1 = pow2ceil(std::max(2, max_chunks * K));
        const size_t l1 = (size_t)M1 * 12;
        if (l1 > 64 * 1024)
            LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_merge_partial),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1));
        hipLaunchKernelGGL(knn_merge_partial, dim3((unsigned)ngroups, (unsigned)nq), dim3(256), l1, s, ix->part_s.p,
                           ix->part_rid.p, ix->part_cnt.p, nchunks, K, max_chunks, M1, ix->part2_s.p,
                           ix->part2_rid.p, ix->part2_cnt.p, ngroups);
        fs = ix->part2_s.p;
        fr = ix->part2_rid.p;
        fc = ix->part2_cnt.p;
        flists = ngroups;
    }
    const int M = pow2ceil(std::max(2, flists * K));
    const size_t mlds = (size_t)M * 12;
    if (mlds > 64 * 1024)
        LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_merge),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds));
    hipLaunchKernelGGL(knn_merge, dim3((unsigned)nq), dim3(256), mlds, s, fs, fr, fc, flists, K, M,
                       ix->ids_by_rank.p, ix->row_of_rid.p, ix->out_ids.p, ix->out_sims.p, ix->out_rows.p,
                       ix->out_cnt.p);
    if (!ix->wide_rows.empty()) {
        // the index's wide rows as candidates (they are padding in the packed images): merged into every K-list
        const int32_t nw = (int32_t)ix->wide_rows.size();
        int cap = 2;
        while (cap < K + nw) cap <<= 1;
        // LDS: the merged list, the query's category table and its place hash (2 x the longest query of the batch, at most
        // 8192 slots; a longer query merges rows instead)
        int hash_cap = 64;
        while (hash_cap < 2 * max_nnz_p && hash_cap < 8192) hash_cap <<= 1;
        const int c_dim = std::max(1, ix->fc.dim);
        const size_t slds = (size_t)cap * 12 + (size_t)c_dim * 8 + (size_t)hash_cap * 12;
        if (slds > 64 * 1024)
            LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_side_topk),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds));
        hipLaunchKernelGGL(knn_side_topk, dim3((unsigned)nq), dim3(256), slds, s, side_csr_of(ix), ix->wide_rows_dev.p, nw, qrows_dev,
                           qrow0, ix->cand_slice0 * 64, (int32_t)std::min<int64_t>(ix->n, (int64_t)ix->cand_slice1 * 64), pw, cw, K,
                           ix->rid.p, ix->ids_by_rank.p, ix->row_of_rid.p, ix->out_ids.p, ix->out_sims.p, ix->out_rows.p,
                           ix->out_cnt.p, c_dim, hash_cap);
    }
    if (mark_absent)
        hipLaunchKernelGGL(knn_mark_absent, dim3((unsigned)nq), dim3(64), 0, s, ix->fp.norm.p, ix->fc.norm.p, qrows_dev,
                           qrow0, (int32_t)nq, K, ix->out_ids.p, ix->out_sims.p, ix->out_rows.p, ix->out_cnt.p);
    LOCREC_HIP_TRY(hipGetLastError());
    ix->last_nq = nq;
    ix->last_k = k;
    ix->have_result = true;
    ix->last_tiled = {qrows_dev, qrow0, nq, max_nnz_p, max_nnz_c, pw, cw, k, mark_absent};
    return LOCREC_OK;
}

// The fast insertion path of the last tiled scan dropped a survivor (a wave queue overflowed):
// redo the scan with synchronous insertion.
int32_t rerun_tiled_sync(locrec_knn_index *ix)
{
    const bool saved = ix->no_fast, saved_single = ix->no_single;
    ix->no_fast = true;
    ix->no_single = true;
    const auto r = ix->last_tiled;
    const int32_t st = enqueue_topk(ix, r.qrows_dev, r.qrow0, r.nq, r.max_p, r.max_c, r.pw, r.cw, r.k, r.mark_absent);
    ix->no_fast = saved;
    ix->no_single = saved_single;
    return st;
}

// A single request ran as a stream and its collect list overflowed (flag read by the caller):
// run it again on the tiled path.
int32_t rerun_single_tiled(locrec_knn_index *ix)
{
    const bool saved = ix->no_single;
    ix->no_single = true;
    const int32_t row = ix->single_qrow;
    const int32_t st = enqueue_topk(ix, nullptr, row, 1, ix->fp.nnz[row], ix->fc.nnz[row], ix->single_pw, ix->single_cw,
                                    ix->last_k);
    ix->no_single = saved;
    return st;
}

// "No such person" (KnnRecommender.scala:83): unknown id, or absent from a family.
int32_t find_query_row(const locrec_knn_index *ix, int64_t person_id, int32_t *row)
{
    const int32_t r = ix->row_of_person(person_id);
    if (r < 0 || ix->fp.nnz[(size_t)r] == 0 || ix->fc.nnz[(size_t)r] == 0)
        return fail(LOCREC_E_NOT_FOUND, "No such person: %lld", (long long)person_id);
    *row = r;
    return LOCREC_OK;
}

// pinned staging buffer (lazy); nullptr if the allocation fails (callers then copy directly)
unsigned char *stage_of(locrec_knn_index *ix)
{
    if (!ix->h_stage) {
        void *p = nullptr;
        if (hipHostMalloc(&p, locrec_knn_index::kStageBytes, hipHostMallocDefault) == hipSuccess) {
            ix->h_stage = static_cast<unsigned char *>(p);
            void *d = nullptr;
            if (hipHostGetDevicePointer(&d, p, 0) == hipSuccess) ix->h_stage_dev = static_cast<unsigned char *>(d);
            else (void)hipGetLastError();
        } else {
            (void)hipGetLastError();
        }
    }
    return ix->h_stage;
}

// The read-back of a small result - a request's K neighbours, its recommendation rows, a few flags - used to be
// one device-to-host copy per array: five or six copy commands of a few hundred bytes, each a packet of its own
// on the stream (~5 us apiece, a third of a 0.066 ms request).  knn_pack_host gathers the arrays straight into
// the pinned staging buffer instead (the device writes host memory over PCIe): one small launch, then the
// request's single synchronisation.  A segment may be cut to a count the device knows (the recommendation rows).
struct PackSeg {
    const void *src;
    const int64_t *count;  // NULL, or: copy only the first *count elements of `elem` bytes
    uint32_t dst_off, bytes, elem, pad;
};
struct PackArgs {
    unsigned char *dst;
    PackSeg seg[6];
    int32_t nseg, pad;
};

__global__ __launch_bounds__(256) void knn_pack_host(PackArgs a)
{
    for (int sgi = 0; sgi < a.nseg; ++sgi) {
        const PackSeg &g = a.seg[sgi];
        uint32_t bytes = g.bytes;
        if (g.count) {
            const int64_t c = *g.count;
            const uint64_t want = c > 0 ? (uint64_t)c * g.elem : 0;
            bytes = (uint32_t)(want < bytes ? want : bytes);
        }
        const uint32_t words = bytes >> 2;  // (every array here is a whole number of 4-byte words)
        const uint32_t *src = static_cast<const uint32_t *>(g.src);
        uint32_t *dst = reinterpret_cast<uint32_t *>(a.dst + g.dst_off);
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < words; i += gridDim.x * 256) dst[i] = src[i];
    }
}

struct PackList {
    PackArgs a{};
    void add(const void *src, size_t dst_off, size_t bytes, const int64_t *count = nullptr, uint32_t elem = 0)
    {
        if (a.nseg < 6) a.seg[a.nseg++] = PackSeg{src, count, (uint32_t)dst_off, (uint32_t)bytes, elem, 0};
    }
    void launch(locrec_knn_index *ix, hipStream_t s)
    {
        if (a.nseg == 0) return;
        a.dst = ix->h_stage_dev;
        size_t most = 0;
        for (int i = 0; i < a.nseg; ++i) most = std::max<size_t>(most, a.seg[i].bytes);
        const unsigned blocks = (unsigned)std::min<size_t>(16, std::max<size_t>(1, most / 4096));
        hipLaunchKernelGGL(knn_pack_host, dim3(blocks), dim3(256), 0, s, a);
    }
};

// Candidate shard `index` of `count`: a contiguous range of slices holding about 1/count of the
// stored elements (rows are sorted by length, so equal slice counts would not be equal work).
void shard_slice_range(locrec_knn_index *ix, int32_t index, int32_t count, int32_t *s0, int32_t *s1)
{
    const int32_t ns = ix->nslices;
    if (ix->slice_cost.empty()) {
        ix->slice_cost.assign((size_t)ns + 1, 0);
        for (int32_t sl = 0; sl < ns; ++sl) {
            int wp = 0, wc = 0;
            for (int64_t r = (int64_t)sl * 64; r < std::min<int64_t>(ix->n, (int64_t)(sl + 1) * 64); ++r) {
                wp = std::max(wp, ix->fp.nnz[r]);
                wc = std::max(wc, ix->fc.nnz[r]);
            }
            ix->slice_cost[sl + 1] = ix->slice_cost[sl] + 64 * (int64_t)(((wp + 3) & ~3) + ((wc + 3) & ~3)) + 64;
        }
    }
    auto cut = [&](int32_t i) -> int32_t {
        if (i <= 0) return 0;
        if (i >= count) return ns;
        const int64_t want = ix->slice_cost[ns] / count * i;
        return (int32_t)(std::lower_bound(ix->slice_cost.begin(), ix->slice_cost.end(), want) - ix->slice_cost.begin());
    };
    *s0 = std::min(cut(index), ns);
    *s1 = std::min(std::max(cut(index + 1), *s0), ns);
}

struct CandRangeGuard {
    locrec_knn_index *ix;
    CandRangeGuard(locrec_knn_index *i, int32_t s0, int32_t s1) : ix(i)
    {
        ix->cand_slice0 = s0;
        ix->cand_slice1 = s1;
    }
    ~CandRangeGuard()
    {
        ix->cand_slice0 = 0;
        ix->cand_slice1 = ix->nslices;
    }
};

}  // namespace

using namespace locrec;

namespace locrec {
int32_t knn_enqueue_dense(locrec_knn_index *ix, int32_t qrow, double pw, double cw)
{
    bool fits = false;
    LOCREC_TRY(enqueue_dense_impl(ix, qrow, pw, cw, &fits));
    if (!fits) return fail(LOCREC_E_INVALID_ARG, "query vector too long for the single-request scan");
    return LOCREC_OK;
}

unsigned char *knn_stage(locrec_knn_index *ix) { return stage_of(ix); }
}  // namespace locrec

namespace locrec {
void knn_read_env(locrec_knn_index *ix)
{
    ix->force_hash = std::getenv("LOCREC_KNN_FORCE_HASH") != nullptr;
    if (const char *e = std::getenv("LOCREC_KNN_QT")) ix->qt_max = std::max(1, std::atoi(e));
    ix->no_single = std::getenv("LOCREC_KNN_NO_SINGLE") != nullptr;
    ix->no_fast = std::getenv("LOCREC_KNN_NO_FAST") != nullptr;
    ix->no_dense_hash = std::getenv("LOCREC_KNN_NO_DENSE_HASH") != nullptr;
    ix->no_wide_block = std::getenv("LOCREC_KNN_NO_WIDE_BLOCK") != nullptr;
    if (const char *e = std::getenv("LOCREC_KNN_WAVES")) ix->waves16 = std::atoi(e) == 4 ? 4 : 8;
    ix->no_ht = std::getenv("LOCREC_KNN_NO_HT") != nullptr;
    ix->no_direct8 = std::getenv("LOCREC_KNN_NO_DIRECT8") != nullptr;  // A/B: a single request through the hashed panel (knn_scan1<1>)
    ix->no_pack = std::getenv("LOCREC_KNN_NO_PACK") != nullptr;
    ix->no_seed = std::getenv("LOCREC_KNN_NO_SEED") != nullptr;  // A/B: knn_scan_ht without the threshold-seeding pass
    ix->no_tile_special = std::getenv("LOCREC_KNN_NO_TILE_SPECIAL") != nullptr;  // A/B: special queries of a batch one by one
    if (const char *e = std::getenv("LOCREC_KNN_SEED_MIN_SLICES")) ix->seed_min_slices = std::max(1, std::atoi(e));  // tests
    if (const char *e = std::getenv("LOCREC_KNN_SEED_SAMPLE")) ix->seed_sample_slices = std::max(8, std::atoi(e));     // tuning
    ix->force_dense_query = std::getenv("LOCREC_KNN_FORCE_DENSE_QUERY") != nullptr;  // tests: every single request takes knn_scan_dense
    ix->ht.v1 = std::getenv("LOCREC_KNN_HT_V1") != nullptr;
    if (const char *e = std::getenv("LOCREC_KNN_HT_W")) {  // tuning: waves per block of knn_scan_ht
        const int w = std::atoi(e);
        if (w == 6 || w == 8 || w == 12) ix->ht.waves = w;
    }
    if (ix->ht.v1) ix->ht.waves = 8;
    ix->force_generic = std::getenv("LOCREC_KNN_FORCE_GENERIC") != nullptr;
    ix->no_pack16 = std::getenv("LOCREC_KNN_NO_PACK16") != nullptr;
    ix->no_pop = std::getenv("LOCREC_KNN_NO_POP") != nullptr;
    ix->no_row_fallback = std::getenv("LOCREC_KNN_NO_ROW_FALLBACK") != nullptr;  // wide rows demote the whole index again
    if (const char *e = std::getenv("LOCREC_KNN_HT_H")) ix->env_ht_h = std::max(4, std::atoi(e));     // tuning: head width
    if (const char *e = std::getenv("LOCREC_KNN_POP_H")) ix->env_pop_h = std::max(64, std::atoi(e));  // tuning: direct slot table
    if (const char *e = std::getenv("LOCREC_KNN_BLOCKS")) ix->env_blocks = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("LOCREC_KNN_FLUSH")) {  // 1, 2, 4, 8, 16, ...
        const int v = std::atoi(e);
        if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) ix->env_flush = v;
    }
    if (const char *e = std::getenv("LOCREC_KNN_ENTER")) ix->env_enter = std::max(0, std::atoi(e));
}

// LOCREC_KNN_HOST_BUILD: locrec_knn_create builds the index with round 1's host code (knn_host_build.h)
bool knn_host_build_requested() { return std::getenv("LOCREC_KNN_HOST_BUILD") != nullptr; }
}  // namespace locrec

#include "knn_host_build.h"

extern "C" int32_t locrec_knn_create_from_device(
    int64_t n, const int64_t *person_ids,
    const int64_t *p_rowptr, const int32_t *p_idx, const double *p_val, int32_t p_dim,
    const int64_t *c_rowptr, const int32_t *c_idx, const double *c_val, int32_t c_dim,
    const int64_t *r_rowptr, const int64_t *r_place, const int64_t *r_rating,
    locrec_knn_index **out) try
{
    return knn_build_device(n, person_ids, p_rowptr, p_idx, p_val, p_dim, c_rowptr, c_idx, c_val, c_dim, r_rowptr, r_place,
                            r_rating, out);
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_create(
    int64_t n, const int64_t *person_ids,
    const int64_t *p_rowptr, const int32_t *p_idx, const double *p_val, int32_t p_dim,
    const int64_t *c_rowptr, const int32_t *c_idx, const double *c_val, int32_t c_dim,
    const int64_t *r_rowptr, const int64_t *r_place, const int64_t *r_rating,
    locrec_knn_index **out) try
{
    if (locrec::knn_host_build_requested())
        return knn_create_host(n, person_ids, p_rowptr, p_idx, p_val, p_dim, c_rowptr, c_idx, c_val, c_dim, r_rowptr, r_place,
                               r_rating, out);
    if (!out) return fail(LOCREC_E_INVALID_ARG, "out_index is NULL");
    *out = nullptr;
    if (n < 0 || n >= ((int64_t)1 << 31) - 64) return fail(LOCREC_E_INVALID_ARG, "bad person count");
    if (n > 0 && (!person_ids || !p_rowptr || !c_rowptr)) return fail(LOCREC_E_INVALID_ARG, "NULL input array");
    if (p_dim <= 0 || c_dim <= 0) return fail(LOCREC_E_INVALID_ARG, "vector sizes must be positive");
    LOCREC_TRY(ensure_device());
    // the row pointers decide how many elements are uploaded: they are checked here, everything else on the device
    auto check_ptr = [&](const char *name, const int64_t *ptr) -> int32_t {
        if (!ptr || n == 0) return LOCREC_OK;
        if (ptr[0] != 0) return fail(LOCREC_E_INVALID_ARG, "%s rowptr must start at 0", name);
        for (int64_t r = 0; r < n; ++r)
            if (ptr[r + 1] < ptr[r]) return fail(LOCREC_E_INVALID_ARG, "%s rowptr not monotone at %lld", name, (long long)r);
        return LOCREC_OK;
    };
    LOCREC_TRY(check_ptr("place", p_rowptr));
    LOCREC_TRY(check_ptr("category", c_rowptr));
    LOCREC_TRY(check_ptr("ratings", r_rowptr));
    const int64_t pe = n > 0 ? p_rowptr[n] : 0, ce = n > 0 ? c_rowptr[n] : 0, re = n > 0 && r_rowptr ? r_rowptr[n] : 0;
    if ((pe > 0 && (!p_idx || !p_val)) || (ce > 0 && (!c_idx || !c_val)) || (re > 0 && (!r_place || !r_rating)))
        return fail(LOCREC_E_INVALID_ARG, "NULL input array");
    DevBuf<int64_t> d_ids, d_pp, d_cp, d_rp, d_rpl, d_rra;
    DevBuf<int32_t> d_pi, d_ci;
    DevBuf<double> d_pv, d_cv;
    LOCREC_TRY(d_ids.upload(person_ids, (size_t)n));
    LOCREC_TRY(d_pp.upload(p_rowptr, n > 0 ? (size_t)n + 1 : 0));
    LOCREC_TRY(d_pi.upload(p_idx, (size_t)pe));
    LOCREC_TRY(d_pv.upload(p_val, (size_t)pe));
    LOCREC_TRY(d_cp.upload(c_rowptr, n > 0 ? (size_t)n + 1 : 0));
    LOCREC_TRY(d_ci.upload(c_idx, (size_t)ce));
    LOCREC_TRY(d_cv.upload(c_val, (size_t)ce));
    if (r_rowptr) {
        LOCREC_TRY(d_rp.upload(r_rowptr, n > 0 ? (size_t)n + 1 : 0));
        LOCREC_TRY(d_rpl.upload(r_place, (size_t)re));
        LOCREC_TRY(d_rra.upload(r_rating, (size_t)re));
    }
    LOCREC_HIP_TRY(hipDeviceSynchronize());
    return knn_build_device(n, d_ids.p, d_pp.p, d_pi.p, d_pv.p, p_dim, d_cp.p, d_ci.p, d_cv.p, c_dim, r_rowptr ? d_rp.p : nullptr,
                            r_rowptr ? d_rpl.p : nullptr, r_rowptr ? d_rra.p : nullptr, out);
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_destroy(locrec_knn_index *ix) try
{
    if (!ix) return LOCREC_OK;
    (void)hipSetDevice(ix->device);
    if (ix->stream) (void)hipStreamSynchronize(ix->stream);
    delete ix;  // the destructor destroys an owned stream
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_info(const locrec_knn_index *ix, int64_t *out_n, int64_t *out_bytes,
                                   int32_t *out_packed) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    if (out_n) *out_n = ix->n;
    if (out_bytes) *out_bytes = scan_bytes_total(ix);
    if (out_packed) *out_packed = ix->packed ? (ix->pack16 ? 2 : 1) : 0;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_batch_scan_bytes(const locrec_knn_index *ix, int64_t *out_bytes) try
{
    if (!ix || !out_bytes) return fail(LOCREC_E_INVALID_ARG, "NULL argument");
    *out_bytes = ix->ht.ready && !ix->no_ht ? ix->ht.scan_bytes : scan_bytes_total(ix);
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_ht_image_info(const locrec_knn_index *ix, int64_t *out_head_words, int64_t *out_tail_postings,
                                            int64_t *out_wide_rows) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "NULL argument");
    const bool have = ix->ht.ready && !ix->no_ht;
    // (both SELL buffers carry kHtNP * 256 words of load padding behind the image)
    const int64_t pad = (int64_t)cfg::kHtNP * 256;
    if (out_head_words) *out_head_words = have ? (int64_t)ix->ht.p_sell.n + (int64_t)ix->ht.c_sell.n - 2 * pad : 0;
    if (out_tail_postings) *out_tail_postings = have && !ix->ht.tail_hits_ps.empty() ? (int64_t)ix->ht.post.n : 0;
    if (out_wide_rows) *out_wide_rows = (int64_t)ix->wide_rows.size();
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_scan_plan(const locrec_knn_index *ix, int32_t *out_kernel, int32_t *out_mode,
                                        int32_t *out_query_tile, int32_t *out_waves_per_block) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    if (out_kernel) *out_kernel = ix->last_plan_kernel;
    if (out_mode) *out_mode = ix->last_plan_mode;
    if (out_query_tile) *out_query_tile = ix->last_plan_qt;
    if (out_waves_per_block) *out_waves_per_block = ix->last_plan_waves;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_vector_lengths(locrec_knn_index *ix, double *out_p, double *out_c) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    LOCREC_HIP_TRY(hipSetDevice(ix->device));
    std::vector<double> np_((size_t)ix->n), nc_((size_t)ix->n);
    LOCREC_HIP_TRY(hipMemcpyAsync(np_.data(), ix->fp.norm.p, (size_t)ix->n * 8, hipMemcpyDeviceToHost, ix->stream));
    LOCREC_HIP_TRY(hipMemcpyAsync(nc_.data(), ix->fc.norm.p, (size_t)ix->n * 8, hipMemcpyDeviceToHost, ix->stream));
    LOCREC_HIP_TRY(hipStreamSynchronize(ix->stream));
    for (int64_t i = 0; i < ix->n; ++i) {
        if (out_p) out_p[i] = np_[ix->row_of_input[i]];
        if (out_c) out_c[i] = nc_[ix->row_of_input[i]];
    }
    return LOCREC_OK;
} LOCREC_CATCH_ALL

// Distance.cosineSimilarity of two rows as the device holds them (fp64 CSR + the norms of create time): one thread,
// the merge in ascending index order, one multiply and one divide (Distance.scala:7-9) - any sign, NaN for an empty vector
__global__ void knn_cosine_pair(const SideCsr C, int32_t a, int32_t b, double *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double dp = side_merge_dot(C.p_ptr, C.p_idx, C.p_val, a, b);
    const double dc = side_merge_dot(C.c_ptr, C.c_idx, C.c_val, a, b);
    const double denp = C.norm_p[a] * C.norm_p[b], denc = C.norm_c[a] * C.norm_c[b];
    out[0] = dp / denp;
    out[1] = dc / denc;
}

extern "C" int32_t locrec_knn_cosine_similarity(locrec_knn_index *ix, int64_t person_a, int64_t person_b,
                                                double *out_place_cosine, double *out_category_cosine) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    const int32_t a = ix->row_of_person(person_a), b = ix->row_of_person(person_b);
    if (a < 0) return fail(LOCREC_E_NOT_FOUND, "No such person: %lld", (long long)person_a);
    if (b < 0) return fail(LOCREC_E_NOT_FOUND, "No such person: %lld", (long long)person_b);
    LOCREC_HIP_TRY(hipSetDevice(ix->device));
    DevBuf<double> out;
    LOCREC_TRY(out.alloc(2));
    hipLaunchKernelGGL(knn_cosine_pair, dim3(1), dim3(64), 0, ix->stream, side_csr_of(ix), a, b, out.p);
    double h[2] = {0.0, 0.0};
    LOCREC_HIP_TRY(hipMemcpyAsync(h, out.p, sizeof(h), hipMemcpyDeviceToHost, ix->stream));
    LOCREC_HIP_TRY(hipStreamSynchronize(ix->stream));
    if (out_place_cosine) *out_place_cosine = h[0];
    if (out_category_cosine) *out_category_cosine = h[1];
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_set_stream(locrec_knn_index *ix, void *s) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    if (ix->own_stream && ix->stream) {
        (void)hipStreamSynchronize(ix->stream);
        (void)hipStreamDestroy(ix->stream);
    }
    ix->stream = reinterpret_cast<hipStream_t>(s);
    ix->own_stream = false;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_synchronize(locrec_knn_index *ix) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    LOCREC_HIP_TRY(hipSetDevice(ix->device));
    LOCREC_HIP_TRY(hipStreamSynchronize(ix->stream));
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_profile_enable(locrec_knn_index *ix, int32_t on) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    ix->prof.on = on != 0;
    ix->prof.used = 0;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_profile_read(locrec_knn_index *ix, double *ms, int64_t *launches) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    LOCREC_HIP_TRY(hipSetDevice(ix->device));
    return ix->prof.read(ix->stream, ms, launches);
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_row_person_ids(locrec_knn_index *ix, int64_t first, int64_t nq, int64_t *out) try
{
    if (!ix || !out) return fail(LOCREC_E_INVALID_ARG, "NULL argument");
    if (first < 0 || nq < 0 || first + nq > ix->n) return fail(LOCREC_E_INVALID_ARG, "row range out of bounds");
    std::copy(ix->ids_row.begin() + first, ix->ids_row.begin() + first + nq, out);
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_topk_range_async(locrec_knn_index *ix, int64_t first, int64_t nq,
                                               double pw, double cw, int64_t k) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    ix->have_result = false;
    LOCREC_TRY(check_params(pw, cw, k));
    if (k > LOCREC_KNN_BATCH_MAX_K) return fail(LOCREC_E_INVALID_ARG, "k_nearest %lld exceeds the batch limit %d", (long long)k, LOCREC_KNN_BATCH_MAX_K);
    if (first < 0 || nq <= 0 || first + nq > ix->n) return fail(LOCREC_E_INVALID_ARG, "row range out of bounds");
    LOCREC_HIP_TRY(hipSetDevice(ix->device));
    // rows are sorted by (nnz_place, nnz_category): the last row of the range has the largest nnz_place
    int max_p = 0, max_c = 0;
    bool any_absent = false;
    for (int64_t r = first; r < first + nq; ++r) {
        any_absent = any_absent || ix->fp.nnz[r] == 0 || ix->fc.nnz[r] == 0;
        max_p = std::max(max_p, ix->fp.nnz[r]);
        max_c = std::max(max_c, ix->fc.nnz[r]);
    }
    LOCREC_TRY(enqueue_topk(ix, nullptr, (int32_t)first, nq, std::max(1, max_p), std::max(1, max_c), pw, cw, k, any_absent));
    return LOCREC_OK;
} LOCREC_CATCH_ALL

// Flush intervals replayed with synchronous insertion inside knn_scan since the index was created
// (include/locrec.h).
extern "C" int32_t locrec_knn_replayed_intervals(locrec_knn_index *ix, int64_t *out_blocks) try
{
    if (!ix || !out_blocks) return fail(LOCREC_E_INVALID_ARG, "NULL argument");
    *out_blocks = 0;
    if (!ix->scan_overflow.p) return LOCREC_OK;
    LOCREC_HIP_TRY(hipSetDevice(ix->device));
    int32_t v = 0;
    LOCREC_HIP_TRY(hipMemcpyAsync(&v, ix->scan_overflow.p + 1, sizeof(int32_t), hipMemcpyDeviceToHost, ix->stream));
    LOCREC_HIP_TRY(hipStreamSynchronize(ix->stream));
    *out_blocks = v;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_fetch_topk(locrec_knn_index *ix, int64_t nq, int64_t k,
                                         int64_t *out_ids, double *out_sims, int64_t *out_counts) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    if (!ix->have_result || nq != ix->last_nq || k != ix->last_k)
        return fail(LOCREC_E_INVALID_ARG, "no matching result to fetch");
    LOCREC_HIP_TRY(hipSetDevice(ix->device));
    hipStream_t s = ix->stream;
    int32_t overflow = 0, qoverflow = 0;
    const size_t rb = (size_t)nq * k * 8, cb = (size_t)nq * 8;
    unsigned char *st = 2 * rb + cb + 16 <= locrec_knn_index::kStageBytes ? stage_of(ix) : nullptr;
    if (st) {
        // small result: everything lands in pinned memory, one synchronisation, then plain memcpys
        int32_t *flags = reinterpret_cast<int32_t *>(st);
        unsigned char *p_ids = st + 16, *p_sims = p_ids + rb, *p_cnt = p_sims + rb;
        const bool direct = ix->single_direct && ix->single_pending && nq == 1;
        if (!direct) flags[0] = flags[1] = 0;  // (direct: the device may already have written its flag)
        if (direct) {
            // knn_final1 has already written flag, ids, similarities and count here: nothing to enqueue
        } else if (ix->h_stage_dev && !ix->no_pack) {  // one gather launch into the pinned buffer (knn_pack_host)
            PackList pk;
            if (out_ids) pk.add(ix->out_ids.p, 16, rb);
            if (out_sims) pk.add(ix->out_sims.p, 16 + rb, rb);
            if (out_counts) pk.add(ix->out_cnt.p, 16 + 2 * rb, cb);
            if (ix->single_pending) pk.add(ix->sel1.p + 4, 0, sizeof(int32_t));
            if (ix->last_scan_fast) pk.add(ix->scan_overflow.p, 4, sizeof(int32_t));
            pk.launch(ix, s);
        } else {
            if (out_ids) LOCREC_HIP_TRY(hipMemcpyAsync(p_ids, ix->out_ids.p, rb, hipMemcpyDeviceToHost, s));
            if (out_sims) LOCREC_HIP_TRY(hipMemcpyAsync(p_sims, ix->out_sims.p, rb, hipMemcpyDeviceToHost, s));
            if (out_counts) LOCREC_HIP_TRY(hipMemcpyAsync(p_cnt, ix->out_cnt.p, cb, hipMemcpyDeviceToHost, s));
            if (ix->single_pending)
                LOCREC_HIP_TRY(hipMemcpyAsync(&flags[0], ix->sel1.p + 4, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            if (ix->last_scan_fast)
                LOCREC_HIP_TRY(hipMemcpyAsync(&flags[1], ix->scan_overflow.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        }
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
        overflow = ix->single_pending ? flags[0] : 0;
        qoverflow = ix->last_scan_fast ? flags[1] : 0;
        if (out_ids) std::memcpy(out_ids, p_ids, rb);
        if (out_sims) std::memcpy(out_sims, p_sims, rb);
        if (out_counts) std::memcpy(out_counts, p_cnt, cb);
    } else {
        if (out_ids) LOCREC_HIP_TRY(hipMemcpyAsync(out_ids, ix->out_ids.p, rb, hipMemcpyDeviceToHost, s));
        if (out_sims) LOCREC_HIP_TRY(hipMemcpyAsync(out_sims, ix->out_sims.p, rb, hipMemcpyDeviceToHost, s));
        if (out_counts) LOCREC_HIP_TRY(hipMemcpyAsync(out_counts, ix->out_cnt.p, cb, hipMemcpyDeviceToHost, s));
        if (ix->single_pending)
            LOCREC_HIP_TRY(hipMemcpyAsync(&overflow, ix->sel1.p + 4, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        if (ix->last_scan_fast)
            LOCREC_HIP_TRY(hipMemcpyAsync(&qoverflow, ix->scan_overflow.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
    }
    if (ix->last_scan_fast && qoverflow) {
        LOCREC_TRY(rerun_tiled_sync(ix));
        return locrec_knn_fetch_topk(ix, nq, k, out_ids, out_sims, out_counts);
    }
    if (ix->single_pending && overflow) {
        LOCREC_TRY(rerun_single_tiled(ix));
        return locrec_knn_fetch_topk(ix, nq, k, out_ids, out_sims, out_counts);
    }
    ix->single_pending = false;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_query_batch(locrec_knn_index *ix, int64_t nq, const int64_t *person_ids,
                                          double pw, double cw, int64_t k, int64_t *out_ids,
                                          double *out_sims, int64_t *out_counts) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    ix->have_result = false;
    LOCREC_TRY(check_params(pw, cw, k));
    if (k > LOCREC_KNN_BATCH_MAX_K) return fail(LOCREC_E_INVALID_ARG, "k_nearest %lld exceeds the batch limit %d", (long long)k, LOCREC_KNN_BATCH_MAX_K);
    if (nq < 0 || (nq > 0 && !person_ids)) return fail(LOCREC_E_INVALID_ARG, "bad query list");
    if (nq == 0) return LOCREC_OK;
    LOCREC_HIP_TRY(hipSetDevice(ix->device));
    // queries are processed in row order so that a tile holds queries of similar length
    std::vector<int32_t> rows((size_t)nq);
    int max_p = 0, max_c = 0;
    for (int64_t i = 0; i < nq; ++i) {
        LOCREC_TRY(find_query_row(ix, person_ids[i], &rows[i]));
        max_p = std::max(max_p, ix->fp.nnz[rows[i]]);
        max_c = std::max(max_c, ix->fc.nnz[rows[i]]);
    }
    std::vector<int32_t> ord((size_t)nq);
    std::iota(ord.begin(), ord.end(), 0);
    std::sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return rows[a] < rows[b]; });
    std::vector<int32_t> sorted_rows((size_t)nq);
    for (int64_t i = 0; i < nq; ++i) sorted_rows[i] = rows[ord[i]];
    LOCREC_TRY(ix->qrows.reserve((size_t)nq));
    LOCREC_HIP_TRY(hipMemcpyAsync(ix->qrows.p, sorted_rows.data(), (size_t)nq * 4, hipMemcpyHostToDevice, ix->stream));
    LOCREC_HIP_TRY(hipStreamSynchronize(ix->stream));
    ix->qrows_host = sorted_rows;
    LOCREC_TRY(enqueue_topk(ix, ix->qrows.p, 0, nq, max_p, max_c, pw, cw, k));
    std::vector<int64_t> t_ids((size_t)nq * k), t_cnt((size_t)nq);
    std::vector<double> t_sims((size_t)nq * k);
    LOCREC_TRY(locrec_knn_fetch_topk(ix, nq, k, t_ids.data(), t_sims.data(), t_cnt.data()));
    for (int64_t i = 0; i < nq; ++i) {
        const int64_t dst = ord[i];
        if (out_ids) std::copy(t_ids.begin() + i * k, t_ids.begin() + (i + 1) * k, out_ids + dst * k);
        if (out_sims) std::copy(t_sims.begin() + i * k, t_sims.begin() + (i + 1) * k, out_sims + dst * k);
        if (out_counts) out_counts[dst] = t_cnt[i];
    }
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_all_pairs_topk(locrec_knn_index *ix, double pw, double cw, int64_t k,
                                             int64_t *out_ids, double *out_sims, int64_t *out_counts) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    LOCREC_TRY(check_params(pw, cw, k));
    if (k > LOCREC_KNN_BATCH_MAX_K) return fail(LOCREC_E_INVALID_ARG, "k_nearest %lld exceeds the batch limit %d", (long long)k, LOCREC_KNN_BATCH_MAX_K);
    const int64_t n = ix->n;
    // bounded batches keep the workspaces small; results are scattered back to input order
    const int64_t batch = 65536;
    std::vector<int32_t> input_of_row((size_t)n);
    for (int64_t p = 0; p < n; ++p) input_of_row[ix->row_of_input[p]] = (int32_t)p;
    std::vector<int64_t> t_ids, t_cnt;
    std::vector<double> t_sims;
    for (int64_t first = 0; first < n; first += batch) {
        const int64_t nq = std::min(batch, n - first);
        LOCREC_TRY(locrec_knn_topk_range_async(ix, first, nq, pw, cw, k));
        t_ids.resize((size_t)nq * k);
        t_sims.resize((size_t)nq * k);
        t_cnt.resize((size_t)nq);
        LOCREC_TRY(locrec_knn_fetch_topk(ix, nq, k, t_ids.data(), t_sims.data(), t_cnt.data()));
        for (int64_t i = 0; i < nq; ++i) {
            const int64_t dst = input_of_row[first + i];
            if (out_ids) std::copy(t_ids.begin() + i * k, t_ids.begin() + (i + 1) * k, out_ids + dst * k);
            if (out_sims) std::copy(t_sims.begin() + i * k, t_sims.begin() + (i + 1) * k, out_sims + dst * k);
            if (out_counts) out_counts[dst] = t_cnt[i];
        }
    }
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_query(locrec_knn_index *ix, int64_t person_id, double pw, double cw,
                                    int64_t k, int64_t *out_ids, double *out_sims, int64_t *inout_count) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    if (!inout_count) return fail(LOCREC_E_INVALID_ARG, "inout_count is NULL");
    LOCREC_TRY(check_params(pw, cw, k));
    int32_t row = 0;
    LOCREC_TRY(find_query_row(ix, person_id, &row));
    // K larger than the number of other persons selects everybody: clamp (H4)
    const int64_t keff = std::min<int64_t>(k, std::max<int64_t>(1, ix->n - 1));
    LOCREC_HIP_TRY(hipSetDevice(ix->device));
    if (keff > LOCREC_KNN_BATCH_MAX_K)  // e.g. the shipped --k-nearest 2000000: sort every candidate
        return knn_large_topk(ix, row, pw, cw, keff, out_ids, out_sims, inout_count);
    LOCREC_TRY(enqueue_topk(ix, nullptr, row, 1, ix->fp.nnz[row], ix->fc.nnz[row], pw, cw, keff));
    std::vector<int64_t> ids((size_t)keff);
    std::vector<double> sims((size_t)keff);
    int64_t cnt = 0;
    LOCREC_TRY(locrec_knn_fetch_topk(ix, 1, keff, ids.data(), sims.data(), &cnt));
    const int64_t cap = *inout_count;
    for (int64_t i = 0; i < std::min(cap, cnt); ++i) {
        if (out_ids) out_ids[i] = ids[i];
        if (out_sims) out_sims[i] = sims[i];
    }
    *inout_count = cnt;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

namespace {

// makeRecommendations0 (:51-70) for the nq neighbour lists the last enqueue_topk left on the device
int32_t enqueue_aggregate(locrec_knn_index *ix, int64_t nq, int K)
{
    hipStream_t s = ix->stream;
    const int64_t tmax = std::min<int64_t>((int64_t)K * std::max<int64_t>(1, ix->max_r_nnz), kAggCap);
    const int M = pow2ceil((int)std::max<int64_t>(2, tmax));
    LOCREC_TRY(ix->agg_place.reserve((size_t)nq * M));
    LOCREC_TRY(ix->agg_est.reserve((size_t)nq * M));
    LOCREC_TRY(ix->agg_n.reserve((size_t)nq));
    LOCREC_TRY(ix->agg_overflow.reserve((size_t)nq));
    // Two passes.  The LDS a block reserves decides how many blocks share a CU, and the typical
    // query needs far less than the worst case: pass 1 runs every query with room for 2048 rating
    // rows (3 blocks of 512 threads per CU instead of one of 1024), pass 2 redoes with the full
    // capacity only the queries pass 1 flagged (its other blocks exit at once).
    const int M1 = std::min(M, 2048);
    const int threads1 = (M1 < M && K <= 512) ? 512 : kAggThreads;
    // 32-bit sort keys where the compact place index and the position fit them
    const int64_t nplaces = (int64_t)ix->cplace_ids.size();
    auto key32 = [&](int cap) { return ceil_log2i(nplaces + 2) + ceil_log2i(cap) <= 32; };
    // (per position: the key, the product rating x similarity, the neighbour's number - 14 or 18 bytes: four 512-thread
    // blocks of 2,048 positions share a CU's LDS)
    auto lds_of = [&](int cap) { return (size_t)cap * (key32(cap) ? 14 : 18) + 8 + (size_t)K * 16 + (size_t)(K + 1) * 4 + 16; };
    const size_t lds_full = lds_of(M);
    auto launch = [&](int cap, int threads, int redo) -> int32_t {
        const size_t lds = lds_of(cap);
        if (key32(cap)) {
            if (lds > 64 * 1024)
                LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_aggregate<uint32_t>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(knn_aggregate<uint32_t>, dim3((unsigned)nq), dim3(threads), lds, s, ix->out_rows.p, ix->out_sims.p,
                               ix->out_cnt.p, K, ix->r_ptr.p, ix->r_pidx.p, ix->r_rating.p, ix->cplace_dev.p, cap,
                               ix->agg_place.p, ix->agg_est.p, ix->agg_n.p, ix->agg_overflow.p, (int64_t)M, redo, ceil_log2i(cap));
        } else {
            if (lds > 64 * 1024)
                LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_aggregate<uint64_t>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(knn_aggregate<uint64_t>, dim3((unsigned)nq), dim3(threads), lds, s, ix->out_rows.p, ix->out_sims.p,
                               ix->out_cnt.p, K, ix->r_ptr.p, ix->r_pidx.p, ix->r_rating.p, ix->cplace_dev.p, cap,
                               ix->agg_place.p, ix->agg_est.p, ix->agg_n.p, ix->agg_overflow.p, (int64_t)M, redo, 16);
        }
        return LOCREC_OK;
    };
    (void)lds_full;
    LOCREC_TRY(launch(M1, threads1, 0));
    if (M1 < M) LOCREC_TRY(launch(M, kAggThreads, 1));
    LOCREC_HIP_TRY(hipGetLastError());
    ix->agg_M = M;
    ix->have_agg = true;
    return LOCREC_OK;
}

// Batched makeRecommendations beyond the LDS lists (K > LOCREC_KNN_BATCH_MAX_K; the shipped --k-nearest 2000000,
// bin/knn_recommender.sh:35).  K >= N - 1 selects every positive-similarity person: tiles of 16 queries without any
// top-K (knn_large.hip, knn_large_recommend_batch).  A K in between (more than 1024 but fewer than all) keeps the
// top-K semantics: such a batch is served query by query by the single-request large-K path when it is fetched.
int32_t enqueue_large_k_batch(locrec_knn_index *ix, const std::vector<int32_t> &rows, double pw, double cw, int64_t k)
{
    const int64_t nq = (int64_t)rows.size();
    ix->agg_rows = rows;
    ix->agg_first = -1;
    ix->agg_pw = pw;
    ix->agg_cw = cw;
    ix->last_nq = nq;
    ix->last_k = k;
    ix->single_pending = false;
    ix->have_result = false;  // (no neighbour lists are produced: locrec_knn_fetch_topk has nothing to read)
    if (k >= ix->n - 1) {
        const int64_t worst = nq * (int64_t)std::max<size_t>(1, ix->cplace_ids.size()) * 16;
        if (worst > ((int64_t)48 << 30))
            return fail(LOCREC_E_INVALID_ARG, "a batch of %lld queries at K = %lld may return %lld GB of rows: split it",
                        (long long)nq, (long long)k, (long long)(worst >> 30));
        ix->lkb_deferred = false;
        return knn_large_recommend_batch(ix, rows.data(), nq, pw, cw);
    }
    ix->lkb_deferred = true;
    ix->have_lkb = true;
    return LOCREC_OK;
}

// rows of a resident large-K batch (processing order), to host arrays
int32_t fetch_large_k_batch(locrec_knn_index *ix, int64_t nq, int64_t *out_offsets, int64_t *out_places, double *out_ratings,
                            int64_t *inout_capacity)
{
    hipStream_t s = ix->stream;
    const int64_t cap = *inout_capacity;
    if (ix->lkb_deferred) {
        std::vector<std::vector<int64_t>> bp((size_t)nq);
        std::vector<std::vector<double>> be((size_t)nq);
        out_offsets[0] = 0;
        for (int64_t q = 0; q < nq; ++q) {
            const int32_t row = ix->agg_rows[(size_t)q];
            int64_t c = 0;
            if (ix->fp.nnz[(size_t)row] > 0 && ix->fc.nnz[(size_t)row] > 0) {
                LOCREC_TRY(knn_large_recommend(ix, row, ix->agg_pw, ix->agg_cw, ix->last_k, nullptr, nullptr, &c));
                bp[(size_t)q].resize((size_t)c);
                be[(size_t)q].resize((size_t)c);
                LOCREC_TRY(knn_large_recommend(ix, row, ix->agg_pw, ix->agg_cw, ix->last_k, bp[(size_t)q].data(), be[(size_t)q].data(), &c));
            }
            out_offsets[q + 1] = out_offsets[q] + c;
        }
        *inout_capacity = out_offsets[nq];
        if (out_offsets[nq] > cap || out_offsets[nq] == 0) return LOCREC_OK;
        if (!out_places || !out_ratings) return fail(LOCREC_E_INVALID_ARG, "NULL output buffer");
        for (int64_t q = 0; q < nq; ++q) {
            std::copy(bp[(size_t)q].begin(), bp[(size_t)q].end(), out_places + out_offsets[q]);
            std::copy(be[(size_t)q].begin(), be[(size_t)q].end(), out_ratings + out_offsets[q]);
        }
        return LOCREC_OK;
    }
    for (int64_t q = 0; q <= nq; ++q) out_offsets[q] = ix->lkb_off[(size_t)q];
    const int64_t total = out_offsets[nq];
    *inout_capacity = total;
    if (total > cap || total == 0) return LOCREC_OK;
    if (!out_places || !out_ratings) return fail(LOCREC_E_INVALID_ARG, "NULL output buffer");
    LOCREC_HIP_TRY(hipMemcpyAsync(out_places, ix->lkb_place.p, (size_t)total * 8, hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipMemcpyAsync(out_ratings, ix->lkb_est.p, (size_t)total * 8, hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    return LOCREC_OK;
}

}  // namespace

// Batched makeRecommendations (the additive surface of SURVEY.md 8b): findSimilarPersons + the
// similarity-weighted aggregation for the persons at internal rows [first, first + nq); everything
// stays on the device until locrec_knn_fetch_recommend.
extern "C" int32_t locrec_knn_recommend_range_async(locrec_knn_index *ix, int64_t first, int64_t nq,
                                                    double pw, double cw, int64_t k) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    ix->have_agg = false;
    ix->have_lkb = false;
    if (k > LOCREC_KNN_BATCH_MAX_K) {
        LOCREC_TRY(check_params(pw, cw, k));
        if (first < 0 || nq <= 0 || first + nq > ix->n) return fail(LOCREC_E_INVALID_ARG, "row range out of bounds");
        LOCREC_HIP_TRY(hipSetDevice(ix->device));
        std::vector<int32_t> rows((size_t)nq);
        std::iota(rows.begin(), rows.end(), (int32_t)first);
        return enqueue_large_k_batch(ix, rows, pw, cw, k);
    }
    LOCREC_TRY(locrec_knn_topk_range_async(ix, first, nq, pw, cw, k));
    ix->agg_first = (int32_t)first;
    ix->agg_rows.clear();
    ix->agg_pw = pw;
    ix->agg_cw = cw;
    return enqueue_aggregate(ix, nq, (int)k);
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_fetch_recommend(locrec_knn_index *ix, int64_t nq, int64_t *out_offsets,
                                              int64_t *out_places, double *out_ratings, int64_t *inout_capacity) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    if (!out_offsets || !inout_capacity) return fail(LOCREC_E_INVALID_ARG, "NULL argument");
    if (ix->have_lkb && nq == ix->last_nq) {  // a large-K batch (K beyond the LDS lists)
        LOCREC_HIP_TRY(hipSetDevice(ix->device));
        return fetch_large_k_batch(ix, nq, out_offsets, out_places, out_ratings, inout_capacity);
    }
    if (!ix->have_agg || !ix->have_result || nq != ix->last_nq)
        return fail(LOCREC_E_INVALID_ARG, "no matching batched recommendation to fetch");
    LOCREC_HIP_TRY(hipSetDevice(ix->device));
    hipStream_t s = ix->stream;
    const int K = (int)ix->last_k;
    std::vector<int64_t> cnt((size_t)nq);
    std::vector<int32_t> ovf((size_t)nq);
    for (int attempt = 0;; ++attempt) {
        int32_t qoverflow = 0;
        LOCREC_HIP_TRY(hipMemcpyAsync(cnt.data(), ix->agg_n.p, (size_t)nq * 8, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(ovf.data(), ix->agg_overflow.p, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
        if (ix->last_scan_fast)
            LOCREC_HIP_TRY(hipMemcpyAsync(&qoverflow, ix->scan_overflow.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
        if (!(ix->last_scan_fast && qoverflow)) break;
        if (attempt > 0) return fail(LOCREC_E_DEVICE, "internal: survivor queue overflow on the synchronous path");
        // a survivor queue of the fast insertion path overflowed: redo scan (synchronous insertion) and aggregation
        LOCREC_TRY(rerun_tiled_sync(ix));
        LOCREC_TRY(enqueue_aggregate(ix, nq, K));
    }
    // queries whose neighbours hold more rating rows than one block sorts: place-major pass, one by one
    std::vector<std::vector<int64_t>> big_p((size_t)nq);
    std::vector<std::vector<double>> big_e((size_t)nq);
    for (int64_t q = 0; q < nq; ++q) {
        if (!ovf[q]) continue;
        const int32_t row = ix->agg_rows.empty() ? ix->agg_first + (int32_t)q : ix->agg_rows[(size_t)q];
        int64_t c = 0;
        LOCREC_TRY(knn_large_recommend(ix, row, ix->agg_pw, ix->agg_cw, ix->last_k, nullptr, nullptr, &c));
        big_p[q].resize((size_t)c);
        big_e[q].resize((size_t)c);
        LOCREC_TRY(knn_large_recommend(ix, row, ix->agg_pw, ix->agg_cw, ix->last_k, big_p[q].data(), big_e[q].data(), &c));
        cnt[q] = c;
    }
    std::vector<int64_t> doff((size_t)nq + 1, 0);  // offsets of the device-compacted rows (overflow queries: none)
    out_offsets[0] = 0;
    for (int64_t q = 0; q < nq; ++q) {
        doff[q + 1] = doff[q] + (ovf[q] ? 0 : cnt[q]);
        out_offsets[q + 1] = out_offsets[q] + cnt[q];
    }
    const int64_t total = out_offsets[nq], dtotal = doff[nq];
    const int64_t cap = *inout_capacity;
    *inout_capacity = total;
    if (total > cap || total == 0) return LOCREC_OK;  // caller looks at the returned total and calls again
    if (!out_places || !out_ratings) return fail(LOCREC_E_INVALID_ARG, "NULL output buffer");
    std::vector<int64_t> hp((size_t)std::max<int64_t>(1, dtotal));
    std::vector<double> he((size_t)std::max<int64_t>(1, dtotal));
    if (dtotal > 0) {
        // agg_n of an overflow query is 0, so the compaction kernel skips it by itself
        LOCREC_TRY(ix->agg_off.reserve((size_t)nq + 1));
        LOCREC_TRY(ix->agg_dense_place.reserve((size_t)dtotal));
        LOCREC_TRY(ix->agg_dense_est.reserve((size_t)dtotal));
        LOCREC_HIP_TRY(hipMemcpyAsync(ix->agg_off.p, doff.data(), (size_t)(nq + 1) * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(knn_agg_compact, dim3((unsigned)nq), dim3(256), 0, s, ix->agg_place.p, ix->agg_est.p,
                           ix->agg_n.p, ix->agg_off.p, (int64_t)ix->agg_M, ix->agg_dense_place.p, ix->agg_dense_est.p);
        LOCREC_HIP_TRY(hipGetLastError());
        LOCREC_HIP_TRY(hipMemcpyAsync(hp.data(), ix->agg_dense_place.p, (size_t)dtotal * 8, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(he.data(), ix->agg_dense_est.p, (size_t)dtotal * 8, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
    }
    for (int64_t q = 0; q < nq; ++q) {
        if (ovf[q]) {
            std::copy(big_p[q].begin(), big_p[q].end(), out_places + out_offsets[q]);
            std::copy(big_e[q].begin(), big_e[q].end(), out_ratings + out_offsets[q]);
        } else {
            std::copy(hp.begin() + doff[q], hp.begin() + doff[q + 1], out_places + out_offsets[q]);
            std::copy(he.begin() + doff[q], he.begin() + doff[q + 1], out_ratings + out_offsets[q]);
        }
    }
    return LOCREC_OK;
} LOCREC_CATCH_ALL

// makeRecommendations for a list of persons; rows of person i are
// [out_offsets[i], out_offsets[i + 1]) of out_places / out_ratings, ordered by place id.
extern "C" int32_t locrec_knn_recommend_batch(locrec_knn_index *ix, int64_t nq, const int64_t *person_ids,
                                              double pw, double cw, int64_t k, int64_t *out_offsets,
                                              int64_t *out_places, double *out_ratings, int64_t *inout_capacity) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    ix->have_result = false;
    ix->have_agg = false;
    ix->have_lkb = false;
    LOCREC_TRY(check_params(pw, cw, k));
    if (nq < 0 || (nq > 0 && !person_ids) || !out_offsets || !inout_capacity) return fail(LOCREC_E_INVALID_ARG, "bad arguments");
    if (nq == 0) {
        out_offsets[0] = 0;
        *inout_capacity = 0;
        return LOCREC_OK;
    }
    LOCREC_HIP_TRY(hipSetDevice(ix->device));
    std::vector<int32_t> rows((size_t)nq);
    int max_p = 0, max_c = 0;
    for (int64_t i = 0; i < nq; ++i) {
        LOCREC_TRY(find_query_row(ix, person_ids[i], &rows[i]));
        max_p = std::max(max_p, ix->fp.nnz[rows[i]]);
        max_c = std::max(max_c, ix->fc.nnz[rows[i]]);
    }
    // processed in row order (tiles of similar length), handed back in input order
    std::vector<int32_t> ord((size_t)nq);
    std::iota(ord.begin(), ord.end(), 0);
    std::sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return rows[a] < rows[b]; });
    ix->agg_rows.resize((size_t)nq);
    for (int64_t i = 0; i < nq; ++i) ix->agg_rows[i] = rows[ord[i]];
    if (k > LOCREC_KNN_BATCH_MAX_K) {
        // beyond the LDS lists (the shipped --k-nearest 2000000): tiles of queries without a top-K (knn_large.hip)
        const std::vector<int32_t> sorted_rows = ix->agg_rows;
        LOCREC_TRY(enqueue_large_k_batch(ix, sorted_rows, pw, cw, k));
    } else {
    LOCREC_TRY(ix->qrows.reserve((size_t)nq));
    LOCREC_HIP_TRY(hipMemcpyAsync(ix->qrows.p, ix->agg_rows.data(), (size_t)nq * 4, hipMemcpyHostToDevice, ix->stream));
    LOCREC_HIP_TRY(hipStreamSynchronize(ix->stream));
    ix->qrows_host = ix->agg_rows;
    LOCREC_TRY(enqueue_topk(ix, ix->qrows.p, 0, nq, max_p, max_c, pw, cw, k));
    ix->agg_first = -1;
    ix->agg_pw = pw;
    ix->agg_cw = cw;
    LOCREC_TRY(enqueue_aggregate(ix, nq, (int)k));
    }
    std::vector<int64_t> t_off((size_t)nq + 1);
    int64_t tcap = 0;
    LOCREC_TRY(locrec_knn_fetch_recommend(ix, nq, t_off.data(), nullptr, nullptr, &tcap));  // sizes only
    const int64_t total = tcap;
    std::vector<int64_t> len((size_t)nq);
    for (int64_t i = 0; i < nq; ++i) len[ord[i]] = t_off[i + 1] - t_off[i];
    out_offsets[0] = 0;
    for (int64_t i = 0; i < nq; ++i) out_offsets[i + 1] = out_offsets[i] + len[i];
    const int64_t cap = *inout_capacity;
    *inout_capacity = total;
    if (total > cap || total == 0) return LOCREC_OK;
    if (!out_places || !out_ratings) return fail(LOCREC_E_INVALID_ARG, "NULL output buffer");
    std::vector<int64_t> tp((size_t)total);
    std::vector<double> te((size_t)total);
    tcap = total;
    LOCREC_TRY(locrec_knn_fetch_recommend(ix, nq, t_off.data(), tp.data(), te.data(), &tcap));
    for (int64_t i = 0; i < nq; ++i) {
        const int64_t dst = out_offsets[ord[i]];
        std::copy(tp.begin() + t_off[i], tp.begin() + t_off[i + 1], out_places + dst);
        std::copy(te.begin() + t_off[i], te.begin() + t_off[i + 1], out_ratings + dst);
    }
    return LOCREC_OK;
} LOCREC_CATCH_ALL

// findSimilarPersons (:27-49) restricted to one shard of the candidates: the local top-K of a
// request whose candidate scan is split over several GPUs (every GPU holds the whole index).
extern "C" int32_t locrec_knn_query_shard(locrec_knn_index *ix, int64_t person_id, double pw, double cw,
                                          int64_t k, int32_t shard_index, int32_t shard_count,
                                          int64_t *out_ids, double *out_sims, int64_t *inout_count) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    if (!inout_count) return fail(LOCREC_E_INVALID_ARG, "inout_count is NULL");
    LOCREC_TRY(check_params(pw, cw, k));
    if (shard_count < 1 || shard_index < 0 || shard_index >= shard_count)
        return fail(LOCREC_E_INVALID_ARG, "shard %d of %d", shard_index, shard_count);
    int32_t row = 0;
    LOCREC_TRY(find_query_row(ix, person_id, &row));
    const int64_t keff = std::min<int64_t>(k, std::max<int64_t>(1, ix->n - 1));
    if (keff > LOCREC_KNN_BATCH_MAX_K)
        return fail(LOCREC_E_INVALID_ARG, "k_nearest %lld exceeds the limit %d of a sharded request", (long long)k,
                    LOCREC_KNN_BATCH_MAX_K);
    LOCREC_HIP_TRY(hipSetDevice(ix->device));
    int32_t s0 = 0, s1 = 0;
    shard_slice_range(ix, shard_index, shard_count, &s0, &s1);
    if (s0 >= s1) {  // more shards than slices: this one is empty
        *inout_count = 0;
        return LOCREC_OK;
    }
    CandRangeGuard guard(ix, s0, s1);  // also covers the reruns inside locrec_knn_fetch_topk
    LOCREC_TRY(enqueue_topk(ix, nullptr, row, 1, ix->fp.nnz[row], ix->fc.nnz[row], pw, cw, keff));
    std::vector<int64_t> ids((size_t)keff);
    std::vector<double> sims((size_t)keff);
    int64_t cnt = 0;
    LOCREC_TRY(locrec_knn_fetch_topk(ix, 1, keff, ids.data(), sims.data(), &cnt));
    const int64_t cap = *inout_count;
    for (int64_t i = 0; i < std::min(cap, cnt); ++i) {
        if (out_ids) out_ids[i] = ids[i];
        if (out_sims) out_sims[i] = sims[i];
    }
    *inout_count = cnt;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

// makeRecommendations0 (:51-70) for a given list of similar persons (e.g. the merged local lists
// of a sharded request), in the given order.
extern "C" int32_t locrec_knn_recommend_neighbours(locrec_knn_index *ix, int64_t n_neighbours,
                                                   const int64_t *neighbour_ids, const double *similarities,
                                                   int64_t *out_places, double *out_ratings, int64_t *inout_count) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    if (!inout_count) return fail(LOCREC_E_INVALID_ARG, "inout_count is NULL");
    if (n_neighbours < 0 || (n_neighbours > 0 && (!neighbour_ids || !similarities)))
        return fail(LOCREC_E_INVALID_ARG, "bad neighbour list");
    if (n_neighbours == 0) {
        *inout_count = 0;
        return LOCREC_OK;
    }
    std::vector<int32_t> rows((size_t)n_neighbours);
    for (int64_t i = 0; i < n_neighbours; ++i) {
        const int32_t nr = ix->row_of_person(neighbour_ids[i]);
        if (nr < 0) return fail(LOCREC_E_NOT_FOUND, "No such person: %lld", (long long)neighbour_ids[i]);
        if (!(similarities[i] > 0)) return fail(LOCREC_E_INVALID_ARG, "similarity of neighbour %lld is not positive", (long long)neighbour_ids[i]);
        rows[i] = nr;
    }
    {
        // findSimilarPersons never lists a person twice; a repeated id would be counted twice by the
        // LDS aggregation and once by the place-major pass: reject it on either path
        std::vector<int32_t> sorted_rows(rows);
        std::sort(sorted_rows.begin(), sorted_rows.end());
        const auto dup = std::adjacent_find(sorted_rows.begin(), sorted_rows.end());
        if (dup != sorted_rows.end())
            return fail(LOCREC_E_INVALID_ARG, "neighbour %lld is listed twice", (long long)ix->ids_row[*dup]);
    }
    LOCREC_HIP_TRY(hipSetDevice(ix->device));
    hipStream_t s = ix->stream;
    ix->have_result = false;
    ix->single_pending = false;
    if (n_neighbours <= LOCREC_KNN_BATCH_MAX_K) {
        const int K = (int)n_neighbours;
        const int64_t cnt = n_neighbours;
        LOCREC_TRY(ix->out_rows.reserve((size_t)K));
        LOCREC_TRY(ix->out_sims.reserve((size_t)K));
        LOCREC_TRY(ix->out_cnt.reserve(1));
        LOCREC_HIP_TRY(hipMemcpyAsync(ix->out_rows.p, rows.data(), (size_t)K * 4, hipMemcpyHostToDevice, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(ix->out_sims.p, similarities, (size_t)K * 8, hipMemcpyHostToDevice, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(ix->out_cnt.p, &cnt, 8, hipMemcpyHostToDevice, s));
        const int64_t tmax = std::min<int64_t>((int64_t)K * std::max<int64_t>(1, ix->max_r_nnz), kAggCap);
        const int M = pow2ceil((int)std::max<int64_t>(2, tmax));
        LOCREC_TRY(ix->agg_place.reserve((size_t)M));
        LOCREC_TRY(ix->agg_est.reserve((size_t)M));
        LOCREC_TRY(ix->agg_n.reserve(1));
        LOCREC_TRY(ix->agg_overflow.reserve(1));
        const size_t lds = (size_t)M * 18 + 8 + (size_t)K * 16 + (size_t)(K + 1) * 4 + 16;
        if (lds > 64 * 1024)
            LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_aggregate<uint64_t>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(knn_aggregate<uint64_t>, dim3(1), dim3(kAggThreads), lds, s, ix->out_rows.p, ix->out_sims.p,
                           ix->out_cnt.p, K, ix->r_ptr.p, ix->r_pidx.p, ix->r_rating.p, ix->cplace_dev.p, M,
                           ix->agg_place.p, ix->agg_est.p, ix->agg_n.p, ix->agg_overflow.p, (int64_t)M, 0, 16);
        LOCREC_HIP_TRY(hipGetLastError());
        int64_t nout = 0;
        int32_t overflow = 0;
        std::vector<int64_t> hp((size_t)M);
        std::vector<double> he((size_t)M);
        LOCREC_HIP_TRY(hipMemcpyAsync(&nout, ix->agg_n.p, 8, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(&overflow, ix->agg_overflow.p, 4, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(hp.data(), ix->agg_place.p, (size_t)M * 8, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(he.data(), ix->agg_est.p, (size_t)M * 8, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
        if (!overflow) {
            const int64_t w = std::min(*inout_count, nout);
            if (w > 0 && out_places) std::copy(hp.begin(), hp.begin() + w, out_places);
            if (w > 0 && out_ratings) std::copy(he.begin(), he.begin() + w, out_ratings);
            *inout_count = nout;
            return LOCREC_OK;
        }
    }
    // many neighbours or many rating rows: place-major pass over the transposed ratings
    std::vector<double> w((size_t)ix->n, 0.0);
    for (int64_t i = 0; i < n_neighbours; ++i) w[rows[i]] = similarities[i];
    return knn_large_aggregate(ix, w.data(), out_places, out_ratings, inout_count);
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_knn_recommend(locrec_knn_index *ix, int64_t person_id, double pw, double cw,
                                        int64_t k, int64_t *out_places, double *out_ratings,
                                        int64_t *inout_count) try
{
    if (!ix) return fail(LOCREC_E_INVALID_ARG, "index is NULL");
    if (!inout_count) return fail(LOCREC_E_INVALID_ARG, "inout_count is NULL");
    LOCREC_TRY(check_params(pw, cw, k));
    int32_t row = 0;
    LOCREC_TRY(find_query_row(ix, person_id, &row));
    const int64_t keff = std::min<int64_t>(k, std::max<int64_t>(1, ix->n - 1));
    LOCREC_HIP_TRY(hipSetDevice(ix->device));
    if (keff > LOCREC_KNN_BATCH_MAX_K)
        return knn_large_recommend(ix, row, pw, cw, keff, out_places, out_ratings, inout_count);
    hipStream_t s = ix->stream;
    static const bool dbg_timing = debug_env("LOCREC_DEBUG_TIMING") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    LOCREC_TRY(enqueue_topk(ix, nullptr, row, 1, ix->fp.nnz[row], ix->fc.nnz[row], pw, cw, keff));
    const auto tp1 = std::chrono::steady_clock::now();
    const int K = (int)keff;
    const int64_t tmax = std::min<int64_t>((int64_t)K * std::max<int64_t>(1, ix->max_r_nnz), kAggCap);
    const int M = pow2ceil((int)std::max<int64_t>(2, tmax));
    LOCREC_TRY(ix->agg_place.reserve((size_t)M));
    LOCREC_TRY(ix->agg_est.reserve((size_t)M));
    LOCREC_TRY(ix->agg_n.reserve(1));
    LOCREC_TRY(ix->agg_overflow.reserve(1));
    const size_t lds = (size_t)M * 18 + 8 + (size_t)K * 16 + (size_t)(K + 1) * 4 + 16;
    if (lds > 64 * 1024)
        LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_aggregate<uint64_t>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // a request on the stream path: knn_aggregate writes its result into the pinned staging buffer itself
    unsigned char *agg_host = nullptr;
    if (ix->single_pending && !ix->last_scan_fast && (size_t)M * 16 + 64 <= locrec_knn_index::kStageBytes && stage_of(ix) &&
        ix->h_stage_dev && !ix->no_pack)
        agg_host = ix->h_stage_dev;
    hipLaunchKernelGGL(knn_aggregate<uint64_t>, dim3(1), dim3(kAggThreads), lds, s, ix->out_rows.p, ix->out_sims.p,
                       ix->out_cnt.p, K, ix->r_ptr.p, ix->r_pidx.p, ix->r_rating.p, ix->cplace_dev.p, M,
                       ix->agg_place.p, ix->agg_est.p, ix->agg_n.p, ix->agg_overflow.p, (int64_t)M, 0, 16, agg_host,
                       ix->sel1.p + 4, M);
    LOCREC_HIP_TRY(hipGetLastError());
    const auto tp2 = std::chrono::steady_clock::now();
    // one batched read-back: counts, flags and the (at most M) rows
    int64_t nout = 0;
    int32_t overflow = 0, overflow1 = 0, qoverflow = 0;
    std::vector<int64_t> hp_v;
    std::vector<double> he_v;
    const int64_t *hp = nullptr;
    const double *he = nullptr;
    unsigned char *st = (size_t)M * 16 + 64 <= locrec_knn_index::kStageBytes ? stage_of(ix) : nullptr;
    int64_t *p_nout = &nout;
    int32_t *p_flags = nullptr;
    int32_t local_flags[3] = {0, 0, 0};
    if (st) {  // pinned: the six copies below do not block, the request synchronises once
        p_nout = reinterpret_cast<int64_t *>(st);
        p_flags = reinterpret_cast<int32_t *>(st + 16);
        hp = reinterpret_cast<const int64_t *>(st + 64);
        he = reinterpret_cast<const double *>(st + 64 + (size_t)M * 8);
    } else {
        p_flags = local_flags;
        hp_v.resize((size_t)M);
        he_v.resize((size_t)M);
        hp = hp_v.data();
        he = he_v.data();
    }
    if (!agg_host) {  // (direct: the device may already have written count and flags)
        *p_nout = 0;
        p_flags[0] = p_flags[1] = p_flags[2] = 0;
    }
    if (agg_host) {
        // nothing to enqueue: knn_aggregate has written count, flags and rows into the staging buffer
    } else if (st && ix->h_stage_dev && !ix->no_pack) {
        // one gather launch into the pinned buffer; only the agg_n rows the request really has cross PCIe
        PackList pk;
        pk.add(ix->agg_n.p, 0, 8);
        pk.add(ix->agg_overflow.p, 16, 4);
        if (ix->single_pending) pk.add(ix->sel1.p + 4, 20, sizeof(int32_t));
        if (ix->last_scan_fast) pk.add(ix->scan_overflow.p, 24, sizeof(int32_t));
        pk.add(ix->agg_place.p, 64, (size_t)M * 8, ix->agg_n.p, 8);
        pk.add(ix->agg_est.p, 64 + (size_t)M * 8, (size_t)M * 8, ix->agg_n.p, 8);
        pk.launch(ix, s);
    } else {
        LOCREC_HIP_TRY(hipMemcpyAsync(p_nout, ix->agg_n.p, 8, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(&p_flags[0], ix->agg_overflow.p, 4, hipMemcpyDeviceToHost, s));
        if (ix->single_pending)
            LOCREC_HIP_TRY(hipMemcpyAsync(&p_flags[1], ix->sel1.p + 4, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        if (ix->last_scan_fast)
            LOCREC_HIP_TRY(hipMemcpyAsync(&p_flags[2], ix->scan_overflow.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(const_cast<int64_t *>(hp), ix->agg_place.p, (size_t)M * 8, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(const_cast<double *>(he), ix->agg_est.p, (size_t)M * 8, hipMemcpyDeviceToHost, s));
    }
    const auto tp3 = std::chrono::steady_clock::now();
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    nout = *p_nout;
    overflow = p_flags[0];
    overflow1 = ix->single_pending ? p_flags[1] : 0;  // (only the flags this request asked for were written)
    qoverflow = ix->last_scan_fast ? p_flags[2] : 0;
    if (dbg_timing) {
        const auto tp4 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "[locrec recommend] enqueue_topk %.1f us, aggregate launch %.1f us, copies enqueued %.1f us, sync %.1f us\n",
                us(tp0, tp1), us(tp1, tp2), us(tp2, tp3), us(tp3, tp4));
    }
    if (ix->last_scan_fast && qoverflow) {  // a wave queue overflowed: synchronous insertion, then again
        const bool saved = ix->no_fast;
        ix->no_fast = true;
        const int32_t st2 = locrec_knn_recommend(ix, person_id, pw, cw, k, out_places, out_ratings, inout_count);
        ix->no_fast = saved;
        return st2;
    }
    if (ix->single_pending && overflow1) {  // tie mass overflowed the stream path: tiled path, then again
        const bool saved = ix->no_single;
        ix->no_single = true;
        const int32_t st2 = locrec_knn_recommend(ix, person_id, pw, cw, k, out_places, out_ratings, inout_count);
        ix->no_single = saved;
        return st2;
    }
    ix->single_pending = false;
    if (overflow)  // more rating rows than one block sorts in LDS: place-major aggregation instead
        return knn_large_recommend(ix, row, pw, cw, keff, out_places, out_ratings, inout_count);
    const int64_t cap = *inout_count;
    const int64_t w = std::min(cap, nout);
    if (w > 0 && out_places) std::copy(hp, hp + w, out_places);
    if (w > 0 && out_ratings) std::copy(he, he + w, out_ratings);
    *inout_count = nout;
    return LOCREC_OK;
} LOCREC_CATCH_ALL
