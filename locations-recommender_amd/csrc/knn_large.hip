// knn_large.hip -- K beyond what the per-query LDS lists hold (LOCREC_KNN_BATCH_MAX_K), and
// aggregation over more rating rows than one block sorts in LDS.
//
// The reference ships --k-nearest 2000000 (bin/knn_recommender.sh:35), i.e. "every person with a
// positive similarity is a neighbour" (KnnRecommender.scala:47-48 with K >= the candidate count).
//   knn_large_topk       S (knn_scan1) -> keys in person-id-rank order -> one STABLE descending device
//                        radix sort (rocPRIM, dev_prims.h; ties keep id-ascending order) -> first K
//   knn_large_recommend  the same selection, then makeRecommendations0 (:51-70) as a place-major
//                        pass over the transposed ratings: est[p] = sum r*s / sum s over the selected
//                        raters of p, each place summed by one wave in a fixed order.
// A library sort is used here on purpose: this is the rare large-K corner, not the hot path.

#include "dev_prims.h"

#include <algorithm>

#include "knn_index.h"

namespace {

using namespace locrec;

// keys[r] = bit pattern of the similarity of the person with id-rank r (all similarities are >= +0,
// so the unsigned order of the bits is the numeric order); vals[r] = r
// rows outside [row0, row1) are not candidates of this request (a candidate shard, locrec_knn_query_shard): S holds
// whatever an earlier request left there
__global__ void lk_gather_keys(const double *S, const int32_t *row_of_rid, int32_t n, int32_t row0, int32_t row1, uint64_t *keys,
                               uint32_t *vals)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int32_t row = row_of_rid[r];
    keys[r] = row >= row0 && row < row1 ? (uint64_t)__double_as_longlong(S[row]) : 0ull;
    vals[r] = (uint32_t)r;
}

// pad_to > m: entries [m, pad_to) of a batch slot are reset to "no neighbour" (-1, 0.0, -1) like every other path leaves
// them - the slot may hold a stand-in query's list
__global__ void lk_emit(const uint64_t *keys, const uint32_t *vals, int32_t m, const int64_t *ids_by_rank,
                        int64_t *out_ids, double *out_sims, const int32_t *row_of_rid = nullptr, int32_t *out_rows = nullptr,
                        int64_t *out_cnt = nullptr, int32_t pad_to = 0)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && out_cnt) *out_cnt = m;
    if (i >= m) {
        if (i < pad_to) {
            out_ids[i] = -1;
            out_sims[i] = 0.0;
            if (out_rows) out_rows[i] = -1;
        }
        return;
    }
    out_ids[i] = ids_by_rank[vals[i]];
    out_sims[i] = __longlong_as_double((long long)keys[i]);
    if (out_rows) out_rows[i] = row_of_rid[vals[i]];
}

// w[row] = similarity if the row is among the first m sorted entries, else 0
__global__ void lk_scatter_weights(const uint64_t *keys, const uint32_t *vals, int32_t m, const int32_t *row_of_rid,
                                   double *w)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    w[row_of_rid[vals[i]]] = __longlong_as_double((long long)keys[i]);
}

// Place-major aggregation in two steps.  A popular place has hundreds of thousands of raters (874 k at
// cfg2): one wave walking all of them was the whole cost of a large-K request (~10 ms).  The raters
// of a place are therefore cut into SEGMENTS of at most kSegRaters entries (a table built once per
// index): one wave per segment (lanes stride it, butterfly), then one thread per place adds its
// segments' sums in segment order - a fixed order, so the result is reproducible.
constexpr int kSegRaters = 4096;

__global__ __launch_bounds__(256) void lk_aggregate_segments(const int64_t *seg_begin, const int64_t *seg_end, int32_t nsegs,
                                                             const int32_t *cp_row, const double *cp_rating, const double *w,
                                                             double *seg_ws, double *seg_ss)
{
    const int lane = threadIdx.x & 63;
    const int sg = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sg >= nsegs) return;
    double ws = 0.0, ss = 0.0;
    // eight raters per lane in flight: the row loads, then the eight dependent gathers of w[] (one after
    // the other this loop was a chain of 64 memory round trips per segment); added in the same ascending
    // order as a plain loop
    const int64_t end = seg_end[sg];
    for (int64_t e0 = seg_begin[sg] + lane; e0 < end; e0 += 64 * 8) {
        int32_t rr[8];
        double rt[8], sv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t e = e0 + 64 * u;
            rr[u] = e < end ? cp_row[e] : -1;
            rt[u] = e < end ? cp_rating[e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) sv[u] = rr[u] >= 0 ? w[rr[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (sv[u] > 0) {
                const double wr = rt[u] * sv[u];  // col("rating") * col("similarity") (:59)
                ws = ws + wr;
                ss = ss + sv[u];
            }
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        ws = ws + __shfl_xor(ws, d);
        ss = ss + __shfl_xor(ss, d);
    }
    if (lane == 0) {
        seg_ws[sg] = ws;
        seg_ss[sg] = ss;
    }
}

__global__ void lk_sum_segments(const int32_t *place_seg0, int32_t nplaces, const double *seg_ws, const double *seg_ss,
                                double *out_ws, double *out_ss)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nplaces) return;
    double ws = 0.0, ss = 0.0;
    for (int sg = place_seg0[p]; sg < place_seg0[p + 1]; ++sg) {
        ws = ws + seg_ws[sg];
        ss = ss + seg_ss[sg];
    }
    out_ws[p] = ws;
    out_ss[p] = ss;
}

// The end of the place-major pass on the device: the places somebody among the neighbours rated (ss > 0), in
// ascending place order, with est = ws / ss (:67).  This was a host loop over all places after copying ws and ss
// back - 100 k iterations with a division each, 0.2 ms of a 0.5 ms request at cfg2.  Two small launches: the rated
// places of every tile of kFinishTile are counted, then every tile writes its rows behind the tiles before it (the
// prefix over at most a few hundred tile counts is recomputed by each block).
constexpr int kFinishTile = 2048;  // places per block: 256 threads x 8 consecutive places

__global__ __launch_bounds__(256) void lk_finish_count(const double *ss, int32_t nplaces, int32_t *tile_cnt)
{
    __shared__ int wsum[4];
    const int p0 = blockIdx.x * kFinishTile + threadIdx.x * 8;
    int c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) c += (p0 + i < nplaces && ss[p0 + i] > 0) ? 1 : 0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void lk_finish_emit(const double *ws, const double *ss, int32_t nplaces,
                                                      const int32_t *tile_cnt, const int64_t *cplace_ids, int64_t *out_place,
                                                      double *out_est, int64_t *out_n, int64_t *host_n)
{
    __shared__ int64_t base_s;
    __shared__ int wtot[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // rows of the tiles before this one
    int64_t before = 0;
    for (int i = t; i < (int)blockIdx.x; i += 256) before += tile_cnt[i];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) before += __shfl_xor(before, d);
    __shared__ int64_t bsum[4];
    if (lane == 0) bsum[wave] = before;
    const int p0 = blockIdx.x * kFinishTile + t * 8;
    bool keep[8];
    int c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        keep[i] = p0 + i < nplaces && ss[p0 + i] > 0;
        c += keep[i] ? 1 : 0;
    }
    // exclusive scan of c over the block: within the wave, then over the four waves
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    if (t == 0) base_s = bsum[0] + bsum[1] + bsum[2] + bsum[3];
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wtot[w];
    int64_t o = base_s + woff + (incl - c);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (keep[i]) {
            out_place[o] = cplace_ids[p0 + i];
            out_est[o] = ws[p0 + i] / ss[p0 + i];  // :67
            ++o;
        }
    }
    if (blockIdx.x == gridDim.x - 1 && t == 255) {  // the last thread of the last tile knows the total
        *out_n = o;
        if (host_n) *host_n = o;
    }
}

// segment table of the place-major ratings (lazy, once per index)
int32_t ensure_segments(locrec_knn_index *ix)
{
    if (ix->lk_nsegs >= 0) return LOCREC_OK;
    const int64_t np = (int64_t)ix->cplace_ids.size();
    std::vector<int64_t> cptr((size_t)np + 1, 0);
    if (np > 0) LOCREC_HIP_TRY(hipMemcpy(cptr.data(), ix->cp_ptr.p, ((size_t)np + 1) * 8, hipMemcpyDeviceToHost));
    std::vector<int64_t> sb, se;
    std::vector<int32_t> seg0((size_t)np + 1, 0);
    for (int64_t p = 0; p < np; ++p) {
        seg0[(size_t)p] = (int32_t)sb.size();
        for (int64_t b = cptr[(size_t)p]; b < cptr[(size_t)p + 1]; b += kSegRaters) {
            sb.push_back(b);
            se.push_back(std::min(b + kSegRaters, cptr[(size_t)p + 1]));
        }
    }
    seg0[(size_t)np] = (int32_t)sb.size();
    LOCREC_TRY(ix->lk_seg_begin.upload(sb, ix->stream));
    LOCREC_TRY(ix->lk_seg_end.upload(se, ix->stream));
    LOCREC_TRY(ix->lk_place_seg0.upload(seg0, ix->stream));
    LOCREC_TRY(ix->lk_seg_ws.alloc(sb.size()));
    LOCREC_TRY(ix->lk_seg_ss.alloc(sb.size()));
    LOCREC_HIP_TRY(hipStreamSynchronize(ix->stream));
    ix->lk_nsegs = (int32_t)sb.size();
    return LOCREC_OK;
}

// Sorted (similarity desc, id asc) list of ALL rows in ix->lk_keys_out / lk_vals_out; *m = number of
// candidates (similarity > 0).
int32_t sort_all(locrec_knn_index *ix, int32_t qrow, double pw, double cw, int64_t *m, bool scanned = false)
{
    hipStream_t s = ix->stream;
    const int32_t n = (int32_t)ix->n;
    if (!scanned) LOCREC_TRY(knn_enqueue_dense(ix, qrow, pw, cw));
    LOCREC_TRY(ix->lk_keys.reserve((size_t)n));
    LOCREC_TRY(ix->lk_keys_out.reserve((size_t)n));
    LOCREC_TRY(ix->lk_vals.reserve((size_t)n));
    LOCREC_TRY(ix->lk_vals_out.reserve((size_t)n));
    hipLaunchKernelGGL(lk_gather_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ix->S1.p, ix->row_of_rid.p, n,
                       ix->cand_slice0 * 64, (int32_t)std::min<int64_t>(ix->n, (int64_t)ix->cand_slice1 * 64), ix->lk_keys.p,
                       ix->lk_vals.p);
    size_t temp_bytes = 0;
    LOCREC_HIP_TRY(prim::sort_pairs_desc(nullptr, temp_bytes, ix->lk_keys.p, ix->lk_keys_out.p,
                                                                ix->lk_vals.p, ix->lk_vals_out.p, n, 0, 64, s));
    LOCREC_TRY(ix->lk_temp.reserve(temp_bytes));
    LOCREC_HIP_TRY(prim::sort_pairs_desc(ix->lk_temp.p, temp_bytes, ix->lk_keys.p,
                                                                ix->lk_keys_out.p, ix->lk_vals.p, ix->lk_vals_out.p, n,
                                                                0, 64, s));
    // number of candidates = histogram total = position of the first zero key: binary search on the host
    // would need the keys; count on the device side instead via the histogram the scan already filled
    std::vector<uint32_t> hist(4096);  // kHistBins of knn.hip
    LOCREC_HIP_TRY(hipMemcpyAsync(hist.data(), ix->hist1.p, hist.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    int64_t total = 0;
    for (uint32_t h : hist) total += h;
    *m = total;
    return LOCREC_OK;
}

// =====================================================================================================
// Batched large K.  bin/knn_recommender.sh:35 ships --k-nearest 2000000: with K >= N - 1
// findSimilarPersons (KnnRecommender.scala:47-48) selects EVERY person of positive similarity, so a batch needs no
// top-K at all: the similarities of a TILE of kLkbQt queries against all candidates are computed once
// (lkb_scan: S[row][query], one 128-byte line per candidate) and fed straight into the place-major aggregation
// (makeRecommendations0, :51-70) - the w[row] gather of lk_aggregate_segments, the cost of the single request,
// now serves sixteen queries per cache line.
//
// Dots: every candidate walks its plain CSR row against DENSE query tables qd[index][query] in global memory,
// sum += qd * value left to right in ascending index order - BLAS.dot's order, a product with an absent query
// entry adds +-0.0 - so similarities are the reference's bit for bit in EVERY stored format (the GENERIC fp64
// one included); the sums of the aggregation run in lk_aggregate_segments' order, so a batch's estimates equal
// the single request's bit for bit.
constexpr int kLkbQt = 16;

struct LkbScan {
    const int64_t *p_ptr, *c_ptr;
    const int32_t *p_idx, *c_idx;
    const double *p_val, *c_val;
    const double *norm_p, *norm_c;
    const double *qd_p, *qd_c;  // [dim][kLkbQt]
    int32_t qrow[kLkbQt];       // -1: no query in this slot
    int32_t nrows;
    double pw, cw;
    double *S;                  // [nrows][kLkbQt], or [kLkbQt][nrows] when transposed
    int32_t *cand;              // [kLkbQt] persons of positive similarity per query
    int32_t transposed;
    const uint32_t *bits;       // which place indices (mod the bitmap size) ANY query of the tile holds
    uint32_t bit_mask;          // bitmap size in bits - 1 (a power of two, <= kLkbBitmapBits)
};

// (the same arithmetic as knn.hip's exact_similarity: one multiply and one divide per family, "> 0", ps*pw + cs*cw)
__device__ __forceinline__ bool lkb_similarity(double dp, double dc, double cnp, double cnc, double qnp, double qnc, double pw,
                                               double cw, double &s)
{
    double ps = 0.0, cs = 0.0;
    bool have = false;
    if (cnp > 0.0) {
        const double den = cnp * qnp;
        const double t = dp / den;
        if (t > 0) { ps = t; have = true; }
    }
    if (cnc > 0.0) {
        const double den = cnc * qnc;
        const double t = dc / den;
        if (t > 0) { cs = t; have = true; }
    }
    const double a = ps * pw;
    const double b = cs * cw;
    s = a + b;
    return have;
}

struct LkbFill {
    const int64_t *ptr;
    const int32_t *idx;
    const double *val;
    double *qd;
    uint32_t *bits;     // presence bitmap of the place table (nullptr for the categories), bit = index & bit_mask
    uint32_t bit_mask;
    int32_t qrow[kLkbQt];
    int32_t set;
};

// grid (element blocks, kLkbQt): query t's vector scattered into (or wiped from) column t of the dense table
__global__ void lkb_fill(const LkbFill a)
{
    const int t = blockIdx.y;
    const int r = a.qrow[t];
    if (r < 0) return;
    const int64_t e = a.ptr[r] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < a.ptr[r + 1]) {
        const int32_t i = a.idx[e];
        a.qd[(int64_t)i * kLkbQt + t] = a.set ? a.val[e] : 0.0;
        if (a.bits) {
            const uint32_t b = (uint32_t)i & a.bit_mask;
            if (a.set) atomicOr(&a.bits[b >> 5], 1u << (b & 31u));
            else a.bits[b >> 5] = 0u;  // (wiping: whole words, by whoever comes)
        }
    }
}

// The tile's 16 queries hold a few hundred of the place dimensions between them, a candidate row is mostly OTHER places:
// the 128-byte line of the dense table (16 fp64 query values, nearly always all zero) was fetched from L2 for every
// candidate element - 3.2 GB per tile at cfg2, the whole cost of the kernel.  A bitmap of the indices any query of
// the tile holds (one bit per place, folded to at most 32 KB; lkb_fill sets it) sits in LDS: an element whose bit is
// clear adds value * (+0.0) to every dot - nothing, for finite values - and is skipped without touching the table.
constexpr uint32_t kLkbBitmapBits = 1u << 18;

__global__ __launch_bounds__(256) void lkb_scan(const LkbScan P)
{
    __shared__ int s_cand[kLkbQt];
    extern __shared__ uint32_t s_bits[];
    if (threadIdx.x < kLkbQt) s_cand[threadIdx.x] = 0;
    for (uint32_t i = threadIdx.x; i < (P.bit_mask + 1u) / 32u; i += blockDim.x) s_bits[i] = P.bits[i];
    __syncthreads();
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row < P.nrows) {
        double dp[kLkbQt], dc[kLkbQt];
#pragma unroll
        for (int t = 0; t < kLkbQt; ++t) dp[t] = dc[t] = 0.0;
        for (int64_t e = P.p_ptr[row]; e < P.p_ptr[row + 1]; ++e) {
            const int32_t i = P.p_idx[e];
            const uint32_t b = (uint32_t)i & P.bit_mask;
            if (!((s_bits[b >> 5] >> (b & 31u)) & 1u)) continue;  // no query of the tile holds this place
            const double v = P.p_val[e];
            const double *q = P.qd_p + (int64_t)i * kLkbQt;
#pragma unroll
            for (int t = 0; t < kLkbQt; ++t) {
                const double x = q[t] * v;  // x(kx) * y(ky), x = the query (Distance.scala:8)
                dp[t] = dp[t] + x;
            }
        }
        for (int64_t e = P.c_ptr[row]; e < P.c_ptr[row + 1]; ++e) {
            const double v = P.c_val[e];
            const double *q = P.qd_c + (int64_t)P.c_idx[e] * kLkbQt;
#pragma unroll
            for (int t = 0; t < kLkbQt; ++t) {
                const double x = q[t] * v;
                dc[t] = dc[t] + x;
            }
        }
        const double cnp = P.norm_p[row], cnc = P.norm_c[row];
#pragma unroll
        for (int t = 0; t < kLkbQt; ++t) {
            const int qr = P.qrow[t];
            double sx = 0.0;
            bool have = false;
            if (qr >= 0 && row != qr)  // person_id =!= personId (KnnRecommender.scala:89)
                have = lkb_similarity(dp[t], dc[t], cnp, cnc, P.norm_p[qr], P.norm_c[qr], P.pw, P.cw, sx);
            if (!have) sx = 0.0;
            // (transposed: one contiguous column per query - what the single request's selection kernels read)
            P.S[P.transposed ? (int64_t)t * P.nrows + row : (int64_t)row * kLkbQt + t] = sx;
            if (have) atomicAdd(&s_cand[t], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < kLkbQt && s_cand[threadIdx.x]) atomicAdd(&P.cand[threadIdx.x], s_cand[threadIdx.x]);
}

// lk_aggregate_segments for a tile of queries: one wave per segment of <= kSegRaters raters of a place, lane l takes
// raters l, l + 64, ... in ascending order (the single request's order), one 128-byte line of S per rater
__global__ __launch_bounds__(256) void lkb_aggregate_segments(const int64_t *seg_begin, const int64_t *seg_end, int32_t nsegs,
                                                              const int32_t *cp_row, const double *cp_rating, const double *S,
                                                              double *seg_ws, double *seg_ss)
{
    const int lane = threadIdx.x & 63;
    const int sg = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sg >= nsegs) return;
    double ws[kLkbQt], ss[kLkbQt];
#pragma unroll
    for (int t = 0; t < kLkbQt; ++t) ws[t] = ss[t] = 0.0;
    const int64_t end = seg_end[sg];
    for (int64_t e0 = seg_begin[sg] + lane; e0 < end; e0 += 64 * 2) {
        int32_t rr[2];
        double rt[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t e = e0 + 64 * u;
            rr[u] = e < end ? cp_row[e] : -1;
            rt[u] = e < end ? cp_rating[e] : 0.0;
        }
        double sv[2][kLkbQt];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < kLkbQt; ++t) sv[u][t] = rr[u] >= 0 ? S[(int64_t)rr[u] * kLkbQt + t] : 0.0;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < kLkbQt; ++t)
                if (sv[u][t] > 0) {
                    const double wr = rt[u] * sv[u][t];  // col("rating") * col("similarity") (:59)
                    ws[t] = ws[t] + wr;
                    ss[t] = ss[t] + sv[u][t];
                }
    }
#pragma unroll
    for (int t = 0; t < kLkbQt; ++t) {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            ws[t] = ws[t] + __shfl_xor(ws[t], d);
            ss[t] = ss[t] + __shfl_xor(ss[t], d);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < kLkbQt; ++t) {
            seg_ws[(int64_t)sg * kLkbQt + t] = ws[t];
            seg_ss[(int64_t)sg * kLkbQt + t] = ss[t];
        }
    }
}

// thread per (place, query): the place's segments in segment order; output query-major [query][place]
__global__ void lkb_sum_segments(const int32_t *place_seg0, int32_t nplaces, const double *seg_ws, const double *seg_ss,
                                 double *out_ws, double *out_ss)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nplaces * kLkbQt) return;
    const int p = (int)(i / kLkbQt), t = (int)(i % kLkbQt);
    double ws = 0.0, ss = 0.0;
    for (int sg = place_seg0[p]; sg < place_seg0[p + 1]; ++sg) {
        ws = ws + seg_ws[(int64_t)sg * kLkbQt + t];
        ss = ss + seg_ss[(int64_t)sg * kLkbQt + t];
    }
    out_ws[(int64_t)t * nplaces + p] = ws;
    out_ss[(int64_t)t * nplaces + p] = ss;
}

struct LkbBases {
    int64_t base[kLkbQt];
};

// the single request's lk_finish_count / lk_finish_emit per query of the tile: grid (tiles, kLkbQt)
__global__ __launch_bounds__(256) void lkb_finish_count(const double *ss, int32_t nplaces, int32_t tiles, int32_t *tile_cnt)
{
    __shared__ int wsum[4];
    const double *sq = ss + (int64_t)blockIdx.y * nplaces;
    const int p0 = blockIdx.x * kFinishTile + threadIdx.x * 8;
    int c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) c += (p0 + i < nplaces && sq[p0 + i] > 0) ? 1 : 0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.y * tiles + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void lkb_finish_emit(const double *ws, const double *ss, int32_t nplaces, int32_t tiles,
                                                       const int32_t *tile_cnt, const int64_t *cplace_ids, const LkbBases bases,
                                                       int64_t *out_place, double *out_est)
{
    __shared__ int64_t bsum[4];
    __shared__ int wtot[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = blockIdx.y;
    if (bases.base[q] < 0) return;
    const double *wq = ws + (int64_t)q * nplaces, *sq = ss + (int64_t)q * nplaces;
    const int32_t *tc = tile_cnt + q * tiles;
    int64_t before = 0;
    for (int i = t; i < (int)blockIdx.x; i += 256) before += tc[i];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) before += __shfl_xor(before, d);
    if (lane == 0) bsum[wave] = before;
    const int p0 = blockIdx.x * kFinishTile + t * 8;
    bool keep[8];
    int c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        keep[i] = p0 + i < nplaces && sq[p0 + i] > 0;
        c += keep[i] ? 1 : 0;
    }
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wtot[w];
    int64_t o = bases.base[q] + bsum[0] + bsum[1] + bsum[2] + bsum[3] + woff + (incl - c);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (keep[i]) {
            out_place[o] = cplace_ids[p0 + i];
            out_est[o] = wq[p0 + i] / sq[p0 + i];  // :67
            ++o;
        }
    }
}

// the histogram of one query's column of a TRANSPOSED tile (what knn_select1 of knn.hip reads; knn_collect1 then
// reads the column itself): how a batch serves the few queries that cannot ride a packed tile
constexpr int kLkHistBins = 4096;  // kHistBins of knn.hip

__global__ __launch_bounds__(256) void lkb_column_hist(const double *col, int32_t nrows, uint32_t *hist)
{
    __shared__ uint32_t s_hist[kLkHistBins];
    col += (size_t)blockIdx.y * (size_t)nrows;   // blockIdx.y = column of the tile
    hist += (size_t)blockIdx.y * kLkHistBins;
    for (int i = threadIdx.x; i < kLkHistBins; i += blockDim.x) s_hist[i] = 0u;
    __syncthreads();
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < nrows; row += gridDim.x * blockDim.x) {
        const double sx = col[row];
        if (sx > 0) {
            const int b = (int)(sx * (double)kLkHistBins);
            atomicAdd(&s_hist[b < kLkHistBins - 1 ? b : kLkHistBins - 1], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kLkHistBins; i += blockDim.x) {
        const uint32_t h = s_hist[i];
        if (h) atomicAdd(&hist[i], h);
    }
}

}  // namespace

namespace locrec {

int32_t knn_large_topk(locrec_knn_index *ix, int32_t qrow, double pw, double cw, int64_t k, int64_t *out_ids,
                       double *out_sims, int64_t *inout_count)
{
    hipStream_t s = ix->stream;
    int64_t cand = 0;
    LOCREC_TRY(sort_all(ix, qrow, pw, cw, &cand));
    const int64_t m = std::min(cand, k);
    const int64_t wr = std::min(m, *inout_count);
    if (wr > 0) {
        LOCREC_TRY(ix->out_ids.reserve((size_t)wr));
        LOCREC_TRY(ix->out_sims.reserve((size_t)wr));
        hipLaunchKernelGGL(lk_emit, dim3((unsigned)((wr + 255) / 256)), dim3(256), 0, s, ix->lk_keys_out.p,
                           ix->lk_vals_out.p, (int32_t)wr, ix->ids_by_rank.p, ix->out_ids.p, ix->out_sims.p);
        if (out_ids) LOCREC_HIP_TRY(hipMemcpyAsync(out_ids, ix->out_ids.p, (size_t)wr * 8, hipMemcpyDeviceToHost, s));
        if (out_sims) LOCREC_HIP_TRY(hipMemcpyAsync(out_sims, ix->out_sims.p, (size_t)wr * 8, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
    }
    *inout_count = m;
    return LOCREC_OK;
}

// The same top-K, delivered into slot `slot` of the DEVICE result arrays of a batch (ids, similarities,
// neighbour rows, count): how a batch serves a query that is too long for any LDS tile (knn.hip,
// enqueue_topk).  The arrays must already hold (slot + 1) * k entries.
int32_t knn_large_topk_device(locrec_knn_index *ix, int32_t qrow, double pw, double cw, int64_t k, int64_t slot)
{
    hipStream_t s = ix->stream;
    int64_t cand = 0;
    LOCREC_TRY(sort_all(ix, qrow, pw, cw, &cand));
    const int32_t m = (int32_t)std::min(cand, k);
    if ((size_t)((slot + 1) * k) > ix->out_ids.n || (size_t)((slot + 1) * k) > ix->out_rows.n || (size_t)slot >= ix->out_cnt.n)
        return fail(LOCREC_E_DEVICE, "result arrays too small for slot %lld", (long long)slot);
    hipLaunchKernelGGL(lk_emit, dim3((unsigned)std::max<int64_t>(1, (k + 255) / 256)), dim3(256), 0, s, ix->lk_keys_out.p,
                       ix->lk_vals_out.p, m, ix->ids_by_rank.p, ix->out_ids.p + slot * k, ix->out_sims.p + slot * k,
                       ix->row_of_rid.p, ix->out_rows.p + slot * k, ix->out_cnt.p + slot, (int32_t)k);
    LOCREC_HIP_TRY(hipGetLastError());
    return LOCREC_OK;
}

// est[p] = sum r*w / sum w over the raters of p with w > 0 (device weights, one per row)
static int32_t aggregate_places(locrec_knn_index *ix, const double *w, int64_t *out_places, double *out_ratings,
                                int64_t *inout_count)
{
    hipStream_t s = ix->stream;
    const int32_t np = (int32_t)ix->cplace_ids.size();
    LOCREC_TRY(ix->lk_ws.reserve((size_t)np));
    LOCREC_TRY(ix->lk_ss.reserve((size_t)np));
    LOCREC_TRY(ensure_segments(ix));
    if (np > 0) {
        if (ix->lk_nsegs > 0)
            hipLaunchKernelGGL(lk_aggregate_segments, dim3((unsigned)((ix->lk_nsegs + 3) / 4)), dim3(256), 0, s, ix->lk_seg_begin.p,
                               ix->lk_seg_end.p, ix->lk_nsegs, ix->cp_row.p, ix->cp_rating.p, w, ix->lk_seg_ws.p, ix->lk_seg_ss.p);
        hipLaunchKernelGGL(lk_sum_segments, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, ix->lk_place_seg0.p, np,
                           ix->lk_seg_ws.p, ix->lk_seg_ss.p, ix->lk_ws.p, ix->lk_ss.p);
    }
    LOCREC_HIP_TRY(hipGetLastError());
    const int64_t cap = *inout_count;
    int64_t outn = 0;
    if (np > 0) {
        // the rated places, compacted in ascending order, with their estimates: on the device (lk_finish_*)
        const int tiles = (np + kFinishTile - 1) / kFinishTile;
        LOCREC_TRY(ix->lk_tile_cnt.reserve((size_t)tiles));
        LOCREC_TRY(ix->lk_out_place.reserve((size_t)np));
        LOCREC_TRY(ix->lk_out_est.reserve((size_t)np));
        LOCREC_TRY(ix->lk_out_n.reserve(1));
        int64_t *pinned_n = nullptr, *pinned_n_dev = nullptr;
        if (knn_stage(ix) && ix->h_stage_dev && !ix->no_pack) {
            pinned_n = reinterpret_cast<int64_t *>(ix->h_stage);
            pinned_n_dev = reinterpret_cast<int64_t *>(ix->h_stage_dev);
        }
        hipLaunchKernelGGL(lk_finish_count, dim3((unsigned)tiles), dim3(256), 0, s, ix->lk_ss.p, np, ix->lk_tile_cnt.p);
        hipLaunchKernelGGL(lk_finish_emit, dim3((unsigned)tiles), dim3(256), 0, s, ix->lk_ws.p, ix->lk_ss.p, np, ix->lk_tile_cnt.p,
                           ix->cplace_dev.p, ix->lk_out_place.p, ix->lk_out_est.p, ix->lk_out_n.p, pinned_n_dev);
        LOCREC_HIP_TRY(hipGetLastError());
        if (!pinned_n) LOCREC_HIP_TRY(hipMemcpyAsync(&outn, ix->lk_out_n.p, 8, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
        if (pinned_n) outn = *pinned_n;
        const int64_t wr = std::min(cap, outn);
        if (wr > 0 && out_places)
            LOCREC_HIP_TRY(hipMemcpyAsync(out_places, ix->lk_out_place.p, (size_t)wr * 8, hipMemcpyDeviceToHost, s));
        if (wr > 0 && out_ratings)
            LOCREC_HIP_TRY(hipMemcpyAsync(out_ratings, ix->lk_out_est.p, (size_t)wr * 8, hipMemcpyDeviceToHost, s));
        if (wr > 0) LOCREC_HIP_TRY(hipStreamSynchronize(s));
    }
    *inout_count = outn;
    return LOCREC_OK;
}

// the same pass for weights given on the host (one per row, 0 = not a neighbour)
int32_t knn_large_aggregate(locrec_knn_index *ix, const double *w_host, int64_t *out_places, double *out_ratings,
                            int64_t *inout_count)
{
    LOCREC_TRY(ix->lk_w.reserve((size_t)ix->n));
    LOCREC_HIP_TRY(hipMemcpyAsync(ix->lk_w.p, w_host, (size_t)ix->n * sizeof(double), hipMemcpyHostToDevice, ix->stream));
    LOCREC_HIP_TRY(hipStreamSynchronize(ix->stream));  // w_host may be freed by the caller
    return aggregate_places(ix, ix->lk_w.p, out_places, out_ratings, inout_count);
}

// number of candidates (similarity > 0) of the scan that was just enqueued: the total of its histogram
static int32_t candidate_count(locrec_knn_index *ix, int64_t *m)
{
    std::vector<uint32_t> hist(4096);  // kHistBins of knn.hip
    LOCREC_HIP_TRY(hipMemcpyAsync(hist.data(), ix->hist1.p, hist.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ix->stream));
    LOCREC_HIP_TRY(hipStreamSynchronize(ix->stream));
    int64_t total = 0;
    for (uint32_t h : hist) total += h;
    *m = total;
    return LOCREC_OK;
}

int32_t knn_large_recommend(locrec_knn_index *ix, int32_t qrow, double pw, double cw, int64_t k,
                            int64_t *out_places, double *out_ratings, int64_t *inout_count)
{
    hipStream_t s = ix->stream;
    const int32_t n = (int32_t)ix->n;
    // K >= the number of positive-similarity persons - the shipped --k-nearest 2000000 - selects them all:
    // the similarities of the stream scan ARE the weights, nothing needs sorting
    LOCREC_TRY(knn_enqueue_dense(ix, qrow, pw, cw));
    int64_t cand = 0;
    LOCREC_TRY(candidate_count(ix, &cand));
    if (k >= cand) return aggregate_places(ix, ix->S1.p, out_places, out_ratings, inout_count);
    LOCREC_TRY(sort_all(ix, qrow, pw, cw, &cand, true));
    const int64_t m = std::min(cand, k);
    LOCREC_TRY(ix->lk_w.reserve((size_t)n));
    LOCREC_HIP_TRY(hipMemsetAsync(ix->lk_w.p, 0, (size_t)n * sizeof(double), s));
    if (m > 0)
        hipLaunchKernelGGL(lk_scatter_weights, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, ix->lk_keys_out.p,
                           ix->lk_vals_out.p, (int32_t)m, ix->row_of_rid.p, ix->lk_w.p);
    return aggregate_places(ix, ix->lk_w.p, out_places, out_ratings, inout_count);
}

}  // namespace locrec

namespace locrec {

// makeRecommendations for the persons at the internal rows rows[0 .. nq) with K >= the number of other persons
// ("every positive-similarity person is a neighbour").  Results stay on the device: ix->lkb_place / lkb_est, rows
// of query i at [lkb_off[i], lkb_off[i + 1]) ordered by place id.  A row that is not a valid query (an empty place
// or category vector: KnnRecommender.scala:77-83 throws for it) gets no rows.
// the presence bitmap of the place table lives behind the table itself (kLkbBitmapBits bits, zero between tiles)
static void lkb_bitmap_of(locrec_knn_index *ix, uint32_t **bits, uint32_t *mask)
{
    uint32_t nbits = 32;
    while (nbits < (uint32_t)std::max(1, ix->fp.dim) && nbits < kLkbBitmapBits) nbits <<= 1;
    *bits = reinterpret_cast<uint32_t *>(ix->lkb_qd_p.p + (size_t)std::max(1, ix->fp.dim) * kLkbQt);
    *mask = nbits - 1u;
}

int32_t knn_large_recommend_batch(locrec_knn_index *ix, const int32_t *rows, int64_t nq, double pw, double cw)
{
    hipStream_t s = ix->stream;
    const int32_t n = (int32_t)ix->n;
    const int32_t np = (int32_t)ix->cplace_ids.size();
    ix->have_lkb = false;
    ix->lkb_off.assign((size_t)nq + 1, 0);
    LOCREC_TRY(ensure_segments(ix));
    const int tiles = std::max(1, (np + kFinishTile - 1) / kFinishTile);
    LOCREC_TRY(ix->lkb_S.reserve((size_t)n * kLkbQt));
    if (!ix->lkb_qd_p.p) {
        LOCREC_TRY(ix->lkb_qd_p.alloc((size_t)std::max(1, ix->fp.dim) * kLkbQt + kLkbBitmapBits / 64));  // (+ the presence bitmap)
        LOCREC_TRY(ix->lkb_qd_c.alloc((size_t)std::max(1, ix->fc.dim) * kLkbQt));
        LOCREC_HIP_TRY(hipMemsetAsync(ix->lkb_qd_p.p, 0, ix->lkb_qd_p.bytes(), s));
        LOCREC_HIP_TRY(hipMemsetAsync(ix->lkb_qd_c.p, 0, ix->lkb_qd_c.bytes(), s));
    }
    LOCREC_TRY(ix->lkb_cand.reserve(kLkbQt));
    LOCREC_TRY(ix->lkb_seg_ws.reserve((size_t)std::max(1, ix->lk_nsegs) * kLkbQt));
    LOCREC_TRY(ix->lkb_seg_ss.reserve((size_t)std::max(1, ix->lk_nsegs) * kLkbQt));
    LOCREC_TRY(ix->lkb_ws.reserve((size_t)std::max(1, np) * kLkbQt));
    LOCREC_TRY(ix->lkb_ss.reserve((size_t)std::max(1, np) * kLkbQt));
    LOCREC_TRY(ix->lkb_tile_cnt.reserve((size_t)tiles * kLkbQt));
    std::vector<int32_t> tc((size_t)tiles * kLkbQt);
    int64_t total = 0;
    for (int64_t q0 = 0; q0 < nq; q0 += kLkbQt) {
        const int nt = (int)std::min<int64_t>(kLkbQt, nq - q0);
        LkbScan P{};
        LkbFill fp{}, fc{};
        int maxp = 1, maxc = 1;
        for (int t = 0; t < kLkbQt; ++t) {
            int32_t r = t < nt ? rows[q0 + t] : -1;
            if (r >= 0 && (ix->fp.nnz[(size_t)r] == 0 || ix->fc.nnz[(size_t)r] == 0)) r = -1;  // not a valid query
            P.qrow[t] = fp.qrow[t] = fc.qrow[t] = r;
            if (r >= 0) {
                maxp = std::max(maxp, ix->fp.nnz[(size_t)r]);
                maxc = std::max(maxc, ix->fc.nnz[(size_t)r]);
            }
        }
        fp.ptr = ix->fp.csr_ptr.p; fp.idx = ix->fp.csr_idx.p; fp.val = ix->fp.csr_val.p; fp.qd = ix->lkb_qd_p.p;
        lkb_bitmap_of(ix, &fp.bits, &fp.bit_mask);
        P.bits = fp.bits;
        P.bit_mask = fp.bit_mask;
        fc.ptr = ix->fc.csr_ptr.p; fc.idx = ix->fc.csr_idx.p; fc.val = ix->fc.csr_val.p; fc.qd = ix->lkb_qd_c.p;
        const dim3 gp((unsigned)((maxp + 255) / 256), kLkbQt), gc((unsigned)((maxc + 255) / 256), kLkbQt);
        fp.set = fc.set = 1;
        hipLaunchKernelGGL(lkb_fill, gp, dim3(256), 0, s, fp);
        hipLaunchKernelGGL(lkb_fill, gc, dim3(256), 0, s, fc);
        P.p_ptr = ix->fp.csr_ptr.p; P.p_idx = ix->fp.csr_idx.p; P.p_val = ix->fp.csr_val.p;
        P.c_ptr = ix->fc.csr_ptr.p; P.c_idx = ix->fc.csr_idx.p; P.c_val = ix->fc.csr_val.p;
        P.norm_p = ix->fp.norm.p; P.norm_c = ix->fc.norm.p;
        P.qd_p = ix->lkb_qd_p.p; P.qd_c = ix->lkb_qd_c.p;
        P.nrows = n;
        P.pw = pw; P.cw = cw;
        P.S = ix->lkb_S.p;
        P.cand = ix->lkb_cand.p;
        LOCREC_HIP_TRY(hipMemsetAsync(ix->lkb_cand.p, 0, kLkbQt * sizeof(int32_t), s));
        LOCREC_LAUNCH_PROFILED(ix->prof, lkb_scan, dim3((unsigned)((n + 255) / 256)), dim3(256), (P.bit_mask + 1u) / 8u, s, P);
        fp.set = fc.set = 0;  // the dense tables go back to all zero for the next tile
        hipLaunchKernelGGL(lkb_fill, gp, dim3(256), 0, s, fp);
        hipLaunchKernelGGL(lkb_fill, gc, dim3(256), 0, s, fc);
        if (np > 0) {
            if (ix->lk_nsegs > 0)
                hipLaunchKernelGGL(lkb_aggregate_segments, dim3((unsigned)((ix->lk_nsegs + 3) / 4)), dim3(256), 0, s,
                                   ix->lk_seg_begin.p, ix->lk_seg_end.p, ix->lk_nsegs, ix->cp_row.p, ix->cp_rating.p,
                                   ix->lkb_S.p, ix->lkb_seg_ws.p, ix->lkb_seg_ss.p);
            hipLaunchKernelGGL(lkb_sum_segments, dim3((unsigned)(((int64_t)np * kLkbQt + 255) / 256)), dim3(256), 0, s,
                               ix->lk_place_seg0.p, np, ix->lkb_seg_ws.p, ix->lkb_seg_ss.p, ix->lkb_ws.p, ix->lkb_ss.p);
            hipLaunchKernelGGL(lkb_finish_count, dim3((unsigned)tiles, kLkbQt), dim3(256), 0, s, ix->lkb_ss.p, np, tiles,
                               ix->lkb_tile_cnt.p);
            LOCREC_HIP_TRY(hipGetLastError());
            LOCREC_HIP_TRY(hipMemcpyAsync(tc.data(), ix->lkb_tile_cnt.p, tc.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            LOCREC_HIP_TRY(hipStreamSynchronize(s));
            LkbBases B{};
            for (int t = 0; t < kLkbQt; ++t) {
                int64_t c = 0;
                if (t < nt && P.qrow[t] >= 0)
                    for (int i = 0; i < tiles; ++i) c += tc[(size_t)t * tiles + i];
                B.base[t] = t < nt ? total : -1;
                if (t < nt) {
                    ix->lkb_off[(size_t)(q0 + t)] = total;
                    total += c;
                }
            }
            // grow-only output with headroom: the rows of the tiles emitted so far must survive a growth
            if ((size_t)total > ix->lkb_place.n) {
                DevBuf<int64_t> np_buf;
                DevBuf<double> ne_buf;
                const size_t want = (size_t)total + (size_t)total / 2 + 1024;
                LOCREC_TRY(np_buf.alloc(want));
                LOCREC_TRY(ne_buf.alloc(want));
                const int64_t keep = ix->lkb_off[(size_t)q0];
                if (keep > 0) {
                    LOCREC_HIP_TRY(hipMemcpyAsync(np_buf.p, ix->lkb_place.p, (size_t)keep * 8, hipMemcpyDeviceToDevice, s));
                    LOCREC_HIP_TRY(hipMemcpyAsync(ne_buf.p, ix->lkb_est.p, (size_t)keep * 8, hipMemcpyDeviceToDevice, s));
                    LOCREC_HIP_TRY(hipStreamSynchronize(s));
                }
                std::swap(ix->lkb_place.p, np_buf.p); std::swap(ix->lkb_place.n, np_buf.n);
                std::swap(ix->lkb_est.p, ne_buf.p); std::swap(ix->lkb_est.n, ne_buf.n);
            }
            hipLaunchKernelGGL(lkb_finish_emit, dim3((unsigned)tiles, kLkbQt), dim3(256), 0, s, ix->lkb_ws.p, ix->lkb_ss.p, np,
                               tiles, ix->lkb_tile_cnt.p, ix->cplace_dev.p, B, ix->lkb_place.p, ix->lkb_est.p);
            LOCREC_HIP_TRY(hipGetLastError());
        } else {
            for (int t = 0; t < nt; ++t) ix->lkb_off[(size_t)(q0 + t)] = total;
        }
    }
    ix->lkb_off[(size_t)nq] = total;
    ix->have_lkb = true;
    return LOCREC_OK;
}

}  // namespace locrec

namespace locrec {

// The similarities of up to kLkbQt (16) queries against every row -> ix->lkb_S[row][16] (dense fp64 query tables, every
// candidate walks its plain CSR row: exact in every stored format and for every row, wide ones included).
int32_t knn_large_scan_tile(locrec_knn_index *ix, const int32_t *rows, int nt, double pw, double cw)
{
    hipStream_t s = ix->stream;
    const int32_t n = (int32_t)ix->n;
    if (nt < 1 || nt > kLkbQt) return fail(LOCREC_E_INVALID_ARG, "a tile holds 1 .. %d queries", kLkbQt);
    LOCREC_TRY(ix->lkb_S.reserve((size_t)n * kLkbQt));
    if (!ix->lkb_qd_p.p) {
        LOCREC_TRY(ix->lkb_qd_p.alloc((size_t)std::max(1, ix->fp.dim) * kLkbQt + kLkbBitmapBits / 64));  // (+ the presence bitmap)
        LOCREC_TRY(ix->lkb_qd_c.alloc((size_t)std::max(1, ix->fc.dim) * kLkbQt));
        LOCREC_HIP_TRY(hipMemsetAsync(ix->lkb_qd_p.p, 0, ix->lkb_qd_p.bytes(), s));
        LOCREC_HIP_TRY(hipMemsetAsync(ix->lkb_qd_c.p, 0, ix->lkb_qd_c.bytes(), s));
    }
    LOCREC_TRY(ix->lkb_cand.reserve(kLkbQt));
    LkbScan P{};
    LkbFill fp{}, fc{};
    int maxp = 1, maxc = 1;
    for (int t = 0; t < kLkbQt; ++t) {
        int32_t r = t < nt ? rows[t] : -1;
        if (r >= 0 && (ix->fp.nnz[(size_t)r] == 0 || ix->fc.nnz[(size_t)r] == 0)) r = -1;
        P.qrow[t] = fp.qrow[t] = fc.qrow[t] = r;
        if (r >= 0) {
            maxp = std::max(maxp, ix->fp.nnz[(size_t)r]);
            maxc = std::max(maxc, ix->fc.nnz[(size_t)r]);
        }
    }
    fp.ptr = ix->fp.csr_ptr.p; fp.idx = ix->fp.csr_idx.p; fp.val = ix->fp.csr_val.p; fp.qd = ix->lkb_qd_p.p;
    lkb_bitmap_of(ix, &fp.bits, &fp.bit_mask);
    P.bits = fp.bits;
    P.bit_mask = fp.bit_mask;
    fc.ptr = ix->fc.csr_ptr.p; fc.idx = ix->fc.csr_idx.p; fc.val = ix->fc.csr_val.p; fc.qd = ix->lkb_qd_c.p;
    const dim3 gp((unsigned)((maxp + 255) / 256), kLkbQt), gc((unsigned)((maxc + 255) / 256), kLkbQt);
    fp.set = fc.set = 1;
    hipLaunchKernelGGL(lkb_fill, gp, dim3(256), 0, s, fp);
    hipLaunchKernelGGL(lkb_fill, gc, dim3(256), 0, s, fc);
    P.p_ptr = ix->fp.csr_ptr.p; P.p_idx = ix->fp.csr_idx.p; P.p_val = ix->fp.csr_val.p;
    P.c_ptr = ix->fc.csr_ptr.p; P.c_idx = ix->fc.csr_idx.p; P.c_val = ix->fc.csr_val.p;
    P.norm_p = ix->fp.norm.p; P.norm_c = ix->fc.norm.p;
    P.qd_p = ix->lkb_qd_p.p; P.qd_c = ix->lkb_qd_c.p;
    P.nrows = n;
    P.pw = pw; P.cw = cw;
    P.S = ix->lkb_S.p;
    P.cand = ix->lkb_cand.p;
    P.transposed = 1;
    LOCREC_HIP_TRY(hipMemsetAsync(ix->lkb_cand.p, 0, kLkbQt * sizeof(int32_t), s));
    hipLaunchKernelGGL(lkb_scan, dim3((unsigned)((n + 255) / 256)), dim3(256), (P.bit_mask + 1u) / 8u, s, P);
    fp.set = fc.set = 0;  // the dense tables go back to all zero
    hipLaunchKernelGGL(lkb_fill, gp, dim3(256), 0, s, fp);
    hipLaunchKernelGGL(lkb_fill, gc, dim3(256), 0, s, fc);
    LOCREC_HIP_TRY(hipGetLastError());
    ix->have_lkb = false;  // (a resident large-K batch shares these workspaces)
    return LOCREC_OK;
}

// the histograms of columns 0 .. nt - 1 of the (transposed) tile into hist[t][kLkHistBins] (zeroed by the caller's last
// knn_select1); *cols = the tile itself, columns ix->n similarities apart
int32_t knn_large_tile_hists(locrec_knn_index *ix, int nt, uint32_t *hist, const double **cols)
{
    hipStream_t s = ix->stream;
    const int32_t n = (int32_t)ix->n;
    *cols = ix->lkb_S.p;
    hipLaunchKernelGGL(lkb_column_hist, dim3((unsigned)std::min(512, (n + 255) / 256), (unsigned)nt), dim3(256), 0, s, ix->lkb_S.p, n,
                       hist);
    LOCREC_HIP_TRY(hipGetLastError());
    return LOCREC_OK;
}

}  // namespace locrec
