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

// ---------------------------------------------------------------------------
// small device helpers

__device__ __forceinline__ uint32_t hash_idx(uint32_t idx, int hlog2)
{
    return (idx * 0x9E3779B1u) >> (32 - hlog2);
}

// GENERIC format: 64-bit hash entries {index, panel row}
__device__ __forceinline__ int panel_slot(const uint2 *hash, int hlog2, int zero_row, uint32_t idx)
{
    const uint32_t mask = (1u << hlog2) - 1u;
    uint32_t h = hash_idx(idx, hlog2);
    for (;;) {
        const uint2 k = hash[h];
        if (k.x == idx) return (int)k.y;
        if (k.x == kEmpty) return zero_row;
        h = (h + 1) & mask;
    }
}

// PACKED formats: 32-bit hash entries, index << 12 | panel row (index < 2^20 - 1, row < 4096), kept
// in BUCKETS of four (16 B): a lookup is one ds_read_b128 plus four compares, branch-free; only a
// full bucket without the key (about 2 % of buckets at the load factor used) sends a lane on to the
// next bucket.  Entries of a bucket fill left to right, so "has room" == last entry empty.
// The hash is one full-rate 24-bit multiply (v_mul_u32_u24), not the quarter-rate v_mul_lo_u32.
constexpr uint32_t kSlotMask = 0xFFFu;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash20(uint32_t idx, int hshift)
{
    // HIP declares __umul24 as returning int: without the cast the shift is arithmetic and half
    // of the keys get a sign-extended, out-of-range first probe position.
    return static_cast<uint32_t>(__umul24(idx, 0x9E3779u)) >> hshift;  // hshift = 32 - log2(buckets)
}

// slot of idx, or -1 if the bucket is full and does not hold it (look in the next bucket).
// entry - (idx << 12) is the panel row (< 4096) exactly for the matching entry and >= 4096 for
// every other one (including the all-ones empty entry), so two v_min3_u32 replace four compares.
__device__ __forceinline__ int bucket_find(const u32x4 k, uint32_t idx, int zero_row)
{
    const uint32_t key12 = idx << 12;
    const uint32_t d0 = k.x - key12, d1 = k.y - key12, d2 = k.z - key12, d3 = k.w - key12;
    const uint32_t d = min(min(d0, d1), min(d2, d3));
    const int miss = k.w == kEmpty ? zero_row : -1;
    return d < 4096u ? (int)d : miss;
}

__device__ __forceinline__ int panel_slot32(const uint32_t *hash, uint32_t bmask, int zero_row, uint32_t idx,
                                            uint32_t b)
{
    for (;;) {
        const int slot = bucket_find(reinterpret_cast<const u32x4 *>(hash)[b], idx, zero_row);
        if (slot >= 0) return slot;
        b = (b + 1) & bmask;
    }
}

// (s desc, rid asc): is a strictly better than b?
__device__ __forceinline__ bool better(double sa, uint32_t ra, double sb, uint32_t rb)
{
    return sa > sb || (sa == sb && ra < rb);
}

// Block-wide bitonic sort of n2 (pow2) entries in LDS, best first.
__device__ void block_sort_desc(double *s, uint32_t *r, int n2)
{
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (n2 >> 1); t += blockDim.x) {
                const int i = 2 * t - (t & (j - 1));
                const int l = i + j;
                const bool first_better = (i & k) == 0;  // this pair: better element first
                const double si = s[i], sl = s[l];
                const uint32_t ri = r[i], rl = r[l];
                const bool l_better = better(sl, rl, si, ri);
                if (l_better == first_better) {
                    s[i] = sl; s[l] = si;
                    r[i] = rl; r[l] = ri;
                }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------
// a1: Distance.vectorLength (Distance.scala:11-16), once per row at create time.
__global__ void knn_norms(const int64_t *ptr, const double *val, int32_t nrows, double *norm, float *inorm32)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    double sum = 0.0;
    for (int64_t e = ptr[r]; e < ptr[r + 1]; ++e) {
        const double sq = val[e] * val[e];
        sum = sum + sq;
    }
    const double len = sqrt(sum);
    norm[r] = len;
    inorm32[r] = len > 0.0 ? (float)(1.0 / len) : 0.0f;
}

// ---------------------------------------------------------------------------
// panel construction (once per block)

template <int QT>
__device__ void build_panel_generic(const Family &f, const int *s_qrow, int nqt, uint2 *hash, double *panel,
                                    int *s_nrows)
{
    const int tid = threadIdx.x;
    const int hcap = f.direct ? 0 : (1 << f.hlog2);
    for (int i = tid; i < f.rows_cap * QT; i += blockDim.x) panel[i] = 0.0;
    for (int i = tid; i < hcap; i += blockDim.x) hash[i] = make_uint2(kEmpty, 0u);
    if (tid == 0) *s_nrows = 0;
    __syncthreads();
    if (!f.direct) {
        const uint32_t mask = (uint32_t)hcap - 1u;
        for (int q = 0; q < nqt; ++q) {
            const int row = s_qrow[q];
            for (int64_t e = f.csr_ptr[row] + tid; e < f.csr_ptr[row + 1]; e += blockDim.x) {
                const uint32_t idx = (uint32_t)f.csr_idx[e];
                uint32_t h = hash_idx(idx, f.hlog2);
                for (;;) {
                    const uint32_t old = atomicCAS(&hash[h].x, kEmpty, idx);
                    if (old == kEmpty || old == idx) break;
                    h = (h + 1) & mask;
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < hcap; i += blockDim.x)
            if (hash[i].x != kEmpty) hash[i].y = (uint32_t)atomicAdd(s_nrows, 1);
        __syncthreads();
    }
    const int zero_row = f.rows_cap - 1;
    for (int q = 0; q < nqt; ++q) {
        const int row = s_qrow[q];
        for (int64_t e = f.csr_ptr[row] + tid; e < f.csr_ptr[row + 1]; e += blockDim.x) {
            const uint32_t idx = (uint32_t)f.csr_idx[e];
            const int slot = f.direct ? (int)idx : panel_slot(hash, f.hlog2, zero_row, idx);
            panel[slot * QT + q] = f.csr_val[e];
        }
    }
    __syncthreads();
}

template <int QT, class PanelT>
__device__ void build_panel_packed(const Family &f, const int *s_qrow, int nqt, uint32_t *hash, PanelT *panel,
                                   int *s_nrows, unsigned short *pop = nullptr)
{
    const int tid = threadIdx.x;
    const int nbuckets = f.direct ? 0 : (1 << f.hlog2);
    const int hcap = nbuckets * 4;
    const uint32_t bmask = (uint32_t)nbuckets - 1u;
    const int hshift = 32 - f.hlog2;
    for (int i = tid; i < f.rows_cap * QT; i += blockDim.x) panel[i] = PanelT(0);
    for (int i = tid; i < hcap; i += blockDim.x) hash[i] = kEmpty;
    if (tid == 0) *s_nrows = 0;
    __syncthreads();
    if (!f.direct) {
        for (int q = 0; q < nqt; ++q) {
            const int row = s_qrow[q];
            for (int64_t e = f.csr_ptr[row] + tid; e < f.csr_ptr[row + 1]; e += blockDim.x) {
                const uint32_t idx = (uint32_t)f.csr_idx[e];
                uint32_t b = hash20(idx, hshift);
                for (bool placed = false; !placed;) {
                    for (int j = 0; j < 4 && !placed; ++j) {
                        const uint32_t old = atomicCAS(&hash[b * 4 + j], kEmpty, (idx << 12) | kSlotMask);
                        placed = old == kEmpty || (old >> 12) == idx;
                    }
                    b = (b + 1) & bmask;
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < hcap; i += blockDim.x)
            if (hash[i] != kEmpty) hash[i] = (hash[i] & ~kSlotMask) | (uint32_t)atomicAdd(s_nrows, 1);
        if (pop)
            for (int i = tid; i < f.pop_h; i += blockDim.x) pop[i] = (unsigned short)(f.rows_cap - 1);  // all-zero row
        __syncthreads();
        if (pop)  // popular indices of the tile: slot straight from the index, no hash
            for (int i = tid; i < hcap; i += blockDim.x) {
                const uint32_t e = hash[i];
                if (e != kEmpty && (e >> 12) < (uint32_t)f.pop_h) pop[e >> 12] = (unsigned short)(e & kSlotMask);
            }
        __syncthreads();
    }
    const int zero_row = f.rows_cap - 1;
    for (int q = 0; q < nqt; ++q) {
        const int row = s_qrow[q];
        for (int64_t e = f.csr_ptr[row] + tid; e < f.csr_ptr[row + 1]; e += blockDim.x) {
            const uint32_t idx = (uint32_t)f.csr_idx[e];
            const int slot = f.direct ? (int)idx : panel_slot32(hash, bmask, zero_row, idx, hash20(idx, hshift));
            panel[slot * QT + q] = PanelT(f.csr_val[e]);
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// PACKED: one family's dot products of this lane's candidate row against the tile.
// The row is walked in groups of four dwordx4 loads with the next group already in
// flight (the compiler's counted vmcnt keeps them outstanding across the work).
//   MODE 1 (PACK32): u32 panel, u32 accumulators, one v_mul_u32_u24 (+ half a v_add3) per pair.
//   MODE 2 (PACK16): u16 panel, packed u16 accumulators, one v_pk_mad_u16 per TWO pairs; legal
//                    when every possible dot is < 65536 (max over rows of sum v^2 < 65536,
//                    Cauchy-Schwarz), decided at create time.

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

struct Group4 {
    u32x4 a0, a1, a2, a3;
};

// w4 and j are wave-uniform, so these are scalar branches around whole dwordx4 loads
__device__ __forceinline__ Group4 load_group(const u32x4 *lane_base, int j, int w4)
{
    Group4 g;
    g.a0 = g.a1 = g.a2 = g.a3 = u32x4{0u, 0u, 0u, 0u};
    if (j + 0 < w4) g.a0 = lane_base[(int64_t)(j + 0) * 64];
    if (j + 1 < w4) g.a1 = lane_base[(int64_t)(j + 1) * 64];
    if (j + 2 < w4) g.a2 = lane_base[(int64_t)(j + 2) * 64];
    if (j + 3 < w4) g.a3 = lane_base[(int64_t)(j + 3) * 64];
    return g;
}

// the few per-family scalars the inner loop needs, copied out of the parameter block once
struct HotFam {
    const uint32_t *hash;
    const unsigned char *panel;
    int vbits;
    uint32_t vmask;
    int hshift;
    uint32_t hmask;
    int zero_row;
    int direct;
    const unsigned short *pop;  // direct slot table of the popular indices, or nullptr
};

__device__ __forceinline__ HotFam make_hot(const Family &f, unsigned char *smem)
{
    HotFam h;
    h.hash = reinterpret_cast<const uint32_t *>(smem + f.off_hash);
    h.panel = smem + f.off_panel;
    h.vbits = f.vbits;
    h.vmask = (1u << f.vbits) - 1u;
    h.hshift = 32 - f.hlog2;
    h.hmask = (1u << f.hlog2) - 1u;
    h.zero_row = f.rows_cap - 1;
    h.direct = f.direct;
    h.pop = (f.pop_h > 0 && !f.direct) ? reinterpret_cast<const unsigned short *>(smem + f.off_pop) : nullptr;
    return h;
}

template <int MODE, int QT>
struct Acc;
template <int QT>
struct Acc<1, QT> {
    uint32_t a[QT];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int q = 0; q < QT; ++q) a[q] = 0u;
    }
    __device__ __forceinline__ uint32_t get(int q) const { return a[q]; }
};
template <int QT>
struct Acc<2, QT> {
    u16x2 a[QT / 2];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int q = 0; q < QT / 2; ++q) a[q] = u16x2{0, 0};
    }
    __device__ __forceinline__ uint32_t get(int q) const { return (q & 1) ? a[q >> 1].y : a[q >> 1].x; }
};
template <int QT>
struct Acc<3, QT> : Acc<2, QT> {};  // head / tail form: the same packed u16 accumulators

template <bool POP>
__device__ __forceinline__ void slots4(const u32x4 e4, const HotFam &f, int (&slot)[4])
{
    const uint32_t ee[4] = {e4.x, e4.y, e4.z, e4.w};
    if constexpr (POP) {
        // every index of this element group is < pop_h in every lane (sell_split): one 2-byte LDS read
#pragma unroll
        for (int t = 0; t < 4; ++t) slot[t] = f.pop[ee[t] >> f.vbits];
    } else if (f.direct) {
#pragma unroll
        for (int t = 0; t < 4; ++t) slot[t] = (int)(ee[t] >> f.vbits);
    } else {
        const u32x4 *buckets = reinterpret_cast<const u32x4 *>(f.hash);
        uint32_t idx[4], b[4];
        u32x4 k[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            idx[t] = ee[t] >> f.vbits;
            b[t] = hash20(idx[t], f.hshift);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) k[t] = buckets[b[t]];  // four independent ds_read_b128 in flight
        bool walk = false;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            slot[t] = bucket_find(k[t], idx[t], f.zero_row);
            walk |= slot[t] < 0;
        }
        if (walk) {  // a full bucket without the key: rare, look in the following buckets
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (slot[t] < 0) slot[t] = panel_slot32(f.hash, f.hmask, f.zero_row, idx[t], (b[t] + 1) & f.hmask);
        }
    }
}

// pop (wave-uniform): every index of this element group is popular in every lane
template <int MODE, int QT>
__device__ __forceinline__ void accum4(const u32x4 e4, const HotFam &f, Acc<MODE, QT> &acc, bool pop)
{
    const uint32_t ee[4] = {e4.x, e4.y, e4.z, e4.w};
    int slot[4];
    if (pop)
        slots4<true>(e4, f, slot);
    else
        slots4<false>(e4, f, slot);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t v = ee[t] & f.vmask;
        if constexpr (MODE == 1) {
            if constexpr (QT >= 4) {
                const u32x4 *r = reinterpret_cast<const u32x4 *>(f.panel) + slot[t] * (QT / 4);
#pragma unroll
                for (int i = 0; i < QT / 4; ++i) {
                    const u32x4 pv = r[i];
                    acc.a[4 * i + 0] += __umul24(v, pv.x);
                    acc.a[4 * i + 1] += __umul24(v, pv.y);
                    acc.a[4 * i + 2] += __umul24(v, pv.z);
                    acc.a[4 * i + 3] += __umul24(v, pv.w);
                }
            } else {
                const uint32_t *r = reinterpret_cast<const uint32_t *>(f.panel) + slot[t] * QT;
#pragma unroll
                for (int q = 0; q < QT; ++q) acc.a[q] += __umul24(v, r[q]);
            }
        } else {
            static_assert(MODE != 2 || QT % 8 == 0, "PACK16 tiles are multiples of 8 queries");
            const u16x2 vv = {(unsigned short)v, (unsigned short)v};
            const u32x4 *r = reinterpret_cast<const u32x4 *>(f.panel) + slot[t] * (QT / 8);
#pragma unroll
            for (int i = 0; i < QT / 8; ++i) {
                const u32x4 pv = r[i];
                // (bit_cast straight from pv.y silently reads element 0 with this clang: go through scalars)
                const uint32_t w0 = pv.x, w1 = pv.y, w2 = pv.z, w3 = pv.w;
#ifdef LOCREC_NO_PKMAD
                const uint32_t ww[4] = {w0, w1, w2, w3};
#pragma unroll
                for (int z = 0; z < 4; ++z) {
                    u16x2 &A = acc.a[4 * i + z];
                    A.x = (unsigned short)(A.x + (ww[z] & 0xFFFFu) * v);
                    A.y = (unsigned short)(A.y + (ww[z] >> 16) * v);
                }
#else
                acc.a[4 * i + 0] = acc.a[4 * i + 0] + __builtin_bit_cast(u16x2, w0) * vv;
                acc.a[4 * i + 1] = acc.a[4 * i + 1] + __builtin_bit_cast(u16x2, w1) * vv;
                acc.a[4 * i + 2] = acc.a[4 * i + 2] + __builtin_bit_cast(u16x2, w2) * vv;
                acc.a[4 * i + 3] = acc.a[4 * i + 3] + __builtin_bit_cast(u16x2, w3) * vv;
#endif
            }
        }
    }
}

// split4 (wave-uniform): element groups [0, split4) hold popular indices only, in every lane
template <int MODE, int QT>
__device__ __forceinline__ void family_dots_packed(const HotFam &f, const u32x4 *lane_base, int w4, Group4 cur,
                                                   Acc<MODE, QT> &acc, int split4 = 0)
{
    for (int j = 0; j < w4; j += 4) {
        const Group4 nxt = load_group(lane_base, j + 4, w4);
        accum4<MODE, QT>(cur.a0, f, acc, j < split4);
        if (j + 1 < w4) accum4<MODE, QT>(cur.a1, f, acc, j + 1 < split4);
        if (j + 2 < w4) accum4<MODE, QT>(cur.a2, f, acc, j + 2 < split4);
        if (j + 3 < w4) accum4<MODE, QT>(cur.a3, f, acc, j + 3 < split4);
        cur = nxt;
    }
}

template <int QT>
__device__ __forceinline__ void dots_generic(const Family &f, const uint2 *hash, const double *panel,
                                             int slice, int lane, double (&acc)[QT])
{
    const int64_t off = f.sell_off[slice];
    const int32_t *ib = reinterpret_cast<const int32_t *>(f.sell) + off + lane;
    const double *vb = f.sell_val + off + lane;
    const int w = f.sell_w[slice];
    const int zero_row = f.rows_cap - 1;
    const bool direct = f.direct != 0;
    for (int j = 0; j < w; ++j) {
        const uint32_t idx = (uint32_t)ib[(int64_t)j * 64];
        const double v = vb[(int64_t)j * 64];
        const int slot = direct ? (int)idx : panel_slot(hash, f.hlog2, zero_row, idx);
        const double *r = panel + slot * QT;
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            const double prod = v * r[q];  // sum += x(kx) * y(ky), ascending index order
            acc[q] = acc[q] + prod;
        }
    }
}

// a2 + a3 + a4 for one (candidate, query): ps = dot/(|c|*|q|) (Distance.scala:8: one multiply,
// one divide), keep "> 0" (KnnRecommender.scala:91), ps*pw + cs*cw (:43-45).  Returns whether the
// candidate appears in the outer join at all.
template <class AccT>
__device__ __forceinline__ bool exact_similarity(AccT dp, AccT dc, double cnp, double cnc, double qnp, double qnc,
                                                 double pw, double cw, double &s)
{
    double ps = 0.0, cs = 0.0;
    bool have = false;
    if (cnp > 0.0) {  // present in the place frame
        const double den = cnp * qnp;
        const double t = (double)dp / den;
        if (t > 0) { ps = t; have = true; }
    }
    if (cnc > 0.0) {
        const double den = cnc * qnc;
        const double t = (double)dc / den;
        if (t > 0) { cs = t; have = true; }
    }
    const double a = ps * pw;
    const double b = cs * cw;
    s = a + b;
    return have;
}

// ---------------------------------------------------------------------------
// per-query LDS top-K list: compaction of query q's list to its best K

// ntau (knn_scan_ht): a negated copy of tau32 - the accumulator operand of its packed bound (v_dot2c_f32_f16)
__device__ void compact_query(double *cs, uint32_t *cr, int *cnt, double *tau_s, uint32_t *tau_r, float *tau32,
                              int q, int S, int K, float *ntau = nullptr)
{
    double *s = cs + q * S;
    uint32_t *r = cr + q * S;
    const int n = min(cnt[q], S);
    for (int i = n + threadIdx.x; i < S; i += blockDim.x) {
        s[i] = -1.0;
        r[i] = 0xFFFFFFFFu;
    }
    __syncthreads();
    block_sort_desc(s, r, S);
    if (threadIdx.x == 0) {
        const int m = min(n, K);
        cnt[q] = m;
        if (m >= K) {
            tau_s[q] = s[K - 1];
            tau_r[q] = r[K - 1];
            tau32[q] = fmaxf((float)s[K - 1] / 1.0001f, 1.17549435e-38f);
            if (ntau) ntau[q] = -tau32[q];
        }
    }
    __syncthreads();
}

constexpr int kPopTable = cfg::kPopTable;  // indices below this (after the popularity renumbering) get a direct u16 slot table
constexpr int kQueueCap = 96;   // entries per wave queue (LDS: at 192 a block no longer shares the CU with a second one: 18.6 -> 27.4 ms)
// slices between block-wide drains of the queues in the barrier-free mode, and the entry threshold:
// the mode is entered after kCalmIters consecutive iterations in which at most kEnterFastThreads
// threads of an 8-wave block held a survivor.  Measured on cfg2 (ms per 16,384-query batch; replayed
// intervals per 4 batches): 2/16 48.5, 4/16 45.5 (0), 8/16 44.0 (0), 8/32 43.2 (7), 16/32 42.6 (51),
// 16/64 43.1 (1473), 32/32 44.4 (2513), 64/64 50.3 (10378).  LOCREC_KNN_FLUSH / LOCREC_KNN_ENTER override.
constexpr int kFlushEvery = 16;
constexpr int kEnterFastThreads = 32;
constexpr int kCalmIters = 2;

// One synchronous insertion round set for at most one candidate per thread (s, rid for query q;
// have = this thread holds one): places it into the query's list, compacting full lists.
__device__ void insert_sync(bool have, double s, uint32_t rid, int q, double *cand_s, uint32_t *cand_r, int *cnt,
                            double *tau_s, uint32_t *tau_r, float *tau32, int nqt, int S, int K, float *ntau = nullptr)
{
    bool pend = have && better(s, rid, tau_s[q], tau_r[q]);
    while (__syncthreads_or(pend)) {
        if (pend) {
            if (!better(s, rid, tau_s[q], tau_r[q])) {  // the list tightened meanwhile
                pend = false;
            } else {
                const int pos = atomicAdd(&cnt[q], 1);
                if (pos < S) {
                    cand_s[q * S + pos] = s;
                    cand_r[q * S + pos] = rid;
                    pend = false;
                }
            }
        }
        __syncthreads();
        for (int qq = 0; qq < nqt; ++qq)
            if (cnt[qq] >= S) compact_query(cand_s, cand_r, cnt, tau_s, tau_r, tau32, qq, S, K, ntau);
    }
}

#include "knn_ht.h"

// MODE 0 = GENERIC (fp64 values), 1 = PACK32, 2 = PACK16; W = waves per block (all share the tile).
// second launch bound = waves per SIMD: an 8-wave block must fit twice per CU (<= 128 VGPRs)
template <int MODE, int QT, int W>
__global__ __launch_bounds__(W * 64, W >= 8 ? (QT >= 32 && W == 8 ? 2 : 4) : 1) void knn_scan(const ScanParams P)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool PACKED = MODE != 0;
    double *cand_s = reinterpret_cast<double *>(smem + P.off_cand_s);
    uint32_t *cand_r = reinterpret_cast<uint32_t *>(smem + P.off_cand_rid);
    // misc block: doubles first (alignment)
    double *s_qnp = reinterpret_cast<double *>(smem + P.off_misc);
    double *s_qnc = s_qnp + QT;
    double *tau_s = s_qnc + QT;
    uint32_t *tau_r = reinterpret_cast<uint32_t *>(tau_s + QT);
    int *s_qrow = reinterpret_cast<int *>(tau_r + QT);
    int *cnt = s_qrow + QT;
    float *s_qfp = reinterpret_cast<float *>(cnt + QT);  // pw / |q_place|   (prefilter)
    float *s_qfc = s_qfp + QT;                           // cw / |q_category|
    float *tau32 = s_qfc + QT;
    int *s_nrows = reinterpret_cast<int *>(tau32 + QT);
    int *s_flags = s_nrows + 1;  // [0] overflow seen
    // per-wave survivor queues of the fast path
    double *wq_s = reinterpret_cast<double *>(smem + P.off_queue);
    uint32_t *wq_r = reinterpret_cast<uint32_t *>(wq_s + W * kQueueCap);
    uint32_t *wq_q = wq_r + W * kQueueCap;
    int *wq_cnt = reinterpret_cast<int *>(wq_q + W * kQueueCap);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q0 = blockIdx.y * QT;
    const int nqt = min(QT, P.nq - q0);
    const int K = P.K, S = P.S;
    const double pw = P.pw, cw = P.cw;

    if (P.poison & 1) {
        for (int i = tid; i < P.lds_bytes / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(smem)[i] = 0xA5A5A5A5u;
        __syncthreads();
    }
    if (tid < QT) {
        int row = -1;
        if (tid < nqt) row = P.qrows ? P.qrows[q0 + tid] : P.qrow0 + q0 + tid;
        s_qrow[tid] = row;
        const double np_ = row >= 0 ? P.fp.norm[row] : 0.0;
        const double nc_ = row >= 0 ? P.fc.norm[row] : 0.0;
        s_qnp[tid] = np_;
        s_qnc[tid] = nc_;
        s_qfp[tid] = np_ > 0.0 ? (float)(pw / np_) : 0.0f;
        s_qfc[tid] = nc_ > 0.0 ? (float)(cw / nc_) : 0.0f;
        tau_s[tid] = 0.0;  // every candidate has s > 0, so (0, 0) admits them all
        tau_r[tid] = 0u;
        tau32[tid] = 1.17549435e-38f;  // prefilter threshold / 1.0001, floored at FLT_MIN
        cnt[tid] = 0;
    }
    if (tid < W) wq_cnt[tid] = 0;
    if (tid == 0) {
        s_flags[0] = 0;
        s_flags[1] = 0;
        s_flags[2] = 0;
    }
    __syncthreads();
    // MODE 3 (knn_ht.h): category panel at LDS offset 0, place head panel right behind it - both at
    // compile-time offsets, so that an element's low half IS the ds_read address
    constexpr int kHtCatBytes = kHtCatRows * QT * 2;
    if constexpr (MODE == 3) {
        uint32_t *t32 = reinterpret_cast<uint32_t *>(smem + P.ht.off_tail);
        for (int i = tid; i < W * 64 * QT / 2; i += blockDim.x) t32[i] = 0u;  // (synchronised by the panel builds)
        ht_build_panel<QT>(P.fc, P.ht.c_rows, kHtCatRows, s_qrow, nqt, reinterpret_cast<unsigned short *>(smem));
        ht_build_panel<QT>(P.fp, P.ht.h, cfg::ht_plane_rows(P.ht.h), s_qrow, nqt, reinterpret_cast<unsigned short *>(smem + kHtCatBytes));
    } else if constexpr (MODE == 1) {
        build_panel_packed<QT, uint32_t>(P.fp, s_qrow, nqt, reinterpret_cast<uint32_t *>(smem + P.fp.off_hash),
                                         reinterpret_cast<uint32_t *>(smem + P.fp.off_panel), s_nrows, P.fp.pop_h > 0 && !P.fp.direct ? reinterpret_cast<unsigned short *>(smem + P.fp.off_pop) : nullptr);
        build_panel_packed<QT, uint32_t>(P.fc, s_qrow, nqt, reinterpret_cast<uint32_t *>(smem + P.fc.off_hash),
                                         reinterpret_cast<uint32_t *>(smem + P.fc.off_panel), s_nrows, P.fc.pop_h > 0 && !P.fc.direct ? reinterpret_cast<unsigned short *>(smem + P.fc.off_pop) : nullptr);
    } else if constexpr (MODE == 2) {
        build_panel_packed<QT, uint16_t>(P.fp, s_qrow, nqt, reinterpret_cast<uint32_t *>(smem + P.fp.off_hash),
                                         reinterpret_cast<uint16_t *>(smem + P.fp.off_panel), s_nrows, P.fp.pop_h > 0 && !P.fp.direct ? reinterpret_cast<unsigned short *>(smem + P.fp.off_pop) : nullptr);
        build_panel_packed<QT, uint16_t>(P.fc, s_qrow, nqt, reinterpret_cast<uint32_t *>(smem + P.fc.off_hash),
                                         reinterpret_cast<uint16_t *>(smem + P.fc.off_panel), s_nrows, P.fc.pop_h > 0 && !P.fc.direct ? reinterpret_cast<unsigned short *>(smem + P.fc.off_pop) : nullptr);
    } else {
        build_panel_generic<QT>(P.fp, s_qrow, nqt, reinterpret_cast<uint2 *>(smem + P.fp.off_hash),
                                reinterpret_cast<double *>(smem + P.fp.off_panel), s_nrows);
        build_panel_generic<QT>(P.fc, s_qrow, nqt, reinterpret_cast<uint2 *>(smem + P.fc.off_hash),
                                reinterpret_cast<double *>(smem + P.fc.off_panel), s_nrows);
    }

    const int slice_begin = P.slice0 + blockIdx.x * P.slices_per_chunk;
    const int slice_end = min(slice_begin + P.slices_per_chunk, P.nslices);
    const int iters = (P.slices_per_chunk + W - 1) / W;

    // Insertion mode (block-uniform).  Survivors are inserted synchronously, slice by slice, until
    // every query has a full list AND the block has seen few survivors for kCalmIters iterations
    // in a row; then they go to per-wave LDS queues drained every kFlushEvery slices.  A queue found
    // more than half full at a drain sends the block back to synchronous insertion.
    bool fastmode = false;
    int calm = 0;
    // MODE 3: hits of this tile (knn_ht.h), software-pipelined: while slice s is processed the first 64
    // hits of slice s + W are in flight and the offsets of slice s + 2W are being fetched
    const uint32_t *ht_off_row = nullptr, *ht_hits = nullptr;
    unsigned char *my_tail = nullptr;
    int ht_primed = -1;                          // slice the pipeline registers below are valid for
    uint32_t ht_c0 = 0, ht_c1 = 0, ht_hcur = 0;  // current slice: hit range and its first 64 hits
    uint32_t ht_n0 = 0, ht_n1 = 0;               // next slice (s + W): hit range
    if constexpr (MODE == 3) {
        ht_off_row = P.ht.off + (int64_t)blockIdx.y * P.ht.off_stride - P.slice0;
        ht_hits = P.ht.hits + P.ht.tile_base[blockIdx.y];
        my_tail = smem + P.ht.off_tail + wave * (64 * QT * 2);
    }
    for (int it = 0; it < iters; ++it) {
        const int slice = slice_begin + it * W + wave;  // wave-uniform
        const bool live = slice < slice_end;
        const int row = slice * 64 + lane;
        const bool valid = live && row < P.nrows;
        unsigned pend = 0;
        if constexpr (PACKED) {
            Acc<MODE, QT> accp, accc;
            accp.zero();
            accc.zero();
            float icnp = 0.0f, icnc = 0.0f;
            double cnp = 0.0, cnc = 0.0;
            uint32_t myrid = 0u;
            if constexpr (MODE == 3) {
                if (live) {
                    const u32x4 *bp = reinterpret_cast<const u32x4 *>(P.fp.sell + P.fp.sell_off[slice]) + lane;
                    const u32x4 *bc = reinterpret_cast<const u32x4 *>(P.fc.sell + P.fc.sell_off[slice]) + lane;
                    const int w4p = __builtin_amdgcn_readfirstlane(P.fp.sell_w[slice] >> 2);
                    const int w4c = __builtin_amdgcn_readfirstlane(P.fc.sell_w[slice] >> 2);
                    const Group4 gp = load_group(bp, 0, w4p);
                    const Group4 gc = load_group(bc, 0, w4c);
                    if (valid) {
                        icnp = P.fp.inorm32[row];
                        icnc = P.fc.inorm32[row];
                        cnp = P.fp.norm[row];
                        cnc = P.fc.norm[row];
                        myrid = P.rid[row];
                    }
                    // --- hit pipeline (see the declarations in front of the loop)
                    const int nslice = slice + W;
                    const bool have_next = it + 1 < iters && nslice < slice_end;
                    if (ht_primed != slice) {  // first iteration of the block, or an interval is being replayed
                        ht_c0 = __builtin_amdgcn_readfirstlane(ht_off_row[slice]);
                        ht_c1 = __builtin_amdgcn_readfirstlane(ht_off_row[slice + 1]);
                        ht_hcur = (uint32_t)lane < ht_c1 - ht_c0 ? ht_hits[ht_c0 + lane] : 0u;
                        if (have_next) {
                            ht_n0 = __builtin_amdgcn_readfirstlane(ht_off_row[nslice]);
                            ht_n1 = __builtin_amdgcn_readfirstlane(ht_off_row[nslice + 1]);
                        }
                    }
                    const uint32_t hn = ht_c1 - ht_c0;  // hits of this slice and tile (wave-uniform)
                    const uint32_t hbase = ht_c0;
                    const uint32_t hfirst = ht_hcur;
                    uint32_t hnext = 0u, m0 = 0u, m1 = 0u;
                    if (have_next) {
                        hnext = (uint32_t)lane < ht_n1 - ht_n0 ? ht_hits[ht_n0 + lane] : 0u;
                        if (it + 2 < iters && nslice + W < slice_end) {  // unwaited here: read in the next iteration
                            m0 = ht_off_row[nslice + W];
                            m1 = ht_off_row[nslice + W + 1];
                        }
                    }
                    uint32_t ap[QT / 2], ac[QT / 2];
#pragma unroll
                    for (int i = 0; i < QT / 2; ++i) ap[i] = ac[i] = 0u;
                    ht_family_dots<QT>(smem + kHtCatBytes, cfg::ht_plane_rows(P.ht.h) * 16, bp, w4p, gp, ap);
                    ht_family_dots<QT>(smem, kHtCatRows * 16, bc, w4c, gc, ac);
                    if (hn > 0) {
                        // the tail: this slice's hits go into the wave's private accumulator (a wave's LDS
                        // operations execute in order), each lane folds its own row into its head dots
                        // and the words that were touched are cleared again
                        uint32_t hh = hfirst, waddr = 0u;
                        for (uint32_t done = 0;;) {
                            const uint32_t nb = min(64u, hn - done);
                            if ((uint32_t)lane < nb) {
                                const uint32_t q = (hh >> 16) & 31u;
                                waddr = ht_tail_word<QT>(hh >> 21, q >> 1);
                                atomicAdd(reinterpret_cast<uint32_t *>(my_tail + waddr), (hh & 0xFFFFu) << ((q & 1u) * 16u));
                            }
                            done += nb;
                            if (done >= hn) break;
                            hh = (uint32_t)lane < hn - done ? ht_hits[hbase + done + lane] : 0u;  // > 64 hits: rare
                        }
                        constexpr uint32_t chunks = QT / 8;
                        const uint32_t sw = chunks == 2 ? (((uint32_t)lane >> 3) & 1u) : (((uint32_t)lane >> 2) & 3u);
                        u32x4 trow[chunks];
#pragma unroll
                        for (uint32_t c = 0; c < chunks; ++c)
                            trow[c] = *reinterpret_cast<const u32x4 *>(my_tail + lane * (QT * 2) + (((c ^ sw) & (chunks - 1)) << 4));
                        if (hn <= 64u) {
                            if ((uint32_t)lane < hn) *reinterpret_cast<uint32_t *>(my_tail + waddr) = 0u;
                        } else {
#pragma unroll
                            for (uint32_t c = 0; c < chunks; ++c)
                                *reinterpret_cast<u32x4 *>(my_tail + lane * (QT * 2) + (c << 4)) = u32x4{0u, 0u, 0u, 0u};
                        }
#pragma unroll
                        for (uint32_t c = 0; c < chunks; ++c) {
                            const uint32_t tw[4] = {trow[c].x, trow[c].y, trow[c].z, trow[c].w};
#pragma unroll
                            for (int z = 0; z < 4; ++z)
                                ap[4 * c + z] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, ap[4 * c + z]) +
                                                                                 __builtin_bit_cast(u16x2, tw[z]));
                        }
                    }
#pragma unroll
                    for (int i = 0; i < QT / 2; ++i) {
                        accp.a[i] = __builtin_bit_cast(u16x2, ap[i]);
                        accc.a[i] = __builtin_bit_cast(u16x2, ac[i]);
                    }
                    // shift the pipeline
                    ht_c0 = ht_n0;
                    ht_c1 = ht_n1;
                    ht_hcur = hnext;
                    ht_n0 = __builtin_amdgcn_readfirstlane(m0);
                    ht_n1 = __builtin_amdgcn_readfirstlane(m1);
                    ht_primed = have_next ? nslice : -1;
                }
            } else if (live) {
                const HotFam hp = make_hot(P.fp, smem);
                const HotFam hc = make_hot(P.fc, smem);
                const u32x4 *bp = reinterpret_cast<const u32x4 *>(P.fp.sell + P.fp.sell_off[slice]) + lane;
                const u32x4 *bc = reinterpret_cast<const u32x4 *>(P.fc.sell + P.fc.sell_off[slice]) + lane;
                const int w4p = __builtin_amdgcn_readfirstlane(P.fp.sell_w[slice] >> 2);
                const int w4c = __builtin_amdgcn_readfirstlane(P.fc.sell_w[slice] >> 2);
                // both families' first groups and the row scalars go out before any use
                const Group4 gp = load_group(bp, 0, w4p);
                const Group4 gc = load_group(bc, 0, w4c);
                if (valid) {  // all of the row's scalars now: a load issued in the epilogue would stall the wave
                    icnp = P.fp.inorm32[row];
                    icnc = P.fc.inorm32[row];
                    cnp = P.fp.norm[row];
                    cnc = P.fc.norm[row];
                    myrid = P.rid[row];
                }
                const int sp4 = P.fp.sell_split ? __builtin_amdgcn_readfirstlane(P.fp.sell_split[slice]) : 0;
                if constexpr (MODE != 3) {
                    family_dots_packed<MODE, QT>(hp, bp, w4p, gp, accp, sp4);
                    family_dots_packed<MODE, QT>(hc, bc, w4c, gc, accc);
                }
            }
            // f32 upper-bound prefilter: only pairs that can still enter the query's list pay for
            // the fp64 divide.  Relative error of s32 < 1e-6; the 1e-4 margin makes it one-sided.
            unsigned maybe = 0;
            if constexpr (MODE == 3) {
                // the same f32 bound, but every query is tested and - rarely - resolved on the spot: one
                // v_cmp and a scalar branch per pair instead of building a per-lane bit mask
                float fq[QT], gq[QT], tq[QT];
#pragma unroll
                for (int i = 0; i < QT / 4; ++i) {
                    const float4 a = reinterpret_cast<const float4 *>(s_qfp)[i];
                    const float4 b = reinterpret_cast<const float4 *>(s_qfc)[i];
                    const float4 c = reinterpret_cast<const float4 *>(tau32)[i];
                    fq[4 * i] = a.x; fq[4 * i + 1] = a.y; fq[4 * i + 2] = a.z; fq[4 * i + 3] = a.w;
                    gq[4 * i] = b.x; gq[4 * i + 1] = b.y; gq[4 * i + 2] = b.z; gq[4 * i + 3] = b.w;
                    tq[4 * i] = c.x; tq[4 * i + 1] = c.y; tq[4 * i + 2] = c.z; tq[4 * i + 3] = c.w;
                }
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    const float sp = ((float)accp.get(q) * icnp) * fq[q];
                    const float s32 = __builtin_fmaf((float)accc.get(q) * icnc, gq[q], sp);
                    if (s32 >= tq[q]) {  // tq = threshold / 1.0001, never below FLT_MIN: s32 == 0 fails
                        if (q < nqt && row != s_qrow[q]) {  // person_id =!= personId (:89)
                            double s;
                            if (exact_similarity(accp.get(q), accc.get(q), cnp, cnc, s_qnp[q], s_qnc[q], pw, cw, s) &&
                                better(s, myrid, tau_s[q], tau_r[q]))
                                pend |= 1u << q;
                        }
                    }
                }
            } else if (!(P.poison & 4)) {  // (bit 4: timing experiment without the epilogue; results are wrong)
                // per-query constants come out of LDS in wide reads, all before the arithmetic, and
                // the mask is built without branches (16 dependent LDS round trips otherwise)
                float fq[QT], gq[QT], tq[QT];
                if constexpr (QT >= 4) {
#pragma unroll
                    for (int i = 0; i < QT / 4; ++i) {
                        const float4 a = reinterpret_cast<const float4 *>(s_qfp)[i];
                        const float4 b = reinterpret_cast<const float4 *>(s_qfc)[i];
                        const float4 c = reinterpret_cast<const float4 *>(tau32)[i];
                        fq[4 * i] = a.x; fq[4 * i + 1] = a.y; fq[4 * i + 2] = a.z; fq[4 * i + 3] = a.w;
                        gq[4 * i] = b.x; gq[4 * i + 1] = b.y; gq[4 * i + 2] = b.z; gq[4 * i + 3] = b.w;
                        tq[4 * i] = c.x; tq[4 * i + 1] = c.y; tq[4 * i + 2] = c.z; tq[4 * i + 3] = c.w;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        fq[q] = s_qfp[q];
                        gq[q] = s_qfc[q];
                        tq[q] = tau32[q];
                    }
                }
                const bool force = (P.poison & 2) != 0;
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    const float sp = ((float)accp.get(q) * icnp) * fq[q];
                    const float s32 = __builtin_fmaf((float)accc.get(q) * icnc, gq[q], sp);
                    // tq = threshold / 1.0001, never below FLT_MIN: s32 == 0 (no overlap at all) fails
                    const bool pass = (s32 >= tq[q]) | (force & (s32 > 0.0f));
                    maybe |= pass ? (1u << q) : 0u;
                }
            } else {
#pragma unroll
                for (int q = 0; q < QT; ++q) asm volatile("" ::"v"(accp.get(q)), "v"(accc.get(q)));
            }
            if (maybe) {
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    if ((maybe & (1u << q)) && q < nqt && row != s_qrow[q]) {  // person_id =!= personId (:89)
                        double s;
                        if (exact_similarity(accp.get(q), accc.get(q), cnp, cnc, s_qnp[q], s_qnc[q], pw, cw, s) &&
                            better(s, myrid, tau_s[q], tau_r[q]))
                            pend |= 1u << q;
                    }
                }
            }
            // Survivors (see "Insertion mode" above).
            if (!fastmode) {
                int np = __syncthreads_count(pend != 0);
                if (P.fast) {
                    bool warm = np <= P.enter_threads * W / 8;
#pragma unroll
                    for (int q = 0; q < QT; ++q) warm = warm && (q >= nqt || tau32[q] > 1.17549435e-38f);
                    calm = warm ? calm + 1 : 0;
                }
                while (np) {
                    if (pend) {
#pragma unroll
                        for (int q = 0; q < QT; ++q) {
                            if (pend & (1u << q)) {
                                double s;
                                exact_similarity(accp.get(q), accc.get(q), cnp, cnc, s_qnp[q], s_qnc[q], pw, cw, s);
                                if (!better(s, myrid, tau_s[q], tau_r[q])) {  // the list tightened meanwhile
                                    pend &= ~(1u << q);
                                    continue;
                                }
                                const int pos = atomicAdd(&cnt[q], 1);
                                if (pos < S) {
                                    cand_s[q * S + pos] = s;
                                    cand_r[q * S + pos] = myrid;
                                    pend &= ~(1u << q);
                                }
                            }
                        }
                    }
                    __syncthreads();
                    for (int q = 0; q < nqt; ++q)
                        if (cnt[q] >= S) compact_query(cand_s, cand_r, cnt, tau_s, tau_r, tau32, q, S, K);
                    np = __syncthreads_count(pend != 0);
                }
                // queues are drained at multiples of kFlushEvery: enter the fast mode on such a boundary
                if (calm >= kCalmIters && ((it + 1) & P.flush_mask) == 0) fastmode = true;
            } else {
                if (pend) {
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        if (pend & (1u << q)) {
                            double s;
                            exact_similarity(accp.get(q), accc.get(q), cnp, cnc, s_qnp[q], s_qnc[q], pw, cw, s);
                            const int pos = atomicAdd(&wq_cnt[wave], 1);
                            if (pos < kQueueCap) {
                                wq_s[wave * kQueueCap + pos] = s;
                                wq_r[wave * kQueueCap + pos] = myrid;
                                wq_q[wave * kQueueCap + pos] = (uint32_t)q;
                            } else {
                                s_flags[1] = 1;  // no room: this interval is replayed synchronously (below)
                            }
                        }
                    }
                }
                if (((it + 1) & P.flush_mask) == 0 || it == iters - 1) {
                    __syncthreads();
                    if (s_flags[1]) {
                        // A burst (typically a run of tied candidates) overran a wave's queue.  Nothing
                        // of this interval has reached the lists yet (queues are only drained here) and
                        // the thresholds have not moved, so the interval is simply run again with
                        // synchronous insertion: discard the queues and go back to its first slice.
                        __syncthreads();
                        if (tid < W) wq_cnt[tid] = 0;
                        if (tid == 0) {
                            s_flags[1] = 0;
                            s_flags[2] += 1;  // statistics: replayed intervals of this block
                        }
                        __syncthreads();
                        fastmode = false;
                        calm = 0;
                        it = (it & ~P.flush_mask) - 1;  // ++it -> first iteration of the interval
                        continue;
                    }
                    int rounds = 0, maxfill = 0;
#pragma unroll
                    for (int w = 0; w < W; ++w) {
                        maxfill = max(maxfill, wq_cnt[w]);
                        rounds = max(rounds, (min(wq_cnt[w], kQueueCap) + 63) >> 6);
                    }
                    for (int r = 0; r < rounds; ++r) {
                        const int i = lane + 64 * r;
                        const bool have = i < min(wq_cnt[wave], kQueueCap);
                        const double es = have ? wq_s[wave * kQueueCap + i] : 0.0;
                        const uint32_t er = have ? wq_r[wave * kQueueCap + i] : 0u;
                        const int eq = have ? (int)wq_q[wave * kQueueCap + i] : 0;
                        insert_sync(have, es, er, eq, cand_s, cand_r, cnt, tau_s, tau_r, tau32, nqt, S, K);
                    }
                    __syncthreads();
                    if (tid < W) wq_cnt[tid] = 0;
                    __syncthreads();
                    if (maxfill > kQueueCap / 2) {  // a burst of survivors: back to synchronous insertion
                        fastmode = false;
                        calm = 0;
                    }
                }
            }
        } else {
            double accp[QT], accc[QT];
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                accp[q] = 0.0;
                accc[q] = 0.0;
            }
            if (live) {
                dots_generic<QT>(P.fp, reinterpret_cast<const uint2 *>(smem + P.fp.off_hash),
                                 reinterpret_cast<const double *>(smem + P.fp.off_panel), slice, lane, accp);
                dots_generic<QT>(P.fc, reinterpret_cast<const uint2 *>(smem + P.fc.off_hash),
                                 reinterpret_cast<const double *>(smem + P.fc.off_panel), slice, lane, accc);
            }
            double cnp = 0.0, cnc = 0.0;
            uint32_t myrid = 0u;
            if (valid) {
                cnp = P.fp.norm[row];
                cnc = P.fc.norm[row];
                myrid = P.rid[row];
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    if (q < nqt && row != s_qrow[q]) {
                        double s;
                        if (exact_similarity(accp[q], accc[q], cnp, cnc, s_qnp[q], s_qnc[q], pw, cw, s) &&
                            better(s, myrid, tau_s[q], tau_r[q]))
                            pend |= 1u << q;
                    }
                }
            }
            while (__syncthreads_or(pend != 0)) {
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    if (pend & (1u << q)) {
                        double s;
                        exact_similarity(accp[q], accc[q], cnp, cnc, s_qnp[q], s_qnc[q], pw, cw, s);
                        if (!better(s, myrid, tau_s[q], tau_r[q])) {
                            pend &= ~(1u << q);
                            continue;
                        }
                        const int pos = atomicAdd(&cnt[q], 1);
                        if (pos < S) {
                            cand_s[q * S + pos] = s;
                            cand_r[q * S + pos] = myrid;
                            pend &= ~(1u << q);
                        }
                    }
                }
                __syncthreads();
                for (int q = 0; q < nqt; ++q)
                    if (cnt[q] >= S) compact_query(cand_s, cand_r, cnt, tau_s, tau_r, tau32, q, S, K);
            }
        }
    }
    if (tid == 0 && s_flags[0] && P.overflow) atomicAdd(P.overflow, 1);  // (no path sets it any more: kept as a tripwire)
    if (tid == 0 && s_flags[2] && P.overflow) atomicAdd(P.overflow + 1, s_flags[2]);  // locrec_knn_replayed_intervals
    // final compaction and write-out of this chunk's lists
    for (int q = 0; q < nqt; ++q) {
        compact_query(cand_s, cand_r, cnt, tau_s, tau_r, tau32, q, S, K);
        const int m = cnt[q];
        const int64_t base = ((int64_t)(q0 + q) * P.nchunks + blockIdx.x) * K;
        for (int i = tid; i < m; i += blockDim.x) {
            P.part_s[base + i] = cand_s[q * S + i];
            P.part_rid[base + i] = cand_r[q * S + i];
        }
        if (tid == 0) P.part_cnt[(int64_t)(q0 + q) * P.nchunks + blockIdx.x] = m;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// Single-request path (the reference's own operator: one person against everybody,
// KnnRecommender.scala:22-25).  A per-block top-K has no time to warm its threshold up when a
// block sees only a few hundred candidates, so one request runs as a pure stream instead:
//   knn_scan1    every candidate's exact similarity -> S[row] (fp64, 0 = not a candidate) and a
//                65536-bin histogram of s (global atomics; s <= pw + cw = 1)
//   knn_select1  one block walks the histogram from the top to the bin b* that holds the K-th value
//   knn_collect1 rows with bin(s) >= b* are appended to a short list (K + the population of b*)
//   knn_final1   one block sorts the list (s desc, rid asc) and writes the K best
// S is also the input of the large-K path (K >= #candidates: every positive row is a neighbour).

constexpr int kHistBins = 4096;   // block-private in LDS, flushed once per block
constexpr int kCollectCap = 8192;

struct Scan1Params {
    Family fp, fc;
    int32_t qrow;
    int32_t nrows, nslices;   // nslices = END of the scanned slice range (exclusive)
    int32_t slice0;
    double pw, cw;
    double *S;            // [nrows]
    uint32_t *hist;       // [kHistBins]
    const uint32_t *ss;   // or nullptr: [nrows] integer sums of squares, place | category << 16 (knn_scan1_direct8)
};

__device__ __forceinline__ int sim_bin(double s)
{
    const int b = (int)(s * (double)kHistBins);
    return b < kHistBins - 1 ? b : kHistBins - 1;
}

// One block of 16 waves per CU (measured: 47 us; two blocks of 8 waves: 57 us); each wave strides over the slices with the next slice's first
// load groups already in flight, and the block keeps a private histogram in LDS (flushed once).
constexpr int kScan1Waves = 16;

template <int MODE>
__global__ __launch_bounds__(kScan1Waves * 64) void knn_scan1(const Scan1Params P)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int s_qrow[1];
    __shared__ int s_nrows;
    __shared__ double s_qn[2];
    __shared__ uint32_t s_hist[kHistBins];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) {
        s_qrow[0] = P.qrow;
        s_qn[0] = P.fp.norm[P.qrow];
        s_qn[1] = P.fc.norm[P.qrow];
    }
    for (int i = tid; i < kHistBins; i += blockDim.x) s_hist[i] = 0u;
    // the first slice's loads do not depend on the panel: they go out before it is built (a block
    // per CU and a single round of blocks - the panel build would otherwise be fully exposed)
    const int stride = gridDim.x * kScan1Waves;
    int slice = P.slice0 + blockIdx.x * kScan1Waves + wave;
    const u32x4 *bp = nullptr;
    int w4p = 0;
    Group4 gp{};
    if constexpr (MODE != 0) {
        if (slice < P.nslices) {
            bp = reinterpret_cast<const u32x4 *>(P.fp.sell + P.fp.sell_off[slice]) + lane;
            w4p = __builtin_amdgcn_readfirstlane(P.fp.sell_w[slice] >> 2);
            gp = load_group(bp, 0, w4p);
        }
    }
    __syncthreads();
    if constexpr (MODE != 0) {
        build_panel_packed<1, uint32_t>(P.fp, s_qrow, 1, reinterpret_cast<uint32_t *>(smem + P.fp.off_hash),
                                        reinterpret_cast<uint32_t *>(smem + P.fp.off_panel), &s_nrows, P.fp.pop_h > 0 && !P.fp.direct ? reinterpret_cast<unsigned short *>(smem + P.fp.off_pop) : nullptr);
        build_panel_packed<1, uint32_t>(P.fc, s_qrow, 1, reinterpret_cast<uint32_t *>(smem + P.fc.off_hash),
                                        reinterpret_cast<uint32_t *>(smem + P.fc.off_panel), &s_nrows, P.fc.pop_h > 0 && !P.fc.direct ? reinterpret_cast<unsigned short *>(smem + P.fc.off_pop) : nullptr);
    } else {
        build_panel_generic<1>(P.fp, s_qrow, 1, reinterpret_cast<uint2 *>(smem + P.fp.off_hash),
                               reinterpret_cast<double *>(smem + P.fp.off_panel), &s_nrows);
        build_panel_generic<1>(P.fc, s_qrow, 1, reinterpret_cast<uint2 *>(smem + P.fc.off_hash),
                               reinterpret_cast<double *>(smem + P.fc.off_panel), &s_nrows);
    }
    const double qnp = s_qn[0], qnc = s_qn[1];
    const double pw = P.pw, cw = P.cw;
    if constexpr (MODE != 0) {
        const HotFam hp = make_hot(P.fp, smem);
        const HotFam hc = make_hot(P.fc, smem);
        // software pipeline over this wave's slices: first groups of the NEXT slice are loaded
        // before the current one is processed
        // (only the place family is prefetched across slices: the short category rows are loaded at
        // the top of their own slice and arrive while the place family is being processed; holding a
        // second prefetched group for them spilled registers at the 128-VGPR budget of 16 waves per CU)
        for (; slice < P.nslices; slice += stride) {
            const int row = slice * 64 + lane;
            const bool valid = row < P.nrows;
            const u32x4 *bc = reinterpret_cast<const u32x4 *>(P.fc.sell + P.fc.sell_off[slice]) + lane;
            const int w4c = __builtin_amdgcn_readfirstlane(P.fc.sell_w[slice] >> 2);
            const Group4 gc = load_group(bc, 0, w4c);
            const double cnp = valid ? P.fp.norm[row] : 0.0;
            const double cnc = valid ? P.fc.norm[row] : 0.0;
            const int nslice = slice + stride;
            const u32x4 *nbp = nullptr;
            int nw4p = 0;
            Group4 ngp{};
            if (nslice < P.nslices) {
                nbp = reinterpret_cast<const u32x4 *>(P.fp.sell + P.fp.sell_off[nslice]) + lane;
                nw4p = __builtin_amdgcn_readfirstlane(P.fp.sell_w[nslice] >> 2);
                ngp = load_group(nbp, 0, nw4p);
            }
            Acc<1, 1> accp, accc;
            accp.zero();
            accc.zero();
            const int sp4 = P.fp.sell_split ? __builtin_amdgcn_readfirstlane(P.fp.sell_split[slice]) : 0;
            family_dots_packed<1, 1>(hp, bp, w4p, gp, accp, sp4);
            family_dots_packed<1, 1>(hc, bc, w4c, gc, accc);
            double s = 0.0;
            bool have = false;
            if (valid && row != P.qrow)
                have = exact_similarity(accp.get(0), accc.get(0), cnp, cnc, qnp, qnc, pw, cw, s);
            if (!have) s = 0.0;
            if (valid) P.S[row] = s;
            if (have && P.hist) atomicAdd(&s_hist[sim_bin(s)], 1u);
            bp = nbp; w4p = nw4p; gp = ngp;
        }
    } else {
        for (; slice < P.nslices; slice += stride) {
            const int row = slice * 64 + lane;
            const bool valid = row < P.nrows;
            double accp[1] = {0.0}, accc[1] = {0.0};
            dots_generic<1>(P.fp, reinterpret_cast<const uint2 *>(smem + P.fp.off_hash),
                            reinterpret_cast<const double *>(smem + P.fp.off_panel), slice, lane, accp);
            dots_generic<1>(P.fc, reinterpret_cast<const uint2 *>(smem + P.fc.off_hash),
                            reinterpret_cast<const double *>(smem + P.fc.off_panel), slice, lane, accc);
            double s = 0.0;
            bool have = false;
            if (valid && row != P.qrow) {
                const double cnp = P.fp.norm[row], cnc = P.fc.norm[row];
                have = exact_similarity(accp[0], accc[0], cnp, cnc, qnp, qnc, pw, cw, s);
            }
            if (!have) s = 0.0;
            if (valid) P.S[row] = s;
            if (have && P.hist) atomicAdd(&s_hist[sim_bin(s)], 1u);
        }
    }
    __syncthreads();
    if (P.hist)
        for (int i = tid; i < kHistBins; i += blockDim.x) {
            const uint32_t h = s_hist[i];
            if (h) atomicAdd(&P.hist[i], h);
        }
}

// knn_scan1 with DIRECT byte tables: when every stored value fits a byte (PACK16-legal data) and p_dim + c_dim
// bytes fit the LDS next to the histogram (one 16-wave block per CU: ~140 KB are free), the query's two vectors
// are expanded into dense u8 tables indexed by the (renumbered) dimension - no hash, no slot map: per stored
// element one shift, one mask, one ds_read_u8 and one v_mad_u32_u24 instead of the ~18 instructions of the
// hashed lookup.  knn_scan1<1> issues 9.4 M wave64 VALU instructions per request at cfg2 (half of its 35 us);
// this form leaves the stream.  Same loop structure, same outputs.
constexpr int kDirect8MaxBytes = 128 * 1024;

__device__ __forceinline__ void direct8_accum4(const u32x4 e4, const unsigned char *tab, int vbits, uint32_t vmask, uint32_t &acc)
{
    const uint32_t ee[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) acc += (ee[t] & vmask) * (uint32_t)tab[ee[t] >> vbits];  // a padding element is 0: index 0, value 0
}

__device__ __forceinline__ void direct8_dots(const unsigned char *tab, int vbits, const u32x4 *lane_base, int w4, Group4 cur,
                                             uint32_t &acc)
{
    const uint32_t vmask = (1u << vbits) - 1u;
    for (int j = 0; j < w4; j += 4) {
        const Group4 nxt = load_group(lane_base, j + 4, w4);
        direct8_accum4(cur.a0, tab, vbits, vmask, acc);
        if (j + 1 < w4) direct8_accum4(cur.a1, tab, vbits, vmask, acc);
        if (j + 2 < w4) direct8_accum4(cur.a2, tab, vbits, vmask, acc);
        if (j + 3 < w4) direct8_accum4(cur.a3, tab, vbits, vmask, acc);
        cur = nxt;
    }
}

__global__ __launch_bounds__(kScan1Waves * 64) void knn_scan1_direct8(const Scan1Params P)
{
    extern __shared__ __align__(16) unsigned char smem[];  // [p_dim bytes, padded to 16][c_dim bytes, padded to 16]
    __shared__ double s_qn[2];
    __shared__ uint32_t s_hist[kHistBins];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pbytes = (P.fp.rows_cap + 15) & ~15, cbytes = (P.fc.rows_cap + 15) & ~15;  // rows_cap = the family's dimension here
    unsigned char *tab_p = smem, *tab_c = smem + pbytes;
    if (tid == 0) {
        s_qn[0] = P.fp.norm[P.qrow];
        s_qn[1] = P.fc.norm[P.qrow];
    }
    for (int i = tid; i < kHistBins; i += blockDim.x) s_hist[i] = 0u;
    // the first slice's loads do not depend on the tables: they go out before those are built
    const int stride = gridDim.x * kScan1Waves;
    int slice = P.slice0 + blockIdx.x * kScan1Waves + wave;
    const u32x4 *bp = nullptr;
    int w4p = 0;
    Group4 gp{};
    if (slice < P.nslices) {
        bp = reinterpret_cast<const u32x4 *>(P.fp.sell + P.fp.sell_off[slice]) + lane;
        w4p = __builtin_amdgcn_readfirstlane(P.fp.sell_w[slice] >> 2);
        gp = load_group(bp, 0, w4p);
    }
    {
        u32x4 *z = reinterpret_cast<u32x4 *>(smem);
        for (int i = tid; i < (pbytes + cbytes) / 16; i += blockDim.x) z[i] = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    for (int64_t e = P.fp.csr_ptr[P.qrow] + tid; e < P.fp.csr_ptr[P.qrow + 1]; e += blockDim.x)
        tab_p[P.fp.csr_idx[e]] = (unsigned char)P.fp.csr_val[e];
    for (int64_t e = P.fc.csr_ptr[P.qrow] + tid; e < P.fc.csr_ptr[P.qrow + 1]; e += blockDim.x)
        tab_c[P.fc.csr_idx[e]] = (unsigned char)P.fc.csr_val[e];
    __syncthreads();
    const double qnp = s_qn[0], qnc = s_qn[1];
    const double pw = P.pw, cw = P.cw;
    const int vbp = P.fp.vbits, vbc = P.fc.vbits;
    for (; slice < P.nslices; slice += stride) {
        const int row = slice * 64 + lane;
        const bool valid = row < P.nrows;
        const u32x4 *bc = reinterpret_cast<const u32x4 *>(P.fc.sell + P.fc.sell_off[slice]) + lane;
        const int w4c = __builtin_amdgcn_readfirstlane(P.fc.sell_w[slice] >> 2);
        const Group4 gc = load_group(bc, 0, w4c);
        // the candidate's norms: sqrt of its exact integer sums of squares where the index keeps them (4 bytes per
        // row instead of two doubles; the same bits: Distance.vectorLength is sqrt of that very sum)
        double cnp = 0.0, cnc = 0.0;
        if (P.ss) {
            const uint32_t q2 = valid ? P.ss[row] : 0u;
            cnp = sqrt((double)(q2 & 0xFFFFu));
            cnc = sqrt((double)(q2 >> 16));
        } else if (valid) {
            cnp = P.fp.norm[row];
            cnc = P.fc.norm[row];
        }
        const int nslice = slice + stride;
        const u32x4 *nbp = nullptr;
        int nw4p = 0;
        Group4 ngp{};
        if (nslice < P.nslices) {
            nbp = reinterpret_cast<const u32x4 *>(P.fp.sell + P.fp.sell_off[nslice]) + lane;
            nw4p = __builtin_amdgcn_readfirstlane(P.fp.sell_w[nslice] >> 2);
            ngp = load_group(nbp, 0, nw4p);
        }
        uint32_t dp = 0u, dc = 0u;
        direct8_dots(tab_p, vbp, bp, w4p, gp, dp);
        direct8_dots(tab_c, vbc, bc, w4c, gc, dc);
        double sx = 0.0;
        bool have = false;
        if (valid && row != P.qrow) have = exact_similarity(dp, dc, cnp, cnc, qnp, qnc, pw, cw, sx);
        if (!have) sx = 0.0;
        if (valid) P.S[row] = sx;
        if (have && P.hist) atomicAdd(&s_hist[sim_bin(sx)], 1u);
        bp = nbp;
        w4p = nw4p;
        gp = ngp;
    }
    __syncthreads();
    if (P.hist)
        for (int i = tid; i < kHistBins; i += blockDim.x) {
            const uint32_t h = s_hist[i];
            if (h) atomicAdd(&P.hist[i], h);
        }
}

// sel[0] = b*, sel[1] = number of candidates in bins > b*, sel[2] = total candidates
// Also leaves the workspace clean for the next request: the histogram is zeroed once every thread
// is done with it and the collect counter sel[3] is reset, so a request needs no memset launches.
__global__ __launch_bounds__(1024) void knn_select1(uint32_t *hist, int32_t K, int32_t *sel)
{
    __shared__ uint32_t suf[2][1024];  // suffix sums over the per-thread bin ranges (Hillis-Steele)
    const int t = threadIdx.x;
    constexpr int per = kHistBins / 1024;
    uint32_t mine = 0;
    for (int i = 0; i < per; ++i) mine += hist[t * per + i];
    suf[0][t] = mine;
    __syncthreads();
    int cur = 0;
    for (int d = 1; d < 1024; d <<= 1) {
        suf[cur ^ 1][t] = suf[cur][t] + (t + d < 1024 ? suf[cur][t + d] : 0u);
        cur ^= 1;
        __syncthreads();
    }
    const uint32_t incl = suf[cur][t];        // candidates in this thread's bins and above
    const uint32_t above_me = incl - mine;    // strictly above this thread's range
    const uint32_t total = suf[cur][0];
    if (total <= (uint32_t)K) {               // fewer candidates than K: take them all
        if (t == 0) {
            sel[0] = 0;
            sel[1] = (int32_t)(total - hist[0]);
            sel[2] = (int32_t)total;
        }
    } else if (above_me < (uint32_t)K && incl >= (uint32_t)K) {  // exactly one thread: the K-th value is in its range
        uint32_t above = above_me;
        int b = t * per + per - 1;
        for (; b > t * per; --b) {
            if (above + hist[b] >= (uint32_t)K) break;
            above += hist[b];
        }
        sel[0] = b;
        sel[1] = (int32_t)above;
        sel[2] = (int32_t)total;
    }
    if (t == 0) sel[3] = 0;  // knn_collect1's list counter
    __syncthreads();         // every read of hist above is done
    for (int i = 0; i < per; ++i) hist[t * per + i] = 0u;
}

__global__ __launch_bounds__(256) void knn_collect1(const double *S, const uint32_t *rid, int32_t row0,
                                                    int32_t nrows, const int32_t *sel, double *list_s,
                                                    uint32_t *list_r, int32_t *list_n)
{
    const int row = row0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrows) return;
    const double s = S[row];
    if (s > 0 && sim_bin(s) >= sel[0]) {
        const int pos = atomicAdd(list_n, 1);
        if (pos < kCollectCap) {
            list_s[pos] = s;
            list_r[pos] = rid[row];
        }
    }
}

// host: the request's result ALSO goes straight into the pinned staging buffer in locrec_knn_fetch_topk's layout
// (flag at 0, ids at 16, similarities behind them, then the count), so that reading it back is the request's one
// synchronisation and nothing else.
// (Measured negative result: knn_select1 folded into a fatter collect - every block finding b* for itself, this
// kernel cleaning the histogram afterwards - made the request three launches and exactly as fast, 0.057 ms.)
__global__ __launch_bounds__(256) void knn_final1(const double *list_s, const uint32_t *list_r,
                                                  const int32_t *list_n, int32_t K, const int64_t *ids_by_rank,
                                                  const int32_t *row_of_rid, int64_t *out_ids, double *out_sims,
                                                  int32_t *out_rows, int64_t *out_cnt, int32_t *overflow,
                                                  unsigned char *host)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int n = *list_n;
    const int tid = threadIdx.x;
    int64_t *h_ids = host ? reinterpret_cast<int64_t *>(host + 16) : nullptr;
    double *h_sims = host ? reinterpret_cast<double *>(host + 16 + (size_t)K * 8) : nullptr;
    int64_t *h_cnt = host ? reinterpret_cast<int64_t *>(host + 16 + (size_t)K * 16) : nullptr;
    if (n > kCollectCap) {  // pathological tie mass in the deciding bin: the caller takes the chunked path
        if (tid == 0) {
            *overflow = 1;
            out_cnt[0] = 0;
            if (host) {
                *reinterpret_cast<int32_t *>(host) = 1;
                *h_cnt = 0;
            }
        }
        return;
    }
    int n2 = 2;
    while (n2 < n) n2 <<= 1;
    double *s = reinterpret_cast<double *>(smem);
    uint32_t *r = reinterpret_cast<uint32_t *>(s + kCollectCap);
    for (int i = tid; i < n2; i += blockDim.x) {
        s[i] = i < n ? list_s[i] : -1.0;
        r[i] = i < n ? list_r[i] : 0xFFFFFFFFu;
    }
    __syncthreads();
    block_sort_desc(s, r, n2);
    const int m = min(n, K);
    for (int i = tid; i < K; i += blockDim.x) {
        const bool ok = i < m;
        const uint32_t rr = ok ? r[i] : 0u;
        const int64_t id = ok ? ids_by_rank[rr] : -1;
        const double sim = ok ? s[i] : 0.0;
        out_ids[i] = id;
        out_sims[i] = sim;
        out_rows[i] = ok ? row_of_rid[rr] : -1;
        if (host) {
            h_ids[i] = id;
            h_sims[i] = sim;
        }
    }
    if (tid == 0) {
        out_cnt[0] = m;
        *overflow = 0;
        if (host) {
            *reinterpret_cast<int32_t *>(host) = 0;
            *h_cnt = m;
        }
    }
}

// First level of a two-level merge (a single request is cut into more chunks than one block can
// sort): block (g, q) merges chunks [g*G, (g+1)*G) of query q into one list of <= K entries.
__global__ __launch_bounds__(256) void knn_merge_partial(
    const double *part_s, const uint32_t *part_rid, const int32_t *part_cnt, int32_t nchunks, int32_t K,
    int32_t G, int32_t M /* pow2 >= G*K */, double *out_s, uint32_t *out_rid, int32_t *out_cnt, int32_t ngroups)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *s = reinterpret_cast<double *>(smem);
    uint32_t *r = reinterpret_cast<uint32_t *>(s + M);
    __shared__ int total;
    const int g = blockIdx.x, q = blockIdx.y;
    const int tid = threadIdx.x;
    if (tid == 0) total = 0;
    for (int i = tid; i < M; i += blockDim.x) {
        s[i] = -1.0;
        r[i] = 0xFFFFFFFFu;
    }
    __syncthreads();
    const int c0 = g * G, c1 = min(c0 + G, nchunks);
    for (int c = c0; c < c1; ++c) {
        const int m = part_cnt[(int64_t)q * nchunks + c];
        const int64_t base = ((int64_t)q * nchunks + c) * K;
        for (int i = tid; i < m; i += blockDim.x) {
            s[(c - c0) * K + i] = part_s[base + i];
            r[(c - c0) * K + i] = part_rid[base + i];
        }
        if (tid == 0) total += m;
    }
    __syncthreads();
    block_sort_desc(s, r, M);
    const int m = min(total, K);
    const int64_t ob = ((int64_t)q * ngroups + g) * K;
    for (int i = tid; i < m; i += blockDim.x) {
        out_s[ob + i] = s[i];
        out_rid[ob + i] = r[i];
    }
    if (tid == 0) out_cnt[(int64_t)q * ngroups + g] = m;
}

// Merge the per-chunk lists of one query (orderBy(desc).limit(K), :47-48).
__global__ __launch_bounds__(256) void knn_merge(
    const double *part_s, const uint32_t *part_rid, const int32_t *part_cnt, int32_t nchunks, int32_t K,
    int32_t M /* pow2 >= nchunks*K */, const int64_t *ids_by_rank, const int32_t *row_of_rid,
    int64_t *out_ids, double *out_sims, int32_t *out_rows, int64_t *out_cnt)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *s = reinterpret_cast<double *>(smem);
    uint32_t *r = reinterpret_cast<uint32_t *>(s + M);
    __shared__ int total;
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid == 0) total = 0;
    for (int i = tid; i < M; i += blockDim.x) {
        s[i] = -1.0;
        r[i] = 0xFFFFFFFFu;
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int m = part_cnt[(int64_t)q * nchunks + c];
        const int64_t base = ((int64_t)q * nchunks + c) * K;
        for (int i = tid; i < m; i += blockDim.x) {
            s[c * K + i] = part_s[base + i];
            r[c * K + i] = part_rid[base + i];
        }
        if (tid == 0) total += m;
    }
    __syncthreads();
    block_sort_desc(s, r, M);
    const int m = min(total, K);
    for (int i = tid; i < K; i += blockDim.x) {
        const bool ok = i < m;
        const uint32_t rid = ok ? r[i] : 0u;
        out_ids[(int64_t)q * K + i] = ok ? ids_by_rank[rid] : -1;
        out_sims[(int64_t)q * K + i] = ok ? s[i] : 0.0;
        out_rows[(int64_t)q * K + i] = ok ? row_of_rid[rid] : -1;
    }
    if (tid == 0) out_cnt[q] = m;
}

// ---------------------------------------------------------------------------
// Per-row format fallback (knn_build.hip): the index's few WIDE rows - integer counts that by themselves break the
// head / tail form's legality (a value of 256 or more, a sum of squares of 65,536 or more; counts are unbounded in
// the reference's data, RatingVectorsBuilder.scala:69) - are all padding in the packed images, so no packed kernel
// ever sees them as candidates.  The two kernels below add them back from the plain CSR (true values): the dot of a
// (query, wide row) pair is a merge of two index-sorted rows - integer products and sums, exact in any order - and
// the similarity is exact_similarity's, bit for bit what the row scan would have produced.
struct SideCsr {
    const int64_t *p_ptr, *c_ptr;
    const int32_t *p_idx, *c_idx;
    const double *p_val, *c_val;
    const double *norm_p, *norm_c;
    const float *inorm_p, *inorm_c;  // f32 inverse norms (0 = absent vector): the side kernel's prefilter
    // the wide rows lane-major (knn_index.h, side_p / side_c): element j of wide row w at off[w / 64] + j * 64 + w % 64
    const int2 *side_p, *side_c;
    const int32_t *side_off_p, *side_off_c, *side_w_p, *side_w_c;
};

__device__ __forceinline__ double side_merge_dot(const int64_t *ptr, const int32_t *idx, const double *val, int32_t a, int32_t b)
{
    int64_t i = ptr[a], j = ptr[b];
    const int64_t ie = ptr[a + 1], je = ptr[b + 1];
    double sum = 0.0;
    while (i < ie && j < je) {
        const int32_t x = idx[i], y = idx[j];
        if (x == y) {
            const double t = val[i] * val[j];
            sum = sum + t;
            ++i;
            ++j;
        } else if (x < y) {
            ++i;
        } else {
            ++j;
        }
    }
    return sum;
}

__device__ __forceinline__ bool side_similarity(const SideCsr &C, int32_t qrow, int32_t row, double pw, double cw, double &sx)
{
    const double dp = side_merge_dot(C.p_ptr, C.p_idx, C.p_val, qrow, row);
    const double dc = side_merge_dot(C.c_ptr, C.c_idx, C.c_val, qrow, row);
    return exact_similarity(dp, dc, C.norm_p[row], C.norm_c[row], C.norm_p[qrow], C.norm_c[qrow], pw, cw, sx);
}

// single request (stream path): the wide rows' similarities into S and the histogram, behind the scan kernel
__global__ __launch_bounds__(256) void knn_side_scan1(const SideCsr C, const int32_t *wide_rows, int32_t nwide, int32_t qrow,
                                                      int32_t row0, int32_t row1, double pw, double cw, double *S, uint32_t *hist)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwide) return;
    const int32_t row = wide_rows[w];
    if (row < row0 || row >= row1 || row == qrow) return;  // person_id =!= personId (KnnRecommender.scala:89)
    double sx = 0.0;
    if (!side_similarity(C, qrow, row, pw, cw, sx)) return;
    S[row] = sx;
    if (hist) atomicAdd(&hist[sim_bin(sx)], 1u);
}

// batched path: one block per query, behind knn_merge.  The query's K-list (out arrays) and the wide rows that beat
// its K-th entry are sorted together by (similarity desc, id rank asc) and the best K written back.
// The query's two vectors are first expanded in LDS - the categories as a dense table (c_dim <= 64 in this mode), the
// places as an open-addressing hash of H slots (H >= 2 x the query's non-zeros) - so that a (query, wide row) pair
// costs one walk over the WIDE row's elements with LDS probes instead of a two-pointer merge of two global rows
// (16,384 queries x 1,000 wide rows of a cfg2 batch: 4 ms as merges).  A query too long for the hash (H > hash_cap)
// takes the merge.
__global__ __launch_bounds__(256) void knn_side_topk(const SideCsr C, const int32_t *wide_rows, int32_t nwide,
                                                     const int32_t *qrows, int32_t qrow0, int32_t row0, int32_t row1, double pw,
                                                     double cw, int32_t K, const uint32_t *rid_of_row, const int64_t *ids_by_rank,
                                                     const int32_t *row_of_rid, int64_t *out_ids, double *out_sims,
                                                     int32_t *out_rows, int64_t *out_cnt, int32_t c_dim, int32_t hash_cap)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *s = reinterpret_cast<double *>(smem);
    __shared__ int n_in;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int cnt = (int)out_cnt[q];
    if (cnt < 0) return;  // not a valid query (knn_mark_absent runs after this kernel, but a rerun may come here again)
    int cap = 2;
    while (cap < K + nwide) cap <<= 1;
    uint32_t *r = reinterpret_cast<uint32_t *>(s + cap);
    double *cat = reinterpret_cast<double *>(r + cap);     // [c_dim]
    double *hval = cat + c_dim;                            // [hash_cap]
    int32_t *hkey = reinterpret_cast<int32_t *>(hval + hash_cap);
    const int32_t qrow = qrows ? qrows[q] : qrow0 + q;
    const int64_t qb = C.p_ptr[qrow], qe = C.p_ptr[qrow + 1];
    int H = 64;
    while (H < 2 * (int)(qe - qb)) H <<= 1;
    const bool hashed = H <= hash_cap;
    for (int i = tid; i < cnt; i += blockDim.x) {
        s[i] = out_sims[(int64_t)q * K + i];
        r[i] = rid_of_row[out_rows[(int64_t)q * K + i]];
    }
    for (int i = tid; i < c_dim; i += blockDim.x) cat[i] = 0.0;
    if (hashed)
        for (int i = tid; i < H; i += blockDim.x) hkey[i] = -1;
    if (tid == 0) n_in = cnt;
    __syncthreads();
    for (int64_t e = C.c_ptr[qrow] + tid; e < C.c_ptr[qrow + 1]; e += blockDim.x) cat[C.c_idx[e]] = C.c_val[e];
    if (hashed)
        for (int64_t e = qb + tid; e < qe; e += blockDim.x) {
            const int32_t key = C.p_idx[e];
            uint32_t slot = ((uint32_t)key * 2654435761u) & (uint32_t)(H - 1);
            while (atomicCAS(&hkey[slot], -1, key) != -1) slot = (slot + 1) & (uint32_t)(H - 1);  // (indices of a row are distinct)
            hval[slot] = C.p_val[e];
        }
    __syncthreads();
    // a full list only admits what beats its last entry
    const double tau_s = cnt >= K ? s[K - 1] : -1.0;
    const uint32_t tau_r = cnt >= K ? r[K - 1] : 0xFFFFFFFFu;
    const double qnp = C.norm_p[qrow], qnc = C.norm_c[qrow];
    const float qfp = qnp > 0.0 ? (float)(pw / qnp) * 1.0001f : 0.0f, qfc = qnc > 0.0 ? (float)(cw / qnc) * 1.0001f : 0.0f;
    const float tau32 = (float)(tau_s * (1.0 - 1e-4));
    for (int w = tid; w < nwide; w += blockDim.x) {   // a wave = 64 consecutive wide rows = one slice of the side image
        const int32_t row = wide_rows[w];
        if (row < row0 || row >= row1 || row == qrow) continue;
        double dp = 0.0, dc = 0.0;
        const int sl = w >> 6, ln = w & 63;
        if (hashed) {
            const int2 *img = C.side_p + C.side_off_p[sl] + ln;
            const int width = C.side_w_p[sl];
            for (int j = 0; j < width; ++j) {
                const int2 e = img[(int64_t)j * 64];   // coalesced: lane = wide row
                if (e.x < 0) continue;                 // padding
                uint32_t slot = ((uint32_t)e.x * 2654435761u) & (uint32_t)(H - 1);
                for (;;) {
                    const int32_t k2 = hkey[slot];
                    if (k2 == e.x) {
                        const double t = hval[slot] * (double)e.y;
                        dp = dp + t;
                        break;
                    }
                    if (k2 == -1) break;
                    slot = (slot + 1) & (uint32_t)(H - 1);
                }
            }
        } else {
            dp = side_merge_dot(C.p_ptr, C.p_idx, C.p_val, qrow, row);
        }
        {
            const int2 *img = C.side_c + C.side_off_c[sl] + ln;
            const int width = C.side_w_c[sl];
            for (int j = 0; j < width; ++j) {
                const int2 e = img[(int64_t)j * 64];
                if (e.x < 0) continue;
                const double t = cat[e.x] * (double)e.y;
                dc = dc + t;
            }
        }
        if (!(dp > 0.0) && !(dc > 0.0)) continue;  // no common dimension: not in the outer join (KnnRecommender.scala:91)
        if (cnt >= K) {
            // f32 upper bound against the list's last entry (the batched scans' one-sided 1e-4 margin) before the two
            // fp64 divisions of the exact path: nearly every wide row fails it
            const float ub = (float)dp * C.inorm_p[row] * qfp + (float)dc * C.inorm_c[row] * qfc;
            if (ub < tau32) continue;
        }
        double sx = 0.0;
        if (!exact_similarity(dp, dc, C.norm_p[row], C.norm_c[row], qnp, qnc, pw, cw, sx)) continue;
        const uint32_t rr = rid_of_row[row];
        if (cnt >= K && !better(sx, rr, tau_s, tau_r)) continue;
        const int pos = atomicAdd(&n_in, 1);
        s[pos] = sx;
        r[pos] = rr;
    }
    __syncthreads();
    const int total = n_in;
    if (total == cnt) return;  // no wide row enters: the list stands
    int n2 = 2;
    while (n2 < total) n2 <<= 1;
    for (int i = total + tid; i < n2; i += blockDim.x) {
        s[i] = -1.0;
        r[i] = 0xFFFFFFFFu;
    }
    __syncthreads();
    block_sort_desc(s, r, n2);
    const int m = min(total, K);
    for (int i = tid; i < K; i += blockDim.x) {
        const bool ok = i < m;
        const uint32_t rr = ok ? r[i] : 0u;
        out_ids[(int64_t)q * K + i] = ok ? ids_by_rank[rr] : -1;
        out_sims[(int64_t)q * K + i] = ok ? s[i] : 0.0;
        out_rows[(int64_t)q * K + i] = ok ? row_of_rid[rr] : -1;
    }
    if (tid == 0) out_cnt[q] = m;
}

// a5: makeRecommendations0 (KnnRecommender.scala:51-70) for one query per block.
// The <= K neighbours' rating rows are flattened in neighbour-rank order; one 64-bit key per row,
// compact place index << 16 | sequence number, is sorted in LDS (no payload to move); the
// products rating*similarity are gathered once, in parallel, and each place is then summed left
// to right from LDS -- in neighbour-rank order, the oracle's order.
// Block-wide exclusive prefix sum of one int per thread (blockDim.x <= 1024, a multiple of 64);
// *total receives the sum.  wtot: LDS scratch of 17 ints.
__device__ __forceinline__ int block_exclusive_scan(int v, int *wtot, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int w = 0; w < nw; ++w) {
            const int t = wtot[w];
            wtot[w] = acc;
            acc += t;
        }
        wtot[16] = acc;
    }
    __syncthreads();
    *total = wtot[16];
    return wtot[wave] + inc - v;
}

constexpr int kAggThreads = 1024;

__global__ __launch_bounds__(kAggThreads) void knn_aggregate(
    const int32_t *nb_rows, const double *nb_sims, const int64_t *nb_cnt, int32_t K,
    const int64_t *r_ptr, const int32_t *r_pidx, const double *r_rating, const int64_t *cplace_ids,
    int32_t M /* pow2 LDS capacity, <= kAggCap */, int64_t *out_place, double *out_est, int64_t *out_n,
    int32_t *out_overflow, int64_t out_stride, int32_t redo_only,
    unsigned char *host = nullptr /* one request: the result also goes into the pinned staging buffer, in
    locrec_knn_recommend's layout (count at 0, overflow flag at 16, *host_flag_src at 20, places at 64, estimates
    behind host_cap of them) */, const int32_t *host_flag_src = nullptr, int32_t host_cap = 0)
{
    // second pass of a batch: only the queries whose rows did not fit the first pass's smaller capacity
    if (redo_only && out_overflow[blockIdx.x] == 0) return;
    if (host && threadIdx.x == 0 && host_flag_src) *reinterpret_cast<int32_t *>(host + 20) = *host_flag_src;
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *key = reinterpret_cast<uint64_t *>(smem);       // [M]
    double *wrv = reinterpret_cast<double *>(key + M);        // [M] rating * similarity
    double *sv = wrv + M;                                     // [M] similarity
    int64_t *rbase = reinterpret_cast<int64_t *>(sv + M);     // [K] first rating row of neighbour i
    double *simv = reinterpret_cast<double *>(rbase + K);     // [K]
    int32_t *off = reinterpret_cast<int32_t *>(simv + K);     // [K+1] prefix of neighbour row counts
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int m = max(0, (int)nb_cnt[q]);  // -1: not a valid query (knn_mark_absent)
    const int32_t *rows = nb_rows + (int64_t)q * K;
    const double *sims = nb_sims + (int64_t)q * K;
    __shared__ int wtot[17];
    // neighbour row counts -> exclusive offsets (K <= 1024 = blockDim: one neighbour per thread);
    // counts are clamped to M + 1, so the int sums cannot overflow (<= 1024 * 4097)
    int mycnt = 0;
    if (tid < m) {
        const int64_t b = r_ptr[rows[tid]];
        rbase[tid] = b;
        simv[tid] = sims[tid];
        mycnt = (int32_t)min(r_ptr[rows[tid] + 1] - b, (int64_t)M + 1);
    }
    int Tsum = 0;
    const int myoff = block_exclusive_scan(mycnt, wtot, &Tsum);
    if (tid < m) off[tid] = myoff;
    if (tid == 0) off[m] = Tsum;
    __syncthreads();
    const int T = min(Tsum, M + 1);
    if (T > M) {
        if (tid == 0) {
            out_overflow[q] = 1;
            out_n[q] = 0;
            if (host) {
                *reinterpret_cast<int32_t *>(host + 16) = 1;
                *reinterpret_cast<int64_t *>(host) = 0;
            }
        }
        return;
    }
    int n2 = 2;
    while (n2 < T) n2 <<= 1;  // <= M
    auto neighbour_of = [&](int f) {  // last i with off[i] <= f
        int a = 0, b = m;
        while (b - a > 1) {
            const int mid = (a + b) >> 1;
            if (off[mid] <= f) a = mid; else b = mid;
        }
        return a;
    };
    for (int f = tid; f < n2; f += blockDim.x) {
        uint64_t k = ~0ull;
        if (f < T) {
            const int a = neighbour_of(f);
            k = ((uint64_t)(uint32_t)r_pidx[rbase[a] + (f - off[a])] << 16) | (uint64_t)f;
        }
        key[f] = k;
    }
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1) {  // bitonic sort, ascending
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += blockDim.x) {
                const int i = 2 * t - (t & (j - 1));
                const int l = i + j;
                const uint64_t ki = key[i], kl = key[l];
                if ((kl < ki) == ((i & k) == 0)) {
                    key[i] = kl;
                    key[l] = ki;
                }
            }
            __syncthreads();
        }
    }
    for (int t = tid; t < T; t += blockDim.x) {  // the products, once, in parallel
        const int f = (int)(key[t] & 0xFFFFu);
        const int a = neighbour_of(f);
        const double sim = simv[a];
        wrv[t] = r_rating[rbase[a] + (f - off[a])] * sim;  // col("rating") * col("similarity") (:59)
        sv[t] = sim;
    }
    __syncthreads();
    // heads per thread (each thread owns n2/blockDim consecutive positions)
    const int per = n2 / (int)blockDim.x > 0 ? n2 / (int)blockDim.x : 1;
    const int lo = tid * per, hi = min(lo + per, T);
    int heads = 0;
    for (int i = lo; i < hi; ++i)
        if (i == 0 || (key[i] >> 16) != (key[i - 1] >> 16)) ++heads;
    int nheads = 0;
    int o = block_exclusive_scan(heads, wtot, &nheads);
    if (tid == 0) {
        out_n[q] = nheads;
        out_overflow[q] = 0;
        if (host) {
            *reinterpret_cast<int32_t *>(host + 16) = 0;
            *reinterpret_cast<int64_t *>(host) = nheads;
        }
    }
    int64_t *h_place = host ? reinterpret_cast<int64_t *>(host + 64) : nullptr;
    double *h_est = host ? reinterpret_cast<double *>(host + 64 + (size_t)host_cap * 8) : nullptr;
    for (int i = lo; i < hi; ++i) {
        const uint64_t pk = key[i] >> 16;
        if (i == 0 || pk != (key[i - 1] >> 16)) {
            double ws = 0.0, ss = 0.0;
            for (int t = i; t < T && (key[t] >> 16) == pk; ++t) {
                ws = ws + wrv[t];
                ss = ss + sv[t];
            }
            const int64_t place = cplace_ids[pk];
            const double est = ws / ss;   // :67
            out_place[(int64_t)q * out_stride + o] = place;
            out_est[(int64_t)q * out_stride + o] = est;
            if (host) {
                h_place[o] = place;
                h_est[o] = est;
            }
            ++o;
        }
    }
}

// rows of query q (out_n[q] of them, at q * stride) -> dense[off[q] ..): one block per query
__global__ __launch_bounds__(256) void knn_agg_compact(const int64_t *place, const double *est, const int64_t *out_n,
                                                       const int64_t *off, int64_t stride, int64_t *dense_place,
                                                       double *dense_est)
{
    const int q = blockIdx.x;
    const int64_t n = out_n[q], o = off[q];
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        dense_place[o + i] = place[(int64_t)q * stride + i];
        dense_est[o + i] = est[(int64_t)q * stride + i];
    }
}

// Range / all-pairs forms: a person whose place or category vector is empty is a legitimate
// candidate of the reference's outer join but not a valid QUERY (KnnRecommender.scala:77-83 throws
// "No such person" for it): its list is reported with count -1 instead of failing the whole batch.
__global__ void knn_mark_absent(const double *norm_p, const double *norm_c, const int32_t *qrows, int32_t qrow0,
                                int32_t nq, int32_t K, int64_t *out_ids, double *out_sims, int32_t *out_rows,
                                int64_t *out_cnt)
{
    const int q = blockIdx.x;
    if (q >= nq) return;
    const int row = qrows ? qrows[q] : qrow0 + q;
    if (norm_p[row] > 0.0 && norm_c[row] > 0.0) return;
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        out_ids[(int64_t)q * K + i] = -1;
        out_sims[(int64_t)q * K + i] = 0.0;
        out_rows[(int64_t)q * K + i] = -1;
    }
    if (threadIdx.x == 0) out_cnt[q] = -1;
}

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
        if (!qrows_dev && rows >= nq && nq > 0) {
            int64_t widest = 0;
            for (int64_t i = 0; i + nq <= rows; ++i) widest = std::max(widest, ht.tail_hits_ps[(size_t)(i + nq)] - ht.tail_hits_ps[(size_t)i]);
            want = std::max(widest, total_hits);
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
                       ix->rid.p, row0, row1, ix->sel1.p, ix->list1_s.p, ix->list1_r.p, ix->sel1.p + 3);
    ix->hist1_dirty = nohist;  // knn_select1 cleans up behind itself
    const size_t flds = (size_t)kCollectCap * 12;
    if (!ix->final1_attr) {
        LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_final1),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
        ix->final1_attr = true;
    }
    hipLaunchKernelGGL(knn_final1, dim3(1), dim3(256), flds, s, ix->list1_s.p, ix->list1_r.p, ix->sel1.p + 3, K,
                       ix->ids_by_rank.p, ix->row_of_rid.p, ix->out_ids.p, ix->out_sims.p, ix->out_rows.p,
                       ix->out_cnt.p, ix->sel1.p + 4, host);
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
            LOCREC_TRY(ix->list1_s.reserve(kCollectCap));
            LOCREC_TRY(ix->list1_r.reserve(kCollectCap));
            LOCREC_TRY(ix->sel1.reserve(8));
            LOCREC_TRY(ix->hist1.reserve(kHistBins));
            LOCREC_TRY(ix->tile_ovf.reserve(16));
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
                for (int t = 0; t < nt; ++t) {
                    const int64_t slot = longq[j0 + (size_t)t];
                    // (knn_select1 leaves histogram and counters clean behind itself; the first use cleans them here)
                    if (ix->hist1_dirty) {
                        LOCREC_HIP_TRY(hipMemsetAsync(ix->hist1.p, 0, kHistBins * sizeof(uint32_t), s));
                        LOCREC_HIP_TRY(hipMemsetAsync(ix->sel1.p, 0, 8 * sizeof(int32_t), s));
                        ix->hist1_dirty = false;
                    }
                    const double *col = nullptr;
                    LOCREC_TRY(knn_large_tile_column(ix, t, &col));
                    hipLaunchKernelGGL(knn_select1, dim3(1), dim3(1024), 0, s, ix->hist1.p, K, ix->sel1.p);
                    hipLaunchKernelGGL(knn_collect1, dim3((unsigned)std::max(1, (nrows + 255) / 256)), dim3(256), 0, s, col,
                                       ix->rid.p, 0, nrows, ix->sel1.p, ix->list1_s.p, ix->list1_r.p, ix->sel1.p + 3);
                    hipLaunchKernelGGL(knn_final1, dim3(1), dim3(256), flds, s, ix->list1_s.p, ix->list1_r.p, ix->sel1.p + 3, K,
                                       ix->ids_by_rank.p, ix->row_of_rid.p, ix->out_ids.p + slot * K, ix->out_sims.p + slot * K,
                                       ix->out_rows.p + slot * K, ix->out_cnt.p + slot, ix->tile_ovf.p + t,
                                       static_cast<unsigned char *>(nullptr));
                }
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
    if (const char *e = std::getenv("LOCREC_KNN_BLOCKS")) want = std::max(1, (std::atoi(e) + ntiles - 1) / ntiles);  // tuning
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
    if (const char *e = std::getenv("LOCREC_KNN_FLUSH")) {  // tuning: 1, 2, 4, 8, 16
        const int v = std::atoi(e);
        if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) P.flush_mask = v - 1;
    }
    if (const char *e = std::getenv("LOCREC_KNN_ENTER")) P.enter_threads = std::max(0, std::atoi(e));
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

}
}  // namespace locrec

// The index built on the HOST (the first implementation, single-threaded): kept behind
// LOCREC_KNN_HOST_BUILD=1 as the A/B partner of knn_build.hip's device build (tests/test_gpu_build.py).
static int32_t knn_create_host(
    int64_t n, const int64_t *person_ids,
    const int64_t *p_rowptr, const int32_t *p_idx, const double *p_val, int32_t p_dim,
    const int64_t *c_rowptr, const int32_t *c_idx, const double *c_val, int32_t c_dim,
    const int64_t *r_rowptr, const int64_t *r_place, const int64_t *r_rating,
    locrec_knn_index **out) try
{
    if (!out) return fail(LOCREC_E_INVALID_ARG, "out_index is NULL");
    *out = nullptr;
    if (n < 0 || n >= ((int64_t)1 << 31) - 64) return fail(LOCREC_E_INVALID_ARG, "bad person count");
    if (n > 0 && (!person_ids || !p_rowptr || !c_rowptr)) return fail(LOCREC_E_INVALID_ARG, "NULL input array");
    if (p_dim <= 0 || c_dim <= 0) return fail(LOCREC_E_INVALID_ARG, "vector sizes must be positive");
    LOCREC_TRY(ensure_device());
    std::unique_ptr<locrec_knn_index> ix(new (std::nothrow) locrec_knn_index);
    if (!ix) return fail(LOCREC_E_OOM, "host allocation failed");
    LOCREC_HIP_TRY(hipGetDevice(&ix->device));
    LOCREC_HIP_TRY(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
    ix->own_stream = true;
    ix->n = n;
    ix->nslices = (int32_t)((n + 63) / 64);
    ix->cand_slice0 = 0;
    ix->cand_slice1 = ix->nslices;
    knn_read_env(ix.get());
    const bool force_generic = std::getenv("LOCREC_KNN_FORCE_GENERIC") != nullptr;

    const bool dbg_t = debug_env("LOCREC_DEBUG_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg_t) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[locrec knn_create] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    // ---- validation (SparseVector invariants, RatingVectorsBuilder.scala:74-77)
    auto check_family = [&](const char *name, const int64_t *ptr, const int32_t *idx, const double *val,
                            int32_t dim, bool &integral, double &vmax, double &ssmax) -> int32_t {
        if (n == 0) return LOCREC_OK;
        if (ptr[0] != 0) return fail(LOCREC_E_INVALID_ARG, "%s rowptr must start at 0", name);
        for (int64_t r = 0; r < n; ++r) {
            if (ptr[r + 1] < ptr[r]) return fail(LOCREC_E_INVALID_ARG, "%s rowptr not monotone at %lld", name, (long long)r);
            double ss = 0;
            for (int64_t e = ptr[r]; e < ptr[r + 1]; ++e) {
                if (idx[e] < 0 || idx[e] >= dim)
                    return fail(LOCREC_E_INVALID_ARG, "%s index %d out of range [0,%d)", name, idx[e], dim);
                if (e > ptr[r] && idx[e] <= idx[e - 1])
                    return fail(LOCREC_E_INVALID_ARG, "%s indices of person %lld not strictly ascending", name,
                                (long long)person_ids[r]);
                const double v = val[e];
                if (!std::isfinite(v)) return fail(LOCREC_E_INVALID_ARG, "%s value is not finite", name);
                if (!(v >= 1.0) || v != std::floor(v)) integral = false;
                vmax = std::max(vmax, std::fabs(v));
                ss += v * v;
            }
            if (ptr[r + 1] > ptr[r] && !(ss > 0))
                return fail(LOCREC_E_INVALID_ARG, "%s vector of person %lld has zero norm", name,
                            (long long)person_ids[r]);
            ssmax = std::max(ssmax, ss);
        }
        return LOCREC_OK;
    };
    bool integral = true;
    double pvmax = 0, cvmax = 0, pss = 0, css = 0;
    LOCREC_TRY(check_family("place", p_rowptr, p_idx, p_val, p_dim, integral, pvmax, pss));
    LOCREC_TRY(check_family("category", c_rowptr, c_idx, c_val, c_dim, integral, cvmax, css));
    const int p_vbits = std::min(24, 32 - ceil_log2i(p_dim));
    const int c_vbits = std::min(24, 32 - ceil_log2i(c_dim));
    // exact u32 dots need every dot < 2^32; |dot| <= sqrt(ss_a * ss_b) <= max ss
    ix->packed = !force_generic && integral && p_dim < (1 << 20) - 1 && c_dim < (1 << 20) - 1 &&
                 pvmax < (double)(1u << p_vbits) && cvmax < (double)(1u << c_vbits) &&
                 pss < 4294967296.0 && css < 4294967296.0;

    ix->pack16 = ix->packed && pss < 65536.0 && css < 65536.0 && pvmax < 65536.0 && cvmax < 65536.0 &&
                 std::getenv("LOCREC_KNN_NO_PACK16") == nullptr;

    lap("validation");
    // ---- popularity split of the place family (PACKED formats, hashed panel): place indices are
    // renumbered by descending frequency (a permutation of the dimensions: every dot product is
    // unchanged, and integer sums do not depend on the order of the terms), so that a row's popular
    // indices come first.  new_of_old is empty when the split is not used.
    std::vector<int32_t> new_of_old;
    std::vector<int32_t> npop;  // per input row: number of indices that become < pop_h
    int32_t pop_h = 0;
    // head / tail form (knn_ht.h): PACK16 data whose values fit a byte and whose rows fit 24 bits
    const int ht_qt = 16;
    int32_t ht_h = std::min<int32_t>(p_dim, 512);
    if (const char *e = std::getenv("LOCREC_KNN_HT_H")) ht_h = std::min<int32_t>(p_dim, std::max(4, std::atoi(e)));  // tuning
    ht_h = std::min<int32_t>(ht_h, 65536 / (2 * ht_qt));  // the element's low half is the panel row's byte offset
    const bool want_ht = ix->pack16 && !ix->no_ht && !force_generic && n > 0 && n < ((int64_t)1 << 24) && pvmax < 256.0 &&
                         cvmax < 256.0 && c_dim <= kHtCatRows;
    if (ix->packed && std::getenv("LOCREC_KNN_NO_POP") == nullptr && n > 0 &&
        (ix->force_hash || (size_t)p_dim * 2 > (size_t)kDirectMaxBytes || want_ht)) {
        std::vector<int64_t> freq((size_t)p_dim, 0);
        for (int64_t e = 0; e < p_rowptr[n]; ++e) ++freq[p_idx[e]];
        std::vector<int32_t> by_freq((size_t)p_dim);
        std::iota(by_freq.begin(), by_freq.end(), 0);
        std::stable_sort(by_freq.begin(), by_freq.end(), [&](int32_t a, int32_t b) { return freq[a] > freq[b]; });
        new_of_old.resize((size_t)p_dim);
        for (int32_t i = 0; i < p_dim; ++i) new_of_old[by_freq[i]] = i;
        pop_h = std::min<int32_t>(p_dim, kPopTable);
        if (const char *e = std::getenv("LOCREC_KNN_POP_H")) pop_h = std::min<int32_t>(p_dim, std::max(64, std::atoi(e)));  // tuning
        // third sort key of the rows: their count of popular indices - of HEAD indices when the head / tail
        // form is built, so that the head rows of a slice have (nearly) one length
        const int32_t key_h = want_ht ? ht_h : pop_h;
        npop.assign((size_t)n, 0);
        for (int64_t r = 0; r < n; ++r)
            for (int64_t e = p_rowptr[r]; e < p_rowptr[r + 1]; ++e) npop[r] += new_of_old[p_idx[e]] < key_h ? 1 : 0;
    }

    lap("popularity");
    // ---- row order: ascending (nnz_place, nnz_category[, popular count]), stable
    std::vector<int32_t> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        const int64_t pa = p_rowptr[a + 1] - p_rowptr[a], pb = p_rowptr[b + 1] - p_rowptr[b];
        if (pa != pb) return pa < pb;
        const int64_t ca = c_rowptr[a + 1] - c_rowptr[a], cb = c_rowptr[b + 1] - c_rowptr[b];
        if (ca != cb) return ca < cb;
        return !npop.empty() && npop[a] < npop[b];
    });
    ix->ids_row.resize((size_t)n);
    ix->row_of_input.resize((size_t)n);
    for (int64_t r = 0; r < n; ++r) {
        ix->ids_row[r] = person_ids[order[r]];
        ix->row_of_input[order[r]] = (int32_t)r;
    }
    {
        ix->row_by_rank.resize((size_t)n);
        std::iota(ix->row_by_rank.begin(), ix->row_by_rank.end(), 0);
        std::sort(ix->row_by_rank.begin(), ix->row_by_rank.end(), [&](int32_t a, int32_t b) { return ix->ids_row[a] < ix->ids_row[b]; });
        ix->ids_sorted.resize((size_t)n);
        for (int64_t k = 0; k < n; ++k) ix->ids_sorted[k] = ix->ids_row[ix->row_by_rank[k]];
        const auto dupit = std::adjacent_find(ix->ids_sorted.begin(), ix->ids_sorted.end());
        if (dupit != ix->ids_sorted.end()) return fail(LOCREC_E_INVALID_ARG, "duplicate person_id %lld", (long long)*dupit);
    }
    lap("row order + id map");
    auto gather = [&](const int64_t *ptr, const int32_t *idx, const double *val, int32_t dim, int vbits,
                      HostFamily &h) {
        h.dim = dim;
        h.vbits = vbits;
        h.ptr.assign((size_t)n + 1, 0);
        for (int64_t r = 0; r < n; ++r) h.ptr[r + 1] = h.ptr[r] + (ptr[order[r] + 1] - ptr[order[r]]);
        h.idx.resize((size_t)h.ptr[n]);
        h.val.resize((size_t)h.ptr[n]);
        for (int64_t r = 0; r < n; ++r) {
            const int64_t b = ptr[order[r]], len = ptr[order[r] + 1] - b;
            std::copy(idx + b, idx + b + len, h.idx.begin() + h.ptr[r]);
            std::copy(val + b, val + b + len, h.val.begin() + h.ptr[r]);
            h.max_nnz = std::max(h.max_nnz, (int32_t)len);
        }
    };
    {
        HostFamily hp, hc;
        gather(p_rowptr, p_idx, p_val, p_dim, p_vbits, hp);
        gather(c_rowptr, c_idx, c_val, c_dim, c_vbits, hc);
        if (!new_of_old.empty()) {
            // the device image of the place family in the renumbered dimensions, rows re-sorted by the
            // new index; hp itself keeps the caller's indices (the default ratings below use them)
            HostFamily hq = hp;
            std::vector<std::pair<int32_t, double>> tmp;
            for (int64_t r = 0; r < n; ++r) {
                tmp.clear();
                for (int64_t e = hp.ptr[r]; e < hp.ptr[r + 1]; ++e) tmp.emplace_back(new_of_old[hp.idx[e]], hp.val[e]);
                std::sort(tmp.begin(), tmp.end());
                for (size_t j = 0; j < tmp.size(); ++j) {
                    hq.idx[hp.ptr[r] + j] = tmp[j].first;
                    hq.val[hp.ptr[r] + j] = tmp[j].second;
                }
            }
            LOCREC_TRY(build_family_device(ix.get(), hq, ix->fp, ix->packed));
            if (want_ht) {
                LOCREC_TRY(build_ht(ix.get(), hq, hc, ht_h, ht_qt));
                lap("head / tail image");
            }
            // leading element groups (dwordx4 = 4 elements) that are popular in EVERY lane of the slice;
            // padding elements are index 0, which is popular
            std::vector<int32_t> split((size_t)ix->nslices, 0);
            for (int32_t sl = 0; sl < ix->nslices; ++sl) {
                int w = 0, g = INT32_MAX;
                for (int64_t r = (int64_t)sl * 64; r < std::min<int64_t>(n, (int64_t)sl * 64 + 64); ++r) {
                    const int len = (int)(hq.ptr[r + 1] - hq.ptr[r]);
                    w = std::max(w, len);
                    int np_r = 0;
                    while (np_r < len && hq.idx[hq.ptr[r] + np_r] < pop_h) ++np_r;
                    if (np_r < len) g = std::min(g, np_r / 4);
                }
                split[sl] = std::min(g, ((w + 3) & ~3) / 4);
            }
            LOCREC_TRY(ix->fp.sell_split.upload(split, ix->stream));
            LOCREC_HIP_TRY(hipStreamSynchronize(ix->stream));
            ix->fp.pop_h = pop_h;
            ix->fp.scan_bytes += (int64_t)ix->nslices * 4;
        } else {
            LOCREC_TRY(build_family_device(ix.get(), hp, ix->fp, ix->packed));
        }
        LOCREC_TRY(build_family_device(ix.get(), hc, ix->fc, ix->packed));
        lap("families (gather, SELL, upload)");
        // ratings CSR in row order
        std::vector<int64_t> rp((size_t)n + 1, 0), rplace;
        std::vector<double> rrating;
        if (r_rowptr) {
            if (n > 0 && (!r_place || !r_rating)) return fail(LOCREC_E_INVALID_ARG, "NULL ratings array");
            for (int64_t r = 0; r < n; ++r) {
                const int64_t len = r_rowptr[order[r] + 1] - r_rowptr[order[r]];
                if (len < 0) return fail(LOCREC_E_INVALID_ARG, "ratings rowptr not monotone");
                rp[r + 1] = rp[r] + len;
            }
            rplace.resize((size_t)rp[n]);
            rrating.resize((size_t)rp[n]);
            for (int64_t r = 0; r < n; ++r) {
                const int64_t b = r_rowptr[order[r]];
                for (int64_t e = 0; e < rp[r + 1] - rp[r]; ++e) {
                    rplace[rp[r] + e] = r_place[b + e];
                    rrating[rp[r] + e] = (double)r_rating[b + e];  // Long * Double promotes (:59)
                }
            }
        } else {
            rp = hp.ptr;
            rplace.resize(hp.idx.size());
            for (size_t e = 0; e < hp.idx.size(); ++e) rplace[e] = hp.idx[e];
            rrating = hp.val;
        }
        for (int64_t r = 0; r < n; ++r) ix->max_r_nnz = std::max(ix->max_r_nnz, rp[r + 1] - rp[r]);
        {
            // place-major transpose of the ratings (rows ascending inside a place: a fixed order)
            std::vector<int64_t> &cpl = ix->cplace_ids;
            std::vector<int32_t> pidx_of(rplace.size());
            int64_t mn = 0, mx = -1;
            if (!rplace.empty()) {
                const auto mm = std::minmax_element(rplace.begin(), rplace.end());
                mn = *mm.first;
                mx = *mm.second;
            }
            if (!rplace.empty() && mx - mn < ((int64_t)1 << 26)) {
                // place ids span a moderate range (they do in the reference: one global id space):
                // distinct ids and their ranks from a presence table, no 25 M-element sort
                std::vector<int32_t> rank((size_t)(mx - mn + 1), 0);
                for (const int64_t pl : rplace) rank[(size_t)(pl - mn)] = 1;
                int32_t acc = 0;
                cpl.clear();
                for (size_t i = 0; i < rank.size(); ++i) {
                    if (rank[i]) {
                        rank[i] = acc++;
                        cpl.push_back(mn + (int64_t)i);
                    } else {
                        rank[i] = -1;
                    }
                }
                for (size_t e = 0; e < rplace.size(); ++e) pidx_of[e] = rank[(size_t)(rplace[e] - mn)];
            } else {
                cpl = rplace;
                std::sort(cpl.begin(), cpl.end());
                cpl.erase(std::unique(cpl.begin(), cpl.end()), cpl.end());
                for (size_t e = 0; e < rplace.size(); ++e)
                    pidx_of[e] = (int32_t)(std::lower_bound(cpl.begin(), cpl.end(), rplace[e]) - cpl.begin());
            }
            const int64_t ncp = (int64_t)cpl.size();
            std::vector<int64_t> cptr((size_t)ncp + 1, 0);
            for (size_t e = 0; e < rplace.size(); ++e) ++cptr[pidx_of[e] + 1];
            for (int64_t i = 0; i < ncp; ++i) cptr[i + 1] += cptr[i];
            std::vector<int64_t> cur(cptr.begin(), cptr.end() - 1);
            std::vector<int32_t> crow(rplace.size());
            std::vector<double> crat(rplace.size());
            for (int64_t r = 0; r < n; ++r)
                for (int64_t e = rp[r]; e < rp[r + 1]; ++e) {
                    const int64_t pos = cur[pidx_of[e]]++;
                    crow[pos] = (int32_t)r;
                    crat[pos] = rrating[e];
                }
            LOCREC_TRY(ix->r_pidx.upload(pidx_of, ix->stream));
            LOCREC_TRY(ix->cplace_dev.upload(cpl, ix->stream));
            LOCREC_TRY(ix->cp_ptr.upload(cptr, ix->stream));
            LOCREC_TRY(ix->cp_row.upload(crow, ix->stream));
            LOCREC_TRY(ix->cp_rating.upload(crat, ix->stream));
            LOCREC_HIP_TRY(hipStreamSynchronize(ix->stream));
        }
        LOCREC_TRY(ix->r_ptr.upload(rp, ix->stream));
        LOCREC_TRY(ix->r_place.upload(rplace, ix->stream));
        LOCREC_TRY(ix->r_rating.upload(rrating, ix->stream));
        LOCREC_HIP_TRY(hipStreamSynchronize(ix->stream));
    }
    lap("ratings (CSR + transpose)");
    // ---- rid: rank of each row's person id (tie-break person_id asc)
    {
        const std::vector<int32_t> &by_id = ix->row_by_rank;
        std::vector<uint32_t> rid((size_t)n);
        std::vector<int64_t> ids_sorted((size_t)n);
        for (int64_t k = 0; k < n; ++k) {
            rid[by_id[k]] = (uint32_t)k;
            ids_sorted[k] = ix->ids_row[by_id[k]];
        }
        LOCREC_TRY(ix->rid.upload(rid, ix->stream));
        if (ix->ht.ready) {  // the same ranks padded to whole slices (knn_scan_ht loads them unconditionally)
            rid.resize((size_t)ix->nslices * 64, 0u);
            LOCREC_TRY(ix->ht.rid.upload(rid, ix->stream));
        }
        LOCREC_TRY(ix->ids_by_rank.upload(ids_sorted, ix->stream));
        LOCREC_TRY(ix->row_of_rid.upload(by_id, ix->stream));
        LOCREC_HIP_TRY(hipStreamSynchronize(ix->stream));
    }
    lap("rid + norms + sync");
    *out = ix.release();
    return LOCREC_OK;
} LOCREC_CATCH_ALL

static_assert(sizeof(HtCold) <= (size_t)cfg::kHtColdBytes, "cfg::kHtColdBytes is the size of the device buffer that holds a launch's HtCold");
static_assert(kHtNP == cfg::kHtNP && kHtCatRows == cfg::kHtCatRows, "knn_ht.h and knn_index.h disagree");

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
    if (std::getenv("LOCREC_KNN_HOST_BUILD"))
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
    auto lds_of = [&](int cap) { return (size_t)cap * 24 + (size_t)K * 16 + (size_t)(K + 1) * 4 + 16; };
    const int M1 = std::min(M, 2048);
    const int threads1 = (M1 < M && K <= 512) ? 512 : kAggThreads;
    const size_t lds_full = lds_of(M);
    if (lds_full > 64 * 1024)
        LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_aggregate),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_full));
    hipLaunchKernelGGL(knn_aggregate, dim3((unsigned)nq), dim3(threads1), lds_of(M1), s, ix->out_rows.p, ix->out_sims.p,
                       ix->out_cnt.p, K, ix->r_ptr.p, ix->r_pidx.p, ix->r_rating.p, ix->cplace_dev.p, M1,
                       ix->agg_place.p, ix->agg_est.p, ix->agg_n.p, ix->agg_overflow.p, (int64_t)M, 0);
    if (M1 < M)
        hipLaunchKernelGGL(knn_aggregate, dim3((unsigned)nq), dim3(kAggThreads), lds_full, s, ix->out_rows.p,
                           ix->out_sims.p, ix->out_cnt.p, K, ix->r_ptr.p, ix->r_pidx.p, ix->r_rating.p,
                           ix->cplace_dev.p, M, ix->agg_place.p, ix->agg_est.p, ix->agg_n.p, ix->agg_overflow.p,
                           (int64_t)M, 1);
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
        const size_t lds = (size_t)M * 24 + (size_t)K * 16 + (size_t)(K + 1) * 4 + 16;
        if (lds > 64 * 1024)
            LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_aggregate),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(knn_aggregate, dim3(1), dim3(kAggThreads), lds, s, ix->out_rows.p, ix->out_sims.p,
                           ix->out_cnt.p, K, ix->r_ptr.p, ix->r_pidx.p, ix->r_rating.p, ix->cplace_dev.p, M,
                           ix->agg_place.p, ix->agg_est.p, ix->agg_n.p, ix->agg_overflow.p, (int64_t)M, 0);
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
    const size_t lds = (size_t)M * 24 + (size_t)K * 16 + (size_t)(K + 1) * 4 + 16;
    if (lds > 64 * 1024)
        LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(knn_aggregate),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // a request on the stream path: knn_aggregate writes its result into the pinned staging buffer itself
    unsigned char *agg_host = nullptr;
    if (ix->single_pending && !ix->last_scan_fast && (size_t)M * 16 + 64 <= locrec_knn_index::kStageBytes && stage_of(ix) &&
        ix->h_stage_dev && !ix->no_pack)
        agg_host = ix->h_stage_dev;
    hipLaunchKernelGGL(knn_aggregate, dim3(1), dim3(kAggThreads), lds, s, ix->out_rows.p, ix->out_sims.p,
                       ix->out_cnt.p, K, ix->r_ptr.p, ix->r_pidx.p, ix->r_rating.p, ix->cplace_dev.p, M,
                       ix->agg_place.p, ix->agg_est.p, ix->agg_n.p, ix->agg_overflow.p, (int64_t)M, 0, agg_host,
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
