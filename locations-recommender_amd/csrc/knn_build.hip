// knn_build.hip -- the KNN index is built ON THE DEVICE (the index build is the
// last stage of the rating-vector pipeline, RatingVectorsBuilder.scala:52-84 -> the device layouts of
// knn.hip / knn_ht.h).  Inputs are device arrays (locrec_knn_create_from_device; locrec_knn_create
// uploads the caller's host arrays and comes here too), outputs are the same images the scans read:
//
//   validation          one thread per row: SparseVector invariants (RatingVectorsBuilder.scala:74-77)
//   popularity          place frequencies (atomics), dimensions sorted by (frequency desc, index asc)
//   row order           stable radix sort of the rows by (nnz_place, nnz_category, head count)
//   id rank             radix sort of the person ids -> rid / ids_by_rank / row_of_rid (tie-break H1)
//   place CSR           elements keyed (row, renumbered index) and radix-sorted: rows come out in row
//                       order with ascending renumbered indices in ONE pass over all elements
//   SELL-64 images      packed / generic rows, slice tables, popular-prefix table, head / tail image
//   postings            tail elements keyed (place, row) and radix-sorted
//   ratings             gathered by row; place-major transpose through one more keyed sort
//
// The sorts and scans are rocPRIM device primitives called directly (dev_prims.h): this is the offline build step,
// not a hot path (VERDICT r01 item 6 allows a library call here); every other step is a kernel in this file.
// The host keeps only what request PLANNING needs (row lengths, tail sizes, the id -> row lookup).

#include "dev_prims.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <memory>
#include <new>
#include <numeric>

#include "knn_index.h"

namespace {

using namespace locrec;

// ---- small helpers ---------------------------------------------------------------------------

struct Temp {  // rocPRIM temporary storage, reused
    DevBuf<unsigned char> buf;
};

#define KB_PRIM(tmp, stream, call_with_args)                                           \
    do {                                                                              \
        size_t bytes_ = 0;                                                            \
        void *p_ = nullptr;                                                           \
        LOCREC_HIP_TRY((call_with_args));                                             \
        LOCREC_TRY((tmp).buf.reserve(bytes_ + 256));                                  \
        p_ = (tmp).buf.p;                                                             \
        LOCREC_HIP_TRY((call_with_args));                                             \
    } while (0)

int ceil_log2_64(int64_t v)
{
    int l = 0;
    while (((int64_t)1 << l) < v) ++l;
    return l;
}

dim3 grid_for(int64_t n, int threads = 256) { return dim3((unsigned)std::max<int64_t>(1, (n + threads - 1) / threads)); }

// ---- validation ------------------------------------------------------------------------------

struct CheckOut {
    unsigned long long first_error;  // row << 8 | code, minimum over all failing rows (~0 = none)
    unsigned long long vmax_bits;    // max |value| (bit pattern of a non-negative double orders like the value)
    unsigned long long ssmax_bits;   // max over rows of sum v^2
    int non_integral;                // a value that is not an integer count >= 1
    int bad_index;                   // the offending index of error code 2
    // the same two maxima over the rows that are NOT "wide" (see kb_validate): what the head / tail form may assume
    // when the wide rows are kept out of its images (per-row format fallback)
    unsigned long long lvmax_bits, lssmax_bits;
};

enum { kErrPtr = 1, kErrRange = 2, kErrOrder = 3, kErrFinite = 4, kErrZero = 5, kErrLong = 6 };

// wide[r] (OR-ed over both families): the row by ITSELF breaks the head / tail form's legality - an integer count of 256
// or more (a person with 256 visits to one place: the element format holds a byte) or a sum of squares of 65,536 or
// more (a u16 dot may overflow).  Such rows are few in real visit data; they are kept OUT of the packed images and
// scored by the side kernels of knn.hip instead of demoting the whole index.
__global__ void kb_validate(int64_t n, const int64_t *ptr, const int32_t *idx, const double *val, int32_t dim, CheckOut *out,
                            unsigned char *wide)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t b = ptr[r], e = ptr[r + 1];
    unsigned code = 0;
    int bad = 0;
    double ss = 0.0, vmax = 0.0;
    bool integral = true;
    if (e < b) {
        code = kErrPtr;
    } else if (e - b >= (1 << 21)) {
        code = kErrLong;  // (the row-order sort key packs three 21-bit counts)
    } else {
        int32_t prev = -1;
        for (int64_t i = b; i < e && !code; ++i) {
            const int32_t ix = idx[i];
            const double v = val[i];
            if (ix < 0 || ix >= dim) {
                code = kErrRange;
                bad = ix;
            } else if (i > b && ix <= prev) {
                code = kErrOrder;
            } else if (!isfinite(v)) {
                code = kErrFinite;
            }
            prev = ix;
            if (!(v >= 1.0) || v != floor(v)) integral = false;
            vmax = fmax(vmax, fabs(v));
            ss += v * v;
        }
        if (!code && e > b && !(ss > 0)) code = kErrZero;
    }
    if (code) {
        const unsigned long long packed = ((unsigned long long)r << 8) | code;
        const unsigned long long old = atomicMin(&out->first_error, packed);
        if (packed < old && code == kErrRange) out->bad_index = bad;  // (best effort: the message's detail)
    }
    if (!integral) out->non_integral = 1;
    atomicMax(&out->vmax_bits, (unsigned long long)__double_as_longlong(vmax));
    atomicMax(&out->ssmax_bits, (unsigned long long)__double_as_longlong(ss));
    if (vmax >= 256.0 || ss >= 65536.0) {
        if (wide) wide[r] = 1;
    } else {
        atomicMax(&out->lvmax_bits, (unsigned long long)__double_as_longlong(vmax));
        atomicMax(&out->lssmax_bits, (unsigned long long)__double_as_longlong(ss));
    }
}

// wide flags from input order to row order, and the list of wide rows
__global__ void kb_wide_rows(int64_t n, const uint32_t *order, const unsigned char *wide_in, unsigned char *wide_row)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) wide_row[r] = wide_in[order[r]];
}

// ---- popularity ------------------------------------------------------------------------------

// Place frequencies.  Visits are Zipf-distributed: one global atomic per element made the few popular
// places serialise the whole pass (12.8 ms for 25 M elements).  Every block therefore counts its chunk of
// elements in a small tagged table in LDS - slot = index mod 4096, the first index to arrive owns the slot,
// a later index that finds the slot taken goes to the global counter directly - and flushes the table once.
constexpr int kHistSlots = 4096;
constexpr int kHistChunk = 1 << 16;  // elements per block

__global__ __launch_bounds__(256) void kb_hist(int64_t ne, const int32_t *idx, uint32_t *freq)
{
    __shared__ uint32_t tag[kHistSlots], cnt[kHistSlots];
    for (int i = threadIdx.x; i < kHistSlots; i += blockDim.x) {
        tag[i] = 0xFFFFFFFFu;
        cnt[i] = 0u;
    }
    __syncthreads();
    const int64_t b = (int64_t)blockIdx.x * kHistChunk, e = min(ne, b + kHistChunk);
    for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        const uint32_t d = (uint32_t)idx[i];
        const uint32_t slot = d & (kHistSlots - 1);
        uint32_t t = tag[slot];
        if (t == 0xFFFFFFFFu) {
            const uint32_t old = atomicCAS(&tag[slot], 0xFFFFFFFFu, d);
            t = old == 0xFFFFFFFFu ? d : old;
        }
        if (t == d) atomicAdd(&cnt[slot], 1u);
        else atomicAdd(&freq[d], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kHistSlots; i += blockDim.x)
        if (cnt[i]) atomicAdd(&freq[tag[i]], cnt[i]);
}

__global__ void kb_dim_keys(int32_t dim, const uint32_t *freq, uint64_t *keys)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < dim) keys[i] = ((uint64_t)(~freq[i]) << 32) | (uint32_t)i;  // frequency desc, index asc
}

__global__ void kb_new_of_old(int32_t dim, const uint64_t *sorted, int32_t *new_of_old, uint32_t *freq_new)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dim) return;
    const uint32_t d = (uint32_t)sorted[i];
    new_of_old[d] = i;
    freq_new[i] = ~(uint32_t)(sorted[i] >> 32);
}

// ---- row order -------------------------------------------------------------------------------

// key = nnz_place << 42 | nnz_category << 21 | (count of indices that become < key_h); value = input row
__global__ void kb_row_keys(int64_t n, const int64_t *p_ptr, const int32_t *p_idx, const int64_t *c_ptr,
                            const int32_t *new_of_old, int32_t key_h, uint64_t *keys, uint32_t *vals)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t b = p_ptr[r], e = p_ptr[r + 1];
    uint64_t pop = 0;
    if (new_of_old)
        for (int64_t i = b; i < e; ++i) pop += new_of_old[p_idx[i]] < key_h ? 1 : 0;
    keys[r] = ((uint64_t)(e - b) << 42) | ((uint64_t)(c_ptr[r + 1] - c_ptr[r]) << 21) | pop;
    vals[r] = (uint32_t)r;
}

__global__ void kb_apply_order(int64_t n, const uint64_t *keys_sorted, const uint32_t *order, const int64_t *ids,
                               const int64_t *r_ptr, int64_t *ids_row, int32_t *row_of_input, int32_t *nnz_p, int32_t *nnz_c,
                               int32_t *npop, int64_t *len_p, int64_t *len_c, int64_t *len_r, uint64_t *id_keys, uint32_t *id_vals)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint64_t k = keys_sorted[r];
    const uint32_t src = order[r];
    const int64_t id = ids[src];
    ids_row[r] = id;
    row_of_input[src] = (int32_t)r;
    const int32_t np = (int32_t)(k >> 42), nc = (int32_t)((k >> 21) & 0x1FFFFF);
    nnz_p[r] = np;
    nnz_c[r] = nc;
    npop[r] = (int32_t)(k & 0x1FFFFF);
    len_p[r] = np;
    len_c[r] = nc;
    if (len_r) len_r[r] = r_ptr ? r_ptr[src + 1] - r_ptr[src] : np;
    id_keys[r] = (uint64_t)id ^ 0x8000000000000000ull;  // signed order as unsigned order
    id_vals[r] = (uint32_t)r;
}

__global__ void kb_rank_ids(int64_t n, const uint64_t *id_keys_sorted, const uint32_t *rows_sorted, uint32_t *rid,
                            int64_t *ids_by_rank, int32_t *row_of_rid, unsigned long long *dup)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t row = rows_sorted[k];
    rid[row] = (uint32_t)k;
    ids_by_rank[k] = (int64_t)(id_keys_sorted[k] ^ 0x8000000000000000ull);
    row_of_rid[k] = (int32_t)row;
    if (k > 0 && id_keys_sorted[k] == id_keys_sorted[k - 1]) atomicMin(dup, (unsigned long long)k);
}

// ---- CSR in row order -------------------------------------------------------------------------

// elements of row r (input row order[r]) -> keys (r << 32 | new index) and the source position as payload
__global__ void kb_place_keys(int64_t n, const uint32_t *order, const int64_t *in_ptr, const int64_t *out_ptr,
                              const int32_t *in_idx, const int32_t *new_of_old, uint64_t *keys, uint32_t *src_pos,
                              int32_t *orig_idx /* optional: the caller's indices in row order */)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t b = in_ptr[order[r]], len = in_ptr[order[r] + 1] - b, o = out_ptr[r];
    for (int64_t j = 0; j < len; ++j) {
        const int32_t ix = in_idx[b + j];
        keys[o + j] = ((uint64_t)r << 32) | (uint32_t)(new_of_old ? new_of_old[ix] : ix);
        src_pos[o + j] = (uint32_t)(b + j);
        if (orig_idx) orig_idx[o + j] = ix;
    }
}

__global__ void kb_finish_csr(int64_t ne, const uint64_t *keys, const uint32_t *src_pos, const double *in_val, int32_t *csr_idx,
                              double *csr_val)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ne) return;
    csr_idx[i] = (int32_t)(uint32_t)keys[i];
    csr_val[i] = in_val[src_pos[i]];
}

// Distance.vectorLength (Distance.scala:11-16): left fold of v*v from 0.0, then sqrt -- the same arithmetic as knn_norms
__global__ void kb_norms(const int64_t *ptr, const double *val, int64_t n, double *norm, float *inorm32)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double sum = 0.0;
    for (int64_t e = ptr[r]; e < ptr[r + 1]; ++e) {
        const double sq = val[e] * val[e];
        sum = sum + sq;
    }
    const double len = sqrt(sum);
    norm[r] = len;
    inorm32[r] = len > 0.0 ? (float)(1.0 / len) : 0.0f;
}

// ---- SELL-64 images ---------------------------------------------------------------------------

// width (elements per lane) of every slice: the longest of its 64 rows, optionally rounded up to 4
__global__ void kb_slice_width(int64_t n, int32_t nslices, const int32_t *len, int32_t round4, int32_t *w, int64_t *w64)
{
    const int sl = blockIdx.x * blockDim.x + threadIdx.x;
    if (sl >= nslices) return;
    int m = 0;
    for (int64_t r = (int64_t)sl * 64; r < min(n, (int64_t)sl * 64 + 64); ++r) m = max(m, len[r]);
    if (round4) m = (m + 3) & ~3;
    w[sl] = m;
    w64[sl] = (int64_t)m * 64;
}

// PACKED rows: index << vbits | value at [slice][j / 4][lane][4]; limit (optional) caps a row's elements
// HT rows (ht_rsh > 0): value << 16 | index << ht_rsh
// skip (optional): rows flagged there stay all padding (index 0, value 0): the wide rows of the per-row fallback
__global__ void kb_sell_packed(int64_t n, const int64_t *ptr, const int32_t *idx, const double *val, const int32_t *limit,
                               const int64_t *off, int32_t vbits, int32_t ht_rsh, uint32_t *sell, const unsigned char *skip)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || (skip && skip[r])) return;
    const int64_t b = ptr[r];
    const int len = limit ? limit[r] : (int)(ptr[r + 1] - b);
    const int64_t base = off[r >> 6] + (r & 63) * 4;
    for (int j = 0; j < len; ++j) {
        const uint32_t ix = (uint32_t)idx[b + j], v = (uint32_t)val[b + j];
        sell[base + (int64_t)(j >> 2) * 256 + (j & 3)] = ht_rsh ? (v << 16) | (ix << ht_rsh) : (ix << vbits) | v;
    }
}

__global__ void kb_sell_generic(int64_t n, const int64_t *ptr, const int32_t *idx, const double *val, const int64_t *off,
                                uint32_t *sell, double *sval)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t b = ptr[r], len = ptr[r + 1] - b, base = off[r >> 6] + (r & 63);
    for (int64_t j = 0; j < len; ++j) {
        sell[base + j * 64] = (uint32_t)idx[b + j];
        sval[base + j * 64] = val[b + j];
    }
}

// leading dwordx4 element groups of a slice in which EVERY lane holds popular (< pop_h) indices only
__global__ void kb_split(int64_t n, int32_t nslices, const int64_t *ptr, const int32_t *idx, int32_t pop_h, const int32_t *w,
                         int32_t *split)
{
    const int sl = blockIdx.x * blockDim.x + threadIdx.x;
    if (sl >= nslices) return;
    int g = INT32_MAX;
    for (int64_t r = (int64_t)sl * 64; r < min(n, (int64_t)sl * 64 + 64); ++r) {
        const int64_t b = ptr[r];
        const int len = (int)(ptr[r + 1] - b);
        int np = 0;
        while (np < len && idx[b + np] < pop_h) ++np;
        if (np < len) g = min(g, np / 4);
    }
    split[sl] = min(g, w[sl] / 4);
}

// the wide rows' vectors lane-major for the side kernels (knn_index.h, side_p / side_c)
__global__ void kb_side_fill(int32_t nw, const int32_t *wide_rows, const int64_t *ptr, const int32_t *idx, const double *val,
                             const int32_t *off, int2 *out)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw) return;
    const int32_t r = wide_rows[w];
    const int64_t b = ptr[r];
    const int len = (int)(ptr[r + 1] - b);
    int2 *o = out + off[w >> 6] + (w & 63);
    for (int j = 0; j < len; ++j) o[(int64_t)j * 64] = make_int2(idx[b + j], (int)val[b + j]);
}

// ---- head / tail image ------------------------------------------------------------------------

// the per-row fallback keeps its wide rows out of the postings: their elements leave the place frequencies, from which
// the posting lists' sizes and every query row's hit count are derived
__global__ void kb_freq_drop_rows(int64_t n, const int64_t *p_ptr, const int32_t *p_idx, const unsigned char *skip, uint32_t *freq_new)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || !skip[r]) return;
    for (int64_t i = p_ptr[r]; i < p_ptr[r + 1]; ++i) atomicSub(&freq_new[p_idx[i]], 1u);
}

__global__ void kb_ht_rows(int64_t n, const int64_t *p_ptr, const int32_t *p_idx, const double *p_val, const int64_t *c_ptr,
                           const double *c_val, int32_t h, const uint32_t *freq_new, int32_t *nhead, int32_t *tail_nnz,
                           int64_t *tail_len, int64_t *tail_hits, uint32_t *ss, const unsigned char *skip)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    if (skip && skip[r]) {  // a wide row: no head elements, no postings, ss = 0 ("no row" to knn_scan_ht and knn_scan1_direct8)
        nhead[r] = 0;
        tail_nnz[r] = 0;
        tail_len[r] = 0;
        tail_hits[r] = 0;
        ss[r] = 0u;
        return;
    }
    const int64_t b = p_ptr[r], e = p_ptr[r + 1];
    int nh = 0;
    int64_t hits = 0;
    double sp = 0.0, sc = 0.0;
    for (int64_t i = b; i < e; ++i) {
        const int32_t ix = p_idx[i];
        if (ix < h) ++nh; else hits += freq_new[ix];
        sp += p_val[i] * p_val[i];
    }
    for (int64_t i = c_ptr[r]; i < c_ptr[r + 1]; ++i) sc += c_val[i] * c_val[i];
    nhead[r] = nh;
    tail_nnz[r] = (int32_t)(e - b) - nh;
    tail_len[r] = (e - b) - nh;
    tail_hits[r] = hits;
    ss[r] = (uint32_t)sp | ((uint32_t)sc << 16);
}

__global__ void kb_tail_keys(int64_t n, const int64_t *p_ptr, const int32_t *p_idx, const double *p_val, const int32_t *nhead,
                             const int64_t *tail_ptr, int32_t h, uint64_t *keys, uint32_t *vals, const unsigned char *skip)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || (skip && skip[r])) return;
    const int64_t b = p_ptr[r] + nhead[r], e = p_ptr[r + 1], o = tail_ptr[r];
    for (int64_t i = b; i < e; ++i) {
        keys[o + (i - b)] = ((uint64_t)(uint32_t)(p_idx[i] - h) << 32) | (uint32_t)r;  // (place, row)
        vals[o + (i - b)] = (uint32_t)p_val[i];
    }
}

__global__ void kb_postings(int64_t nt, const uint64_t *keys, const uint32_t *vals, uint32_t *post)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nt) post[i] = ((uint32_t)keys[i] << 8) | vals[i];
}

__global__ void kb_tail_freq(int64_t ntail, const uint32_t *freq_new, int32_t h, int64_t *cnt)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ntail) cnt[t] = freq_new[h + t];
}

__global__ void kb_desc(int32_t nslices, const int64_t *off_p, const int32_t *w_p, const int64_t *off_c, const int32_t *w_c,
                        uint4 *desc)
{
    const int sl = blockIdx.x * blockDim.x + threadIdx.x;
    if (sl >= nslices) return;
    desc[sl] = make_uint4((uint32_t)(off_p[sl] / 4), (uint32_t)(off_c[sl] / 4), (uint32_t)(w_p[sl] / 4) | ((uint32_t)(w_c[sl] / 4) << 16), 0u);
}

__global__ void kb_pad_rid(int64_t n, int64_t npad, const uint32_t *rid, uint32_t *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) out[i] = i < n ? rid[i] : 0u;
}

// ---- ratings ----------------------------------------------------------------------------------

__global__ void kb_gather_ratings(int64_t n, const uint32_t *order, const int64_t *in_ptr, const int64_t *out_ptr,
                                  const int64_t *in_place, const int64_t *in_rating, int64_t *place, double *rating)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t b = in_ptr[order[r]], len = in_ptr[order[r] + 1] - b, o = out_ptr[r];
    for (int64_t j = 0; j < len; ++j) {
        place[o + j] = in_place[b + j];
        rating[o + j] = (double)in_rating[b + j];  // Long * Double promotes (KnnRecommender.scala:59)
    }
}

__global__ void kb_default_ratings(int64_t ne, const int32_t *orig_idx, int64_t *place)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ne) place[i] = orig_idx[i];
}

__global__ void kb_place_sortkeys(int64_t ne, const int64_t *place, uint64_t *keys)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ne) keys[i] = (uint64_t)place[i] ^ 0x8000000000000000ull;
}

__global__ void kb_unbias(int64_t nc, const uint64_t *keys, int64_t *ids)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nc) ids[i] = (int64_t)(keys[i] ^ 0x8000000000000000ull);
}

// pidx_of[e] = index of the rating's place among the distinct places; transpose keys (place index, row)
__global__ void kb_place_index(int64_t n, const int64_t *r_ptr, const int64_t *place, const int64_t *cplace, int64_t ncp,
                               int32_t *pidx_of, uint64_t *tkeys, uint32_t *tvals)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    for (int64_t e = r_ptr[r]; e < r_ptr[r + 1]; ++e) {
        const int64_t pl = place[e];
        int64_t a = 0, b = ncp;
        while (b - a > 1) {
            const int64_t m = (a + b) >> 1;
            if (cplace[m] <= pl) a = m; else b = m;
        }
        pidx_of[e] = (int32_t)a;
        tkeys[e] = ((uint64_t)(uint32_t)a << 32) | (uint32_t)r;
        tvals[e] = (uint32_t)e;
    }
}

__global__ void kb_transpose_out(int64_t ne, const uint64_t *tkeys, const uint32_t *tvals, const double *rating, int32_t *cp_row,
                                 double *cp_rating)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ne) return;
    cp_row[i] = (int32_t)(uint32_t)tkeys[i];
    cp_rating[i] = rating[tvals[i]];
}

// cp_ptr[p] = first position of place index p in the sorted transpose keys
__global__ void kb_cp_ptr(int64_t ncp, int64_t ne, const uint64_t *tkeys, int64_t *cp_ptr)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p > ncp) return;
    const uint64_t want = (uint64_t)p << 32;
    int64_t a = 0, b = ne;
    while (a < b) {
        const int64_t m = (a + b) >> 1;
        if (tkeys[m] < want) a = m + 1; else b = m;
    }
    cp_ptr[p] = a;
}

struct MaxI64 {
    __host__ __device__ int64_t operator()(int64_t a, int64_t b) const { return a > b ? a : b; }
};

// one family's images from its CSR in row order (device)
int32_t build_family(locrec_knn_index *ix, DevFamily &d, int32_t dim, int32_t vbits, bool packed, const DevBuf<int32_t> &nnz_dev,
                     Temp &tmp, const unsigned char *skip)
{
    const int64_t n = ix->n;
    const int32_t nslices = ix->nslices;
    hipStream_t s = ix->stream;
    d.dim = dim;
    d.vbits = vbits;
    DevBuf<int64_t> w64;
    LOCREC_TRY(d.sell_w.alloc((size_t)nslices));
    LOCREC_TRY(w64.alloc((size_t)nslices + 1));
    LOCREC_TRY(d.sell_off.alloc((size_t)nslices + 1));
    LOCREC_HIP_TRY(hipMemsetAsync(w64.p, 0, ((size_t)nslices + 1) * 8, s));
    if (nslices > 0)
        hipLaunchKernelGGL(kb_slice_width, grid_for(nslices), dim3(256), 0, s, n, nslices, nnz_dev.p, packed ? 1 : 0, d.sell_w.p, w64.p);
    KB_PRIM(tmp, s, prim::exclusive_sum(p_, bytes_, w64.p, d.sell_off.p, nslices + 1, s));
    int64_t total = 0;
    LOCREC_HIP_TRY(hipMemcpyAsync(&total, d.sell_off.p + nslices, 8, hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    LOCREC_TRY(d.sell.alloc((size_t)total));
    LOCREC_HIP_TRY(hipMemsetAsync(d.sell.p, 0, (size_t)std::max<int64_t>(1, total) * 4, s));
    if (packed) {
        if (n > 0)
            hipLaunchKernelGGL(kb_sell_packed, grid_for(n), dim3(256), 0, s, n, d.csr_ptr.p, d.csr_idx.p, d.csr_val.p, nullptr,
                               d.sell_off.p, vbits, 0, d.sell.p, skip);
    } else {
        LOCREC_TRY(d.sell_val.alloc((size_t)total));
        LOCREC_HIP_TRY(hipMemsetAsync(d.sell_val.p, 0, (size_t)std::max<int64_t>(1, total) * 8, s));
        if (n > 0)
            hipLaunchKernelGGL(kb_sell_generic, grid_for(n), dim3(256), 0, s, n, d.csr_ptr.p, d.csr_idx.p, d.csr_val.p, d.sell_off.p,
                               d.sell.p, d.sell_val.p);
    }
    LOCREC_TRY(d.norm.alloc((size_t)n));
    LOCREC_TRY(d.inorm32.alloc((size_t)n));
    if (n > 0) hipLaunchKernelGGL(kb_norms, grid_for(n), dim3(256), 0, s, d.csr_ptr.p, d.csr_val.p, n, d.norm.p, d.inorm32.p);
    d.scan_bytes = total * (packed ? 4 : 12) + (int64_t)nslices * 12 + n * (packed ? 4 : 8);
    LOCREC_HIP_TRY(hipGetLastError());
    return LOCREC_OK;
}

}  // namespace

namespace locrec {

// The whole build.  Every pointer is a DEVICE pointer valid on the current device; r_ptr may be null
// (ratings = the place vectors themselves, what RatingVectorsBuilderMain.scala:41-73 writes).
int32_t knn_build_device(int64_t n, const int64_t *ids, const int64_t *p_ptr, const int32_t *p_idx, const double *p_val,
                         int32_t p_dim, const int64_t *c_ptr, const int32_t *c_idx, const double *c_val, int32_t c_dim,
                         const int64_t *r_ptr, const int64_t *r_place, const int64_t *r_rating, locrec_knn_index **out)
{
    if (!out) return fail(LOCREC_E_INVALID_ARG, "out_index is NULL");
    *out = nullptr;
    if (n < 0 || n >= ((int64_t)1 << 31) - 64) return fail(LOCREC_E_INVALID_ARG, "bad person count");
    if (n > 0 && (!ids || !p_ptr || !c_ptr)) return fail(LOCREC_E_INVALID_ARG, "NULL input array");
    if (p_dim <= 0 || c_dim <= 0) return fail(LOCREC_E_INVALID_ARG, "vector sizes must be positive");
    LOCREC_TRY(ensure_device());
    std::unique_ptr<locrec_knn_index> ix(new (std::nothrow) locrec_knn_index);
    if (!ix) return fail(LOCREC_E_OOM, "host allocation failed");
    LOCREC_HIP_TRY(hipGetDevice(&ix->device));
    LOCREC_HIP_TRY(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
    ix->own_stream = true;
    hipStream_t s = ix->stream;
    ix->n = n;
    ix->nslices = (int32_t)((n + 63) / 64);
    ix->cand_slice0 = 0;
    ix->cand_slice1 = ix->nslices;
    knn_read_env(ix.get());
    const bool force_generic = ix->force_generic;
    const bool dbg_t = debug_env("LOCREC_DEBUG_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg_t) return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[locrec knn_build_device] %-30s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    Temp tmp;

    // ---- sizes of the element arrays: the last row pointers
    int64_t pe = 0, ce = 0, re = 0, p0 = 0, c0 = 0;
    if (n > 0) {
        LOCREC_HIP_TRY(hipMemcpyAsync(&pe, p_ptr + n, 8, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(&ce, c_ptr + n, 8, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(&p0, p_ptr, 8, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(&c0, c_ptr, 8, hipMemcpyDeviceToHost, s));
        if (r_ptr) LOCREC_HIP_TRY(hipMemcpyAsync(&re, r_ptr + n, 8, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
        if (p0 != 0) return fail(LOCREC_E_INVALID_ARG, "place rowptr must start at 0");
        if (c0 != 0) return fail(LOCREC_E_INVALID_ARG, "category rowptr must start at 0");
        if (pe < 0 || ce < 0 || re < 0 || pe >= ((int64_t)1 << 31) || ce >= ((int64_t)1 << 31) || re >= ((int64_t)1 << 31))
            return fail(LOCREC_E_INVALID_ARG, "element count out of range (the build sorts 32-bit positions)");
        if (r_ptr && re > 0 && (!r_place || !r_rating)) return fail(LOCREC_E_INVALID_ARG, "NULL ratings array");
    }

    // ---- validation (SparseVector invariants, RatingVectorsBuilder.scala:74-77)
    DevBuf<CheckOut> chk;
    LOCREC_TRY(chk.alloc(2));
    CheckOut init{~0ull, 0ull, 0ull, 0, 0, 0ull, 0ull};
    CheckOut h_chk[2] = {init, init};
    LOCREC_HIP_TRY(hipMemcpyAsync(chk.p, h_chk, sizeof h_chk, hipMemcpyHostToDevice, s));
    DevBuf<unsigned char> wide_in;  // per INPUT row: breaks the head / tail legality by itself (kb_validate)
    LOCREC_TRY(wide_in.alloc((size_t)std::max<int64_t>(1, n)));
    LOCREC_HIP_TRY(hipMemsetAsync(wide_in.p, 0, (size_t)std::max<int64_t>(1, n), s));
    if (n > 0) {
        hipLaunchKernelGGL(kb_validate, grid_for(n), dim3(256), 0, s, n, p_ptr, p_idx, p_val, p_dim, chk.p, wide_in.p);
        hipLaunchKernelGGL(kb_validate, grid_for(n), dim3(256), 0, s, n, c_ptr, c_idx, c_val, c_dim, chk.p + 1, wide_in.p);
    }
    LOCREC_HIP_TRY(hipMemcpyAsync(h_chk, chk.p, sizeof h_chk, hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    for (int f = 0; f < 2; ++f) {
        const char *name = f == 0 ? "place" : "category";
        if (h_chk[f].first_error == ~0ull) continue;
        const int64_t row = (int64_t)(h_chk[f].first_error >> 8);
        const int code = (int)(h_chk[f].first_error & 0xFF);
        int64_t pid = 0;
        LOCREC_HIP_TRY(hipMemcpy(&pid, ids + row, 8, hipMemcpyDeviceToHost));
        switch (code) {
        case kErrPtr: return fail(LOCREC_E_INVALID_ARG, "%s rowptr not monotone at %lld", name, (long long)row);
        case kErrRange:
            return fail(LOCREC_E_INVALID_ARG, "%s index %d out of range [0,%d)", name, h_chk[f].bad_index, f == 0 ? p_dim : c_dim);
        case kErrOrder: return fail(LOCREC_E_INVALID_ARG, "%s indices of person %lld not strictly ascending", name, (long long)pid);
        case kErrFinite: return fail(LOCREC_E_INVALID_ARG, "%s value is not finite", name);
        case kErrLong: return fail(LOCREC_E_INVALID_ARG, "%s vector of person %lld has 2^21 or more entries", name, (long long)pid);
        default: return fail(LOCREC_E_INVALID_ARG, "%s vector of person %lld has zero norm", name, (long long)pid);
        }
    }
    auto as_double = [](unsigned long long bits) {
        double d;
        std::memcpy(&d, &bits, 8);
        return d;
    };
    const bool integral = !h_chk[0].non_integral && !h_chk[1].non_integral;
    double pvmax = as_double(h_chk[0].vmax_bits), cvmax = as_double(h_chk[1].vmax_bits);
    double pss = as_double(h_chk[0].ssmax_bits), css = as_double(h_chk[1].ssmax_bits);
    // ---- per-row format fallback.  Counts are unbounded in the reference's data (RatingVectorsBuilder.scala:69
    // `rating.toDouble` of count("*")): ONE person with 256 visits to a place, or one row with a sum of squares of
    // 65,536, used to demote the WHOLE index from the head / tail form to PACK32 (3.6 x slower per pair).  When such
    // "wide" rows are few - at most max(256, n / 512), capped at 4096: the side kernels of knn.hip score every
    // (query, wide row) pair by merging two CSR rows - they are kept out of the packed images (all padding, ss = 0),
    // the rest of the index is judged on its own maxima, and knn.hip adds the wide rows back: as candidates through
    // knn_side_topk / knn_side_scan1, as queries through the dense CSR scan.  Integer data only: a non-integer value
    // still selects the GENERIC format for everybody (the reference's builders never produce one).
    std::vector<unsigned char> h_wide;
    int64_t n_wide = 0;
    const bool any_wide = pvmax >= 256.0 || cvmax >= 256.0 || pss >= 65536.0 || css >= 65536.0;
    if (any_wide && integral && !force_generic && !ix->no_row_fallback && !ix->no_pack16 && !ix->no_ht && n < ((int64_t)1 << 24) && c_dim <= cfg::kHtCatRows) {
        h_wide.resize((size_t)n);
        LOCREC_HIP_TRY(hipMemcpy(h_wide.data(), wide_in.p, (size_t)n, hipMemcpyDeviceToHost));
        for (unsigned char w : h_wide) n_wide += w ? 1 : 0;
        const int64_t cap = std::min<int64_t>(4096, std::max<int64_t>(256, n / 512));
        if (n_wide > 0 && n_wide <= cap && n_wide < n) {
            pvmax = as_double(h_chk[0].lvmax_bits);
            cvmax = as_double(h_chk[1].lvmax_bits);
            pss = as_double(h_chk[0].lssmax_bits);
            css = as_double(h_chk[1].lssmax_bits);
        } else {
            n_wide = 0;
        }
    }
    // (the SELL image's own limits - values below 2^vbits, sums of squares below 2^32 - hold for the non-wide rows a
    // fortiori; the wide rows never enter it)
    const int p_vbits = std::min(24, 32 - ceil_log2_64(p_dim));
    const int c_vbits = std::min(24, 32 - ceil_log2_64(c_dim));
    ix->packed = !force_generic && integral && p_dim < (1 << 20) - 1 && c_dim < (1 << 20) - 1 &&
                 pvmax < (double)(1u << p_vbits) && cvmax < (double)(1u << c_vbits) && pss < 4294967296.0 && css < 4294967296.0;
    ix->pack16 = ix->packed && pss < 65536.0 && css < 65536.0 && pvmax < 65536.0 && cvmax < 65536.0 &&
                 !ix->no_pack16;
    lap("validation");

    // ---- popularity renumbering of the place dimensions (see knn.hip / knn_ht.h)
    const int ht_qt = cfg::kHtQt;
    int32_t ht_h = std::min<int32_t>(p_dim, cfg::kHtHead);
    if (ix->env_ht_h > 0) ht_h = std::min<int32_t>(p_dim, ix->env_ht_h);
    ht_h = std::min<int32_t>(ht_h, 65536 / (2 * ht_qt));
    const bool want_ht = ix->pack16 && !ix->no_ht && !force_generic && n > 0 && n < ((int64_t)1 << 24) && pvmax < 256.0 &&
                         cvmax < 256.0 && c_dim <= cfg::kHtCatRows;
    const bool use_pop = ix->packed && !ix->no_pop && n > 0 &&
                         (ix->force_hash || (size_t)p_dim * 2 > (size_t)cfg::kDirectMaxBytes || want_ht);
    int32_t pop_h = 0;
    DevBuf<int32_t> new_of_old;
    DevBuf<uint32_t> freq, freq_new;
    if (use_pop) {
        LOCREC_TRY(freq.alloc((size_t)p_dim));
        LOCREC_TRY(freq_new.alloc((size_t)p_dim));
        LOCREC_TRY(new_of_old.alloc((size_t)p_dim));
        LOCREC_HIP_TRY(hipMemsetAsync(freq.p, 0, (size_t)p_dim * 4, s));
        if (pe > 0) hipLaunchKernelGGL(kb_hist, grid_for(pe, kHistChunk), dim3(256), 0, s, pe, p_idx, freq.p);
        DevBuf<uint64_t> dk, dk2;
        LOCREC_TRY(dk.alloc((size_t)p_dim));
        LOCREC_TRY(dk2.alloc((size_t)p_dim));
        hipLaunchKernelGGL(kb_dim_keys, grid_for(p_dim), dim3(256), 0, s, p_dim, freq.p, dk.p);
        KB_PRIM(tmp, s, prim::sort_keys(p_, bytes_, dk.p, dk2.p, p_dim, 0, 64, s));
        hipLaunchKernelGGL(kb_new_of_old, grid_for(p_dim), dim3(256), 0, s, p_dim, dk2.p, new_of_old.p, freq_new.p);
        pop_h = std::min<int32_t>(p_dim, cfg::kPopTable);
        if (ix->env_pop_h > 0) pop_h = std::min<int32_t>(p_dim, ix->env_pop_h);
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
    }
    const int32_t key_h = want_ht ? ht_h : pop_h;
    lap("popularity");

    // ---- row order: ascending (nnz_place, nnz_category[, head / popular count]), stable
    DevBuf<uint64_t> rk, rk2, idk, idk2;
    DevBuf<uint32_t> rv, order, idv, idv2;
    DevBuf<int64_t> ids_row, len_p, len_c, len_r;
    DevBuf<int32_t> row_of_input, nnz_p, nnz_c, npop;
    const size_t nn = (size_t)std::max<int64_t>(1, n);
    LOCREC_TRY(rk.alloc(nn));
    LOCREC_TRY(rk2.alloc(nn));
    LOCREC_TRY(rv.alloc(nn));
    LOCREC_TRY(order.alloc(nn));
    LOCREC_TRY(idk.alloc(nn));
    LOCREC_TRY(idk2.alloc(nn));
    LOCREC_TRY(idv.alloc(nn));
    LOCREC_TRY(idv2.alloc(nn));
    LOCREC_TRY(ids_row.alloc(nn));
    LOCREC_TRY(row_of_input.alloc(nn));
    LOCREC_TRY(nnz_p.alloc(nn));
    LOCREC_TRY(nnz_c.alloc(nn));
    LOCREC_TRY(npop.alloc(nn));
    LOCREC_TRY(len_p.alloc(nn + 1));
    LOCREC_TRY(len_c.alloc(nn + 1));
    LOCREC_TRY(len_r.alloc(nn + 1));
    LOCREC_HIP_TRY(hipMemsetAsync(len_p.p, 0, (nn + 1) * 8, s));
    LOCREC_HIP_TRY(hipMemsetAsync(len_c.p, 0, (nn + 1) * 8, s));
    LOCREC_HIP_TRY(hipMemsetAsync(len_r.p, 0, (nn + 1) * 8, s));
    if (n > 0) {
        hipLaunchKernelGGL(kb_row_keys, grid_for(n), dim3(256), 0, s, n, p_ptr, p_idx, c_ptr, use_pop ? new_of_old.p : nullptr, key_h,
                           rk.p, rv.p);
        KB_PRIM(tmp, s, prim::sort_pairs(p_, bytes_, rk.p, rk2.p, rv.p, order.p, (int)n, 0, 64, s));
        hipLaunchKernelGGL(kb_apply_order, grid_for(n), dim3(256), 0, s, n, rk2.p, order.p, ids, r_ptr, ids_row.p, row_of_input.p,
                           nnz_p.p, nnz_c.p, npop.p, len_p.p, len_c.p, len_r.p, idk.p, idv.p);
    }
    // ---- rid: rank of each row's person id (tie-break person_id asc)
    DevBuf<unsigned long long> dup;
    LOCREC_TRY(dup.alloc(1));
    LOCREC_HIP_TRY(hipMemsetAsync(dup.p, 0xFF, 8, s));
    LOCREC_TRY(ix->rid.alloc(nn));
    LOCREC_TRY(ix->ids_by_rank.alloc(nn));
    LOCREC_TRY(ix->row_of_rid.alloc(nn));
    if (n > 0) {
        KB_PRIM(tmp, s, prim::sort_pairs(p_, bytes_, idk.p, idk2.p, idv.p, idv2.p, (int)n, 0, 64, s));
        hipLaunchKernelGGL(kb_rank_ids, grid_for(n), dim3(256), 0, s, n, idk2.p, idv2.p, ix->rid.p, ix->ids_by_rank.p,
                           ix->row_of_rid.p, dup.p);
    }
    unsigned long long h_dup = ~0ull;
    LOCREC_HIP_TRY(hipMemcpyAsync(&h_dup, dup.p, 8, hipMemcpyDeviceToHost, s));
    ix->ids_row.resize((size_t)n);
    ix->row_of_input.resize((size_t)n);
    ix->fp.nnz.resize((size_t)n);
    ix->fc.nnz.resize((size_t)n);
    ix->ids_sorted.resize((size_t)n);
    ix->row_by_rank.resize((size_t)n);
    if (n > 0) {
        LOCREC_HIP_TRY(hipMemcpyAsync(ix->ids_row.data(), ids_row.p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(ix->row_of_input.data(), row_of_input.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(ix->fp.nnz.data(), nnz_p.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(ix->fc.nnz.data(), nnz_c.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(ix->ids_sorted.data(), ix->ids_by_rank.p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(ix->row_by_rank.data(), ix->row_of_rid.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    }
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    if (h_dup != ~0ull) return fail(LOCREC_E_INVALID_ARG, "duplicate person_id %lld", (long long)ix->ids_sorted[(size_t)h_dup]);
    lap("row order + id rank");
    DevBuf<unsigned char> wide_row;  // per ROW (non-empty only in the per-row fallback)
    const unsigned char *skip = nullptr;
    if (n_wide > 0) {
        LOCREC_TRY(wide_row.alloc((size_t)n));
        hipLaunchKernelGGL(kb_wide_rows, grid_for(n), dim3(256), 0, s, n, order.p, wide_in.p, wide_row.p);
        ix->is_wide.resize((size_t)n);
        LOCREC_HIP_TRY(hipMemcpyAsync(ix->is_wide.data(), wide_row.p, (size_t)n, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
        for (int64_t r = 0; r < n; ++r)
            if (ix->is_wide[(size_t)r]) ix->wide_rows.push_back((int32_t)r);
        LOCREC_TRY(ix->wide_rows_dev.upload(ix->wide_rows, s));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
        skip = wide_row.p;
    }

    // ---- CSR of both families in row order (place: renumbered and re-sorted inside every row)
    auto csr_of = [&](DevFamily &d, const int64_t *in_ptr, const int32_t *in_idx, const double *in_val, int64_t ne,
                      DevBuf<int64_t> &len, const int32_t *renumber, DevBuf<int32_t> *orig_idx) -> int32_t {
        LOCREC_TRY(d.csr_ptr.alloc(nn + 1));
        KB_PRIM(tmp, s, prim::exclusive_sum(p_, bytes_, len.p, d.csr_ptr.p, (int)(n + 1), s));
        const size_t nee = (size_t)std::max<int64_t>(1, ne);
        LOCREC_TRY(d.csr_idx.alloc(nee));
        LOCREC_TRY(d.csr_val.alloc(nee));
        if (orig_idx) LOCREC_TRY(orig_idx->alloc(nee));
        if (ne == 0 || n == 0) return LOCREC_OK;
        DevBuf<uint64_t> k1, k2;
        DevBuf<uint32_t> v1, v2;
        LOCREC_TRY(k1.alloc(nee));
        LOCREC_TRY(v1.alloc(nee));
        hipLaunchKernelGGL(kb_place_keys, grid_for(n), dim3(256), 0, s, n, order.p, in_ptr, d.csr_ptr.p, in_idx, renumber, k1.p, v1.p,
                           orig_idx ? orig_idx->p : nullptr);
        const uint64_t *ks = k1.p;
        const uint32_t *vs = v1.p;
        if (renumber) {  // the renumbered indices of a row are no longer ascending: sort by (row, new index)
            LOCREC_TRY(k2.alloc(nee));
            LOCREC_TRY(v2.alloc(nee));
            KB_PRIM(tmp, s, prim::sort_pairs(p_, bytes_, k1.p, k2.p, v1.p, v2.p, (int)ne, 0,
                                                              32 + std::max(1, ceil_log2_64(n + 1)), s));
            ks = k2.p;
            vs = v2.p;
        }
        hipLaunchKernelGGL(kb_finish_csr, grid_for(ne), dim3(256), 0, s, ne, ks, vs, in_val, d.csr_idx.p, d.csr_val.p);
        LOCREC_HIP_TRY(hipStreamSynchronize(s));  // k / v go out of scope
        return LOCREC_OK;
    };
    LOCREC_TRY(csr_of(ix->fp, p_ptr, p_idx, p_val, pe, len_p, use_pop ? new_of_old.p : nullptr, nullptr));
    LOCREC_TRY(csr_of(ix->fc, c_ptr, c_idx, c_val, ce, len_c, nullptr, nullptr));
    lap("CSR (gather, renumber, sort)");

    // ---- SELL-64 images, norms
    LOCREC_TRY(build_family(ix.get(), ix->fp, p_dim, p_vbits, ix->packed, nnz_p, tmp, skip));
    LOCREC_TRY(build_family(ix.get(), ix->fc, c_dim, c_vbits, ix->packed, nnz_c, tmp, skip));
    if (use_pop) {
        LOCREC_TRY(ix->fp.sell_split.alloc((size_t)std::max(1, ix->nslices)));
        if (ix->nslices > 0)
            hipLaunchKernelGGL(kb_split, grid_for(ix->nslices), dim3(256), 0, s, n, ix->nslices, ix->fp.csr_ptr.p, ix->fp.csr_idx.p, pop_h,
                               ix->fp.sell_w.p, ix->fp.sell_split.p);
        ix->fp.pop_h = pop_h;
        ix->fp.scan_bytes += (int64_t)ix->nslices * 4;
    }
    if (n_wide > 0) {
        // the side kernels' lane-major copy of the wide rows (both families; values are integer counts)
        auto side_of = [&](const DevFamily &src, DevBuf<int2> &img, DevBuf<int32_t> &off_d, DevBuf<int32_t> &w_d) -> int32_t {
            const int32_t nw = (int32_t)ix->wide_rows.size(), ns = (nw + 63) / 64;
            std::vector<int32_t> off((size_t)ns + 1, 0), wv((size_t)ns, 0);
            for (int32_t sl = 0; sl < ns; ++sl) {
                int m = 0;
                for (int32_t w = sl * 64; w < std::min(nw, sl * 64 + 64); ++w) m = std::max(m, src.nnz[(size_t)ix->wide_rows[(size_t)w]]);
                wv[(size_t)sl] = m;
                off[(size_t)sl + 1] = off[(size_t)sl] + m * 64;
            }
            const size_t total = (size_t)std::max(1, off[(size_t)ns]);
            LOCREC_TRY(img.alloc(total));
            LOCREC_HIP_TRY(hipMemsetAsync(img.p, 0xFF, total * sizeof(int2), s));  // index -1 = padding
            LOCREC_TRY(off_d.upload(off, s));
            LOCREC_TRY(w_d.upload(wv, s));
            hipLaunchKernelGGL(kb_side_fill, grid_for(nw), dim3(256), 0, s, nw, ix->wide_rows_dev.p, src.csr_ptr.p, src.csr_idx.p,
                               src.csr_val.p, off_d.p, img.p);
            LOCREC_HIP_TRY(hipStreamSynchronize(s));  // (off / wv are locals)
            return LOCREC_OK;
        };
        LOCREC_TRY(side_of(ix->fp, ix->side_p, ix->side_off_p, ix->side_w_p));
        LOCREC_TRY(side_of(ix->fc, ix->side_c, ix->side_off_c, ix->side_w_c));
    }
    lap("SELL images + norms");

    // ---- head / tail image (knn_ht.h)
    if (want_ht) {
        HtIndex &ht = ix->ht;
        const int rsh = 4;  // byte offset of a 16-byte plane row (knn_ht.h)
        DevBuf<int32_t> nhead, tail_nnz;
        DevBuf<int64_t> tail_len, tail_ptr;
        LOCREC_TRY(nhead.alloc(nn));
        LOCREC_TRY(tail_nnz.alloc(nn));
        LOCREC_TRY(tail_len.alloc(nn + 1));
        LOCREC_TRY(tail_ptr.alloc(nn + 1));
        LOCREC_TRY(ht.tail_hits.alloc(nn));
        LOCREC_TRY(ht.ss.alloc((size_t)ix->nslices * 64));
        LOCREC_HIP_TRY(hipMemsetAsync(tail_len.p, 0, (nn + 1) * 8, s));
        LOCREC_HIP_TRY(hipMemsetAsync(ht.ss.p, 0, (size_t)ix->nslices * 64 * 4, s));
        if (skip) hipLaunchKernelGGL(kb_freq_drop_rows, grid_for(n), dim3(256), 0, s, n, ix->fp.csr_ptr.p, ix->fp.csr_idx.p, skip, freq_new.p);
        hipLaunchKernelGGL(kb_ht_rows, grid_for(n), dim3(256), 0, s, n, ix->fp.csr_ptr.p, ix->fp.csr_idx.p, ix->fp.csr_val.p,
                           ix->fc.csr_ptr.p, ix->fc.csr_val.p, ht_h, freq_new.p, nhead.p, tail_nnz.p, tail_len.p, ht.tail_hits.p, ht.ss.p, skip);
        KB_PRIM(tmp, s, prim::exclusive_sum(p_, bytes_, tail_len.p, tail_ptr.p, (int)(n + 1), s));
        // head rows and category rows in the head / tail element format
        auto ht_sell = [&](const DevFamily &src, const DevBuf<int32_t> &lens, const int32_t *limit, DevBuf<uint32_t> &sell,
                           DevBuf<int64_t> &off, DevBuf<int32_t> &w, int64_t &elements) -> int32_t {
            DevBuf<int64_t> w64;
            LOCREC_TRY(w.alloc((size_t)ix->nslices));
            LOCREC_TRY(w64.alloc((size_t)ix->nslices + 1));
            LOCREC_TRY(off.alloc((size_t)ix->nslices + 1));
            LOCREC_HIP_TRY(hipMemsetAsync(w64.p, 0, ((size_t)ix->nslices + 1) * 8, s));
            hipLaunchKernelGGL(kb_slice_width, grid_for(ix->nslices), dim3(256), 0, s, n, ix->nslices, lens.p, 1, w.p, w64.p);
            KB_PRIM(tmp, s, prim::exclusive_sum(p_, bytes_, w64.p, off.p, ix->nslices + 1, s));
            LOCREC_HIP_TRY(hipMemcpyAsync(&elements, off.p + ix->nslices, 8, hipMemcpyDeviceToHost, s));
            LOCREC_HIP_TRY(hipStreamSynchronize(s));
            const size_t padded = (size_t)elements + (size_t)cfg::kHtNP * 256;  // knn_scan_ht always loads kHtNP groups
            LOCREC_TRY(sell.alloc(padded));
            LOCREC_HIP_TRY(hipMemsetAsync(sell.p, 0, padded * 4, s));
            hipLaunchKernelGGL(kb_sell_packed, grid_for(n), dim3(256), 0, s, n, src.csr_ptr.p, src.csr_idx.p, src.csr_val.p, limit,
                               off.p, 0, rsh, sell.p, skip);
            return LOCREC_OK;
        };
        int64_t hpe = 0, hce = 0;
        LOCREC_TRY(ht_sell(ix->fp, nhead, nhead.p, ht.p_sell, ht.p_off, ht.p_w, hpe));
        LOCREC_TRY(ht_sell(ix->fc, nnz_c, nullptr, ht.c_sell, ht.c_off, ht.c_w, hce));
        if (hpe / 4 < ((int64_t)1 << 32) && hce / 4 < ((int64_t)1 << 32)) {
            LOCREC_TRY(ht.desc.alloc((size_t)std::max(1, ix->nslices)));
            hipLaunchKernelGGL(kb_desc, grid_for(ix->nslices), dim3(256), 0, s, ix->nslices, ht.p_off.p, ht.p_w.p, ht.c_off.p, ht.c_w.p,
                               ht.desc.p);
            LOCREC_TRY(ht.cold.alloc(cfg::kHtColdBytes));
            // postings of the tail places: (place, row) keys, radix-sorted -> rows ascending inside a place
            int64_t nt_el = 0;
            LOCREC_HIP_TRY(hipMemcpyAsync(&nt_el, tail_ptr.p + n, 8, hipMemcpyDeviceToHost, s));
            LOCREC_HIP_TRY(hipStreamSynchronize(s));
            const int64_t ntail = std::max<int64_t>(0, (int64_t)p_dim - ht_h);
            LOCREC_TRY(ht.post.alloc((size_t)std::max<int64_t>(1, nt_el)));
            LOCREC_TRY(ht.post_ptr.alloc((size_t)ntail + 1));
            {
                DevBuf<int64_t> cnt;
                LOCREC_TRY(cnt.alloc((size_t)ntail + 1));
                LOCREC_HIP_TRY(hipMemsetAsync(cnt.p, 0, ((size_t)ntail + 1) * 8, s));
                if (ntail > 0) hipLaunchKernelGGL(kb_tail_freq, grid_for(ntail), dim3(256), 0, s, ntail, freq_new.p, ht_h, cnt.p);
                KB_PRIM(tmp, s, prim::exclusive_sum(p_, bytes_, cnt.p, ht.post_ptr.p, (int)(ntail + 1), s));
                LOCREC_HIP_TRY(hipStreamSynchronize(s));
            }
            if (nt_el > 0) {
                DevBuf<uint64_t> k1, k2;
                DevBuf<uint32_t> v1, v2;
                LOCREC_TRY(k1.alloc((size_t)nt_el));
                LOCREC_TRY(k2.alloc((size_t)nt_el));
                LOCREC_TRY(v1.alloc((size_t)nt_el));
                LOCREC_TRY(v2.alloc((size_t)nt_el));
                hipLaunchKernelGGL(kb_tail_keys, grid_for(n), dim3(256), 0, s, n, ix->fp.csr_ptr.p, ix->fp.csr_idx.p, ix->fp.csr_val.p,
                                   nhead.p, tail_ptr.p, ht_h, k1.p, v1.p, skip);
                KB_PRIM(tmp, s, prim::sort_pairs(p_, bytes_, k1.p, k2.p, v1.p, v2.p, (int)nt_el, 0,
                                                                  32 + std::max(1, ceil_log2_64(std::max<int64_t>(2, ntail))), s));
                hipLaunchKernelGGL(kb_postings, grid_for(nt_el), dim3(256), 0, s, nt_el, k2.p, v2.p, ht.post.p);
                LOCREC_HIP_TRY(hipStreamSynchronize(s));
            }
            LOCREC_TRY(ht.rid.alloc((size_t)ix->nslices * 64));
            hipLaunchKernelGGL(kb_pad_rid, grid_for((int64_t)ix->nslices * 64), dim3(256), 0, s, n, (int64_t)ix->nslices * 64, ix->rid.p,
                               ht.rid.p);
            // what request planning needs on the host: tail sizes per row
            ht.tail_nnz.resize((size_t)n);
            std::vector<int64_t> th((size_t)n);
            LOCREC_HIP_TRY(hipMemcpyAsync(ht.tail_nnz.data(), tail_nnz.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
            LOCREC_HIP_TRY(hipMemcpyAsync(th.data(), ht.tail_hits.p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
            LOCREC_HIP_TRY(hipStreamSynchronize(s));
            ht.tail_hits_ps.assign((size_t)n + 1, 0);
            for (int64_t r = 0; r < n; ++r) ht.tail_hits_ps[(size_t)r + 1] = ht.tail_hits_ps[(size_t)r] + th[(size_t)r];
            ht.scan_bytes = (hpe + hce) * 4 + nt_el * 4 + (int64_t)ix->nslices * 24 + n * 8;
            ht.h = ht_h;
            ht.qt = ht_qt;
            ht.ready = true;
        }
        LOCREC_HIP_TRY(hipGetLastError());
        lap("head / tail image + postings");
    }

    // ---- ratings CSR in row order and its place-major transpose
    {
        const int64_t ne = r_ptr ? re : pe;
        const size_t nee = (size_t)std::max<int64_t>(1, ne);
        LOCREC_TRY(ix->r_ptr.alloc(nn + 1));
        KB_PRIM(tmp, s, prim::exclusive_sum(p_, bytes_, len_r.p, ix->r_ptr.p, (int)(n + 1), s));
        LOCREC_TRY(ix->r_place.alloc(nee));
        LOCREC_TRY(ix->r_rating.alloc(nee));
        if (n > 0 && ne > 0) {
            if (r_ptr) {
                hipLaunchKernelGGL(kb_gather_ratings, grid_for(n), dim3(256), 0, s, n, order.p, r_ptr, ix->r_ptr.p, r_place, r_rating,
                                   ix->r_place.p, ix->r_rating.p);
            } else {  // the place vectors themselves: caller's indices and values, in row order (the row's ORIGINAL element order)
                DevBuf<uint64_t> k1;
                DevBuf<uint32_t> v1;
                LOCREC_TRY(k1.alloc(nee));
                LOCREC_TRY(v1.alloc(nee));
                hipLaunchKernelGGL(kb_place_keys, grid_for(n), dim3(256), 0, s, n, order.p, p_ptr, ix->r_ptr.p, p_idx, nullptr, k1.p, v1.p,
                                   nullptr);
                DevBuf<int32_t> oi;
                LOCREC_TRY(oi.alloc(nee));
                hipLaunchKernelGGL(kb_finish_csr, grid_for(ne), dim3(256), 0, s, ne, k1.p, v1.p, p_val, oi.p, ix->r_rating.p);
                hipLaunchKernelGGL(kb_default_ratings, grid_for(ne), dim3(256), 0, s, ne, oi.p, ix->r_place.p);
                LOCREC_HIP_TRY(hipStreamSynchronize(s));
            }
        }
        {
            int64_t *mx = nullptr;
            DevBuf<int64_t> mxb;
            LOCREC_TRY(mxb.alloc(1));
            mx = mxb.p;
            if (n > 0) {
                KB_PRIM(tmp, s, prim::reduce(p_, bytes_, len_r.p, mx, (int)n, MaxI64(), (int64_t)0, s));
                LOCREC_HIP_TRY(hipMemcpyAsync(&ix->max_r_nnz, mx, 8, hipMemcpyDeviceToHost, s));
                LOCREC_HIP_TRY(hipStreamSynchronize(s));
            }
        }
        // distinct places, ascending
        DevBuf<uint64_t> pk, pk2, uk;
        DevBuf<int64_t> nuniq;
        LOCREC_TRY(pk.alloc(nee));
        LOCREC_TRY(pk2.alloc(nee));
        LOCREC_TRY(uk.alloc(nee));
        LOCREC_TRY(nuniq.alloc(1));
        int64_t ncp = 0;
        if (ne > 0) {
            hipLaunchKernelGGL(kb_place_sortkeys, grid_for(ne), dim3(256), 0, s, ne, ix->r_place.p, pk.p);
            KB_PRIM(tmp, s, prim::sort_keys(p_, bytes_, pk.p, pk2.p, (int)ne, 0, 64, s));
            KB_PRIM(tmp, s, prim::unique(p_, bytes_, pk2.p, uk.p, nuniq.p, (int)ne, s));
            LOCREC_HIP_TRY(hipMemcpyAsync(&ncp, nuniq.p, 8, hipMemcpyDeviceToHost, s));
            LOCREC_HIP_TRY(hipStreamSynchronize(s));
        }
        LOCREC_TRY(ix->cplace_dev.alloc((size_t)std::max<int64_t>(1, ncp)));
        if (ncp > 0) hipLaunchKernelGGL(kb_unbias, grid_for(ncp), dim3(256), 0, s, ncp, uk.p, ix->cplace_dev.p);
        ix->cplace_ids.resize((size_t)ncp);
        if (ncp > 0) LOCREC_HIP_TRY(hipMemcpyAsync(ix->cplace_ids.data(), ix->cplace_dev.p, (size_t)ncp * 8, hipMemcpyDeviceToHost, s));
        // place index of every rating row, and the transpose (rows ascending inside a place: a fixed order)
        LOCREC_TRY(ix->r_pidx.alloc(nee));
        LOCREC_TRY(ix->cp_ptr.alloc((size_t)ncp + 1));
        LOCREC_TRY(ix->cp_row.alloc(nee));
        LOCREC_TRY(ix->cp_rating.alloc(nee));
        if (ne > 0) {
            DevBuf<uint32_t> tv, tv2;
            LOCREC_TRY(tv.alloc(nee));
            LOCREC_TRY(tv2.alloc(nee));
            hipLaunchKernelGGL(kb_place_index, grid_for(n), dim3(256), 0, s, n, ix->r_ptr.p, ix->r_place.p, ix->cplace_dev.p, ncp,
                               ix->r_pidx.p, pk.p, tv.p);
            KB_PRIM(tmp, s, prim::sort_pairs(p_, bytes_, pk.p, pk2.p, tv.p, tv2.p, (int)ne, 0, 64, s));
            hipLaunchKernelGGL(kb_transpose_out, grid_for(ne), dim3(256), 0, s, ne, pk2.p, tv2.p, ix->r_rating.p, ix->cp_row.p,
                               ix->cp_rating.p);
            hipLaunchKernelGGL(kb_cp_ptr, grid_for(ncp + 1), dim3(256), 0, s, ncp, ne, pk2.p, ix->cp_ptr.p);
        } else {
            LOCREC_HIP_TRY(hipMemsetAsync(ix->cp_ptr.p, 0, ((size_t)ncp + 1) * 8, s));
        }
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
        LOCREC_HIP_TRY(hipGetLastError());
    }
    lap("ratings (CSR + transpose)");
    *out = ix.release();
    return LOCREC_OK;
}

}  // namespace locrec
