// knn_device.h -- device helpers shared by the KNN scan kernels: index hashing, Distance.vectorLength, the LDS panel of a query tile, the packed / generic family dots, exact_similarity and the per-query LDS top-K lists
// A fragment of knn.hip's translation unit: included by knn.hip inside its anonymous namespace, after the parameter
// blocks and the headers it names (it is not a stand-alone header; the split only keeps every file readable).
#pragma once

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
