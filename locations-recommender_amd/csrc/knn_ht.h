// knn_ht.h -- "head / tail" form of the batched KNN scan (knn_scan MODE 3); textually included by
// knn.hip inside its anonymous namespace.
//
// Why.  The row scan (MODE 1/2) pays, for EVERY stored element of every candidate, a hash lookup of
// the query tile's slot plus QT multiply-adds - although, after the popularity renumbering of the
// place dimensions, an element beyond the first few hundred places meets a query of the tile only
// a few times per thousand rows.  VALU issue, not HBM, bounds that kernel (profiles/r01_g_pmc_*).
//
// What.  The place dimensions are split at H (popularity rank):
//   HEAD  (index < H):  stays in the SELL-64 row image, one u32 per element =
//                       value << 16 | index << 4.  The query tile's head is a DIRECT panel in LDS,
//                       planes of [rows][8 queries] u16 (ht_build_panel) with the category panel at
//                       LDS offset 0, so a panel row's byte address is the element's low half (one
//                       v_and) and the multiplier is its high half (read in place by v_pk_mad_u16
//                       through op_sel): 1 + QT/2 VALU and QT/8 ds_read_b128 per element, no hash.
//   TAIL  (index >= H): stored INVERTED, one posting list per place (row << 8 | value, rows
//                       ascending).  Before a batch is scanned, ht_build_hits walks, for every
//                       tail element of every query of a tile, that place's posting list and emits
//                       one HIT (candidate lane, query, product) per posting, counting-sorted by
//                       candidate slice.  The scan adds a slice's hits (a handful per slice and
//                       tile) into a wave-private LDS accumulator that each lane then folds into
//                       its head dots - the tail costs a few instructions per SLICE instead of
//                       ~40 per element and lane.
//   The category family (20 dimensions) is all head.
// Integer products and sums are exact in any order, so the dots - and hence similarities - are the
// reference's bit for bit (same argument as the packed formats; PACK16 legality is required: every
// dot < 65536, values <= 255, rows < 2^24).
//
// Replaces the same reference lines as knn_scan: KnnRecommender.scala:27-49,76-96, Distance.scala:7-9.

constexpr int kHtCatRows = 64;        // rows reserved for the category panel (c_dim <= 64)
constexpr int kHtMaxEntries = 1024;   // (query, tail place) pairs of one tile the pre-pass holds in LDS
constexpr int kHtSubSlices = 4096;    // candidate slices one counting pass covers (u32 counters in LDS)
constexpr int kHtPreThreads = 512;

// (struct HtParams - the scan-side parameters - is declared in knn.hip in front of ScanParams)

// ---- element / hit formats -------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t ht_pack_hit(uint32_t lane, uint32_t q, uint32_t prod)
{
    return (lane << 21) | (q << 16) | prod;
}

// byte offset, inside one wave's tail accumulator, of the 32-bit word that holds queries (2w, 2w+1)
// of candidate lane r.  Rows are QT*2 bytes; the 16-byte chunks of a row are XOR-swizzled with bits
// of the row number so that the 16 lanes ds_read_b128 services together (MI355X_MICROARCH.md, LDS:
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) touch 64 different banks.
template <int QT>
__device__ __forceinline__ uint32_t ht_tail_word(uint32_t r, uint32_t w)
{
    constexpr uint32_t chunks = QT / 8;  // 16-byte chunks per row
    const uint32_t sw = chunks == 2 ? ((r >> 3) & 1u) : ((r >> 2) & 3u);
    return r * (QT * 2) + ((((w >> 2) ^ sw) & (chunks - 1)) << 4) + ((w & 3u) << 2);
}

// ---- pre-pass: hits of one tile ---------------------------------------------------------------
struct HtPreParams {
    const int64_t *csr_ptr;     // place family, renumbered indices, ascending inside a row
    const int32_t *csr_idx;
    const double *csr_val;
    const int64_t *post_ptr;    // [p_dim - h + 1]
    const uint32_t *post;       // row << 8 | value
    const int32_t *qrows;       // or nullptr: rows qrow0 ..
    int32_t qrow0, nq;
    int32_t h;
    int32_t slice0, nslices;    // scanned candidate slices [slice0, nslices)
    const int64_t *tile_base;
    uint32_t *hits;
    uint32_t *off;
    int32_t off_stride;
    int32_t *error;             // set when a tile has more than kHtMaxEntries entries (host checks first; tripwire)
};

// tile_base[t] = number of hits of the tiles before t: a posting of a tail place is one hit for every
// query of the tile that holds the place.  One block; tail_hits[row] is precomputed at create time.
__global__ __launch_bounds__(1024) void ht_tile_bases(const int64_t *tail_hits, const int32_t *qrows, int32_t qrow0,
                                                      int32_t nq, int32_t qt, int32_t ntiles, int64_t *tile_base)
{
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int t0 = 0; t0 < ntiles; t0 += 1024) {
        const int t = t0 + tid;
        int64_t mine = 0;
        if (t < ntiles)
            for (int q = t * qt; q < min(nq, (t + 1) * qt); ++q) mine += tail_hits[qrows ? qrows[q] : qrow0 + q];
        int64_t inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int64_t before = carry;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (t < ntiles) tile_base[t] = before + inc - mine;
        __syncthreads();
        if (tid == 1023) carry = before + inc;
        __syncthreads();
    }
    if (tid == 0) tile_base[ntiles] = carry;
}

// Measured at cfg2 (136 M hits per 16,384-query batch, 8.5 per slice and tile): 1.55 ms per launch =
// 35 us of entry set-up, 75 us of zeroing / prefix scans, 450 us per pass of LDS atomics (~2 cycles per
// lane-atomic per CU: the LDS's atomic rate, not latency - a version with four 64-posting loads of a wave in
// flight over a flattened work list was no faster) and 535 us for the scattered 4-byte stores of pass B.
template <int QT>
__global__ __launch_bounds__(kHtPreThreads) void ht_build_hits(const HtPreParams P)
{
    __shared__ int64_t e_cur[kHtMaxEntries];   // next posting of the entry
    __shared__ int64_t e_end[kHtMaxEntries];
    __shared__ uint32_t e_qv[kHtMaxEntries];   // q << 16 | query value
    __shared__ uint32_t cnt[kHtSubSlices];     // per slice of the current sub-range: count, then cursor
    __shared__ uint32_t wtot[kHtPreThreads / 64 + 1];
    __shared__ int s_ne;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = kHtPreThreads / 64;
    const int tile = blockIdx.x;
    const int q0 = tile * QT;
    const int nqt = min(QT, P.nq - q0);
    if (tid == 0) s_ne = 0;
    __syncthreads();
    // entries: one per (query of the tile, tail element of its place vector)
    for (int q = 0; q < nqt; ++q) {
        const int row = P.qrows ? P.qrows[q0 + q] : P.qrow0 + q0 + q;
        for (int64_t e = P.csr_ptr[row] + tid; e < P.csr_ptr[row + 1]; e += kHtPreThreads) {
            const int idx = P.csr_idx[e];
            if (idx >= P.h) {
                const int slot = atomicAdd(&s_ne, 1);
                if (slot < kHtMaxEntries) {
                    const int64_t b = P.post_ptr[idx - P.h];
                    e_cur[slot] = b;
                    e_end[slot] = P.post_ptr[idx - P.h + 1];
                    e_qv[slot] = ((uint32_t)q << 16) | (uint32_t)P.csr_val[e];
                }
            }
        }
    }
    __syncthreads();
    const int ne = min(s_ne, kHtMaxEntries);
    if (s_ne > kHtMaxEntries && tid == 0 && P.error) atomicAdd(P.error, 1);
    // the scan may cover a candidate shard only: skip the postings before its first row
    if (P.slice0 > 0) {
        const uint32_t first_row = (uint32_t)P.slice0 * 64u;
        for (int e = tid; e < ne; e += kHtPreThreads) {
            int64_t a = e_cur[e], b = e_end[e];
            while (a < b) {  // lower bound of first_row
                const int64_t m = (a + b) >> 1;
                if ((P.post[m] >> 8) < first_row) a = m + 1; else b = m;
            }
            e_cur[e] = a;
        }
        __syncthreads();
    }
    uint32_t *off_row = P.off + (int64_t)tile * P.off_stride;
    uint32_t *hits = P.hits + P.tile_base[tile];
    uint32_t run = 0;  // hits of this tile before the current sub-range (same value in every thread)
    for (int sl0 = P.slice0; sl0 < P.nslices; sl0 += kHtSubSlices) {
        const int nsl = min(kHtSubSlices, P.nslices - sl0);
        const uint32_t row_end = (uint32_t)(sl0 + nsl) * 64u;
        for (int i = tid; i < nsl; i += kHtPreThreads) cnt[i] = 0u;
        __syncthreads();
        // pass A: count the postings of every entry that fall into this sub-range, per slice
        for (int e = wave; e < ne; e += NW) {
            const int64_t end = e_end[e];
            for (int64_t p = e_cur[e]; p < end; p += 64) {
                const bool in = p + lane < end;
                const uint32_t row = in ? (P.post[p + lane] >> 8) : 0xFFFFFFFFu;
                const bool mine = in && row < row_end;
                if (mine) atomicAdd(&cnt[(row >> 6) - sl0], 1u);
                if (!__all(mine)) break;
            }
        }
        __syncthreads();
        // exclusive prefix of the counts -> offsets; every thread owns nsl / threads consecutive slices
        constexpr int PER = kHtSubSlices / kHtPreThreads;
        uint32_t local[PER];
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = tid * PER + j;
            local[j] = i < nsl ? cnt[i] : 0u;
            sum += local[j];
        }
        uint32_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        if (tid == 0) {
            uint32_t acc = 0;
            for (int w = 0; w < NW; ++w) {
                const uint32_t t = wtot[w];
                wtot[w] = acc;
                acc += t;
            }
            wtot[NW] = acc;
        }
        __syncthreads();
        uint32_t excl = run + wtot[wave] + inc - sum;
        const uint32_t total = wtot[NW];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = tid * PER + j;
            if (i < nsl) {
                cnt[i] = excl;                           // becomes the slice's write cursor
                off_row[(sl0 - P.slice0) + i] = excl;
            }
            excl += local[j];
        }
        __syncthreads();
        // pass B: the hits themselves, and the entries' cursors move past this sub-range
        for (int e = wave; e < ne; e += NW) {
            const int64_t end = e_end[e];
            const uint32_t qv = e_qv[e];
            const uint32_t q = qv >> 16, qval = qv & 0xFFFFu;
            int64_t p = e_cur[e];
            for (; p < end; p += 64) {
                const bool in = p + lane < end;
                const uint32_t pv = in ? P.post[p + lane] : 0xFFFFFFFFu;
                const uint32_t row = pv >> 8;
                const bool mine = in && row < row_end;
                if (mine) {
                    const uint32_t pos = atomicAdd(&cnt[(row >> 6) - sl0], 1u);
                    hits[pos] = ht_pack_hit(row & 63u, q, (pv & 0xFFu) * qval);
                }
                const uint64_t m = __ballot(mine);
                if (m != ~0ull) {  // postings are sorted by row: the lanes inside the sub-range form a prefix
                    p += __popcll(m);
                    break;
                }
            }
            if (lane == 0) e_cur[e] = p < end ? p : end;
        }
        run += total;
        __syncthreads();
    }
    if (tid == 0) off_row[P.nslices - P.slice0] = run;
}

// ---- scan side ---------------------------------------------------------------------------------

// u16 panel of one family's head, in PLANES of 8 queries: plane k = [rows][8] u16 (16-byte rows) holds
// queries 8k .. 8k + 7, planes `rows * 16` bytes apart.  A lane reads one 16-byte row per plane and element
// (ds_read_b128); the 16 lanes the LDS services together then spread over all 16 slots of the 256-byte
// bank row by `index mod 16`.  (With the queries of a row side by side - 32-byte rows - one read used only 8
// of the 16 slots: 3.5 LDS cycles per read on Zipf-distributed indices instead of 2.5, and the LDS was as
// busy as the vector ALU; profiles/r02_pmc_knn.txt.)  Zero, then the tile's queries.
// rows = indices the panel holds (idx_limit), plane_rows >= rows = rows a plane has room for.
template <int QT>
__device__ void ht_build_panel(const Family &f, int rows, int plane_rows, const int *s_qrow, int nqt, unsigned short *panel)
{
    const int tid = threadIdx.x;
    uint32_t *p32 = reinterpret_cast<uint32_t *>(panel);
    for (int i = tid; i < plane_rows * QT / 2; i += blockDim.x) p32[i] = 0u;
    __syncthreads();
    for (int q = 0; q < nqt; ++q) {
        const int row = s_qrow[q];
        unsigned short *plane = panel + (q >> 3) * plane_rows * 8 + (q & 7);
        for (int64_t e = f.csr_ptr[row] + tid; e < f.csr_ptr[row + 1]; e += blockDim.x) {
            const int idx = f.csr_idx[e];
            if (idx < rows) plane[idx * 8] = (unsigned short)f.csr_val[e];
        }
    }
    __syncthreads();
}

// acc (two queries per register) += value * panel row, for the four elements of one dwordx4 group.
// Element = value << 16 | byte offset of the row inside a plane (index << 4): the row address is one v_and,
// the value is read from the element's HIGH half by both halves of v_pk_mad_u16 (op_sel), so nothing is
// unpacked.  `stride` = bytes between planes.
template <int QT>
__device__ __forceinline__ void ht_accum4(const u32x4 e4, const unsigned char *panel, int stride, uint32_t (&acc)[QT / 2])
{
    const uint32_t ee[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const unsigned char *r = panel + (ee[t] & 0xFFFFu);
#pragma unroll
        for (int i = 0; i < QT / 8; ++i) {
            const u32x4 pv = *reinterpret_cast<const u32x4 *>(r + i * stride);
            asm("v_pk_mad_u16 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[4 * i + 0]) : "v"(pv.x), "v"(ee[t]));
            asm("v_pk_mad_u16 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[4 * i + 1]) : "v"(pv.y), "v"(ee[t]));
            asm("v_pk_mad_u16 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[4 * i + 2]) : "v"(pv.z), "v"(ee[t]));
            asm("v_pk_mad_u16 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[4 * i + 3]) : "v"(pv.w), "v"(ee[t]));
        }
    }
}

template <int QT>
__device__ __forceinline__ void ht_family_dots(const unsigned char *panel, int stride, const u32x4 *lane_base, int w4, Group4 cur,
                                               uint32_t (&acc)[QT / 2])
{
    for (int j = 0; j < w4; j += 4) {
        const Group4 nxt = load_group(lane_base, j + 4, w4);
        ht_accum4<QT>(cur.a0, panel, stride, acc);
        if (j + 1 < w4) ht_accum4<QT>(cur.a1, panel, stride, acc);
        if (j + 2 < w4) ht_accum4<QT>(cur.a2, panel, stride, acc);
        if (j + 3 < w4) ht_accum4<QT>(cur.a3, panel, stride, acc);
        cur = nxt;
    }
}

// =================================================================================================
// knn_scan_ht: the dedicated head / tail scan.  Same tile / chunk / top-K structure as knn_scan
// (lists, adaptive insertion mode, in-kernel interval replay), but the hot path is written for the
// instruction count: per candidate element one v_and + QT/2 v_pk_mad_u16, per (candidate, query)
// pair ~7 VALU of f32 prefilter, and everything a survivor needs (fp64 norms, id rank, query norms)
// is loaded only on the rare path.  The row data is software-pipelined ACROSS slices with rotating
// registers: as soon as a dwordx4 group of the current slice has been consumed, the same registers
// receive the group of the wave's next slice.

struct HtSliceDesc {   // per candidate slice, built at create time (one 16-byte scalar load per slice)
    uint32_t off_p4;   // first dwordx4 of the slice's place-head image (in units of 16 bytes)
    uint32_t off_c4;   // ... of its category image
    uint32_t w4;       // dwordx4 groups per lane: place | category << 16
    uint32_t pad;
};

struct HtCold {        // parameters of the rare paths, read from memory where they are needed
    const double *norm_p, *norm_c;   // [nrows] Distance.vectorLength
    const uint32_t *rid;             // [nrows] rank of the person id
    const int64_t *pcsr_ptr;         // place family, renumbered, for the panel build
    const int32_t *pcsr_idx;
    const double *pcsr_val;
    const int64_t *ccsr_ptr;
    const int32_t *ccsr_idx;
    const double *ccsr_val;
    double *part_s;                  // [nq][nchunks][K]
    uint32_t *part_rid;
    int32_t *part_cnt;
    int32_t *overflow;               // [1] replayed intervals (statistics)
    const int32_t *qrows;            // or nullptr
    double pw, cw;
    int32_t qrow0, nq, nrows, K, S, h, c_rows, nchunks;
    int32_t off_tail, off_cand_s, off_cand_rid, off_misc, off_queue;
    int32_t flush_mask, enter_threads, fast;
    const double *seed;              // [nq] or nullptr: a LOWER bound of each query's final K-th best similarity (0 = none)
    double *seed_out;                // SEED pass: [nq] the bounds it derives
    int32_t stride, off_hist;        // SEED pass: every stride-th slice is sampled; LDS offset of the histograms
    unsigned long long *dbg_out;  // DEBUG_SWITCHES builds: [16] phase clocks / event counts summed over waves
    int32_t dbg;  // LOCREC_DEBUG_HT (DEBUG_SWITCHES builds only): 1 no place dots, 2 no category dots, 4 no tail, 8 no prefilter
};

// "all but the N youngest vector-memory operations of this wave are complete" - stated explicitly where
// the compiler cannot count across the loop's back edge (it would wait with vmcnt(0), i.e. also for
// the prefetches that were issued a moment ago).  gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14.
template <int N>
__device__ __forceinline__ void ht_wait_vm()
{
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

typedef _Float16 ht_h2 __attribute__((ext_vector_type(2)));
constexpr float kHtBoundScale = 1.006f;  // see the packed f16 bound in knn_scan_ht

// x >= 0 as f16, rounded UP (never below x - also where f16 is subnormal or x underflows it)
__device__ __forceinline__ _Float16 ht_f16_up(float x)
{
    _Float16 h = (_Float16)x;
    if ((float)h < x) h = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned short, h) + 1u));
    return h;
}
constexpr int kHtNP = 4;  // place groups (of 4 elements) held in rotating registers; wider slices take the slow loop
constexpr int kHtNC = 2;  // category groups

template <int QT>
__device__ __forceinline__ void ht_mad_elem(const u32x4 (&pv)[QT / 8], uint32_t e, uint32_t (&acc)[QT / 2])
{
#pragma unroll
    for (int i = 0; i < QT / 8; ++i) {
        asm("v_pk_mad_u16 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[4 * i + 0]) : "v"(pv[i].x), "v"(e));
        asm("v_pk_mad_u16 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[4 * i + 1]) : "v"(pv[i].y), "v"(e));
        asm("v_pk_mad_u16 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[4 * i + 2]) : "v"(pv[i].z), "v"(e));
        asm("v_pk_mad_u16 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[4 * i + 3]) : "v"(pv[i].w), "v"(e));
    }
}

// STRIDE = bytes between the planes of the panel (a compile-time constant: the second read is the same
// address register with an immediate offset)
template <int QT, int STRIDE>
__device__ __forceinline__ void ht_read_row(const unsigned char *panel, uint32_t e, u32x4 (&pv)[QT / 8])
{
    const unsigned char *r = panel + (e & 0xFFFFu);
#pragma unroll
    for (int i = 0; i < QT / 8; ++i) pv[i] = *reinterpret_cast<const u32x4 *>(r + i * STRIDE);
}

// acc = value * panel row: the FIRST element of a slice's first group starts the accumulators (no zeroing pass: 16
// v_mov per slice and tile otherwise)
template <int QT>
__device__ __forceinline__ void ht_mul_elem(const u32x4 (&pv)[QT / 8], uint32_t e, uint32_t (&acc)[QT / 2])
{
#pragma unroll
    for (int i = 0; i < QT / 8; ++i) {
        asm("v_pk_mul_lo_u16 %0, %1, %2 op_sel:[0,1]" : "=v"(acc[4 * i + 0]) : "v"(pv[i].x), "v"(e));
        asm("v_pk_mul_lo_u16 %0, %1, %2 op_sel:[0,1]" : "=v"(acc[4 * i + 1]) : "v"(pv[i].y), "v"(e));
        asm("v_pk_mul_lo_u16 %0, %1, %2 op_sel:[0,1]" : "=v"(acc[4 * i + 2]) : "v"(pv[i].z), "v"(e));
        asm("v_pk_mul_lo_u16 %0, %1, %2 op_sel:[0,1]" : "=v"(acc[4 * i + 3]) : "v"(pv[i].w), "v"(e));
    }
}

// one dwordx4 group = four elements; the panel rows of element t + 1 are read while element t is multiplied
template <int QT, int STRIDE, bool FIRST = false>
__device__ __forceinline__ void ht_group(const unsigned char *panel, const u32x4 g, uint32_t (&acc)[QT / 2])
{
    u32x4 a[QT / 8], b[QT / 8];
    ht_read_row<QT, STRIDE>(panel, g.x, a);
    ht_read_row<QT, STRIDE>(panel, g.y, b);
    if constexpr (FIRST) ht_mul_elem<QT>(a, g.x, acc);
    else ht_mad_elem<QT>(a, g.x, acc);
    ht_read_row<QT, STRIDE>(panel, g.z, a);
    ht_mad_elem<QT>(b, g.y, acc);
    ht_read_row<QT, STRIDE>(panel, g.w, b);
    ht_mad_elem<QT>(a, g.z, acc);
    ht_mad_elem<QT>(b, g.w, acc);
}

// ---- survivors.  About K*ln(N/K) candidates per query enter its list over a scan - roughly one pair per
// slice and tile - so a pair that passes the f32 bound must be resolved without touching global
// memory and without 16 unrolled copies of the fp64 code: the wave loops over the (wave-uniform)
// queries that have a passing pair in some lane.
struct HtLds {          // LDS pointers of the block (computed once per thread, kept in a few registers)
    double *cand_s;
    uint32_t *cand_r;
    double *s_qnp, *s_qnc, *tau_s;
    uint32_t *tau_r;
    int *s_qrow, *cnt;
    float *tau32;
};

// one packed u16 dot out of eight words (a select chain on SSA values: the accumulators must never be
// indexed dynamically, or the compiler moves the whole array to scratch memory)
__device__ __forceinline__ uint32_t ht_dot_of(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5,
                                              uint32_t a6, uint32_t a7, int q)
{
    const int w = q >> 1;
    const uint32_t lo = w == 0 ? a0 : w == 1 ? a1 : w == 2 ? a2 : a3;
    const uint32_t hi = w == 4 ? a4 : w == 5 ? a5 : w == 6 ? a6 : a7;
    const uint32_t v = w < 4 ? lo : hi;
    return (q & 1) ? v >> 16 : v & 0xFFFFu;
}

// exact similarity of this lane's candidate against query q.  The candidate's norms come from its sums of
// squares - exact integers, so sqrt((double)ss) is Distance.vectorLength bit for bit - and nothing
// is loaded from global memory.
#define LOCREC_HT_EXACT(q, sx)                                                                                        \
    exact_similarity(ht_dot_of(ap[0], ap[1], ap[2], ap[3], ap[4], ap[5], ap[6], ap[7], (q)),                          \
                     ht_dot_of(ac[0], ac[1], ac[2], ac[3], ac[4], ac[5], ac[6], ac[7], (q)), cnp, cnc, L.s_qnp[(q)],  \
                     L.s_qnc[(q)], pw, cw, (sx))

constexpr int kHtSeedSampleSlices = 1024;  // candidate slices the SEED pass samples per tile

// SEED = true is the threshold-seeding pass (one launch in front of the scan proper, over every stride-th
// slice): same panels, pipeline, dots and tail fold, but instead of bounding / resolving / inserting, every
// lane keeps the MAXIMUM f32 similarity it has seen per query.  At the end the block has W * 64 lane maxima
// per query - each the similarity of a distinct sampled candidate - and the (K + 1)-th largest of them
// (K + 1: the query itself may be in the sample) is a lower bound of the final K-th best similarity: the
// K-th best of a subset cannot exceed the K-th best of the whole.  The scan proper starts every block with
// that threshold instead of zero: no cold-start flood in which every positive pair is a survivor (14 % of
// the wave time went into the ~70 synchronous warm-up iterations), and K * ln(rank of the seed / K)
// insertions per query instead of K * ln(N / K).  Measured at cfg2: scan 18.3 -> 15.0 ms for a pass of
// ~1 ms; with IDEAL seeds (the previous launch's own results, an experiment) 13.7 ms
// (profiles/r02_knn_scan_ht_ideal_seeds.log).  (A per-query LDS histogram of all sampled pairs gives the
// exact K-th of the sample but costs 16 LDS atomics per lane and slice, with heavy same-bin conflicts.)
template <int QT, int W, bool SEED = false>
__global__ __launch_bounds__(W * 64, W == 8 ? 4 : 3) void knn_scan_ht(
    const u32x4 *__restrict__ sell_p, const u32x4 *__restrict__ sell_c, const HtSliceDesc *__restrict__ desc,
    const uint32_t *__restrict__ ss_all, const uint32_t *__restrict__ rid_all, const uint32_t *__restrict__ hits_all,
    const uint32_t *__restrict__ hoff, const int64_t *__restrict__ tile_base, int32_t off_stride, int32_t slice0,
    int32_t nslices, int32_t slices_per_chunk, const HtCold *__restrict__ cold)
{
    static_assert(QT == 16, "the out-of-line survivor paths are written for a tile of 16 queries");
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int kCatBytes = kHtCatRows * QT * 2;
    constexpr int kCatStride = kHtCatRows * 16, kPlaceStride = locrec::cfg::kHtHead * 16;  // plane strides (host: h <= kHtHead)
    HtLds L;
    L.cand_s = reinterpret_cast<double *>(smem + cold->off_cand_s);
    L.cand_r = reinterpret_cast<uint32_t *>(smem + cold->off_cand_rid);
    L.s_qnp = reinterpret_cast<double *>(smem + cold->off_misc);
    L.s_qnc = L.s_qnp + QT;
    L.tau_s = L.s_qnc + QT;
    L.tau_r = reinterpret_cast<uint32_t *>(L.tau_s + QT);
    L.s_qrow = reinterpret_cast<int *>(L.tau_r + QT);
    L.cnt = L.s_qrow + QT;
    float *s_qfp = reinterpret_cast<float *>(L.cnt + QT);
    float *s_qfc = s_qfp + QT;
    L.tau32 = s_qfc + QT;
    int *s_flags = reinterpret_cast<int *>(L.tau32 + QT) + 1;
    // the packed bound's per-query operands: (pw / |q_place|, cw / |q_category|) as an f16 pair, scaled up by
    // kHtBoundScale, and the negated threshold (the accumulator of v_dot2c_f32_f16)
    uint32_t *s_qf2 = reinterpret_cast<uint32_t *>(L.tau32 + QT) + 4;
    float *s_ntau = reinterpret_cast<float *>(s_qf2 + QT);
    double *wq_s = reinterpret_cast<double *>(smem + cold->off_queue);
    uint32_t *wq_r = reinterpret_cast<uint32_t *>(wq_s + W * kQueueCap);
    uint32_t *wq_q = wq_r + W * kQueueCap;
    int *wq_cnt = reinterpret_cast<int *>(wq_q + W * kQueueCap);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q0 = blockIdx.y * QT;
    const int nqt = min(QT, cold->nq - q0);
    const int K = cold->K, S = cold->S;

    if (tid < QT) {
        int row = -1;
        if (tid < nqt) row = cold->qrows ? cold->qrows[q0 + tid] : cold->qrow0 + q0 + tid;
        L.s_qrow[tid] = row;
        const double np_ = row >= 0 ? cold->norm_p[row] : 0.0;
        const double nc_ = row >= 0 ? cold->norm_c[row] : 0.0;
        L.s_qnp[tid] = np_;
        L.s_qnc[tid] = nc_;
        s_qfp[tid] = np_ > 0.0 ? (float)(cold->pw / np_) : 0.0f;
        s_qfc[tid] = nc_ > 0.0 ? (float)(cold->cw / nc_) : 0.0f;
        {
            const ht_h2 f2 = {ht_f16_up(s_qfp[tid] * kHtBoundScale), ht_f16_up(s_qfc[tid] * kHtBoundScale)};
            s_qf2[tid] = __builtin_bit_cast(uint32_t, f2);
        }
        // A seeded threshold: the list starts empty, but a candidate below the seed can never reach the
        // final top K, so the block skips the cold-start flood (every positive pair a survivor).  The seed is
        // strictly below the K-th best, so "better than (seed, worst id rank)" accepts everything that matters.
        const double sd = (cold->seed && tid < nqt) ? cold->seed[q0 + tid] : 0.0;
        L.tau_s[tid] = sd;
        L.tau_r[tid] = sd > 0.0 ? 0xFFFFFFFFu : 0u;
        L.tau32[tid] = sd > 0.0 ? (float)(sd * (1.0 - 1e-4)) : 1.17549435e-38f;
        s_ntau[tid] = -L.tau32[tid];
        L.cnt[tid] = 0;
    }
    if (tid < W) wq_cnt[tid] = 0;
    if (tid == 0) {
        s_flags[0] = 0;
        s_flags[1] = 0;
        s_flags[2] = 0;
    }
    {
        uint32_t *t32 = reinterpret_cast<uint32_t *>(smem + cold->off_tail);
        for (int i = tid; i < W * 64 * QT / 2; i += W * 64) t32[i] = 0u;
    }
    float seed_max[SEED ? QT : 1];  // SEED pass: this lane's largest f32 similarity per query
#pragma unroll
    for (int i = 0; i < (SEED ? QT : 1); ++i) seed_max[i] = 0.0f;
    __syncthreads();
    {
        Family fc{}, fp{};
        fc.csr_ptr = cold->ccsr_ptr; fc.csr_idx = cold->ccsr_idx; fc.csr_val = cold->ccsr_val;
        fp.csr_ptr = cold->pcsr_ptr; fp.csr_idx = cold->pcsr_idx; fp.csr_val = cold->pcsr_val;
        ht_build_panel<QT>(fc, cold->c_rows, kHtCatRows, L.s_qrow, nqt, reinterpret_cast<unsigned short *>(smem));
        ht_build_panel<QT>(fp, cold->h, locrec::cfg::kHtHead, L.s_qrow, nqt, reinterpret_cast<unsigned short *>(smem + kCatBytes));
    }
    const unsigned char *pan_c = smem;
    const unsigned char *pan_p = smem + kCatBytes;
    unsigned char *my_tail = smem + cold->off_tail + wave * (64 * QT * 2);
    const uint32_t *hoff_row = hoff + (int64_t)blockIdx.y * off_stride - slice0;
    const uint32_t *hits = hits_all + tile_base[blockIdx.y];
    const int fast_allowed = cold->fast, flush_mask = cold->flush_mask, enter_threads = cold->enter_threads;

    const int slice_begin = slice0 + blockIdx.x * slices_per_chunk;
    const int slice_end = min(slice_begin + slices_per_chunk, nslices);
    const int stride = SEED ? cold->stride : 1;  // (a constant 1 in the scan proper)
    const int iters = (slices_per_chunk + W * stride - 1) / (W * stride);

#ifdef LOCREC_DEBUG_SWITCHES
    const int dbg_bits = cold->dbg;
#define LOCREC_HT_DBG(bit) (dbg_bits & (bit))
    unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_mark = __builtin_amdgcn_s_memtime();
#define LOCREC_HT_LAP(i)                                              \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        tk[i] += now_ - t_mark;                                       \
        t_mark = now_;                                                \
    } while (0)
#define LOCREC_HT_COUNT(i, v) tk[i] += (v)
#else
#define LOCREC_HT_DBG(bit) 0
#define LOCREC_HT_LAP(i)
#define LOCREC_HT_COUNT(i, v)
#endif
    // ---- pipeline registers (valid for slice `primed`)
    int primed = -1;
    u32x4 gp[kHtNP], gc[kHtNC];
    uint32_t ssrow = 0u, ridrow = 0u;               // sums of squares (place | category << 16) and id rank of the lane's row
    uint32_t dcur_p4 = 0, dcur_c4 = 0, dcur_w4 = 0;   // descriptor of the current slice
    uint32_t dnxt_p4 = 0, dnxt_c4 = 0, dnxt_w4 = 0;   // ... of the wave's next slice
    uint32_t hc0 = 0, hc1 = 0, hcur = 0, hn0 = 0, hn1 = 0;
#pragma unroll
    for (int g = 0; g < kHtNP; ++g) gp[g] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int g = 0; g < kHtNC; ++g) gc[g] = u32x4{0u, 0u, 0u, 0u};

    bool fastmode = LOCREC_HT_DBG(32) != 0;
    int calm = 0;
    const double pw = cold->pw, cw = cold->cw;
    for (int it = 0; it < iters; ++it) {
        // The SIMD's arbiter favours its older waves: with a fixed priority the block's first four waves did the
        // same work 11 % faster than its last four and then idled at every drain barrier.  The younger half gets
        // the raised priority in 9 of 16 iterations, the older half in the other 7 (measured work per half:
        // fixed priority 17.1 : 19.1, alternating 18.2 : 19.0, 5 of 8 for the younger 18.9 : 18.4, 3 of 4 19.3 : 18.0).
        if (((wave >> 2) != 0) == ((it & 15) < 9)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        const int slice = slice_begin + (it * W + wave) * stride;  // wave-uniform
        const bool live = slice < slice_end;
        const int row = slice * 64 + lane;
        unsigned pend = 0;
        uint32_t cur_ss = 0u, cur_rid = 0u;
        uint32_t ap[QT / 2], ac[QT / 2];  // (started by the first group of each family, zeroed where a slice has none)
        if (live) {
            const int nslice = slice + W * stride;
            const bool have_next = it + 1 < iters && nslice < slice_end;
            const bool have_next2 = it + 2 < iters && nslice + W * stride < slice_end;
            // The pipeline's loads are UNCONDITIONAL vector loads in straight-line code (a slice without a
            // successor reloads itself; every array is padded for it): only then can the compiler count
            // them and wait with vmcnt(N) for an older load while the prefetches behind it stay in flight.
            const int pf = have_next ? nslice : slice;            // slice whose row data is prefetched in this iteration
            const int pf2 = have_next2 ? nslice + W * stride : pf;  // slice whose descriptor / hit range is fetched
            if (primed != slice) {  // first slice of the wave, or an interval is being replayed: fill synchronously
                const HtSliceDesc d = desc[slice];
                dcur_p4 = __builtin_amdgcn_readfirstlane(d.off_p4);
                dcur_c4 = __builtin_amdgcn_readfirstlane(d.off_c4);
                dcur_w4 = __builtin_amdgcn_readfirstlane(d.w4);
                const u32x4 *sp = sell_p + dcur_p4, *sc = sell_c + dcur_c4;  // uniform bases: 32-bit lane offsets below
#pragma unroll
                for (int g = 0; g < kHtNP; ++g) gp[g] = sp[g * 64 + lane];
#pragma unroll
                for (int g = 0; g < kHtNC; ++g) gc[g] = sc[g * 64 + lane];
                ssrow = (ss_all + slice * 64)[lane];
                ridrow = (rid_all + slice * 64)[lane];
                hc0 = __builtin_amdgcn_readfirstlane(hoff_row[slice]);
                hc1 = __builtin_amdgcn_readfirstlane(hoff_row[slice + 1]);
                hcur = (hits + hc0)[min((uint32_t)lane, max(hc1 - hc0, 1u) - 1u)];
                const HtSliceDesc dn = desc[pf];
                dnxt_p4 = __builtin_amdgcn_readfirstlane(dn.off_p4);
                dnxt_c4 = __builtin_amdgcn_readfirstlane(dn.off_c4);
                dnxt_w4 = __builtin_amdgcn_readfirstlane(dn.w4);
                hn0 = __builtin_amdgcn_readfirstlane(hoff_row[pf]);
                hn1 = __builtin_amdgcn_readfirstlane(hoff_row[pf + 1]);
            }
            const int w4p = (int)(dcur_w4 & 0xFFFFu), w4c = (int)(dcur_w4 >> 16);
            const u32x4 *spn = sell_p + dnxt_p4, *scn = sell_c + dnxt_c4;  // the prefetched slice's images (uniform)
            // Load order of one iteration: [descriptor, hit range, hits] (3), place groups (kHtNP), category
            // groups (kHtNC), [ss, rid] (2).  The first place group was followed by kHtNP - 1 + kHtNC + 2 loads
            // in the previous iteration:
            ht_wait_vm<kHtNP - 1 + kHtNC + 2>();
            // descriptor / hit range two slices ahead: VECTOR loads on purpose (a scalar load would sit in the
            // LDS reads' lgkmcnt and stall the first of them), read at the end of this iteration
            int vpf2 = pf2;
            asm volatile("" : "+v"(vpf2));  // (a per-lane index: keeps the loads vector loads, and global ones)
            const u32x4 vd2 = reinterpret_cast<const u32x4 *>(desc)[vpf2];
            const uint32_t vm0 = hoff_row[vpf2], vm1 = hoff_row[vpf2 + 1];
            const uint32_t hn = LOCREC_HT_DBG(16) ? 0u : hc1 - hc0, hbase = hc0, hfirst = hcur;
            const uint32_t hnext = (hits + hn0)[min((uint32_t)lane, max(hn1 - hn0, 1u) - 1u)];
            // ---- dots.  Group g of the current slice is consumed, then its registers receive group g of the
            // wave's next slice.
#pragma unroll
            for (int g = 0; g < kHtNP; ++g) {
                // group g (g >= 1) is followed by the previous iteration's later groups, its [ss, rid], this
                // iteration's three leading loads and g reloads: kHtNP + kHtNC + 4 in all; one less, to be safe
                if (g > 0) ht_wait_vm<kHtNP + kHtNC + 3>();
                if (g == 0) {
                    if (0 < w4p && !LOCREC_HT_DBG(1)) {
                        ht_group<QT, kPlaceStride, true>(pan_p, gp[0], ap);
                    } else {
#pragma unroll
                        for (int i = 0; i < QT / 2; ++i) ap[i] = 0u;
                    }
                } else if (g < w4p && !LOCREC_HT_DBG(1)) {
                    ht_group<QT, kPlaceStride>(pan_p, gp[g], ap);
                }
                gp[g] = spn[g * 64 + lane];
            }
            for (int g = kHtNP; g < w4p; ++g)  // a slice of long rows: the remaining groups, loaded on demand
                ht_group<QT, kPlaceStride>(pan_p, (sell_p + dcur_p4)[g * 64 + lane], ap);
#pragma unroll
            for (int g = 0; g < kHtNC; ++g) {
                ht_wait_vm<kHtNP + kHtNC + 3>();
                if (g == 0) {
                    if (0 < w4c && !LOCREC_HT_DBG(2)) {
                        ht_group<QT, kCatStride, true>(pan_c, gc[0], ac);
                    } else {
#pragma unroll
                        for (int i = 0; i < QT / 2; ++i) ac[i] = 0u;
                    }
                } else if (g < w4c && !LOCREC_HT_DBG(2)) {
                    ht_group<QT, kCatStride>(pan_c, gc[g], ac);
                }
                gc[g] = scn[g * 64 + lane];
            }
            for (int g = kHtNC; g < w4c; ++g)
                ht_group<QT, kCatStride>(pan_c, (sell_c + dcur_c4)[g * 64 + lane], ac);
            const uint32_t ss = ssrow, myrid = ridrow;
            ssrow = (ss_all + pf * 64)[lane];   // (padded to whole slices: 0 = no row)
            ridrow = (rid_all + pf * 64)[lane];
            // everything older than this iteration's kHtNP + kHtNC + 2 reloads has arrived: the previous [ss, rid],
            // the hits of this slice, the descriptor / hit range fetched above
            ht_wait_vm<kHtNP + kHtNC + 2>();
            // f32 inverse norms for the bound: v_rsq of the integer sums of squares (0 = absent vector / padding row)
            const float icnp_cur = (ss & 0xFFFFu) ? __builtin_amdgcn_rsqf((float)(ss & 0xFFFFu)) : 0.0f;
            const float icnc_cur = (ss >> 16) ? __builtin_amdgcn_rsqf((float)(ss >> 16)) : 0.0f;
            // ---- tail: this slice's hits -> wave-private accumulator -> folded into the head dots
            if (hn > 0 && !LOCREC_HT_DBG(4)) {
                uint32_t waddr = 0u;
                if ((uint32_t)lane < hn) {  // the first 64 hits were prefetched (hn <= 64 almost always)
                    const uint32_t q = (hfirst >> 16) & 31u;
                    waddr = ht_tail_word<QT>(hfirst >> 21, q >> 1);
                    atomicAdd(reinterpret_cast<uint32_t *>(my_tail + waddr), (hfirst & 0xFFFFu) << ((q & 1u) * 16u));
                }
                for (uint32_t done = 64; done < hn; done += 64) {  // more than 64 hits in one slice and tile: rare
                    const uint32_t hh = (hits + hbase + done)[min((uint32_t)lane, hn - done - 1u)];
                    if ((uint32_t)lane < hn - done) {
                        const uint32_t q = (hh >> 16) & 31u;
                        atomicAdd(reinterpret_cast<uint32_t *>(my_tail + ht_tail_word<QT>(hh >> 21, q >> 1)),
                                  (hh & 0xFFFFu) << ((q & 1u) * 16u));
                    }
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
            if constexpr (SEED) {
                // f32 similarity of every sampled pair (a padding row gives 0)
#pragma unroll
                for (int i = 0; i < QT / 4; ++i) {
                    const float4 fq = reinterpret_cast<const float4 *>(s_qfp)[i];
                    const float4 gq = reinterpret_cast<const float4 *>(s_qfc)[i];
                    const float fqa[4] = {fq.x, fq.y, fq.z, fq.w}, gqa[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
                    for (int z = 0; z < 4; ++z) {
                        const int q = 4 * i + z;
                        const uint32_t dpq = (q & 1) ? ap[q >> 1] >> 16 : ap[q >> 1] & 0xFFFFu;
                        const uint32_t dcq = (q & 1) ? ac[q >> 1] >> 16 : ac[q >> 1] & 0xFFFFu;
                        const float a = __builtin_fmaf((float)dcq * icnc_cur, gqa[z], ((float)dpq * icnp_cur) * fqa[z]);
                        seed_max[q] = fmaxf(seed_max[q], a);
                    }
                }
            } else {
            // ---- f32 upper bound of the combined similarity against the query's current K-th value (1e-4
            // one-sided margin), branch-free: d = s32 - threshold, and v_alignbit shifts d's sign bit into
            // a per-lane mask (bit q set = the pair FAILS).  Only a slice with a passing pair leaves the
            // hot path.
            unsigned fails = 0;
            if (LOCREC_HT_DBG(8)) {
#pragma unroll
                for (int i = 0; i < QT / 2; ++i) asm volatile("" ::"v"(ap[i]), "v"(ac[i]));
                fails = 0xFFFFu;
            } else {
                // PACKED f16: per pair two SDWA converts build (dp, dc) as an f16 pair, one v_pk_mul_f16 scales it by the
                // lane's inverse norms, one v_dot2c_f32_f16 multiplies by the query's pair and adds the negated
                // threshold in f32 - 5 VALU per pair instead of the 7 of the f32 form.  Each product carries at
                // most three f16 roundings (< 1.5e-3 relative: the dot, the lane's inverse norm - always a normal
                // f16, norms are <= 256 - and the product); the query factors are scaled up by kHtBoundScale (1.006)
                // and rounded UP to f16 (so a tiny weight cannot be quantised away), and the sum is formed in
                // f32: the bound never falls below the exact similarity * (1 - 1e-4).  A dot above 65504
                // converts to +inf and passes.
                const ht_h2 inorm2 = {(_Float16)icnp_cur, (_Float16)icnc_cur};
                // the sixteen queries' factor pairs and negated thresholds: all eight LDS reads in flight together (read
                // four queries at a time where they are used, each read was followed by its own wait for the LDS)
                u32x4 q4v[QT / 4];
                float4 t4v[QT / 4];
#pragma unroll
                for (int i = 0; i < QT / 4; ++i) {
                    q4v[i] = reinterpret_cast<const u32x4 *>(s_qf2)[i];
                    t4v[i] = reinterpret_cast<const float4 *>(s_ntau)[i];
                }
#pragma unroll
                for (int i = QT / 4 - 1; i >= 0; --i) {  // descending, so that query 0 ends in bit 0
                    const u32x4 q4 = q4v[i];
                    const float4 t4 = t4v[i];
                    const uint32_t qfa[4] = {q4.x, q4.y, q4.z, q4.w};
                    const float nta[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                    for (int zz = 1; zz >= 0; --zz) {
                        // the odd and the even query of one accumulator word side by side: two independent chains of
                        // convert - convert - scale - dot - shift, so that each fills the other's wait states (as one
                        // chain per query the compiler padded every step with s_nop)
                        const int w = 2 * i + zz;
                        uint32_t pa, pb;
                        asm("v_cvt_f16_u16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(pa) : "v"(ap[w]));
                        asm("v_cvt_f16_u16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(pb) : "v"(ap[w]));
                        asm("v_cvt_f16_u16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(pa) : "v"(ac[w]));
                        asm("v_cvt_f16_u16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(pb) : "v"(ac[w]));
                        const ht_h2 sa = __builtin_bit_cast(ht_h2, pa) * inorm2;
                        const ht_h2 sb = __builtin_bit_cast(ht_h2, pb) * inorm2;
                        const float da = __builtin_amdgcn_fdot2(sa, __builtin_bit_cast(ht_h2, qfa[2 * zz + 1]), nta[2 * zz + 1], false);
                        const float db = __builtin_amdgcn_fdot2(sb, __builtin_bit_cast(ht_h2, qfa[2 * zz]), nta[2 * zz], false);
                        fails = __builtin_amdgcn_alignbit(fails, __builtin_bit_cast(uint32_t, da), 31);
                        fails = __builtin_amdgcn_alignbit(fails, __builtin_bit_cast(uint32_t, db), 31);
                    }
                }
            }
            LOCREC_HT_LAP(0);
            unsigned maybe = ~fails & 0xFFFFu;
            if (LOCREC_HT_DBG(256)) {
                asm volatile("" ::"v"(maybe));
                maybe = 0;
            }
            cur_ss = ss;
            cur_rid = myrid;
            if (maybe && fastmode) {
                // Barrier-free mode: a pair that passes the bound goes to the wave's queue AS IT IS - its two u16 dots,
                // the row's sums of squares, its id rank, the query - and the drain resolves it exactly, 64 entries
                // at a time with every lane busy.  (Resolving here cost two fp64 square roots and two divisions
                // executed 64 lanes wide for the one or two lanes that hold a survivor, in every second slice:
                // 7 % of the kernel's issue cycles, profiles/r03_valu_classes.txt.)
                unsigned rem = maybe;
                for (;;) {
                    const uint64_t anym = __ballot(rem != 0u);
                    if (!anym) break;
                    const unsigned bits = __builtin_amdgcn_readlane(rem, __ffsll((unsigned long long)anym) - 1);
                    const int q = __ffs(bits) - 1;  // wave-uniform
                    LOCREC_HT_COUNT(5, 1);
                    if (rem & (1u << q)) {
                        rem &= ~(1u << q);
                        const uint32_t dots = ht_dot_of(ap[0], ap[1], ap[2], ap[3], ap[4], ap[5], ap[6], ap[7], q) |
                                              (ht_dot_of(ac[0], ac[1], ac[2], ac[3], ac[4], ac[5], ac[6], ac[7], q) << 16);
                        if (q < nqt && row != L.s_qrow[q] && dots != 0u) {  // person_id =!= personId (:89); a common dimension (:91)
                            const int pos = atomicAdd(&wq_cnt[wave], 1);
                            if (pos < kQueueCap) {
                                reinterpret_cast<uint2 *>(wq_s)[wave * kQueueCap + pos] = make_uint2(dots, ss);
                                wq_r[wave * kQueueCap + pos] = myrid;
                                wq_q[wave * kQueueCap + pos] = (uint32_t)q;
                            } else {
                                s_flags[1] = 1;  // no room: this interval is replayed synchronously (below)
                            }
                        }
                    }
                }
            } else if (maybe) {
                const double cnp = sqrt((double)(ss & 0xFFFFu)), cnc = sqrt((double)(ss >> 16));
                unsigned rem = maybe;
                for (;;) {
                    const uint64_t anym = __ballot(rem != 0u);
                    if (!anym) break;
                    const unsigned bits = __builtin_amdgcn_readlane(rem, __ffsll((unsigned long long)anym) - 1);
                    const int q = __ffs(bits) - 1;  // wave-uniform
                    LOCREC_HT_COUNT(5, 1);
                    if (rem & (1u << q)) {
                        rem &= ~(1u << q);
                        double sx;
                        if (q < nqt && row != L.s_qrow[q] &&  // person_id =!= personId (:89)
                            LOCREC_HT_EXACT(q, sx) && better(sx, myrid, L.tau_s[q], L.tau_r[q]))
                            pend |= 1u << q;
                    }
                }
            }
            LOCREC_HT_COUNT(4, __ballot(maybe != 0u) != 0ull ? 1 : 0);
            LOCREC_HT_LAP(1);
            }  // !SEED
            // ---- rotate the pipeline
            dcur_p4 = dnxt_p4;
            dcur_c4 = dnxt_c4;
            dcur_w4 = dnxt_w4;
            // (the descriptor's fourth word is dead, and the compiler would reuse its register right behind the load - a
            // write-after-write hazard it guards with s_waitcnt vmcnt(0), i.e. every iteration waiting for the loads it
            // has just issued: keeping the word "used" until here keeps the register, and the counted waits, intact)
            asm volatile("" ::"v"(vd2.w));
            dnxt_p4 = __builtin_amdgcn_readfirstlane(vd2.x);
            dnxt_c4 = __builtin_amdgcn_readfirstlane(vd2.y);
            dnxt_w4 = __builtin_amdgcn_readfirstlane(vd2.z);
            hc0 = hn0;
            hc1 = hn1;
            hcur = hnext;
            hn0 = __builtin_amdgcn_readfirstlane(vm0);
            hn1 = __builtin_amdgcn_readfirstlane(vm1);
            primed = have_next ? nslice : -1;
        }
        // ---- survivors (same protocol as knn_scan: synchronous until calm, then per-wave queues drained
        // every flush interval, an overrun interval is replayed synchronously)
        if constexpr (SEED) {
            continue;  // (the seeding pass inserts nothing)
        } else if (!fastmode) {
            LOCREC_HT_COUNT(6, 1);
            int np = __syncthreads_count(pend != 0);
            if (fast_allowed) {
                bool warm = np <= enter_threads * W / 8;
#pragma unroll
                for (int q = 0; q < QT; ++q) warm = warm && (q >= nqt || L.tau32[q] > 1.17549435e-38f);
                calm = warm ? calm + 1 : 0;
            }
            while (np) {
                LOCREC_HT_COUNT(7, 1);
                if (pend) {
                    const double cnp = sqrt((double)(cur_ss & 0xFFFFu)), cnc = sqrt((double)(cur_ss >> 16));
                    unsigned rem = pend;
                    for (;;) {
                        const uint64_t anym = __ballot(rem != 0u);
                        if (!anym) break;
                        const unsigned bits = __builtin_amdgcn_readlane(rem, __ffsll((unsigned long long)anym) - 1);
                        const int q = __ffs(bits) - 1;  // wave-uniform
                        if (rem & (1u << q)) {
                            rem &= ~(1u << q);
                            double sx;
                            LOCREC_HT_EXACT(q, sx);
                            if (!better(sx, cur_rid, L.tau_s[q], L.tau_r[q])) {  // the list tightened meanwhile
                                pend &= ~(1u << q);
                            } else {
                                const int pos = atomicAdd(&L.cnt[q], 1);
                                if (pos < S) {
                                    L.cand_s[q * S + pos] = sx;
                                    L.cand_r[q * S + pos] = cur_rid;
                                    pend &= ~(1u << q);
                                }
                            }
                        }
                    }
                }
                __syncthreads();
                for (int q = 0; q < nqt; ++q)
                    if (L.cnt[q] >= S) compact_query(L.cand_s, L.cand_r, L.cnt, L.tau_s, L.tau_r, L.tau32, q, S, K, s_ntau);
                np = __syncthreads_count(pend != 0);
            }
            if (calm >= kCalmIters && ((it + 1) & flush_mask) == 0) fastmode = true;
            LOCREC_HT_LAP(2);
        } else {
            if (((it + 1) & flush_mask) == 0 || it == iters - 1) {
                __syncthreads();
                if (s_flags[1]) {
                    __syncthreads();
                    if (tid < W) wq_cnt[tid] = 0;
                    if (tid == 0) {
                        s_flags[1] = 0;
                        s_flags[2] += 1;
                    }
                    __syncthreads();
                    fastmode = false;
                    calm = 0;
                    it = (it & ~flush_mask) - 1;  // ++it -> first iteration of the interval (the pipeline re-primes)
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
                    const uint2 raw = have ? reinterpret_cast<const uint2 *>(wq_s)[wave * kQueueCap + i] : make_uint2(0u, 0u);
                    const uint32_t er = have ? wq_r[wave * kQueueCap + i] : 0u;
                    const int eq = have ? (int)wq_q[wave * kQueueCap + i] : 0;
                    // the exact similarity of the queued pair (the candidate's norms from its integer sums of squares, as
                    // in the synchronous path: Distance.vectorLength bit for bit), one entry per lane
                    double es = 0.0;
                    const bool ok = have && exact_similarity(raw.x & 0xFFFFu, raw.x >> 16, sqrt((double)(raw.y & 0xFFFFu)),
                                                             sqrt((double)(raw.y >> 16)), L.s_qnp[eq], L.s_qnc[eq], pw, cw, es);
                    insert_sync(ok, es, er, eq, L.cand_s, L.cand_r, L.cnt, L.tau_s, L.tau_r, L.tau32, nqt, S, K, s_ntau);
                }
                __syncthreads();
                if (tid < W) wq_cnt[tid] = 0;
                __syncthreads();
                if (maxfill > kQueueCap / 2) {
                    fastmode = false;
                    calm = 0;
                }
            }
            LOCREC_HT_LAP(3);
        }
    }
    if constexpr (SEED) {
        // per query: the (K + 1)-th largest of the block's W * 64 lane maxima, by bisection on the bit pattern
        // (non-negative floats order like their bits)
        uint32_t *mxbuf = reinterpret_cast<uint32_t *>(smem + cold->off_hist);  // [QT][W * 64]
        constexpr int kPer = W;  // values per lane when one wave selects for one query
#pragma unroll
        for (int q = 0; q < QT; ++q) mxbuf[q * (W * 64) + tid] = __builtin_bit_cast(uint32_t, seed_max[q]);
        __syncthreads();
        const int need = K + 1;
        for (int q = wave; q < nqt; q += W) {
            uint32_t v[kPer];
#pragma unroll
            for (int j = 0; j < kPer; ++j) v[j] = mxbuf[q * (W * 64) + j * 64 + lane];
            uint32_t t = 0u;
            if (need <= W * 64) {
                for (int bit = 30; bit >= 0; --bit) {
                    const uint32_t cand = t | (1u << bit);
                    int c = 0;
#pragma unroll
                    for (int j = 0; j < kPer; ++j) c += v[j] >= cand ? 1 : 0;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) c += __shfl_xor(c, d);
                    if (c >= need) t = cand;  // (wave-uniform)
                }
            }
            // the f32 similarity is within 1e-6 (relative) of the exact one: the margin keeps the seed below it
            if (lane == 0) cold->seed_out[q0 + q] = t ? (double)__builtin_bit_cast(float, t) * (1.0 - 1e-5) : 0.0;
        }
        return;
    }
#ifdef LOCREC_DEBUG_SWITCHES
    if (lane == 0 && cold->dbg_out) {
        for (int i = 0; i < 8; ++i) atomicAdd(&cold->dbg_out[i], tk[i]);
        atomicAdd(&cold->dbg_out[8 + (wave & 7)], tk[0] + tk[1]);  // work (hot + resolve) by wave index: imbalance inside a block
    }
#endif
    if (tid == 0 && s_flags[2] && cold->overflow) atomicAdd(cold->overflow + 1, s_flags[2]);
    const int nchunks = cold->nchunks;
    double *part_s = cold->part_s;
    uint32_t *part_rid = cold->part_rid;
    int32_t *part_cnt = cold->part_cnt;
    for (int q = 0; q < nqt; ++q) {
        compact_query(L.cand_s, L.cand_r, L.cnt, L.tau_s, L.tau_r, L.tau32, q, S, K, s_ntau);
        const int m = L.cnt[q];
        const int64_t base = ((int64_t)(q0 + q) * nchunks + blockIdx.x) * K;
        for (int i = tid; i < m; i += W * 64) {
            part_s[base + i] = L.cand_s[q * S + i];
            part_rid[base + i] = L.cand_r[q * S + i];
        }
        if (tid == 0) part_cnt[(int64_t)(q0 + q) * nchunks + blockIdx.x] = m;
        __syncthreads();
    }
}
