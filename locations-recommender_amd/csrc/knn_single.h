// knn_single.h -- the single-request stream path (KnnRecommender.scala:22-25 for one person): knn_scan1 / knn_scan1_direct8, knn_select1, knn_collect1, knn_final1
// A fragment of knn.hip's translation unit: included by knn.hip inside its anonymous namespace, after the parameter
// blocks and the headers it names (it is not a stand-alone header; the split only keeps every file readable).
#pragma once

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
// (Block b works on histogram b and selection record b: one block for a request, one per column for a tile of special
// queries of a batch.)
__global__ __launch_bounds__(1024) void knn_select1(uint32_t *hist, int32_t K, int32_t *sel)
{
    __shared__ uint32_t suf[2][1024];  // suffix sums over the per-thread bin ranges (Hillis-Steele)
    const int t = threadIdx.x;
    hist += (size_t)blockIdx.x * kHistBins;
    sel += blockIdx.x * 8;
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

// (blockIdx.y = column of a tile: S columns col_stride apart, one selection record / list per column)
__global__ __launch_bounds__(256) void knn_collect1(const double *S, const uint32_t *rid, int32_t row0,
                                                    int32_t nrows, const int32_t *sel, double *list_s,
                                                    uint32_t *list_r, int32_t *list_n, int64_t col_stride)
{
    const int row = row0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrows) return;
    S += (size_t)blockIdx.y * (size_t)col_stride;
    sel += blockIdx.y * 8;
    list_s += (size_t)blockIdx.y * kCollectCap;
    list_r += (size_t)blockIdx.y * kCollectCap;
    list_n += blockIdx.y * 8;
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
                                                  unsigned char *host, const int64_t *slots)
{
    extern __shared__ __align__(16) unsigned char smem[];
    if (slots) {  // a tile of special queries of a batch: block b sorts list b into result slot slots[b]
        const int64_t sl = slots[blockIdx.x];
        list_s += (size_t)blockIdx.x * kCollectCap;
        list_r += (size_t)blockIdx.x * kCollectCap;
        list_n += blockIdx.x * 8;
        out_ids += sl * K;
        out_sims += sl * K;
        out_rows += sl * K;
        out_cnt += sl;
        overflow += blockIdx.x;
    }
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
