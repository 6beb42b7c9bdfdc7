// knn_side.h -- the per-row format fallback's side kernels: the index's few wide rows against a request (knn_side_scan1) or a batch (knn_side_topk), from the plain CSR
// A fragment of knn.hip's translation unit: included by knn.hip inside its anonymous namespace, after the parameter
// blocks and the headers it names (it is not a stand-alone header; the split only keeps every file readable).
#pragma once

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
        // the categories first (a handful of elements against a dense table): with a full list, a wide row whose place
        // cosine could be 1 and still falls short of the list's last entry needs no place dot at all
        {
            const int2 *img = C.side_c + C.side_off_c[sl] + ln;
            const int width = C.side_w_c[sl];
            for (int j0 = 0; j0 < width; j0 += 8) {
                int2 ev[8];
#pragma unroll
                for (int b = 0; b < 8; ++b) ev[b] = j0 + b < width ? img[(int64_t)(j0 + b) * 64] : make_int2(-1, 0);
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    if (ev[b].x < 0) continue;
                    const double t = cat[ev[b].x] * (double)ev[b].y;
                    dc = dc + t;
                }
            }
        }
        if (cnt >= K && (float)pw * 1.0001f + (float)dc * C.inorm_c[row] * qfc < tau32) continue;  // (cosine <= 1)
        if (hashed) {
            const int2 *img = C.side_p + C.side_off_p[sl] + ln;
            const int width = C.side_w_p[sl];
            for (int j0 = 0; j0 < width; j0 += 8) {   // eight coalesced loads (lane = wide row) in flight, then the probes
                int2 ev[8];
#pragma unroll
                for (int b = 0; b < 8; ++b) ev[b] = j0 + b < width ? img[(int64_t)(j0 + b) * 64] : make_int2(-1, 0);
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const int2 e = ev[b];
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
            }
        } else {
            dp = side_merge_dot(C.p_ptr, C.p_idx, C.p_val, qrow, row);
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
