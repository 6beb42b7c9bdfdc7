// knn_merge.h -- knn_merge_partial / knn_merge: the per-chunk top-K lists of a query merged into its final list
// A fragment of knn.hip's translation unit: included by knn.hip inside its anonymous namespace, after the parameter
// blocks and the headers it names (it is not a stand-alone header; the split only keeps every file readable).
#pragma once

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
