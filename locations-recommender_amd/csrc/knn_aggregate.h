// knn_aggregate.h -- a5 makeRecommendations0 (KnnRecommender.scala:51-70): knn_aggregate and its helpers
// A fragment of knn.hip's translation unit: included by knn.hip inside its anonymous namespace, after the parameter
// blocks and the headers it names (it is not a stand-alone header; the split only keeps every file readable).
#pragma once

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

// KeyT: uint32_t where (compact place index, position) fit 32 bits - half the LDS traffic of the bitonic sort, which is
// what this kernel's time goes into (2,048 keys x 66 stages per query) - else uint64_t; fbits = bits of the position.
template <typename KeyT>
__global__ __launch_bounds__(kAggThreads) void knn_aggregate(
    const int32_t *nb_rows, const double *nb_sims, const int64_t *nb_cnt, int32_t K,
    const int64_t *r_ptr, const int32_t *r_pidx, const double *r_rating, const int64_t *cplace_ids,
    int32_t M /* pow2 LDS capacity, <= kAggCap */, int64_t *out_place, double *out_est, int64_t *out_n,
    int32_t *out_overflow, int64_t out_stride, int32_t redo_only, int32_t fbits,
    unsigned char *host = nullptr /* one request: the result also goes into the pinned staging buffer, in
    locrec_knn_recommend's layout (count at 0, overflow flag at 16, *host_flag_src at 20, places at 64, estimates
    behind host_cap of them) */, const int32_t *host_flag_src = nullptr, int32_t host_cap = 0)
{
    // second pass of a batch: only the queries whose rows did not fit the first pass's smaller capacity
    if (redo_only && out_overflow[blockIdx.x] == 0) return;
    if (host && threadIdx.x == 0 && host_flag_src) *reinterpret_cast<int32_t *>(host + 20) = *host_flag_src;
    extern __shared__ __align__(16) unsigned char smem[];
    KeyT *key = reinterpret_cast<KeyT *>(smem);               // [M]
    double *wrv = reinterpret_cast<double *>(smem + (((size_t)M * sizeof(KeyT) + 7) & ~(size_t)7));  // [M] rating * similarity
    unsigned short *av = reinterpret_cast<unsigned short *>(wrv + M);  // [M] neighbour of the position (its similarity: simv[])
    int64_t *rbase = reinterpret_cast<int64_t *>(reinterpret_cast<unsigned char *>(av) + (((size_t)M * 2 + 7) & ~(size_t)7));  // [K] first rating row of neighbour i
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
    // (wrv / av are filled here, by ORIGINAL position f, and read through the sorted key's low bits afterwards: the rating
    // is loaded beside the place index instead of by a second search and a second round trip after the sort)
    for (int f = tid; f < n2; f += blockDim.x) {
        KeyT k = ~(KeyT)0;
        if (f < T) {
            const int a = neighbour_of(f);
            const int64_t e = rbase[a] + (f - off[a]);
            const double sim = simv[a];
            k = ((KeyT)(uint32_t)r_pidx[e] << fbits) | (KeyT)f;
            wrv[f] = r_rating[e] * sim;  // col("rating") * col("similarity") (:59)
            av[f] = (unsigned short)a;
        }
        key[f] = k;
    }
    __syncthreads();
    // Bitonic sort, ascending.  A stage whose partners lie less than C = n2 / waves apart exchanges only inside the
    // chunk of C consecutive keys a wave owns: all such stages of a level run back to back inside the wave (its LDS
    // operations complete in order) and the block meets once behind them - 21 barriers instead of 66 for 2,048 keys
    // on 16 waves, which was most of this kernel's time (1,024 threads per query: a barrier per stage is 16 waves waiting).
    {
        const int nwaves = (int)blockDim.x >> 6, wv = tid >> 6, ln = tid & 63;
        const int C = n2 / nwaves;  // keys per wave chunk (a power of two, possibly < 2)
        for (int k = 2; k <= n2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (j < C) {
                    for (int jj = j; jj > 0; jj >>= 1) {
                        for (int t = ln; t < (C >> 1); t += 64) {
                            const int tt = wv * (C >> 1) + t;
                            const int i = 2 * tt - (tt & (jj - 1));
                            const int l = i + jj;
                            const KeyT ki = key[i], kl = key[l];
                            if ((kl < ki) == ((i & k) == 0)) {
                                key[i] = kl;
                                key[l] = ki;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                    break;
                }
                for (int t = tid; t < (n2 >> 1); t += blockDim.x) {
                    const int i = 2 * t - (t & (j - 1));
                    const int l = i + j;
                    const KeyT ki = key[i], kl = key[l];
                    if ((kl < ki) == ((i & k) == 0)) {
                        key[i] = kl;
                        key[l] = ki;
                    }
                }
                __syncthreads();
            }
            __syncthreads();
        }
    }
    // heads per thread (each thread owns n2/blockDim consecutive positions)
    const int per = n2 / (int)blockDim.x > 0 ? n2 / (int)blockDim.x : 1;
    const int lo = tid * per, hi = min(lo + per, T);
    int heads = 0;
    for (int i = lo; i < hi; ++i)
        if (i == 0 || (key[i] >> fbits) != (key[i - 1] >> fbits)) ++heads;
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
        const KeyT pk = key[i] >> fbits;
        if (i == 0 || pk != (key[i - 1] >> fbits)) {
            double ws = 0.0, ss = 0.0;
            for (int t = i; t < T && (key[t] >> fbits) == pk; ++t) {
                const int f = (int)(key[t] & (((KeyT)1 << fbits) - 1));
                ws = ws + wrv[f];
                ss = ss + simv[av[f]];
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
