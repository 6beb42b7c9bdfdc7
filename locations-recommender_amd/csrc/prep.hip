// prep.hip -- the producers of the two hot paths' inputs, on the device (SURVEY.md 8f "next" rows):
//
//   f-2  RatingsBuilder.calcRatings              knn/RatingsBuilder.scala:32-48
//        RatingVectorsBuilder.calcRatingVectors  knn/RatingVectorsBuilder.scala:10-25,52-84
//        StochasticGraphBuilder.buildWithBalancedWeights   stochastic/StochasticGraphBuilder.scala:8-28
//   f-4  PlaceVisits.calcPlaceVisits             PlaceVisits.scala:11-46 (+ Location.scala:7-8,30-43)
//
// Every function takes either host arrays (copied in and out) or device arrays of the current
// device (mem = LOCREC_MEM_DEVICE): the device form lets visits -> ratings -> rating vectors ->
// locrec_knn_create_from_device run without a host hop.  These are offline, once-per-dataset
// steps: grouping and ranking are keyed rocPRIM radix sorts / scans with small kernels between them
// (as in knn_build.hip); the spatial join replaces the reference's "very inefficient almost
// cross-join" (PlaceVisits.scala:30) with a band / cell grid whose cells are at least one search
// radius wide, so a visit meets only the places of at most 3 x 3 cells, each with the exact
// fp64 haversine of Location.scala.

#include "dev_prims.h"

#include <algorithm>
#include <cmath>
#include <limits>

#include "common.h"

namespace {

using namespace locrec;

struct Temp {
    DevBuf<unsigned char> buf;
};

#define PR_PRIM(tmp, call_with_args)                   \
    do {                                              \
        size_t bytes_ = 0;                            \
        void *p_ = nullptr;                           \
        LOCREC_HIP_TRY((call_with_args));             \
        LOCREC_TRY((tmp).buf.reserve(bytes_ + 256));  \
        p_ = (tmp).buf.p;                             \
        LOCREC_HIP_TRY((call_with_args));             \
    } while (0)

dim3 grid_for(int64_t n, int threads = 256) { return dim3((unsigned)std::max<int64_t>(1, (n + threads - 1) / threads)); }

constexpr int64_t kMaxRows = (int64_t)1 << 31;  // row numbers travel as u32 sort payloads

// An input column: the caller's array, on the device.  Host arrays are uploaded into `own`.
template <class T>
struct In {
    DevBuf<T> own;
    const T *p = nullptr;
    int32_t bind(const T *src, int64_t n, int32_t mem, hipStream_t s)
    {
        if (mem == LOCREC_MEM_DEVICE || n == 0) {
            p = src;
            return LOCREC_OK;
        }
        LOCREC_TRY(own.upload(src, (size_t)n, s));
        p = own.p;
        return LOCREC_OK;
    }
};

// An output column: the caller's device array, or a staging buffer copied back to the host array.
template <class T>
struct Out {
    DevBuf<T> own;
    T *p = nullptr;
    T *host = nullptr;
    int32_t bind(T *dst, int64_t cap, int32_t mem)
    {
        if (mem == LOCREC_MEM_DEVICE) {
            p = dst;
            return LOCREC_OK;
        }
        host = dst;
        LOCREC_TRY(own.alloc((size_t)std::max<int64_t>(cap, 1)));
        p = own.p;
        return LOCREC_OK;
    }
    int32_t deliver(int64_t count, hipStream_t s)
    {
        if (host && count > 0) LOCREC_HIP_TRY(hipMemcpyAsync(host, p, (size_t)count * sizeof(T), hipMemcpyDeviceToHost, s));
        return LOCREC_OK;
    }
};

__device__ __forceinline__ uint64_t ordered_key(int64_t v) { return (uint64_t)v ^ 0x8000000000000000ull; }  // signed order

// ---- sort rows by (person, entity), stable in the input order ------------------------------------

__global__ void pr_iota_keys(int64_t n, const int64_t *col, uint64_t *keys, uint32_t *rows)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = ordered_key(col[i]);
    rows[i] = (uint32_t)i;
}

__global__ void pr_gather_keys(int64_t n, const int64_t *col, const uint32_t *rows, uint64_t *keys)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = ordered_key(col[rows[i]]);
}

struct SortedRows {
    DevBuf<uint64_t> k0, k1;
    DevBuf<uint32_t> r0, r1;
    const uint32_t *rows = nullptr;  // input row of sorted position i
};

// two stable LSD passes: by entity, then by person
int32_t sort_person_entity(int64_t n, const int64_t *person, const int64_t *entity, SortedRows &S, Temp &tmp, hipStream_t s)
{
    LOCREC_TRY(S.k0.alloc((size_t)n));
    LOCREC_TRY(S.k1.alloc((size_t)n));
    LOCREC_TRY(S.r0.alloc((size_t)n));
    LOCREC_TRY(S.r1.alloc((size_t)n));
    hipLaunchKernelGGL(pr_iota_keys, grid_for(n), dim3(256), 0, s, n, entity, S.k0.p, S.r0.p);
    PR_PRIM(tmp, prim::sort_pairs(p_, bytes_, S.k0.p, S.k1.p, S.r0.p, S.r1.p, (int)n, 0, 64, s));
    hipLaunchKernelGGL(pr_gather_keys, grid_for(n), dim3(256), 0, s, n, person, S.r1.p, S.k0.p);
    PR_PRIM(tmp, prim::sort_pairs(p_, bytes_, S.k0.p, S.k1.p, S.r1.p, S.r0.p, (int)n, 0, 64, s));
    S.rows = S.r0.p;
    return LOCREC_OK;
}

// ---- calcRatings ---------------------------------------------------------------------------------

// flags of the sorted rows: a new (person, entity) group / a new person
__global__ void pr_group_flags(int64_t n, const int64_t *person, const int64_t *entity, const uint32_t *rows,
                               unsigned char *gfirst, uint32_t *pfirst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t p = person[rows[i]], e = entity[rows[i]];
    bool np = true, ng = true;
    if (i > 0) {
        const int64_t pp = person[rows[i - 1]], pe = entity[rows[i - 1]];
        np = pp != p;
        ng = np || pe != e;
    }
    gfirst[i] = ng ? 1 : 0;
    pfirst[i] = np ? 1u : 0u;
}

// per group: its visit count and the key (person rank, count descending)
__global__ void pr_group_keys(int64_t g, int64_t n, const uint32_t *gstart, const uint32_t *prank_of_pos, uint32_t *cnt,
                              uint64_t *keys, uint32_t *gid)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g) return;
    const uint32_t b = gstart[i];
    const uint32_t e = i + 1 < g ? gstart[i + 1] : (uint32_t)n;
    const uint32_t c = e - b;
    cnt[i] = c;
    keys[i] = ((uint64_t)(prank_of_pos[b] - 1u) << 32) | (uint32_t)(~c);
    gid[i] = (uint32_t)i;
}

// over the groups sorted by (person, count desc): position of the person's first group and of the
// first group of the run of equal counts (as values for two running maxima)
__global__ void pr_run_marks(int64_t g, const uint64_t *keys, uint32_t *pmark, uint32_t *rmark)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g) return;
    const bool np = i == 0 || (keys[i] >> 32) != (keys[i - 1] >> 32);
    const bool nr = i == 0 || keys[i] != keys[i - 1];
    pmark[i] = np ? (uint32_t)i : 0u;
    rmark[i] = nr ? (uint32_t)i : 0u;
}

// SQL rank() = 1 + rows of the partition sorting strictly before = 1 + (first of the run - first of
// the person); where(rank <= topN)
__global__ void pr_keep_top(int64_t g, const uint32_t *pstart, const uint32_t *rstart, const uint32_t *gid, int64_t top_n,
                            uint32_t *keep)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g) return;
    const int64_t rank = 1 + (int64_t)(rstart[i] - pstart[i]);
    keep[gid[i]] = rank <= top_n ? 1u : 0u;
}

__global__ void pr_emit_ratings(int64_t g, const uint32_t *keep, const uint32_t *pos, const uint32_t *gstart,
                                const uint32_t *rows, const int64_t *person, const int64_t *entity, const uint32_t *cnt,
                                int64_t *out_person, int64_t *out_entity, int64_t *out_rating)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g || !keep[i]) return;
    const uint32_t r = rows[gstart[i]], at = pos[i];
    out_person[at] = person[r];
    out_entity[at] = entity[r];
    out_rating[at] = (int64_t)cnt[i];
}

int32_t calc_ratings(int64_t n, const int64_t *person, const int64_t *entity, int64_t top_n, int64_t *out_person,
                     int64_t *out_entity, int64_t *out_rating, int64_t *out_count, hipStream_t s)
{
    Temp tmp;
    SortedRows S;
    LOCREC_TRY(sort_person_entity(n, person, entity, S, tmp, s));
    DevBuf<unsigned char> gfirst;
    DevBuf<uint32_t> pfirst, prank, gstart, ng_dev;
    LOCREC_TRY(gfirst.alloc((size_t)n));
    LOCREC_TRY(pfirst.alloc((size_t)n));
    LOCREC_TRY(prank.alloc((size_t)n));
    LOCREC_TRY(gstart.alloc((size_t)n));
    LOCREC_TRY(ng_dev.alloc(1));
    hipLaunchKernelGGL(pr_group_flags, grid_for(n), dim3(256), 0, s, n, person, entity, S.rows, gfirst.p, pfirst.p);
    PR_PRIM(tmp, prim::inclusive_sum(p_, bytes_, pfirst.p, prank.p, (int)n, s));
    prim::counting_iterator<uint32_t> iota(0u);
    PR_PRIM(tmp, prim::select_flagged(p_, bytes_, iota, gfirst.p, gstart.p, ng_dev.p, (int)n, s));
    uint32_t g32 = 0;
    LOCREC_HIP_TRY(hipMemcpyAsync(&g32, ng_dev.p, sizeof g32, hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    const int64_t g = g32;

    DevBuf<uint32_t> cnt, gid0, gid1, pmark, rmark, pstart, rstart, keep, pos;
    DevBuf<uint64_t> key0, key1;
    LOCREC_TRY(cnt.alloc((size_t)g));
    LOCREC_TRY(gid0.alloc((size_t)g));
    LOCREC_TRY(gid1.alloc((size_t)g));
    LOCREC_TRY(key0.alloc((size_t)g));
    LOCREC_TRY(key1.alloc((size_t)g));
    LOCREC_TRY(pmark.alloc((size_t)g));
    LOCREC_TRY(rmark.alloc((size_t)g));
    LOCREC_TRY(pstart.alloc((size_t)g));
    LOCREC_TRY(rstart.alloc((size_t)g));
    LOCREC_TRY(keep.alloc((size_t)g));
    LOCREC_TRY(pos.alloc((size_t)g + 1));
    hipLaunchKernelGGL(pr_group_keys, grid_for(g), dim3(256), 0, s, g, n, gstart.p, prank.p, cnt.p, key0.p, gid0.p);
    PR_PRIM(tmp, prim::sort_pairs(p_, bytes_, key0.p, key1.p, gid0.p, gid1.p, (int)g, 0, 64, s));
    hipLaunchKernelGGL(pr_run_marks, grid_for(g), dim3(256), 0, s, g, key1.p, pmark.p, rmark.p);
    PR_PRIM(tmp, prim::inclusive_max(p_, bytes_, pmark.p, pstart.p, (int)g, s));
    PR_PRIM(tmp, prim::inclusive_max(p_, bytes_, rmark.p, rstart.p, (int)g, s));
    hipLaunchKernelGGL(pr_keep_top, grid_for(g), dim3(256), 0, s, g, pstart.p, rstart.p, gid1.p, top_n, keep.p);
    PR_PRIM(tmp, prim::exclusive_sum(p_, bytes_, keep.p, pos.p, (int)g, s));
    hipLaunchKernelGGL(pr_emit_ratings, grid_for(g), dim3(256), 0, s, g, keep.p, pos.p, gstart.p, S.rows, person, entity,
                       cnt.p, out_person, out_entity, out_rating);
    uint32_t last_pos = 0, last_keep = 0;
    LOCREC_HIP_TRY(hipMemcpyAsync(&last_pos, pos.p + (g - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipMemcpyAsync(&last_keep, keep.p + (g - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    *out_count = (int64_t)last_pos + last_keep;
    return LOCREC_OK;
}

// ---- calcRatingVectors ---------------------------------------------------------------------------

struct RangeOut {
    unsigned long long max_key, min_key;  // ordered_key() of the largest / smallest entity id
};

__global__ void pr_entity_range(int64_t n, const int64_t *entity, RangeOut *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long k = i < n ? ordered_key(entity[i]) : ordered_key(entity[0]);
    unsigned long long mx = k, mn = k;
    for (int off = 32; off > 0; off >>= 1) {
        mx = max(mx, (unsigned long long)__shfl_xor((long long)mx, off));
        mn = min(mn, (unsigned long long)__shfl_xor((long long)mn, off));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&out->max_key, mx);
        atomicMin(&out->min_key, mn);
    }
}

// flags of the sorted rows: first row of a person / a row that the person's TreeSet keeps (the first
// of equal indices, in input order: the sort is stable)
__global__ void pr_vector_flags(int64_t n, const int64_t *person, const int64_t *entity, const uint32_t *rows,
                                uint32_t *pfirst, uint32_t *keep)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool np = true, dup = false;
    if (i > 0) {
        np = person[rows[i - 1]] != person[rows[i]];
        dup = !np && entity[rows[i - 1]] == entity[rows[i]];
    }
    pfirst[i] = np ? 1u : 0u;
    keep[i] = dup ? 0u : 1u;
}

__global__ void pr_emit_vectors(int64_t n, const uint32_t *rows, const uint32_t *pfirst, const uint32_t *keep,
                                const uint32_t *prank, const uint32_t *pos, const int64_t *person, const int64_t *entity,
                                const int64_t *rating, int64_t *out_ids, int64_t *out_rowptr, int32_t *out_idx, double *out_val)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !keep[i]) return;
    const uint32_t r = rows[i], at = pos[i];
    out_idx[at] = (int32_t)entity[r];
    out_val[at] = (double)rating[r];  // rating.toDouble (RatingVectorsBuilder.scala:69)
    if (pfirst[i]) {
        out_ids[prank[i] - 1u] = person[r];
        out_rowptr[prank[i] - 1u] = (int64_t)at;
    }
}

__global__ void pr_set_i64(int64_t *p, int64_t v) { *p = v; }

int32_t calc_rating_vectors(int64_t n, const int64_t *person, const int64_t *entity, const int64_t *rating, int64_t *out_ids,
                            int64_t *out_rowptr, int32_t *out_idx, double *out_val, int64_t *out_npersons, int64_t *out_nnz,
                            int64_t *out_size, hipStream_t s)
{
    Temp tmp;
    DevBuf<RangeOut> range;
    LOCREC_TRY(range.alloc(1));
    const RangeOut init{0ull, ~0ull};
    LOCREC_HIP_TRY(hipMemcpyAsync(range.p, &init, sizeof init, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(pr_entity_range, grid_for(n), dim3(256), 0, s, n, entity, range.p);
    RangeOut got;
    LOCREC_HIP_TRY(hipMemcpyAsync(&got, range.p, sizeof got, hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    const int64_t max_id = (int64_t)(got.max_key ^ 0x8000000000000000ull), min_id = (int64_t)(got.min_key ^ 0x8000000000000000ull);
    // checkedCast (RatingVectorsBuilder.scala:36-41): of max(id) first (:27-34), then of every id (:69)
    if (max_id > std::numeric_limits<int32_t>::max() || max_id < std::numeric_limits<int32_t>::min())
        return fail(LOCREC_E_ARITHMETIC, "Index out of Int range: %lld", (long long)max_id);
    if (min_id < std::numeric_limits<int32_t>::min())
        return fail(LOCREC_E_ARITHMETIC, "Index out of Int range: %lld", (long long)min_id);
    // the SparseVector constructor's own require()s (third party: spark-mllib-local_2.12 3.1.2, ml/linalg/Vectors.scala)
    if (min_id < 0) return fail(LOCREC_E_INVALID_ARG, "requirement failed: Found negative index: %lld.", (long long)min_id);
    if (max_id == std::numeric_limits<int32_t>::max())  // checkedCast(maxId) + 1 wraps to Int.MinValue (:34)
        return fail(LOCREC_E_INVALID_ARG, "requirement failed: The size of the requested sparse vector must be no less than 0.");

    SortedRows S;
    LOCREC_TRY(sort_person_entity(n, person, entity, S, tmp, s));
    DevBuf<uint32_t> pfirst, keep, prank, pos;
    LOCREC_TRY(pfirst.alloc((size_t)n));
    LOCREC_TRY(keep.alloc((size_t)n));
    LOCREC_TRY(prank.alloc((size_t)n));
    LOCREC_TRY(pos.alloc((size_t)n));
    hipLaunchKernelGGL(pr_vector_flags, grid_for(n), dim3(256), 0, s, n, person, entity, S.rows, pfirst.p, keep.p);
    PR_PRIM(tmp, prim::inclusive_sum(p_, bytes_, pfirst.p, prank.p, (int)n, s));
    PR_PRIM(tmp, prim::exclusive_sum(p_, bytes_, keep.p, pos.p, (int)n, s));
    hipLaunchKernelGGL(pr_emit_vectors, grid_for(n), dim3(256), 0, s, n, S.rows, pfirst.p, keep.p, prank.p, pos.p, person,
                       entity, rating, out_ids, out_rowptr, out_idx, out_val);
    uint32_t tail[3] = {0, 0, 0};
    LOCREC_HIP_TRY(hipMemcpyAsync(&tail[0], prank.p + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipMemcpyAsync(&tail[1], pos.p + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipMemcpyAsync(&tail[2], keep.p + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    *out_npersons = tail[0];
    *out_nnz = (int64_t)tail[1] + tail[2];
    *out_size = max_id + 1;
    hipLaunchKernelGGL(pr_set_i64, dim3(1), dim3(1), 0, s, out_rowptr + *out_npersons, *out_nnz);
    return LOCREC_OK;
}

// ---- buildWithBalancedWeights --------------------------------------------------------------------

__global__ void pr_balance(int64_t n, const int64_t *src, const int64_t *dst, const double *w, double beta, int64_t *out_src,
                           int64_t *out_dst, double *out_w)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_src[i] = src[i];
    out_dst[i] = dst[i];
    out_w[i] = w[i] * beta;  // col("weight") * beta (StochasticGraphBuilder.scala:14,23)
}

// ---- calcPlaceVisits -----------------------------------------------------------------------------

constexpr double kEarthRadiusMeters = 6371.0 * 1000.0;  // Location.scala:28
constexpr double kPi = 3.14159265358979323846;
constexpr int kCellBits = 20;                            // bands and cells per band: < 2^20 each; regions < 2^24

__device__ __forceinline__ double to_radians(double deg) { return deg / 180.0 * kPi; }

__device__ __forceinline__ double haversine(double theta)  // Location.scala:40-43
{
    const double sn = sin(theta / 2);
    return sn * sn;
}

__device__ double distance_meters(double lat1d, double lon1d, double lat2d, double lon2d)  // Location.scala:30-38
{
    const double lat1 = to_radians(lat1d), lat2 = to_radians(lat2d);
    const double lon1 = to_radians(lon1d), lon2 = to_radians(lon2d);
    const double h1 = haversine(lat2 - lat1);
    const double cc = cos(lat1) * cos(lat2);
    const double h2 = cc * haversine(lon2 - lon1);
    const double hav = h1 + h2;
    return (kEarthRadiusMeters * 2) * asin(sqrt(hav));
}

__device__ __forceinline__ bool location_ok(double lat, double lon)  // Location.scala:7-8 (NaN fails the require)
{
    return lat >= -90.0 && lat <= 90.0 && lon >= -180.0 && lon <= 180.0;
}

// The grid.  Latitude bands of `band_deg` degrees (>= the search radius as an angle, so a match lies in
// the visit's band or a neighbouring one).  Band b is cut into nx(b) longitude cells of 360 / nx(b)
// degrees, at least as wide as the largest longitude difference a match can have when the place is in
// band b and the visit in bands b - 1 .. b + 1:  hav(d / R) >= cos(lat1) cos(lat2) hav(dlon)  =>
// sin(dlon / 2) <= sin(d / 2R) / cos(latmax).  Near the poles that bound exceeds 1: one cell.
struct Grid {
    double band_deg, sin_half;  // sin(d / 2R), with a 1e-9 relative safety margin
    int32_t nbands;
};

__device__ __forceinline__ int32_t band_of(const Grid &g, double lat)
{
    const int32_t b = (int32_t)floor((lat + 90.0) / g.band_deg);
    return min(max(b, 0), g.nbands - 1);
}

// longitude half-window (degrees) and cell count of band b
__device__ __forceinline__ void band_cells(const Grid &g, int32_t b, double *half_window_deg, int32_t *nx)
{
    const double lo = -90.0 + (b - 1) * g.band_deg, hi = -90.0 + (b + 2) * g.band_deg;
    const double latmax = fmin(fmax(fabs(lo), fabs(hi)), 90.0);
    const double c = cos(to_radians(latmax));
    double win = 180.0;
    if (c > 0.0) {
        const double ratio = g.sin_half / c;
        if (ratio < 1.0) win = fmin(180.0, 2.0 * asin(ratio) * (180.0 / kPi) * (1.0 + 1e-9) + 1e-12);
    }
    *half_window_deg = win;
    *nx = (int32_t)fmin(fmax(floor(360.0 / win), 1.0), (double)((1 << kCellBits) - 1));
}

__device__ __forceinline__ int64_t rank_of_region(const int64_t *regions, int32_t nr, int64_t region)
{
    int32_t lo = 0, hi = nr;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (regions[mid] < region) lo = mid + 1; else hi = mid;
    }
    return lo < nr && regions[lo] == region ? lo : -1;
}

__global__ void pr_place_keys(int64_t np, const double *lat, const double *lon, const int64_t *region, const int64_t *regions,
                              int32_t nr, Grid g, uint64_t *keys, uint32_t *rows)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= np) return;
    rows[j] = (uint32_t)j;
    const int64_t r = rank_of_region(regions, nr, region[j]);
    if (!location_ok(lat[j], lon[j])) {  // never a match; reported by pr_check_places if a visit would meet it
        keys[j] = ~0ull;
        return;
    }
    const int32_t b = band_of(g, lat[j]);
    double win;
    int32_t nx;
    band_cells(g, b, &win, &nx);
    const double w = 360.0 / nx;
    const int32_t cx = min(max((int32_t)floor((lon[j] + 180.0) / w), 0), nx - 1);
    keys[j] = ((uint64_t)r << (2 * kCellBits)) | ((uint64_t)b << kCellBits) | (uint64_t)cx;
}

struct JoinError {
    unsigned long long first_bad_visit, first_bad_place;  // ~0 = none
};

// a visit that passes the time filter and whose region has places takes part in the join: mark its region,
// and its Location must be valid (the reference's UDF constructs it for every joined pair)
__global__ void pr_check_visits(int64_t nv, const int64_t *ts, const double *lat, const double *lon, const int64_t *region,
                                int64_t visits_from, const int64_t *regions, int32_t nr, uint32_t *region_visited,
                                JoinError *err)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv || ts[i] < visits_from) return;
    const int64_t r = rank_of_region(regions, nr, region[i]);
    if (r < 0) return;
    region_visited[r] = 1u;
    if (!location_ok(lat[i], lon[i])) atomicMin(&err->first_bad_visit, (unsigned long long)i);
}

__global__ void pr_check_places(int64_t np, const double *lat, const double *lon, const int64_t *region, const int64_t *regions,
                                int32_t nr, const uint32_t *region_visited, JoinError *err)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= np) return;
    const int64_t r = rank_of_region(regions, nr, region[j]);
    if (r >= 0 && region_visited[r] && !location_ok(lat[j], lon[j])) atomicMin(&err->first_bad_place, (unsigned long long)j);
}

__device__ __forceinline__ int64_t lower_bound_key(const uint64_t *keys, int64_t n, uint64_t key)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// One thread per visit: the places of the (at most) 3 bands x 3 cells around it, exact distance each.
// WRITE = false counts the matches; WRITE = true stores the place rows at the visit's offset (ascending
// place row: the candidates of a visit are few, an insertion sort orders them) and fills the columns.
template <bool WRITE>
__global__ void pr_join(int64_t nv, const int64_t *v_person, const int64_t *v_ts, const double *v_lat, const double *v_lon,
                        const int64_t *v_region, int64_t visits_from, const int64_t *regions, int32_t nr, Grid g,
                        double max_meters, int64_t np, const uint64_t *keys, const uint32_t *place_rows, const int64_t *p_id,
                        const double *p_lat, const double *p_lon, const int64_t *p_category, unsigned long long *counts,
                        const unsigned long long *offsets, int64_t cap, uint32_t *scratch_rows, int64_t *out_person,
                        int64_t *out_ts, int64_t *out_place, int64_t *out_region, int64_t *out_category)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    unsigned long long found = 0;
    const unsigned long long base = WRITE ? offsets[i] : 0ull;
    if (v_ts[i] >= visits_from) {  // locationVisits.where(timestamp >= visitsFrom) (PlaceVisits.scala:24)
        const int64_t r = rank_of_region(regions, nr, v_region[i]);  // join(places, "region_id") (:31)
        const double lat = v_lat[i], lon = v_lon[i];
        if (r >= 0 && location_ok(lat, lon)) {
            const int32_t bv = band_of(g, lat);
            for (int32_t b = max(bv - 1, 0); b <= min(bv + 1, g.nbands - 1); ++b) {
                double win;
                int32_t nx;
                band_cells(g, b, &win, &nx);
                const double w = 360.0 / nx;
                const int64_t c_lo = (int64_t)floor((lon - win + 180.0) / w), c_hi = (int64_t)floor((lon + win + 180.0) / w);
                const int64_t ncell = min(c_hi - c_lo + 1, (int64_t)nx);
                for (int64_t t = 0; t < ncell; ++t) {
                    const int64_t cx = ((c_lo + t) % nx + nx) % nx;  // cells wrap around the antimeridian
                    const uint64_t key = ((uint64_t)r << (2 * kCellBits)) | ((uint64_t)b << kCellBits) | (uint64_t)cx;
                    for (int64_t at = lower_bound_key(keys, np, key); at < np && keys[at] == key; ++at) {
                        const uint32_t j = place_rows[at];
                        if (distance_meters(lat, lon, p_lat[j], p_lon[j]) <= max_meters) {  // (:15-21,127)
                            if (WRITE && base + found < (unsigned long long)cap) scratch_rows[base + found] = j;
                            ++found;
                        }
                    }
                }
            }
        }
    }
    if (!WRITE) {
        counts[i] = found;
        return;
    }
    const unsigned long long room = base < (unsigned long long)cap ? (unsigned long long)cap - base : 0ull;
    const unsigned long long m = min(found, room);
    uint32_t *mine = scratch_rows + base;
    for (unsigned long long a = 1; a < m; ++a) {
        const uint32_t v = mine[a];
        unsigned long long b = a;
        for (; b > 0 && mine[b - 1] > v; --b) mine[b] = mine[b - 1];
        mine[b] = v;
    }
    for (unsigned long long a = 0; a < m; ++a) {  // select(person_id, timestamp, id as place_id, region_id, category_id) (:40-46)
        const uint32_t j = mine[a];
        out_person[base + a] = v_person[i];
        out_ts[base + a] = v_ts[i];
        out_place[base + a] = p_id[j];
        out_region[base + a] = v_region[i];
        out_category[base + a] = p_category[j];
    }
}

int32_t mem_ok(int32_t mem)
{
    if (mem != LOCREC_MEM_HOST && mem != LOCREC_MEM_DEVICE) return fail(LOCREC_E_INVALID_ARG, "mem must be LOCREC_MEM_HOST or LOCREC_MEM_DEVICE");
    return LOCREC_OK;
}

}  // namespace

extern "C" int32_t locrec_calc_ratings(int64_t n, const int64_t *person_ids, const int64_t *entity_ids, int64_t top_n,
                                       int32_t mem, int64_t *out_person_ids, int64_t *out_entity_ids, int64_t *out_ratings,
                                       int64_t *out_count)
try {
    LOCREC_TRY(mem_ok(mem));
    if (!out_count) return fail(LOCREC_E_INVALID_ARG, "out_count is required");
    *out_count = 0;
    if (n < 0 || n >= kMaxRows) return fail(LOCREC_E_INVALID_ARG, "visit count %lld out of range [0, 2^31)", (long long)n);
    if (n == 0) return LOCREC_OK;
    if (!person_ids || !entity_ids || !out_person_ids || !out_entity_ids || !out_ratings)
        return fail(LOCREC_E_INVALID_ARG, "null array");
    LOCREC_TRY(ensure_device());
    hipStream_t s = nullptr;
    In<int64_t> p, e;
    Out<int64_t> op, oe, orat;
    LOCREC_TRY(p.bind(person_ids, n, mem, s));
    LOCREC_TRY(e.bind(entity_ids, n, mem, s));
    LOCREC_TRY(op.bind(out_person_ids, n, mem));
    LOCREC_TRY(oe.bind(out_entity_ids, n, mem));
    LOCREC_TRY(orat.bind(out_ratings, n, mem));
    LOCREC_TRY(calc_ratings(n, p.p, e.p, top_n, op.p, oe.p, orat.p, out_count, s));
    LOCREC_TRY(op.deliver(*out_count, s));
    LOCREC_TRY(oe.deliver(*out_count, s));
    LOCREC_TRY(orat.deliver(*out_count, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    return LOCREC_OK;
}
LOCREC_CATCH_ALL

extern "C" int32_t locrec_calc_rating_vectors(int64_t n, const int64_t *person_ids, const int64_t *entity_ids,
                                              const int64_t *ratings, int32_t mem, int64_t *out_person_ids,
                                              int64_t *out_rowptr, int32_t *out_idx, double *out_val, int64_t *out_npersons,
                                              int64_t *out_nnz, int64_t *out_size)
try {
    LOCREC_TRY(mem_ok(mem));
    if (!out_npersons || !out_nnz || !out_size) return fail(LOCREC_E_INVALID_ARG, "count outputs are required");
    *out_npersons = *out_nnz = *out_size = 0;
    if (n < 0 || n >= kMaxRows) return fail(LOCREC_E_INVALID_ARG, "rating count %lld out of range [0, 2^31)", (long long)n);
    if (!out_rowptr) return fail(LOCREC_E_INVALID_ARG, "null array");
    LOCREC_TRY(ensure_device());
    hipStream_t s = nullptr;
    if (n == 0) {
        if (mem == LOCREC_MEM_DEVICE) {
            hipLaunchKernelGGL(pr_set_i64, dim3(1), dim3(1), 0, s, out_rowptr, (int64_t)0);
            LOCREC_HIP_TRY(hipStreamSynchronize(s));
        } else {
            out_rowptr[0] = 0;
        }
        return LOCREC_OK;
    }
    if (!person_ids || !entity_ids || !ratings || !out_person_ids || !out_idx || !out_val)
        return fail(LOCREC_E_INVALID_ARG, "null array");
    In<int64_t> p, e, r;
    Out<int64_t> oid, optr;
    Out<int32_t> oidx;
    Out<double> oval;
    LOCREC_TRY(p.bind(person_ids, n, mem, s));
    LOCREC_TRY(e.bind(entity_ids, n, mem, s));
    LOCREC_TRY(r.bind(ratings, n, mem, s));
    LOCREC_TRY(oid.bind(out_person_ids, n, mem));
    LOCREC_TRY(optr.bind(out_rowptr, n + 1, mem));
    LOCREC_TRY(oidx.bind(out_idx, n, mem));
    LOCREC_TRY(oval.bind(out_val, n, mem));
    LOCREC_TRY(calc_rating_vectors(n, p.p, e.p, r.p, oid.p, optr.p, oidx.p, oval.p, out_npersons, out_nnz, out_size, s));
    LOCREC_TRY(oid.deliver(*out_npersons, s));
    LOCREC_TRY(optr.deliver(*out_npersons + 1, s));
    LOCREC_TRY(oidx.deliver(*out_nnz, s));
    LOCREC_TRY(oval.deliver(*out_nnz, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    return LOCREC_OK;
}
LOCREC_CATCH_ALL

extern "C" int32_t locrec_build_balanced_edges(int32_t n_families, const double *betas, const int64_t *counts,
                                               const int64_t *const *source_ids, const int64_t *const *target_ids,
                                               const double *const *weights, int32_t mem, int64_t *out_source_ids,
                                               int64_t *out_target_ids, double *out_balanced_weights)
try {
    LOCREC_TRY(mem_ok(mem));
    // betas.head / allEdges.head of an empty Seq throw (StochasticGraphBuilder.scala:9-10)
    if (n_families <= 0 || !betas || !counts || !source_ids || !target_ids || !weights)
        return fail(LOCREC_E_INVALID_ARG, "one beta per edge family is required");
    int64_t total = 0;
    for (int32_t f = 0; f < n_families; ++f) {
        if (counts[f] < 0) return fail(LOCREC_E_INVALID_ARG, "negative edge count in family %d", f);
        total += counts[f];
    }
    if (total == 0) return LOCREC_OK;
    if (!out_source_ids || !out_target_ids || !out_balanced_weights) return fail(LOCREC_E_INVALID_ARG, "null array");
    LOCREC_TRY(ensure_device());
    hipStream_t s = nullptr;
    Out<int64_t> os, ot;
    Out<double> ow;
    LOCREC_TRY(os.bind(out_source_ids, total, mem));
    LOCREC_TRY(ot.bind(out_target_ids, total, mem));
    LOCREC_TRY(ow.bind(out_balanced_weights, total, mem));
    int64_t at = 0;
    for (int32_t f = 0; f < n_families; ++f) {
        const int64_t m = counts[f];
        if (m == 0) continue;
        if (!source_ids[f] || !target_ids[f] || !weights[f]) return fail(LOCREC_E_INVALID_ARG, "null array in family %d", f);
        In<int64_t> a, b;
        In<double> w;
        LOCREC_TRY(a.bind(source_ids[f], m, mem, s));
        LOCREC_TRY(b.bind(target_ids[f], m, mem, s));
        LOCREC_TRY(w.bind(weights[f], m, mem, s));
        hipLaunchKernelGGL(pr_balance, grid_for(m), dim3(256), 0, s, m, a.p, b.p, w.p, betas[f], os.p + at, ot.p + at, ow.p + at);
        LOCREC_HIP_TRY(hipStreamSynchronize(s));  // (the family's staging buffers are released at the end of this iteration)
        at += m;
    }
    LOCREC_TRY(os.deliver(total, s));
    LOCREC_TRY(ot.deliver(total, s));
    LOCREC_TRY(ow.deliver(total, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    return LOCREC_OK;
}
LOCREC_CATCH_ALL

extern "C" int32_t locrec_calc_place_visits(int64_t n_visits, const int64_t *v_person_ids, const int64_t *v_timestamps,
                                            const double *v_latitudes, const double *v_longitudes, const int64_t *v_region_ids,
                                            int64_t n_places, const int64_t *p_ids, const double *p_latitudes,
                                            const double *p_longitudes, const int64_t *p_region_ids,
                                            const int64_t *p_category_ids, int64_t visits_from, double max_meters, int32_t mem,
                                            int64_t *out_person_ids, int64_t *out_timestamps, int64_t *out_place_ids,
                                            int64_t *out_region_ids, int64_t *out_category_ids, int64_t *inout_count)
try {
    LOCREC_TRY(mem_ok(mem));
    if (!inout_count) return fail(LOCREC_E_INVALID_ARG, "inout_count is required");
    const int64_t cap = *inout_count;
    *inout_count = 0;
    if (cap < 0) return fail(LOCREC_E_INVALID_ARG, "negative capacity");
    if (n_visits < 0 || n_visits >= kMaxRows || n_places < 0 || n_places >= kMaxRows)
        return fail(LOCREC_E_INVALID_ARG, "row count out of range [0, 2^31)");
    if (!(max_meters >= 0.0) || !(max_meters < kEarthRadiusMeters))
        return fail(LOCREC_E_INVALID_ARG, "the search radius %g m must be within [0, earth radius)", max_meters);
    if (n_visits == 0 || n_places == 0) return LOCREC_OK;
    if (!v_person_ids || !v_timestamps || !v_latitudes || !v_longitudes || !v_region_ids || !p_ids || !p_latitudes ||
        !p_longitudes || !p_region_ids || !p_category_ids)
        return fail(LOCREC_E_INVALID_ARG, "null array");
    if (cap > 0 && (!out_person_ids || !out_timestamps || !out_place_ids || !out_region_ids || !out_category_ids))
        return fail(LOCREC_E_INVALID_ARG, "null output array");
    LOCREC_TRY(ensure_device());
    hipStream_t s = nullptr;
    Temp tmp;
    In<int64_t> vp, vt, vr, pi, pr, pc;
    In<double> vlat, vlon, plat, plon;
    LOCREC_TRY(vp.bind(v_person_ids, n_visits, mem, s));
    LOCREC_TRY(vt.bind(v_timestamps, n_visits, mem, s));
    LOCREC_TRY(vlat.bind(v_latitudes, n_visits, mem, s));
    LOCREC_TRY(vlon.bind(v_longitudes, n_visits, mem, s));
    LOCREC_TRY(vr.bind(v_region_ids, n_visits, mem, s));
    LOCREC_TRY(pi.bind(p_ids, n_places, mem, s));
    LOCREC_TRY(plat.bind(p_latitudes, n_places, mem, s));
    LOCREC_TRY(plon.bind(p_longitudes, n_places, mem, s));
    LOCREC_TRY(pr.bind(p_region_ids, n_places, mem, s));
    LOCREC_TRY(pc.bind(p_category_ids, n_places, mem, s));

    // distinct place regions, ascending
    DevBuf<uint64_t> k0, k1;
    DevBuf<uint32_t> r0, r1;
    DevBuf<int64_t> regions;
    DevBuf<int32_t> nr_dev;
    LOCREC_TRY(k0.alloc((size_t)n_places));
    LOCREC_TRY(k1.alloc((size_t)n_places));
    LOCREC_TRY(r0.alloc((size_t)n_places));
    LOCREC_TRY(r1.alloc((size_t)n_places));
    LOCREC_TRY(regions.alloc((size_t)n_places));
    LOCREC_TRY(nr_dev.alloc(1));
    hipLaunchKernelGGL(pr_iota_keys, grid_for(n_places), dim3(256), 0, s, n_places, pr.p, k0.p, r0.p);
    PR_PRIM(tmp, prim::sort_keys(p_, bytes_, k0.p, k1.p, (int)n_places, 0, 64, s));
    PR_PRIM(tmp, prim::unique(p_, bytes_, k1.p, k0.p, nr_dev.p, (int)n_places, s));
    int32_t nr = 0;
    LOCREC_HIP_TRY(hipMemcpyAsync(&nr, nr_dev.p, sizeof nr, hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    if (nr >= (1 << 24)) return fail(LOCREC_E_INVALID_ARG, "%d distinct regions: at most 2^24 - 1 are supported", nr);
    {   // ordered_key() back to the signed ids (x ^ sign bit is its own inverse and keeps the order)
        std::vector<uint64_t> hk((size_t)nr);
        LOCREC_HIP_TRY(hipMemcpy(hk.data(), k0.p, (size_t)nr * 8, hipMemcpyDeviceToHost));
        std::vector<int64_t> hr((size_t)nr);
        for (int32_t i = 0; i < nr; ++i) hr[(size_t)i] = (int64_t)(hk[(size_t)i] ^ 0x8000000000000000ull);
        LOCREC_HIP_TRY(hipMemcpy(regions.p, hr.data(), (size_t)nr * 8, hipMemcpyHostToDevice));
    }

    Grid g;
    const double ang = max_meters / kEarthRadiusMeters;                       // the radius as an angle
    g.band_deg = std::max(ang * (180.0 / kPi) * (1.0 + 1e-9) + 1e-12, 180.0 / (double)((1 << kCellBits) - 2));
    g.nbands = (int32_t)std::floor(180.0 / g.band_deg) + 1;
    g.sin_half = std::sin(ang / 2) * (1.0 + 1e-9);

    DevBuf<uint32_t> visited;
    DevBuf<JoinError> err;
    LOCREC_TRY(visited.alloc((size_t)nr));
    LOCREC_TRY(err.alloc(1));
    LOCREC_HIP_TRY(hipMemsetAsync(visited.p, 0, (size_t)nr * 4, s));
    LOCREC_HIP_TRY(hipMemsetAsync(err.p, 0xFF, sizeof(JoinError), s));
    hipLaunchKernelGGL(pr_check_visits, grid_for(n_visits), dim3(256), 0, s, n_visits, vt.p, vlat.p, vlon.p, vr.p, visits_from,
                       regions.p, nr, visited.p, err.p);
    hipLaunchKernelGGL(pr_check_places, grid_for(n_places), dim3(256), 0, s, n_places, plat.p, plon.p, pr.p, regions.p, nr,
                       visited.p, err.p);
    JoinError je;
    LOCREC_HIP_TRY(hipMemcpyAsync(&je, err.p, sizeof je, hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    if (je.first_bad_visit != ~0ull || je.first_bad_place != ~0ull) {
        const bool visit = je.first_bad_visit != ~0ull;
        const int64_t row = (int64_t)(visit ? je.first_bad_visit : je.first_bad_place);
        double lat = 0, lon = 0;
        LOCREC_HIP_TRY(hipMemcpy(&lat, (visit ? vlat.p : plat.p) + row, 8, hipMemcpyDeviceToHost));
        LOCREC_HIP_TRY(hipMemcpy(&lon, (visit ? vlon.p : plon.p) + row, 8, hipMemcpyDeviceToHost));
        *inout_count = visit ? -(1 + row) : -(1 + n_visits + row);
        if (!(lat >= -90.0 && lat <= 90.0))  // the reference's messages (Location.scala:7-8)
            return fail(LOCREC_E_INVALID_ARG, "requirement failed: Latitude %.17g must be within range [-90.0, 90.0] (%s %lld)", lat,
                        visit ? "location visit" : "place", (long long)row);
        return fail(LOCREC_E_INVALID_ARG, "requirement failed: Longitude %.17g must be within range [-180.0, 180.0] (%s %lld)", lon,
                    visit ? "location visit" : "place", (long long)row);
    }

    hipLaunchKernelGGL(pr_place_keys, grid_for(n_places), dim3(256), 0, s, n_places, plat.p, plon.p, pr.p, regions.p, nr, g, k0.p,
                       r0.p);
    PR_PRIM(tmp, prim::sort_pairs(p_, bytes_, k0.p, k1.p, r0.p, r1.p, (int)n_places, 0, 64, s));

    DevBuf<unsigned long long> counts, offsets;
    LOCREC_TRY(counts.alloc((size_t)n_visits));
    LOCREC_TRY(offsets.alloc((size_t)n_visits));
    hipLaunchKernelGGL((pr_join<false>), grid_for(n_visits), dim3(256), 0, s, n_visits, vp.p, vt.p, vlat.p, vlon.p, vr.p,
                       visits_from, regions.p, nr, g, max_meters, n_places, k1.p, r1.p, pi.p, plat.p, plon.p, pc.p, counts.p,
                       nullptr, (int64_t)0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    PR_PRIM(tmp, prim::exclusive_sum(p_, bytes_, counts.p, offsets.p, (int)n_visits, s));
    unsigned long long last_off = 0, last_cnt = 0;
    LOCREC_HIP_TRY(hipMemcpyAsync(&last_off, offsets.p + (n_visits - 1), 8, hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipMemcpyAsync(&last_cnt, counts.p + (n_visits - 1), 8, hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    const int64_t total = (int64_t)(last_off + last_cnt);
    *inout_count = total;
    const int64_t rows = std::min(total, cap);
    if (rows == 0) return LOCREC_OK;

    Out<int64_t> op, ots, opl, org, oca;
    DevBuf<uint32_t> scratch;
    LOCREC_TRY(op.bind(out_person_ids, rows, mem));
    LOCREC_TRY(ots.bind(out_timestamps, rows, mem));
    LOCREC_TRY(opl.bind(out_place_ids, rows, mem));
    LOCREC_TRY(org.bind(out_region_ids, rows, mem));
    LOCREC_TRY(oca.bind(out_category_ids, rows, mem));
    LOCREC_TRY(scratch.alloc((size_t)rows));
    hipLaunchKernelGGL((pr_join<true>), grid_for(n_visits), dim3(256), 0, s, n_visits, vp.p, vt.p, vlat.p, vlon.p, vr.p,
                       visits_from, regions.p, nr, g, max_meters, n_places, k1.p, r1.p, pi.p, plat.p, plon.p, pc.p, nullptr,
                       offsets.p, rows, scratch.p, op.p, ots.p, opl.p, org.p, oca.p);
    LOCREC_TRY(op.deliver(rows, s));
    LOCREC_TRY(ots.deliver(rows, s));
    LOCREC_TRY(opl.deliver(rows, s));
    LOCREC_TRY(org.deliver(rows, s));
    LOCREC_TRY(oca.deliver(rows, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    return LOCREC_OK;
}
LOCREC_CATCH_ALL

// Location.distanceMeters (Location.scala:30-38) of n pairs, computed by the device code the join uses.
__global__ void pr_distances(int64_t n, const double *lat1, const double *lon1, const double *lat2, const double *lon2, double *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = location_ok(lat1[i], lon1[i]) && location_ok(lat2[i], lon2[i])
                            ? distance_meters(lat1[i], lon1[i], lat2[i], lon2[i])
                            : __longlong_as_double(0x7FF8000000000000ll);
}

extern "C" int32_t locrec_distance_meters(int64_t n, const double *lat1, const double *lon1, const double *lat2,
                                          const double *lon2, int32_t mem, double *out_meters)
try {
    LOCREC_TRY(mem_ok(mem));
    if (n < 0 || n >= kMaxRows) return fail(LOCREC_E_INVALID_ARG, "pair count out of range [0, 2^31)");
    if (n == 0) return LOCREC_OK;
    if (!lat1 || !lon1 || !lat2 || !lon2 || !out_meters) return fail(LOCREC_E_INVALID_ARG, "null array");
    LOCREC_TRY(ensure_device());
    hipStream_t s = nullptr;
    In<double> a, b, c, d;
    Out<double> o;
    LOCREC_TRY(a.bind(lat1, n, mem, s));
    LOCREC_TRY(b.bind(lon1, n, mem, s));
    LOCREC_TRY(c.bind(lat2, n, mem, s));
    LOCREC_TRY(d.bind(lon2, n, mem, s));
    LOCREC_TRY(o.bind(out_meters, n, mem));
    hipLaunchKernelGGL(pr_distances, grid_for(n), dim3(256), 0, s, n, a.p, b.p, c.p, d.p, o.p);
    LOCREC_TRY(o.deliver(n, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    return LOCREC_OK;
}
LOCREC_CATCH_ALL

// ---- the mains' final ranking ----------------------------------------------------
// printRecommendations of both mains (KnnRecommenderMain.scala:90-101, StochasticRecommenderMain.scala:64-75):
// places.where(region_id === target) JOIN recommendations ON id, ORDER BY score DESC, LIMIT n.  Rows whose id
// is not a place of the target region (persons, categories, places elsewhere) drop out in the join.

__global__ void pr_region_place_keys(int64_t np, const int64_t *place_ids, const int64_t *place_regions, int64_t target,
                                     uint64_t *keys, unsigned char *in_region)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= np) return;
    keys[j] = ordered_key(place_ids[j]);
    in_region[j] = place_regions[j] == target ? 1 : 0;
}

// key of a kept row: score descending (bit pattern of a double made monotone, then inverted), id ascending is
// the second, earlier sort pass; rows that are not places of the region get the flag 0
__global__ void pr_rank_flags(int64_t n, const int64_t *ids, const uint64_t *allowed, int32_t nallowed, unsigned char *keep)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = ordered_key(ids[i]);
    int32_t lo = 0, hi = nallowed;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (allowed[mid] < k) lo = mid + 1; else hi = mid;
    }
    keep[i] = lo < nallowed && allowed[lo] == k ? 1 : 0;
}

__device__ __forceinline__ uint64_t score_desc_key(double s)
{
    uint64_t b = (uint64_t)__double_as_longlong(s);
    b = (b >> 63) ? ~b : b | 0x8000000000000000ull;  // ascending order of the doubles (NaN sorts above +inf, as in Spark)
    return ~b;                                       // ... descending
}

__global__ void pr_rank_keys_by_id(int64_t m, const uint32_t *rows, const int64_t *ids, uint64_t *keys)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) keys[i] = ordered_key(ids[rows[i]]);
}

__global__ void pr_rank_keys_by_score(int64_t m, const uint32_t *rows, const double *scores, uint64_t *keys)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) keys[i] = score_desc_key(scores[rows[i]]);
}

__global__ void pr_rank_emit(int64_t w, const uint32_t *rows, const int64_t *ids, const double *scores, int64_t *out_ids,
                             double *out_scores)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w) return;
    out_ids[i] = ids[rows[i]];
    out_scores[i] = scores[rows[i]];
}

extern "C" int32_t locrec_rank_recommendations(int64_t n, const int64_t *ids, const double *scores, int64_t n_places,
                                               const int64_t *place_ids, const int64_t *place_region_ids,
                                               int64_t target_region_id, int64_t max_recommendations, int32_t mem,
                                               int64_t *out_ids, double *out_scores, int64_t *out_count)
try {
    LOCREC_TRY(mem_ok(mem));
    if (!out_count) return fail(LOCREC_E_INVALID_ARG, "out_count is required");
    *out_count = 0;
    if (n < 0 || n >= kMaxRows || n_places < 0 || n_places >= kMaxRows)
        return fail(LOCREC_E_INVALID_ARG, "row count out of range [0, 2^31)");
    const int64_t limit = std::max<int64_t>(0, max_recommendations);  // limit(n <= 0) is empty
    if (n == 0 || n_places == 0 || limit == 0) return LOCREC_OK;
    if (!ids || !scores || !place_ids || !place_region_ids || !out_ids || !out_scores) return fail(LOCREC_E_INVALID_ARG, "null array");
    LOCREC_TRY(ensure_device());
    hipStream_t s = nullptr;
    Temp tmp;
    In<int64_t> rid, pid, preg;
    In<double> rsc;
    LOCREC_TRY(rid.bind(ids, n, mem, s));
    LOCREC_TRY(rsc.bind(scores, n, mem, s));
    LOCREC_TRY(pid.bind(place_ids, n_places, mem, s));
    LOCREC_TRY(preg.bind(place_region_ids, n_places, mem, s));
    // the target region's place ids: selected, sorted, distinct
    DevBuf<uint64_t> a0, a1;
    DevBuf<unsigned char> in_region;
    DevBuf<int32_t> cnt_dev;
    LOCREC_TRY(a0.alloc((size_t)n_places));
    LOCREC_TRY(a1.alloc((size_t)n_places));
    LOCREC_TRY(in_region.alloc((size_t)n_places));
    LOCREC_TRY(cnt_dev.alloc(1));
    hipLaunchKernelGGL(pr_region_place_keys, grid_for(n_places), dim3(256), 0, s, n_places, pid.p, preg.p, target_region_id, a0.p,
                       in_region.p);
    PR_PRIM(tmp, prim::select_flagged(p_, bytes_, a0.p, in_region.p, a1.p, cnt_dev.p, (int)n_places, s));
    int32_t nallowed = 0;
    LOCREC_HIP_TRY(hipMemcpyAsync(&nallowed, cnt_dev.p, sizeof nallowed, hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    if (nallowed > 0) {
        PR_PRIM(tmp, prim::sort_keys(p_, bytes_, a1.p, a0.p, nallowed, 0, 64, s));
        PR_PRIM(tmp, prim::unique(p_, bytes_, a0.p, a1.p, cnt_dev.p, nallowed, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(&nallowed, cnt_dev.p, sizeof nallowed, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
        std::swap(a0.p, a1.p);  // a0 = the distinct sorted keys
        std::swap(a0.n, a1.n);
    }
    if (nallowed == 0) return LOCREC_OK;
    // rows of the recommendations that are places of the region
    DevBuf<unsigned char> keep;
    DevBuf<uint32_t> r0, r1;
    DevBuf<uint64_t> k0, k1;
    LOCREC_TRY(keep.alloc((size_t)n));
    LOCREC_TRY(r0.alloc((size_t)n));
    LOCREC_TRY(r1.alloc((size_t)n));
    hipLaunchKernelGGL(pr_rank_flags, grid_for(n), dim3(256), 0, s, n, rid.p, a0.p, nallowed, keep.p);
    prim::counting_iterator<uint32_t> iota(0u);
    PR_PRIM(tmp, prim::select_flagged(p_, bytes_, iota, keep.p, r0.p, cnt_dev.p, (int)n, s));
    int32_t m = 0;
    LOCREC_HIP_TRY(hipMemcpyAsync(&m, cnt_dev.p, sizeof m, hipMemcpyDeviceToHost, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    if (m == 0) return LOCREC_OK;
    // order by (score desc, id asc): stable LSD - by id, then by score
    LOCREC_TRY(k0.alloc((size_t)m));
    LOCREC_TRY(k1.alloc((size_t)m));
    hipLaunchKernelGGL(pr_rank_keys_by_id, grid_for(m), dim3(256), 0, s, (int64_t)m, r0.p, rid.p, k0.p);
    PR_PRIM(tmp, prim::sort_pairs(p_, bytes_, k0.p, k1.p, r0.p, r1.p, m, 0, 64, s));
    hipLaunchKernelGGL(pr_rank_keys_by_score, grid_for(m), dim3(256), 0, s, (int64_t)m, r1.p, rsc.p, k0.p);
    PR_PRIM(tmp, prim::sort_pairs(p_, bytes_, k0.p, k1.p, r1.p, r0.p, m, 0, 64, s));
    const int64_t w = std::min<int64_t>(m, limit);
    Out<int64_t> oid;
    Out<double> osc;
    LOCREC_TRY(oid.bind(out_ids, w, mem));
    LOCREC_TRY(osc.bind(out_scores, w, mem));
    hipLaunchKernelGGL(pr_rank_emit, grid_for(w), dim3(256), 0, s, w, r0.p, rid.p, rsc.p, oid.p, osc.p);
    LOCREC_TRY(oid.deliver(w, s));
    LOCREC_TRY(osc.deliver(w, s));
    LOCREC_HIP_TRY(hipStreamSynchronize(s));
    *out_count = w;
    return LOCREC_OK;
}
LOCREC_CATCH_ALL
