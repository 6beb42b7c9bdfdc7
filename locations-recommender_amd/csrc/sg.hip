// sg.hip -- stochastic-graph power iteration x <- alpha*u + (1-alpha)*P^T x on gfx950.
//
// Replaces the body of StochasticRecommender.makeRecommendations
// (recommender/src/main/scala/com/github/tashoyan/recommender/stochastic/
//  StochasticRecommender.scala:66-141): vertexes (:42-49), x0 (:51-54),
// calcNextX (:108-128), isConverged (:130-141), step (:92-106).
//
// Device layout (all built once in locrec_sg_create):
//
// 1. x is stored COMPACT.  A vertex without inbound edges has sigma = 0 forever, so
//    x'[v] = alpha*u[v] (:118-122): all such "source-only" vertices carry the SAME value
//    (1/V before the first sweep, 0 after it), except the request's own vertex (alpha).
//    x therefore has one entry per LIVE vertex (in-degree > 0, T of them, ascending id)
//    plus two shared slots: D = T for every source-only vertex and Q = T + 1 for the
//    request's vertex when it is source-only.  In the reference's graphs no edge ever
//    targets a person, so the gather table shrinks from V*8 B (2.3 MB at
//    cfg3) to (T+2)*8 B (80 KB): it stays in L2/L1 and the 64 lanes of a gather mostly
//    hit one address.  EVERY edge is still streamed and multiplied -- this is a layout,
//    not the "skip x[s] == 0" shortcut.  A request re-points its vertex's out-edge slots
//    from D to Q with a tiny patch kernel (and the next request points them back).
//
// 2. "pow2-segmented pieces".  The rows of P^T (one per live vertex: few, long, skewed)
//    are cut into PIECES of 256 edge slots = one 64-lane wave x 4 consecutive edges per
//    lane.  A row of in-degree d owns floor(d/256) FULL pieces plus one REMAINDER segment
//    of pow2 size (4..256 slots) that shares a piece with other remainders of the same
//    size class.  Every piece is homogeneous: 64 >> cls segments of (1 << cls) lanes, and
//    the in-wave reduction is a butterfly of exactly cls steps -- wave-uniform control
//    flow, no per-edge row ids, a fixed summation order (bitwise reproducible).
//      col[]  int32  [piece][lane][4]          one dwordx4 per lane, 1 KiB per wave
//      w[]    fp64   [piece][half][lane][2]    two dwordx4 per lane, 1 KiB each
//    Padding slots are (col D, w 0.0): they add x*0.0 = +0.0, exactly nothing.
//    Within a row the slots keep EDGE-LIST order and each lane adds its four products
//    left to right, so a row that fits one lane (in-degree <= 4) is summed in exactly the
//    order that reproduces the reference KATs bit for bit.
//
// One sweep = two launches (a kernel boundary, ~1.5 us, is the cheapest grid-wide sync):
//   sg_sweep     one wave per 4 pieces: stream col/w (non-temporal, so the stream does not
//                evict x from L2), gather x, multiply, butterfly, one partial per segment
//   sg_finalize  per live vertex: sigma = its partials in fixed order,
//                x' = u*alpha + sigma*(1-alpha), diff^2 -> kParts block sums
// The convergence test of sweep i (StochasticRecommender.scala:99) is evaluated
// redundantly by every wave at the START of the next sg_finalize from the kParts block
// sums, in one fixed order: no atomic, no host round trip, no extra launch.  A sticky
// `done` word turns the remaining launches of a batch into no-ops.  The host repeats
// the same fixed-order sum at the end to report the iteration count the reference prints.
//
// All arithmetic is fp64 with contraction off (the JVM never fuses a*b+c).

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>
#include <new>
#include <numeric>

#include "common.h"
#include "dev_prims.h"

namespace {

using namespace locrec;

constexpr int kSlots = 256;       // edge slots per piece (64 lanes x 4)
constexpr int kParts = 64;        // finalize blocks == diff^2 partial sums
constexpr int kMaxGraphRounds = 512;   // iterations per replayed hipGraph (even)
constexpr size_t kMaxRoundGraphs = 48; // cached hipGraphExec_t per handle
constexpr int kBeginBlocks = 32;  // blocks of a request's set-up launch (sg_begin), per graph
constexpr int kLongRow = 8;       // rows with more full pieces are summed by a whole wave
constexpr int kWaveRowLoads = 16; // partial loads a lane of such a wave has in flight (one round covers 1,024 pieces)
constexpr int kCheckEvery = 16;   // host looks at `done` this often when epsilon > 0
// pieces per wave of sg_sweep.  Measured per cfg3 ITERATION (sweep + finalize, no events):
// 1 -> 15.2 us, 2 -> 15.7 us, 4 -> 17.0 us; the grid-stride form sg_sweep_gs (LOCREC_SG_GS = blocks
// per CU) 2 -> 20.7, 4 -> 16.9, 6 -> 15.9, 8 -> 16.2 us.  Many short waves win: each wave is one
// dependent chain (stream loads -> gather -> butterfly -> store) and only more waves hide it.
constexpr int kPiecesPerWave = 1;
constexpr int kDictMax = 8192;     // distinct edge weights the dictionary form of the sweep takes (64 KB of LDS)

struct SgState {
    int32_t done;    // sticky: a converged sweep has been observed
    int32_t sweeps;  // number of executed finalize passes
};

struct RowMeta {
    int32_t full_begin;  // first partial index of the row's full pieces
    int32_t nfull;       // number of full pieces
    int32_t rem;         // partial index of the remainder segment, -1 if none
};

// One step of the xor butterfly, s[l] + s[l ^ (1 << STEP)], without the LDS round trips of __shfl_xor (two
// ds_bpermute per fp64 value and step: six dependent steps were ~1.3 us of every sweep wave's life,
// profiles/r03_sg_dict_knockouts.log).  Steps 0 and 1 are DPP quad permutes.  From step 2 on every lane of an aligned
// group of 1 << STEP lanes already holds the same sum, so ANY lane of the partner group serves: the mirror of a half row
// (lane l <- 7 - l) for step 2, of a row (l <- 15 - l) for step 3 - DPP again -, a swizzle across rows for step 4
// and the two half-wave values through SGPRs for step 5.  Same operands, same (commutative) addition: the same bits
// as the shuffle form.  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double s)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(s), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(s), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

template <int STEP>
__device__ __forceinline__ double butterfly_step(double s)
{
    if constexpr (STEP == 0) return s + dpp_f64<0xB1>(s);        // quad_perm [1, 0, 3, 2]
    else if constexpr (STEP == 1) return s + dpp_f64<0x4E>(s);   // quad_perm [2, 3, 0, 1]
    else if constexpr (STEP == 2) return s + dpp_f64<0x141>(s);  // row_half_mirror
    else if constexpr (STEP == 3) return s + dpp_f64<0x140>(s);  // row_mirror
    else if constexpr (STEP == 4) {
        const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(s), 0x401F);  // lane ^ 16
        const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(s), 0x401F);
        return s + __hiloint2double(hi, lo);
    } else {
        const double a = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 0), __builtin_amdgcn_readlane(__double2loint(s), 0));
        const double b = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 32), __builtin_amdgcn_readlane(__double2loint(s), 32));
        return (threadIdx.x & 32) ? b + a : a + b;
    }
}

__device__ __forceinline__ double wave_butterfly_sum(double s)
{
    s = butterfly_step<0>(s);
    s = butterfly_step<1>(s);
    s = butterfly_step<2>(s);
    s = butterfly_step<3>(s);
    s = butterfly_step<4>(s);
    return butterfly_step<5>(s);
}

// the butterfly over aligned segments of 1 << cls lanes (cls wave-uniform, 0 .. 6)
__device__ __forceinline__ double segment_butterfly_sum(double s, int cls)
{
    if (cls >= 1) s = butterfly_step<0>(s);
    if (cls >= 2) s = butterfly_step<1>(s);
    if (cls >= 3) s = butterfly_step<2>(s);
    if (cls >= 4) s = butterfly_step<3>(s);
    if (cls >= 5) s = butterfly_step<4>(s);
    if (cls >= 6) s = butterfly_step<5>(s);
    return s;
}

// Fixed-order total of the kParts (= 64) block sums; every lane returns the same bits.
// host_total_d2() repeats it on the CPU.
__device__ __forceinline__ double device_total_d2(const double *parts)
{
    return wave_butterfly_sum(parts[threadIdx.x & 63]);
}

double host_total_d2(const double *parts)
{
    double s[64];
    for (int l = 0; l < 64; ++l) s[l] = parts[l];
    for (int d = 1; d < 64; d <<= 1) {
        double t[64];
        for (int l = 0; l < 64; ++l) t[l] = s[l] + s[l ^ d];
        for (int l = 0; l < 64; ++l) s[l] = t[l];
    }
    return s[0];
}

// x0 (:51-54) and state reset.
// calcNextX, the sigma part (StochasticRecommender.scala:109-114).
// One wave owns kPiecesPerWave consecutive pieces and issues all of their loads before
// the first use.  The sweep never evaluates the convergence test itself: it only reads
// the sticky `done` word; sg_finalize decides.  A sweep launched after convergence only
// rewrites the scratch partials.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned short v4h __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// COL16: the compact x table has <= 65536 entries, so the column indices are stored as uint16
// (8 B per lane instead of 16: one sixth less traffic per sweep).
template <bool COL16, int PPW>
__device__ __forceinline__ void sg_sweep_body(
    const void *__restrict__ colv, const v2d *__restrict__ w2, const int2 *__restrict__ pinfo,
    const int32_t *__restrict__ seg_out, const double *__restrict__ x_in, double *__restrict__ partial,
    int32_t npieces, const SgState *__restrict__ st, const int p0 /* first piece of this wave, wave-uniform */)
{
    const int lane = threadIdx.x & 63;
    if (p0 >= npieces) return;
    const int done = st->done;
    int c[PPW][4];
    v2d wa[PPW], wb[PPW];
    int2 info[PPW];
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
        const int p = min(p0 + u, npieces - 1);  // clamped duplicates are computed and dropped
        if constexpr (COL16) {
            const v4h cc = __builtin_nontemporal_load(&reinterpret_cast<const v4h *>(colv)[(int64_t)p * 64 + lane]);
            c[u][0] = cc.x; c[u][1] = cc.y; c[u][2] = cc.z; c[u][3] = cc.w;
        } else {
            const v4i cc = __builtin_nontemporal_load(&reinterpret_cast<const v4i *>(colv)[(int64_t)p * 64 + lane]);
            c[u][0] = cc.x; c[u][1] = cc.y; c[u][2] = cc.z; c[u][3] = cc.w;
        }
        wa[u] = __builtin_nontemporal_load(&w2[(int64_t)p * 128 + lane]);
        wb[u] = __builtin_nontemporal_load(&w2[(int64_t)p * 128 + 64 + lane]);
        info[u] = pinfo[p];  // x = partial base, y = cls (log2 lanes per segment)
    }
    if (done) return;
    double xs[PPW][4];
    int tgt[PPW];  // row-major partial slot of this lane's segment (meaningful for its leader lane)
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xs[u][j] = x_in[c[u][j]];
        tgt[u] = seg_out[info[u].x + (lane >> info[u].y)];
    }
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
        if (p0 + u >= npieces) break;
        double s = xs[u][0] * wa[u].x;   // col("probability") * col("balanced_weight") (:112)
        s = s + xs[u][1] * wa[u].y;
        s = s + xs[u][2] * wb[u].x;
        s = s + xs[u][3] * wb[u].y;
        const int cls = __builtin_amdgcn_readfirstlane(info[u].y);
        s = segment_butterfly_sum(s, cls);
        if ((lane & ((1 << cls) - 1)) == 0 && tgt[u] >= 0) partial[tgt[u]] = s;
    }
}

template <bool COL16, int PPW>
__global__ __launch_bounds__(256) void sg_sweep(
    const void *__restrict__ colv, const v2d *__restrict__ w2, const int2 *__restrict__ pinfo,
    const int32_t *__restrict__ seg_out, const double *__restrict__ x_in, double *__restrict__ partial,
    int32_t npieces, const SgState *__restrict__ st)
{
    const int p0 = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW);
    sg_sweep_body<COL16, PPW>(colv, w2, pinfo, seg_out, x_in, partial, npieces, st, p0);
}

// every edge slot's index into the table of distinct weight bit patterns (bisection in the SORTED values, then the value's
// place in the frequency-ordered table), in slot order ([piece][lane][4], like
// the uint16 columns); the weights themselves are read in the sweep's own layout [piece][half][lane][2]
__global__ __launch_bounds__(256) void sg_build_widx(const double *w2, int64_t nslots, const uint64_t *sorted,
                                                     const unsigned short *rank, int32_t ndict, unsigned short *widx)
{
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nslots) return;
    const int64_t piece = slot >> 8;
    const int k = (int)(slot & 255), lane = k >> 2, j = k & 3;
    const uint64_t bits = (uint64_t)__double_as_longlong(w2[piece * kSlots + (j >> 1) * 128 + lane * 2 + (j & 1)]);
    int lo = 0, hi = ndict - 1;
    while (lo < hi) {  // (the value is in the table)
        const int mid = (lo + hi) >> 1;
        if (sorted[mid] < bits) lo = mid + 1; else hi = mid;
    }
    widx[slot] = rank[lo];
}

// Dictionary form of the sweep.  The balanced weights are (count / total of the source) x beta of the edge type: few
// distinct fp64 values (1,048 over 4.8 M edges at cfg3).  When there are at most kDictMax of them an edge streams a
// uint16 index instead of the fp64 weight - 4 B per edge slot instead of 10 - and the block keeps the value table in
// LDS (filled while the stream loads are in flight).  The looked-up value is the edge's own fp64 weight, bit for bit:
// same products, same sums, same order as sg_sweep.  cfg3: 11.4 -> 9.7-10.0 us per sweep (1024 threads, two pieces per
// wave; profiles/r03_sg_dict_forms.log) - 60 % fewer bytes buy 12 % because the sweep is a chain of latencies, not a
// stream: knocked out one by one, the stream loads, the x gather, the table lookup, the butterfly and the store are
// worth ~1 us each, and 5.1-5.3 us remain with ALL of them gone (profiles/r03_sg_dict_knockouts.log): what a launch of
// this many waves with one memory round trip costs on this chip.  (x in LDS as well - one block per CU holding the
// value table and all of x, 88 KB - measured 10.5 us: the LDS pipe then carries eight gathers per piece.)
template <bool COL16, int PPW>
__global__ __launch_bounds__(1024) void sg_sweep_dict(
    const void *__restrict__ colv, const v4h *__restrict__ widx, const double *__restrict__ dict, const int32_t ndict,
    const int2 *__restrict__ pinfo, const int32_t *__restrict__ seg_out, const double *__restrict__ x_in,
    double *__restrict__ partial, const int32_t npieces, const SgState *__restrict__ st)
{
    extern __shared__ double tbl[];
    const int lane = threadIdx.x & 63;
    const int p0 = __builtin_amdgcn_readfirstlane((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * PPW);
    int c[PPW][4];
    v4h wi[PPW];
    int2 info[PPW];
#pragma unroll
    for (int u = 0; u < PPW; ++u) {  // (npieces >= 1: the launch has no blocks otherwise)
        const int p = min(p0 + u, npieces - 1);  // clamped duplicates are computed and dropped
        if constexpr (COL16) {
            const v4h cc = __builtin_nontemporal_load(&reinterpret_cast<const v4h *>(colv)[(int64_t)p * 64 + lane]);
            c[u][0] = cc.x; c[u][1] = cc.y; c[u][2] = cc.z; c[u][3] = cc.w;
        } else {
            const v4i cc = __builtin_nontemporal_load(&reinterpret_cast<const v4i *>(colv)[(int64_t)p * 64 + lane]);
            c[u][0] = cc.x; c[u][1] = cc.y; c[u][2] = cc.z; c[u][3] = cc.w;
        }
        wi[u] = __builtin_nontemporal_load(&widx[(int64_t)p * 64 + lane]);
        info[u] = pinfo[p];
    }
    const int done = st->done;
    for (int i = threadIdx.x; i < ndict; i += blockDim.x) tbl[i] = dict[i];
    double xs[PPW][4];
    int tgt[PPW];
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xs[u][j] = x_in[c[u][j]];
        tgt[u] = seg_out[info[u].x + (lane >> info[u].y)];
    }
    __syncthreads();
    if (done || p0 >= npieces) return;
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
        if (p0 + u >= npieces) break;
        double s = xs[u][0] * tbl[wi[u].x];   // col("probability") * col("balanced_weight") (:112)
        s = s + xs[u][1] * tbl[wi[u].y];
        s = s + xs[u][2] * tbl[wi[u].z];
        s = s + xs[u][3] * tbl[wi[u].w];
        const int cls = __builtin_amdgcn_readfirstlane(info[u].y);
        s = segment_butterfly_sum(s, cls);
        if ((lane & ((1 << cls) - 1)) == 0 && tgt[u] >= 0) partial[tgt[u]] = s;
    }
}

// Grid-stride form of the sweep: a fixed number of waves (a few blocks per CU) each walk pieces
// p, p + nwaves, ...; the column/weight loads of the NEXT piece are issued before the current one
// is gathered and reduced.  Same arithmetic, same slots, same order as sg_sweep.
template <bool COL16>
__global__ __launch_bounds__(256) void sg_sweep_gs(
    const void *__restrict__ colv, const v2d *__restrict__ w2, const int2 *__restrict__ pinfo,
    const int32_t *__restrict__ seg_out, const double *__restrict__ x_in, double *__restrict__ partial,
    int32_t npieces, const SgState *__restrict__ st)
{
    const int lane = threadIdx.x & 63;
    const int nwaves = gridDim.x * 4;
    int p = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (p >= npieces) return;
    const int done = st->done;
    int c0, c1, c2, c3;
    v2d wa, wb;
    int2 info;
    auto load_piece = [&](int q, int &d0, int &d1, int &d2, int &d3, v2d &ua, v2d &ub, int2 &inf) {
        if constexpr (COL16) {
            const v4h cc = __builtin_nontemporal_load(&reinterpret_cast<const v4h *>(colv)[(int64_t)q * 64 + lane]);
            d0 = cc.x; d1 = cc.y; d2 = cc.z; d3 = cc.w;
        } else {
            const v4i cc = __builtin_nontemporal_load(&reinterpret_cast<const v4i *>(colv)[(int64_t)q * 64 + lane]);
            d0 = cc.x; d1 = cc.y; d2 = cc.z; d3 = cc.w;
        }
        ua = __builtin_nontemporal_load(&w2[(int64_t)q * 128 + lane]);
        ub = __builtin_nontemporal_load(&w2[(int64_t)q * 128 + 64 + lane]);
        inf = pinfo[q];
    };
    load_piece(p, c0, c1, c2, c3, wa, wb, info);
    if (done) return;
    for (; p < npieces; p += nwaves) {
        const int q = p + nwaves;
        int n0 = 0, n1 = 0, n2 = 0, n3 = 0;
        v2d na = {0.0, 0.0}, nb = {0.0, 0.0};
        int2 ninfo = make_int2(0, 0);
        if (q < npieces) load_piece(q, n0, n1, n2, n3, na, nb, ninfo);
        const double x0 = x_in[c0], x1 = x_in[c1], x2 = x_in[c2], x3 = x_in[c3];
        const int cls = __builtin_amdgcn_readfirstlane(info.y);
        const int tgt = seg_out[info.x + (lane >> cls)];
        double s = x0 * wa.x;   // col("probability") * col("balanced_weight") (:112)
        s = s + x1 * wa.y;
        s = s + x2 * wb.x;
        s = s + x3 * wb.y;
        s = segment_butterfly_sum(s, cls);
        if ((lane & ((1 << cls) - 1)) == 0 && tgt >= 0) partial[tgt] = s;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        wa = na; wb = nb;
        info = ninfo;
    }
}

// patch kernels see the column array in its stored width
__device__ __forceinline__ double sg_next_x(double sigma, bool is_target, double alpha, double oma)
{
    const double u = is_target ? 1.0 : 0.0;
    const double a = u * alpha;     // col("u_probability") * alpha   (:119)
    const double b = sigma * oma;   // col("sigma") * (1 - alpha)     (:120)
    return a + b;
}

// calcNextX, the combine part (:115-126), fused with isConverged's sum (:130-141).
// x has T live entries, then D (all source-only vertices: n_plain_dead of them share it) and Q (the
// request's vertex when it is source-only).  Partials sit in row-major slots: live rows
// [0, n_short) own slots 3l .. 3l+2 (<= 2 full pieces + the remainder; unused slots stay 0.0), so a
// thread needs no index load before its three partial loads; the few longer rows are listed in lrows.
__device__ __forceinline__ void sg_finalize_body(
    const int bx /* this block's index among the kParts blocks of its graph */,
    int32_t n_short, const int4 *__restrict__ lrows, int32_t nlrows, int32_t n_crows, int32_t nlive,
    const double *__restrict__ partial, const double *__restrict__ x_in, double *__restrict__ x_out,
    int32_t target_x /* index into x of the request's vertex */, int32_t n_plain_dead, int32_t q_in_use,
    double alpha, double oma, const double *__restrict__ parts_prev, double *__restrict__ parts_out,
    SgState *st, double eps2, int32_t first)
{
    // this thread's loads go out before the convergence decision is known (they are independent of it)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l = bx * 256 + threadIdx.x;  // first row of this thread; further ones below
    const bool mine = l < n_short;
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, xv = 0.0;
    if (mine) {
        p0 = partial[3 * l + 0];
        p1 = partial[3 * l + 1];
        p2 = partial[3 * l + 2];
        xv = x_in[l];
    }
    // longer rows: lrows[0, n_crows) have more than kLongRow full pieces (one WAVE each),
    // lrows[n_crows, nlrows) have 3..kLongRow (one THREAD each, spread over the blocks so that their
    // loads are in flight together instead of one row after another on lane 0)
    const int n_brows = nlrows - n_crows;
    const int ci = bx * 4 + wave;
    const int bi = bx + kParts * (int)threadIdx.x;
    int4 rc = make_int4(0, 0, 0, 0), rb = make_int4(0, 0, 0, 0);
    if (ci < n_crows) rc = lrows[ci];
    if (bi < n_brows) rb = lrows[n_crows + bi];
    if (!first) {
        // isConverged of the PREVIOUS sweep (:99): every wave takes the same decision from the
        // same block sums in the same order; once true it sticks and x is never touched again
        if (st->done != 0 || device_total_d2(parts_prev) <= eps2) {
            if (bx == 0 && threadIdx.x == 0) st->done = 1;
            return;
        }
    }
    __shared__ double wsum[4];
    double d2 = 0.0;
    // one thread per medium row: all of its (at most kLongRow + 1) partials in flight at once,
    // added in slot order
    for (int b = bi; b < n_brows; b += kParts * 256) {
        const int4 r = b == bi ? rb : lrows[n_crows + b];
        double pv[kLongRow + 1];
#pragma unroll
        for (int j = 0; j < kLongRow; ++j) pv[j] = j < r.z ? partial[r.y + j] : 0.0;
        pv[kLongRow] = r.w ? partial[r.y + r.z] : 0.0;
        const double xo = x_in[r.x];
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < kLongRow; ++j)
            if (j < r.z) s = s + pv[j];
        if (r.w) s = s + pv[kLongRow];
        const double nx = sg_next_x(s, r.x == target_x, alpha, oma);
        const double diff = nx - xo;
        x_out[r.x] = nx;
        d2 = d2 + diff * diff;
    }
    // one wave per long row: lanes stride the partials, kWaveRowLoads loads in flight per lane (cfg3's largest row has
    // 790 pieces: one round), added in the same (ascending) order as a plain loop, then a butterfly and the remainder
    for (int i = ci; i < n_crows; i += kParts * 4) {
        const int4 r = i == ci ? rc : lrows[i];
        const double xo = x_in[r.x];
        const double prem = r.w ? partial[r.y + r.z] : 0.0;
        double s = 0.0;
        for (int j0 = lane; j0 < r.z; j0 += 64 * kWaveRowLoads) {
            double pv[kWaveRowLoads];
#pragma unroll
            for (int b = 0; b < kWaveRowLoads; ++b) {
                const int j = j0 + 64 * b;
                pv[b] = j < r.z ? partial[r.y + j] : 0.0;
            }
#pragma unroll
            for (int b = 0; b < kWaveRowLoads; ++b)
                if (j0 + 64 * b < r.z) s = s + pv[b];
        }
        s = wave_butterfly_sum(s);
        if (r.w) s = s + prem;
        if (lane == 0) {
            const double nx = sg_next_x(s, r.x == target_x, alpha, oma);
            const double diff = nx - xo;
            x_out[r.x] = nx;
            d2 = d2 + diff * diff;
        }
    }
    if (mine) {
        double s = 0.0;
        s = s + p0;
        s = s + p1;
        s = s + p2;
        const double nx = sg_next_x(s, l == target_x, alpha, oma);
        const double diff = nx - xv;
        x_out[l] = nx;
        d2 = d2 + diff * diff;
    }
    for (int l2 = l + kParts * 256; l2 < n_short; l2 += kParts * 256) {  // graphs with more than 16384 short rows
        double s = 0.0;
        s = s + partial[3 * l2 + 0];
        s = s + partial[3 * l2 + 1];
        s = s + partial[3 * l2 + 2];
        const double nx = sg_next_x(s, l2 == target_x, alpha, oma);
        const double diff = nx - x_in[l2];
        x_out[l2] = nx;
        d2 = d2 + diff * diff;
    }
    if (bx == 0 && threadIdx.x == 0) {
        // the shared slots: sigma = 0 for a vertex nobody points at
        const double xd = sg_next_x(0.0, false, alpha, oma);
        const double dd = xd - x_in[nlive];
        x_out[nlive] = xd;
        d2 = d2 + (double)n_plain_dead * (dd * dd);
        const double xq = sg_next_x(0.0, q_in_use != 0, alpha, oma);
        const double dq = xq - x_in[nlive + 1];
        x_out[nlive + 1] = xq;
        if (q_in_use) d2 = d2 + dq * dq;
    }
    d2 = wave_butterfly_sum(d2);
    if (lane == 0) wsum[wave] = d2;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = wsum[0];
        t = t + wsum[1];
        t = t + wsum[2];
        t = t + wsum[3];
        parts_out[bx] = t;
        if (bx == 0) st->sweeps = st->sweeps + 1;  // nobody reads it inside this launch
    }
}

// What changes from request to request lives in device memory, not in the kernel arguments: the launches of a
// run of iterations are then the same for every request and can be replayed as ONE hipGraph (enqueue_iterations).
struct SgReq {
    int32_t target_x, n_plain_dead, q_in_use, pad;
    double alpha, oma, eps2;
};

// The set-up of one request in ONE launch (round 2; it was x0 fill + two patch launches + a host-to-device copy of
// the slot list): x0 (:51-54), state reset, the previous request's out-edge slots back to D and this request's to Q
// (the slot lists of every source-only vertex stay resident: dead_slots), the request's values into SgReq.
struct SgBegin {
    double *x;            // first x buffer, nx entries
    double *parts;        // 2 * kParts block sums
    SgState *st;
    void *col;            // uint16 or int32 column indices
    const int32_t *slots; // slot lists of all source-only vertices
    SgReq *req_dst;       // may be NULL (groups and shards carry the values elsewhere)
    double x0;
    int32_t nx, col16, old_off, n_old, new_off, n_new, T, pad;
    SgReq req;
};

__device__ __forceinline__ void sg_begin_body(const SgBegin &b, const int bx, const int nblocks)
{
    const int i = bx * 256 + (int)threadIdx.x;
    const int stride = nblocks * 256;
    for (int j = i; j < b.nx; j += stride) b.x[j] = b.x0;
    // (the two slot ranges belong to different vertices, so they are disjoint; the host passes none when the vertex
    // is the previous request's)
    for (int j = i; j < b.n_old; j += stride) {
        const int32_t slot = b.slots[b.old_off + j];
        if (b.col16) static_cast<unsigned short *>(b.col)[slot] = (unsigned short)b.T;
        else static_cast<int32_t *>(b.col)[slot] = b.T;
    }
    for (int j = i; j < b.n_new; j += stride) {
        const int32_t slot = b.slots[b.new_off + j];
        if (b.col16) static_cast<unsigned short *>(b.col)[slot] = (unsigned short)(b.T + 1);
        else static_cast<int32_t *>(b.col)[slot] = b.T + 1;
    }
    if (bx == 0) {
        if (threadIdx.x == 0) {
            b.st->done = 0;
            b.st->sweeps = 0;
            if (b.req_dst) *b.req_dst = b.req;
        }
        for (int j = threadIdx.x; j < 2 * kParts; j += 256) b.parts[j] = 0.0;
    }
}

__global__ __launch_bounds__(256) void sg_begin(SgBegin b)  // (by value: in stream order, no host buffer to keep)
{
    sg_begin_body(b, (int)blockIdx.x, (int)gridDim.x);
}

__global__ __launch_bounds__(256) void sg_begin_group(const SgBegin *__restrict__ tab)
{
    sg_begin_body(tab[blockIdx.y], (int)blockIdx.x, (int)gridDim.x);
}

// The host's look at the convergence word, and the read-back of a result, without copy packets on the stream (each
// device-to-host copy of a few bytes is a command of its own, ~5-8 us): the kernels below write pinned host memory
// directly, and the host reads it after the stream synchronisation it needs anyway.
__global__ void sg_poll(const SgState *__restrict__ st, int32_t *host_word)
{
    if (threadIdx.x == 0) *host_word = st->done;
}

// state (16 B) at 0, the 2 * kParts block sums at 64, then x of the CURRENT parity (the one the last executed sweep
// wrote: sweeps & 1) - nx doubles instead of both buffers
__global__ __launch_bounds__(256) void sg_pack_result(const SgState *__restrict__ st, const double *__restrict__ parts,
                                                      const double *__restrict__ xbuf, int32_t nx, unsigned char *host)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int stride = gridDim.x * 256;
    const SgState s = *st;
    if (t == 0) *reinterpret_cast<SgState *>(host) = s;
    double *hp = reinterpret_cast<double *>(host + 64);
    for (int i = t; i < 2 * kParts; i += stride) hp[i] = parts[i];
    double *hx = hp + 2 * kParts;
    const double *x = xbuf + (size_t)(s.sweeps & 1) * nx;
    for (int i = t; i < nx; i += stride) hx[i] = x[i];
}

__global__ __launch_bounds__(256) void sg_finalize(
    int32_t n_short, const int4 *__restrict__ lrows, int32_t nlrows, int32_t n_crows, int32_t nlive,
    const double *__restrict__ partial, const double *__restrict__ x_in, double *__restrict__ x_out,
    const SgReq *__restrict__ rq, const double *__restrict__ parts_prev, double *__restrict__ parts_out, SgState *st,
    int32_t first)
{
    sg_finalize_body((int)blockIdx.x, n_short, lrows, nlrows, n_crows, nlive, partial, x_in, x_out, rq->target_x,
                     rq->n_plain_dead, rq->q_in_use, rq->alpha, rq->oma, parts_prev, parts_out, st, rq->eps2, first);
}

// ---------------------------------------------------------------------------------------------------
// Fused iteration (round 3): no sg_finalize on the critical path.  sg_finalize moves 0.3 MB in 5 us -
// a launch plus two dependent round trips, a third of every 15 us iteration - and no kernel ever needs the whole of x:
//   * a class-A row (<= 2 full pieces: 97 % of the rows at cfg3) has at most four partials; x'[c] of such a SOURCE is
//     recomputed on the fly by the sweep that needs it: s = 0 + p[3c] + p[3c+1] + p[3c+2] + p[X c] (sg_finalize's order,
//     then the contribution of the row's long-source edges, below), x' = u alpha + s (1 - alpha);
//   * a source-only vertex (95 % of the edges at cfg3) needs no load at all: x' = alpha u;
//   * the LONGER rows (508 of 10,020 at cfg3, but they own most pieces) do need a reduction over many partials - the
//     part of sg_finalize that stays, as sg_fused_long - yet only 14 k of the 4.8 M edges have such a row as their
//     SOURCE.  Those edges are given a second, tiny piece list of their own (119 pieces at cfg3; in the main pieces
//     their lanes multiply by x = 0), swept by sg_fused_k2 into one extra partial slot per target row (X).
// One iteration = [ sg_sweep_fused(i): all main pieces ]  in parallel with  [ sg_fused_long(i-1) -> sg_fused_k2(i) ] on
// a second stream, joined by events (a replayed hipGraph carries the fork and the join): the long rows' reduction
// and the tiny second sweep hide behind the main sweep.  Partials live in FOUR generations (i mod 4).
// isConverged's sum of iteration i-1 (:130-141) rides along: in sweep i the first ceil(n_short / 64) waves ("duty")
// each take 64 class-A rows and form x'_{i-1} and x'_{i-2} from the two previous generations; sg_fused_long adds the
// longer rows' share (it has both of their x' at hand).  Wave 0 of sweep i totals both shares of iteration i-2 in a
// fixed order and, when <= eps^2, sets the sticky `done` word and records the iteration: sweeps i-1 and i have run
// by then - harmlessly, four generations keep the converged iteration's partials intact - and sweep i + 1 exits.
// A request ends with sg_fused_long and the duty pass for the last iteration (sg_fused_tail) and sg_fused_result,
// which decides step()'s exit (:92-106) on the device and leaves x, the sweep count and isConverged's sum exactly
// where locrec_sg_fetch expects them.
// All sums are fp64 in a fixed order; against the two-launch form only the position of a row's long-source products
// inside its sum changes (they are added last): probabilities agree to ~1e-15 relative, not bit for bit.
// (A first form with ONE launch - the longer rows' partials added as 62-bit fixed-point integers with fire-and-forget
// atomics into 16 stripe accumulators, summed on the fly by every reader - was correct and SLOWER: 20.5 us per
// iteration against 15.2; knocked out, the atomics cost 5.3 us (50 same-address atomics per stripe of a category row) and
// the divergent stripe reads 3.9 us; without both the sweep ran 11.3 us = 12.2 us per iteration, which is what this
// form goes for.  profiles/r03_sg_fused_stripes.log, r03_sg_fused_knockouts.log.)
struct SgFused {
    const void *colv;
    const v2d *w2;
    const int2 *pinfo;
    const int32_t *seg_fa;       // main pieces: segment -> partial slot (4 c + j for a class-A row c)
    // the second piece list: the edges whose source is a longer row
    const int32_t *col2;         // [npieces2][64][4] live index of the source
    const v2d *w2b;
    const int2 *pinfo2;
    const int32_t *seg2;         // segment -> the X slot of its target row
    const int4 *lrows;           // the longer rows (as sg_finalize reads them)
    double *PA4;                 // [4][pa4]: class-A row c owns slots 4 c .. 4 c + 3 (two full pieces, the remainder, X);
                                 // a longer row a run behind them (its full pieces, the remainder, X)
    double *XL;                  // [4][n_long] x' of the longer rows
    double *D2W;                 // [4][nduty + kParts] diff^2 sums: duty waves, then sg_fused_long's blocks
    const SgReq *rq;
    SgState *st;
    int32_t *conv;               // [0] the iteration isConverged first held for (valid when st->done); [1] the iteration
                                 // number of the current run's first sweep (sg_fused_advance): the launches of a replayed
                                 // run carry only their position inside the run
    int32_t npieces, npieces2, n_short, T, nduty, pa4, nlrows, n_crows;
    double x0;
};

// sigma of a class-A row from generation `gen` (sg_finalize's order, then the long-source share)
__device__ __forceinline__ double fused_sigma_a(const SgFused &F, int gen, int c)
{
    const v2d *p = reinterpret_cast<const v2d *>(F.PA4 + (size_t)gen * F.pa4 + 4 * (size_t)c);  // one 32-byte line
    const v2d a = p[0], b = p[1];
    double s = 0.0;
    s = s + a.x;
    s = s + a.y;
    s = s + b.x;
    s = s + b.y;
    return s;
}

// fixed-order total of one generation's diff^2 sums (duty waves, then sg_fused_long's blocks); the same bits in every lane
__device__ __forceinline__ double fused_total_d2(const SgFused &F, int it, int lane)
{
    const double *p = F.D2W + (size_t)(it & 3) * (F.nduty + kParts);
    double s = 0.0;
    for (int j = lane; j < F.nduty + kParts; j += 64) s = s + p[j];
    return wave_butterfly_sum(s);
}

// the duty pass for iteration `it` (run inside sweep it + 1, or by sg_fused_tail): wave w takes class-A rows 64 w ..
__device__ __forceinline__ void fused_duty(const SgFused &F, int it, int w, int lane, const SgReq &rq)
{
    const int r = w * 64 + lane;
    double d2 = 0.0;
    if (r < F.n_short) {
        const double xa = sg_next_x(fused_sigma_a(F, it & 3, r), r == rq.target_x, rq.alpha, rq.oma);
        const double xb = it == 0 ? F.x0 : sg_next_x(fused_sigma_a(F, (it - 1) & 3, r), r == rq.target_x, rq.alpha, rq.oma);
        const double diff = xa - xb;
        d2 = d2 + diff * diff;
    }
    if (w == 0 && lane == 0) {  // the shared slots: every source-only vertex, and the request's own when it is one
        const double xd = sg_next_x(0.0, false, rq.alpha, rq.oma);
        const double dd = xd - (it == 0 ? F.x0 : xd);
        d2 = d2 + (double)rq.n_plain_dead * (dd * dd);
        if (rq.q_in_use) {
            const double xq = sg_next_x(0.0, true, rq.alpha, rq.oma);
            const double dq = xq - (it == 0 ? F.x0 : xq);
            d2 = d2 + dq * dq;
        }
    }
    d2 = wave_butterfly_sum(d2);
    if (lane == 0) F.D2W[(size_t)(it & 3) * (F.nduty + kParts) + w] = d2;
}

// end of a run of `len` sweeps: the next run's launches count from here
__global__ void sg_fused_advance(int32_t *conv, int32_t len)
{
    if (threadIdx.x == 0) conv[1] += len;
}

// the main sweep of iteration it = conv[1] + j
template <bool COL16>
__global__ __launch_bounds__(256) void sg_sweep_fused(const SgFused F, const int32_t j /* position inside its run */)
{
    const int it = F.conv[1] + j;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const bool has_piece = w < F.npieces, has_duty = it >= 1 && w < F.nduty;
    if (!has_piece && !has_duty) return;
    int c[4] = {0, 0, 0, 0};
    v2d wa = {0.0, 0.0}, wb = {0.0, 0.0};
    int2 info = make_int2(0, 0);
    if (has_piece) {  // the stream first: it depends on nothing
        if constexpr (COL16) {
            const v4h cc = __builtin_nontemporal_load(&reinterpret_cast<const v4h *>(F.colv)[(int64_t)w * 64 + lane]);
            c[0] = cc.x; c[1] = cc.y; c[2] = cc.z; c[3] = cc.w;
        } else {
            const v4i cc = __builtin_nontemporal_load(&reinterpret_cast<const v4i *>(F.colv)[(int64_t)w * 64 + lane]);
            c[0] = cc.x; c[1] = cc.y; c[2] = cc.z; c[3] = cc.w;
        }
        wa = __builtin_nontemporal_load(&F.w2[(int64_t)w * 128 + lane]);
        wb = __builtin_nontemporal_load(&F.w2[(int64_t)w * 128 + 64 + lane]);
        info = F.pinfo[w];
    }
    const int done = F.st->done;
    const SgReq rq = *F.rq;
    if (w == 0 && it >= 2 && !done) {  // isConverged of iteration it - 2 (:99): one wave decides, the next launch obeys
        if (fused_total_d2(F, it - 2, lane) <= rq.eps2 && lane == 0) {
            F.conv[0] = it - 2;
            F.st->done = 1;
        }
    }
    if (done) return;
    if (has_piece) {
        const int prev = (it - 1) & 3;
        double xs[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cc = c[e];
            // a longer row as the source: this lane contributes nothing here - the edge sits in the second piece list too
            xs[e] = it == 0 ? (cc >= F.n_short && cc < F.T ? 0.0 : F.x0)
                   : cc < F.n_short ? sg_next_x(fused_sigma_a(F, prev, cc), cc == rq.target_x, rq.alpha, rq.oma)
                   : cc < F.T ? 0.0 : sg_next_x(0.0, cc == rq.target_x, rq.alpha, rq.oma);
        }
        double s = xs[0] * wa.x;   // col("probability") * col("balanced_weight") (:112)
        s = s + xs[1] * wa.y;
        s = s + xs[2] * wb.x;
        s = s + xs[3] * wb.y;
        const int cls = __builtin_amdgcn_readfirstlane(info.y);
        s = segment_butterfly_sum(s, cls);
        const int tgt = F.seg_fa[info.x + (lane >> cls)];
        if ((lane & ((1 << cls) - 1)) == 0 && tgt >= 0) F.PA4[(size_t)(it & 3) * F.pa4 + tgt] = s;
    }
    if (has_duty) fused_duty(F, it - 1, w, lane, rq);
}

// the second sweep of iteration it: the edges whose source is a longer row, into the X slot of their target row
__global__ __launch_bounds__(256) void sg_fused_k2(const SgFused F, const int32_t j)
{
    const int it = F.conv[1] + j;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (w >= F.npieces2 || F.st->done) return;
    const v4i cc = reinterpret_cast<const v4i *>(F.col2)[(int64_t)w * 64 + lane];
    const v2d wa = F.w2b[(int64_t)w * 128 + lane], wb = F.w2b[(int64_t)w * 128 + 64 + lane];
    const int2 info = F.pinfo2[w];
    const double *xl = F.XL + (size_t)((it - 1) & 3) * max(1, F.T - F.n_short);
    const int c[4] = {cc.x, cc.y, cc.z, cc.w};
    double xs[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) xs[e] = c[e] < 0 ? 0.0 : it == 0 ? F.x0 : xl[c[e] - F.n_short];  // (-1 = padding slot)
    double s = xs[0] * wa.x;
    s = s + xs[1] * wa.y;
    s = s + xs[2] * wb.x;
    s = s + xs[3] * wb.y;
    const int cls = __builtin_amdgcn_readfirstlane(info.y);
    s = segment_butterfly_sum(s, cls);
    const int tgt = F.seg2[info.x + (lane >> cls)];  // (the X slot itself)
    if ((lane & ((1 << cls) - 1)) == 0 && tgt >= 0) F.PA4[(size_t)(it & 3) * F.pa4 + tgt] = s;
}

// the longer rows of iteration `it`: sg_finalize's reduction (same order, then the X slot), x' into XL, their share of
// isConverged's sum into the block slots behind the duty slots.  it = conv[1] + j for launches inside a run, or j itself
// for the request's last one (`absolute`).
__global__ __launch_bounds__(256) void sg_fused_long(const SgFused F, const int32_t j, const int32_t absolute)
{
    const int it = absolute ? j : F.conv[1] + j;
    __shared__ double wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, bx = blockIdx.x;
    if (F.st->done) return;
    const SgReq rq = *F.rq;
    const int n_long = max(1, F.T - F.n_short);
    const double *partial = F.PA4 + (size_t)(it & 3) * F.pa4;
    double *xl = F.XL + (size_t)(it & 3) * n_long;
    const double *xl_prev = F.XL + (size_t)((it - 1) & 3) * n_long;
    const int n_brows = F.nlrows - F.n_crows;
    double d2 = 0.0;
    for (int b = bx + kParts * (int)threadIdx.x; b < n_brows; b += kParts * 256) {  // one thread per medium row
        const int4 r = F.lrows[F.n_crows + b];
        double pv[kLongRow + 1];
#pragma unroll
        for (int q = 0; q < kLongRow; ++q) pv[q] = q < r.z ? partial[r.y + q] : 0.0;
        pv[kLongRow] = r.w ? partial[r.y + r.z] : 0.0;
        const double px = partial[r.y + r.z + 1];
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < kLongRow; ++q)
            if (q < r.z) s = s + pv[q];
        if (r.w) s = s + pv[kLongRow];
        s = s + px;
        const double nx = sg_next_x(s, r.x == rq.target_x, rq.alpha, rq.oma);
        const double diff = nx - (it == 0 ? F.x0 : xl_prev[r.x - F.n_short]);
        xl[r.x - F.n_short] = nx;
        d2 = d2 + diff * diff;
    }
    for (int i = bx * 4 + wave; i < F.n_crows; i += kParts * 4) {  // one wave per long row
        const int4 r = F.lrows[i];
        const double prem = r.w ? partial[r.y + r.z] : 0.0;
        const double px = partial[r.y + r.z + 1];
        double s = 0.0;
        for (int j0 = lane; j0 < r.z; j0 += 64 * kWaveRowLoads) {
            double pv[kWaveRowLoads];
#pragma unroll
            for (int b = 0; b < kWaveRowLoads; ++b) {
                const int q = j0 + 64 * b;
                pv[b] = q < r.z ? partial[r.y + q] : 0.0;
            }
#pragma unroll
            for (int b = 0; b < kWaveRowLoads; ++b)
                if (j0 + 64 * b < r.z) s = s + pv[b];
        }
        s = wave_butterfly_sum(s);
        if (r.w) s = s + prem;
        s = s + px;
        if (lane == 0) {
            const double nx = sg_next_x(s, r.x == rq.target_x, rq.alpha, rq.oma);
            const double diff = nx - (it == 0 ? F.x0 : xl_prev[r.x - F.n_short]);
            xl[r.x - F.n_short] = nx;
            d2 = d2 + diff * diff;
        }
    }
    d2 = wave_butterfly_sum(d2);
    if (lane == 0) wsum[wave] = d2;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = wsum[0];
        t = t + wsum[1];
        t = t + wsum[2];
        t = t + wsum[3];
        F.D2W[(size_t)(it & 3) * (F.nduty + kParts) + F.nduty + bx] = t;
    }
}

// the duty pass of the LAST iteration (no sweep follows it)
__global__ __launch_bounds__(256) void sg_fused_tail(const SgFused F, const int32_t last_it)
{
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= F.nduty || F.st->done) return;
    const SgReq rq = *F.rq;
    fused_duty(F, last_it, w, lane, rq);
}

// step()'s exit (:92-106) decided on the device, and the request's result where locrec_sg_fetch reads it: x of the
// returned iteration k at parity (k + 1) & 1 of xbuf, st->sweeps = k + 1, isConverged's sum of iteration k in block-sum
// slot 0 of parity k & 1 (the other slots zero: the host's fixed-order total of the 64 slots is then that value)
__global__ __launch_bounds__(256) void sg_fused_result(const SgFused F, const int32_t max_it, double *xbuf, double *parts)
{
    const int lane = threadIdx.x & 63;
    const SgReq rq = *F.rq;
    int k = max_it - 1;
    if (F.st->done) {
        k = F.conv[0];
    } else if (max_it >= 2 && fused_total_d2(F, max_it - 2, lane) <= rq.eps2) {
        k = max_it - 2;  // (its sums were complete after the last sweep; no later sweep was there to act on them)
    }
    const double d2k = fused_total_d2(F, k, lane);
    const int nx = F.T + 2;
    double *xo = xbuf + (size_t)((k + 1) & 1) * nx;
    const double *xl = F.XL + (size_t)(k & 3) * max(1, F.T - F.n_short);
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nx; r += gridDim.x * blockDim.x)
        xo[r] = r < F.n_short ? sg_next_x(fused_sigma_a(F, k & 3, r), r == rq.target_x, rq.alpha, rq.oma)
                : r < F.T    ? xl[r - F.n_short]
                             : sg_next_x(0.0, r == rq.target_x, rq.alpha, rq.oma);
    if (blockIdx.x == 0) {
        double *po = parts + (size_t)(k & 1) * kParts;
        for (int q = threadIdx.x; q < kParts; q += blockDim.x) po[q] = q == 0 ? d2k : 0.0;
        if (threadIdx.x == 0) F.st->sweeps = k + 1;
    }
}

// ---- several graphs in one launch (locrec_sg_group_*: BASELINE.json configs[4], many independent graphs per
// GPU).  One cfg3-sized graph is ~19 k short waves: its sweep is mostly ramp and tail, and eight graphs on eight
// streams still pay sixteen launches per round.  The group kernels walk a table of per-graph views: the sweep's
// wave w belongs to the graph whose wave range holds it, the finalize's block (b, g) is block b of graph g -
// the same bodies, the same order of operations, bit-identical results, two launches per round for all graphs.
struct SgGraphView {
    const void *colv;
    const v2d *w2;
    const int2 *pinfo;
    const int32_t *seg_out;
    double *xbuf;      // 2 * nx
    double *partial;
    double *parts;     // 2 * kParts
    SgState *st;
    const int4 *lrows;
    int32_t npieces, nx, wave_base;
    int32_t n_short, nlrows, n_crows, nlive, target_x, n_plain_dead, q_in_use;
    double alpha, oma, eps2;
};

template <bool COL16>
__global__ __launch_bounds__(256) void sg_sweep_group(const SgGraphView *__restrict__ G, int32_t ngraphs, int32_t par)
{
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    // the last graph whose wave range starts at or before w: a scalar bisection (a walk over 64 graphs was 64
    // dependent scalar loads, ~3 us in front of a ~10 us sweep)
    int gi = 0, hi = ngraphs;
    while (hi - gi > 1) {
        const int mid = (gi + hi) >> 1;
        if (G[mid].wave_base <= w) gi = mid; else hi = mid;
    }
    const SgGraphView &v = G[gi];
    sg_sweep_body<COL16, 1>(v.colv, v.w2, v.pinfo, v.seg_out, v.xbuf + (size_t)par * v.nx, v.partial, v.npieces, v.st,
                            w - v.wave_base);
}

__global__ __launch_bounds__(256) void sg_finalize_group(const SgGraphView *__restrict__ G, int32_t par, int32_t first)
{
    const SgGraphView &v = G[blockIdx.y];
    sg_finalize_body((int)blockIdx.x, v.n_short, v.lrows, v.nlrows, v.n_crows, v.nlive, v.partial,
                     v.xbuf + (size_t)par * v.nx, v.xbuf + (size_t)(par ^ 1) * v.nx, v.target_x, v.n_plain_dead, v.q_in_use,
                     v.alpha, v.oma, v.parts + (size_t)(par ^ 1) * kParts, v.parts + (size_t)par * kParts, v.st, v.eps2, first);
}

// ---------------------------------------------------------------------------
// Row-sharded form (several GPUs on ONE graph, BASELINE.json configs[4]): every shard sweeps its own
// edges into sigma[T] (sg_sigma), the shards' sigmas are summed by an all-reduce the host enqueues on
// the same stream (RCCL over xGMI through torch.distributed -- only the T live entries of x ever need
// exchanging, 80 KB at cfg3), and every shard applies the same total (sg_apply_sigma).

__global__ __launch_bounds__(256) void sg_sigma(int32_t n_short, const int4 *__restrict__ lrows, int32_t nlrows,
                                                const double *__restrict__ partial, double *__restrict__ sigma)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int i = blockIdx.x * 4 + wave; i < nlrows; i += gridDim.x * 4) {
        const int4 r = lrows[i];
        double s = 0.0;
        if (r.z > kLongRow) {
            for (int j = lane; j < r.z; j += 64) s = s + partial[r.y + j];
            s = wave_butterfly_sum(s);
        } else {
            for (int j = 0; j < r.z; ++j) s = s + partial[r.y + j];
        }
        if (r.w) s = s + partial[r.y + r.z];
        if (lane == 0) sigma[r.x] = s;
    }
    for (int l = blockIdx.x * 256 + threadIdx.x; l < n_short; l += gridDim.x * 256) {
        double s = 0.0;
        s = s + partial[3 * l + 0];
        s = s + partial[3 * l + 1];
        s = s + partial[3 * l + 2];
        sigma[l] = s;
    }
}

__global__ __launch_bounds__(256) void sg_apply_sigma(
    int32_t nlive, const double *__restrict__ sigma, const double *__restrict__ x_in, double *__restrict__ x_out,
    int32_t target_x, int32_t n_plain_dead, int32_t q_in_use, double alpha, double oma,
    double *__restrict__ parts_out, SgState *st)
{
    __shared__ double wsum[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double d2 = 0.0;
    for (int l = blockIdx.x * 256 + threadIdx.x; l < nlive; l += kParts * 256) {
        const double nx = sg_next_x(sigma[l], l == target_x, alpha, oma);
        const double diff = nx - x_in[l];
        x_out[l] = nx;
        d2 = d2 + diff * diff;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double xd = sg_next_x(0.0, false, alpha, oma);
        const double dd = xd - x_in[nlive];
        x_out[nlive] = xd;
        d2 = d2 + (double)n_plain_dead * (dd * dd);
        const double xq = sg_next_x(0.0, q_in_use != 0, alpha, oma);
        const double dq = xq - x_in[nlive + 1];
        x_out[nlive + 1] = xq;
        if (q_in_use) d2 = d2 + dq * dq;
    }
    d2 = wave_butterfly_sum(d2);
    if (lane == 0) wsum[wave] = d2;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = wsum[0];
        t = t + wsum[1];
        t = t + wsum[2];
        t = t + wsum[3];
        parts_out[blockIdx.x] = t;
        if (blockIdx.x == 0) st->sweeps = st->sweeps + 1;
    }
}

// ---------------------------------------------------------------------------
// Persistent form: the whole edge list lives in REGISTERS (the chip's vector register file is
// 128 MB; cfg3's matrix is ~50 MB) and the compact x table in LDS, for the whole request.
// One block of 8 waves per CU; each wave owns PW pieces for the lifetime of the launch.  A sweep:
//   1. sigma partials from registers + LDS gathers -> global (row-major slots, double-buffered)
//   2. ONE grid barrier (monotonic counter, agent-scope release / acquire, bounded wait)
//   3. every block rebuilds the full x table in its own LDS from the partials (L2 reads, the
//      same fixed order in every block) and so takes the same convergence decision: no second
//      barrier, no host round trip.
// Summation orders equal the streaming kernels' (full pieces in order, then the remainder), so
// both forms return the same bits.
// STATUS: correct (tests/test_gpu_sg.py::test_persistent_form_matches) but OPT-IN
// (LOCREC_SG_PERSIST=1): measured on cfg3 per sweep, compute 5.3 us + barrier 13 us + rebuild 59 us
// = 77 us against 23.7 us for the streaming pair.  The rebuild is latency-bound on the ~500 rows
// with more than two full pieces that every block re-sums one row per wave at a time, and the
// single-counter barrier costs as much as eight kernel boundaries.  What it needs to win: an
// XCD-hierarchical barrier (~4 us), per-wave run-merged partials so that long rows become
// thread-per-row batched loads, DPP instead of ds_bpermute butterflies.  If a block does not see the others within the time limit
// (not all blocks resident) every block leaves and the host falls back to the streaming form.

struct PersistParams {
    const void *col;
    const v2d *w2;
    const int2 *pinfo;
    const int32_t *lane_out;   // [npieces * 64] target partial slot of a segment's leader lane, else -1
    int32_t n_short;           // live rows [0, n_short) have <= 2 full pieces: row-major slots l*3 + j
    const int4 *lrows;         // rows with more: x = row, y = first slot, z = nfull, w = has remainder
    int32_t nlrows;
    double *PA;                // 2 * pa_stride partial slots
    int32_t pa_stride;
    int32_t npieces, T;
    int32_t target_x, n_plain_dead, q_in_use;
    double alpha, oma, eps2, x0;
    int32_t max_it;
    unsigned *barrier;
    SgState *st;
    double *parts;
    double *xbuf;
    int32_t nblocks;
    unsigned long long *dbg;   // optional: block 0 accumulates 100 MHz ticks per phase (compute, barrier, rebuild)
};

// Guideline-16 style grid barrier on one monotonic counter.  Returns false on time-out.
__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned target, int *s_ok)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's partial stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
        int ok = 1;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ull) {  // 0.2 s
                ok = 0;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *s_ok = ok;
    }
    __syncthreads();
    return *s_ok != 0;
}

template <int PW, bool COL16>
__global__ __launch_bounds__(512) void sg_persistent(const PersistParams P)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *xt = reinterpret_cast<double *>(smem);  // [T + 2]
    __shared__ double wsum[8];
    __shared__ int s_ok;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int T = P.T;
    const int gw = blockIdx.x * 8 + wave;

    // resident pieces: column byte offsets into xt, weights, output slot, segment class
    int cb[PW][4];
    v2d wa[PW], wb[PW];
    int out[PW], cls[PW];
#pragma unroll
    for (int u = 0; u < PW; ++u) {
        const int p = gw * PW + u;
        if (p < P.npieces) {
            if constexpr (COL16) {
                const v4h cc = reinterpret_cast<const v4h *>(P.col)[(int64_t)p * 64 + lane];
                cb[u][0] = cc.x * 8; cb[u][1] = cc.y * 8; cb[u][2] = cc.z * 8; cb[u][3] = cc.w * 8;
            } else {
                const v4i cc = reinterpret_cast<const v4i *>(P.col)[(int64_t)p * 64 + lane];
                cb[u][0] = cc.x * 8; cb[u][1] = cc.y * 8; cb[u][2] = cc.z * 8; cb[u][3] = cc.w * 8;
            }
            wa[u] = P.w2[(int64_t)p * 128 + lane];
            wb[u] = P.w2[(int64_t)p * 128 + 64 + lane];
            out[u] = P.lane_out[(int64_t)p * 64 + lane];
            cls[u] = P.pinfo[p].y;
        } else {
            cb[u][0] = cb[u][1] = cb[u][2] = cb[u][3] = T * 8;
            wa[u] = v2d{0.0, 0.0};
            wb[u] = v2d{0.0, 0.0};
            out[u] = -1;
            cls[u] = 0;
        }
    }
    for (int i = tid; i < T + 2; i += 512) xt[i] = P.x0;
    __syncthreads();

    int sweeps = 0, converged = 0, ok = 1;
    double d2 = 0.0;
    unsigned long long tk[3] = {0, 0, 0};
    for (int it = 0; it < P.max_it; ++it) {
        double *pa = P.PA + (size_t)(it & 1) * P.pa_stride;
        const unsigned long long t_a = P.dbg ? __builtin_amdgcn_s_memrealtime() : 0;
        // 1. sigma partials (StochasticRecommender.scala:109-114)
#pragma unroll
        for (int u = 0; u < PW; ++u) {
            const double x0 = *reinterpret_cast<const double *>(smem + cb[u][0]);
            const double x1 = *reinterpret_cast<const double *>(smem + cb[u][1]);
            const double x2 = *reinterpret_cast<const double *>(smem + cb[u][2]);
            const double x3 = *reinterpret_cast<const double *>(smem + cb[u][3]);
            double s = x0 * wa[u].x;
            s = s + x1 * wa[u].y;
            s = s + x2 * wb[u].x;
            s = s + x3 * wb[u].y;
            const int c = __builtin_amdgcn_readfirstlane(cls[u]);
            for (int d = 1; d < (1 << c); d <<= 1) s = s + __shfl_xor(s, d);
            if (out[u] >= 0) pa[out[u]] = s;
        }
        // 2. every block's partials are in memory
        const unsigned long long t_b = P.dbg ? __builtin_amdgcn_s_memrealtime() : 0;
        if (!grid_barrier(P.barrier, (unsigned)(it + 1) * (unsigned)P.nblocks, &s_ok)) {
            ok = 0;
            break;
        }
        const unsigned long long t_c = P.dbg ? __builtin_amdgcn_s_memrealtime() : 0;
        // 3. x' for every live row (:115-126) and the convergence sum (:130-141), in this block's LDS
        double my = 0.0;
        for (int i = wave; i < P.nlrows; i += 8) {  // rows with more than two full pieces: one wave each
            const int4 r = P.lrows[i];
            double s = 0.0;
            if (r.z > kLongRow) {
                for (int j = lane; j < r.z; j += 64) s = s + pa[r.y + j];
                s = wave_butterfly_sum(s);
            } else {
                for (int j = 0; j < r.z; ++j) s = s + pa[r.y + j];
            }
            if (r.w) s = s + pa[r.y + r.z];
            if (lane == 0) {
                const double nx = sg_next_x(s, r.x == P.target_x, P.alpha, P.oma);
                const double diff = nx - xt[r.x];
                xt[r.x] = nx;
                my = my + diff * diff;
            }
        }
        // rows with <= 2 full pieces: four rows per thread in flight (the loads are independent L2
        // reads; issued one row at a time their latency would dominate the sweep)
        for (int l0 = tid; l0 < P.n_short; l0 += 4 * 512) {
            double pv[4][3];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int l = l0 + b * 512;
                const bool in = l < P.n_short;
                pv[b][0] = in ? pa[3 * l + 0] : 0.0;
                pv[b][1] = in ? pa[3 * l + 1] : 0.0;
                pv[b][2] = in ? pa[3 * l + 2] : 0.0;
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int l = l0 + b * 512;
                if (l < P.n_short) {
                    double s = 0.0;
                    s = s + pv[b][0];
                    s = s + pv[b][1];
                    s = s + pv[b][2];
                    const double nx = sg_next_x(s, l == P.target_x, P.alpha, P.oma);
                    const double diff = nx - xt[l];
                    xt[l] = nx;
                    my = my + diff * diff;
                }
            }
        }
        if (tid == 0) {  // the shared slots: sigma = 0 for a vertex nobody points at
            const double xd = sg_next_x(0.0, false, P.alpha, P.oma);
            const double dd = xd - xt[T];
            xt[T] = xd;
            my = my + (double)P.n_plain_dead * (dd * dd);
            const double xq = sg_next_x(0.0, P.q_in_use != 0, P.alpha, P.oma);
            const double dq = xq - xt[T + 1];
            xt[T + 1] = xq;
            if (P.q_in_use) my = my + dq * dq;
        }
        my = wave_butterfly_sum(my);
        if (lane == 0) wsum[wave] = my;
        __syncthreads();
        d2 = wsum[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) d2 = d2 + wsum[i];
        sweeps = it + 1;
        __syncthreads();  // xt complete, wsum free
        if (P.dbg) {
            const unsigned long long t_d = __builtin_amdgcn_s_memrealtime();
            tk[0] += t_b - t_a;
            tk[1] += t_c - t_b;
            tk[2] += t_d - t_c;
        }
        if (d2 <= P.eps2) {  // isConverged (:99): the same bits in every block
            converged = 1;
            break;
        }
    }
    if (blockIdx.x == 0) {
        double *xo = P.xbuf + (size_t)(sweeps & 1) * (T + 2);
        for (int i = tid; i < T + 2; i += 512) xo[i] = xt[i];
        for (int i = tid; i < 2 * kParts; i += 512) P.parts[i] = 0.0;
        __syncthreads();
        if (tid == 0) {
            P.st->sweeps = sweeps;
            P.st->done = ok ? converged : -1;
            if (sweeps > 0) P.parts[(size_t)((sweeps - 1) & 1) * kParts] = d2;
            if (P.dbg) {
                P.dbg[0] = tk[0];
                P.dbg[1] = tk[1];
                P.dbg[2] = tk[2];
            }
        }
    }
}

int ceil_log2(int v)
{
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace

struct locrec_sg_graph {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // pinned host staging (lazy): the convergence poll and the read-back of a result copy into it
    // asynchronously, so a request synchronises once per poll / once per fetch
    unsigned char *h_stage = nullptr;
    size_t h_stage_bytes = 0;
    ~locrec_sg_graph()
    {
        if (h_stage) (void)hipHostFree(h_stage);
        if (h_poll) (void)hipHostFree(h_poll);
        for (auto &kv : round_graphs) (void)hipGraphExecDestroy(kv.second);
        for (hipEvent_t e : fused_events) (void)hipEventDestroy(e);
        if (stream2) (void)hipStreamDestroy(stream2);
        // also reached by every early `return fail(...)` of sg_create_impl (unique_ptr)
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
    unsigned char *stage(size_t bytes)
    {
        if (h_stage_bytes < bytes) {
            if (h_stage) (void)hipHostFree(h_stage);
            h_stage = nullptr;
            h_stage_bytes = 0;
            void *p = nullptr;
            if (hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess) {
                h_stage = static_cast<unsigned char *>(p);
                h_stage_bytes = bytes;
            } else {
                (void)hipGetLastError();
            }
        }
        return h_stage;
    }
    // runs of iterations replayed as hipGraphs: key = ((rounds * 4 + (the run starts the request) + 2 * (it ends with the
    // poll kernel)) * 2 + (fused form)), see enqueue_iterations
    std::map<int64_t, hipGraphExec_t> round_graphs;
    bool no_graph = false;         // LOCREC_SG_NO_GRAPH
    int32_t *h_poll = nullptr;     // pinned: the convergence word as sg_poll last wrote it (64 B)
    int32_t *h_poll_dev = nullptr; // ... as the device addresses it
    bool no_pack = false;          // LOCREC_SG_NO_PACK: polls and read-back through device-to-host copies
    DevBuf<SgReq> req_dev;
    int64_t ne = 0;
    int64_t nv = 0;
    int32_t nlive = 0;             // T: vertices with inbound edges
    std::vector<int64_t> vid;      // sorted distinct vertex ids
    std::vector<int32_t> live_of;  // vertex -> live index or -1
    std::vector<int32_t> live_vertex;  // live index -> vertex
    std::vector<int32_t> live_sorted;  // live vertices in ascending vertex order (lazy, locrec_sg_fetch)
    // out-edge slots of source-only vertices (CSR over all vertices, empty ranges for live ones)
    std::vector<int64_t> dead_ptr;
    std::vector<int32_t> dead_slots;
    int32_t npieces = 0, nlong = 0;
    int64_t layout_bytes = 0;
    int64_t device_sweep_bytes = 0;  // locrec_sg_device_bytes
    DevBuf<int4> col4;            // int32 columns (COL16 off)
    DevBuf<unsigned short> col16;  // uint16 columns (COL16 on)
    bool use16 = false;
    int ppw = kPiecesPerWave;
    DevBuf<double2> w2;
    DevBuf<unsigned short> widx;   // dictionary form: uint16 weight index per edge slot (slot order, like col16)
    DevBuf<double> dict;
    int32_t ndict = 0;             // > 0: the sweep runs in its dictionary form
    int dict_threads = 1024;       // LOCREC_SG_DICT_THREADS
    int dict_ppw = 2;              // LOCREC_SG_DICT_PPW: pieces per wave of the dictionary form (1, 2 or 4)
    DevBuf<int2> pinfo;
    DevBuf<RowMeta> meta;
    DevBuf<int32_t> long_rows;
    DevBuf<double> partial;
    DevBuf<double> xbuf;    // 2 * (nlive + 2)
    DevBuf<double> parts;   // 2 * kParts
    DevBuf<SgState> state;
    DevBuf<int32_t> dead_slots_dev;    // dead_slots, resident: sg_begin patches straight from it
    int64_t patched_off = 0;           // the range of dead_slots currently pointing at Q
    int32_t n_patched = 0;
    int32_t shard_index = 0, shard_count = 1;
    // row-sharded iteration driven by the host (locrec_sg_shard_*)
    bool shard_active = false, shard_done = false;
    int64_t shard_it = 0, shard_iterations = 0;
    int32_t shard_converged = 0, shard_target_x = 0, shard_n_plain_dead = 0, shard_q_dead = 0;
    // persistent form
    bool persist_ok = false;
    int persist_pw = 0, persist_blocks = 0;
    size_t persist_lds = 0;
    int32_t pa_stride = 0, nlrows = 0, n_crows = 0;
    int gs_blocks = 0;  // > 0: grid-stride sweep with this many blocks (LOCREC_SG_GS = blocks per CU)
    DevBuf<int32_t> lane_out, seg_out;
    int32_t n_short = 0;
    DevBuf<int4> lrows;
    DevBuf<double> PA;
    DevBuf<unsigned> barrier;
    DevBuf<unsigned long long> dbg;  // LOCREC_SG_DEBUG_PHASES
    bool used_persistent = false;
    bool persist_failed = false;   // a barrier timed out once: streaming form from then on
    double req_alpha = 0;
    int64_t req_vertex = 0;
    int64_t persist_units = 0;
    // fused iteration (sg_sweep_fused and friends)
    bool fused_ok = false;          // the layout supports it (one handle holds all rows; <= 256 long-source edges per row)
    bool use_fused = false;         // LOCREC_SG_FUSED=1 (opt-in while it is being measured)
    bool fused_one_stream = false;  // LOCREC_SG_FUSED_ONE_STREAM: the three kernels of an iteration one after the other
    DevBuf<int32_t> seg_fa, col2, seg2, fused_conv;
    DevBuf<int2> pinfo2;
    DevBuf<double2> w2b;
    DevBuf<int4> lrows_f;
    DevBuf<double> PA4, XL, D2W;
    int32_t pa4 = 0, npieces2 = 0, nduty = 0;
    hipStream_t stream2 = nullptr;  // the longer rows' reduction and the second sweep run beside the main sweep
    std::vector<hipEvent_t> fused_events;
    // what sg_read_env found (the only place of this file that looks at the environment)
    bool env_no_dense_ids = false, env_no_dict = false, env_no_col16 = false, env_fused = false, env_persist = false;
    int env_gs = 0;                 // LOCREC_SG_GS: blocks per CU of the grid-stride sweep (0 = the one-shot sweep)
    KernelProfile prof;
    // last request
    bool have_result = false;
    int32_t target_vertex = -1;
    int64_t req_max_it = 0;
    double req_eps2 = 0;
};

using namespace locrec;

// Every environment switch of the SG path ("Environment switches"), read once when a handle is created.
static void sg_read_env(locrec_sg_graph *g)
{
    g->env_no_dense_ids = std::getenv("LOCREC_SG_NO_DENSE_IDS") != nullptr;  // rank the vertex ids by sorting
    g->env_no_dict = std::getenv("LOCREC_SG_NO_DICT") != nullptr;            // stream the fp64 weights
    g->env_no_col16 = std::getenv("LOCREC_SG_NO_COL16") != nullptr;          // int32 columns
    g->no_graph = std::getenv("LOCREC_SG_NO_GRAPH") != nullptr;              // no hipGraph replay of runs
    g->no_pack = std::getenv("LOCREC_SG_NO_PACK") != nullptr;                // polls / read-back through copies
    if (const char *e = std::getenv("LOCREC_SG_GS")) g->env_gs = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("LOCREC_SG_PPW")) {
        const int v = std::atoi(e);
        g->ppw = v == 1 || v == 2 || v == 4 || v == 8 ? v : kPiecesPerWave;
    }
    if (const char *e = std::getenv("LOCREC_SG_DICT_THREADS")) {
        const int v = std::atoi(e);
        if (v == 256 || v == 512 || v == 1024) g->dict_threads = v;
    }
    if (const char *e = std::getenv("LOCREC_SG_DICT_PPW")) {
        const int v = std::atoi(e);
        if (v == 1 || v == 2 || v == 4) g->dict_ppw = v;
    }
    g->env_fused = std::getenv("LOCREC_SG_FUSED") != nullptr;                // the fused iteration (an experiment)
    g->fused_one_stream = std::getenv("LOCREC_SG_FUSED_ONE_STREAM") != nullptr;
    g->env_persist = std::getenv("LOCREC_SG_PERSIST") != nullptr;            // the persistent form (an experiment)
}

// shard_index / shard_count: this handle holds the edges whose SOURCE vertex (its index in the
// sorted vertex set) is congruent to shard_index modulo shard_count -- "rows of P sharded"; the
// vertex set, the live set and the x layout are built from ALL edges and identical on every shard.
static int32_t sg_create_impl(int64_t ne, const int64_t *src, const int64_t *dst, const double *w,
                              int32_t shard_index, int32_t shard_count, bool by_target, locrec_sg_graph **out)
{
    if (!out) return fail(LOCREC_E_INVALID_ARG, "out_graph is NULL");
    *out = nullptr;
    if (ne < 0 || (ne > 0 && (!src || !dst || !w)))
        return fail(LOCREC_E_INVALID_ARG, "bad edge arrays");
    LOCREC_TRY(ensure_device());
    std::unique_ptr<locrec_sg_graph> g(new (std::nothrow) locrec_sg_graph);
    if (!g) return fail(LOCREC_E_OOM, "host allocation failed");
    LOCREC_HIP_TRY(hipGetDevice(&g->device));
    LOCREC_HIP_TRY(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    g->own_stream = true;
    g->ne = ne;
    sg_read_env(g.get());

    // vertexes = distinct(source_id U target_id), StochasticRecommender.scala:42-49
    std::vector<int64_t> &vid = g->vid;
    try {
        vid.resize((size_t)(2 * ne));
    } catch (...) {
        return fail(LOCREC_E_OOM, "host allocation failed");
    }
    // Ids that sit close together (the generator's and most real id spaces: categories, places, persons numbered
    // one after the other) are ranked through a table over [min, max] - three linear passes - instead of sorting
    // 2E ids and bisecting twice per edge (0.5 s of the 0.55 s locrec_sg_create took at cfg3); any other id space
    // takes the sort.  Both give the same ascending `vid` and the same indices.
    int64_t id_lo = INT64_MAX, id_hi = INT64_MIN;
    for (int64_t e = 0; e < ne; ++e) {
        id_lo = std::min(id_lo, std::min(src[e], dst[e]));
        id_hi = std::max(id_hi, std::max(src[e], dst[e]));
    }
    const uint64_t id_span = ne > 0 ? (uint64_t)id_hi - (uint64_t)id_lo : 0;  // (max - min, exact in unsigned arithmetic)
    const bool dense_ids = ne > 0 && id_span < (uint64_t)(8 * ne) + (1u << 20) && !g->env_no_dense_ids;
    std::vector<int32_t> rank_of;  // dense_ids: id - id_lo -> vertex index
    if (dense_ids) {
        try {
            rank_of.assign((size_t)id_span + 1, 0);
        } catch (...) {
            return fail(LOCREC_E_OOM, "host allocation failed");
        }
        for (int64_t e = 0; e < ne; ++e) {
            rank_of[(size_t)((uint64_t)src[e] - (uint64_t)id_lo)] = 1;
            rank_of[(size_t)((uint64_t)dst[e] - (uint64_t)id_lo)] = 1;
        }
        size_t k = 0;
        for (size_t i = 0; i <= (size_t)id_span; ++i)
            if (rank_of[i]) {
                rank_of[i] = (int32_t)k;
                vid[k++] = (int64_t)((uint64_t)id_lo + i);
            }
        vid.resize(k);
        vid.shrink_to_fit();
    } else {
        for (int64_t e = 0; e < ne; ++e) {
            vid[2 * e] = src[e];
            vid[2 * e + 1] = dst[e];
        }
        std::sort(vid.begin(), vid.end());
        vid.erase(std::unique(vid.begin(), vid.end()), vid.end());
    }
    const int64_t nv = (int64_t)vid.size();
    if (nv >= ((int64_t)1 << 31) - 2) return fail(LOCREC_E_INVALID_ARG, "too many vertices");
    g->nv = nv;

    if (shard_count < 1 || shard_index < 0 || shard_index >= shard_count)
        return fail(LOCREC_E_INVALID_ARG, "bad shard specification");
    g->shard_index = shard_index;
    g->shard_count = shard_count;
    std::vector<int32_t> cs((size_t)ne), ct((size_t)ne);
    std::vector<int32_t> deg((size_t)nv + 1, 0);   // in-degree over this shard's edges (piece plan)
    std::vector<int32_t> gdeg((size_t)nv + 1, 0);  // in-degree over all edges (live set, row classes)
    for (int64_t e = 0; e < ne; ++e) {
        if (dense_ids) {
            cs[e] = rank_of[(size_t)((uint64_t)src[e] - (uint64_t)id_lo)];
            ct[e] = rank_of[(size_t)((uint64_t)dst[e] - (uint64_t)id_lo)];
        } else {
            cs[e] = (int32_t)(std::lower_bound(vid.begin(), vid.end(), src[e]) - vid.begin());
            ct[e] = (int32_t)(std::lower_bound(vid.begin(), vid.end(), dst[e]) - vid.begin());
        }
        ++gdeg[ct[e]];
    }
    std::vector<int32_t>().swap(rank_of);
    // live vertices, rows with at most two full pieces first (ascending id inside each class): the
    // first `n_short` rows are then treated uniformly (three row-major partial slots each)
    g->live_of.assign((size_t)nv, -1);
    int32_t n_short_global = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (int64_t v = 0; v < nv; ++v)
            if (gdeg[v] > 0 && (gdeg[v] / kSlots > 2) == (pass == 1)) {
                g->live_of[v] = (int32_t)g->live_vertex.size();
                g->live_vertex.push_back((int32_t)v);
            }
        if (pass == 0) n_short_global = (int32_t)g->live_vertex.size();
    }
    const int32_t T = (int32_t)g->live_vertex.size();
    g->nlive = T;
    const int32_t slot_d = T;
    // which edges this handle keeps: rows of P (sources, by vertex index) or rows of P^T (targets, by
    // LIVE index: the live rows are sorted by degree, so a stride spreads the heavy ones evenly)
    auto owned = [&](int64_t e) {
        return by_target ? g->live_of[ct[e]] % shard_count == shard_index : cs[e] % shard_count == shard_index;
    };
    for (int64_t e = 0; e < ne; ++e)
        if (owned(e)) ++deg[ct[e]];

    // piece plan over the live rows: full pieces first (row order), then remainder pieces by class 6..0
    std::vector<RowMeta> meta((size_t)T);
    std::vector<int32_t> long_rows;
    int64_t nfull_total = 0;
    int64_t nseg_cls[7] = {0, 0, 0, 0, 0, 0, 0};
    std::vector<int8_t> rcls((size_t)T, -1);
    for (int32_t l = 0; l < T; ++l) {
        const int d = deg[g->live_vertex[l]];
        RowMeta m{0, 0, -1};
        m.nfull = d / kSlots;
        m.full_begin = (int32_t)nfull_total;
        nfull_total += m.nfull;
        const int rem = d % kSlots;
        if (rem > 0) {
            const int c = ceil_log2((rem + 3) / 4);
            rcls[l] = (int8_t)c;
            ++nseg_cls[c];
        }
        if (m.nfull > kLongRow) long_rows.push_back(l);
        meta[l] = m;
    }
    int64_t piece_begin_cls[7], part_begin_cls[7];
    int64_t np = nfull_total, npart = nfull_total;
    for (int c = 6; c >= 0; --c) {
        const int segs_per_piece = 64 >> c;
        const int64_t pieces = (nseg_cls[c] + segs_per_piece - 1) / segs_per_piece;
        piece_begin_cls[c] = np;
        part_begin_cls[c] = npart;
        np += pieces;
        npart += pieces * segs_per_piece;
    }
    if (np >= ((int64_t)1 << 31) / kSlots) return fail(LOCREC_E_INVALID_ARG, "graph too large for int32 slot ids");
    g->npieces = (int32_t)np;
    g->nlong = (int32_t)long_rows.size();

    std::vector<int32_t> col((size_t)np * kSlots, slot_d);
    std::vector<double> wv((size_t)np * kSlots, 0.0);
    std::vector<int2> pinfo((size_t)np);
    for (int64_t p = 0; p < nfull_total; ++p) pinfo[p] = make_int2((int)p, 6);
    for (int c = 6; c >= 0; --c) {
        const int segs_per_piece = 64 >> c;
        const int64_t pieces = (nseg_cls[c] + segs_per_piece - 1) / segs_per_piece;
        for (int64_t i = 0; i < pieces; ++i)
            pinfo[piece_begin_cls[c] + i] = make_int2((int)(part_begin_cls[c] + i * segs_per_piece), c);
    }
    std::vector<int64_t> cursor((size_t)T, 0);      // edges placed so far in the row
    std::vector<int64_t> rem_slot0((size_t)T, -1);  // absolute slot of remainder element 0
    {
        int64_t seg_used[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int32_t l = 0; l < T; ++l) {
            const int c = rcls[l];
            if (c < 0) continue;
            const int segs_per_piece = 64 >> c;
            const int64_t s = seg_used[c]++;
            const int64_t piece = piece_begin_cls[c] + s / segs_per_piece;
            const int seg = (int)(s % segs_per_piece);
            meta[l].rem = (int32_t)(part_begin_cls[c] + s);
            rem_slot0[l] = piece * kSlots + (int64_t)seg * (4 << c);
        }
    }
    // scatter the edges in edge-list order (stable within a row); remember where the
    // out-edges of source-only vertices landed
    auto wofs = [](int64_t slot) {
        const int64_t piece = slot / kSlots;
        const int k = (int)(slot % kSlots);
        const int lane = k >> 2, j = k & 3;
        return piece * kSlots + (j >> 1) * 128 + lane * 2 + (j & 1);
    };
    g->dead_ptr.assign((size_t)nv + 1, 0);
    for (int64_t e = 0; e < ne; ++e)
        if (owned(e) && g->live_of[cs[e]] < 0) ++g->dead_ptr[cs[e] + 1];
    for (int64_t v = 0; v < nv; ++v) g->dead_ptr[v + 1] += g->dead_ptr[v];
    g->dead_slots.resize((size_t)g->dead_ptr[nv]);
    std::vector<int64_t> dcur(g->dead_ptr.begin(), g->dead_ptr.end() - 1);
    for (int64_t e = 0; e < ne; ++e) {
        if (!owned(e)) continue;
        const int32_t l = g->live_of[ct[e]];
        const int64_t k = cursor[l]++;
        const RowMeta &m = meta[l];
        int64_t slot;
        if (k < (int64_t)m.nfull * kSlots)
            slot = (int64_t)m.full_begin * kSlots + k;
        else
            slot = rem_slot0[l] + (k - (int64_t)m.nfull * kSlots);
        const int32_t sl = g->live_of[cs[e]];
        if (sl >= 0) {
            col[slot] = sl;
        } else {
            col[slot] = slot_d;
            g->dead_slots[dcur[cs[e]]++] = (int32_t)slot;
        }
        wv[wofs(slot)] = w[e];
    }

    g->use16 = T + 2 <= 65536 && !g->env_no_col16;
    g->device_sweep_bytes = 0;
    if (!g->no_pack) {
        void *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocDefault) == hipSuccess && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
            g->h_poll = static_cast<int32_t *>(hp);
            g->h_poll_dev = static_cast<int32_t *>(dp);
        } else {
            (void)hipGetLastError();
            if (hp) (void)hipHostFree(hp);
            g->no_pack = true;
        }
    }
    if (g->env_gs > 0) {
        int dev = 0, ncu = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        g->gs_blocks = g->env_gs * std::max(1, ncu);
    }
    if (g->use16) {
        std::vector<unsigned short> col16(col.size());
        for (size_t i = 0; i < col.size(); ++i) col16[i] = (unsigned short)col[i];
        LOCREC_TRY(g->col16.upload(col16, g->stream));
        LOCREC_HIP_TRY(hipStreamSynchronize(g->stream));
    } else {
        LOCREC_TRY(g->col4.upload(reinterpret_cast<const int4 *>(col.data()), (size_t)np * 64, g->stream));
    }
    LOCREC_TRY(g->w2.upload(reinterpret_cast<const double2 *>(wv.data()), (size_t)np * 128, g->stream));
    // The dictionary of the edge weights, built on the device from the weights just uploaded: their bit patterns sorted
    // (rocPRIM radix sort) and run-length encoded; when there are at most kDictMax distinct ones the table is ordered by
    // FREQUENCY (the lanes of a wave mostly ask for the common values: the 32 most common ones then sit in 32 different
    // LDS bank pairs) and every slot's index found by bisection in the sorted values (sg_build_widx).  (A host pass with a
    // hash table took 37 ms of a 100 ms create at cfg3; this takes ~2.)
    if (!g->env_no_dict && np > 0) {
        const size_t nslots = (size_t)np * kSlots;
        hipStream_t s = g->stream;
        DevBuf<uint64_t> ka, kb;
        DevBuf<unsigned int> counts;
        DevBuf<int32_t> nuniq;
        DevBuf<unsigned char> tmp;
        LOCREC_TRY(g->widx.alloc(nslots));  // (before the temporaries: what stays resident is allocated first)
        LOCREC_TRY(g->dict.alloc(kDictMax));
        LOCREC_TRY(ka.alloc(nslots));
        LOCREC_TRY(kb.alloc(nslots));
        LOCREC_TRY(counts.alloc(nslots));
        LOCREC_TRY(nuniq.alloc(1));
        LOCREC_HIP_TRY(hipMemcpyAsync(ka.p, g->w2.p, nslots * 8, hipMemcpyDeviceToDevice, s));
        size_t b1 = 0, b2 = 0;
        LOCREC_HIP_TRY(prim::sort_keys(nullptr, b1, ka.p, kb.p, nslots, 0u, 64u, s));
        LOCREC_HIP_TRY(prim::run_length_encode(nullptr, b2, kb.p, nslots, ka.p, counts.p, nuniq.p, s));
        LOCREC_TRY(tmp.alloc(std::max(b1, b2)));
        LOCREC_HIP_TRY(prim::sort_keys(tmp.p, b1, ka.p, kb.p, nslots, 0u, 64u, s));
        LOCREC_HIP_TRY(prim::run_length_encode(tmp.p, b2, kb.p, nslots, ka.p, counts.p, nuniq.p, s));
        int32_t nu = 0;
        LOCREC_HIP_TRY(hipMemcpyAsync(&nu, nuniq.p, sizeof(nu), hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
        if (nu >= 1 && nu <= kDictMax) {
            std::vector<uint64_t> vals((size_t)nu);
            std::vector<unsigned int> cnt((size_t)nu);
            LOCREC_HIP_TRY(hipMemcpyAsync(vals.data(), ka.p, (size_t)nu * 8, hipMemcpyDeviceToHost, s));
            LOCREC_HIP_TRY(hipMemcpyAsync(cnt.data(), counts.p, (size_t)nu * 4, hipMemcpyDeviceToHost, s));
            LOCREC_HIP_TRY(hipStreamSynchronize(s));
            std::vector<int32_t> order((size_t)nu);
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cnt[(size_t)a] > cnt[(size_t)b]; });
            std::vector<double> dict_h((size_t)nu);
            std::vector<unsigned short> rank_h((size_t)nu);  // position in the sorted values -> table index
            for (int32_t r = 0; r < nu; ++r) {
                std::memcpy(&dict_h[(size_t)r], &vals[(size_t)order[(size_t)r]], 8);
                rank_h[(size_t)order[(size_t)r]] = (unsigned short)r;
            }
            DevBuf<unsigned short> rank;
            LOCREC_HIP_TRY(hipMemcpyAsync(g->dict.p, dict_h.data(), (size_t)nu * 8, hipMemcpyHostToDevice, s));
            LOCREC_TRY(rank.upload(rank_h, s));
            hipLaunchKernelGGL(sg_build_widx, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, s,
                               reinterpret_cast<const double *>(g->w2.p), (int64_t)nslots, ka.p, rank.p, nu, g->widx.p);
            LOCREC_HIP_TRY(hipGetLastError());
            LOCREC_HIP_TRY(hipStreamSynchronize(s));  // (locals)
            g->ndict = nu;
        } else {
            g->widx.release();
            g->dict.release();
        }
    }
    LOCREC_TRY(g->pinfo.upload(pinfo, g->stream));
    LOCREC_TRY(g->xbuf.alloc((size_t)(2 * (T + 2))));
    LOCREC_TRY(g->parts.alloc(2 * kParts));
    LOCREC_TRY(g->state.alloc(1));
    if (g->dead_slots.size() > (size_t)INT32_MAX) return fail(LOCREC_E_INVALID_ARG, "too many out-edges of source-only vertices");
    LOCREC_TRY(g->dead_slots_dev.upload(g->dead_slots, g->stream));
    g->layout_bytes = np * kSlots * 12 + np * 8 + (int64_t)T * 12;
    // what one sweep + finalize really moves in THIS layout: columns (2 or 4 B) and fp64 weights of
    // every slot (padding included), piece descriptors, one partial written and read back per
    // segment, x read and x' written once per live vertex
    // (dictionary form: a uint16 index per slot instead of the fp64 weight; the value table is read once per block)
    g->device_sweep_bytes = np * kSlots * (int64_t)((g->use16 ? 2 : 4) + (g->ndict > 0 ? 2 : 8)) + np * 8 + npart * 16 + (int64_t)T * 16;
    {
        // row-major partial slots (l*3 + j for rows with <= 2 full pieces, a contiguous run behind
        // them for the others); seg_out maps a segment (old contiguous numbering: pinfo.x + seg) to
        // its slot, lane_out does the same per leader lane for the persistent kernel
        std::vector<int32_t> long_begin((size_t)T, -1);
        std::vector<int4> lrows;
        int64_t pa = 3 * (int64_t)T;
        g->n_short = n_short_global;
        for (int32_t l = 0; l < T; ++l) {
            if (l >= n_short_global) {  // a long row somewhere: it lives in the long area on every shard
                long_begin[l] = (int32_t)pa;
                lrows.push_back(make_int4(l, (int)pa, meta[l].nfull, meta[l].rem >= 0 ? 1 : 0));
                pa += meta[l].nfull + 1;
            }
        }
        std::vector<int32_t> lane_out((size_t)np * 64, -1);
        std::vector<int32_t> seg_out((size_t)npart, -1);
        std::vector<int32_t> rem_owner((size_t)npart, -1);  // old partial index -> live row
        for (int32_t l = 0; l < T; ++l) {
            const RowMeta &m = meta[l];
            for (int j = 0; j < m.nfull; ++j) {  // full piece id == its old partial index
                const int32_t slot = long_begin[l] >= 0 ? long_begin[l] + j : 3 * l + j;
                lane_out[(size_t)(m.full_begin + j) * 64] = slot;
                seg_out[m.full_begin + j] = slot;
            }
            if (m.rem >= 0) rem_owner[m.rem] = l;
        }
        for (int64_t p = nfull_total; p < np; ++p) {
            const int c = pinfo[p].y, base = pinfo[p].x;
            for (int sgm = 0; sgm < (64 >> c); ++sgm) {
                const int32_t l = rem_owner[base + sgm];
                if (l < 0) continue;
                const int32_t slot = long_begin[l] >= 0 ? long_begin[l] + meta[l].nfull : 3 * l + 2;
                lane_out[(size_t)p * 64 + ((size_t)sgm << c)] = slot;
                seg_out[base + sgm] = slot;
            }
        }
        // fused iteration (sg_sweep_fused): its own slot map - four slots per class-A row, a run per longer row with the X
        // slot at its end - and the second piece list: the edges whose SOURCE is a longer row, grouped by target row
        // into one pow2 segment each (at most 256 of them per row, or the handle keeps the two-launch form).  Built only
        // where it is asked for: an experiment (LOCREC_SG_FUSED), not the product path.
        if (shard_count == 1 && g->env_fused) {
            const int32_t n_long = T - n_short_global;
            std::vector<int32_t> seg_fa((size_t)npart, -1), xslot((size_t)T, -1), lbegin((size_t)T, -1);
            std::vector<int4> lrows_f;
            int64_t pa4 = 4 * (int64_t)n_short_global;
            for (int32_t l = n_short_global; l < T; ++l) {
                lbegin[l] = (int32_t)pa4;
                lrows_f.push_back(make_int4(l, (int)pa4, meta[l].nfull, meta[l].rem >= 0 ? 1 : 0));
                pa4 += meta[l].nfull + 2;
            }
            for (int32_t l = 0; l < T; ++l) {
                const RowMeta &m = meta[l];
                for (int j = 0; j < m.nfull; ++j) seg_fa[m.full_begin + j] = l < n_short_global ? 4 * l + j : lbegin[l] + j;
                xslot[l] = l < n_short_global ? 4 * l + 3 : lbegin[l] + m.nfull + 1;
            }
            for (int64_t p = nfull_total; p < np; ++p) {
                const int c = pinfo[p].y, base = pinfo[p].x;
                for (int sgm = 0; sgm < (64 >> c); ++sgm) {
                    const int32_t l = rem_owner[base + sgm];
                    if (l >= 0) seg_fa[base + sgm] = l < n_short_global ? 4 * l + 2 : lbegin[l] + meta[l].nfull;
                }
            }
            std::stable_partition(lrows_f.begin(), lrows_f.end(), [](const int4 &r) { return r.z > kLongRow; });
            // the second piece list
            std::vector<int32_t> cnt2((size_t)T, 0);
            for (int64_t e = 0; e < ne; ++e) {
                const int32_t sl = g->live_of[cs[e]];
                if (sl >= n_short_global) ++cnt2[g->live_of[ct[e]]];
            }
            bool ok = pa4 < ((int64_t)1 << 30);
            int64_t seg_n[7] = {0, 0, 0, 0, 0, 0, 0};
            std::vector<int8_t> cls2((size_t)T, -1);
            for (int32_t l = 0; l < T && ok; ++l) {
                if (cnt2[l] == 0) continue;
                if (cnt2[l] > kSlots) { ok = false; break; }
                int c = 0;
                while ((4 << c) < cnt2[l]) ++c;
                cls2[l] = (int8_t)c;
                ++seg_n[c];
            }
            g->fused_ok = ok;
            g->use_fused = ok;
            if (ok) {
                int64_t piece0[7], segbase[7], np2 = 0, nseg2 = 0;
                for (int c = 0; c < 7; ++c) {
                    piece0[c] = np2;
                    segbase[c] = nseg2;
                    const int64_t per = 64 >> c, pcs = (seg_n[c] + per - 1) / per;
                    np2 += pcs;
                    nseg2 += pcs * per;
                }
                std::vector<int32_t> col2((size_t)np2 * kSlots, -1), seg2((size_t)nseg2, -1);
                std::vector<double> wv2((size_t)np2 * kSlots, 0.0);
                std::vector<int2> pinfo2((size_t)np2);
                for (int c = 0; c < 7; ++c)
                    for (int64_t p = piece0[c]; p < (c < 6 ? piece0[c + 1] : np2); ++p)
                        pinfo2[p] = make_int2((int)(segbase[c] + (p - piece0[c]) * (64 >> c)), c);
                std::vector<int64_t> slot0((size_t)T, -1);
                int64_t used[7] = {0, 0, 0, 0, 0, 0, 0};
                for (int32_t l = 0; l < T; ++l) {
                    const int c = cls2[l];
                    if (c < 0) continue;
                    const int64_t sidx = used[c]++, per = 64 >> c;
                    slot0[l] = (piece0[c] + sidx / per) * kSlots + (sidx % per) * (4 << c);
                    seg2[segbase[c] + sidx] = xslot[l];
                }
                std::vector<int32_t> cur2((size_t)T, 0);
                for (int64_t e = 0; e < ne; ++e) {  // edge-list order inside a row, like the main pieces
                    const int32_t sl = g->live_of[cs[e]];
                    if (sl < n_short_global) continue;
                    const int32_t l = g->live_of[ct[e]];
                    const int64_t slot = slot0[l] + cur2[l]++;
                    col2[slot] = sl;
                    wv2[wofs(slot)] = w[e];
                }
                g->pa4 = (int32_t)std::max<int64_t>(1, pa4);
                g->npieces2 = (int32_t)np2;
                g->nduty = std::max(1, (n_short_global + 63) / 64);
                LOCREC_TRY(g->seg_fa.upload(seg_fa, g->stream));
                LOCREC_TRY(g->lrows_f.upload(lrows_f, g->stream));
                LOCREC_TRY(g->col2.upload(col2, g->stream));
                LOCREC_TRY(g->seg2.upload(seg2, g->stream));
                LOCREC_TRY(g->pinfo2.upload(pinfo2, g->stream));
                LOCREC_TRY(g->w2b.upload(reinterpret_cast<const double2 *>(wv2.data()), (size_t)np2 * 128, g->stream));
                LOCREC_TRY(g->PA4.alloc((size_t)4 * g->pa4));
                LOCREC_TRY(g->XL.alloc((size_t)4 * std::max(1, n_long)));
                LOCREC_TRY(g->D2W.alloc((size_t)4 * (g->nduty + kParts)));
                LOCREC_TRY(g->fused_conv.alloc(2));
                // (slots no kernel writes - a remainder or X slot of a row without one - stay 0.0 for good)
                LOCREC_HIP_TRY(hipMemsetAsync(g->PA4.p, 0, g->PA4.bytes(), g->stream));
                LOCREC_HIP_TRY(hipMemsetAsync(g->XL.p, 0, g->XL.bytes(), g->stream));
                LOCREC_HIP_TRY(hipMemsetAsync(g->D2W.p, 0, g->D2W.bytes(), g->stream));
                LOCREC_HIP_TRY(hipStreamSynchronize(g->stream));  // (the vectors are locals)
                if (g->use_fused && !g->fused_one_stream)
                    LOCREC_HIP_TRY(hipStreamCreateWithFlags(&g->stream2, hipStreamNonBlocking));
            }
        }
        if (pa >= ((int64_t)1 << 30)) return fail(LOCREC_E_INVALID_ARG, "graph too large for int32 partial slots");
        // rows summed by a whole wave first, then the ones a single thread sums (slot order is
        // not affected: every entry carries its own first slot)
        std::stable_partition(lrows.begin(), lrows.end(), [](const int4 &r) { return r.z > kLongRow; });
        g->n_crows = 0;
        for (const int4 &r : lrows) g->n_crows += r.z > kLongRow ? 1 : 0;
        g->pa_stride = (int32_t)pa;
        g->nlrows = (int32_t)lrows.size();
        LOCREC_TRY(g->seg_out.upload(seg_out, g->stream));
        LOCREC_TRY(g->lrows.upload(lrows, g->stream));
        LOCREC_TRY(g->PA.alloc((size_t)(2 * pa)));
        LOCREC_HIP_TRY(hipMemsetAsync(g->PA.p, 0, (size_t)(2 * pa) * sizeof(double), g->stream));
        LOCREC_HIP_TRY(hipStreamSynchronize(g->stream));
        int dev = 0, ncu = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        const int64_t waves = (int64_t)ncu * 8;
        const int64_t need = waves > 0 ? (np + waves - 1) / waves : 1 << 30;
        const size_t lds = (size_t)(T + 2) * 8;
        g->persist_pw = need <= 4 ? 4 : need <= 12 ? 12 : 0;  // 8 and 16 spill registers: not built
        g->persist_ok = shard_count == 1 && g->persist_pw > 0 && lds <= 128 * 1024 && ncu > 0 && pa < ((int64_t)1 << 30) &&
                        g->env_persist;  // opt-in: see the note above sg_persistent
        if (g->persist_ok) {
            g->persist_blocks = ncu;
            g->persist_lds = lds;
            LOCREC_TRY(g->lane_out.upload(lane_out, g->stream));
            LOCREC_TRY(g->barrier.alloc(1));
            if (debug_env("LOCREC_SG_DEBUG_PHASES")) LOCREC_TRY(g->dbg.alloc(4));
            LOCREC_HIP_TRY(hipStreamSynchronize(g->stream));
        }
    }
    LOCREC_HIP_TRY(hipStreamSynchronize(g->stream));
    std::vector<int32_t>().swap(g->dead_slots);  // resident on the device now (dead_slots_dev); dead_ptr stays on the host
    *out = g.release();
    return LOCREC_OK;
}

extern "C" int32_t locrec_sg_create(int64_t ne, const int64_t *src, const int64_t *dst, const double *w,
                                    locrec_sg_graph **out) try
{
    return sg_create_impl(ne, src, dst, w, 0, 1, false, out);
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_create_sharded(int64_t ne, const int64_t *src, const int64_t *dst, const double *w,
                                            int32_t shard_index, int32_t shard_count, locrec_sg_graph **out) try
{
    return sg_create_impl(ne, src, dst, w, shard_index, shard_count, false, out);
} LOCREC_CATCH_ALL

// Rows of P^T (targets) sharded instead: live row l belongs to shard l % shard_count, which then holds
// ALL inbound edges of its rows, so locrec_sg_shard_sigma's output is complete (and summed in the
// single-GPU order) for the rows it owns and 0 elsewhere; the caller all-GATHERs the owned entries
// instead of all-reducing sigma (half the traffic, bit-identical to the unsharded result).
extern "C" int32_t locrec_sg_create_target_sharded(int64_t ne, const int64_t *src, const int64_t *dst,
                                                   const double *w, int32_t shard_index, int32_t shard_count,
                                                   locrec_sg_graph **out) try
{
    return sg_create_impl(ne, src, dst, w, shard_index, shard_count, true, out);
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_destroy(locrec_sg_graph *g) try
{
    if (!g) return LOCREC_OK;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    delete g;  // the destructor destroys an owned stream
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_info(const locrec_sg_graph *g, int64_t *out_v, int64_t *out_e,
                                  int64_t *out_bytes) try
{
    if (!g) return fail(LOCREC_E_INVALID_ARG, "graph is NULL");
    if (out_v) *out_v = g->nv;
    if (out_e) *out_e = g->ne;
    // E*(ib+wb) + T*rb + V*8 (read x) + T*8 (write x' rows), ib=4, wb=8, rb=8
    if (out_bytes) *out_bytes = g->ne * 12 + (int64_t)g->nlive * 8 + g->nv * 8 + (int64_t)g->nlive * 8;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_device_bytes(const locrec_sg_graph *g, int64_t *out_bytes) try
{
    if (!g || !out_bytes) return fail(LOCREC_E_INVALID_ARG, "NULL argument");
    *out_bytes = g->device_sweep_bytes;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_weight_dictionary(const locrec_sg_graph *g, int32_t *out_entries) try
{
    if (!g || !out_entries) return fail(LOCREC_E_INVALID_ARG, "NULL argument");
    *out_entries = g->ndict;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_set_stream(locrec_sg_graph *g, void *s) try
{
    if (!g) return fail(LOCREC_E_INVALID_ARG, "graph is NULL");
    if (g->own_stream && g->stream) {
        (void)hipStreamSynchronize(g->stream);
        (void)hipStreamDestroy(g->stream);
    }
    g->stream = reinterpret_cast<hipStream_t>(s);
    g->own_stream = false;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_synchronize(locrec_sg_graph *g) try
{
    if (!g) return fail(LOCREC_E_INVALID_ARG, "graph is NULL");
    LOCREC_HIP_TRY(hipSetDevice(g->device));
    LOCREC_HIP_TRY(hipStreamSynchronize(g->stream));
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_profile_enable(locrec_sg_graph *g, int32_t on) try
{
    if (!g) return fail(LOCREC_E_INVALID_ARG, "graph is NULL");
    g->prof.on = on != 0;
    g->prof.used = 0;
    g->persist_units = 0;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_profile_read(locrec_sg_graph *g, double *ms, int64_t *launches) try
{
    if (!g) return fail(LOCREC_E_INVALID_ARG, "graph is NULL");
    LOCREC_HIP_TRY(hipSetDevice(g->device));
    LOCREC_TRY(g->prof.read(g->stream, ms, launches));
    if (launches && g->persist_units > 0) *launches = g->persist_units;  // one launch = many sweeps
    g->persist_units = 0;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

namespace {

// step() (StochasticRecommender.scala:92-106) as a stream of launches.  eps2 < 0 disables the
// convergence exit (fixed number of sweeps, used by the benchmark entry point).
struct RequestSetup {
    int32_t tv, target_x, n_plain_dead;
    bool q_dead;
};

// isVertexExist (:73-77 / :70), x0 (:51-54), state reset, and the request's out-edge slots D -> Q: one sg_begin
// launch on the handle's stream - or, for a group, the graph's row of the sg_begin_group table (`row`).
// `alpha`, `eps2`: the request's values for SgReq (used by the single-graph iteration only).
int32_t begin_request(locrec_sg_graph *g, int64_t vertex_id, RequestSetup *rs, double alpha = 0, double eps2 = -1,
                      SgBegin *row = nullptr)
{
    auto it = std::lower_bound(g->vid.begin(), g->vid.end(), vertex_id);
    if (it == g->vid.end() || *it != vertex_id)
        return fail(LOCREC_E_NOT_FOUND, "No such vertex in the graph: %lld", (long long)vertex_id);
    const int32_t tv = (int32_t)(it - g->vid.begin());
    LOCREC_HIP_TRY(hipSetDevice(g->device));
    const int32_t T = g->nlive;
    const bool q_dead = g->live_of[tv] < 0;
    rs->tv = tv;
    rs->q_dead = q_dead;
    rs->target_x = q_dead ? T + 1 : g->live_of[tv];
    rs->n_plain_dead = (int32_t)(g->nv - T) - (q_dead ? 1 : 0);
    if (!row && !g->req_dev.p) LOCREC_TRY(g->req_dev.alloc(1));
    SgBegin b{};
    b.x = g->xbuf.p;
    b.parts = g->parts.p;
    b.st = g->state.p;
    b.col = g->use16 ? static_cast<void *>(g->col16.p) : static_cast<void *>(g->col4.p);
    b.slots = g->dead_slots_dev.p;
    b.req_dst = row ? nullptr : g->req_dev.p;
    b.x0 = 1.0 / (double)g->nv;     // :51-54
    b.nx = T + 2;
    b.col16 = g->use16 ? 1 : 0;
    b.T = T;
    // point the previous request's out-edge slots back at D, this request's at Q (nothing to do when it is the same
    // source-only vertex again: its slots already point at Q)
    const int64_t new_off = q_dead ? g->dead_ptr[tv] : 0;
    const int32_t n_new = q_dead ? (int32_t)(g->dead_ptr[tv + 1] - g->dead_ptr[tv]) : 0;
    if (!(n_new == g->n_patched && (n_new == 0 || new_off == g->patched_off))) {
        b.old_off = (int32_t)g->patched_off;
        b.n_old = g->n_patched;
        b.new_off = (int32_t)new_off;
        b.n_new = n_new;
    }
    g->patched_off = new_off;
    g->n_patched = n_new;
    b.req = SgReq{rs->target_x, rs->n_plain_dead, q_dead ? 1 : 0, 0, alpha, 1 - alpha /* :121 */, eps2};
    if (row) {
        *row = b;
        return LOCREC_OK;
    }
    hipLaunchKernelGGL(sg_begin, dim3(kBeginBlocks), dim3(256), 0, g->stream, b);
    return LOCREC_OK;
}

// one sg_sweep launch over this handle's pieces
void launch_sweep(locrec_sg_graph *g, const double *x_in)
{
    hipStream_t s = g->stream;
    const int sweep_blocks = (g->npieces + 4 * g->ppw - 1) / (4 * g->ppw);
    if (sweep_blocks <= 0) return;
    const void *colv = g->use16 ? static_cast<const void *>(g->col16.p) : static_cast<const void *>(g->col4.p);
    const v2d *wv2 = reinterpret_cast<const v2d *>(g->w2.p);
    SgState *st = g->state.p;
    if (g->gs_blocks > 0) {
        const int blocks = std::min(g->gs_blocks, (g->npieces + 3) / 4);
        if (g->use16)
            LOCREC_LAUNCH_PROFILED(g->prof, (sg_sweep_gs<true>), dim3(blocks), dim3(256), 0, s, colv, wv2, g->pinfo.p,
                                   g->seg_out.p, x_in, g->PA.p, g->npieces, st);
        else
            LOCREC_LAUNCH_PROFILED(g->prof, (sg_sweep_gs<false>), dim3(blocks), dim3(256), 0, s, colv, wv2, g->pinfo.p,
                                   g->seg_out.p, x_in, g->PA.p, g->npieces, st);
        return;
    }
    if (g->ndict > 0) {
        const int wpb = g->dict_threads / 64;
        const int ppw = g->dict_ppw;
        const int blocks = (g->npieces + wpb * ppw - 1) / (wpb * ppw);
        const size_t lds = (size_t)g->ndict * sizeof(double);
        const v4h *wi = reinterpret_cast<const v4h *>(g->widx.p);
#define LOCREC_SWEEP_DICT(C16, PPW)                                                                                     \
    LOCREC_LAUNCH_PROFILED(g->prof, (sg_sweep_dict<C16, PPW>), dim3(blocks), dim3(g->dict_threads), lds, s, colv, wi,    \
                           g->dict.p, g->ndict, g->pinfo.p, g->seg_out.p, x_in, g->PA.p, g->npieces, st)
        if (g->use16) {
            if (ppw == 1) LOCREC_SWEEP_DICT(true, 1); else if (ppw == 2) LOCREC_SWEEP_DICT(true, 2); else LOCREC_SWEEP_DICT(true, 4);
        } else {
            if (ppw == 1) LOCREC_SWEEP_DICT(false, 1); else if (ppw == 2) LOCREC_SWEEP_DICT(false, 2); else LOCREC_SWEEP_DICT(false, 4);
        }
#undef LOCREC_SWEEP_DICT
        return;
    }
#define LOCREC_SWEEP(C16, PPW)                                                                                      \
    LOCREC_LAUNCH_PROFILED(g->prof, (sg_sweep<C16, PPW>), dim3(sweep_blocks), dim3(256), 0, s, colv, wv2, g->pinfo.p, \
                           g->seg_out.p, x_in, g->PA.p, g->npieces, st)
    if (g->use16) {
        if (g->ppw == 1) LOCREC_SWEEP(true, 1); else if (g->ppw == 2) LOCREC_SWEEP(true, 2);
        else if (g->ppw == 8) LOCREC_SWEEP(true, 8); else LOCREC_SWEEP(true, 4);
    } else {
        if (g->ppw == 1) LOCREC_SWEEP(false, 1); else if (g->ppw == 2) LOCREC_SWEEP(false, 2);
        else if (g->ppw == 8) LOCREC_SWEEP(false, 8); else LOCREC_SWEEP(false, 4);
    }
#undef LOCREC_SWEEP
}

int32_t enqueue_iterations(locrec_sg_graph *g, int64_t vertex_id, double alpha, double eps2,
                           int64_t max_iterations, bool poll)
{
    g->have_result = false;
    g->shard_active = false;
    if (g->shard_count != 1)
        return fail(LOCREC_E_INVALID_ARG, "a sharded graph is iterated with locrec_sg_shard_* (it holds only part of the edges)");
    if (max_iterations > INT32_MAX) max_iterations = INT32_MAX;
    RequestSetup rs{};
    LOCREC_TRY(begin_request(g, vertex_id, &rs, alpha, eps2));
    hipStream_t s = g->stream;
    const int32_t tv = rs.tv;
    const int32_t T = g->nlive;
    const int32_t nx = T + 2;
    const bool q_dead = rs.q_dead;
    const int32_t target_x = rs.target_x;
    const int32_t n_plain_dead = rs.n_plain_dead;
    const double x0 = 1.0 / (double)g->nv;     // :51-54
    const double oma = 1 - alpha;               // :121
    double *xb = g->xbuf.p;
    double *parts = g->parts.p;
    SgState *st = g->state.p;

    g->used_persistent = false;
    if (g->persist_ok && !g->persist_failed && max_iterations > 0) {
        PersistParams PP{};
        PP.col = g->use16 ? static_cast<const void *>(g->col16.p) : static_cast<const void *>(g->col4.p);
        PP.w2 = reinterpret_cast<const v2d *>(g->w2.p);
        PP.pinfo = g->pinfo.p;
        PP.lane_out = g->lane_out.p;
        PP.n_short = g->n_short;
        PP.lrows = g->lrows.p;
        PP.nlrows = g->nlrows;
        PP.PA = g->PA.p;
        PP.pa_stride = g->pa_stride;
        PP.npieces = g->npieces;
        PP.T = T;
        PP.target_x = target_x;
        PP.n_plain_dead = n_plain_dead;
        PP.q_in_use = (int32_t)q_dead;
        PP.alpha = alpha;
        PP.oma = oma;
        PP.eps2 = eps2;
        PP.x0 = x0;
        PP.max_it = (int32_t)max_iterations;
        PP.barrier = g->barrier.p;
        PP.st = st;
        PP.parts = parts;
        PP.xbuf = xb;
        PP.nblocks = g->persist_blocks;
        PP.dbg = g->dbg.p;
        LOCREC_HIP_TRY(hipMemsetAsync(g->barrier.p, 0, sizeof(unsigned), s));
        const dim3 grid((unsigned)g->persist_blocks), block(512);
        const size_t lds = g->persist_lds;
        LOCREC_TRY(g->prof.begin(s));
#define LOCREC_PERSIST(PWV)                                                                                    \
    do {                                                                                                       \
        if (g->use16) {                                                                                        \
            if (lds > 64 * 1024)                                                                               \
                LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(sg_persistent<PWV, true>),   \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
            hipLaunchKernelGGL((sg_persistent<PWV, true>), grid, block, lds, s, PP);                           \
        } else {                                                                                               \
            if (lds > 64 * 1024)                                                                               \
                LOCREC_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(sg_persistent<PWV, false>),  \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
            hipLaunchKernelGGL((sg_persistent<PWV, false>), grid, block, lds, s, PP);                          \
        }                                                                                                      \
    } while (0)
        if (g->persist_pw == 4) LOCREC_PERSIST(4);
        else LOCREC_PERSIST(12);
#undef LOCREC_PERSIST
        LOCREC_TRY(g->prof.end(s));
        LOCREC_HIP_TRY(hipGetLastError());
        g->persist_units += max_iterations;
        g->used_persistent = true;
        g->target_vertex = tv;
        g->req_max_it = max_iterations;
        g->req_eps2 = eps2;
        g->req_alpha = alpha;
        g->req_vertex = vertex_id;
        g->have_result = true;
        return LOCREC_OK;
    }

    const int sweep_blocks = (g->npieces + 4 * g->ppw - 1) / (4 * g->ppw);
    // isConverged is decided on the device; when epsilon > 0 the host looks at the sticky `done` word
    // after 4, 6, 8, 12, 16 and then every kCheckEvery iterations (the shipped epsilon stops after a
    // handful), through pinned memory that lives with the handle
    int32_t *pinned_done = nullptr;
    const bool poll_by_kernel = g->h_poll_dev != nullptr && !g->no_pack;  // sg_poll at the end of every polled run
    if (poll && max_iterations > 4) pinned_done = poll_by_kernel ? g->h_poll : reinterpret_cast<int32_t *>(g->stage(64));
    // (the request's own values - target slot, dead-vertex counts, alpha, 1 - alpha, eps2 - are in device memory:
    // sg_begin wrote SgReq)
    // the fused form (sg_sweep_fused) where the graph allows it and the request is long enough to need no special cases
    const bool fused = g->use_fused && g->fused_ok && max_iterations >= 3;
    SgFused F{};
    hipStream_t sb = s;  // where the longer rows' reduction and the second sweep run
    if (fused) {
        F.colv = g->use16 ? static_cast<const void *>(g->col16.p) : static_cast<const void *>(g->col4.p);
        F.w2 = reinterpret_cast<const v2d *>(g->w2.p);
        F.pinfo = g->pinfo.p;
        F.seg_fa = g->seg_fa.p;
        F.col2 = g->col2.p;
        F.w2b = reinterpret_cast<const v2d *>(g->w2b.p);
        F.pinfo2 = g->pinfo2.p;
        F.seg2 = g->seg2.p;
        F.lrows = g->lrows_f.p;
        F.PA4 = g->PA4.p;
        F.XL = g->XL.p;
        F.D2W = g->D2W.p;
        F.rq = g->req_dev.p;
        F.st = st;
        F.conv = g->fused_conv.p;
        F.npieces = g->npieces;
        F.npieces2 = g->npieces2;
        F.n_short = g->n_short;
        F.T = T;
        F.nduty = g->nduty;
        F.pa4 = g->pa4;
        F.nlrows = g->nlrows;
        F.n_crows = g->n_crows;
        F.x0 = x0;
        LOCREC_HIP_TRY(hipMemsetAsync(g->fused_conv.p, 0, 2 * sizeof(int32_t), s));
        if (g->stream2 && s != nullptr) {
            sb = g->stream2;
            while (g->fused_events.size() < (size_t)(2 * kMaxGraphRounds + 1)) {
                hipEvent_t e = nullptr;
                LOCREC_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                g->fused_events.push_back(e);
            }
        }
    }
    const bool two_streams = sb != s;
    const int fused_blocks = (std::max(g->npieces, g->nduty) + 3) / 4;
    // position j of a run of the fused form.  Two streams: the main sweep of iteration j on `s`, beside it on `sb` the
    // longer rows of iteration j - 1 and then the second sweep of iteration j; the main sweep waits for the previous
    // second sweep, the longer rows for the previous main sweep.  (Events 2 j + 1 / 2 j + 2: main / second sweep j done;
    // event 0 forks the run, its last second sweep joins it.)
    auto launch_fused = [&](int64_t j, bool starts_request, bool last_of_run) {
        hipEvent_t *ev = g->fused_events.data();
        if (two_streams) {
            if (j == 0) {
                (void)hipEventRecord(ev[0], s);
                (void)hipStreamWaitEvent(sb, ev[0], 0);
            } else {
                (void)hipStreamWaitEvent(sb, ev[2 * j - 1], 0);
            }
        }
        if (!(starts_request && j == 0))
            hipLaunchKernelGGL(sg_fused_long, dim3(kParts), dim3(256), 0, sb, F, (int32_t)j - 1, 0);
        if (g->npieces2 > 0) hipLaunchKernelGGL(sg_fused_k2, dim3((unsigned)((g->npieces2 + 3) / 4)), dim3(256), 0, sb, F, (int32_t)j);
        if (two_streams) {
            (void)hipEventRecord(ev[2 * j + 2], sb);
            if (j > 0) (void)hipStreamWaitEvent(s, ev[2 * j], 0);
        }
        if (g->use16) LOCREC_LAUNCH_PROFILED(g->prof, (sg_sweep_fused<true>), dim3(fused_blocks), dim3(256), 0, s, F, (int32_t)j);
        else LOCREC_LAUNCH_PROFILED(g->prof, (sg_sweep_fused<false>), dim3(fused_blocks), dim3(256), 0, s, F, (int32_t)j);
        if (two_streams) {
            if (last_of_run) (void)hipStreamWaitEvent(s, ev[2 * j + 2], 0);
            else (void)hipEventRecord(ev[2 * j + 1], s);
        }
    };
    auto launch_round = [&](int64_t i) {
        const int par = (int)(i & 1);
        const double *x_in = xb + (size_t)par * nx;
        double *x_out = xb + (size_t)(par ^ 1) * nx;
        const double *parts_prev = parts + (size_t)(par ^ 1) * kParts;
        double *parts_out = parts + (size_t)par * kParts;
        if (sweep_blocks > 0) launch_sweep(g, x_in);  // (timed by its own dispatch when profiling is on)
        hipLaunchKernelGGL(sg_finalize, dim3(kParts), dim3(256), 0, s, g->n_short, g->lrows.p, g->nlrows, g->n_crows, T,
                           g->PA.p, x_in, x_out, g->req_dev.p, parts_prev, parts_out, st, i == 0 ? 1 : 0);
    };
    // Iterations i0 .. i0 + len - 1 (i0 even: the x / block-sum buffers alternate by parity).  A run of at least 4
    // rounds is captured once per (length, starts-the-request) and replayed as ONE hipGraph launch: two launches per
    // ~15 us iteration are otherwise at the mercy of the host's enqueue rate (a slower host measured 47 k instead
    // of 64 k iterations/s at cfg3), and eight graphs on eight streams ask for 100+ k launches per second.
    auto run_rounds_once = [&](int64_t i0, int64_t len, bool with_poll) -> int32_t {
        // (the legacy null stream cannot be captured: a handle moved onto it launches one by one)
        const bool graph_ok = len >= 4 && !g->no_graph && !g->prof.on && (i0 & 1) == 0 && s != nullptr;
        // (fused: the launches of a run depend on i mod 4 - the buffer generation - and on i < 2)
        const int64_t key = (len * 4 + (i0 == 0 ? 1 : 0) + (with_poll ? 2 : 0)) * 2 + (fused ? 1 : 0);
        auto it = graph_ok ? g->round_graphs.find(key) : g->round_graphs.end();
        if (graph_ok && it == g->round_graphs.end()) {
            if (g->round_graphs.size() >= kMaxRoundGraphs) {  // (callers that ask for ever new run lengths)
                LOCREC_HIP_TRY(hipStreamSynchronize(s));
                for (auto &kv : g->round_graphs) (void)hipGraphExecDestroy(kv.second);
                g->round_graphs.clear();
            }
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
                (void)hipGetLastError();  // a stream that refuses capture: the same launches, issued directly, from now on
                g->no_graph = true;
            } else {
                // (only parity and "i == 0" matter; a fused launch carries its position inside the run, the run's first
                // iteration number sits in device memory)
                for (int64_t i = 0; i < len; ++i) {
                    if (fused) launch_fused(i, i0 == 0, i == len - 1);
                    else launch_round(i0 == 0 ? i : i + 2);
                }
                if (fused) hipLaunchKernelGGL(sg_fused_advance, dim3(1), dim3(64), 0, s, g->fused_conv.p, (int32_t)len);
                if (with_poll) hipLaunchKernelGGL(sg_poll, dim3(1), dim3(64), 0, s, st, g->h_poll_dev);
                LOCREC_HIP_TRY(hipStreamEndCapture(s, &graph));
                const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (e != hipSuccess) return fail(LOCREC_E_DEVICE, "hipGraphInstantiate failed: %s", hipGetErrorName(e));
                it = g->round_graphs.emplace(key, exec).first;
            }
        }
        if (it == g->round_graphs.end()) {
            for (int64_t i = i0; i < i0 + len; ++i) {
                if (fused) launch_fused(i - i0, i0 == 0, i == i0 + len - 1);
                else launch_round(i);
            }
            if (fused) hipLaunchKernelGGL(sg_fused_advance, dim3(1), dim3(64), 0, s, g->fused_conv.p, (int32_t)len);
            if (with_poll) hipLaunchKernelGGL(sg_poll, dim3(1), dim3(64), 0, s, st, g->h_poll_dev);
            return LOCREC_OK;
        }
        LOCREC_HIP_TRY(hipGraphLaunch(it->second, s));
        return LOCREC_OK;
    };
    // (a long run is replayed in stretches of kMaxGraphRounds: the graphs stay small, the parity of i0 is kept)
    auto run_rounds = [&](int64_t i0, int64_t len, bool with_poll) -> int32_t {
        while (len > 0) {
            const int64_t part = std::min<int64_t>(len, kMaxGraphRounds);
            LOCREC_TRY(run_rounds_once(i0, part, with_poll && part == len));
            i0 += part;
            len -= part;
        }
        return LOCREC_OK;
    };
    int64_t next_check = 4;
    int32_t status = LOCREC_OK;
    int64_t fused_iterations_enqueued = 0;
    for (int64_t i = 0; i < max_iterations;) {
        const int64_t stop = pinned_done ? std::min(max_iterations, next_check) : max_iterations;
        const bool look = pinned_done && stop == next_check && stop < max_iterations;  // the host looks after this run
        if ((status = run_rounds(i, stop - i, look && poll_by_kernel)) != LOCREC_OK) break;
        i = stop;
        fused_iterations_enqueued = i;
        if (look) {
            if ((!poll_by_kernel &&
                 hipMemcpyAsync(pinned_done, &st->done, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess) ||
                hipStreamSynchronize(s) != hipSuccess) {
                status = fail(LOCREC_E_DEVICE, "convergence poll failed");
                break;
            }
            if (*pinned_done) break;
            next_check += next_check < 8 ? 2 : (next_check < 16 ? 4 : kCheckEvery);
        }
    }
    if (status != LOCREC_OK) return status;
    if (fused) {
        // the longer rows and the duty pass of the last iteration, then step()'s exit decided on the device: x, sweep count and isConverged's
        // sum land where locrec_sg_fetch reads them
        const int32_t ran = (int32_t)fused_iterations_enqueued;
        hipLaunchKernelGGL(sg_fused_long, dim3(kParts), dim3(256), 0, s, F, ran - 1, 1);
        hipLaunchKernelGGL(sg_fused_tail, dim3((unsigned)((g->nduty + 3) / 4)), dim3(256), 0, s, F, ran - 1);
        hipLaunchKernelGGL(sg_fused_result, dim3((unsigned)std::min(64, (nx + 255) / 256)), dim3(256), 0, s, F, ran, xb, parts);
    }
    LOCREC_HIP_TRY(hipGetLastError());
    g->target_vertex = tv;
    g->req_max_it = max_iterations;
    g->req_eps2 = eps2;
    g->have_result = true;
    return LOCREC_OK;
}

}  // namespace

extern "C" int32_t locrec_sg_iterate_async(locrec_sg_graph *g, int64_t vertex_id, double alpha,
                                           double epsilon, int64_t max_iterations) try
{
    if (!g) return fail(LOCREC_E_INVALID_ARG, "graph is NULL");
    g->have_result = false;
    // require()s of the constructor, StochasticRecommender.scala:33-34
    if (!(epsilon >= 0)) return fail(LOCREC_E_INVALID_ARG, "requirement failed: epsilon must be non-negative");
    if (max_iterations < 0)
        return fail(LOCREC_E_INVALID_ARG, "requirement failed: max iterations number must be non-negative");
    return enqueue_iterations(g, vertex_id, alpha, epsilon * epsilon /* :40 */, max_iterations, epsilon > 0);
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_sweeps_async(locrec_sg_graph *g, int64_t vertex_id, double alpha, int64_t sweeps) try
{
    if (!g) return fail(LOCREC_E_INVALID_ARG, "graph is NULL");
    if (sweeps < 0) return fail(LOCREC_E_INVALID_ARG, "sweeps must be non-negative");
    return enqueue_iterations(g, vertex_id, alpha, -1.0, sweeps, false);
} LOCREC_CATCH_ALL

// ---- a group of independent graphs iterated together (sg_sweep_group / sg_finalize_group) ----

// *out = 1 when every graph of the group has observed its convergence (one thread)
__global__ void sg_group_all_done(const SgGraphView *__restrict__ G, int32_t ngraphs, int32_t *out)
{
    int all = 1;
    for (int i = 0; i < ngraphs; ++i) all &= G[i].st->done != 0 ? 1 : 0;
    *out = all;
}

struct locrec_sg_group {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<locrec_sg_graph *> graphs;  // not owned
    std::vector<SgGraphView> host;
    DevBuf<SgGraphView> dev;
    std::vector<SgBegin> begin_host;  // the graphs' rows of sg_begin_group
    DevBuf<SgBegin> begin_dev;
    int32_t total_waves = 0;
    bool use16 = false;
    hipEvent_t done = nullptr;  // end of the last enqueued rounds: the graphs' own streams wait for it
    DevBuf<int32_t> all_done;   // sg_group_all_done's answer
    int32_t *h_all_done = nullptr;  // ... in pinned host memory
    ~locrec_sg_group()
    {
        if (done) (void)hipEventDestroy(done);
        if (h_all_done) (void)hipHostFree(h_all_done);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

extern "C" int32_t locrec_sg_group_create(locrec_sg_graph *const *graphs, int32_t n_graphs, locrec_sg_group **out) try
{
    if (!out) return fail(LOCREC_E_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!graphs || n_graphs <= 0 || n_graphs > 65535) return fail(LOCREC_E_INVALID_ARG, "a group needs 1 .. 65535 graphs");
    auto grp = std::make_unique<locrec_sg_group>();
    for (int32_t i = 0; i < n_graphs; ++i) {
        locrec_sg_graph *g = graphs[i];
        if (!g) return fail(LOCREC_E_INVALID_ARG, "graph %d is NULL", i);
        for (int32_t j = 0; j < i; ++j)
            if (graphs[j] == g) return fail(LOCREC_E_INVALID_ARG, "graph %d appears twice in the group", i);
        if (g->shard_count != 1) return fail(LOCREC_E_INVALID_ARG, "graph %d is a shard: it is iterated with locrec_sg_shard_*", i);
        if (g->ppw != 1 || g->gs_blocks > 0)
            return fail(LOCREC_E_INVALID_ARG, "graph %d was created with a non-default sweep form (LOCREC_SG_PPW / LOCREC_SG_GS)", i);
        if (i == 0) {
            grp->device = g->device;
            grp->use16 = g->use16;
        } else if (g->device != grp->device) {
            return fail(LOCREC_E_INVALID_ARG, "graph %d lives on another device", i);
        } else if (g->use16 != grp->use16) {
            return fail(LOCREC_E_INVALID_ARG, "graph %d stores its columns in another width than graph 0", i);
        }
        grp->graphs.push_back(g);
    }
    LOCREC_HIP_TRY(hipSetDevice(grp->device));
    LOCREC_HIP_TRY(hipStreamCreateWithFlags(&grp->stream, hipStreamNonBlocking));
    LOCREC_HIP_TRY(hipEventCreateWithFlags(&grp->done, hipEventDisableTiming));
    grp->host.resize((size_t)n_graphs);
    LOCREC_TRY(grp->dev.alloc((size_t)n_graphs));
    grp->begin_host.resize((size_t)n_graphs);
    LOCREC_TRY(grp->begin_dev.alloc((size_t)n_graphs));
    LOCREC_TRY(grp->all_done.alloc(1));
    {
        void *hp = nullptr;
        LOCREC_HIP_TRY(hipHostMalloc(&hp, sizeof(int32_t), hipHostMallocDefault));
        grp->h_all_done = static_cast<int32_t *>(hp);
    }
    *out = grp.release();
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" void locrec_sg_group_destroy(locrec_sg_group *grp)
{
    if (!grp) return;
    (void)hipSetDevice(grp->device);
    if (grp->stream) (void)hipStreamSynchronize(grp->stream);
    delete grp;
}

extern "C" int32_t locrec_sg_group_synchronize(locrec_sg_group *grp) try
{
    if (!grp) return fail(LOCREC_E_INVALID_ARG, "group is NULL");
    LOCREC_HIP_TRY(hipSetDevice(grp->device));
    LOCREC_HIP_TRY(hipStreamSynchronize(grp->stream));
    return LOCREC_OK;
} LOCREC_CATCH_ALL

// Up to `sweeps` rounds of calcNextX (StochasticRecommender.scala:108-128) on every graph of the group, graph i
// from vertex_ids[i]: two launches per round for all graphs.  eps2 < 0: exactly `sweeps` rounds; eps2 >= 0: step()'s
// isConverged (:92-106, :130-141) per graph - a converged graph's sticky `done` word turns its share of the
// remaining rounds into no-ops, exactly as in the single-graph loop.  Each graph's result is then read with
// locrec_sg_fetch, as after locrec_sg_sweeps_async / locrec_sg_iterate_async.
static int32_t group_run(locrec_sg_group *grp, const int64_t *vertex_ids, double alpha, double eps2, int64_t sweeps)
{
    if (sweeps > INT32_MAX) sweeps = INT32_MAX;
    LOCREC_HIP_TRY(hipSetDevice(grp->device));
    hipStream_t s = grp->stream;
    const int32_t n = (int32_t)grp->graphs.size();
    const double oma = 1 - alpha;  // :121
    int32_t waves = 0;
    std::vector<RequestSetup> setups((size_t)n);
    for (int32_t i = 0; i < n; ++i) {
        locrec_sg_graph *g = grp->graphs[(size_t)i];
        g->have_result = false;
        g->shard_active = false;
        LOCREC_HIP_TRY(hipStreamSynchronize(g->stream));  // whatever the graph was doing on its own stream is over
        // the request's set-up (x0, state reset, the vertex's out-edge slots): this graph's row of ONE sg_begin_group launch
        const int32_t st_rc = begin_request(g, vertex_ids[i], &setups[(size_t)i], alpha, eps2, &grp->begin_host[(size_t)i]);
        if (st_rc != LOCREC_OK) {
            // graphs 0 .. i-1 already count their slots as re-pointed: do that before reporting the failure
            if (i > 0) {
                LOCREC_HIP_TRY(hipMemcpyAsync(grp->begin_dev.p, grp->begin_host.data(), (size_t)i * sizeof(SgBegin), hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(sg_begin_group, dim3(kBeginBlocks, (unsigned)i), dim3(256), 0, s, grp->begin_dev.p);
                LOCREC_HIP_TRY(hipStreamSynchronize(s));
            }
            return st_rc;
        }
        const RequestSetup &rs = setups[(size_t)i];
        SgGraphView &v = grp->host[(size_t)i];
        v.colv = g->use16 ? static_cast<const void *>(g->col16.p) : static_cast<const void *>(g->col4.p);
        v.w2 = reinterpret_cast<const v2d *>(g->w2.p);
        v.pinfo = g->pinfo.p;
        v.seg_out = g->seg_out.p;
        v.xbuf = g->xbuf.p;
        v.partial = g->PA.p;
        v.parts = g->parts.p;
        v.st = g->state.p;
        v.lrows = g->lrows.p;
        v.npieces = g->npieces;
        v.nx = g->nlive + 2;
        v.wave_base = waves;
        v.n_short = g->n_short;
        v.nlrows = g->nlrows;
        v.n_crows = g->n_crows;
        v.nlive = g->nlive;
        v.target_x = rs.target_x;
        v.n_plain_dead = rs.n_plain_dead;
        v.q_in_use = rs.q_dead ? 1 : 0;
        v.alpha = alpha;
        v.oma = oma;
        v.eps2 = eps2;  // (< 0: a fixed number of sweeps, isConverged never fires)
        waves += g->npieces;
    }
    grp->total_waves = waves;
    LOCREC_HIP_TRY(hipMemcpyAsync(grp->dev.p, grp->host.data(), (size_t)n * sizeof(SgGraphView), hipMemcpyHostToDevice, s));
    LOCREC_HIP_TRY(hipMemcpyAsync(grp->begin_dev.p, grp->begin_host.data(), (size_t)n * sizeof(SgBegin), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(sg_begin_group, dim3(kBeginBlocks, (unsigned)n), dim3(256), 0, s, grp->begin_dev.p);
    const dim3 sweep_grid((unsigned)std::max(1, (waves + 3) / 4)), fin_grid(kParts, (unsigned)n);
    auto launch_round = [&](int64_t i) {
        const int par = (int)(i & 1);
        if (waves > 0) {
            if (grp->use16) hipLaunchKernelGGL((sg_sweep_group<true>), sweep_grid, dim3(256), 0, s, grp->dev.p, n, par);
            else hipLaunchKernelGGL((sg_sweep_group<false>), sweep_grid, dim3(256), 0, s, grp->dev.p, n, par);
        }
        hipLaunchKernelGGL(sg_finalize_group, fin_grid, dim3(256), 0, s, grp->dev.p, par, i == 0 ? 1 : 0);
    };
    // (Launched one by one on purpose: replaying the rounds of a group as hipGraphs - as enqueue_iterations does for a
    // single graph - measured SLOWER, 1.14 M -> 0.79 M graph-iterations/s on 16 graphs of 36 k edges and 2.21 M ->
    // 2.14 M on 64: two launches per round never starve the stream, and the graph's kernel nodes start further apart
    // than back-to-back launches do.)
    auto run_rounds = [&](int64_t i0, int64_t len) -> int32_t {
        for (int64_t i = i0; i < i0 + len; ++i) launch_round(i);
        return LOCREC_OK;
    };
    int64_t next_check = 4;
    for (int64_t i = 0; i < sweeps;) {
        const int64_t stop = eps2 >= 0 ? std::min(sweeps, next_check) : sweeps;
        LOCREC_TRY(run_rounds(i, stop - i));
        i = stop;
        // with an epsilon, look now and then whether EVERY graph has converged (the shipped epsilon stops after a
        // handful of rounds): the rest of the rounds would be empty launches
        if (eps2 >= 0 && i == next_check && i < sweeps) {
            hipLaunchKernelGGL(sg_group_all_done, dim3(1), dim3(1), 0, s, grp->dev.p, n, grp->all_done.p);
            LOCREC_HIP_TRY(hipMemcpyAsync(grp->h_all_done, grp->all_done.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            LOCREC_HIP_TRY(hipStreamSynchronize(s));
            if (*grp->h_all_done) break;
            next_check += next_check < 16 ? 4 : kCheckEvery;
        }
    }
    LOCREC_HIP_TRY(hipGetLastError());
    // a fetch on a graph synchronises ITS stream: make that stream wait for the group's rounds
    LOCREC_HIP_TRY(hipEventRecord(grp->done, s));
    for (int32_t i = 0; i < n; ++i) {
        locrec_sg_graph *g = grp->graphs[(size_t)i];
        LOCREC_HIP_TRY(hipStreamWaitEvent(g->stream, grp->done, 0));
        g->used_persistent = false;
        g->target_vertex = setups[(size_t)i].tv;
        g->req_max_it = sweeps;
        g->req_eps2 = eps2;
        g->have_result = true;
    }
    return LOCREC_OK;
}

extern "C" int32_t locrec_sg_group_sweeps_async(locrec_sg_group *grp, const int64_t *vertex_ids, double alpha, int64_t sweeps) try
{
    if (!grp || !vertex_ids) return fail(LOCREC_E_INVALID_ARG, "NULL argument");
    if (sweeps < 0) return fail(LOCREC_E_INVALID_ARG, "sweeps must be non-negative");
    return group_run(grp, vertex_ids, alpha, -1.0, sweeps);
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_group_iterate_async(locrec_sg_group *grp, const int64_t *vertex_ids, double alpha, double epsilon,
                                                 int64_t max_iterations) try
{
    if (!grp || !vertex_ids) return fail(LOCREC_E_INVALID_ARG, "NULL argument");
    // require()s of the constructor, StochasticRecommender.scala:33-34
    if (!(epsilon >= 0)) return fail(LOCREC_E_INVALID_ARG, "requirement failed: epsilon must be non-negative");
    if (max_iterations < 0)
        return fail(LOCREC_E_INVALID_ARG, "requirement failed: max iterations number must be non-negative");
    return group_run(grp, vertex_ids, alpha, epsilon * epsilon /* :40 */, max_iterations);
} LOCREC_CATCH_ALL

// ---- row-sharded iteration, driven step by step by the host (which owns the all-reduce) ----

extern "C" int32_t locrec_sg_live_count(const locrec_sg_graph *g, int64_t *out_live) try
{
    if (!g || !out_live) return fail(LOCREC_E_INVALID_ARG, "NULL argument");
    *out_live = g->nlive;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_shard_begin(locrec_sg_graph *g, int64_t vertex_id) try
{
    if (!g) return fail(LOCREC_E_INVALID_ARG, "graph is NULL");
    g->have_result = false;
    g->shard_active = false;
    RequestSetup rs{};
    LOCREC_TRY(begin_request(g, vertex_id, &rs));
    g->target_vertex = rs.tv;
    g->shard_target_x = rs.target_x;
    g->shard_n_plain_dead = rs.n_plain_dead;
    g->shard_q_dead = rs.q_dead ? 1 : 0;
    g->shard_it = 0;
    g->shard_done = false;
    g->shard_active = true;
    g->used_persistent = false;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_shard_sigma(locrec_sg_graph *g, double *sigma_dev) try
{
    if (!g || !sigma_dev) return fail(LOCREC_E_INVALID_ARG, "NULL argument");
    if (!g->shard_active) return fail(LOCREC_E_INVALID_ARG, "locrec_sg_shard_begin has not been called");
    LOCREC_HIP_TRY(hipSetDevice(g->device));
    hipStream_t s = g->stream;
    const int32_t nx = g->nlive + 2;
    const double *x_in = g->xbuf.p + (size_t)(g->shard_it & 1) * nx;
    launch_sweep(g, x_in);
    hipLaunchKernelGGL(sg_sigma, dim3(kParts), dim3(256), 0, s, g->n_short, g->lrows.p, g->nlrows, g->PA.p, sigma_dev);
    LOCREC_HIP_TRY(hipGetLastError());
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_shard_apply(locrec_sg_graph *g, const double *sigma_dev, double alpha) try
{
    if (!g || !sigma_dev) return fail(LOCREC_E_INVALID_ARG, "NULL argument");
    if (!g->shard_active) return fail(LOCREC_E_INVALID_ARG, "locrec_sg_shard_begin has not been called");
    LOCREC_HIP_TRY(hipSetDevice(g->device));
    hipStream_t s = g->stream;
    const int32_t nx = g->nlive + 2;
    const int par = (int)(g->shard_it & 1);
    const double *x_in = g->xbuf.p + (size_t)par * nx;
    double *x_out = g->xbuf.p + (size_t)(par ^ 1) * nx;
    hipLaunchKernelGGL(sg_apply_sigma, dim3(kParts), dim3(256), 0, s, g->nlive, sigma_dev, x_in, x_out,
                       g->shard_target_x, g->shard_n_plain_dead, g->shard_q_dead, alpha, 1 - alpha,
                       g->parts.p + (size_t)par * kParts, g->state.p);
    LOCREC_HIP_TRY(hipGetLastError());
    ++g->shard_it;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

// isConverged's sum (:130-141) of the last applied sweep; synchronises the stream
extern "C" int32_t locrec_sg_shard_d2(locrec_sg_graph *g, double *out_d2) try
{
    if (!g || !out_d2) return fail(LOCREC_E_INVALID_ARG, "NULL argument");
    if (!g->shard_active || g->shard_it == 0) return fail(LOCREC_E_INVALID_ARG, "no sweep has been applied");
    LOCREC_HIP_TRY(hipSetDevice(g->device));
    double parts[kParts];
    LOCREC_HIP_TRY(hipMemcpyAsync(parts, g->parts.p + (size_t)((g->shard_it - 1) & 1) * kParts, sizeof parts,
                                  hipMemcpyDeviceToHost, g->stream));
    LOCREC_HIP_TRY(hipStreamSynchronize(g->stream));
    *out_d2 = host_total_d2(parts);
    return LOCREC_OK;
} LOCREC_CATCH_ALL

// records what step() (:92-106) decided, so that locrec_sg_fetch() can report it
extern "C" int32_t locrec_sg_shard_finish(locrec_sg_graph *g, int64_t iterations, int32_t converged) try
{
    if (!g) return fail(LOCREC_E_INVALID_ARG, "graph is NULL");
    if (!g->shard_active) return fail(LOCREC_E_INVALID_ARG, "locrec_sg_shard_begin has not been called");
    g->shard_iterations = iterations;
    g->shard_converged = converged;
    g->shard_done = true;
    g->have_result = true;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_fetch(locrec_sg_graph *g, int64_t *out_ids, double *out_probs,
                                   int64_t *inout_count, int64_t *out_iterations, int32_t *out_converged) try
{
    if (!g) return fail(LOCREC_E_INVALID_ARG, "graph is NULL");
    if (!g->have_result) return fail(LOCREC_E_INVALID_ARG, "no iteration has been enqueued");
    if (!inout_count) return fail(LOCREC_E_INVALID_ARG, "inout_count is NULL");
    LOCREC_HIP_TRY(hipSetDevice(g->device));
    hipStream_t s = g->stream;
    SgState st{};
    std::vector<double> parts(2 * kParts);
    // small x (both parities fit the staging buffer): state, block sums and x in ONE round trip
    const size_t nx_all = (size_t)(g->nlive + 2);
    const bool one_trip = nx_all <= 65536;
    unsigned char *stg = one_trip ? g->stage(64 + 2 * kParts * 8 + 2 * nx_all * 8) : nullptr;
    const double *x_both = nullptr;
    const double *x_packed = nullptr;  // x of the current parity only (sg_pack_result)
    void *stg_dev = nullptr;
    const bool packed = stg && !g->no_pack && !g->used_persistent && hipHostGetDevicePointer(&stg_dev, stg, 0) == hipSuccess;
    if (packed) {
        // one gather launch writes state, block sums and the current x into the pinned buffer (sg_pack_result)
        const int32_t nx1 = g->nlive + 2;
        hipLaunchKernelGGL(sg_pack_result, dim3((unsigned)std::min(16, (nx1 + 2047) / 2048)), dim3(256), 0, s, g->state.p,
                           g->parts.p, g->xbuf.p, nx1, static_cast<unsigned char *>(stg_dev));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
        st = *reinterpret_cast<SgState *>(stg);
        const double *pparts = reinterpret_cast<const double *>(stg + 64);
        std::copy(pparts, pparts + 2 * kParts, parts.begin());
        x_packed = pparts + 2 * kParts;
    } else if (stg) {
        (void)hipGetLastError();
        SgState *pst = reinterpret_cast<SgState *>(stg);
        double *pparts = reinterpret_cast<double *>(stg + 64);
        double *px = pparts + 2 * kParts;
        LOCREC_HIP_TRY(hipMemcpyAsync(pst, g->state.p, sizeof st, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(pparts, g->parts.p, 2 * kParts * sizeof(double), hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(px, g->xbuf.p, 2 * nx_all * sizeof(double), hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
        st = *pst;
        std::copy(pparts, pparts + 2 * kParts, parts.begin());
        x_both = px;
    } else {
        LOCREC_HIP_TRY(hipMemcpyAsync(&st, g->state.p, sizeof st, hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipMemcpyAsync(parts.data(), g->parts.p, parts.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
    }
    if (g->used_persistent && g->dbg.p) {
        unsigned long long tk[3] = {0, 0, 0};
        LOCREC_HIP_TRY(hipMemcpy(tk, g->dbg.p, sizeof tk, hipMemcpyDeviceToHost));
        fprintf(stderr, "[locrec sg persistent] sweeps %d: compute %.2f us, barrier %.2f us, rebuild %.2f us per sweep\n",
                st.sweeps, tk[0] / 100.0 / std::max(1, st.sweeps), tk[1] / 100.0 / std::max(1, st.sweeps),
                tk[2] / 100.0 / std::max(1, st.sweeps));
    }
    if (g->used_persistent && st.done < 0) {
        // the persistent launch could not gather all of its blocks in time: rerun in streaming form
        g->persist_failed = true;
        LOCREC_TRY(enqueue_iterations(g, g->req_vertex, g->req_alpha, g->req_eps2, g->req_max_it, g->req_eps2 > 0));
        return locrec_sg_fetch(g, out_ids, out_probs, inout_count, out_iterations, out_converged);
    }
    const int64_t sweeps = st.sweeps;
    const int32_t T = g->nlive;
    const int32_t nx = T + 2;
    std::vector<double> x_own;
    const double *x = nullptr;
    if (x_packed) {
        x = x_packed;
    } else if (x_both && !(g->used_persistent && st.done < 0)) {
        x = x_both + (size_t)(sweeps & 1) * nx;
    } else {
        x_own.resize((size_t)nx);
        LOCREC_HIP_TRY(hipMemcpyAsync(x_own.data(), g->xbuf.p + (size_t)(sweeps & 1) * nx, (size_t)nx * sizeof(double),
                                      hipMemcpyDeviceToHost, s));
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
        x = x_own.data();
    }
    // step(), :92-106: which of the two exits was taken
    int32_t converged = 0;
    int64_t iterations = g->req_max_it;
    if (g->shard_active) {
        if (!g->shard_done) return fail(LOCREC_E_INVALID_ARG, "locrec_sg_shard_finish has not been called");
        converged = g->shard_converged;
        iterations = g->shard_iterations;
    } else if (sweeps > 0) {
        const double d2 = host_total_d2(parts.data() + (size_t)((sweeps - 1) & 1) * kParts);
        if (d2 <= g->req_eps2) {
            converged = 1;
            iterations = sweeps - 1;
        }
    }
    if (!g->shard_active && !converged && sweeps != g->req_max_it)
        return fail(LOCREC_E_DEVICE, "internal: %lld sweeps executed, %lld expected",
                    (long long)sweeps, (long long)g->req_max_it);
    // :84-88  id != vertexId and probability > 0, ascending id.  A source-only vertex holds the
    // shared value x[D] (1/V before the first sweep, 0 afterwards).
    const int64_t cap = *inout_count;
    const double xdead = x[T];
    int64_t n = 0;
    auto emit = [&](int64_t v, double xv) {
        if (v == g->target_vertex || !(xv > 0)) return;
        if (n < cap) {
            if (out_ids) out_ids[n] = g->vid[v];
            if (out_probs) out_probs[n] = xv;
        }
        ++n;
    };
    if (xdead > 0) {  // before the first sweep every vertex still holds x0
        for (int64_t v = 0; v < g->nv; ++v) {
            const int32_t l = g->live_of[v];
            emit(v, l >= 0 ? x[l] : xdead);
        }
    } else {
        // after a sweep the source-only vertices are all 0: only the live ones can appear (at cfg3
        // 10 k of 290 k vertices), in ascending id order
        if (g->live_sorted.size() != (size_t)T) {
            g->live_sorted.clear();
            for (int64_t v = 0; v < g->nv; ++v)
                if (g->live_of[v] >= 0) g->live_sorted.push_back((int32_t)v);
        }
        for (const int32_t v : g->live_sorted) emit(v, x[g->live_of[v]]);
    }
    *inout_count = n;
    if (out_iterations) *out_iterations = iterations;
    if (out_converged) *out_converged = converged;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_recommend(locrec_sg_graph *g, int64_t vertex_id, double alpha,
                                       double epsilon, int64_t max_iterations, int64_t *out_ids,
                                       double *out_probs, int64_t *inout_count,
                                       int64_t *out_iterations, int32_t *out_converged) try
{
    LOCREC_TRY(locrec_sg_iterate_async(g, vertex_id, alpha, epsilon, max_iterations));
    return locrec_sg_fetch(g, out_ids, out_probs, inout_count, out_iterations, out_converged);
} LOCREC_CATCH_ALL
