// multi.hip -- several GPUs of one node driven from ONE process, inside the library (include/locrec.h, "Several
// devices in one process").  The Scala host the reference prescribes is one JVM:
// it cannot start one process per GPU and has no torch.distributed, so the multi-GPU forms that bench.py drives with
// one rank per GPU exist here a second time as plain C entry points over per-device streams, events and peer access.
//
//   locrec_knn_replicas_*   KNN all-pairs / batches: every device holds the whole candidate set (3.9 GB at cfg4 - nothing
//                           next to 288 GB) and takes a contiguous share of the QUERIES; no communication per query.
//                           Set-up is the block all-gather north_star names, tile-wise and overlapped: device r uploads
//                           tile r of every CSR array over PCIe, and as soon as a tile has landed the other devices
//                           pull it over xGMI (hipMemcpyPeerAsync on their own streams, ordered by the tile's event)
//                           while the later tiles are still uploading; then each device builds its index from its
//                           DEVICE arrays (locrec_knn_create_from_device) on a thread of its own.
//   locrec_sg_sharded_*     ONE graph over the devices (BASELINE.json configs[4]): the C-ABI shard protocol
//                           (locrec_sg_shard_begin / sigma / apply / d2 / finish) driven from here, the exchange of the
//                           T live entries of sigma done by a kernel that READS THE PEERS' buffers directly (peer
//                           access: 80 KB per sweep at cfg3; staged hipMemcpyPeerAsync copies where peer access is
//                           not available), ordered across devices by events only - no host hop per sweep, no
//                           collective library.  rows of P^T sharded -> an all-gather of owned entries, bit-identical to
//                           one GPU; rows of P sharded -> an all-reduce whose additions run in DEVICE ORDER on every
//                           device (deterministic, unlike a ring).
// Host-side composition of the public C ABI: nothing here reaches into a handle.
#include "common.h"

#include <algorithm>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>
#include <vector>

using namespace locrec;

namespace {

constexpr int kMaxDevices = 16;

std::mutex g_dev_mu;
std::vector<int32_t> g_devices;  // locrec_set_devices

int32_t resolve_devices(int32_t n, const int32_t *ids, std::vector<int32_t> &out)
{
    out.clear();
    if (n > 0 && ids) {
        out.assign(ids, ids + n);
    } else {
        std::lock_guard<std::mutex> g(g_dev_mu);
        out = g_devices;
    }
    if (out.empty()) {
        int cur = 0;
        LOCREC_TRY(ensure_device());
        LOCREC_HIP_TRY(hipGetDevice(&cur));
        out.push_back(cur);
    }
    if ((int)out.size() > kMaxDevices) return fail(LOCREC_E_INVALID_ARG, "at most %d devices", kMaxDevices);
    int count = 0;
    LOCREC_TRY(ensure_device());
    LOCREC_HIP_TRY(hipGetDeviceCount(&count));
    for (int32_t d : out)
        if (d < 0 || d >= count) return fail(LOCREC_E_INVALID_ARG, "device %d is not one of the %d visible devices", d, count);
    return LOCREC_OK;
}

// peer access from every listed device to every other DISTINCT one; false if some pair cannot (the callers then stage
// copies instead).  The same physical device listed twice (the one-GPU rehearsal) needs nothing.
bool enable_peer_access(const std::vector<int32_t> &devs)
{
    bool all = true;
    for (int32_t a : devs)
        for (int32_t b : devs) {
            if (a == b) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) {
                all = false;
                continue;
            }
            if (hipSetDevice(a) != hipSuccess) return false;
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) all = false;
            (void)hipGetLastError();
        }
    return all;
}

struct PerDevice {
    int32_t device = 0;
    hipStream_t stream = nullptr;
    ~PerDevice()
    {
        if (stream) {
            (void)hipSetDevice(device);
            (void)hipStreamDestroy(stream);
        }
    }
};

// runs fn(r) for r = 0 .. n-1 on one thread each (the calls block: index builds, batched requests); the first failure's
// status and message are handed to the calling thread
template <class F>
int32_t parallel_over(int n, F fn)
{
    std::vector<int32_t> st((size_t)n, LOCREC_OK);
    std::vector<std::string> msg((size_t)n);
    std::vector<std::thread> th;
    for (int r = 0; r < n; ++r)
        th.emplace_back([&, r] {
            try {
                st[(size_t)r] = fn(r);
            } catch (...) {
                st[(size_t)r] = status_of_current_exception();
            }
            if (st[(size_t)r] != LOCREC_OK) msg[(size_t)r] = last_error_ref();  // (thread-local: copy it out)
        });
    for (auto &t : th) t.join();
    for (int r = 0; r < n; ++r)
        if (st[(size_t)r] != LOCREC_OK) {
            last_error_ref() = msg[(size_t)r];
            return st[(size_t)r];
        }
    return LOCREC_OK;
}

}  // namespace

extern "C" int32_t locrec_set_devices(int32_t n_devices, const int32_t *device_ids) try
{
    std::vector<int32_t> devs;
    if (n_devices < 0 || (n_devices > 0 && !device_ids)) return fail(LOCREC_E_INVALID_ARG, "bad device list");
    if (n_devices > 0) LOCREC_TRY(resolve_devices(n_devices, device_ids, devs));
    std::lock_guard<std::mutex> g(g_dev_mu);
    g_devices = devs;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

// =====================================================================================================
// KNN: one replica of the index per device, queries sharded

struct locrec_knn_replicas {
    std::vector<int32_t> devices;
    std::vector<locrec_knn_index *> ix;
    ~locrec_knn_replicas()
    {
        for (size_t r = 0; r < ix.size(); ++r)
            if (ix[r]) {
                (void)hipSetDevice(devices[r]);
                (void)locrec_knn_destroy(ix[r]);
            }
    }
};

namespace {

// one CSR array of the set-up: a full-size buffer on every device, filled tile by tile
struct GatherArray {
    const void *host = nullptr;
    size_t bytes = 0;
    std::vector<void *> dev;  // [device]
};

}  // namespace

extern "C" int32_t locrec_knn_replicas_create(
    int32_t n_devices, const int32_t *device_ids, int64_t n, const int64_t *person_ids,
    const int64_t *p_rowptr, const int32_t *p_idx, const double *p_val, int32_t p_dim,
    const int64_t *c_rowptr, const int32_t *c_idx, const double *c_val, int32_t c_dim,
    const int64_t *r_rowptr, const int64_t *r_place, const int64_t *r_rating, locrec_knn_replicas **out) try
{
    if (!out) return fail(LOCREC_E_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (n < 0 || (n > 0 && (!person_ids || !p_rowptr || !c_rowptr))) return fail(LOCREC_E_INVALID_ARG, "NULL input array");
    std::unique_ptr<locrec_knn_replicas> h(new locrec_knn_replicas);
    LOCREC_TRY(resolve_devices(n_devices, device_ids, h->devices));
    const int D = (int)h->devices.size();
    h->ix.assign((size_t)D, nullptr);
    if (D == 1) {  // nothing to gather
        LOCREC_HIP_TRY(hipSetDevice(h->devices[0]));
        LOCREC_TRY(locrec_knn_create(n, person_ids, p_rowptr, p_idx, p_val, p_dim, c_rowptr, c_idx, c_val, c_dim, r_rowptr, r_place,
                                     r_rating, &h->ix[0]));
        *out = h.release();
        return LOCREC_OK;
    }
    (void)enable_peer_access(h->devices);  // (hipMemcpyPeerAsync works without it, through the host, only slower)
    if (n > 0 && (p_rowptr[0] != 0 || c_rowptr[0] != 0 || (r_rowptr && r_rowptr[0] != 0)))
        return fail(LOCREC_E_INVALID_ARG, "row pointers must start at 0");
    const int64_t pe = n > 0 ? p_rowptr[n] : 0, ce = n > 0 ? c_rowptr[n] : 0, re = (n > 0 && r_rowptr) ? r_rowptr[n] : 0;
    if (pe < 0 || ce < 0 || re < 0) return fail(LOCREC_E_INVALID_ARG, "negative element count");
    if (pe > 0 && (!p_idx || !p_val)) return fail(LOCREC_E_INVALID_ARG, "NULL place arrays");
    if (ce > 0 && (!c_idx || !c_val)) return fail(LOCREC_E_INVALID_ARG, "NULL category arrays");
    if (re > 0 && (!r_place || !r_rating)) return fail(LOCREC_E_INVALID_ARG, "NULL ratings arrays");
    std::vector<GatherArray> arrays = {
        {person_ids, (size_t)n * 8, {}},       {p_rowptr, (size_t)(n + 1) * 8, {}}, {p_idx, (size_t)pe * 4, {}},
        {p_val, (size_t)pe * 8, {}},           {c_rowptr, (size_t)(n + 1) * 8, {}}, {c_idx, (size_t)ce * 4, {}},
        {c_val, (size_t)ce * 8, {}},           {r_rowptr, r_rowptr ? (size_t)(n + 1) * 8 : 0, {}},
        {r_place, (size_t)re * 8, {}},         {r_rating, (size_t)re * 8, {}}};
    std::vector<PerDevice> pd((size_t)D);
    struct Freer {  // the gathered input arrays are only needed until the indices are built
        std::vector<GatherArray> *a;
        std::vector<int32_t> *devs;
        ~Freer()
        {
            for (auto &g : *a)
                for (size_t r = 0; r < g.dev.size(); ++r)
                    if (g.dev[r]) {
                        (void)hipSetDevice((*devs)[r]);
                        (void)hipFree(g.dev[r]);
                    }
        }
    } freer{&arrays, &h->devices};
    for (int r = 0; r < D; ++r) {
        pd[(size_t)r].device = h->devices[(size_t)r];
        LOCREC_HIP_TRY(hipSetDevice(pd[(size_t)r].device));
        LOCREC_HIP_TRY(hipStreamCreateWithFlags(&pd[(size_t)r].stream, hipStreamNonBlocking));
        for (auto &g : arrays) {
            g.dev.resize((size_t)D, nullptr);
            if (g.bytes) LOCREC_HIP_TRY(hipMalloc(&g.dev[(size_t)r], g.bytes));
        }
    }
    // tile t of an array = bytes [t * B / D, (t + 1) * B / D), cut at 16-byte multiples
    auto tile = [&](size_t bytes, int t, size_t &off, size_t &len) {
        auto cut = [&](int k) { return k >= D ? bytes : (bytes * (size_t)k / (size_t)D) & ~(size_t)15; };
        off = cut(t);
        len = cut(t + 1) - off;
    };
    // 1. every device uploads ITS tile of every array (PCIe, all devices at once) and records an event per array
    std::vector<std::vector<hipEvent_t>> landed((size_t)D, std::vector<hipEvent_t>(arrays.size(), nullptr));
    struct EvFree {
        std::vector<std::vector<hipEvent_t>> *e;
        ~EvFree()
        {
            for (auto &v : *e)
                for (hipEvent_t x : v)
                    if (x) (void)hipEventDestroy(x);
        }
    } evfree{&landed};
    for (int r = 0; r < D; ++r) {
        LOCREC_HIP_TRY(hipSetDevice(pd[(size_t)r].device));
        for (size_t a = 0; a < arrays.size(); ++a) {
            size_t off = 0, len = 0;
            tile(arrays[a].bytes, r, off, len);
            if (len)
                LOCREC_HIP_TRY(hipMemcpyAsync(static_cast<char *>(arrays[a].dev[(size_t)r]) + off,
                                              static_cast<const char *>(arrays[a].host) + off, len, hipMemcpyHostToDevice,
                                              pd[(size_t)r].stream));
            LOCREC_HIP_TRY(hipEventCreateWithFlags(&landed[(size_t)r][a], hipEventDisableTiming));
            LOCREC_HIP_TRY(hipEventRecord(landed[(size_t)r][a], pd[(size_t)r].stream));
        }
    }
    // 2. the block all-gather: device r pulls tile j of array a from device j as soon as THAT tile has landed - the
    // pulls of the first arrays run over xGMI while the later arrays are still coming up over PCIe
    for (size_t a = 0; a < arrays.size(); ++a)
        for (int step = 1; step < D; ++step)      // (ring order: in every step each device reads from a different peer)
            for (int r = 0; r < D; ++r) {
                const int j = (r + step) % D;
                size_t off = 0, len = 0;
                tile(arrays[a].bytes, j, off, len);
                if (!len) continue;
                LOCREC_HIP_TRY(hipSetDevice(pd[(size_t)r].device));
                LOCREC_HIP_TRY(hipStreamWaitEvent(pd[(size_t)r].stream, landed[(size_t)j][a], 0));
                LOCREC_HIP_TRY(hipMemcpyPeerAsync(static_cast<char *>(arrays[a].dev[(size_t)r]) + off, pd[(size_t)r].device,
                                                  static_cast<const char *>(arrays[a].dev[(size_t)j]) + off, pd[(size_t)j].device,
                                                  len, pd[(size_t)r].stream));
            }
    for (int r = 0; r < D; ++r) {
        LOCREC_HIP_TRY(hipSetDevice(pd[(size_t)r].device));
        LOCREC_HIP_TRY(hipStreamSynchronize(pd[(size_t)r].stream));
    }
    // 3. every device builds its index from its own device arrays, all at once
    const int32_t st = parallel_over(D, [&](int r) -> int32_t {
        LOCREC_HIP_TRY(hipSetDevice(pd[(size_t)r].device));
        auto at = [&](size_t a) { return arrays[a].dev[(size_t)r]; };
        return locrec_knn_create_from_device(
            n, static_cast<const int64_t *>(at(0)), static_cast<const int64_t *>(at(1)), static_cast<const int32_t *>(at(2)),
            static_cast<const double *>(at(3)), p_dim, static_cast<const int64_t *>(at(4)), static_cast<const int32_t *>(at(5)),
            static_cast<const double *>(at(6)), c_dim, static_cast<const int64_t *>(at(7)), static_cast<const int64_t *>(at(8)),
            static_cast<const int64_t *>(at(9)), &h->ix[(size_t)r]);
    });
    LOCREC_TRY(st);
    *out = h.release();
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" void locrec_knn_replicas_destroy(locrec_knn_replicas *h) { delete h; }

extern "C" int32_t locrec_knn_replicas_info(const locrec_knn_replicas *h, int32_t *out_devices, locrec_knn_index **out_first) try
{
    if (!h) return fail(LOCREC_E_INVALID_ARG, "handle is NULL");
    if (out_devices) *out_devices = (int32_t)h->devices.size();
    if (out_first) *out_first = h->ix.empty() ? nullptr : h->ix[0];
    return LOCREC_OK;
} LOCREC_CATCH_ALL

namespace {

// contiguous share of nq queries for replica r of D
void share_of(int64_t nq, int r, int D, int64_t &first, int64_t &count)
{
    first = nq * r / D;
    count = nq * (r + 1) / D - first;
}

}  // namespace

// Batched makeRecommendations over the replicas: replica r takes queries [nq r / D, nq (r + 1) / D) of the list (each
// replica sorts its share by its own row order inside locrec_knn_recommend_batch); rows come back in input order.
extern "C" int32_t locrec_knn_replicas_recommend_batch(locrec_knn_replicas *h, int64_t nq, const int64_t *person_ids, double pw,
                                                       double cw, int64_t k, int64_t *out_offsets, int64_t *out_places,
                                                       double *out_ratings, int64_t *inout_capacity) try
{
    if (!h || !out_offsets || !inout_capacity || nq < 0 || (nq > 0 && !person_ids)) return fail(LOCREC_E_INVALID_ARG, "bad arguments");
    const int D = (int)h->ix.size();
    if (D == 1) {
        LOCREC_HIP_TRY(hipSetDevice(h->devices[0]));
        return locrec_knn_recommend_batch(h->ix[0], nq, person_ids, pw, cw, k, out_offsets, out_places, out_ratings, inout_capacity);
    }
    std::vector<std::vector<int64_t>> off((size_t)D), pl((size_t)D);
    std::vector<std::vector<double>> ra((size_t)D);
    LOCREC_TRY(parallel_over(D, [&](int r) -> int32_t {
        int64_t first = 0, cnt = 0;
        share_of(nq, r, D, first, cnt);
        off[(size_t)r].assign((size_t)cnt + 1, 0);
        if (cnt == 0) return LOCREC_OK;
        LOCREC_HIP_TRY(hipSetDevice(h->devices[(size_t)r]));
        int64_t cap = 0;
        for (int pass = 0; pass < 2; ++pass) {  // sizes, then rows
            LOCREC_TRY(locrec_knn_recommend_batch(h->ix[(size_t)r], cnt, person_ids + first, pw, cw, k, off[(size_t)r].data(),
                                                  pl[(size_t)r].data(), ra[(size_t)r].data(), &cap));
            if ((size_t)cap <= pl[(size_t)r].size()) break;
            pl[(size_t)r].resize((size_t)cap);
            ra[(size_t)r].resize((size_t)cap);
        }
        return LOCREC_OK;
    }));
    int64_t total = 0;
    out_offsets[0] = 0;
    for (int r = 0; r < D; ++r) {
        int64_t first = 0, cnt = 0;
        share_of(nq, r, D, first, cnt);
        for (int64_t i = 0; i < cnt; ++i) out_offsets[first + i + 1] = total + off[(size_t)r][(size_t)i + 1];
        total += off[(size_t)r][(size_t)cnt];
    }
    const int64_t cap = *inout_capacity;
    *inout_capacity = total;
    if (total > cap || total == 0) return LOCREC_OK;
    if (!out_places || !out_ratings) return fail(LOCREC_E_INVALID_ARG, "NULL output buffer");
    int64_t at = 0;
    for (int r = 0; r < D; ++r) {
        int64_t first = 0, cnt = 0;
        share_of(nq, r, D, first, cnt);
        const int64_t rows = off[(size_t)r][(size_t)cnt];
        std::copy(pl[(size_t)r].begin(), pl[(size_t)r].begin() + rows, out_places + at);
        std::copy(ra[(size_t)r].begin(), ra[(size_t)r].begin() + rows, out_ratings + at);
        at += rows;
    }
    return LOCREC_OK;
} LOCREC_CATCH_ALL

// Batched findSimilarPersons over the replicas (layout of locrec_knn_query_batch).
extern "C" int32_t locrec_knn_replicas_query_batch(locrec_knn_replicas *h, int64_t nq, const int64_t *person_ids, double pw, double cw,
                                                   int64_t k, int64_t *out_person_ids, double *out_similarities, int64_t *out_counts) try
{
    if (!h || nq < 0 || (nq > 0 && (!person_ids || !out_person_ids || !out_similarities || !out_counts)))
        return fail(LOCREC_E_INVALID_ARG, "bad arguments");
    const int D = (int)h->ix.size();
    return parallel_over(D, [&](int r) -> int32_t {
        int64_t first = 0, cnt = 0;
        share_of(nq, r, D, first, cnt);
        if (cnt == 0 && !(r == 0 && nq == 0)) return LOCREC_OK;
        LOCREC_HIP_TRY(hipSetDevice(h->devices[(size_t)r]));
        const int64_t kk = std::max<int64_t>(k, 0);
        return locrec_knn_query_batch(h->ix[(size_t)r], cnt, person_ids + first, pw, cw, k, out_person_ids + first * kk,
                                      out_similarities + first * kk, out_counts + first);
    });
} LOCREC_CATCH_ALL

// =====================================================================================================
// SG: one graph, rows sharded over the devices

namespace {

struct SigmaPtrs {
    const double *p[kMaxDevices];
};

// total[l] = the owner's entry (by_target: l % D) or the shards' entries added in device order
__global__ __launch_bounds__(256) void sgm_exchange(const SigmaPtrs sig, int32_t D, int32_t T, int32_t by_target, double *total)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= T) return;
    if (by_target) {
        total[l] = sig.p[l % D][l];
    } else {
        double s = sig.p[0][l];
        for (int d = 1; d < D; ++d) s = s + sig.p[d][l];
        total[l] = s;
    }
}

}  // namespace

struct locrec_sg_sharded {
    std::vector<int32_t> devices;
    std::vector<locrec_sg_graph *> g;
    std::vector<PerDevice> pd;
    int32_t by_target = 0;
    int64_t live = 0;
    bool peer = false;
    // per device: its own sigma (two parities: a peer may still read sweep i while sweep i + 1 is written), the
    // exchanged total, and - without peer access - staged copies of the peers' sigmas
    std::vector<DevBuf<double>> sigma, total, staged;
    std::vector<hipEvent_t> ev_sigma;  // [device * 2 + parity]
    bool have_result = false;
    ~locrec_sg_sharded()
    {
        for (hipEvent_t e : ev_sigma)
            if (e) (void)hipEventDestroy(e);
        for (size_t r = 0; r < g.size(); ++r)
            if (g[r]) {
                (void)hipSetDevice(devices[r]);
                (void)locrec_sg_destroy(g[r]);
            }
        for (size_t r = 0; r < devices.size(); ++r) {  // the DevBufs free on their own device
            (void)hipSetDevice(devices[r]);
            if (r < sigma.size()) sigma[r].release();
            if (r < total.size()) total[r].release();
            if (r < staged.size()) staged[r].release();
        }
    }
};

extern "C" int32_t locrec_sg_sharded_create(int32_t n_devices, const int32_t *device_ids, int64_t n_edges, const int64_t *source_ids,
                                            const int64_t *target_ids, const double *balanced_weights, int32_t by_target,
                                            locrec_sg_sharded **out) try
{
    if (!out) return fail(LOCREC_E_INVALID_ARG, "out is NULL");
    *out = nullptr;
    std::unique_ptr<locrec_sg_sharded> h(new locrec_sg_sharded);
    LOCREC_TRY(resolve_devices(n_devices, device_ids, h->devices));
    const int D = (int)h->devices.size();
    h->by_target = by_target ? 1 : 0;
    h->g.assign((size_t)D, nullptr);
    h->pd = std::vector<PerDevice>((size_t)D);
    h->peer = enable_peer_access(h->devices);
    LOCREC_TRY(parallel_over(D, [&](int r) -> int32_t {
        LOCREC_HIP_TRY(hipSetDevice(h->devices[(size_t)r]));
        if (D == 1) return locrec_sg_create(n_edges, source_ids, target_ids, balanced_weights, &h->g[0]);
        return (by_target ? locrec_sg_create_target_sharded : locrec_sg_create_sharded)(n_edges, source_ids, target_ids,
                                                                                     balanced_weights, r, D, &h->g[(size_t)r]);
    }));
    if (D == 1) {
        *out = h.release();
        return LOCREC_OK;
    }
    LOCREC_TRY(locrec_sg_live_count(h->g[0], &h->live));
    const size_t T = (size_t)std::max<int64_t>(1, h->live);
    h->sigma = std::vector<DevBuf<double>>((size_t)D);
    h->total = std::vector<DevBuf<double>>((size_t)D);
    h->staged = std::vector<DevBuf<double>>((size_t)D);
    h->ev_sigma.assign((size_t)D * 2, nullptr);
    for (int r = 0; r < D; ++r) {
        h->pd[(size_t)r].device = h->devices[(size_t)r];
        LOCREC_HIP_TRY(hipSetDevice(h->devices[(size_t)r]));
        LOCREC_HIP_TRY(hipStreamCreateWithFlags(&h->pd[(size_t)r].stream, hipStreamNonBlocking));
        LOCREC_TRY(locrec_sg_set_stream(h->g[(size_t)r], h->pd[(size_t)r].stream));
        LOCREC_TRY(h->sigma[(size_t)r].alloc(2 * T));
        LOCREC_TRY(h->total[(size_t)r].alloc(T));
        if (!h->peer) LOCREC_TRY(h->staged[(size_t)r].alloc((size_t)D * T));
        for (int par = 0; par < 2; ++par)
            LOCREC_HIP_TRY(hipEventCreateWithFlags(&h->ev_sigma[(size_t)r * 2 + par], hipEventDisableTiming));
    }
    *out = h.release();
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" void locrec_sg_sharded_destroy(locrec_sg_sharded *h) { delete h; }

namespace {

// one sweep on every device: sigma -> exchange -> apply, ordered across the devices by events
int32_t sharded_sweep(locrec_sg_sharded *h, int64_t it, double alpha)
{
    const int D = (int)h->g.size();
    const int par = (int)(it & 1);
    const int32_t T = (int32_t)h->live;
    const size_t Ts = (size_t)std::max<int64_t>(1, h->live);
    for (int r = 0; r < D; ++r) {
        LOCREC_HIP_TRY(hipSetDevice(h->devices[(size_t)r]));
        LOCREC_TRY(locrec_sg_shard_sigma(h->g[(size_t)r], h->sigma[(size_t)r].p + (size_t)par * Ts));
        LOCREC_HIP_TRY(hipEventRecord(h->ev_sigma[(size_t)r * 2 + par], h->pd[(size_t)r].stream));
    }
    for (int r = 0; r < D; ++r) {
        LOCREC_HIP_TRY(hipSetDevice(h->devices[(size_t)r]));
        hipStream_t s = h->pd[(size_t)r].stream;
        SigmaPtrs sp{};
        for (int j = 0; j < D; ++j) {
            if (j != r) LOCREC_HIP_TRY(hipStreamWaitEvent(s, h->ev_sigma[(size_t)j * 2 + par], 0));
            const double *src = h->sigma[(size_t)j].p + (size_t)par * Ts;
            if (!h->peer && j != r && h->devices[(size_t)j] != h->devices[(size_t)r]) {
                double *dst = h->staged[(size_t)r].p + (size_t)j * Ts;
                LOCREC_HIP_TRY(hipMemcpyPeerAsync(dst, h->devices[(size_t)r], src, h->devices[(size_t)j], Ts * sizeof(double), s));
                src = dst;
            }
            sp.p[j] = src;
        }
        if (T > 0) hipLaunchKernelGGL(sgm_exchange, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, sp, D, T, h->by_target,
                                      h->total[(size_t)r].p);
        LOCREC_HIP_TRY(hipGetLastError());
        LOCREC_TRY(locrec_sg_shard_apply(h->g[(size_t)r], h->total[(size_t)r].p, alpha));
    }
    return LOCREC_OK;
}

int32_t sharded_run(locrec_sg_sharded *h, int64_t vertex_id, double alpha, double epsilon, int64_t max_iterations, bool fixed)
{
    const int D = (int)h->g.size();
    h->have_result = false;
    if (!(epsilon >= 0)) return fail(LOCREC_E_INVALID_ARG, "requirement failed: epsilon must be non-negative");
    if (max_iterations < 0) return fail(LOCREC_E_INVALID_ARG, "requirement failed: max iterations number must be non-negative");
    if (D == 1) {
        LOCREC_HIP_TRY(hipSetDevice(h->devices[0]));
        LOCREC_TRY(fixed ? locrec_sg_sweeps_async(h->g[0], vertex_id, alpha, max_iterations)
                         : locrec_sg_iterate_async(h->g[0], vertex_id, alpha, epsilon, max_iterations));
        h->have_result = true;
        return LOCREC_OK;
    }
    for (int r = 0; r < D; ++r) {
        LOCREC_HIP_TRY(hipSetDevice(h->devices[(size_t)r]));
        LOCREC_TRY(locrec_sg_shard_begin(h->g[(size_t)r], vertex_id));
    }
    // step(), StochasticRecommender.scala:92-106
    int64_t it = 0;
    int32_t converged = 0;
    const double eps2 = epsilon * epsilon;
    while (it < max_iterations) {
        LOCREC_TRY(sharded_sweep(h, it, alpha));
        if (!fixed) {
            double d2 = 0.0;
            LOCREC_HIP_TRY(hipSetDevice(h->devices[0]));
            LOCREC_TRY(locrec_sg_shard_d2(h->g[0], &d2));  // (every device holds the same x: device 0 decides)
            if (d2 <= eps2) {
                converged = 1;
                break;
            }
        }
        ++it;
    }
    for (int r = 0; r < D; ++r) {
        LOCREC_HIP_TRY(hipSetDevice(h->devices[(size_t)r]));
        LOCREC_TRY(locrec_sg_shard_finish(h->g[(size_t)r], it, converged));
    }
    h->have_result = true;
    return LOCREC_OK;
}

}  // namespace

extern "C" int32_t locrec_sg_sharded_iterate_async(locrec_sg_sharded *h, int64_t vertex_id, double alpha, double epsilon,
                                                   int64_t max_iterations) try
{
    if (!h) return fail(LOCREC_E_INVALID_ARG, "handle is NULL");
    return sharded_run(h, vertex_id, alpha, epsilon, max_iterations, false);
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_sharded_sweeps_async(locrec_sg_sharded *h, int64_t vertex_id, double alpha, int64_t sweeps) try
{
    if (!h) return fail(LOCREC_E_INVALID_ARG, "handle is NULL");
    return sharded_run(h, vertex_id, alpha, 0.0, sweeps, true);
} LOCREC_CATCH_ALL

// the result as device `which` holds it (every device holds the same x; which = 0 is the usual choice)
extern "C" int32_t locrec_sg_sharded_fetch(locrec_sg_sharded *h, int32_t which, int64_t *out_ids, double *out_probabilities,
                                           int64_t *inout_count, int64_t *out_iterations, int32_t *out_converged) try
{
    if (!h) return fail(LOCREC_E_INVALID_ARG, "handle is NULL");
    if (!h->have_result) return fail(LOCREC_E_INVALID_ARG, "no iteration has been enqueued");
    if (which < 0 || which >= (int32_t)h->g.size()) return fail(LOCREC_E_INVALID_ARG, "no such device in the handle");
    LOCREC_HIP_TRY(hipSetDevice(h->devices[(size_t)which]));
    return locrec_sg_fetch(h->g[(size_t)which], out_ids, out_probabilities, inout_count, out_iterations, out_converged);
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_sharded_recommend(locrec_sg_sharded *h, int64_t vertex_id, double alpha, double epsilon,
                                               int64_t max_iterations, int64_t *out_ids, double *out_probabilities,
                                               int64_t *inout_count, int64_t *out_iterations, int32_t *out_converged) try
{
    LOCREC_TRY(locrec_sg_sharded_iterate_async(h, vertex_id, alpha, epsilon, max_iterations));
    return locrec_sg_sharded_fetch(h, 0, out_ids, out_probabilities, inout_count, out_iterations, out_converged);
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_sg_sharded_info(const locrec_sg_sharded *h, int32_t *out_devices, int32_t *out_peer_access,
                                          int64_t *out_exchanged_entries, int64_t *out_vertices) try
{
    if (!h) return fail(LOCREC_E_INVALID_ARG, "handle is NULL");
    if (out_devices) *out_devices = (int32_t)h->devices.size();
    if (out_peer_access) *out_peer_access = h->peer ? 1 : 0;
    if (out_exchanged_entries) *out_exchanged_entries = h->live;
    if (out_vertices) {
        (void)hipSetDevice(h->devices[0]);
        LOCREC_TRY(locrec_sg_info(h->g[0], out_vertices, nullptr, nullptr));
    }
    return LOCREC_OK;
} LOCREC_CATCH_ALL
