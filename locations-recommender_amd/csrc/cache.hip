// cache.hip -- process-wide cache of device-resident handles (include/locrec.h, "Handle cache").
//
// The reference's mains build a NEW recommender from freshly read DataFrames for every stdin request
// (knn/KnnRecommenderMain.scala:53-67, stochastic/StochasticRecommenderMain.scala:53-62) and nothing they
// construct is ever closed.  Behind those UNCHANGED mains a device handle must therefore outlive the
// recommender object that first asked for it: the host classes derive a key from what the DataFrames ARE
// (their input files, sizes and modification times - not from the weights, K, epsilon or maxIterations, which
// are per-request parameters of the C ABI), look the key up here, and only a miss pays for collect + upload +
// index build.  The cache owns the handles; a recommender object holds a reference.  Least-recently-used
// entries that nobody references are destroyed when the library's live device bytes exceed the budget.
//
// Host-only code (no kernels); one mutex; handles themselves stay single-threaded as the header says.
#include "common.h"

#include <algorithm>
#include <list>
#include <mutex>
#include <string>
#include <unordered_map>

using namespace locrec;

namespace {

struct Entry {
    int32_t kind;
    std::string key;
    void *handle;
    int64_t refs;
    int64_t bytes;   // live device bytes the create call added (work buffers that grow later are in the global count)
    bool evicted;    // dropped from the key map while referenced: destroyed by the last release
};

struct Cache {
    std::mutex mu;
    std::list<Entry> lru;  // front = most recently used
    std::unordered_map<std::string, std::list<Entry>::iterator> by_key;
    std::unordered_map<void *, std::list<Entry>::iterator> by_handle;
    int64_t limit_bytes = (int64_t)64 << 30;  // a quarter of the 288 GB of one MI355X; LOCREC_CACHE_BYTES overrides
    int64_t limit_entries = 64;               // the reference has one graph per region and per region pair
    int64_t hits = 0, misses = 0, evictions = 0;
    Cache()
    {
        if (const char *e = std::getenv("LOCREC_CACHE_BYTES")) limit_bytes = std::max<long long>(0, atoll(e));
        if (const char *e = std::getenv("LOCREC_CACHE_ENTRIES")) limit_entries = std::max<long long>(0, atoll(e));
    }
};

Cache &cache()
{
    static Cache *c = new Cache;  // never destroyed: handles must not be torn down after the HIP runtime at exit
    return *c;
}

std::string map_key(int32_t kind, const char *key) { return std::string(kind == LOCREC_CACHE_KNN ? "K:" : "S:") + key; }

void destroy_handle(int32_t kind, void *h)
{
    if (kind == LOCREC_CACHE_KNN)
        (void)locrec_knn_destroy(static_cast<locrec_knn_index *>(h));
    else
        (void)locrec_sg_destroy(static_cast<locrec_sg_graph *>(h));
}

// Evict unreferenced entries, least recently used first, until both limits hold.  `keep` is never evicted.
// Called with the mutex held; the victims are destroyed by the caller AFTER unlocking (a destroy synchronises
// the handle's stream).
void collect_victims(Cache &c, void *keep, std::vector<std::pair<int32_t, void *>> &victims)
{
    int64_t live = device_bytes_in_use();
    int64_t entries = (int64_t)c.by_key.size();
    for (auto it = c.lru.end(); it != c.lru.begin() && (live > c.limit_bytes || entries > c.limit_entries);) {
        --it;
        if (it->refs > 0 || it->handle == keep || it->evicted) continue;
        victims.emplace_back(it->kind, it->handle);
        live -= it->bytes;
        --entries;
        ++c.evictions;
        c.by_key.erase(map_key(it->kind, it->key.c_str()));
        c.by_handle.erase(it->handle);
        it = c.lru.erase(it);
    }
}

bool kind_ok(int32_t kind) { return kind == LOCREC_CACHE_KNN || kind == LOCREC_CACHE_SG; }

}  // namespace

extern "C" int32_t locrec_cache_acquire(int32_t kind, const char *key, void **out_handle) try
{
    if (!kind_ok(kind) || !key || !out_handle) return fail(LOCREC_E_INVALID_ARG, "locrec_cache_acquire: bad argument");
    Cache &c = cache();
    std::lock_guard<std::mutex> g(c.mu);
    auto f = c.by_key.find(map_key(kind, key));
    if (f == c.by_key.end()) {
        ++c.misses;
        *out_handle = nullptr;
        return LOCREC_OK;
    }
    ++c.hits;
    ++f->second->refs;
    c.lru.splice(c.lru.begin(), c.lru, f->second);
    *out_handle = f->second->handle;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_cache_publish(int32_t kind, const char *key, void *handle, int64_t device_bytes,
                                        void **out_handle) try
{
    if (!kind_ok(kind) || !key || !handle || !out_handle)
        return fail(LOCREC_E_INVALID_ARG, "locrec_cache_publish: bad argument");
    Cache &c = cache();
    std::vector<std::pair<int32_t, void *>> victims;
    {
        std::lock_guard<std::mutex> g(c.mu);
        if (c.by_handle.count(handle)) return fail(LOCREC_E_INVALID_ARG, "locrec_cache_publish: the handle is cached already");
        auto f = c.by_key.find(map_key(kind, key));
        if (f != c.by_key.end()) {
            // another thread built the same data meanwhile: keep the first, drop the newcomer
            ++f->second->refs;
            c.lru.splice(c.lru.begin(), c.lru, f->second);
            *out_handle = f->second->handle;
            victims.emplace_back(kind, handle);
        } else {
            c.lru.push_front(Entry{kind, key, handle, 1, std::max<int64_t>(0, device_bytes), false});
            c.by_key[map_key(kind, key)] = c.lru.begin();
            c.by_handle[handle] = c.lru.begin();
            *out_handle = handle;
            collect_victims(c, handle, victims);
        }
    }
    for (auto &v : victims) destroy_handle(v.first, v.second);
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_cache_release(int32_t kind, void *handle) try
{
    if (!kind_ok(kind)) return fail(LOCREC_E_INVALID_ARG, "locrec_cache_release: bad kind");
    if (!handle) return LOCREC_OK;
    Cache &c = cache();
    std::vector<std::pair<int32_t, void *>> victims;
    {
        std::lock_guard<std::mutex> g(c.mu);
        auto f = c.by_handle.find(handle);
        if (f == c.by_handle.end()) {
            victims.emplace_back(kind, handle);  // never published: the caller's own handle, close() destroys it
        } else {
            auto it = f->second;
            if (it->refs > 0) --it->refs;
            if (it->evicted && it->refs == 0) {
                victims.emplace_back(it->kind, it->handle);
                c.by_handle.erase(f);
                c.lru.erase(it);
            } else {
                collect_victims(c, nullptr, victims);
            }
        }
    }
    for (auto &v : victims) destroy_handle(v.first, v.second);
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_cache_set_limits(int64_t max_device_bytes, int64_t max_entries) try
{
    Cache &c = cache();
    std::vector<std::pair<int32_t, void *>> victims;
    {
        std::lock_guard<std::mutex> g(c.mu);
        if (max_device_bytes >= 0) c.limit_bytes = max_device_bytes;
        if (max_entries >= 0) c.limit_entries = max_entries;
        collect_victims(c, nullptr, victims);
    }
    for (auto &v : victims) destroy_handle(v.first, v.second);
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_cache_clear(void) try
{
    Cache &c = cache();
    std::vector<std::pair<int32_t, void *>> victims;
    {
        std::lock_guard<std::mutex> g(c.mu);
        for (auto it = c.lru.begin(); it != c.lru.end();) {
            if (!it->evicted) c.by_key.erase(map_key(it->kind, it->key.c_str()));
            if (it->refs > 0) {
                it->evicted = true;  // the last release destroys it
                ++it;
            } else {
                victims.emplace_back(it->kind, it->handle);
                c.by_handle.erase(it->handle);
                it = c.lru.erase(it);
            }
        }
    }
    for (auto &v : victims) destroy_handle(v.first, v.second);
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_cache_stats(int64_t *out_entries, int64_t *out_entry_bytes, int64_t *out_hits,
                                      int64_t *out_misses, int64_t *out_evictions) try
{
    Cache &c = cache();
    std::lock_guard<std::mutex> g(c.mu);
    int64_t bytes = 0;
    for (const Entry &e : c.lru) bytes += e.bytes;
    if (out_entries) *out_entries = (int64_t)c.by_key.size();
    if (out_entry_bytes) *out_entry_bytes = bytes;
    if (out_hits) *out_hits = c.hits;
    if (out_misses) *out_misses = c.misses;
    if (out_evictions) *out_evictions = c.evictions;
    return LOCREC_OK;
} LOCREC_CATCH_ALL
