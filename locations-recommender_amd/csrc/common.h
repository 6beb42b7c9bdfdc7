// common.h -- shared host-side plumbing of liblocrec.so (status codes, error
// text, device buffers, event profiling).  gfx950 only; no CPU fallback: every
// entry point fails with LOCREC_E_DEVICE when HIP cannot run the kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "../../include/locrec.h"

namespace locrec {

std::string &last_error_ref();
// device allocations made by this process so far (every DevBuf::alloc): a steady-state step must make none
void count_device_allocation(size_t bytes);  // (LOCREC_TRACE_ALLOC=1: also one line on stderr per allocation)
void count_device_release(size_t bytes);     // bytes of live DevBufs: locrec_device_bytes_in_use, the handle cache's budget
int64_t device_bytes_in_use();

inline int32_t fail(int32_t code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

#define LOCREC_HIP_TRY(expr)                                                              \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return ::locrec::fail(e_ == hipErrorOutOfMemory ? LOCREC_E_OOM : LOCREC_E_DEVICE, \
                                  "HIP error %s at %s:%d: %s", hipGetErrorName(e_), __FILE__, \
                                  __LINE__, #expr);                                       \
    } while (0)

// Nothing may unwind through the C / JNI boundary: every extern "C" entry point is a
// function-try-block ending in LOCREC_CATCH_ALL (bad_alloc -> LOCREC_E_OOM, anything else ->
// LOCREC_E_DEVICE), so a failed host allocation at cfg4 sizes never terminates the JVM.
int32_t status_of_current_exception() noexcept;
#define LOCREC_CATCH_ALL                                   \
    catch (...)                                            \
    {                                                      \
        return ::locrec::status_of_current_exception();    \
    }

#define LOCREC_TRY(expr)              \
    do {                              \
        int32_t s_ = (expr);          \
        if (s_ != LOCREC_OK) return s_; \
    } while (0)

// Owning device allocation.
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) {
            (void)hipFree(p);
            count_device_release(n * sizeof(T));
        }
        p = nullptr;
        n = 0;
    }
    int32_t alloc(size_t count)
    {
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            return fail(e == hipErrorOutOfMemory ? LOCREC_E_OOM : LOCREC_E_DEVICE,
                        "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorName(e));
        }
        n = count;
        count_device_allocation(count * sizeof(T));
        return LOCREC_OK;
    }
    // grow-only, with an eighth of headroom: a batch buffer sized by the data (the hits of a batch of queries, the
    // recommendation rows) would otherwise be freed and allocated again - a device-wide synchronisation plus a
    // half-gigabyte hipMalloc - at every batch that is a little larger than all before it
    int32_t reserve(size_t count) { return count <= n && p ? LOCREC_OK : alloc(count + (count >> 3)); }
    int32_t upload(const T *host, size_t count, hipStream_t s = nullptr)
    {
        LOCREC_TRY(alloc(count));
        if (count) LOCREC_HIP_TRY(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, s));
        return LOCREC_OK;
    }
    int32_t upload(const std::vector<T> &v, hipStream_t s = nullptr) { return upload(v.data(), v.size(), s); }
    size_t bytes() const { return n * sizeof(T); }
};

// Times the launches of one kernel with HIP events (bench.py's roofline.achieved is computed from this, not
// from host clocks).  The events are attached to the kernel's OWN dispatch (hipExtLaunchKernelGGL's start / stop
// events: the packet's begin / end timestamps, what rocprofv3 reports too) - two separate event records around a
// 10 - 30 us kernel add 2 - 3 us of marker processing to the interval (round 1 measured sg_sweep at 13.5 us
// where rocprofv3 saw 10.6).  LOCREC_LAUNCH_PROFILED launches through whichever form applies.
struct KernelProfile {
    bool on = false;
    std::vector<hipEvent_t> pool;  // pairs
    size_t used = 0;               // events handed out since last read
    ~KernelProfile()
    {
        for (hipEvent_t e : pool) (void)hipEventDestroy(e);
    }
    int32_t begin(hipStream_t s)
    {
        if (!on) return LOCREC_OK;
        if (used + 2 > pool.size()) {
            hipEvent_t a, b;
            LOCREC_HIP_TRY(hipEventCreate(&a));
            LOCREC_HIP_TRY(hipEventCreate(&b));
            pool.push_back(a);
            pool.push_back(b);
        }
        LOCREC_HIP_TRY(hipEventRecord(pool[used], s));
        return LOCREC_OK;
    }
    // the next (start, stop) pair for a profiled launch; false when profiling is off
    bool pair(hipEvent_t *a, hipEvent_t *b)
    {
        if (!on) return false;
        if (used + 2 > pool.size()) {
            hipEvent_t x, y;
            if (hipEventCreate(&x) != hipSuccess) return false;
            if (hipEventCreate(&y) != hipSuccess) {
                (void)hipEventDestroy(x);
                return false;
            }
            pool.push_back(x);
            pool.push_back(y);
        }
        *a = pool[used];
        *b = pool[used + 1];
        used += 2;
        return true;
    }
    int32_t end(hipStream_t s)
    {
        if (!on) return LOCREC_OK;
        LOCREC_HIP_TRY(hipEventRecord(pool[used + 1], s));
        used += 2;
        return LOCREC_OK;
    }
    int32_t read(hipStream_t s, double *ms, int64_t *launches)
    {
        LOCREC_HIP_TRY(hipStreamSynchronize(s));
        double tot = 0;
        for (size_t i = 0; i + 1 < used; i += 2) {
            float t = 0;
            LOCREC_HIP_TRY(hipEventElapsedTime(&t, pool[i], pool[i + 1]));
            tot += t;
        }
        if (ms) *ms = tot;
        if (launches) *launches = (int64_t)(used / 2);
        used = 0;
        return LOCREC_OK;
    }
};

#define LOCREC_LAUNCH_PROFILED(prof, kern, grid, block, lds, stream, ...)                                 \
    do {                                                                                                  \
        hipEvent_t ev_a_, ev_b_;                                                                          \
        if ((prof).pair(&ev_a_, &ev_b_))                                                                  \
            hipExtLaunchKernelGGL(kern, grid, block, lds, stream, ev_a_, ev_b_, 0, __VA_ARGS__);          \
        else                                                                                              \
            hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                              \
    } while (0)

int32_t ensure_device();

// LOCREC_DEBUG_* switches (measurement / fault finding; some give WRONG results by design) exist
// only in a library built with `make DEBUG_SWITCHES=1`; the release liblocrec.so never reads them.
inline const char *debug_env(const char *name)
{
#ifdef LOCREC_DEBUG_SWITCHES
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}  // LOCREC_E_DEVICE unless a gfx950-capable HIP device is usable

}  // namespace locrec
