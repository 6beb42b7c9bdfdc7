// common.hip -- process-wide pieces of the C ABI (include/locrec.h): error text,
// version, device selection.
#include "common.h"

#include <atomic>

namespace locrec {

std::string &last_error_ref()
{
    thread_local std::string err;
    return err;
}

int32_t status_of_current_exception() noexcept
{
    try {
        throw;
    } catch (const std::bad_alloc &) {
        try {
            return fail(LOCREC_E_OOM, "host allocation failed");
        } catch (...) {
            return LOCREC_E_OOM;
        }
    } catch (const std::exception &e) {
        try {
            return fail(LOCREC_E_DEVICE, "internal error: %s", e.what());
        } catch (...) {
            return LOCREC_E_DEVICE;
        }
    } catch (...) {
        return LOCREC_E_DEVICE;
    }
}

int32_t ensure_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(LOCREC_E_DEVICE, "no usable HIP device (%s): liblocrec has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorName(e));
    return LOCREC_OK;
}

static std::atomic<int64_t> g_device_allocations{0};
static std::atomic<int64_t> g_device_bytes{0};
void count_device_release(size_t bytes) { g_device_bytes.fetch_sub((int64_t)bytes, std::memory_order_relaxed); }
int64_t device_bytes_in_use() { return g_device_bytes.load(std::memory_order_relaxed); }
void count_device_allocation(size_t bytes)
{
    g_device_bytes.fetch_add((int64_t)bytes, std::memory_order_relaxed);
    static const bool trace = std::getenv("LOCREC_TRACE_ALLOC") != nullptr;
    const int64_t k = g_device_allocations.fetch_add(1, std::memory_order_relaxed) + 1;
    if (trace) fprintf(stderr, "locrec: device allocation %lld: %zu bytes\n", (long long)k, bytes);
}

}  // namespace locrec

using namespace locrec;

extern "C" const char *locrec_last_error(void) { return last_error_ref().c_str(); }

extern "C" int32_t locrec_device_allocations(int64_t *out_count)
{
    if (!out_count) return fail(LOCREC_E_INVALID_ARG, "out_count is NULL");
    *out_count = g_device_allocations.load(std::memory_order_relaxed);
    return LOCREC_OK;
}

extern "C" int32_t locrec_device_bytes_in_use(int64_t *out_bytes)
{
    if (!out_bytes) return fail(LOCREC_E_INVALID_ARG, "out_bytes is NULL");
    *out_bytes = device_bytes_in_use();
    return LOCREC_OK;
}

extern "C" const char *locrec_version(void) { return "locrec 0.1 (gfx950)"; }

extern "C" int32_t locrec_device_count(int32_t *out_count) try
{
    if (!out_count) return fail(LOCREC_E_INVALID_ARG, "out_count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out_count = 0;
        return fail(LOCREC_E_DEVICE, "hipGetDeviceCount: %s", hipGetErrorName(e));
    }
    *out_count = n;
    return LOCREC_OK;
} LOCREC_CATCH_ALL

extern "C" int32_t locrec_set_device(int32_t ordinal) try
{
    LOCREC_TRY(ensure_device());
    LOCREC_HIP_TRY(hipSetDevice(ordinal));
    return LOCREC_OK;
} LOCREC_CATCH_ALL
