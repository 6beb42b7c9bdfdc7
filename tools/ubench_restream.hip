// ubench_restream.hip -- what does the memory hierarchy deliver when MANY blocks re-read the same
// ~100 MB array (the batched KNN scan's access pattern: 2 chunks x 1024 tiles, every block of a chunk
// streams the same 50 MB)?  Variants: loads in flight per wave, cache policy, phase alignment.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_restream.hip -o /tmp/ubench_restream && /tmp/ubench_restream
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                    \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// slice = G groups of 1 KiB; block (chunk, tile) of 8 waves; wave w reads slices w, w+8, ... of its chunk
template <int G, int DEPTH, bool NT>
__global__ __launch_bounds__(512, 4) void k_stream(const u32x4 *data, int slices_per_chunk, unsigned *out, int work)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4 *base = data + (size_t)blockIdx.x * slices_per_chunk * G * 64;
    u32x4 buf[DEPTH][G];
    unsigned acc = 0;
    const int iters = slices_per_chunk / 8;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const u32x4 *p = base + ((size_t)(d * 8 + wave) * G + g) * 64 + lane;
            buf[d][g] = NT ? __builtin_nontemporal_load(p) : *p;
        }
    for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                acc ^= buf[d][g].x + buf[d][g].y + buf[d][g].z + buf[d][g].w;
                const int nit = it + d + DEPTH;
                if (nit < iters) {
                    const u32x4 *p = base + ((size_t)(nit * 8 + wave) * G + g) * 64 + lane;
                    buf[d][g] = NT ? __builtin_nontemporal_load(p) : *p;
                }
            }
            for (int k = 0; k < work; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc) : "v"(k));  // stand-in for compute
        }
    }
    out[(blockIdx.y * gridDim.x + blockIdx.x) * 512 + threadIdx.x] = acc;
}

template <int G, int DEPTH, bool NT>
void run(const char *name, const u32x4 *data, int slices_per_chunk, int tiles, unsigned *out, int work)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_stream<G, DEPTH, NT>), dim3(2, tiles), dim3(512), 0, 0, data, slices_per_chunk, out, work);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_stream<G, DEPTH, NT>), dim3(2, tiles), dim3(512), 0, 0, data, slices_per_chunk, out, work);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 2.0 * tiles * slices_per_chunk * G * 1024.0;
    printf("%-44s G=%d depth=%d work=%4d tiles=%4d: %7.2f ms, %6.2f TB/s requested\n", name, G, DEPTH, work, tiles, ms, bytes / ms / 1e9);
}

int main()
{
    const int slices_per_chunk = 7808;  // ~500k rows per chunk
    const size_t n16 = (size_t)2 * slices_per_chunk * 6 * 64;
    u32x4 *data;
    unsigned *out;
    CHECK(hipMalloc(&data, n16 * 16));
    CHECK(hipMemset(data, 1, n16 * 16));
    CHECK(hipMalloc(&out, (size_t)2 * 2048 * 512 * 4));
    for (int tiles : {256, 1024}) {
        run<6, 1, false>("default loads", data, slices_per_chunk, tiles, out, 0);
        run<6, 2, false>("default loads", data, slices_per_chunk, tiles, out, 0);
        run<6, 1, true>("nontemporal loads", data, slices_per_chunk, tiles, out, 0);
        run<6, 1, false>("default loads + compute", data, slices_per_chunk, tiles, out, 300);
        run<6, 2, false>("default loads + compute", data, slices_per_chunk, tiles, out, 300);
        run<3, 2, false>("half the bytes per slice", data, slices_per_chunk, tiles, out, 0);
        run<3, 2, false>("half the bytes per slice + compute", data, slices_per_chunk, tiles, out, 300);
    }
    return 0;
}
