// ubench_valu.hip -- issue-rate microbenchmarks behind bench.py's "valu" roofline (gfx950).
// Measures, per SIMD, how many wave64 instructions per second the chip sustains for the
// instruction mix of the KNN scan: v_pk_mad_u16, v_mad_u32_u24, ds_read_b128 (+ packed MADs).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o gpurun_out/ubench_valu && gpurun_out/ubench_valu
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kUnroll = 16;

// 16 independent accumulators, one v_pk_mad_u16 each per iteration
__global__ void k_pkmad(unsigned *out, int iters, unsigned seed)
{
    unsigned a[kUnroll];
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) a[i] = seed + threadIdx.x * 7 + i;
    unsigned b = seed ^ threadIdx.x, c = seed + 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mad24(unsigned *out, int iters, unsigned seed)
{
    unsigned a[kUnroll];
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) a[i] = seed + threadIdx.x * 7 + i;
    unsigned b = seed ^ threadIdx.x, c = seed + 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_fma32(unsigned *out, int iters, unsigned seed)
{
    float a[kUnroll];
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) a[i] = (float)(seed + threadIdx.x * 7 + i);
    float b = (float)(seed ^ threadIdx.x) * 1e-9f, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(s);
}

// the scan's inner pattern: one ds_read_b128 of a per-lane panel row, then four (or eight) packed MADs
template <int MADS, int STRIDE_DW>
__global__ void k_lds_mad(unsigned *out, int iters, unsigned seed)
{
    __shared__ __align__(16) unsigned panel[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) panel[i] = seed + i;
    __syncthreads();
    unsigned a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + i;
    unsigned row = (threadIdx.x * 37u + seed) & 255u;
    unsigned v = seed | 1u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32x4 p = *reinterpret_cast<const u32x4 *>(&panel[(row * STRIDE_DW + u * 4) & 16380u]);
            asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(a[0]) : "v"(p.x), "v"(v));
            asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(a[1]) : "v"(p.y), "v"(v));
            asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(a[2]) : "v"(p.z), "v"(v));
            asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(a[3]) : "v"(p.w), "v"(v));
            if (MADS == 8) {
                asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(a[4]) : "v"(p.x), "v"(v));
                asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(a[5]) : "v"(p.y), "v"(v));
                asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(a[6]) : "v"(p.z), "v"(v));
                asm volatile("v_pk_mad_u16 %0, %1, %2, %0" : "+v"(a[7]) : "v"(p.w), "v"(v));
            }
        }
        row = (row * 5u + 1u) & 255u;
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
double run(K kern, int blocks, int threads, int iters, unsigned *out)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters / 8, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3;
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, ncu, p.clockRate);
    unsigned *out;
    CHECK(hipMalloc(&out, (size_t)ncu * 8 * 1024 * sizeof(unsigned)));
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD
        const int threads = 256;     // 4 waves = one per SIMD
        const int blocks = ncu * wps;
        const double simds = ncu * 4.0;
        double t = run(k_pkmad, blocks, threads, iters, out);
        const double n = (double)blocks * 4 * iters * kUnroll;
        printf("v_pk_mad_u16  %d waves/SIMD: %.1f G wave-instr/s chip, %.2f cycles/instr/SIMD at 2.4 GHz\n", wps, n / t / 1e9,
               2.4e9 * t * simds / n);
        t = run(k_mad24, blocks, threads, iters, out);
        printf("v_mad_u32_u24 %d waves/SIMD: %.1f G wave-instr/s chip, %.2f cycles/instr/SIMD\n", wps, n / t / 1e9,
               2.4e9 * t * simds / n);
        t = run(k_fma32, blocks, threads, iters, out);
        printf("v_fma_f32     %d waves/SIMD: %.1f G wave-instr/s chip, %.2f cycles/instr/SIMD\n", wps, n / t / 1e9,
               2.4e9 * t * simds / n);
    }
    for (int wps : {2, 4}) {
        const int blocks = ncu * wps;
        const double simds = ncu * 4.0;
        const int it2 = 4000;
        double t = run(k_lds_mad<4, 8>, blocks, 256, it2, out);
        double reads = (double)blocks * 4 * it2 * 4;
        printf("ds_read_b128 + 4 pk_mad (32-B rows)  %d waves/SIMD: %.1f G reads/s chip = %.2f cycles/read/CU, %.1f G pk_mad/s\n", wps,
               reads / t / 1e9, 2.4e9 * t * ncu / reads, reads * 4 / t / 1e9);
        t = run(k_lds_mad<8, 8>, blocks, 256, it2, out);
        printf("ds_read_b128 + 8 pk_mad (32-B rows)  %d waves/SIMD: %.1f G reads/s chip = %.2f cycles/read/CU, %.1f G pk_mad/s\n", wps,
               reads / t / 1e9, 2.4e9 * t * ncu / reads, reads * 8 / t / 1e9);
        t = run(k_lds_mad<4, 20>, blocks, 256, it2, out);
        printf("ds_read_b128 + 4 pk_mad (80-B rows)  %d waves/SIMD: %.1f G reads/s chip = %.2f cycles/read/CU\n", wps,
               reads / t / 1e9, 2.4e9 * t * ncu / reads);
        (void)simds;
    }
    CHECK(hipFree(out));
    return 0;
}
