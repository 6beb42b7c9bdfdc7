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


// ---- one kernel per opcode CLASS the batched scan issues: 16 independent chains of the same
// instruction, so that the figure is the SIMD's issue rate for that class, not a dependency latency.
#define UB_KERNEL(NAME, DECL, ASM)                                                                 \
    __global__ void NAME(unsigned *out, int iters, unsigned seed)                                  \
    {                                                                                              \
        unsigned a[kUnroll];                                                                       \
        _Pragma("unroll") for (int i = 0; i < kUnroll; ++i) a[i] = seed + threadIdx.x * 7 + i;     \
        unsigned b = (seed ^ threadIdx.x) | 0x3c003c00u, c = seed + 3;                             \
        DECL;                                                                                      \
        (void)b; (void)c;                                                                          \
        for (int it = 0; it < iters; ++it) {                                                       \
            _Pragma("unroll") for (int i = 0; i < kUnroll; ++i) { ASM; }                           \
        }                                                                                          \
        unsigned s = 0;                                                                            \
        _Pragma("unroll") for (int i = 0; i < kUnroll; ++i) s ^= a[i];                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                            \
    }
UB_KERNEL(k_and, , asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b)))
UB_KERNEL(k_add, , asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b)))
UB_KERNEL(k_mov, , asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b)))
UB_KERNEL(k_add_sdwa, , asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(a[i]) : "v"(b)))
UB_KERNEL(k_cvt_sdwa, , asm volatile("v_cvt_f16_u16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(a[i]) : "v"(b)))
UB_KERNEL(k_pk_mul_f16, , asm volatile("v_pk_mul_f16 %0, %1, %0" : "+v"(a[i]) : "v"(b)))
UB_KERNEL(k_pk_add_u16, , asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(a[i]) : "v"(b)))
UB_KERNEL(k_dot2c, , asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)))
UB_KERNEL(k_rsq_f32, , asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i])))
UB_KERNEL(k_cvt_f32_u32, , asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i])))
UB_KERNEL(k_alignbit, , asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(b)))
// (one asm statement for the whole group: declared per instruction, the vcc operand makes the compiler put an s_nop
// between them and the figure measures that)
__global__ void k_cndmask(unsigned *out, int iters, unsigned seed)
{
    unsigned a[kUnroll];
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) a[i] = seed + threadIdx.x * 7 + i;
    const unsigned b = seed ^ threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_cmp_lt_u32 vcc, %16, %0\n"
                     "v_cndmask_b32 %0, %0, %16, vcc\n v_cndmask_b32 %1, %1, %16, vcc\n v_cndmask_b32 %2, %2, %16, vcc\n"
                     "v_cndmask_b32 %3, %3, %16, vcc\n v_cndmask_b32 %4, %4, %16, vcc\n v_cndmask_b32 %5, %5, %16, vcc\n"
                     "v_cndmask_b32 %6, %6, %16, vcc\n v_cndmask_b32 %7, %7, %16, vcc\n v_cndmask_b32 %8, %8, %16, vcc\n"
                     "v_cndmask_b32 %9, %9, %16, vcc\n v_cndmask_b32 %10, %10, %16, vcc\n v_cndmask_b32 %11, %11, %16, vcc\n"
                     "v_cndmask_b32 %12, %12, %16, vcc\n v_cndmask_b32 %13, %13, %16, vcc\n v_cndmask_b32 %14, %14, %16, vcc\n"
                     "v_cndmask_b32 %15, %15, %16, vcc\n"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
                       "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                     : "v"(b)
                     : "vcc");
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
UB_KERNEL(k_cmp, , asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc"))
UB_KERNEL(k_readlane, unsigned sacc = 0, { unsigned t; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(t) : "v"(a[i])); sacc ^= t; a[0] ^= sacc & 1u; })

__global__ void k_lshl_add_u64(unsigned *out, int iters, unsigned seed)
{
    unsigned long long w[kUnroll];
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) w[i] = seed + threadIdx.x * 7 + i;
    const unsigned long long b = seed ^ threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) asm volatile("v_lshl_add_u64 %0, %0, 4, %1" : "+v"(w[i]) : "v"(b));
    }
    unsigned long long s = 0;
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) s ^= w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(s ^ (s >> 32));
}

__global__ void k_fma_f64(unsigned *out, int iters, unsigned seed)
{
    double w[kUnroll];
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) w[i] = (double)(seed + threadIdx.x * 7 + i);
    const double b = 1.0000001, c = 1e-9 * seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(w[i]) : "v"(b), "v"(c));
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) s += w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(long long)s;
}

struct ClassEntry {
    const char *name;
    void (*kern)(unsigned *, int, unsigned);
};
static const ClassEntry kClasses[] = {
    {"v_pk_mad_u16", k_pkmad},       {"v_mad_u32_u24", k_mad24},     {"v_fma_f32", k_fma32},          {"v_and_b32", k_and},
    {"v_add_u32", k_add},            {"v_mov_b32", k_mov},           {"v_add_u32_sdwa", k_add_sdwa},  {"v_cvt_f16_u16_sdwa", k_cvt_sdwa},
    {"v_pk_mul_f16", k_pk_mul_f16},  {"v_pk_add_u16", k_pk_add_u16}, {"v_dot2c_f32_f16", k_dot2c},    {"v_alignbit_b32", k_alignbit},
    {"v_cndmask_b32(+1/16 v_cmp)", k_cndmask},    {"v_cmp_lt_u32", k_cmp},        {"v_lshl_add_u64", k_lshl_add_u64}, {"v_readlane_b32", k_readlane},
    {"v_fma_f64", k_fma_f64},       {"v_rsq_f32", k_rsq_f32},       {"v_cvt_f32_u32", k_cvt_f32_u32},
};

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

int main(int argc, char **argv)
{
    const bool classes_only = argc > 1;  // (the --pmc calibration pass: one launch pair per class kernel, nothing else)
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, ncu, p.clockRate);
    unsigned *out;
    CHECK(hipMalloc(&out, (size_t)ncu * 8 * 1024 * sizeof(unsigned)));
    const int iters = 20000;
    // the per-class table bench.py's mix-weighted issue peak is computed from (profiles/r03_valu_classes.json):
    // "CLASS name waves_per_simd G_wave_instr_per_s cycles_per_instr_per_simd"
    for (int wps : {4, 8}) {
        if (classes_only && wps != 8) continue;
        for (const ClassEntry &c : kClasses) {
            const int blocks = ncu * wps;
            const double t = run(c.kern, blocks, 256, iters, out);
            const double n = (double)blocks * 4 * iters * kUnroll;
            printf("CLASS %-20s %d %.1f %.3f\n", c.name, wps, n / t / 1e9, p.clockRate * 1e3 * t * ncu * 4.0 / n);
        }
    }
    if (classes_only) return 0;
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
