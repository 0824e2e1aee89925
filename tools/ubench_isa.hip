// Per-instruction VALU throughput on gfx950 by inline asm (no compiler folding): 8 independent
// dependency chains per lane, 8 waves per SIMD resident.  Prints cycles per wave-instruction
// per SIMD at the measured in-kernel clock.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_isa.hip -o tools/_build/ubench_isa
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int ITERS = 4096;

#define KERNEL3(NAME, ASM)                                                                        \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned long long* st, unsigned y) { \
        unsigned x0 = threadIdx.x + 1, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 11, x5 = x0 * 13,  \
                 x6 = x0 * 17, x7 = x0 * 19;                                                      \
        unsigned long long c0 = clock64(), w0 = wall_clock64();                                   \
        for (int i = 0; i < ITERS; i++) {                                                         \
            asm volatile(ASM " %0, %0, %8\n" ASM " %1, %1, %8\n" ASM " %2, %2, %8\n" ASM " %3, %3, %8\n" \
                         ASM " %4, %4, %8\n" ASM " %5, %5, %8\n" ASM " %6, %6, %8\n" ASM " %7, %7, %8\n" \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
                         : "v"(y));                                                               \
        }                                                                                         \
        unsigned long long c1 = clock64(), w1 = wall_clock64();                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;       \
        if (threadIdx.x == 0) { st[2 * blockIdx.x] = c1 - c0; st[2 * blockIdx.x + 1] = w1 - w0; }  \
    }

KERNEL3(k_add, "v_add_u32")
KERNEL3(k_min, "v_min_u32")
KERNEL3(k_mul_lo, "v_mul_lo_u32")
KERNEL3(k_mul_hi_u, "v_mul_hi_u32")
KERNEL3(k_mul_hi_i, "v_mul_hi_i32")
KERNEL3(k_mul_u24, "v_mul_u32_u24")
KERNEL3(k_mul_hi_u24, "v_mul_hi_u32_u24")
KERNEL3(k_sub, "v_sub_u32")
KERNEL3(k_lshl, "v_lshlrev_b32")
KERNEL3(k_max, "v_max_i32")

// same with the second source in an SGPR / as a 32-bit literal (how compiled code feeds constants)
#define KERNELS(NAME, ASM)                                                                        \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned long long* st, unsigned y) { \
        unsigned x0 = threadIdx.x + 1, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 11, x5 = x0 * 13,  \
                 x6 = x0 * 17, x7 = x0 * 19;                                                      \
        unsigned long long c0 = clock64(), w0 = wall_clock64();                                   \
        for (int i = 0; i < ITERS; i++) {                                                         \
            asm volatile(ASM " %0, %0, %8\n" ASM " %1, %1, %8\n" ASM " %2, %2, %8\n" ASM " %3, %3, %8\n" \
                         ASM " %4, %4, %8\n" ASM " %5, %5, %8\n" ASM " %6, %6, %8\n" ASM " %7, %7, %8\n" \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
                         : "s"(y));                                                               \
        }                                                                                         \
        unsigned long long c1 = clock64(), w1 = wall_clock64();                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;       \
        if (threadIdx.x == 0) { st[2 * blockIdx.x] = c1 - c0; st[2 * blockIdx.x + 1] = w1 - w0; }  \
    }
KERNELS(k_add_s, "v_add_u32")
KERNELS(k_mul_lo_s, "v_mul_lo_u32")
KERNELS(k_mul_hi_u_s, "v_mul_hi_u32")
KERNELS(k_mul_hi_i_s, "v_mul_hi_i32")
#define KERNELL(NAME, ASM)                                                                        \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned long long* st, unsigned y) { \
        unsigned x0 = threadIdx.x + 1, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 11, x5 = x0 * 13,  \
                 x6 = x0 * 17, x7 = x0 * 19;                                                      \
        unsigned long long c0 = clock64(), w0 = wall_clock64();                                   \
        for (int i = 0; i < ITERS; i++) {                                                         \
            asm volatile(ASM " %0, 0x87ffffff, %0\n" ASM " %1, 0x87ffffff, %1\n" ASM " %2, 0x87ffffff, %2\n" \
                         ASM " %3, 0x87ffffff, %3\n" ASM " %4, 0x87ffffff, %4\n" ASM " %5, 0x87ffffff, %5\n" \
                         ASM " %6, 0x87ffffff, %6\n" ASM " %7, 0x87ffffff, %7\n"                  \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); \
        }                                                                                         \
        unsigned long long c1 = clock64(), w1 = wall_clock64();                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ y;   \
        if (threadIdx.x == 0) { st[2 * blockIdx.x] = c1 - c0; st[2 * blockIdx.x + 1] = w1 - w0; }  \
    }
KERNELL(k_add_lit, "v_add_u32")
// one dependent chain per lane: issue-to-use latency when run with ONE wave per SIMD
#define KERNELDEP(NAME, ASM)                                                                      \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned long long* st, unsigned y) { \
        unsigned x0 = threadIdx.x + 1;                                                            \
        unsigned long long c0 = clock64(), w0 = wall_clock64();                                   \
        for (int i = 0; i < ITERS; i++) {                                                         \
            asm volatile(ASM " %0, %0, %1\n" ASM " %0, %0, %1\n" ASM " %0, %0, %1\n" ASM " %0, %0, %1\n" \
                         ASM " %0, %0, %1\n" ASM " %0, %0, %1\n" ASM " %0, %0, %1\n" ASM " %0, %0, %1\n" \
                         : "+v"(x0) : "v"(y));                                                    \
        }                                                                                         \
        unsigned long long c1 = clock64(), w1 = wall_clock64();                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0;                                          \
        if (threadIdx.x == 0) { st[2 * blockIdx.x] = c1 - c0; st[2 * blockIdx.x + 1] = w1 - w0; }  \
    }
KERNELDEP(k_dep_add, "v_add_u32")
KERNELDEP(k_dep_mul_lo, "v_mul_lo_u32")
KERNELDEP(k_dep_mul_hi, "v_mul_hi_u32")

// straight-line body of 8*REPT instructions per loop trip: does throughput survive when the
// loop no longer fits the wave's instruction buffer (instruction fetch / I-cache bound)?
#define KERNELBIG(NAME, ASM, REPT, LOOPS)                                                         \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned long long* st, unsigned y) { \
        unsigned x0 = threadIdx.x + 1, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 11, x5 = x0 * 13,  \
                 x6 = x0 * 17, x7 = x0 * 19;                                                      \
        unsigned long long c0 = clock64(), w0 = wall_clock64();                                   \
        for (int i = 0; i < LOOPS; i++) {                                                         \
            asm volatile(".rept " #REPT "\n" ASM " %0, %0, %8\n" ASM " %1, %1, %8\n" ASM " %2, %2, %8\n" \
                         ASM " %3, %3, %8\n" ASM " %4, %4, %8\n" ASM " %5, %5, %8\n" ASM " %6, %6, %8\n" \
                         ASM " %7, %7, %8\n.endr\n"                                                \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
                         : "v"(y));                                                               \
        }                                                                                         \
        unsigned long long c1 = clock64(), w1 = wall_clock64();                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;       \
        if (threadIdx.x == 0) { st[2 * blockIdx.x] = c1 - c0; st[2 * blockIdx.x + 1] = w1 - w0; }  \
    }
KERNELBIG(k_big_mul_128, "v_mul_hi_u32", 16, 256)
KERNELBIG(k_big_mul_1k, "v_mul_hi_u32", 128, 32)
KERNELBIG(k_big_mul_4k, "v_mul_hi_u32", 512, 8)
KERNELBIG(k_big_min_4k, "v_min_u32", 512, 8)
KERNELBIG(k_big_add_4k, "v_add_u32", 512, 8)

// 64-bit mad: dst pair, src0, src1, 64-bit addend (the chain runs through the addend)
#define KERNELMAD(NAME, ASM)                                                                      \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned long long* st, unsigned y) { \
        unsigned long long x0 = threadIdx.x + 1, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7;           \
        unsigned long long c0 = clock64(), w0 = wall_clock64();                                   \
        for (int i = 0; i < ITERS; i++) {                                                         \
            asm volatile(ASM " %0, vcc, %4, %4, %0\n" ASM " %1, vcc, %4, %4, %1\n"                 \
                         ASM " %2, vcc, %4, %4, %2\n" ASM " %3, vcc, %4, %4, %3\n"                 \
                         ASM " %0, vcc, %4, %4, %0\n" ASM " %1, vcc, %4, %4, %1\n"                 \
                         ASM " %2, vcc, %4, %4, %2\n" ASM " %3, vcc, %4, %4, %3\n"                 \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)                                 \
                         : "v"(y)                                                                 \
                         : "vcc");                                                                \
        }                                                                                         \
        unsigned long long c1 = clock64(), w1 = wall_clock64();                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(x0 ^ x1 ^ x2 ^ x3);               \
        if (threadIdx.x == 0) { st[2 * blockIdx.x] = c1 - c0; st[2 * blockIdx.x + 1] = w1 - w0; }  \
    }
KERNELMAD(k_mad_u64, "v_mad_u64_u32")
// 64-bit shift-add: d = (a << k) + b on register pairs
#define KERNELLSHL(NAME, SH)                                                                      \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned long long* st, unsigned y) { \
        unsigned long long x0 = threadIdx.x + 1, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, z = y;     \
        unsigned long long c0 = clock64(), w0 = wall_clock64();                                   \
        for (int i = 0; i < ITERS; i++) {                                                         \
            asm volatile("v_lshl_add_u64 %0, %0, " #SH ", %4\n v_lshl_add_u64 %1, %1, " #SH ", %4\n"  \
                         "v_lshl_add_u64 %2, %2, " #SH ", %4\n v_lshl_add_u64 %3, %3, " #SH ", %4\n"  \
                         "v_lshl_add_u64 %0, %0, " #SH ", %4\n v_lshl_add_u64 %1, %1, " #SH ", %4\n"  \
                         "v_lshl_add_u64 %2, %2, " #SH ", %4\n v_lshl_add_u64 %3, %3, " #SH ", %4\n"  \
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(z));                       \
        }                                                                                         \
        unsigned long long c1 = clock64(), w1 = wall_clock64();                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(x0 ^ x1 ^ x2 ^ x3);               \
        if (threadIdx.x == 0) { st[2 * blockIdx.x] = c1 - c0; st[2 * blockIdx.x + 1] = w1 - w0; }  \
    }
KERNELLSHL(k_lshl_add_u64_0, 0)
KERNELLSHL(k_lshl_add_u64_2, 2)
// mad with an inline-constant multiplier (5*x + acc)
__global__ __launch_bounds__(256) void k_mad_u64_lit(unsigned* out, unsigned long long* st, unsigned y) {
    unsigned long long x0 = threadIdx.x + 1, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7;
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < ITERS; i++) {
        asm volatile("v_mad_u64_u32 %0, vcc, %4, 5, %0\n v_mad_u64_u32 %1, vcc, %4, 7, %1\n"
                     "v_mad_u64_u32 %2, vcc, %4, 3, %2\n v_mad_u64_u32 %3, vcc, %4, 1, %3\n"
                     "v_mad_u64_u32 %0, vcc, %4, 5, %0\n v_mad_u64_u32 %1, vcc, %4, 7, %1\n"
                     "v_mad_u64_u32 %2, vcc, %4, 3, %2\n v_mad_u64_u32 %3, vcc, %4, 1, %3\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y) : "vcc");
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(x0 ^ x1 ^ x2 ^ x3);
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = c1 - c0; st[2 * blockIdx.x + 1] = w1 - w0; }
}
KERNELMAD(k_mad_i64, "v_mad_i64_i32")

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename K>
int run(const char* name, K kern, int cus, int blocks_per_cu = 64) {
    // many workgroups per CU: steady-state throughput from the host clock (hipEvent), no tail effects
    const int blocks = cus * blocks_per_cu, threads = 256;
    unsigned* out;
    unsigned long long* st;
    CK(hipMalloc(&out, (size_t)blocks * threads * 4));
    CK(hipMalloc(&st, (size_t)blocks * 16));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, st, 12345u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, st, 12345u);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    double ops = (double)blocks * threads * ITERS * 8.0;
    printf("%-28s %8.3f ms  %7.2f Tlane-op/s  = %5.1f lanes/clk/CU at 2.4 GHz\n", name, ms, ops / ms / 1e9,
           ops / ms / 1e9 * 1e12 / cus / 2.4e9);
    CK(hipFree(out));
    CK(hipFree(st));
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("%d CUs\n", cus);
    run("v_add_u32", k_add, cus);
    run("v_min_u32", k_min, cus);
    run("v_sub_u32", k_sub, cus);
    run("v_lshlrev_b32", k_lshl, cus);
    run("v_max_i32", k_max, cus);
    run("v_mul_lo_u32", k_mul_lo, cus);
    run("v_mul_hi_u32", k_mul_hi_u, cus);
    run("v_mul_hi_i32", k_mul_hi_i, cus);
    run("v_mul_u32_u24", k_mul_u24, cus);
    run("v_mul_hi_u32_u24", k_mul_hi_u24, cus);
    run("v_add_u32 (sgpr src)", k_add_s, cus);
    run("v_add_u32 (literal)", k_add_lit, cus);
    run("v_mul_lo_u32 (sgpr)", k_mul_lo_s, cus);
    run("v_mul_hi_u32 (sgpr)", k_mul_hi_u_s, cus);
    run("v_mul_hi_i32 (sgpr)", k_mul_hi_i_s, cus);
    printf("straight-line bodies (ITERS*8 instructions per wave in all cases):\n");
    run("mul_hi body 128 instr", k_big_mul_128, cus);
    run("mul_hi body 1024 instr", k_big_mul_1k, cus);
    run("mul_hi body 4096 instr", k_big_mul_4k, cus);
    run("v_min  body 4096 instr", k_big_min_4k, cus);
    run("v_add  body 4096 instr", k_big_add_4k, cus);
    printf("dependent chains, ONE wave per SIMD (cycles = issue-to-use latency):\n");
    run("dep v_add_u32", k_dep_add, cus);
    run("dep v_mul_lo_u32", k_dep_mul_lo, cus);
    run("dep v_mul_hi_u32", k_dep_mul_hi, cus);
    printf("dependent chains, 4 waves per SIMD:\n");
    run("v_mad_u64_u32", k_mad_u64, cus);
    run("v_mad_u64_u32 (inline const)", k_mad_u64_lit, cus);
    run("v_lshl_add_u64 (shift 0)", k_lshl_add_u64_0, cus);
    run("v_lshl_add_u64 (shift 2)", k_lshl_add_u64_2, cus);
    run("v_mad_i64_i32", k_mad_i64, cus);
    return 0;
}
