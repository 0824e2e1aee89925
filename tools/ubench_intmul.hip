// Micro-benchmark: integer-multiply and BabyBear mulmod throughput on gfx950.
// Decides how the Poseidon2 / NTT inner loops should form their Montgomery products.
//   hipcc --offload-arch=gfx950 -O3 -I raiko_amd/csrc tools/ubench_intmul.hip -o tools/_build/ubench_intmul
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "bb.hpp"
#include "poseidon2_core.hpp"
#include "poseidon2_consts.inc"

constexpr int ITERS = 2048;
constexpr int ILP = 8;

template <int OP>
__global__ __launch_bounds__(256) void k_op(uint32_t* out, uint32_t seed) {
    uint32_t x[ILP], y = seed | 1u;
#pragma unroll
    for (int j = 0; j < ILP; j++) x[j] = threadIdx.x * 2654435761u + j * 40503u + seed;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) {
            if (OP == 0) x[j] = x[j] * (x[(j + 1) % ILP] | 1u); // v_mul_lo_u32 (+v_or)
            if (OP == 1) x[j] = __umulhi(x[j], y);              // v_mul_hi_u32
            if (OP == 2) {                                       // v_mad_u64_u32
                uint64_t t = (uint64_t)x[j] * y + x[(j + 1) % ILP];
                x[j] = (uint32_t)t ^ (uint32_t)(t >> 32);
            }
            if (OP == 3) x[j] = __umul24(x[j], y);              // v_mul_u32_u24
            if (OP == 4) x[j] = x[j] + (x[(j + 1) % ILP] ^ y);  // v_add_u32 + v_xor (full-rate reference)
            if (OP == 5) x[j] = bb::mul(x[j] % bb::P, y % bb::P);  // not used
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc ^= x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_montmul(uint32_t* out, uint32_t seed) {
    uint32_t x[ILP], y = (seed | 1u) % bb::P;
#pragma unroll
    for (int j = 0; j < ILP; j++) x[j] = (threadIdx.x * 2654435761u + j * 40503u + seed) % bb::P;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) x[j] = bb::mul(x[j], y);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc ^= x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_smul(uint32_t* out, uint32_t seed) {
    int32_t x[ILP], y = (int32_t)((seed | 1u) % bb::P);
#pragma unroll
    for (int j = 0; j < ILP; j++) x[j] = (int32_t)((threadIdx.x * 2654435761u + j * 40503u + seed) % bb::P);
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) x[j] = bb::smul(x[j], x[(j + 1) % ILP]);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc ^= (uint32_t)x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + y;
}

__global__ __launch_bounds__(256) void k_fma64(double* out, double seed) {
    double x[ILP], y = seed;
#pragma unroll
    for (int j = 0; j < ILP; j++) x[j] = threadIdx.x + j;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) x[j] = fma(x[j], y, 1.0);
    }
    double acc = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

constexpr int P2_ITERS = 64;
__global__ __launch_bounds__(256) void k_poseidon2(uint32_t* out, const p2::Consts* __restrict__ kc) {
    uint32_t s[p2::CELLS];
#pragma unroll
    for (int i = 0; i < p2::CELLS; i++) s[i] = (threadIdx.x * 977u + i * 131u + blockIdx.x) % bb::P;
    for (int it = 0; it < P2_ITERS; it++) p2::permute(s, *kc);
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < p2::CELLS; i++) acc ^= s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// same loop, stamped: shader clock (s_memtime) against the 100 MHz constant clock gives the
// clock the chip actually holds under this integer load (stamps go to their own buffer)
__global__ __launch_bounds__(256) void k_poseidon2_clk(uint32_t* out, const p2::Consts* __restrict__ kc,
                                                       unsigned long long* stamps) {
    uint32_t s[p2::CELLS];
#pragma unroll
    for (int i = 0; i < p2::CELLS; i++) s[i] = (threadIdx.x * 977u + i * 131u + blockIdx.x) % bb::P;
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < P2_ITERS; it++) p2::permute(s, *kc);
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < p2::CELLS; i++) acc ^= s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = w1 - w0;
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <typename F>
float time_ms(F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();  // warm-up
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms / 5;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    const int blocks = cus * 8, threads = 256;
    uint32_t* out; double* outd;
    CK(hipMalloc(&out, (size_t)blocks * threads * 4));
    CK(hipMalloc(&outd, (size_t)blocks * threads * 8));
    p2::Consts h; p2::Consts* d;
    memcpy(h.rc_ext, P2_RC_EXT_MONT, sizeof h.rc_ext);
    memcpy(h.rc_int, P2_RC_INT_MONT, sizeof h.rc_int);
    memcpy(h.diag, P2_INT_DIAG_MONT, sizeof h.diag);
    p2::derive(h);
    CK(hipMalloc(&d, sizeof h));
    CK(hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice));
    double nops = (double)blocks * threads * ITERS * ILP;
    const char* names[] = {"v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32(+xor)", "v_mul_u32_u24", "v_add_u32"};
    float ms;
    ms = time_ms([&] { hipLaunchKernelGGL(k_op<0>, dim3(blocks), dim3(threads), 0, 0, out, 12345u); });
    printf("%-22s %8.3f ms  %8.1f Gop/s  %.2f op/clk/CU@2.4GHz\n", names[0], ms, nops / ms / 1e6, nops / ms / 1e6 / cus / 2.4);
    ms = time_ms([&] { hipLaunchKernelGGL(k_op<1>, dim3(blocks), dim3(threads), 0, 0, out, 12345u); });
    printf("%-22s %8.3f ms  %8.1f Gop/s  %.2f op/clk/CU@2.4GHz\n", names[1], ms, nops / ms / 1e6, nops / ms / 1e6 / cus / 2.4);
    ms = time_ms([&] { hipLaunchKernelGGL(k_op<2>, dim3(blocks), dim3(threads), 0, 0, out, 12345u); });
    printf("%-22s %8.3f ms  %8.1f Gop/s  %.2f op/clk/CU@2.4GHz\n", names[2], ms, nops / ms / 1e6, nops / ms / 1e6 / cus / 2.4);
    ms = time_ms([&] { hipLaunchKernelGGL(k_op<3>, dim3(blocks), dim3(threads), 0, 0, out, 12345u); });
    printf("%-22s %8.3f ms  %8.1f Gop/s  %.2f op/clk/CU@2.4GHz\n", names[3], ms, nops / ms / 1e6, nops / ms / 1e6 / cus / 2.4);
    ms = time_ms([&] { hipLaunchKernelGGL(k_op<4>, dim3(blocks), dim3(threads), 0, 0, out, 12345u); });
    printf("%-22s %8.3f ms  %8.1f Gop/s  %.2f op/clk/CU@2.4GHz\n", names[4], ms, nops / ms / 1e6, nops / ms / 1e6 / cus / 2.4);
    ms = time_ms([&] { hipLaunchKernelGGL(k_montmul, dim3(blocks), dim3(threads), 0, 0, out, 12345u); });
    printf("%-22s %8.3f ms  %8.1f Gmulmod/s  %.2f /clk/CU@2.4GHz\n", "bb::mul (Montgomery)", ms, nops / ms / 1e6, nops / ms / 1e6 / cus / 2.4);
    ms = time_ms([&] { hipLaunchKernelGGL(k_smul, dim3(blocks), dim3(threads), 0, 0, out, 12345u); });
    printf("%-22s %8.3f ms  %8.1f Gmulmod/s  %.2f /clk/CU@2.4GHz\n", "bb::smul (signed, var)", ms, nops / ms / 1e6, nops / ms / 1e6 / cus / 2.4);
    ms = time_ms([&] { hipLaunchKernelGGL(k_fma64, dim3(blocks), dim3(threads), 0, 0, outd, 1.0000001); });
    printf("%-22s %8.3f ms  %8.1f Gop/s  %.2f op/clk/CU@2.4GHz\n", "v_fma_f64", ms, nops / ms / 1e6, nops / ms / 1e6 / cus / 2.4);
    double nperm = (double)blocks * threads * P2_ITERS;
    ms = time_ms([&] { hipLaunchKernelGGL(k_poseidon2, dim3(blocks), dim3(threads), 0, 0, out, d); });
    printf("%-22s %8.3f ms  %8.3f Gperm/s  (%.1f ns/perm/CU-lane)\n", "poseidon2 t=24 permute", ms, nperm / ms / 1e6, ms * 1e6 / P2_ITERS);
    {
        unsigned long long* d_st;
        CK(hipMalloc(&d_st, (size_t)blocks * 16));
        for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k_poseidon2, dim3(blocks), dim3(threads), 0, 0, out, d);  // heat up
        hipLaunchKernelGGL(k_poseidon2_clk, dim3(blocks), dim3(threads), 0, 0, out, d, d_st);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> st((size_t)blocks * 2);
        CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> mhz;
        double cyc = 0;
        for (int b = 0; b < blocks; b++) {
            if (st[2 * b + 1]) mhz.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 100.0);
            cyc += (double)st[2 * b];
        }
        std::sort(mhz.begin(), mhz.end());
        printf("in-kernel clock under poseidon2 load: median %.0f MHz (min %.0f, max %.0f); %.0f shader cycles per wave-permutation-block\n",
               mhz[mhz.size() / 2], mhz.front(), mhz.back(), cyc / blocks / P2_ITERS);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
