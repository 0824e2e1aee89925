// Poseidon2 permutation throughput harness: runs p2::permute (the product's shared header) at a
// chosen residency and prints cycles per wave-permutation per SIMD, so code variants of
// poseidon2_core.hpp / bb.hpp can be A/B-ed in one gpurun call.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I raiko_amd/csrc [-DP2_...] tools/ubench_p2.hip -o tools/_build/ubench_p2
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "bb.hpp"
#include "poseidon2_core.hpp"
#include "poseidon2_consts.inc"

constexpr int P2_ITERS = 32;

__global__ __launch_bounds__(256) void k_perm(uint32_t* out, const p2::Consts* __restrict__ kc,
                                              unsigned long long* stamps) {
    uint32_t s[p2::CELLS];
#pragma unroll
    for (int i = 0; i < p2::CELLS; i++) s[i] = (threadIdx.x * 977u + i * 131u + blockIdx.x) % bb::P;
    unsigned long long c0 = clock64();
    for (int it = 0; it < P2_ITERS; it++) p2::permute(s, *kc);
    unsigned long long c1 = clock64();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < p2::CELLS; i++) acc ^= s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) stamps[blockIdx.x] = c1 - c0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void*)k_perm));
    printf("k_perm: %d VGPRs, %zu B scratch\n", fa.numRegs, (size_t)fa.localSizeBytes);
    p2::Consts h;
    p2::Consts* d;
    memcpy(h.rc_ext, P2_RC_EXT_MONT, sizeof h.rc_ext);
    memcpy(h.rc_int, P2_RC_INT_MONT, sizeof h.rc_int);
    memcpy(h.diag, P2_INT_DIAG_MONT, sizeof h.diag);
    p2::derive(h);
    CK(hipMalloc(&d, sizeof h));
    CK(hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice));
    for (int per_cu : {1, 2, 4, 8}) {
        int blocks = cus * per_cu;
        uint32_t* out;
        unsigned long long* st;
        CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
        CK(hipMalloc(&st, (size_t)blocks * 8));
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        hipLaunchKernelGGL(k_perm, dim3(blocks), dim3(256), 0, 0, out, d, st);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_perm, dim3(blocks), dim3(256), 0, 0, out, d, st);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        std::vector<unsigned long long> hs(blocks);
        CK(hipMemcpy(hs.data(), st, (size_t)blocks * 8, hipMemcpyDeviceToHost));
        std::sort(hs.begin(), hs.end());
        double wave_cycles = (double)hs[blocks / 2] / P2_ITERS;  // latency of one permutation for a wave
        printf("%d wave(s)/SIMD: %8.3f ms, %7.0f cycles per permutation per wave, %7.0f SIMD-cycles per wave-permutation, "
               "%.3f Gperm/s\n",
               per_cu, ms, wave_cycles, wave_cycles / per_cu, (double)blocks * 256 * P2_ITERS / ms / 1e6);
        CK(hipFree(out));
        CK(hipFree(st));
    }
    return 0;
}
