// VERDICT round-1 item 9, the bounded experiment: could the linear layers of Poseidon2 run on byte limbs through
// v_dot4_u32_u8 or the int8 matrix cores?  Measures on gfx950, at 8 waves per SIMD with 8 independent chains:
//   * v_mad_u64_u32 (what a 32 x 32 -> 64 product costs today: one instruction),
//   * v_dot4_u32_u8 (four byte products per instruction: a 32 x 32 product is 16 byte products = 4 of these
//     plus the shifts that recombine seven partial sums),
//   * v_mfma_i32_16x16x64_i8 (16384 byte MACs per wave instruction).
// and prints what the external layer of one width-24 permutation would cost in each form.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_dot4_mfma.hip -o tools/_build/ubench_dot4_mfma
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int ITERS = 4096;
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mad(unsigned* out, unsigned long long* st, unsigned y) {
    unsigned long long x[8];
    for (int j = 0; j < 8; j++) x[j] = threadIdx.x * (2 * j + 3) + 1;
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[j]) : "v"(y), "v"((unsigned)x[j]) : "vcc");
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    unsigned acc = 0;
    for (int j = 0; j < 8; j++) acc ^= (unsigned)x[j] ^ (unsigned)(x[j] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = c1 - c0; st[2 * blockIdx.x + 1] = w1 - w0; }
}

__global__ __launch_bounds__(256) void k_dot4(unsigned* out, unsigned long long* st, unsigned y) {
    unsigned x[8];
    for (int j = 0; j < 8; j++) x[j] = threadIdx.x * (2 * j + 3) + 1;
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(x[j]) : "v"(y));
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    unsigned acc = 0;
    for (int j = 0; j < 8; j++) acc ^= x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = c1 - c0; st[2 * blockIdx.x + 1] = w1 - w0; }
}

// 4 independent accumulators per wave: A, B = 16 bytes per lane (the 16x16x64 form), C = 4 x i32 per lane
__global__ __launch_bounds__(256) void k_mfma(unsigned* out, unsigned long long* st, unsigned y) {
    v4i a = {(int)threadIdx.x, (int)y, 3, 4}, b = {5, (int)threadIdx.x, 7, (int)y};
    v4i c[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) c[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[j], 0, 0, 0);
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    unsigned acc = 0;
    for (int j = 0; j < 4; j++) acc ^= (unsigned)(c[j].x ^ c[j].y ^ c[j].z ^ c[j].w);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = c1 - c0; st[2 * blockIdx.x + 1] = w1 - w0; }
}

template <class K>
double run(K kernel, const char* name, int per_iter, int cus, double* cycles_per_instr) {
    const int blocks = cus * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    unsigned* out;
    unsigned long long* st;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMalloc(&st, (size_t)blocks * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, st, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, st, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks * 4 waves * ITERS * per_iter / (cus * 4 SIMDs)
    const double wave_instr_per_simd = (double)blocks * 4 * ITERS * per_iter / (cus * 4.0);
    const double cyc = ms * 1e-3 * 2.4e9 / wave_instr_per_simd;
    printf("%-28s %8.3f ms   %6.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, cyc);
    *cycles_per_instr = cyc;
    hipFree(out);
    hipFree(st);
    return ms;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs\n", p.gcnArchName, cus);
    double mad, dot, mfma;
    run(k_mad, "v_mad_u64_u32", 8, cus, &mad);
    run(k_dot4, "v_dot4_u32_u8", 8, cus, &dot);
    run(k_mfma, "v_mfma_i32_16x16x64_i8", 4, cus, &mfma);
    // the external layer of one width-24 permutation = a 24 x 24 matrix of one-byte entries times a state of
    // 31-bit cells (4 byte limbs each).  Today: ~110 VALU instructions per lane (one permutation per lane).
    const double valu_now = 110 * 4.0 / 64;  // cycles per permutation per SIMD at 4 cycles per wave-instruction
    // dot4 form: per output cell 4 limb sums x 6 dot4 (24 inputs / 4) = 24 dot4 + 7 recombination -> 24 * 31
    const double dot_form = 24 * 31 * dot / 64;
    // MFMA form: 16 permutations per 16x16 tile column block; rows 24 -> 2 tiles; K = 24 cells -> one x64 op is 37 % full;
    // 4 limb planes: 2 * 4 = 8 MFMAs per 16 permutations, plus >= 10 VALU per cell to split and recombine limbs
    const double mfma_form = 8 * mfma / 16 + 24 * 10 * 4.0 / 64;
    printf("external layer, cycles per permutation per SIMD: VALU as shipped %.1f | v_dot4 limbs %.1f | int8 MFMA %.1f "
           "(of which %.1f on the matrix core, the rest VALU limb handling)\n",
           valu_now, dot_form, mfma_form, 8 * mfma / 16);
    return 0;
}
