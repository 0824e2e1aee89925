// Residency census: how many 256-thread workgroups are co-resident per CU for a small kernel,
// and what a v_mul_hi_u32 stream costs per SIMD at exactly that residency.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
constexpr int ITERS = 4096;
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned long long* st, unsigned y) {
    unsigned x0 = threadIdx.x + 1, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 11, x5 = x0 * 13, x6 = x0 * 17, x7 = x0 * 19;
    unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < ITERS; i++) {
        asm volatile("v_mul_hi_u32 %0, %0, %8\nv_mul_hi_u32 %1, %1, %8\nv_mul_hi_u32 %2, %2, %8\nv_mul_hi_u32 %3, %3, %8\n"
                     "v_mul_hi_u32 %4, %4, %8\nv_mul_hi_u32 %5, %5, %8\nv_mul_hi_u32 %6, %6, %8\nv_mul_hi_u32 %7, %7, %8\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
    if (threadIdx.x == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        st[4 * blockIdx.x] = w0; st[4 * blockIdx.x + 1] = w1; st[4 * blockIdx.x + 2] = c1 - c0;
        st[4 * blockIdx.x + 3] = ((unsigned long long)xcc << 32) | hwid;
    }
}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    for (int per_cu : {1, 2, 4, 8, 16}) {
        int blocks = cus * per_cu;
        unsigned* out; unsigned long long* st;
        hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&st, (size_t)blocks * 32);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, st, 12345u);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, st, 12345u);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)blocks * 4);
        hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int b = 0; b < blocks; b++) { t0 = std::min(t0, h[4 * b]); t1 = std::max(t1, h[4 * b + 1]); }
        // concurrency at 25% of the kernel's span
        unsigned long long probe = t0 + (t1 - t0) / 4;
        int live = 0; double cyc = 0;
        std::vector<int> percu(1 << 16, 0);
        for (int b = 0; b < blocks; b++) {
            cyc += (double)h[4 * b + 2];
            if (h[4 * b] <= probe && h[4 * b + 1] >= probe) {
                live++;
                unsigned hwid = (unsigned)h[4 * b + 3], xcc = (unsigned)(h[4 * b + 3] >> 32);
                unsigned cu = (hwid >> 8) & 0xf, sh = (hwid >> 12) & 1, se = (hwid >> 13) & 0x7;
                percu[(xcc & 15) << 8 | se << 5 | sh << 4 | cu]++;
            }
        }
        int maxcu = 0, used = 0;
        for (int v : percu) { maxcu = std::max(maxcu, v); used += v > 0; }
        cyc /= blocks;
        printf("%2d blocks/CU launched: span %.3f ms, live at 25%%: %d blocks (%.2f per CU; %d distinct CU ids, max %d on one), "
               "wave cycles/instr %.2f -> per SIMD at live residency %.2f\n",
               per_cu, (t1 - t0) / 100000.0, live, (double)live / cus, used, maxcu, cyc / (ITERS * 8.0),
               cyc / (ITERS * 8.0) / ((double)live / cus));
        hipFree(out); hipFree(st);
    }
    return 0;
}
