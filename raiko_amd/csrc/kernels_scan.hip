// Witness-side helpers of the Hal trait (risc0-zkp 1.0.1 hal/mod.rs): `prefix_products` (the
// running product behind the accum group's grand-product columns) and `scatter` (writes of the
// executor's sparse per-cycle values into the dense witness).  Both are HBM-bound.
#include "internal.hpp"

#include <cstring>

namespace {

using bb::Ext;
constexpr int SCAN_TPB = 256;
constexpr int SCAN_CH = 8;                        // consecutive elements per lane
constexpr int SCAN_BLOCK = SCAN_TPB * SCAN_CH;    // elements per workgroup

__device__ __forceinline__ Ext load_ext(const uint32_t* p) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    return Ext{{v.x, v.y, v.z, v.w}};
}
__device__ __forceinline__ void store_ext(uint32_t* p, const Ext& e) {
    *reinterpret_cast<uint4*>(p) = make_uint4(e.c[0], e.c[1], e.c[2], e.c[3]);
}

// inclusive scan of one value per lane across the workgroup (Hillis-Steele through LDS)
__device__ Ext block_scan(Ext v, Ext* sh, uint32_t wm) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int d = 1; d < SCAN_TPB; d <<= 1) {
        Ext o = t >= d ? sh[t - d] : bb::ext_one();
        __syncthreads();
        if (t >= d) {
            v = bb::mul(o, v, wm);
            sh[t] = v;
        }
        __syncthreads();
    }
    return v;
}

// product of each workgroup's SCAN_BLOCK elements
__global__ void scan_totals_kernel(uint32_t* totals, const uint32_t* io, size_t count, uint32_t wm) {
    __shared__ Ext sh[SCAN_TPB];
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_CH;
    Ext acc = bb::ext_one();
#pragma unroll
    for (int j = 0; j < SCAN_CH; j++)
        if (base + j < count) acc = bb::mul(acc, load_ext(io + (base + j) * 4), wm);
    acc = block_scan(acc, sh, wm);
    if (threadIdx.x == SCAN_TPB - 1) store_ext(totals + (size_t)blockIdx.x * 4, acc);
}
// in-place inclusive scan of n totals by ONE workgroup: lanes take contiguous runs
__global__ void scan_carry_kernel(uint32_t* totals, size_t n, uint32_t wm) {
    __shared__ Ext sh[SCAN_TPB];
    size_t per = (n + SCAN_TPB - 1) / SCAN_TPB;
    size_t lo = (size_t)threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    Ext acc = bb::ext_one();
    for (size_t i = lo; i < hi; i++) acc = bb::mul(acc, load_ext(totals + i * 4), wm);
    Ext incl = block_scan(acc, sh, wm);
    __syncthreads();
    sh[threadIdx.x] = incl;
    __syncthreads();
    Ext run = threadIdx.x ? sh[threadIdx.x - 1] : bb::ext_one();
    for (size_t i = lo; i < hi; i++) {
        run = bb::mul(run, load_ext(totals + i * 4), wm);
        store_ext(totals + i * 4, run);
    }
}
// final pass: carry-in of the workgroup (scanned totals of the groups before it) times the local scan
__global__ void scan_apply_kernel(uint32_t* io, const uint32_t* totals, size_t count, uint32_t wm) {
    __shared__ Ext sh[SCAN_TPB];
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_CH;
    Ext v[SCAN_CH];
    Ext acc = bb::ext_one();
#pragma unroll
    for (int j = 0; j < SCAN_CH; j++) {
        v[j] = base + j < count ? load_ext(io + (base + j) * 4) : bb::ext_one();
        acc = bb::mul(acc, v[j], wm);
        v[j] = acc;
    }
    Ext incl = block_scan(acc, sh, wm);
    __syncthreads();
    sh[threadIdx.x] = incl;
    __syncthreads();
    Ext carry = blockIdx.x ? load_ext(totals + ((size_t)blockIdx.x - 1) * 4) : bb::ext_one();
    if (threadIdx.x) carry = bb::mul(carry, sh[threadIdx.x - 1], wm);
#pragma unroll
    for (int j = 0; j < SCAN_CH; j++)
        if (base + j < count) store_ext(io + (base + j) * 4, bb::mul(carry, v[j], wm));
}

// scatter with "last write wins": stamp = 1 + the largest entry index targeting each word
__global__ void scatter_stamp_kernel(uint32_t* stamp, const uint32_t* offsets, size_t n) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; k < n; k += st) atomicMax(&stamp[offsets[k]], (uint32_t)k + 1);
}
__global__ void scatter_write_kernel(uint32_t* into, const uint32_t* stamp, const uint32_t* offsets,
                                     const uint32_t* values, size_t n) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; k < n; k += st)
        if (stamp[offsets[k]] == (uint32_t)k + 1) into[offsets[k]] = values[k];
}

}  // namespace

namespace rk {

int prefix_products(rk_ctx* ctx, uint32_t* d_io_ext, size_t count) {
    if (count <= 1) return RK_OK;
    if (((uintptr_t)d_io_ext & 15) != 0) return RK_ERR_INVALID;
    size_t blocks = (count + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (blocks > 0x7fffffffu) return RK_ERR_INVALID;
    void* totals = nullptr;
    RK_TRY(dev_alloc(ctx, blocks * 16, &totals));
    int st = RK_OK;
    do {
        KTimer kt(ctx, RK_KCLASS_POLY, (double)count * 48);
        if (blocks > 1) {
            hipLaunchKernelGGL(scan_totals_kernel, dim3((unsigned)blocks), dim3(SCAN_TPB), 0, ctx->stream,
                               (uint32_t*)totals, d_io_ext, count, ctx->sys.wm);
            st = post_launch(ctx, "scan_totals_kernel");
            if (st != RK_OK) break;
            hipLaunchKernelGGL(scan_carry_kernel, dim3(1), dim3(SCAN_TPB), 0, ctx->stream, (uint32_t*)totals, blocks, ctx->sys.wm);
            st = post_launch(ctx, "scan_carry_kernel");
            if (st != RK_OK) break;
        }
        hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)blocks), dim3(SCAN_TPB), 0, ctx->stream, d_io_ext,
                           (const uint32_t*)totals, count, ctx->sys.wm);
        st = post_launch(ctx, "scan_apply_kernel");
    } while (0);
    dev_free(ctx, totals);  // stream-ordered reuse within this ctx
    return st;
}

int scatter(rk_ctx* ctx, uint32_t* d_into, size_t into_words, const uint32_t* h_index, size_t n_cycles,
            const uint32_t* h_offsets, const uint32_t* h_values) {
    if (n_cycles == 0) return RK_OK;
    for (size_t c = 0; c < n_cycles; c++)
        if (h_index[c + 1] < h_index[c]) return RK_ERR_INVALID;
    const size_t lo = h_index[0], n = h_index[n_cycles] - lo;
    if (n == 0) return RK_OK;
    if (into_words > 0xffffffffull) return RK_ERR_INVALID;
    for (size_t k = 0; k < n; k++)
        if (h_offsets[lo + k] >= into_words) return RK_ERR_INVALID;
    void *d_pack = nullptr, *d_stamp = nullptr;
    RK_TRY(dev_alloc(ctx, n * 8, &d_pack));
    int st = dev_alloc(ctx, into_words * 4, &d_stamp);
    if (st != RK_OK) {
        dev_free(ctx, d_pack);
        return st;
    }
    uint32_t* d_off = (uint32_t*)d_pack;
    uint32_t* d_val = d_off + n;
    do {
        hipError_t e = hipMemcpyAsync(d_off, h_offsets + lo, n * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_val, h_values + lo, n * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_stamp, 0, into_words * 4, ctx->stream);
        if (e != hipSuccess) {
            ctx->last_error = std::string("scatter h2d: ") + hipGetErrorString(e);
            st = RK_ERR_HIP;
            break;
        }
        unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 16384);
        hipLaunchKernelGGL(scatter_stamp_kernel, dim3(grid), dim3(256), 0, ctx->stream, (uint32_t*)d_stamp, d_off, n);
        st = post_launch(ctx, "scatter_stamp_kernel");
        if (st != RK_OK) break;
        hipLaunchKernelGGL(scatter_write_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_into, (const uint32_t*)d_stamp,
                           d_off, d_val, n);
        st = post_launch(ctx, "scatter_write_kernel");
    } while (0);
    // the host arrays are the caller's: drain the copies before returning
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess && st == RK_OK) {
        ctx->last_error = std::string("scatter sync: ") + hipGetErrorString(e);
        st = RK_ERR_HIP;
    }
    dev_free(ctx, d_stamp);
    dev_free(ctx, d_pack);
    return st;
}

}  // namespace rk

extern "C" {

int rk_prefix_products(rk_ctx* ctx, uint32_t* d_io_ext, size_t count) {
    RK_GUARD_BEGIN
    if (!ctx || (count && !d_io_ext)) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::prefix_products(ctx, d_io_ext, count);
    RK_GUARD_END
}
int rk_scatter(rk_ctx* ctx, uint32_t* d_into, size_t into_words, const uint32_t* h_index, size_t n_cycles,
               const uint32_t* h_offsets, const uint32_t* h_values) {
    RK_GUARD_BEGIN
    if (!ctx) return RK_ERR_INVALID;
    if (n_cycles == 0) return RK_OK;
    if (!d_into || !h_index || !h_offsets || !h_values) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::scatter(ctx, d_into, into_words, h_index, n_cycles, h_offsets, h_values);
    RK_GUARD_END
}

}  // extern "C"
