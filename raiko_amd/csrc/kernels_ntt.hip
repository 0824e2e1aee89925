// NTT / iNTT / LDE launches (Hal::batch_interpolate_ntt, batch_evaluate_ntt,
// batch_expand_into_evaluate_ntt, zk_shift, batch_bit_reverse of risc0-zkp 1.0.1).
// Each pass: one workgroup per 2^g x T tile, tile staged in LDS (<= 64 KiB so two
// workgroups share a CU's 160 KiB), coalesced tile rows of T consecutive elements.
#include "internal.hpp"
#include "ntt_r16.hpp"

namespace {

template <bool FWD>
__global__ __launch_bounds__(1024) void ntt_pass_kernel(ntt::PassArgs a, ntt::Tables tb) {
    extern __shared__ uint32_t lds[];
    const unsigned tid = threadIdx.x, nthr = blockDim.x;
    const ntt::Tile t = ntt::tile_of(a, blockIdx.x);
    if (FWD) {
        ntt::fwd_load(a, tb, t, lds, tid, nthr);
        __syncthreads();
        for (unsigned s = a.expand_bits; s < a.g; s++) {
            ntt::fwd_stage(a, tb, lds, tid, nthr, s);
            __syncthreads();
        }
        ntt::fwd_store(a, t, lds, tid, nthr);
    } else {
        ntt::rev_load(a, t, lds, tid, nthr);
        __syncthreads();
        for (unsigned s = 0; s < a.g; s++) {
            ntt::rev_stage(a, tb, lds, tid, nthr, s);
            __syncthreads();
        }
        ntt::rev_store(a, tb, t, lds, tid, nthr);
    }
}

// unrolled form: tile == EPT * blockDim.x, every phase issues its memory operations up front
template <bool FWD, int EPT, bool VEC>
__global__ __launch_bounds__(1024) void ntt_pass_kernel_u(ntt::PassArgs a, ntt::Tables tb) {
    extern __shared__ uint32_t lds[];
    const unsigned tid = threadIdx.x, nthr = blockDim.x;
    const ntt::Tile t = ntt::tile_of(a, blockIdx.x);
    if (FWD) {
        ntt::fwd_load_t<EPT, VEC>(a, tb, t, lds, tid, nthr);
        __syncthreads();
        for (unsigned s = a.expand_bits; s < a.g; s++) {
            ntt::fwd_stage_t<EPT / 2>(a, tb, lds, tid, nthr, s);
            __syncthreads();
        }
        ntt::fwd_store_t<EPT, VEC>(a, t, lds, tid, nthr);
    } else {
        ntt::rev_load_t<EPT, VEC>(a, t, lds, tid, nthr);
        __syncthreads();
        for (unsigned s = 0; s < a.g; s++) {
            ntt::rev_stage_t<EPT / 2>(a, tb, lds, tid, nthr, s);
            __syncthreads();
        }
        ntt::rev_store_t<EPT, VEC>(a, tb, t, lds, tid, nthr);
    }
}

// register-blocked passes on full 2^14 tiles (ntt_r16.hpp)
template <bool FWD, int NST>
__device__ __forceinline__ void r16_round(uint32_t* v, const ntt::Tables& tb, unsigned ls, unsigned rlow) {
    if (FWD) r16::round_dit<NST>(v, tb.small[0], ls, rlow);
    else r16::round_dif<NST>(v, tb.small[1], ls, rlow);
}
// 8 waves per SIMD (<= 64 VGPRs): two 1024-lane workgroups per CU, so one group's barriers and LDS
// round trips are covered by the other's arithmetic
template <bool FWD, bool CONTIG>
__global__ __launch_bounds__(1024, 8) void ntt_r16_kernel(r16::Args a, ntt::Tables tb, r16::Sched sc) {
    __shared__ uint32_t lds[r16::LDS_WORDS];
    const unsigned tid = threadIdx.x;
    const r16::Tile t = r16::tile_of(a, blockIdx.x);
    if (CONTIG) {
        if (FWD) r16::load_fwd_contig(a, t, lds, tid);
        else r16::load_rev_contig(a, tb, t, lds, tid);
    } else {
        r16::load_plain(a, t, lds, tid);
    }
    __syncthreads();
    for (unsigned rd = 0; rd < sc.n; rd++) {
        const unsigned ls = sc.ls[rd];
        const r16::RoundIdx x = r16::round_idx(tid, a.g, ls);
        uint32_t v[16];
        r16::round_read(v, lds, x);
        switch (sc.nst[rd]) {
            case 4: r16_round<FWD, 4>(v, tb, ls, x.rlow); break;
            case 3: r16_round<FWD, 3>(v, tb, ls, x.rlow); break;
            case 2: r16_round<FWD, 2>(v, tb, ls, x.rlow); break;
            default: r16_round<FWD, 1>(v, tb, ls, x.rlow); break;
        }
        r16::round_write(v, lds, x);
        __syncthreads();
    }
    if (CONTIG) {
        if (FWD) r16::store_fwd_contig(a, tb, t, lds, tid);
        else r16::store_rev_contig(a, tb, t, lds, tid);
    } else {
        r16::store_plain<FWD>(a, t, lds, tid);
    }
}

__global__ void zk_shift_kernel(uint32_t* io, size_t total, size_t size, unsigned bits, ntt::Tables tb) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; idx < total; idx += stride) {
        uint32_t pos = (uint32_t)(idx & (size - 1));
        io[idx] = bb::mul(io[idx], ntt::pow3(tb, bb::bitrev(pos, bits)));
    }
}

__global__ void bit_reverse_kernel(uint32_t* io, size_t total, size_t size, unsigned bits) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; idx < total; idx += stride) {
        uint32_t pos = (uint32_t)(idx & (size - 1));
        uint32_t rev = bb::bitrev(pos, bits);
        if (pos < rev) {
            size_t other = idx - pos + rev;
            uint32_t a = io[idx], b = io[other];
            io[idx] = b;
            io[other] = a;
        }
    }
}

template <bool FWD>
int launch_pass(rk_ctx* ctx, const ntt::PassArgs& a, size_t count) {
    constexpr int EPT = 16;
    size_t tile = (size_t)1 << (a.g + a.logT);
    size_t blocks = count * (a.n >> (a.g + a.logT));
    if (blocks == 0 || blocks > 0x7fffffffu) return RK_ERR_INVALID;
    size_t lds_bytes = tile * sizeof(uint32_t);
    rk::KTimer kt(ctx, RK_KCLASS_NTT_PASS, (double)count * 4 * ((a.expand_bits ? a.n_src : a.n) + a.n));
    if (ntt::can_unroll(a, EPT)) {
        unsigned threads = (unsigned)(tile / EPT);
        if (ntt::can_vec(a))
            hipLaunchKernelGGL((ntt_pass_kernel_u<FWD, EPT, true>), dim3((unsigned)blocks), dim3(threads), lds_bytes,
                               ctx->stream, a, ctx->tb);
        else
            hipLaunchKernelGGL((ntt_pass_kernel_u<FWD, EPT, false>), dim3((unsigned)blocks), dim3(threads), lds_bytes,
                               ctx->stream, a, ctx->tb);
    } else {
        hipLaunchKernelGGL(ntt_pass_kernel<FWD>, dim3((unsigned)blocks), dim3(64), lds_bytes, ctx->stream, a, ctx->tb);
    }
    return rk::post_launch(ctx, FWD ? "ntt_pass_kernel<fwd>" : "ntt_pass_kernel<rev>");
}

template <bool FWD, bool CONTIG>
int launch_r16(rk_ctx* ctx, const r16::Args& a, size_t count) {
    size_t blocks = count * (a.n >> r16::TILE_LOG);
    if (blocks == 0 || blocks > 0x7fffffffu) return RK_ERR_INVALID;
    r16::Sched sc = FWD ? r16::sched_dit(a.g, a.expand_bits) : r16::sched_dif(a.g);
    rk::KTimer kt(ctx, RK_KCLASS_NTT_PASS, (double)count * 4 * (a.n_src + a.n));
    hipLaunchKernelGGL((ntt_r16_kernel<FWD, CONTIG>), dim3((unsigned)blocks), dim3(r16::NTHR), 0, ctx->stream, a, ctx->tb,
                       sc);
    return rk::post_launch(ctx, "ntt_r16_kernel");
}
inline bool aligned16(const void* p, const void* q) { return ((((uintptr_t)p) | ((uintptr_t)q)) & 15) == 0; }

}  // namespace

namespace rk {

int ntt_reverse(rk_ctx* ctx, uint32_t* d_io, size_t size, size_t count, bool fuse_zk_shift) {
    if (!is_pow2(size) || size > ((size_t)1 << ntt::LAMBDA) || count == 0) return RK_ERR_INVALID;
    unsigned k = log2u(size);
    uint32_t scale = bb::inv(bb::encode((uint32_t)size));
    if (k == 0) {
        // size-1 transform: identity (1/1 scale, 3^0 shift)
        return RK_OK;
    }
    if (r16::usable(k, 0, aligned16(d_io, d_io))) {
        r16::Args a{};
        a.dst = d_io;
        a.src = d_io;
        a.n = a.n_src = size;
        a.k = k;
        a.g = k - r16::TILE_LOG;
        RK_TRY((launch_r16<false, false>(ctx, a, count)));
        a.g_outer = a.g;
        a.g = r16::TILE_LOG;
        a.scale = scale;
        a.zk = fuse_zk_shift ? 1 : 0;
        return launch_r16<false, true>(ctx, a, count);
    }
    ntt::Plan plan = ntt::make_plan(k);
    for (unsigned p = 0; p < plan.npass; p++) {
        ntt::PassArgs a{};
        a.dst = d_io;
        a.src = d_io;
        a.n = size;
        a.n_src = size;
        a.mu = plan.mu[p];
        a.g = plan.g[p];
        a.logT = plan.logT[p];
        a.expand_bits = 0;
        bool last = p + 1 == plan.npass;
        a.scale = last ? scale : 0;
        a.zk_bits = (last && fuse_zk_shift) ? k : 0;
        RK_TRY(launch_pass<false>(ctx, a, count));
    }
    return RK_OK;
}

int ntt_forward(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t in_size, size_t count,
                unsigned expand_bits) {
    if (!is_pow2(in_size) || count == 0 || expand_bits > 4) return RK_ERR_INVALID;
    size_t size = in_size << expand_bits;
    if (size > ((size_t)1 << ntt::LAMBDA)) return RK_ERR_INVALID;
    unsigned k = log2u(size);
    if (k == 0) {
        if (d_out != d_in)
            RK_HIP_TRY(ctx, hipMemcpyAsync(d_out, d_in, count * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
        return RK_OK;
    }
    if (r16::usable(k, expand_bits, aligned16(d_out, d_in))) {
        r16::Args a{};
        a.dst = d_out;
        a.src = d_in;
        a.n = size;
        a.n_src = in_size;
        a.k = k;
        a.g = r16::TILE_LOG;
        a.g_outer = k - r16::TILE_LOG;
        a.expand_bits = expand_bits;
        RK_TRY((launch_r16<true, true>(ctx, a, count)));
        a.src = d_out;
        a.n_src = size;
        a.g = k - r16::TILE_LOG;
        a.g_outer = 0;
        a.expand_bits = 0;
        return launch_r16<true, false>(ctx, a, count);
    }
    if (expand_bits == 0 && d_out != d_in) {
        RK_HIP_TRY(ctx, hipMemcpyAsync(d_out, d_in, count * size * sizeof(uint32_t), hipMemcpyDeviceToDevice,
                                       ctx->stream));
        d_in = d_out;
    }
    ntt::Plan plan = ntt::make_plan(k);
    for (unsigned pi = plan.npass; pi-- > 0;) {
        ntt::PassArgs a{};
        bool first = pi + 1 == plan.npass;  // innermost (contiguous) pass runs first
        a.dst = d_out;
        a.src = first ? d_in : d_out;
        a.n = size;
        a.n_src = first ? in_size : size;
        a.mu = plan.mu[pi];
        a.g = plan.g[pi];
        a.logT = plan.logT[pi];
        a.expand_bits = first ? expand_bits : 0;
        a.scale = 0;
        a.zk_bits = 0;
        if (first && expand_bits > a.g) return RK_ERR_INVALID;
        RK_TRY(launch_pass<true>(ctx, a, count));
    }
    return RK_OK;
}

int zk_shift(rk_ctx* ctx, uint32_t* d_io, size_t size, size_t count) {
    if (!is_pow2(size) || size > ((size_t)1 << ntt::LAMBDA) || count == 0) return RK_ERR_INVALID;
    size_t total = size * count;
    unsigned blocks = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    KTimer kt(ctx, RK_KCLASS_POLY, (double)total * 8);
    hipLaunchKernelGGL(zk_shift_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_io, total, size, log2u(size), ctx->tb);
    return post_launch(ctx, "zk_shift_kernel");
}

int bit_reverse(rk_ctx* ctx, uint32_t* d_io, size_t size, size_t count) {
    if (!is_pow2(size) || size > ((size_t)1 << 31) || count == 0) return RK_ERR_INVALID;
    if (size <= 2) return RK_OK;
    size_t total = size * count;
    unsigned blocks = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    KTimer kt(ctx, RK_KCLASS_BIT_REVERSE, (double)total * 8);
    hipLaunchKernelGGL(bit_reverse_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_io, total, size, log2u(size));
    return post_launch(ctx, "bit_reverse_kernel");
}

}  // namespace rk

extern "C" {

int rk_batch_interpolate_ntt(rk_ctx* ctx, uint32_t* d_io, size_t size, size_t count) {
    if (!ctx || !d_io) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::ntt_reverse(ctx, d_io, size, count, false);
}
int rk_batch_evaluate_ntt(rk_ctx* ctx, uint32_t* d_io, size_t size, size_t count, uint32_t expand_bits) {
    if (!ctx || !d_io || !is_pow2(size) || (size >> expand_bits) == 0) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (expand_bits != 0) {
        // in-place evaluate of an already-expanded (broadcast) buffer: the low stages are a no-op
        // only when the input really is a broadcast; risc0 calls this form with expand_bits = 0
        return RK_ERR_INVALID;
    }
    return rk::ntt_forward(ctx, d_io, d_io, size, count, 0);
}
int rk_zk_shift(rk_ctx* ctx, uint32_t* d_io, size_t size, size_t count) {
    if (!ctx || !d_io) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::zk_shift(ctx, d_io, size, count);
}
int rk_batch_expand_into_evaluate_ntt(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t in_size,
                                      size_t count, uint32_t expand_bits) {
    if (!ctx || !d_out || !d_in || d_out == d_in) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::ntt_forward(ctx, d_out, d_in, in_size, count, expand_bits);
}
int rk_batch_bit_reverse(rk_ctx* ctx, uint32_t* d_io, size_t size, size_t count) {
    if (!ctx || !d_io) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::bit_reverse(ctx, d_io, size, count);
}

}  // extern "C"
