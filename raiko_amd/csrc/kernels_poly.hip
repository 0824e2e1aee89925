// Polynomial / element-wise operators of the Hal trait (risc0-zkp 1.0.1 hal/mod.rs):
// eltwise_*, fri_fold, gather_sample, batch_evaluate_any, mix_poly_coeffs, plus the
// device form of core/poly.rs poly_divide.  All are HBM-bound streaming kernels:
// one lane per coefficient index, consecutive lanes on consecutive addresses.
#include "internal.hpp"

#include <algorithm>
#include <cstring>

namespace {

using bb::Ext;
constexpr int TPB = 256;

inline unsigned grid_for(size_t n, unsigned cap = 16384) {
    size_t b = (n + TPB - 1) / TPB;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

__device__ __forceinline__ Ext load_ext(const uint32_t* p) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    return Ext{{v.x, v.y, v.z, v.w}};
}
__device__ __forceinline__ void store_ext(uint32_t* p, const Ext& e) {
    *reinterpret_cast<uint4*>(p) = make_uint4(e.c[0], e.c[1], e.c[2], e.c[3]);
}

__global__ void add_kernel(uint32_t* out, const uint32_t* a, const uint32_t* b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) out[i] = bb::add(a[i], b[i]);
}
__global__ void zeroize_kernel(uint32_t* io, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st)
        if (io[i] == bb::INVALID) io[i] = 0;
}
__global__ void sum_ext_kernel(uint32_t* out, const uint32_t* in, size_t count, size_t to_add) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < count; i += st) {
        Ext tot = bb::ext_zero();
        for (size_t j = 0; j < to_add; j++) tot = bb::add(tot, load_ext(in + (j * count + i) * 4));
#pragma unroll
        for (int k = 0; k < 4; k++) out[(size_t)k * count + i] = tot.c[k];
    }
}

struct FoldPows {
    Ext p[16];
};
// fold by A = 2^LOG_A: out[idx] = sum_{i < A} mix^i * in[bitrev(i) * count + idx] over 4 planes of A * count
template <int LOG_A>
__global__ void fri_fold_kernel(uint32_t* out, const uint32_t* in, size_t count, FoldPows pw, uint32_t wm) {
    constexpr unsigned A = 1u << LOG_A;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; idx < count; idx += st) {
        Ext tot = bb::ext_zero();
#pragma unroll
        for (unsigned i = 0; i < A; i++) {
            unsigned rev_i = bb::bitrev(i, LOG_A);
            size_t ri = (size_t)rev_i * count + idx;
            Ext f{{in[ri], in[count * A + ri], in[count * 2 * A + ri], in[count * 3 * A + ri]}};
            tot = bb::add(tot, bb::mul(pw.p[i], f, wm));
        }
#pragma unroll
        for (int k = 0; k < 4; k++) out[(size_t)k * count + idx] = tot.c[k];
    }
}

__global__ void fri_fold_evals_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ in, size_t n_out, unsigned k, Ext beta,
                                      uint32_t half, uint32_t wm, ntt::Tables tb) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const Ext a = load_ext(in + 8 * i), b = load_ext(in + 8 * i + 4);       // p(x), p(-x)
    const uint32_t j = bb::bitrev((uint32_t)i, k - 1);                      // x = g^j, g of order 2^k
    const uint32_t xinv_half = bb::mul(half, ntt::root_pow(tb, 1, j << (ntt::LAMBDA - k)));
    const Ext even = bb::scale(bb::add(a, b), half);
    const Ext odd = bb::scale(bb::sub(a, b), xinv_half);
    store_ext(out + 4 * i, bb::add(even, bb::mul(beta, odd, wm)));
}

__global__ void gather_sample_kernel(uint32_t* dst, const uint32_t* src, size_t idx, size_t size, size_t stride) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; g < size; g += st) dst[g] = src[g * stride + idx];
}
__global__ void gather_rows_kernel(uint32_t* dst, const uint32_t* matrix, size_t rows, size_t cols,
                                   const uint32_t* idx) {
    size_t q = blockIdx.y;
    size_t r = idx[q];
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < cols; c += (size_t)gridDim.x * blockDim.x)
        dst[q * cols + c] = matrix[c * rows + r];
}
__global__ void gather_digests_kernel(uint32_t* dst, const uint32_t* nodes, const uint32_t* idx, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 8) return;
    dst[i] = nodes[(size_t)idx[i >> 3] * 8 + (i & 7)];
}

// the two gathers for every tree of a proof in one launch each: blockIdx.z = job
__global__ void gather_rows_many_kernel(uint32_t* dst, const uint32_t* idx, const rk::GatherJob* jobs) {
    const rk::GatherJob j = jobs[blockIdx.z];
    const size_t q = blockIdx.y;
    if (q >= j.n) return;
    const size_t r = idx[j.idx_off + q];
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < j.cols; c += (size_t)gridDim.x * blockDim.x)
        dst[j.dst_off + q * j.cols + c] = j.matrix[c * j.rows + r];
}
__global__ void gather_digests_many_kernel(uint32_t* dst, const uint32_t* idx, const rk::GatherJob* jobs) {
    const rk::GatherJob j = jobs[blockIdx.z];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= j.n * j.path_len * 8) return;
    dst[j.dst_off + j.n * j.cols + i] = j.nodes[(size_t)idx[j.idx_off + j.n + (i >> 3)] * 8 + (i & 7)];
}

// pw[k] = x^k: each lane seeds x^(lane_start) by square-and-multiply, then walks CH powers
constexpr int PW_CH = 32;
// rev_bits != 0: x^k is stored at position bitrev(k), matching bit-reversed coefficient storage
__global__ void ext_powers_kernel(uint32_t* pw, Ext x, size_t n, unsigned rev_bits, uint32_t wm) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t start = t * PW_CH;
    if (start >= n) return;
    Ext cur = bb::pow(x, (uint64_t)start, wm);
    size_t end = start + PW_CH < n ? start + PW_CH : n;
    for (size_t k = start; k < end; k++) {
        size_t pos = rev_bits ? (size_t)bb::bitrev((uint32_t)k, rev_bits) : k;
        store_ext(pw + pos * 4, cur);
        cur = bb::mul(cur, x, wm);
    }
}
// the same for blockIdx.y = table index, base points read from device memory
__global__ void ext_powers_many_kernel(uint32_t* pw, const Ext* xs, size_t n, unsigned rev_bits, uint32_t wm) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t start = t * PW_CH;
    if (start >= n) return;
    const Ext x = xs[blockIdx.y];
    pw += (size_t)blockIdx.y * n * 4;
    Ext cur = bb::pow(x, (uint64_t)start, wm);
    size_t end = start + PW_CH < n ? start + PW_CH : n;
    for (size_t k = start; k < end; k++) {
        size_t pos = rev_bits ? (size_t)bb::bitrev((uint32_t)k, rev_bits) : k;
        store_ext(pw + pos * 4, cur);
        cur = bb::mul(cur, x, wm);
    }
}
// in-place bit reversal of `count` polynomials of `size` extension elements (16-byte items)
__global__ void bit_reverse_ext_kernel(uint32_t* io, size_t total, size_t size, unsigned bits) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; idx < total; idx += st) {
        uint32_t pos = (uint32_t)(idx & (size - 1));
        uint32_t rev = bb::bitrev(pos, bits);
        if (pos < rev) {
            size_t other = idx - pos + rev;
            Ext a = load_ext(io + idx * 4), b = load_ext(io + other * 4);
            store_ext(io + idx * 4, b);
            store_ext(io + other * 4, a);
        }
    }
}

// partial[e][blk] = sum over this block's slice of coeffs[which[e]][k] * pw[sel[e]][k]
constexpr int DOT_BLOCKS = 64;
__global__ __launch_bounds__(TPB) void eval_dot_kernel(uint32_t* __restrict__ partial, const uint32_t* __restrict__ coeffs,
                                                       size_t size, const uint32_t* __restrict__ which,
                                                       const uint32_t* __restrict__ pw, const uint32_t* __restrict__ sel) {
    __shared__ uint32_t red[TPB * 4];
    size_t e = blockIdx.y;
    const uint32_t* c = coeffs + (size_t)which[e] * size;
    const uint32_t* p = pw + (size_t)sel[e] * size * 4;
    Ext acc = bb::ext_zero();
    // four independent (coefficient, table element) loads in flight per lane
    constexpr size_t STEP = (size_t)DOT_BLOCKS * TPB;
    size_t k = (size_t)blockIdx.x * TPB + threadIdx.x;
    for (; k + 3 * STEP < size; k += 4 * STEP) {
        uint32_t v[4];
        Ext t[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            v[u] = c[k + u * STEP];
            t[u] = load_ext(p + (k + u * STEP) * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc = bb::add(acc, bb::scale(t[u], v[u]));
    }
    for (; k < size; k += STEP) acc = bb::add(acc, bb::scale(load_ext(p + k * 4), c[k]));
#pragma unroll
    for (int j = 0; j < 4; j++) red[threadIdx.x * 4 + j] = acc.c[j];
    __syncthreads();
    for (int s = TPB / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                red[threadIdx.x * 4 + j] = bb::add(red[threadIdx.x * 4 + j], red[(threadIdx.x + s) * 4 + j]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) partial[(e * DOT_BLOCKS + blockIdx.x) * 4 + threadIdx.x] = red[threadIdx.x];
}
__global__ void eval_reduce_kernel(uint32_t* out, const uint32_t* partial, size_t n_eval) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_eval * 4) return;
    size_t e = i >> 2, j = i & 3;
    uint32_t acc = 0;
    for (int b = 0; b < DOT_BLOCKS; b++) acc = bb::add(acc, partial[(e * DOT_BLOCKS + b) * 4 + j]);
    out[i] = acc;
}

// one grid.y slot per distinct combo: out[combo][idx] += sum_t pows[t] * in[cols[t]][idx].
// A slot can hold a couple of hundred columns (every register read at the current row only); the
// loads of eight of them are issued together so that a lane keeps eight coefficient loads in
// flight instead of one.
__global__ void mix_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ in, size_t count,
                           const uint32_t* __restrict__ slot_combo, const uint32_t* __restrict__ slot_off,
                           const uint32_t* __restrict__ cols, const uint32_t* __restrict__ pows) {
    constexpr unsigned U = 8;
    unsigned slot = blockIdx.y;
    size_t combo = slot_combo[slot];
    unsigned t0 = slot_off[slot], t1 = slot_off[slot + 1];
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; idx < count; idx += st) {
        uint32_t* o = out + (combo * count + idx) * 4;
        Ext acc = load_ext(o);
        unsigned t = t0;
        for (; t + U <= t1; t += U) {
            uint32_t v[U];
#pragma unroll
            for (unsigned u = 0; u < U; u++) v[u] = in[(size_t)cols[t + u] * count + idx];
#pragma unroll
            for (unsigned u = 0; u < U; u++) acc = bb::add(acc, bb::scale(load_ext(pows + (size_t)(t + u) * 4), v[u]));
        }
        for (; t < t1; t++) {
            Ext pw = load_ext(pows + (size_t)t * 4);
            acc = bb::add(acc, bb::scale(pw, in[(size_t)cols[t] * count + idx]));
        }
        store_ext(o, acc);
    }
}

// ---- synthetic division by (x - z): q[i-1] = c[i] + z*q[i], chunked Horner -----------------
// Several polynomials (each with its own z) per launch: blockIdx.y = item.  Three launches per
// batch: per-chunk Horner tops, one workgroup per item that turns the tops into the carry entering
// every chunk (three-level scan: 16 + 32 + 32 + 32 + 16 dependent steps for 2^20 coefficients),
// per-chunk apply.
constexpr int DIV_CH = 64;
constexpr int DIV_NT = 1024;   // lanes of the carry workgroup
constexpr int DIV_GRP = 32;    // segments per leader lane
struct DivItem {
    Ext z, zL;               // divisor point, z^DIV_CH
    unsigned long long off;  // first ext element of the polynomial
    uint32_t wm;             // the extension's W (Montgomery form)
    uint32_t pad_;
};
// tops[item][b] = sum_{k in chunk b} c[k] z^(k - start_b)
__global__ void div_tops_kernel(uint32_t* tops, const uint32_t* base, size_t count, size_t nchunks, const DivItem* items) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nchunks) return;
    const DivItem it = items[blockIdx.y];
    const uint32_t* poly = base + it.off * 4;
    size_t start = b * DIV_CH;
    size_t end = start + DIV_CH < count ? start + DIV_CH : count;
    Ext s = bb::ext_zero();
    for (size_t k = end; k-- > start;) s = bb::add(bb::mul(s, it.z, it.wm), load_ext(poly + k * 4));
    store_ext(tops + ((size_t)blockIdx.y * nchunks + b) * 4, s);
}
// carry[b] = T[b+1] with T[b] = tops[b] + zL * T[b+1], T[nchunks] = 0; rem = T[0] = f(z)
__global__ __launch_bounds__(DIV_NT) void div_carry_kernel(uint32_t* carry, const uint32_t* tops, size_t nchunks,
                                                           const DivItem* items, uint32_t* rem_out) {
    __shared__ uint32_t seg[DIV_NT * 4];   // segment tops, later the carry entering each segment
    __shared__ uint32_t sup[DIV_GRP * 4];  // same for groups of DIV_GRP segments
    const unsigned t = threadIdx.x;
    const DivItem it = items[blockIdx.y];
    tops += (size_t)blockIdx.y * nchunks * 4;
    carry += (size_t)blockIdx.y * nchunks * 4;
    const size_t per = (nchunks + DIV_NT - 1) / DIV_NT;
    const size_t s0 = (size_t)t * per, s1 = s0 + per < nchunks ? s0 + per : nchunks;
    // A: value at the segment's first chunk with zero carry-in (missing chunks count as zero)
    Ext acc = bb::ext_zero();
    if (s0 < nchunks)
        for (size_t b = s1; b-- > s0;) acc = bb::add(bb::mul(acc, it.zL, it.wm), load_ext(tops + b * 4));
#pragma unroll
    for (int j = 0; j < 4; j++) seg[t * 4 + j] = acc.c[j];
    __syncthreads();
    const Ext zseg = bb::pow(it.zL, (uint64_t)per, it.wm);
    // B: leaders fold their DIV_GRP segments
    if (t < DIV_GRP) {
        Ext g = bb::ext_zero();
        for (int i = DIV_GRP - 1; i >= 0; i--) {
            unsigned sidx = t * DIV_GRP + i;
            Ext top{{seg[sidx * 4], seg[sidx * 4 + 1], seg[sidx * 4 + 2], seg[sidx * 4 + 3]}};
            g = bb::add(top, bb::mul(zseg, g, it.wm));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) sup[t * 4 + j] = g.c[j];
    }
    __syncthreads();
    // C: one lane walks the groups from the top; sup[] becomes the carry entering each group
    if (t == 0) {
        const Ext zsup = bb::pow(zseg, (uint64_t)DIV_GRP, it.wm);
        Ext run = bb::ext_zero();
        for (int u = DIV_GRP - 1; u >= 0; u--) {
            Ext top{{sup[u * 4], sup[u * 4 + 1], sup[u * 4 + 2], sup[u * 4 + 3]}};
#pragma unroll
            for (int j = 0; j < 4; j++) sup[u * 4 + j] = run.c[j];
            run = bb::add(top, bb::mul(zsup, run, it.wm));
        }
        store_ext(rem_out + (size_t)blockIdx.y * 4, run);
    }
    __syncthreads();
    // B': leaders turn seg[] into the carry entering each segment
    if (t < DIV_GRP) {
        Ext run{{sup[t * 4], sup[t * 4 + 1], sup[t * 4 + 2], sup[t * 4 + 3]}};
        for (int i = DIV_GRP - 1; i >= 0; i--) {
            unsigned sidx = t * DIV_GRP + i;
            Ext top{{seg[sidx * 4], seg[sidx * 4 + 1], seg[sidx * 4 + 2], seg[sidx * 4 + 3]}};
#pragma unroll
            for (int j = 0; j < 4; j++) seg[sidx * 4 + j] = run.c[j];
            run = bb::add(top, bb::mul(zseg, run, it.wm));
        }
    }
    __syncthreads();
    // A': carries of the segment's chunks
    if (s0 < nchunks) {
        Ext run{{seg[t * 4], seg[t * 4 + 1], seg[t * 4 + 2], seg[t * 4 + 3]}};
        for (size_t b = s1; b-- > s0;) {
            store_ext(carry + b * 4, run);
            run = bb::add(load_ext(tops + b * 4), bb::mul(it.zL, run, it.wm));
        }
    }
}
// within chunk b: cur = carry[b]; for k from top: next = z*cur + c[k]; c[k] = cur; cur = next
__global__ void div_apply_kernel(uint32_t* base, const uint32_t* carry, size_t count, size_t nchunks, const DivItem* items) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nchunks) return;
    const DivItem it = items[blockIdx.y];
    uint32_t* poly = base + it.off * 4;
    size_t start = b * DIV_CH;
    size_t end = start + DIV_CH < count ? start + DIV_CH : count;
    Ext cur = load_ext(carry + ((size_t)blockIdx.y * nchunks + b) * 4);
    for (size_t k = end; k-- > start;) {
        Ext next = bb::add(bb::mul(it.z, cur, it.wm), load_ext(poly + k * 4));
        store_ext(poly + k * 4, cur);
        cur = next;
    }
}

__global__ void ext_sub_at_kernel(uint32_t* data, const uint32_t* idx, const uint32_t* delta, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t* p = data + (size_t)idx[i] * 4;
    store_ext(p, bb::sub(load_ext(p), load_ext(delta + i * 4)));
}

}  // namespace

namespace rk {

int eltwise_add(rk_ctx* ctx, uint32_t* d_out, const uint32_t* a, const uint32_t* b, size_t n) {
    if (n == 0) return RK_OK;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n)), dim3(TPB), 0, ctx->stream, d_out, a, b, n);
    return post_launch(ctx, "add_kernel");
}
int eltwise_zeroize(rk_ctx* ctx, uint32_t* d_io, size_t n) {
    if (n == 0) return RK_OK;
    hipLaunchKernelGGL(zeroize_kernel, dim3(grid_for(n)), dim3(TPB), 0, ctx->stream, d_io, n);
    return post_launch(ctx, "zeroize_kernel");
}
int eltwise_sum_ext(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t count, size_t to_add) {
    if (count == 0) return RK_OK;
    KTimer kt(ctx, RK_KCLASS_POLY, (double)count * (to_add + 1) * 16);
    hipLaunchKernelGGL(sum_ext_kernel, dim3(grid_for(count)), dim3(TPB), 0, ctx->stream, d_out, d_in, count, to_add);
    return post_launch(ctx, "sum_ext_kernel");
}
int fri_fold(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t out_count, const bb::Ext& mix) {
    if (out_count == 0) return RK_ERR_INVALID;
    const uint32_t wm = ctx->sys.wm;
    const unsigned log_a = ctx->sys.fri_fold_log2, a = 1u << log_a;
    FoldPows pw;
    pw.p[0] = bb::ext_one();
    for (unsigned i = 1; i < 16; i++) pw.p[i] = i < a ? bb::mul(pw.p[i - 1], mix, wm) : bb::ext_zero();
    KTimer kt(ctx, RK_KCLASS_POLY, (double)out_count * (a + 1) * 16);
    const dim3 grid(grid_for(out_count)), blk(TPB);
    switch (log_a) {
        case 1: hipLaunchKernelGGL(fri_fold_kernel<1>, grid, blk, 0, ctx->stream, d_out, d_in, out_count, pw, wm); break;
        case 2: hipLaunchKernelGGL(fri_fold_kernel<2>, grid, blk, 0, ctx->stream, d_out, d_in, out_count, pw, wm); break;
        case 3: hipLaunchKernelGGL(fri_fold_kernel<3>, grid, blk, 0, ctx->stream, d_out, d_in, out_count, pw, wm); break;
        default: hipLaunchKernelGGL(fri_fold_kernel<4>, grid, blk, 0, ctx->stream, d_out, d_in, out_count, pw, wm); break;
    }
    return post_launch(ctx, "fri_fold_kernel");
}
// Plonky3's FRI fold (p3-fri fold_even_odd, RECALLED): on evaluations, arity 2.  `in` = 2 * n_out extension
// elements (4 consecutive words each), the evaluations of p over the subgroup of order 2 * n_out in
// bit-reversed order, so p(x) and p(-x) are neighbours; out[i] = (p(x) + p(-x)) / 2 + beta (p(x) - p(-x)) / (2 x)
// with x = g^bitrev(i): the evaluations of p_even + beta p_odd over the squared subgroup, bit-reversed.
int fri_fold_evals(rk_ctx* ctx, uint32_t* d_out_ext, const uint32_t* d_in_ext, size_t n_out, const bb::Ext& beta) {
    if (!is_pow2(n_out) || 2 * n_out > ((size_t)1 << ntt::LAMBDA)) return RK_ERR_INVALID;
    const unsigned k = log2u(2 * n_out);
    const uint32_t half = bb::inv(bb::encode(2));
    KTimer kt(ctx, RK_KCLASS_POLY, (double)n_out * 48);
    hipLaunchKernelGGL(fri_fold_evals_kernel, dim3(grid_for(n_out)), dim3(TPB), 0, ctx->stream, d_out_ext, d_in_ext, n_out, k, beta,
                       half, ctx->sys.wm, ctx->tb);
    return post_launch(ctx, "fri_fold_evals_kernel");
}
int gather_sample(rk_ctx* ctx, uint32_t* d_dst, const uint32_t* d_src, size_t idx, size_t size, size_t stride) {
    if (size == 0) return RK_OK;
    hipLaunchKernelGGL(gather_sample_kernel, dim3(grid_for(size)), dim3(TPB), 0, ctx->stream, d_dst, d_src, idx, size,
                       stride);
    return post_launch(ctx, "gather_sample_kernel");
}
int gather_rows(rk_ctx* ctx, uint32_t* d_dst, const uint32_t* d_matrix, size_t rows, size_t cols,
                const uint32_t* d_idx, size_t n_idx) {
    if (n_idx == 0 || cols == 0) return RK_OK;
    if (n_idx > 65535) return RK_ERR_INVALID;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(cols, 64), (unsigned)n_idx), dim3(TPB), 0, ctx->stream, d_dst,
                       d_matrix, rows, cols, d_idx);
    return post_launch(ctx, "gather_rows_kernel");
}
int gather_digests(rk_ctx* ctx, uint32_t* d_dst, const uint32_t* d_nodes, const uint32_t* d_idx, size_t n_idx) {
    if (n_idx == 0) return RK_OK;
    size_t n = n_idx * 8;
    hipLaunchKernelGGL(gather_digests_kernel, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, ctx->stream, d_dst,
                       d_nodes, d_idx, n_idx);
    return post_launch(ctx, "gather_digests_kernel");
}
int gather_many(rk_ctx* ctx, uint32_t* d_dst, const uint32_t* d_idx, const GatherJob* d_jobs, const GatherJob* h_jobs, size_t n_jobs) {
    if (n_jobs == 0) return RK_OK;
    size_t max_n = 0, max_cols = 0, max_dig = 0;
    for (size_t k = 0; k < n_jobs; k++) {
        max_n = std::max<size_t>(max_n, h_jobs[k].n);
        max_cols = std::max<size_t>(max_cols, h_jobs[k].cols);
        max_dig = std::max<size_t>(max_dig, (size_t)h_jobs[k].n * h_jobs[k].path_len * 8);
    }
    if (max_n > 65535 || n_jobs > 65535) return RK_ERR_INVALID;
    if (max_n && max_cols) {
        hipLaunchKernelGGL(gather_rows_many_kernel, dim3(grid_for(max_cols, 64), (unsigned)max_n, (unsigned)n_jobs), dim3(TPB), 0,
                           ctx->stream, d_dst, d_idx, d_jobs);
        RK_TRY(post_launch(ctx, "gather_rows_many_kernel"));
    }
    if (max_dig) {
        hipLaunchKernelGGL(gather_digests_many_kernel, dim3((unsigned)((max_dig + TPB - 1) / TPB), 1, (unsigned)n_jobs), dim3(TPB), 0,
                           ctx->stream, d_dst, d_idx, d_jobs);
        RK_TRY(post_launch(ctx, "gather_digests_many_kernel"));
    }
    return RK_OK;
}
int ext_powers(rk_ctx* ctx, uint32_t* d_pw_ext, const bb::Ext& x, size_t n, bool bit_reversed) {
    if (n == 0) return RK_OK;
    if (bit_reversed && !is_pow2(n)) return RK_ERR_INVALID;
    size_t lanes = (n + PW_CH - 1) / PW_CH;
    KTimer kt(ctx, RK_KCLASS_POLY, (double)n * 16);
    hipLaunchKernelGGL(ext_powers_kernel, dim3((unsigned)((lanes + TPB - 1) / TPB)), dim3(TPB), 0, ctx->stream, d_pw_ext,
                       x, n, bit_reversed ? log2u(n) : 0u, ctx->sys.wm);
    return post_launch(ctx, "ext_powers_kernel");
}
int ext_powers_many(rk_ctx* ctx, uint32_t* d_pw_ext, const bb::Ext* h_pts, size_t n_pts, size_t n, bool bit_reversed) {
    if (n == 0 || n_pts == 0) return RK_OK;
    if ((bit_reversed && !is_pow2(n)) || n_pts > 65535) return RK_ERR_INVALID;
    void* d = nullptr;
    RK_TRY(scratch(ctx, n_pts * 16, &d));
    RK_TRY(upload(ctx, d, h_pts, n_pts * 16));
    size_t lanes = (n + PW_CH - 1) / PW_CH;
    KTimer kt(ctx, RK_KCLASS_POLY, (double)n * 16 * n_pts);
    hipLaunchKernelGGL(ext_powers_many_kernel, dim3((unsigned)((lanes + TPB - 1) / TPB), (unsigned)n_pts), dim3(TPB), 0, ctx->stream,
                       d_pw_ext, (const Ext*)d, n, bit_reversed ? log2u(n) : 0u, ctx->sys.wm);
    return post_launch(ctx, "ext_powers_many_kernel");
}
int bit_reverse_ext(rk_ctx* ctx, uint32_t* d_io_ext, size_t size, size_t count) {
    if (!is_pow2(size) || count == 0) return RK_ERR_INVALID;
    if (size <= 2) return RK_OK;
    size_t total = size * count;
    KTimer kt(ctx, RK_KCLASS_BIT_REVERSE, (double)total * 32);
    hipLaunchKernelGGL(bit_reverse_ext_kernel, dim3(grid_for(total)), dim3(TPB), 0, ctx->stream, d_io_ext, total, size,
                       log2u(size));
    return post_launch(ctx, "bit_reverse_ext_kernel");
}
int eval_dot(rk_ctx* ctx, uint32_t* d_out_ext, const uint32_t* d_coeffs, size_t size, const uint32_t* d_which,
             const uint32_t* d_pw_ext, const uint32_t* d_pw_sel, size_t eval_count) {
    if (eval_count == 0) return RK_OK;
    if (eval_count > 65535) return RK_ERR_INVALID;
    void* partial = nullptr;
    RK_TRY(dev_alloc(ctx, eval_count * DOT_BLOCKS * 16, &partial));
    KTimer kt(ctx, RK_KCLASS_POLY, (double)eval_count * size * 20);
    hipLaunchKernelGGL(eval_dot_kernel, dim3(DOT_BLOCKS, (unsigned)eval_count), dim3(TPB), 0, ctx->stream,
                       (uint32_t*)partial, d_coeffs, size, d_which, d_pw_ext, d_pw_sel);
    int st = post_launch(ctx, "eval_dot_kernel");
    if (st == RK_OK) {
        hipLaunchKernelGGL(eval_reduce_kernel, dim3((unsigned)((eval_count * 4 + TPB - 1) / TPB)), dim3(TPB), 0,
                           ctx->stream, d_out_ext, (const uint32_t*)partial, eval_count);
        st = post_launch(ctx, "eval_reduce_kernel");
    }
    // stream-ordered reuse: the block returns to this ctx's pool and any later user runs on the same stream
    dev_free(ctx, partial);
    return st;
}

int mix_poly_coeffs(rk_ctx* ctx, uint32_t* d_out_ext, const bb::Ext& mix_start, const bb::Ext& mix,
                    const uint32_t* d_in, const uint32_t* h_combos, size_t input_size, size_t count) {
    if (input_size == 0 || count == 0) return RK_OK;
    // bucket the input columns by destination combo; each bucket is one grid.y slot
    std::vector<uint32_t> order(input_size);
    for (size_t i = 0; i < input_size; i++) order[i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return h_combos[a] < h_combos[b]; });
    std::vector<bb::Ext> pw(input_size);
    bb::Ext cur = mix_start;
    for (size_t i = 0; i < input_size; i++) {
        pw[i] = cur;
        cur = bb::mul(cur, mix, ctx->sys.wm);
    }
    std::vector<uint32_t> slot_combo, slot_off, cols(input_size), pows(input_size * 4);
    for (size_t t = 0; t < input_size; t++) {
        uint32_t i = order[t];
        if (t == 0 || h_combos[i] != h_combos[order[t - 1]]) {
            slot_combo.push_back(h_combos[i]);
            slot_off.push_back((uint32_t)t);
        }
        cols[t] = i;
        std::memcpy(&pows[t * 4], pw[i].c, 16);
    }
    slot_off.push_back((uint32_t)input_size);
    size_t n_slots = slot_combo.size();
    // pack: [slot_combo | slot_off | cols | pows(16B aligned)]
    size_t o_combo = 0, o_off = o_combo + n_slots, o_cols = o_off + n_slots + 1;
    size_t o_pows = (o_cols + input_size + 3) & ~(size_t)3;
    size_t words = o_pows + input_size * 4;
    std::vector<uint32_t> pack(words, 0);
    std::memcpy(&pack[o_combo], slot_combo.data(), n_slots * 4);
    std::memcpy(&pack[o_off], slot_off.data(), (n_slots + 1) * 4);
    std::memcpy(&pack[o_cols], cols.data(), input_size * 4);
    std::memcpy(&pack[o_pows], pows.data(), input_size * 16);
    void* d = nullptr;
    RK_TRY(scratch(ctx, words * 4, &d));
    // the previous user of the scratch area may still be running: the copy is ordered behind it on the stream
    RK_TRY(upload(ctx, d, pack.data(), words * 4));
    const uint32_t* dp = (const uint32_t*)d;
    KTimer kt(ctx, RK_KCLASS_POLY, (double)count * (input_size * 4 + n_slots * 32));
    hipLaunchKernelGGL(mix_kernel, dim3(grid_for(count, 4096), (unsigned)n_slots), dim3(TPB), 0, ctx->stream, d_out_ext,
                       d_in, count, dp + o_combo, dp + o_off, dp + o_cols, dp + o_pows);
    return post_launch(ctx, "mix_kernel");
}

int poly_divide_many(rk_ctx* ctx, uint32_t* d_base_ext, size_t count, const size_t* h_offsets, const bb::Ext* h_z,
                     size_t n_items, bb::Ext* h_rems, uint32_t* d_rems) {
    if (count == 0 || n_items == 0 || n_items > 65535) return RK_ERR_INVALID;
    size_t nchunks = (count + DIV_CH - 1) / DIV_CH;
    if (nchunks > (size_t)DIV_NT * 4096) return RK_ERR_INVALID;
    std::vector<DivItem> items(n_items);
    for (size_t i = 0; i < n_items; i++) {
        items[i].z = h_z[i];
        items[i].zL = bb::pow(h_z[i], (uint64_t)DIV_CH, ctx->sys.wm);
        items[i].off = h_offsets[i];
        items[i].wm = ctx->sys.wm;
        items[i].pad_ = 0;
    }
    void* buf = nullptr;
    size_t items_bytes = (n_items * sizeof(DivItem) + 15) & ~(size_t)15;
    RK_TRY(dev_alloc(ctx, items_bytes + n_items * (2 * nchunks + 1) * 16, &buf));
    DivItem* d_items = (DivItem*)buf;
    uint32_t* tops = (uint32_t*)((char*)buf + items_bytes);
    uint32_t* carry = tops + n_items * nchunks * 4;
    uint32_t* rem = carry + n_items * nchunks * 4;
    int st = RK_OK;
    do {
        hipError_t e = hipSuccess;
        st = upload(ctx, d_items, items.data(), n_items * sizeof(DivItem));
        if (st != RK_OK) break;
        unsigned blocks = (unsigned)((nchunks + TPB - 1) / TPB);
        KTimer kt(ctx, RK_KCLASS_POLY, (double)count * 48 * n_items);
        hipLaunchKernelGGL(div_tops_kernel, dim3(blocks, (unsigned)n_items), dim3(TPB), 0, ctx->stream, tops, d_base_ext,
                           count, nchunks, d_items);
        st = post_launch(ctx, "div_tops_kernel");
        if (st != RK_OK) break;
        hipLaunchKernelGGL(div_carry_kernel, dim3(1, (unsigned)n_items), dim3(DIV_NT), 0, ctx->stream, carry, tops, nchunks,
                           d_items, rem);
        st = post_launch(ctx, "div_carry_kernel");
        if (st != RK_OK) break;
        hipLaunchKernelGGL(div_apply_kernel, dim3(blocks, (unsigned)n_items), dim3(TPB), 0, ctx->stream, d_base_ext, carry,
                           count, nchunks, d_items);
        st = post_launch(ctx, "div_apply_kernel");
        if (st != RK_OK) break;
        if (d_rems) {  // remainders stay on the device: the caller reads several rounds' worth at once
            e = hipMemcpyAsync(d_rems, rem, n_items * 16, hipMemcpyDeviceToDevice, ctx->stream);
            if (e != hipSuccess) {
                ctx->last_error = std::string("poly_divide d2d: ") + hipGetErrorString(e);
                st = RK_ERR_HIP;
            }
        }
        if (h_rems && st == RK_OK) {
            e = hipMemcpyAsync(h_rems, rem, n_items * 16, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) {
                ctx->last_error = std::string("poly_divide d2h: ") + hipGetErrorString(e);
                st = RK_ERR_HIP;
            }
        }
    } while (0);
    dev_free(ctx, buf);
    return st;
}
int poly_divide(rk_ctx* ctx, uint32_t* d_poly_ext, size_t count, const bb::Ext& z, bb::Ext* h_rem) {
    size_t off = 0;
    return poly_divide_many(ctx, d_poly_ext, count, &off, &z, 1, h_rem, nullptr);
}

int ext_sub_at(rk_ctx* ctx, uint32_t* d_ext, const uint32_t* h_idx, const bb::Ext* h_delta, size_t n) {
    if (n == 0) return RK_OK;
    size_t o_delta = (n + 3) & ~(size_t)3;
    std::vector<uint32_t> pack(o_delta + n * 4);
    std::memcpy(pack.data(), h_idx, n * 4);
    std::memcpy(&pack[o_delta], h_delta, n * 16);
    void* d = nullptr;
    RK_TRY(scratch(ctx, pack.size() * 4, &d));
    RK_TRY(upload(ctx, d, pack.data(), pack.size() * 4));
    const uint32_t* dp = (const uint32_t*)d;
    hipLaunchKernelGGL(ext_sub_at_kernel, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, ctx->stream, d_ext, dp,
                       dp + o_delta, n);
    return post_launch(ctx, "ext_sub_at_kernel");
}

}  // namespace rk

extern "C" {

int rk_eltwise_add_elem(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_a, const uint32_t* d_b, size_t n) {
    RK_GUARD_BEGIN
    if (!ctx || (n && (!d_out || !d_a || !d_b))) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::eltwise_add(ctx, d_out, d_a, d_b, n);
    RK_GUARD_END
}
int rk_eltwise_sum_extelem(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in_ext, size_t count, size_t to_add) {
    RK_GUARD_BEGIN
    if (!ctx || (count && (!d_out || !d_in_ext))) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::eltwise_sum_ext(ctx, d_out, d_in_ext, count, to_add);
    RK_GUARD_END
}
int rk_eltwise_copy_elem(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t n) {
    RK_GUARD_BEGIN
    if (!ctx || (n && (!d_out || !d_in))) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n)
        RK_HIP_TRY(ctx, hipMemcpyAsync(d_out, d_in, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    return RK_OK;
    RK_GUARD_END
}
int rk_eltwise_zeroize_elem(rk_ctx* ctx, uint32_t* d_io, size_t n) {
    RK_GUARD_BEGIN
    if (!ctx || (n && !d_io)) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::eltwise_zeroize(ctx, d_io, n);
    RK_GUARD_END
}
int rk_fri_fold(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t out_count, const uint32_t mix[4]) {
    RK_GUARD_BEGIN
    if (!ctx || !d_out || !d_in || !mix) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    bb::Ext m{{mix[0], mix[1], mix[2], mix[3]}};
    return rk::fri_fold(ctx, d_out, d_in, out_count, m);
    RK_GUARD_END
}
int rk_fri_fold_evals(rk_ctx* ctx, uint32_t* d_out_ext, const uint32_t* d_in_ext, size_t n_out, const uint32_t beta[4]) {
    RK_GUARD_BEGIN
    if (!ctx || !d_out_ext || !d_in_ext || !beta || n_out == 0) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    bb::Ext b{{beta[0], beta[1], beta[2], beta[3]}};
    return rk::fri_fold_evals(ctx, d_out_ext, d_in_ext, n_out, b);
    RK_GUARD_END
}
int rk_gather_sample(rk_ctx* ctx, uint32_t* d_dst, const uint32_t* d_src, size_t idx, size_t size, size_t stride) {
    RK_GUARD_BEGIN
    if (!ctx || (size && (!d_dst || !d_src))) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::gather_sample(ctx, d_dst, d_src, idx, size, stride);
    RK_GUARD_END
}

int rk_batch_evaluate_any(rk_ctx* ctx, const uint32_t* d_coeffs, size_t poly_count, size_t size,
                          const uint32_t* h_which, const uint32_t* h_xs, size_t eval_count, uint32_t* h_out) {
    RK_GUARD_BEGIN
    if (!ctx || !d_coeffs || !is_pow2(size)) return RK_ERR_INVALID;
    if (eval_count == 0) return RK_OK;
    if (!h_which || !h_xs || !h_out) return RK_ERR_INVALID;
    for (size_t e = 0; e < eval_count; e++)
        if (h_which[e] >= poly_count) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // distinct evaluation points share one device table of powers
    std::vector<bb::Ext> pts;
    std::vector<uint32_t> sel(eval_count);
    for (size_t e = 0; e < eval_count; e++) {
        bb::Ext x{{h_xs[e * 4], h_xs[e * 4 + 1], h_xs[e * 4 + 2], h_xs[e * 4 + 3]}};
        size_t j = 0;
        for (; j < pts.size(); j++)
            if (bb::eq(pts[j], x)) break;
        if (j == pts.size()) pts.push_back(x);
        sel[e] = (uint32_t)j;
    }
    void *d_pw = nullptr, *d_small = nullptr;
    RK_TRY(rk::dev_alloc(ctx, pts.size() * size * 16, &d_pw));
    int st = rk::dev_alloc(ctx, eval_count * (4 + 4 + 16) + 32, &d_small);
    if (st != RK_OK) {
        rk::dev_free(ctx, d_pw);
        return st;
    }
    uint32_t* d_which = (uint32_t*)d_small;
    uint32_t* d_sel = d_which + eval_count;
    uint32_t* d_out = d_sel + eval_count;
    // keep d_out 16-byte aligned
    if (((uintptr_t)d_out & 15) != 0) d_out += (16 - ((uintptr_t)d_out & 15)) / 4;
    do {
        hipError_t e1 = hipMemcpyAsync(d_which, h_which, eval_count * 4, hipMemcpyHostToDevice, ctx->stream);
        hipError_t e2 = hipMemcpyAsync(d_sel, sel.data(), eval_count * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e1 != hipSuccess || e2 != hipSuccess) { st = RK_ERR_HIP; ctx->last_error = "evaluate_any h2d"; break; }
        for (size_t j = 0; j < pts.size() && st == RK_OK; j++)
            st = rk::ext_powers(ctx, (uint32_t*)d_pw + j * size * 4, pts[j], size, false);
        if (st != RK_OK) break;
        st = rk::eval_dot(ctx, d_out, d_coeffs, size, d_which, (const uint32_t*)d_pw, d_sel, eval_count);
        if (st != RK_OK) break;
        hipError_t e3 = hipMemcpyAsync(h_out, d_out, eval_count * 16, hipMemcpyDeviceToHost, ctx->stream);
        if (e3 == hipSuccess) e3 = hipStreamSynchronize(ctx->stream);
        if (e3 != hipSuccess) { st = RK_ERR_HIP; ctx->last_error = "evaluate_any d2h"; }
    } while (0);
    if (st != RK_OK) (void)hipStreamSynchronize(ctx->stream);
    rk::dev_free(ctx, d_small);
    rk::dev_free(ctx, d_pw);
    return st;
    RK_GUARD_END
}

int rk_mix_poly_coeffs(rk_ctx* ctx, uint32_t* d_out_ext, const uint32_t mix_start[4], const uint32_t mix[4],
                       const uint32_t* d_in, const uint32_t* h_combos, size_t input_size, size_t count) {
    RK_GUARD_BEGIN
    if (!ctx || !d_out_ext || !d_in || !h_combos || !mix_start || !mix) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    bb::Ext ms{{mix_start[0], mix_start[1], mix_start[2], mix_start[3]}};
    bb::Ext mx{{mix[0], mix[1], mix[2], mix[3]}};
    return rk::mix_poly_coeffs(ctx, d_out_ext, ms, mx, d_in, h_combos, input_size, count);
    RK_GUARD_END
}

int rk_poly_divide(rk_ctx* ctx, uint32_t* d_polys_ext, size_t count, const uint32_t z[4], uint32_t* h_rem) {
    RK_GUARD_BEGIN
    if (!ctx || !d_polys_ext || !z || count == 0) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    bb::Ext zz{{z[0], z[1], z[2], z[3]}};
    bb::Ext rem;
    RK_TRY(rk::poly_divide(ctx, d_polys_ext, count, zz, &rem));
    if (h_rem) std::memcpy(h_rem, rem.c, 16);
    return RK_OK;
    RK_GUARD_END
}

}  // extern "C"
