// The data-parallel steps of Plonky3's two-adic FRI PCS on row-major matrices (p3-fri two_adic_pcs.rs commit / open at
// the revision SP1 pins -- Plonky3@88ea2b8, reference Cargo.lock:4889-5127, reached from
// provers/sp1/driver/src/lib.rs:48-57; RECALLED, the crates are outside the reference tree):
//   rk_pcs_coset_lde_rows     commit: coset LDE of every column, rows in bit-reversed order
//   rk_pcs_eval_at            open: the opened values p_c(z) from the low coset of the LDE (barycentric form)
//   rk_pcs_reduce_openings    open: ro[r] += alpha^(..) (sum_c alpha^c M[r][c] - sum_c alpha^c p_c(z)) / (x_r - z)
// Together with rk_mmcs_commit / rk_mmcs_open (the tree), rk_fri_fold_evals (the commit phase's fold) and
// rk_pow_grind they are what TwoAdicFriPcs spends its time in.  All HBM-bound: the LDE goes through the
// column-major NTT kernels between two tiled transposes (128-byte segments on both sides), the other two read
// the matrix once (lanes along the rows for the loads, one lane per row or per column for the sums).
#include "internal.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

namespace {

using bb::Ext;
constexpr int TPB = 256;
constexpr int TT = 32;  // transpose tile

__device__ __forceinline__ Ext load_ext(const uint32_t* p) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    return Ext{{v.x, v.y, v.z, v.w}};
}
__device__ __forceinline__ void store_ext(uint32_t* p, const Ext& e) {
    *reinterpret_cast<uint4*>(p) = make_uint4(e.c[0], e.c[1], e.c[2], e.c[3]);
}

// row-major h x w -> column-major w x h.  Block (32, 8), tile 32 rows x 32 columns.
__global__ void rows_to_cols_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t h, size_t w) {
    __shared__ uint32_t tile[TT][TT + 1];
    const size_t r0 = (size_t)blockIdx.x * TT, c0 = (size_t)blockIdx.y * TT;
    for (unsigned y = threadIdx.y; y < TT; y += blockDim.y) {
        const size_t r = r0 + y, c = c0 + threadIdx.x;
        if (r < h && c < w) tile[y][threadIdx.x] = src[r * w + c];
    }
    __syncthreads();
    for (unsigned y = threadIdx.y; y < TT; y += blockDim.y) {
        const size_t c = c0 + y, r = r0 + threadIdx.x;
        if (r < h && c < w) dst[c * h + r] = tile[threadIdx.x][y];
    }
}
// column-major w x H -> row-major H x w with dst row bitrev(j) = src index j: a tile of 32 consecutive j lands in
// 32 rows H / 32 apart, each as one run of 32 columns
__global__ void cols_to_rows_bitrev_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t H, size_t w,
                                           unsigned bits) {
    __shared__ uint32_t tile[TT][TT + 1];
    const size_t j0 = (size_t)blockIdx.x * TT, c0 = (size_t)blockIdx.y * TT;
    for (unsigned y = threadIdx.y; y < TT; y += blockDim.y) {
        const size_t c = c0 + y, j = j0 + threadIdx.x;
        if (j < H && c < w) tile[y][threadIdx.x] = src[c * H + j];
    }
    __syncthreads();
    for (unsigned y = threadIdx.y; y < TT; y += blockDim.y) {
        const size_t j = j0 + y, c = c0 + threadIdx.x;
        if (j < H && c < w) dst[(size_t)bb::bitrev((uint32_t)j, bits) * w + c] = tile[threadIdx.x][y];
    }
}

// barycentric weights at the LDE's row positions, blockIdx.y = point: wts[p][bitrev_k(i)] = g^i / (z_p - s g^i), i < h = 2^k
struct BaryPoint {
    Ext z, scaling;  // the point; (z^h - s^h) / (h s^(h-1))
};
constexpr int BARY_MAX_POINTS = 4;
// natural != 0: stored at index i instead (for column-major matrices kept in natural order)
__global__ void bary_weights_kernel(uint32_t* __restrict__ wts, const BaryPoint* __restrict__ pts, size_t h, unsigned k, uint32_t shiftm,
                                    uint32_t wm, ntt::Tables tb, int natural) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h) return;
    const uint32_t gi = ntt::root_pow(tb, 0, (uint32_t)(i << (ntt::LAMBDA - k)));
    Ext d = pts[blockIdx.y].z;
    d.c[0] = bb::sub(d.c[0], bb::mul(shiftm, gi));
    store_ext(wts + ((size_t)blockIdx.y * h + (natural ? i : bb::bitrev((uint32_t)i, k))) * 4, bb::scale(bb::inv(d, wm), gi));
}
// The same sums for a COLUMN-major matrix in natural order (w columns of H words; the low coset is every 2^blow-th
// element): lanes along i, every lane carries BC columns of its i's so that a weight is loaded once per BC matrix words;
// partial[chunk][p][c] as above.
constexpr int BC = 8, BARYC_THREADS = 256;
template <int NP>
__global__ __launch_bounds__(BARYC_THREADS) void bary_dot_cols_kernel(uint32_t* __restrict__ partial, const uint32_t* __restrict__ wts,
                                                                      const uint32_t* __restrict__ M, size_t h, size_t H, size_t w, unsigned blow,
                                                                      size_t rows_per_chunk) {
    __shared__ uint32_t red[BARYC_THREADS / 64][NP * BC * 4];
    const size_t i_begin = (size_t)blockIdx.x * rows_per_chunk, i_end = i_begin + rows_per_chunk < h ? i_begin + rows_per_chunk : h;
    const size_t c0 = (size_t)blockIdx.y * BC;
    Ext acc[NP][BC];
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
        for (int cc = 0; cc < BC; cc++) acc[q][cc] = bb::ext_zero();
    for (size_t i = i_begin + threadIdx.x; i < i_end; i += BARYC_THREADS) {
        Ext wq[NP];
#pragma unroll
        for (int q = 0; q < NP; q++) wq[q] = load_ext(wts + ((size_t)q * h + i) * 4);
#pragma unroll
        for (int cc = 0; cc < BC; cc++) {
            if (c0 + cc >= w) break;
            const uint32_t m = M[(c0 + cc) * H + (i << blow)];
#pragma unroll
            for (int q = 0; q < NP; q++) acc[q][cc] = bb::add(acc[q][cc], bb::scale(wq[q], m));
        }
    }
    // the wave's sum by butterfly shuffles, then the block's through LDS
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
        for (int cc = 0; cc < BC; cc++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                uint32_t v = acc[q][cc].c[e];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) v = bb::add(v, (uint32_t)__shfl_xor((int)v, off, 64));
                if (lane == 0) red[wave][(q * BC + cc) * 4 + e] = v;
            }
    __syncthreads();
    if (threadIdx.x < NP * BC * 4) {
        uint32_t v = red[0][threadIdx.x];
        for (unsigned wv = 1; wv < BARYC_THREADS / 64; wv++) v = bb::add(v, red[wv][threadIdx.x]);
        const unsigned q = threadIdx.x / (BC * 4), cc = (threadIdx.x / 4) % BC, e = threadIdx.x & 3;
        if (c0 + cc < w) partial[(((size_t)blockIdx.x * NP + q) * w + c0 + cc) * 4 + e] = v;
    }
}
// partial[chunk][p][c] = sum over the chunk's rows of wts[p][r] * M[r][c]: lanes along the row (coalesced), the rows of
// a chunk split over the blockDim.y sub-rows, combined through LDS; the matrix is read once for all NP points
constexpr int BARY_X = 64, BARY_Y = 4;
template <int NP>
__global__ void bary_dot_kernel(uint32_t* __restrict__ partial, const uint32_t* __restrict__ wts, const uint32_t* __restrict__ M,
                                size_t h, size_t w, size_t rows_per_chunk) {
    __shared__ uint32_t red[BARY_Y][BARY_X][4];
    const size_t r_begin = (size_t)blockIdx.x * rows_per_chunk, r_end = r_begin + rows_per_chunk < h ? r_begin + rows_per_chunk : h;
    for (size_t c0 = 0; c0 < w; c0 += BARY_X) {
        const size_t c = c0 + threadIdx.x;
        Ext acc[NP];
#pragma unroll
        for (int q = 0; q < NP; q++) acc[q] = bb::ext_zero();
        if (c < w)
            for (size_t r = r_begin + threadIdx.y; r < r_end; r += BARY_Y) {
                const uint32_t m = M[r * w + c];
#pragma unroll
                for (int q = 0; q < NP; q++) acc[q] = bb::add(acc[q], bb::scale(load_ext(wts + ((size_t)q * h + r) * 4), m));
            }
#pragma unroll
        for (int q = 0; q < NP; q++) {
#pragma unroll
            for (int t = 0; t < 4; t++) red[threadIdx.y][threadIdx.x][t] = acc[q].c[t];
            __syncthreads();
            if (threadIdx.y == 0 && c < w) {
                Ext tot = acc[q];
                for (int y = 1; y < BARY_Y; y++)
#pragma unroll
                    for (int t = 0; t < 4; t++) tot.c[t] = bb::add(tot.c[t], red[y][threadIdx.x][t]);
                store_ext(partial + (((size_t)blockIdx.x * NP + q) * w + c) * 4, tot);
            }
            __syncthreads();
        }
    }
}
// out[p][c] = scaling_p * sum over chunks of partial[chunk][p][c]: blockIdx.y = point, 64 columns x 16 chunk classes
// per block, combined through LDS
constexpr int FIN_X = 64, FIN_Y = 16;
__global__ void bary_finish_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ partial, size_t chunks, size_t w,
                                   const BaryPoint* __restrict__ pts, unsigned n_points, uint32_t wm) {
    __shared__ uint32_t red[FIN_Y][FIN_X][4];
    const size_t c = (size_t)blockIdx.x * FIN_X + threadIdx.x;
    const unsigned q = blockIdx.y;
    Ext acc = bb::ext_zero();
    if (c < w)
        for (size_t k = threadIdx.y; k < chunks; k += FIN_Y) acc = bb::add(acc, load_ext(partial + ((k * n_points + q) * w + c) * 4));
#pragma unroll
    for (int t = 0; t < 4; t++) red[threadIdx.y][threadIdx.x][t] = acc.c[t];
    __syncthreads();
    if (threadIdx.y != 0 || c >= w) return;
    for (int y = 1; y < FIN_Y; y++)
#pragma unroll
        for (int t = 0; t < 4; t++) acc.c[t] = bb::add(acc.c[t], red[y][threadIdx.x][t]);
    store_ext(out + ((size_t)q * w + c) * 4, bb::mul(acc, pts[q].scaling, wm));
}

// One lane per row.  A block takes 256 rows; the matrix goes through LDS in tiles of 256 rows x 32 columns so that
// the global loads are 128-byte runs along the rows while every lane walks its own row (stride 33: conflict-free);
// the powers of alpha of a tile sit next to it; the next tile travels from HBM into registers meanwhile.  Then the
// lane adds its row's quotient for every point -- one extension inversion per point, all lanes at once.
struct PcsPoint {
    Ext z, coef, rys;  // the point, alpha^(offset + j w), sum_c alpha^c p_c(z)
};
constexpr int PCS_MAX_POINTS = 8;
constexpr int RO_ROWS = 256, RO_COLS = 32;
__global__ __launch_bounds__(RO_ROWS) void reduce_openings_kernel(uint32_t* __restrict__ ro, const uint32_t* __restrict__ M, size_t H,
                                                                  size_t w, unsigned bits, const uint32_t* __restrict__ apow,
                                                                  const PcsPoint* __restrict__ pts, unsigned n_points, uint32_t shiftm,
                                                                  uint32_t wm, ntt::Tables tb) {
    __shared__ uint32_t tile[RO_ROWS][RO_COLS + 1];
    __shared__ __attribute__((aligned(16))) uint32_t ap[RO_COLS][4];  // one 16-byte broadcast read per column
    const unsigned tid = threadIdx.x;
    const size_t r0 = (size_t)blockIdx.x * RO_ROWS, r = r0 + tid;
    Ext acc = bb::ext_zero();
    // the next tile's words are fetched into registers while the current tile is being summed out of LDS
    constexpr int PER = RO_COLS;  // RO_ROWS * RO_COLS words / RO_ROWS lanes
    uint32_t nxt[PER];
    auto fetch = [&](size_t c0) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const unsigned e = tid + (unsigned)k * RO_ROWS, row = e / RO_COLS, col = e % RO_COLS;
            nxt[k] = (r0 + row < H && c0 + col < w) ? M[(r0 + row) * w + c0 + col] : 0u;
        }
    };
    fetch(0);
    for (size_t c0 = 0; c0 < w; c0 += RO_COLS) {
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const unsigned e = tid + (unsigned)k * RO_ROWS;
            tile[e / RO_COLS][e % RO_COLS] = nxt[k];
        }
        if (tid < RO_COLS * 4) ap[tid >> 2][tid & 3] = (c0 + (tid >> 2) < w) ? apow[(c0 + (tid >> 2)) * 4 + (tid & 3)] : 0u;
        __syncthreads();
        if (c0 + RO_COLS < w) fetch(c0 + RO_COLS);
#pragma unroll 8
        for (int col = 0; col < RO_COLS; col++) {
            const uint32_t m = tile[tid][col];
            const uint4 a4 = *reinterpret_cast<const uint4*>(ap[col]);
            acc.c[0] = bb::add(acc.c[0], bb::mul(a4.x, m));
            acc.c[1] = bb::add(acc.c[1], bb::mul(a4.y, m));
            acc.c[2] = bb::add(acc.c[2], bb::mul(a4.z, m));
            acc.c[3] = bb::add(acc.c[3], bb::mul(a4.w, m));
        }
        __syncthreads();
    }
    if (r >= H) return;
    const uint32_t x = bb::mul(shiftm, ntt::root_pow(tb, 0, bb::bitrev((uint32_t)r, bits) << (ntt::LAMBDA - bits)));
    Ext out = load_ext(ro + r * 4);
    for (unsigned j = 0; j < n_points; j++) {
        const PcsPoint p = pts[j];
        const Ext den = Ext{{bb::sub(x, p.z.c[0]), bb::neg(p.z.c[1]), bb::neg(p.z.c[2]), bb::neg(p.z.c[3])}};
        out = bb::add(out, bb::mul(p.coef, bb::mul(bb::sub(acc, p.rys), bb::inv(den, wm), wm), wm));
    }
    store_ext(ro + r * 4, out);
}

// "reduce rows" for a COLUMN-major matrix in natural order: lane = natural index j, the row's dot product with the powers
// of alpha built from coalesced column loads (the powers come through the scalar cache: the column index is uniform),
// the result added to ro at the committed row bitrev(j)
__global__ __launch_bounds__(256) void reduce_openings_cols_kernel(uint32_t* __restrict__ ro, const uint32_t* __restrict__ M, size_t H, size_t w,
                                                                   unsigned bits, uint64_t apow_addr, const PcsPoint* __restrict__ pts,
                                                                   unsigned n_points, uint32_t shiftm, uint32_t wm, ntt::Tables tb) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= H) return;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) uint32_t* const_u32;
#else
    typedef const uint32_t* const_u32;
#endif
    const const_u32 apow = (const_u32)apow_addr;
    Ext acc = bb::ext_zero();
    for (size_t c = 0; c < w; c++) {
        const uint32_t m = M[c * H + j];
        acc.c[0] = bb::add(acc.c[0], bb::mul(apow[4 * c + 0], m));
        acc.c[1] = bb::add(acc.c[1], bb::mul(apow[4 * c + 1], m));
        acc.c[2] = bb::add(acc.c[2], bb::mul(apow[4 * c + 2], m));
        acc.c[3] = bb::add(acc.c[3], bb::mul(apow[4 * c + 3], m));
    }
    const uint32_t x = bb::mul(shiftm, ntt::root_pow(tb, 0, (uint32_t)j << (ntt::LAMBDA - bits)));
    uint32_t* dst = ro + (size_t)bb::bitrev((uint32_t)j, bits) * 4;
    Ext out = load_ext(dst);
    for (unsigned q = 0; q < n_points; q++) {
        const PcsPoint p = pts[q];
        const Ext den = Ext{{bb::sub(x, p.z.c[0]), bb::neg(p.z.c[1]), bb::neg(p.z.c[2]), bb::neg(p.z.c[3])}};
        out = bb::add(out, bb::mul(p.coef, bb::mul(bb::sub(acc, p.rys), bb::inv(den, wm), wm), wm));
    }
    store_ext(dst, out);
}

}  // namespace

namespace rk {

int pcs_cols_to_rows_bitrev(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_cols, size_t H, size_t w) {
    if (!is_pow2(H) || w == 0 || log2u(H) > ntt::LAMBDA) return RK_ERR_INVALID;
    KTimer kt(ctx, RK_KCLASS_BIT_REVERSE, (double)H * w * 8);
    hipLaunchKernelGGL(cols_to_rows_bitrev_kernel, dim3((unsigned)((H + TT - 1) / TT), (unsigned)((w + TT - 1) / TT)), dim3(TT, 8), 0,
                       ctx->stream, d_out, d_cols, H, w, log2u(H));
    return post_launch(ctx, "cols_to_rows_bitrev_kernel");
}

int pcs_coset_lde_rows(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t h, size_t w, uint32_t** keep_cols) {
    const unsigned blow = ctx->sys.blowup_log2;
    if (!is_pow2(h) || h < 2 || w == 0 || log2u(h) + blow > 24) return RK_ERR_INVALID;
    const size_t H = h << blow;
    void *a = nullptr, *b = nullptr;
    RK_TRY(dev_alloc(ctx, h * w * 4, &a));
    int st = dev_alloc(ctx, H * w * 4, &b);
    if (st == RK_OK) {
        KTimer kt(ctx, RK_KCLASS_BIT_REVERSE, (double)h * w * 8);
        hipLaunchKernelGGL(rows_to_cols_kernel, dim3((unsigned)((h + TT - 1) / TT), (unsigned)((w + TT - 1) / TT)), dim3(TT, 8), 0,
                           ctx->stream, (uint32_t*)a, d_in, h, w);
        st = post_launch(ctx, "rows_to_cols_kernel");
    }
    // the columns' interpolating polynomials (coefficient i times shift^i, bit-reversed), then their values on the
    // coset of the larger subgroup -- the same two calls a risc0 trace group goes through
    if (st == RK_OK) st = ntt_reverse(ctx, (uint32_t*)a, h, w, /*fuse_zk_shift=*/true);
    if (st == RK_OK) st = ntt_forward(ctx, (uint32_t*)b, (const uint32_t*)a, h, w, blow);
    if (st == RK_OK) st = pcs_cols_to_rows_bitrev(ctx, d_out, (const uint32_t*)b, H, w);
    if (b && (st != RK_OK || !keep_cols)) dev_free(ctx, b);
    if (st == RK_OK && keep_cols) *keep_cols = (uint32_t*)b;
    dev_free(ctx, a);
    return st;
}

// the coset LDE of a row-major h x w matrix as `w` columns of H = h << blow-up evaluations in natural order (what the
// two NTTs leave; rk_matrix layout 2 commits it as it is)
int pcs_coset_lde_cols(rk_ctx* ctx, uint32_t* d_cols, const uint32_t* d_in, size_t h, size_t w) {
    const unsigned blow = ctx->sys.blowup_log2;
    if (!is_pow2(h) || h < 2 || w == 0 || log2u(h) + blow > 24) return RK_ERR_INVALID;
    void* a = nullptr;
    RK_TRY(dev_alloc(ctx, h * w * 4, &a));
    int st;
    {
        KTimer kt(ctx, RK_KCLASS_BIT_REVERSE, (double)h * w * 8);
        hipLaunchKernelGGL(rows_to_cols_kernel, dim3((unsigned)((h + TT - 1) / TT), (unsigned)((w + TT - 1) / TT)), dim3(TT, 8), 0,
                           ctx->stream, (uint32_t*)a, d_in, h, w);
        st = post_launch(ctx, "rows_to_cols_kernel");
    }
    if (st == RK_OK) st = ntt_reverse(ctx, (uint32_t*)a, h, w, /*fuse_zk_shift=*/true);
    if (st == RK_OK) st = ntt_forward(ctx, d_cols, (const uint32_t*)a, h, w, blow);
    dev_free(ctx, a);
    return st;
}

int pcs_eval_at(rk_ctx* ctx, uint32_t* d_out_ext, const uint32_t* d_lde, size_t H, size_t w, const uint32_t* h_points, size_t n_points,
                bool cols) {
    const unsigned blow = ctx->sys.blowup_log2;
    if (!is_pow2(H) || (H >> blow) < 1 || w == 0 || log2u(H) > 24 || n_points == 0 || n_points > BARY_MAX_POINTS) return RK_ERR_INVALID;
    const size_t h = H >> blow;
    const unsigned k = log2u(h);
    const uint32_t wm = ctx->sys.wm, s = ctx->sys.shiftm;
    // per point (z^h - s^h) / (h s^(h-1))
    const uint32_t denom_inv = bb::inv(bb::mul(bb::encode((uint32_t)(h % bb::P)), bb::pow(s, (uint64_t)h - 1)));
    BaryPoint pts[BARY_MAX_POINTS];
    for (size_t q = 0; q < n_points; q++) {
        std::memcpy(pts[q].z.c, h_points + 4 * q, 16);
        bb::Ext zer = bb::pow(pts[q].z, (uint64_t)h, wm);
        zer.c[0] = bb::sub(zer.c[0], bb::pow(s, (uint64_t)h));
        pts[q].scaling = bb::scale(zer, denom_inv);
    }
    const size_t rows_per_chunk = cols ? std::max<size_t>(BARYC_THREADS, (h + 63) / 64) : std::max<size_t>(BARY_Y * 16, (h + 1023) / 1024);
    const size_t chunks = (h + rows_per_chunk - 1) / rows_per_chunk;
    void* buf = nullptr;
    RK_TRY(dev_alloc(ctx, n_points * h * 16 + chunks * n_points * w * 16 + sizeof pts, &buf));
    uint32_t* wts = (uint32_t*)buf;
    uint32_t* partial = wts + n_points * h * 4;
    BaryPoint* d_pts = (BaryPoint*)(partial + chunks * n_points * w * 4);
    int st = upload(ctx, d_pts, pts, n_points * sizeof(BaryPoint));
    if (st == RK_OK) {
        KTimer kt(ctx, RK_KCLASS_POLY, (double)h * w * 4 + (double)h * 32 * n_points);
        hipLaunchKernelGGL(bary_weights_kernel, dim3((unsigned)((h + TPB - 1) / TPB), (unsigned)n_points), dim3(TPB), 0, ctx->stream, wts,
                           (const BaryPoint*)d_pts, h, k, s, wm, ctx->tb, cols ? 1 : 0);
        st = post_launch(ctx, "bary_weights_kernel");
        if (st == RK_OK && cols) {
            const dim3 grid((unsigned)chunks, (unsigned)((w + BC - 1) / BC)), block(BARYC_THREADS);
            switch (n_points) {
                case 1: hipLaunchKernelGGL(bary_dot_cols_kernel<1>, grid, block, 0, ctx->stream, partial, (const uint32_t*)wts, d_lde, h, H, w, blow, rows_per_chunk); break;
                case 2: hipLaunchKernelGGL(bary_dot_cols_kernel<2>, grid, block, 0, ctx->stream, partial, (const uint32_t*)wts, d_lde, h, H, w, blow, rows_per_chunk); break;
                case 3: hipLaunchKernelGGL(bary_dot_cols_kernel<3>, grid, block, 0, ctx->stream, partial, (const uint32_t*)wts, d_lde, h, H, w, blow, rows_per_chunk); break;
                default: hipLaunchKernelGGL(bary_dot_cols_kernel<4>, grid, block, 0, ctx->stream, partial, (const uint32_t*)wts, d_lde, h, H, w, blow, rows_per_chunk); break;
            }
            st = post_launch(ctx, "bary_dot_cols_kernel");
        } else if (st == RK_OK) {
            const dim3 grid((unsigned)chunks), block(BARY_X, BARY_Y);
            switch (n_points) {
                case 1: hipLaunchKernelGGL(bary_dot_kernel<1>, grid, block, 0, ctx->stream, partial, (const uint32_t*)wts, d_lde, h, w, rows_per_chunk); break;
                case 2: hipLaunchKernelGGL(bary_dot_kernel<2>, grid, block, 0, ctx->stream, partial, (const uint32_t*)wts, d_lde, h, w, rows_per_chunk); break;
                case 3: hipLaunchKernelGGL(bary_dot_kernel<3>, grid, block, 0, ctx->stream, partial, (const uint32_t*)wts, d_lde, h, w, rows_per_chunk); break;
                default: hipLaunchKernelGGL(bary_dot_kernel<4>, grid, block, 0, ctx->stream, partial, (const uint32_t*)wts, d_lde, h, w, rows_per_chunk); break;
            }
            st = post_launch(ctx, "bary_dot_kernel");
        }
        if (st == RK_OK) {
            hipLaunchKernelGGL(bary_finish_kernel, dim3((unsigned)((w + FIN_X - 1) / FIN_X), (unsigned)n_points), dim3(FIN_X, FIN_Y), 0, ctx->stream,
                               d_out_ext, (const uint32_t*)partial, chunks, w, (const BaryPoint*)d_pts, (unsigned)n_points, wm);
            st = post_launch(ctx, "bary_finish_kernel");
        }
    }
    dev_free(ctx, buf);
    return st;
}

int pcs_reduce_openings(rk_ctx* ctx, uint32_t* d_ro_ext, const uint32_t* d_lde, size_t H, size_t w, size_t n_points,
                        const uint32_t* h_points, const uint32_t* h_ys, const bb::Ext& alpha, uint64_t alpha_offset, bool cols) {
    if (!is_pow2(H) || w == 0 || log2u(H) > 24 || n_points == 0 || n_points > PCS_MAX_POINTS) return RK_ERR_INVALID;
    const uint32_t wm = ctx->sys.wm;
    std::vector<uint32_t> pack(w * 4 + n_points * (sizeof(PcsPoint) / 4));
    bb::Ext cur = bb::ext_one();
    for (size_t c = 0; c < w; c++) {
        std::memcpy(&pack[c * 4], cur.c, 16);
        cur = bb::mul(cur, alpha, wm);
    }
    PcsPoint* pp = (PcsPoint*)&pack[w * 4];
    for (size_t j = 0; j < n_points; j++) {
        std::memcpy(pp[j].z.c, h_points + 4 * j, 16);
        bb::Ext rys = bb::ext_zero();
        for (size_t c = 0; c < w; c++) {
            bb::Ext y, a;
            std::memcpy(y.c, h_ys + (j * w + c) * 4, 16);
            std::memcpy(a.c, &pack[c * 4], 16);
            rys = bb::add(rys, bb::mul(a, y, wm));
        }
        pp[j].rys = rys;
        pp[j].coef = bb::pow(alpha, alpha_offset + (uint64_t)j * w, wm);
    }
    void* d = nullptr;
    RK_TRY(scratch(ctx, pack.size() * 4, &d));
    RK_TRY(upload(ctx, d, pack.data(), pack.size() * 4));
    const uint32_t* dp = (const uint32_t*)d;
    const size_t blocks = (H + RO_ROWS - 1) / RO_ROWS;
    KTimer kt(ctx, RK_KCLASS_POLY, (double)H * w * 4 + (double)H * 32);
    if (cols) {
        hipLaunchKernelGGL(reduce_openings_cols_kernel, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, ctx->stream, d_ro_ext, d_lde, H, w, log2u(H),
                           (uint64_t)(uintptr_t)dp, (const PcsPoint*)(dp + w * 4), (unsigned)n_points, ctx->sys.shiftm, wm, ctx->tb);
        return post_launch(ctx, "reduce_openings_cols_kernel");
    }
    hipLaunchKernelGGL(reduce_openings_kernel, dim3((unsigned)blocks), dim3(RO_ROWS), 0, ctx->stream, d_ro_ext, d_lde, H, w, log2u(H), dp,
                       (const PcsPoint*)(dp + w * 4), (unsigned)n_points, ctx->sys.shiftm, wm, ctx->tb);
    return post_launch(ctx, "reduce_openings_kernel");
}

}  // namespace rk

extern "C" {

int rk_pcs_coset_lde_rows(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t height, size_t width) {
    RK_GUARD_BEGIN
    if (!ctx || !d_out || !d_in) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::pcs_coset_lde_rows(ctx, d_out, d_in, height, width);
    RK_GUARD_END
}
int rk_pcs_eval_at(rk_ctx* ctx, uint32_t* d_out_ext, const uint32_t* d_lde, size_t lde_height, size_t width, const uint32_t z[4]) {
    RK_GUARD_BEGIN
    if (!ctx || !d_out_ext || !d_lde || !z) return RK_ERR_INVALID;
    for (int i = 0; i < 4; i++)
        if (z[i] >= bb::P) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::pcs_eval_at(ctx, d_out_ext, d_lde, lde_height, width, z, 1);
    RK_GUARD_END
}
int rk_pcs_eval_at_many(rk_ctx* ctx, uint32_t* d_out_ext, const uint32_t* d_lde, size_t lde_height, size_t width, uint32_t n_points,
                        const uint32_t* h_points) {
    RK_GUARD_BEGIN
    if (!ctx || !d_out_ext || !d_lde || !h_points || n_points == 0 || n_points > 4) return RK_ERR_INVALID;
    for (uint32_t i = 0; i < 4 * n_points; i++)
        if (h_points[i] >= bb::P) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::pcs_eval_at(ctx, d_out_ext, d_lde, lde_height, width, h_points, n_points);
    RK_GUARD_END
}
int rk_pcs_coset_lde_cols(rk_ctx* ctx, uint32_t* d_cols, const uint32_t* d_in_rows, size_t height, size_t width) {
    RK_GUARD_BEGIN
    if (!ctx || !d_cols || !d_in_rows) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::pcs_coset_lde_cols(ctx, d_cols, d_in_rows, height, width);
    RK_GUARD_END
}
int rk_pcs_eval_at_many_cols(rk_ctx* ctx, uint32_t* d_out_ext, const uint32_t* d_lde_cols, size_t lde_height, size_t width, uint32_t n_points,
                             const uint32_t* h_points) {
    RK_GUARD_BEGIN
    if (!ctx || !d_out_ext || !d_lde_cols || !h_points || n_points == 0 || n_points > 4) return RK_ERR_INVALID;
    for (uint32_t i = 0; i < 4 * n_points; i++)
        if (h_points[i] >= bb::P) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::pcs_eval_at(ctx, d_out_ext, d_lde_cols, lde_height, width, h_points, n_points, /*cols=*/true);
    RK_GUARD_END
}
int rk_pcs_reduce_openings_cols(rk_ctx* ctx, uint32_t* d_ro_ext, const uint32_t* d_lde_cols, size_t lde_height, size_t width, uint32_t n_points,
                                const uint32_t* h_points, const uint32_t* h_opened, const uint32_t alpha[4], uint64_t alpha_offset) {
    RK_GUARD_BEGIN
    if (!ctx || !d_ro_ext || !d_lde_cols || !h_points || !h_opened || !alpha) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::pcs_reduce_openings(ctx, d_ro_ext, d_lde_cols, lde_height, width, n_points, h_points, h_opened,
                                   bb::Ext{{alpha[0], alpha[1], alpha[2], alpha[3]}}, alpha_offset, /*cols=*/true);
    RK_GUARD_END
}
int rk_pcs_reduce_openings(rk_ctx* ctx, uint32_t* d_ro_ext, const uint32_t* d_lde, size_t lde_height, size_t width, uint32_t n_points,
                           const uint32_t* h_points, const uint32_t* h_opened, const uint32_t alpha[4], uint64_t alpha_offset) {
    RK_GUARD_BEGIN
    if (!ctx || !d_ro_ext || !d_lde || !h_points || !h_opened || !alpha) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::pcs_reduce_openings(ctx, d_ro_ext, d_lde, lde_height, width, n_points, h_points, h_opened,
                                   bb::Ext{{alpha[0], alpha[1], alpha[2], alpha[3]}}, alpha_offset);
    RK_GUARD_END
}

}  // extern "C"
