// Poseidon2 row hashing and Merkle folds (Hal::hash_rows / Hal::hash_fold and
// MerkleTreeProver::new of risc0-zkp 1.0.1).  Integer-ALU-bound: one lane owns one
// sponge (24 state words in VGPRs), consecutive lanes own consecutive rows so every
// column load is a coalesced 256-byte wave access of the column-major matrix.
#include <algorithm>
#include <cstring>

#include "internal.hpp"

namespace {

constexpr int HASH_BLOCK = 256;
// levels with at most this many parents go cell-parallel (hash_fold_top): a cell-parallel permutation costs ~6x the
// lane-cycles of a lane-per-parent one, so it only pays where the level is latency-bound anyway
constexpr size_t CELLS_MAX_OUT = rk::HASH_FOLD_TOP_MAX;

// C = the configured p2::Core (width 24 or 16, external 4x4 block): one kernel instance per Core.
// pad_free: the last partial block leaves the remaining rate cells as they are (Plonky3
// PaddingFreeSponge) instead of zero-padding them (risc0); an empty row then takes no permutation.
template <class C>
__global__ __launch_bounds__(HASH_BLOCK) void hash_rows_kernel(uint32_t* __restrict__ out,
                                                               const uint32_t* __restrict__ matrix, size_t rows,
                                                               size_t cols, const typename C::Consts* __restrict__ kc,
                                                               int pad_free) {
    size_t row = (size_t)blockIdx.x * HASH_BLOCK + threadIdx.x;
    if (row >= rows) return;
    const typename C::Consts& k = *kc;
    uint32_t s[C::CELLS];
#pragma unroll
    for (int i = 0; i < C::CELLS; i++) s[i] = 0;
    size_t full = cols / C::RATE;
    const uint32_t* src = matrix + row;
    for (size_t b = 0; b < full; b++) {
#pragma unroll
        for (int i = 0; i < C::RATE; i++) s[i] = src[(b * C::RATE + i) * rows];
        C::permute(s, k);
    }
    size_t rem = cols - full * C::RATE;
    if (rem != 0 || (cols == 0 && !pad_free)) {
#pragma unroll
        for (int i = 0; i < C::RATE; i++) {
            if ((size_t)i < rem) s[i] = src[(full * C::RATE + i) * rows];
            else if (!pad_free) s[i] = 0u;
        }
        C::permute(s, k);
    }
    uint4* o = reinterpret_cast<uint4*>(out + row * p2::OUT);
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// nodes[out_size + i] = H(nodes[2*(out_size+i)] || nodes[2*(out_size+i)+1]): the two digests fill
// cells 0..15, the rest of a wider state is zero; the parent is the first 8 cells
template <class C>
__global__ __launch_bounds__(HASH_BLOCK) void hash_fold_kernel(uint32_t* __restrict__ nodes, size_t out_size,
                                                               const typename C::Consts* __restrict__ kc) {
    size_t i = (size_t)blockIdx.x * HASH_BLOCK + threadIdx.x;
    if (i >= out_size) return;
    const typename C::Consts& k = *kc;
    size_t idx = out_size + i;
    const uint4* in = reinterpret_cast<const uint4*>(nodes + 2 * idx * p2::OUT);
    uint4 a = in[0], b = in[1], c = in[2], d = in[3];
    uint32_t s[C::CELLS];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    s[8] = c.x; s[9] = c.y; s[10] = c.z; s[11] = c.w; s[12] = d.x; s[13] = d.y; s[14] = d.z; s[15] = d.w;
#pragma unroll
    for (int j = 16; j < C::CELLS; j++) s[j] = 0;
    C::permute(s, k);
    uint4* o = reinterpret_cast<uint4*>(nodes + idx * p2::OUT);
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// ---- cell-parallel permutation: one 32-lane half-wave per permutation, one lane per cell --------------
// The levels near the root have too few parents to fill the chip, so a lane-per-permutation launch costs the
// latency of one permutation (~6.7 k dependent instructions, ~15 us) per level whatever its size.  Here the
// 24 (16) cells of one state sit in consecutive lanes, values canonical Montgomery residues:
//   * S-boxes run in all lanes at once (full rounds) or are kept by cell 0 only (partial rounds);
//   * external layer circ(2 M4, M4, ...): out = M4 (x_quad + X), X_j = sum over quads of cell j -- the quad
//     sums are two row rotations (DPP row_ror:4/8) and one exchange of the two 16-lane rows
//     (v_permlane16_swap, gfx950), the 4x4 product takes its operands by DPP quad broadcasts and
//     accumulates exactly in 64 bits (< 16 p), one REDC and one product by 2^64 bring it back;
//   * internal layer: the cell sum is a 5-step DPP reduction, then x_i = d_i x_i + S.
// ~40 dependent instructions per round instead of ~230, nothing goes through LDS.  Lanes beyond the width
// hold zero between layers so that they do not disturb the sums.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
constexpr int DPP_ROR1 = 0x121, DPP_ROR2 = 0x122, DPP_ROR4 = 0x124, DPP_ROR8 = 0x128;
constexpr int DPP_Q0 = 0x00, DPP_Q1 = 0x55, DPP_Q2 = 0xaa, DPP_Q3 = 0xff;
// lane i + lane (i ^ 16), both < p: the sum of the two rows of a half-wave, canonical
__device__ __forceinline__ uint32_t row_pair_sum(uint32_t v) {
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return bb::ucanon(r[0] + r[1]);
}
constexpr uint32_t R2_NQ = bb::R2 * (0u - bb::MPRIME);  // bb::umul_const companion of 2^64 mod p

template <class C>
struct CellPerm {
    static constexpr int W = C::CELLS;
    uint32_t m0, m1, m2, m3;  // this lane's row of the 4x4 block
    uint32_t diag;
    uint32_t rc[2 * p2::ROUNDS_HALF_FULL];  // rc - p of this cell, per full round
    unsigned cell;
    bool active;

    __device__ __forceinline__ void init(const typename C::Consts& k, unsigned cell_) {
        cell = cell_;
        active = cell < (unsigned)W;
        const unsigned j = cell & 3u;
        // rows of the block, one nibble per coefficient (see m_ext_redc)
        const uint32_t pk = C::M4_KIND == 0 ? (j == 0 ? 0x3175u : j == 1 ? 0x1164u : j == 2 ? 0x7531u : 0x6411u)
                                            : (j == 0 ? 0x1132u : j == 1 ? 0x1321u : j == 2 ? 0x3211u : 0x2113u);
        m0 = pk & 15u;
        m1 = (pk >> 4) & 15u;
        m2 = (pk >> 8) & 15u;
        m3 = pk >> 12;
        diag = active ? k.diag[cell] : 0u;
#pragma unroll
        for (int r = 0; r < 2 * p2::ROUNDS_HALF_FULL; r++) rc[r] = (active ? k.rc_ext[r * W + cell] : 0u) - bb::P;
    }
    __device__ __forceinline__ uint32_t ext(uint32_t x) const {
        uint32_t X = bb::ucanon(x + dpp_mov<DPP_ROR4>(x));
        X = bb::ucanon(X + dpp_mov<DPP_ROR8>(X));
        X = row_pair_sum(X);
        const uint32_t z = bb::ucanon(x + X);
        uint64_t w = (uint64_t)dpp_mov<DPP_Q0>(z) * m0;
        w += (uint64_t)dpp_mov<DPP_Q1>(z) * m1;
        w += (uint64_t)dpp_mov<DPP_Q2>(z) * m2;
        w += (uint64_t)dpp_mov<DPP_Q3>(z) * m3;
        // w < 16 p: REDC gives w / 2^32 (< p + 8), the product by 2^64 / 2^32 restores w mod p
        const uint32_t u = bb::ucanon(bb::umul_const(bb::uredc64(w), bb::R2, R2_NQ));
        return active ? u : 0u;
    }
    __device__ __forceinline__ uint32_t internal(uint32_t x, uint32_t rc_mp) const {
        const uint32_t y = bb::sbox7_add(x, rc_mp);
        x = cell == 0 ? y : x;
        uint32_t s = bb::ucanon(x + dpp_mov<DPP_ROR1>(x));
        s = bb::ucanon(s + dpp_mov<DPP_ROR2>(s));
        s = bb::ucanon(s + dpp_mov<DPP_ROR4>(s));
        s = bb::ucanon(s + dpp_mov<DPP_ROR8>(s));
        s = row_pair_sum(s);
        const uint32_t r = bb::add(bb::mul(x, diag), s);
        return active ? r : 0u;
    }
    // x: this lane's cell (canonical, zero beyond the width); every lane of the wave must be here
    __device__ __forceinline__ uint32_t permute(uint32_t x, const typename C::Consts& k) const {
        x = ext(x);
#pragma unroll
        for (int r = 0; r < p2::ROUNDS_HALF_FULL; r++) x = ext(bb::sbox7_add(x, rc[r]));
        // unrolled: the round constants are wave-uniform scalar loads, which a rolled loop would wait for one by one
#pragma unroll
        for (int r = 0; r < C::ROUNDS_PARTIAL; r++) x = internal(x, k.rc_int_mp[r]);
#pragma unroll
        for (int r = p2::ROUNDS_HALF_FULL; r < 2 * p2::ROUNDS_HALF_FULL; r++) x = ext(bb::sbox7_add(x, rc[r]));
        return x;
    }
};

// `levels` (<= CELLS_MAX_LEVELS) levels of a tree in one launch: workgroup b owns the 2^levels nodes
// [b 2^levels, (b + 1) 2^levels) of the level with 2 first_out nodes and everything above them; a level lives
// in LDS (ping-pong) between steps and every parent also goes to its heap slot.  blockDim = 32 lanes per
// parent of the first step (at most 512: a workgroup then has at most two waves per SIMD).
constexpr unsigned CELLS_MAX_LEVELS = 5;
template <class C>
__global__ __launch_bounds__(512) void hash_fold_cells_kernel(uint32_t* __restrict__ nodes, unsigned first_out, unsigned levels,
                                                               const typename C::Consts* __restrict__ kc) {
    __shared__ uint32_t buf[2][(1u << CELLS_MAX_LEVELS) * p2::OUT];
    const typename C::Consts& k = *kc;
    const unsigned tid = threadIdx.x, b = blockIdx.x;
    const unsigned nhw = blockDim.x >> 5, h = tid >> 5, cell = tid & 31u;
    const unsigned in_cnt = 1u << levels;
    for (unsigned i = tid; i < in_cnt * p2::OUT; i += blockDim.x)
        buf[0][i] = nodes[((size_t)2 * first_out + (size_t)b * in_cnt) * p2::OUT + i];
    CellPerm<C> cp;
    cp.init(k, cell);
    __syncthreads();
    for (unsigned l = 0; l < levels; l++) {
        const unsigned cnt = in_cnt >> (l + 1);
        const size_t out_base = (size_t)(first_out >> l) + (size_t)b * cnt;
        const uint32_t* src = buf[l & 1];
        uint32_t* dst = buf[(l + 1) & 1];
        for (unsigned t0 = 0; t0 < cnt; t0 += nhw) {  // uniform trip count: idle half-waves redo parent 0 and drop it
            const unsigned t = t0 + h;
            const bool valid = t < cnt;
            const unsigned tt = valid ? t : 0u;
            uint32_t x = cell < 2 * p2::OUT ? src[2 * tt * p2::OUT + cell] : 0u;
            x = cp.permute(x, k);
            if (valid && cell < (unsigned)p2::OUT) {
                dst[t * p2::OUT + cell] = x;
                nodes[(out_base + t) * p2::OUT + cell] = x;
            }
        }
        __syncthreads();
    }
}

// Proof of work on the transcript: candidate w = base + lane.  digest = hash([w]) (one block of the
// sponge), the generator absorbs it (cells[0..8) += digest, permute) and the next four outputs,
// decoded and xor-ed, must be zero in their low `bits` bits -- exactly what Transcript::commit and
// p2::Rng::random_bits do with pool_used = 0.  The smallest hit of the launch wins (atomicMin).
template <class C>
__global__ __launch_bounds__(HASH_BLOCK) void pow_grind_kernel(uint32_t* __restrict__ best, const uint32_t* __restrict__ cells,
                                                               uint32_t base, uint32_t count, uint32_t mask,
                                                               const typename C::Consts* __restrict__ kc) {
    const uint32_t gid = blockIdx.x * HASH_BLOCK + threadIdx.x;
    if (gid >= count) return;
    const typename C::Consts& k = *kc;
    const uint32_t w = base + gid;
    uint32_t s[C::CELLS];
    s[0] = w;
#pragma unroll
    for (int i = 1; i < C::CELLS; i++) s[i] = 0;
    C::permute(s, k);
    uint32_t c[C::CELLS];
#pragma unroll
    for (int i = 0; i < C::CELLS; i++) c[i] = cells[i];
#pragma unroll
    for (int i = 0; i < p2::OUT; i++) c[i] = bb::add(c[i], s[i]);
    C::permute(c, k);
    const uint32_t v = bb::decode(c[0]) ^ bb::decode(c[1]) ^ bb::decode(c[2]) ^ bb::decode(c[3]);
    if ((v & mask) == 0) atomicMin(best, w);
}

// Plonky3 DuplexChallenger::grind (p3-challenger duplex_challenger.rs + grinding_challenger.rs, RECALLED): candidate
// w = base + lane is observed on a copy of the challenger -- it joins the buffered inputs, which overwrite the first
// cells of the sponge state, and whether the push fills the rate or the following sample forces it, exactly one
// permutation runs before sample_bits pops the last rate cell of the new state.  Smallest hit wins.
template <class C>
__global__ __launch_bounds__(HASH_BLOCK) void duplex_grind_kernel(uint32_t* __restrict__ best, const uint32_t* __restrict__ state,
                                                                  unsigned n_input, uint32_t base, uint32_t count, uint32_t mask,
                                                                  const typename C::Consts* __restrict__ kc) {
    const uint32_t gid = blockIdx.x * HASH_BLOCK + threadIdx.x;
    if (gid >= count) return;
    const uint32_t w = base + gid;
    uint32_t s[C::CELLS];
#pragma unroll
    for (int i = 0; i < C::CELLS; i++) s[i] = (unsigned)i == n_input ? bb::mul(w, bb::R2) : state[i];  // state already holds the buffered inputs
    C::permute(s, *kc);
    if ((bb::decode(s[C::RATE - 1]) & mask) == 0) atomicMin(best, w);
}

// run F<Core> for the context's Poseidon2 instance
#define RK_P2_DISPATCH(ctx, CALL)                         \
    switch ((ctx)->h_p2.kind) {                           \
        case 0: { using C = p2::K0; CALL; } break;        \
        case 1: { using C = p2::K1; CALL; } break;        \
        case 2: { using C = p2::K2; CALL; } break;        \
        default: { using C = p2::K3; CALL; } break;       \
    }

}  // namespace

namespace rk {

int hash_rows(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_matrix, size_t rows, size_t cols) {
    if (rows == 0) return RK_ERR_INVALID;
    size_t blocks = (rows + HASH_BLOCK - 1) / HASH_BLOCK;
    if (blocks > 0x7fffffffu) return RK_ERR_INVALID;
    KTimer kt(ctx, RK_KCLASS_HASH_ROWS, (double)rows * cols * 4 + (double)rows * 32);
    RK_P2_DISPATCH(ctx, hipLaunchKernelGGL(hash_rows_kernel<C>, dim3((unsigned)blocks), dim3(HASH_BLOCK), 0, ctx->stream, d_out,
                                           d_matrix, rows, cols, (const typename C::Consts*)ctx->d_p2,
                                           ctx->h_p2.pad_free ? 1 : 0));
    return post_launch(ctx, "hash_rows_kernel");
}

int hash_fold(rk_ctx* ctx, uint32_t* d_nodes, size_t output_size) {
    if (output_size == 0) return RK_ERR_INVALID;
    size_t blocks = (output_size + HASH_BLOCK - 1) / HASH_BLOCK;
    if (blocks > 0x7fffffffu) return RK_ERR_INVALID;
    KTimer kt(ctx, RK_KCLASS_HASH_FOLD, (double)output_size * 96);
    RK_P2_DISPATCH(ctx, hipLaunchKernelGGL(hash_fold_kernel<C>, dim3((unsigned)blocks), dim3(HASH_BLOCK), 0, ctx->stream, d_nodes,
                                           output_size, (const typename C::Consts*)ctx->d_p2));
    return post_launch(ctx, "hash_fold_kernel");
}

int merkle_build(rk_ctx* ctx, uint32_t* d_nodes, const uint32_t* d_matrix, size_t rows, size_t cols) {
    if (!is_pow2(rows)) return RK_ERR_INVALID;
    RK_TRY(hash_rows(ctx, d_nodes + rows * p2::OUT, d_matrix, rows, cols));
    size_t layer = rows / 2;
    for (; layer > CELLS_MAX_OUT; layer /= 2) RK_TRY(hash_fold(ctx, d_nodes, layer));
    if (layer >= 1) RK_TRY(hash_fold_top(ctx, d_nodes, layer));
    return RK_OK;
}

int pow_grind(rk_ctx* ctx, const uint32_t* h_cells, unsigned bits, uint32_t* nonce) {
    if (bits == 0 || bits > 24 || !h_cells || !nonce) return RK_ERR_INVALID;
    const unsigned width = (unsigned)ctx->h_p2.cells();
    void* d = nullptr;
    RK_TRY(scratch(ctx, (p2::MAX_CELLS + 4) * 4, &d));
    uint32_t* d_cells = (uint32_t*)d;
    uint32_t* d_best = d_cells + p2::MAX_CELLS;
    uint32_t host[p2::MAX_CELLS + 1] = {0};
    std::memcpy(host, h_cells, width * 4);
    host[p2::MAX_CELLS] = 0xffffffffu;
    RK_HIP_TRY(ctx, hipMemcpyAsync(d, host, sizeof host, hipMemcpyHostToDevice, ctx->stream));
    RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // `host` is a stack buffer
    // 2^(bits + 2) candidates per launch: four hits expected, so one launch almost always; candidates run
    // up to p (a nonce is a field element), far beyond what 24 bits need
    const uint64_t batch = (uint64_t)1 << (bits + 2 < 16 ? 16 : bits + 2);
    const uint32_t mask = (uint32_t)(((uint64_t)1 << bits) - 1);
    for (uint64_t base = 0; base < bb::P; base += batch) {
        const uint32_t count = (uint32_t)std::min<uint64_t>(batch, bb::P - base);
        const unsigned blocks = (count + HASH_BLOCK - 1) / HASH_BLOCK;
        RK_P2_DISPATCH(ctx, hipLaunchKernelGGL(pow_grind_kernel<C>, dim3(blocks), dim3(HASH_BLOCK), 0, ctx->stream, d_best, d_cells,
                                               (uint32_t)base, count, mask, (const typename C::Consts*)ctx->d_p2));
        RK_TRY(post_launch(ctx, "pow_grind_kernel"));
        uint32_t best = 0;
        RK_HIP_TRY(ctx, hipMemcpyAsync(&best, d_best, 4, hipMemcpyDeviceToHost, ctx->stream));
        RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (best != 0xffffffffu) {
            *nonce = best;
            return RK_OK;
        }
    }
    return RK_ERR_INTERNAL;
}

int duplex_grind(rk_ctx* ctx, const uint32_t* h_state, const uint32_t* h_input, unsigned n_input, unsigned bits, uint32_t* witness) {
    const unsigned width = (unsigned)ctx->h_p2.cells(), rate = width - p2::OUT;
    if (bits == 0 || bits > 24 || !h_state || !witness || n_input >= rate || (n_input && !h_input)) return RK_ERR_INVALID;
    uint32_t host[p2::MAX_CELLS + 1] = {0};
    std::memcpy(host, h_state, width * 4);
    if (n_input) std::memcpy(host, h_input, n_input * 4);  // duplexing overwrites the first cells with the buffered inputs
    for (unsigned i = 0; i < width; i++)
        if (host[i] >= bb::P) return RK_ERR_INVALID;
    host[p2::MAX_CELLS] = 0xffffffffu;
    void* d = nullptr;
    RK_TRY(scratch(ctx, sizeof host, &d));
    uint32_t* d_cells = (uint32_t*)d;
    uint32_t* d_best = d_cells + p2::MAX_CELLS;
    RK_TRY(upload(ctx, d, host, sizeof host));
    const uint64_t batch = (uint64_t)1 << (bits + 2 < 16 ? 16 : bits + 2);
    const uint32_t mask = (uint32_t)(((uint64_t)1 << bits) - 1);
    for (uint64_t base = 0; base < bb::P; base += batch) {
        const uint32_t count = (uint32_t)std::min<uint64_t>(batch, bb::P - base);
        const unsigned blocks = (count + HASH_BLOCK - 1) / HASH_BLOCK;
        RK_P2_DISPATCH(ctx, hipLaunchKernelGGL(duplex_grind_kernel<C>, dim3(blocks), dim3(HASH_BLOCK), 0, ctx->stream, d_best, d_cells,
                                               n_input, (uint32_t)base, count, mask, (const typename C::Consts*)ctx->d_p2));
        RK_TRY(post_launch(ctx, "duplex_grind_kernel"));
        uint32_t best = 0;
        RK_HIP_TRY(ctx, hipMemcpyAsync(&best, d_best, 4, hipMemcpyDeviceToHost, ctx->stream));
        RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (best != 0xffffffffu) {
            *witness = best;
            return RK_OK;
        }
    }
    return RK_ERR_INTERNAL;
}

// every level from the one with `top_output_size` parents up to the root, cell-parallel, a few levels per launch
int hash_fold_top(rk_ctx* ctx, uint32_t* d_nodes, size_t top_output_size) {
    if (!is_pow2(top_output_size) || top_output_size > CELLS_MAX_OUT) return RK_ERR_INVALID;
    unsigned remaining = 1;  // levels left, the root's included
    for (size_t n = top_output_size; n > 1; n >>= 1) remaining++;
    unsigned first_out = (unsigned)top_output_size;
    while (remaining) {
        // the last launch is one workgroup: four levels keep it at one wave per SIMD (eight parents first); the
        // launches below it take up to five levels each, split evenly
        unsigned m = std::min(remaining, 4u);
        if (remaining > 4) {
            const unsigned rest = remaining - 4, k = (rest + 4) / 5;
            m = (rest + k - 1) / k;
        }
        const unsigned blocks = (2 * first_out) >> m;
        const unsigned threads = std::max(64u, 32u << (m - 1));  // m <= CELLS_MAX_LEVELS: at most 512
        KTimer kt(ctx, RK_KCLASS_HASH_FOLD, (double)blocks * ((1u << m) - 1) * 96);
        RK_P2_DISPATCH(ctx, hipLaunchKernelGGL(hash_fold_cells_kernel<C>, dim3(blocks), dim3(threads), 0, ctx->stream, d_nodes,
                                               first_out, m, (const typename C::Consts*)ctx->d_p2));
        RK_TRY(post_launch(ctx, "hash_fold_cells_kernel"));
        first_out >>= m;
        remaining -= m;
    }
    return RK_OK;
}

}  // namespace rk

extern "C" {

int rk_hash_rows(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_matrix, size_t rows, size_t cols) {
    RK_GUARD_BEGIN
    if (!ctx || !d_out || (!d_matrix && cols)) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::hash_rows(ctx, d_out, d_matrix, rows, cols);
    RK_GUARD_END
}
int rk_hash_fold(rk_ctx* ctx, uint32_t* d_nodes, size_t input_size, size_t output_size) {
    RK_GUARD_BEGIN
    if (!ctx || !d_nodes || input_size != 2 * output_size) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::hash_fold(ctx, d_nodes, output_size);
    RK_GUARD_END
}
int rk_pow_grind(rk_ctx* ctx, const uint32_t* sponge_cells, uint32_t bits, uint32_t* nonce) {
    RK_GUARD_BEGIN
    if (!ctx || !sponge_cells || !nonce) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::pow_grind(ctx, sponge_cells, bits, nonce);
    RK_GUARD_END
}
int rk_duplex_grind(rk_ctx* ctx, const uint32_t* sponge_state, const uint32_t* input_buffer, uint32_t n_input, uint32_t bits,
                    uint32_t* witness) {
    RK_GUARD_BEGIN
    if (!ctx || !sponge_state || !witness) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::duplex_grind(ctx, sponge_state, input_buffer, n_input, bits, witness);
    RK_GUARD_END
}
int rk_merkle_build(rk_ctx* ctx, uint32_t* d_nodes, const uint32_t* d_matrix, size_t rows, size_t cols) {
    RK_GUARD_BEGIN
    if (!ctx || !d_nodes || !d_matrix) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::merkle_build(ctx, d_nodes, d_matrix, rows, cols);
    RK_GUARD_END
}

}  // extern "C"
