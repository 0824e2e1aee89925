// Poseidon2 row hashing and Merkle folds (Hal::hash_rows / Hal::hash_fold and
// MerkleTreeProver::new of risc0-zkp 1.0.1).  Integer-ALU-bound: one lane owns one
// sponge (24 state words in VGPRs), consecutive lanes own consecutive rows so every
// column load is a coalesced 256-byte wave access of the column-major matrix.
#include <algorithm>
#include <cstring>

#include "internal.hpp"

namespace {

constexpr int HASH_BLOCK = 256;

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

// The last levels of a tree (<= 1024 parents) in one launch: the level lives in LDS between
// steps, every parent is also written to its heap slot in HBM.  Replaces ~11 latency-bound
// launches per tree.
constexpr int TAIL_MAX = 1024;
template <class C>
__global__ __launch_bounds__(TAIL_MAX) void hash_fold_tail_kernel(uint32_t* __restrict__ nodes, unsigned top_out,
                                                                  const typename C::Consts* __restrict__ kc) {
    __shared__ uint32_t level[2 * TAIL_MAX * p2::OUT];
    const unsigned tid = threadIdx.x;
    const typename C::Consts& k = *kc;
    for (unsigned i = tid; i < 2 * top_out * p2::OUT; i += blockDim.x) level[i] = nodes[(size_t)2 * top_out * p2::OUT + i];
    __syncthreads();
    for (unsigned out = top_out; out >= 1; out >>= 1) {
        uint32_t s[C::CELLS];
        const bool active = tid < out;
        if (active) {
#pragma unroll
            for (int i = 0; i < 2 * p2::OUT; i++) s[i] = level[2 * tid * p2::OUT + i];
#pragma unroll
            for (int i = 2 * p2::OUT; i < C::CELLS; i++) s[i] = 0;
            C::permute(s, k);
        }
        __syncthreads();  // every child has been read before the level is overwritten
        if (active) {
#pragma unroll
            for (int i = 0; i < p2::OUT; i++) {
                level[tid * p2::OUT + i] = s[i];
                nodes[(size_t)(out + tid) * p2::OUT + i] = s[i];
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
    for (; layer > TAIL_MAX; layer /= 2) RK_TRY(hash_fold(ctx, d_nodes, layer));
    if (layer >= 1) RK_TRY(hash_fold_tail(ctx, d_nodes, layer));
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

int hash_fold_tail(rk_ctx* ctx, uint32_t* d_nodes, size_t top_output_size) {
    if (!is_pow2(top_output_size) || top_output_size > TAIL_MAX) return RK_ERR_INVALID;
    unsigned threads = top_output_size < 64 ? 64u : (unsigned)top_output_size;
    KTimer kt(ctx, RK_KCLASS_HASH_FOLD, (double)(2 * top_output_size - 1) * 96);
    RK_P2_DISPATCH(ctx, hipLaunchKernelGGL(hash_fold_tail_kernel<C>, dim3(1), dim3(threads), 0, ctx->stream, d_nodes,
                                           (unsigned)top_output_size, (const typename C::Consts*)ctx->d_p2));
    return post_launch(ctx, "hash_fold_tail_kernel");
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
int rk_merkle_build(rk_ctx* ctx, uint32_t* d_nodes, const uint32_t* d_matrix, size_t rows, size_t cols) {
    RK_GUARD_BEGIN
    if (!ctx || !d_nodes || !d_matrix) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::merkle_build(ctx, d_nodes, d_matrix, rows, cols);
    RK_GUARD_END
}

}  // extern "C"
