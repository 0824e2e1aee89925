// Mixed-matrix commitment: Plonky3's MerkleTreeMmcs (p3-merkle-tree `MerkleTree::new`,
// `first_digest_layer`, `compress_and_inject`; p3-symmetric PaddingFreeSponge / TruncatedPermutation --
// RECALLED, the crates are outside the reference tree; SP1 reaches them from
// provers/sp1/driver/src/lib.rs:48-57) on this library's Poseidon2 instances.
//
// Leaves: lane = row; the concatenated row of several matrices is absorbed RATE columns at a time, the
// (matrix, column) of every concatenated column comes from a wave-uniform table, so a block's RATE loads
// are issued together and the state stays in registers.  A row-major matrix is read RATE consecutive
// words per lane (32- or 64-byte pieces of one row); the kernel is bound by the permutation's integer
// work, not by that access pattern.  Levels: the existing pair compression, then -- where shorter
// matrices join -- their row hashes and one more compression per node.
#include <algorithm>
#include <cstring>
#include <memory>

#include "internal.hpp"

namespace {

constexpr int HASH_BLOCK = 256;
constexpr uint32_t MAX_MATS = 512;

struct MatDesc {
    const uint32_t* base;
    uint32_t width;
    uint32_t row_major;
};
// one entry per concatenated column: which matrix, which of its columns
struct ColRef {
    uint32_t mat, col;
};

#if defined(__HIP_DEVICE_COMPILE__)
#define RK_CONST_AS __attribute__((address_space(4)))
#else
#define RK_CONST_AS
#endif

// nat_bits != 0: lane t is the NATURAL index of a layout-2 matrix (column-major, committed row r stored at index
// bitrev(r)): its loads are then coalesced and the digest goes to row bitrev(t); otherwise lane t is the committed row
template <class C>
__global__ __launch_bounds__(HASH_BLOCK) void hash_rows_multi_kernel(uint32_t* __restrict__ out, uint64_t mats_addr, uint64_t cols_addr,
                                                                     uint32_t n_cols, size_t rows,
                                                                     const typename C::Consts* __restrict__ kc, int pad_free, unsigned nat_bits,
                                                                     unsigned bits) {
    const size_t t = (size_t)blockIdx.x * HASH_BLOCK + threadIdx.x;
    if (t >= rows) return;
    const size_t other = bb::bitrev((uint32_t)t, bits);           // bits = log2 rows
    const size_t row = nat_bits ? other : t, nat = nat_bits ? t : other;
    const RK_CONST_AS MatDesc* mats = (const RK_CONST_AS MatDesc*)mats_addr;
    const RK_CONST_AS ColRef* cols = (const RK_CONST_AS ColRef*)cols_addr;
    const typename C::Consts& k = *kc;
    uint32_t s[C::CELLS];
#pragma unroll
    for (int i = 0; i < C::CELLS; i++) s[i] = 0;
    auto load = [&](uint32_t g) -> uint32_t {
        const ColRef cr = {cols[g].mat, cols[g].col};
        const MatDesc m = {mats[cr.mat].base, mats[cr.mat].width, mats[cr.mat].row_major};
        return m.row_major == 1 ? m.base[row * m.width + cr.col] : m.base[(size_t)cr.col * rows + (m.row_major == 2 ? nat : row)];
    };
    const uint32_t full = n_cols / C::RATE;
    for (uint32_t b = 0; b < full; b++) {
#pragma unroll
        for (int i = 0; i < C::RATE; i++) s[i] = load(b * C::RATE + i);
        C::permute(s, k);
    }
    const uint32_t rem = n_cols - full * C::RATE;
    if (rem != 0 || (n_cols == 0 && !pad_free)) {
#pragma unroll
        for (int i = 0; i < C::RATE; i++) {
            if ((uint32_t)i < rem) s[i] = load(full * C::RATE + i);
            else if (!pad_free) s[i] = 0u;
        }
        C::permute(s, k);
    }
    uint4* o = reinterpret_cast<uint4*>(out + row * p2::OUT);
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// nodes[base + i] = compress(nodes[base + i], extra[i]), i < n
template <class C>
__global__ __launch_bounds__(HASH_BLOCK) void compress_inject_kernel(uint32_t* __restrict__ nodes, const uint32_t* __restrict__ extra,
                                                                     size_t n, const typename C::Consts* __restrict__ kc) {
    const size_t i = (size_t)blockIdx.x * HASH_BLOCK + threadIdx.x;
    if (i >= n) return;
    const typename C::Consts& k = *kc;
    const uint4* a = reinterpret_cast<const uint4*>(nodes + (n + i) * p2::OUT);
    const uint4* b = reinterpret_cast<const uint4*>(extra + i * p2::OUT);
    const uint4 a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
    uint32_t s[C::CELLS];
    s[0] = a0.x; s[1] = a0.y; s[2] = a0.z; s[3] = a0.w; s[4] = a1.x; s[5] = a1.y; s[6] = a1.z; s[7] = a1.w;
    s[8] = b0.x; s[9] = b0.y; s[10] = b0.z; s[11] = b0.w; s[12] = b1.x; s[13] = b1.y; s[14] = b1.z; s[15] = b1.w;
#pragma unroll
    for (int j = 16; j < C::CELLS; j++) s[j] = 0;
    C::permute(s, k);
    uint4* o = reinterpret_cast<uint4*>(nodes + (n + i) * p2::OUT);
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

#define RK_P2_DISPATCH(ctx, CALL)                         \
    switch ((ctx)->h_p2.kind) {                           \
        case 0: { using C = p2::K0; CALL; } break;        \
        case 1: { using C = p2::K1; CALL; } break;        \
        case 2: { using C = p2::K2; CALL; } break;        \
        default: { using C = p2::K3; CALL; } break;       \
    }

int check_mats(const rk_matrix* mats, uint32_t n, uint32_t* max_h) {
    if (!mats || n == 0 || n > MAX_MATS) return RK_ERR_INVALID;
    uint32_t H = 0;
    for (uint32_t m = 0; m < n; m++) {
        if (!mats[m].d_values || mats[m].height == 0 || (mats[m].height & (mats[m].height - 1)) || mats[m].width == 0 ||
            mats[m].row_major > 2 || mats[m].height > (1u << ntt::LAMBDA))
            return RK_ERR_INVALID;
        H = std::max(H, mats[m].height);
    }
    *max_h = H;
    return RK_OK;
}

// digests of the concatenated rows of the matrices of height `h` (given order) into d_out (h digests)
int hash_level(rk_ctx* ctx, const rk_matrix* mats, uint32_t n, uint32_t h, uint32_t* d_out) {
    std::vector<MatDesc> descs;
    std::vector<ColRef> cols;
    bool all_nat = true;   // every matrix of this level is layout 2: lanes walk the natural index (coalesced loads)
    for (uint32_t m = 0; m < n; m++) {
        if (mats[m].height != h) continue;
        all_nat = all_nat && mats[m].row_major == 2;
        for (uint32_t c = 0; c < mats[m].width; c++) cols.push_back(ColRef{(uint32_t)descs.size(), c});
        descs.push_back(MatDesc{mats[m].d_values, mats[m].width, mats[m].row_major});
    }
    const size_t desc_bytes = (descs.size() * sizeof(MatDesc) + 15) & ~(size_t)15, col_bytes = cols.size() * sizeof(ColRef);
    std::vector<unsigned char> pack(desc_bytes + col_bytes + 16, 0);
    std::memcpy(pack.data(), descs.data(), descs.size() * sizeof(MatDesc));
    std::memcpy(pack.data() + desc_bytes, cols.data(), col_bytes);
    void* d = nullptr;
    RK_TRY(rk::scratch(ctx, pack.size(), &d));
    RK_TRY(rk::upload(ctx, d, pack.data(), pack.size()));  // through the page-locked ring: no wait (21 FRI layers of a proof commit one after the other)
    const unsigned blocks = (unsigned)(((size_t)h + HASH_BLOCK - 1) / HASH_BLOCK);
    rk::KTimer kt(ctx, RK_KCLASS_HASH_ROWS, (double)h * cols.size() * 4 + (double)h * 32);
    RK_P2_DISPATCH(ctx, hipLaunchKernelGGL(hash_rows_multi_kernel<C>, dim3(blocks), dim3(HASH_BLOCK), 0, ctx->stream, d_out,
                                           (uint64_t)(uintptr_t)d, (uint64_t)(uintptr_t)((unsigned char*)d + desc_bytes),
                                           (uint32_t)cols.size(), (size_t)h, (const typename C::Consts*)ctx->d_p2,
                                           ctx->h_p2.pad_free ? 1 : 0, all_nat ? 1u : 0u, log2u(h)));
    return rk::post_launch(ctx, "hash_rows_multi_kernel");
}

}  // namespace

extern "C" {

int rk_mmcs_commit(rk_ctx* ctx, const rk_matrix* mats, uint32_t n_mats, uint32_t* d_nodes, uint32_t h_root[8]) {
    RK_GUARD_BEGIN
    if (!ctx || !d_nodes) return RK_ERR_INVALID;
    uint32_t H = 0;
    RK_TRY(check_mats(mats, n_mats, &H));
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    RK_TRY(hash_level(ctx, mats, n_mats, H, d_nodes + (size_t)H * p2::OUT));
    void* d_extra = nullptr;
    // above the shortest matrix nothing joins any more: those levels (at most HASH_FOLD_TOP_MAX parents) go through
    // the cell-parallel launches of hash_fold_top, a few levels each, instead of one latency-bound launch per level
    uint32_t shortest = H;
    for (uint32_t m = 0; m < n_mats; m++) shortest = std::min(shortest, mats[m].height);
    const uint32_t top = (uint32_t)std::min<size_t>(shortest / 2, rk::HASH_FOLD_TOP_MAX);
    for (uint32_t size = H / 2; size >= 1 && size > top; size /= 2) {
        RK_TRY(rk::hash_fold(ctx, d_nodes, size));  // nodes[size + i] = compress(children)
        bool inject = false;
        for (uint32_t m = 0; m < n_mats; m++) inject |= mats[m].height == size;
        if (inject) {
            if (!d_extra) RK_TRY(rk::dev_alloc(ctx, (size_t)H / 2 * p2::OUT * 4, &d_extra));
            int st = hash_level(ctx, mats, n_mats, size, (uint32_t*)d_extra);
            if (st == RK_OK) {
                const unsigned blocks = (unsigned)(((size_t)size + HASH_BLOCK - 1) / HASH_BLOCK);
                rk::KTimer kt(ctx, RK_KCLASS_HASH_FOLD, (double)size * 96);
                RK_P2_DISPATCH(ctx, hipLaunchKernelGGL(compress_inject_kernel<C>, dim3(blocks), dim3(HASH_BLOCK), 0, ctx->stream, d_nodes,
                                                       (const uint32_t*)d_extra, (size_t)size, (const typename C::Consts*)ctx->d_p2));
                st = rk::post_launch(ctx, "compress_inject_kernel");
            }
            if (st != RK_OK) {
                (void)rk::dev_free(ctx, d_extra);
                return st;
            }
        }
    }
    if (d_extra) RK_TRY(rk::dev_free(ctx, d_extra));
    if (top >= 1) RK_TRY(rk::hash_fold_top(ctx, d_nodes, top));
    if (h_root) {
        RK_HIP_TRY(ctx, hipMemcpyAsync(h_root, d_nodes + p2::OUT, p2::OUT * 4, hipMemcpyDeviceToHost, ctx->stream));
        RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RK_OK;
    RK_GUARD_END
}

int rk_mmcs_open(rk_ctx* ctx, const rk_matrix* mats, uint32_t n_mats, const uint32_t* d_nodes, uint32_t index, uint32_t* h_rows,
                 uint32_t* h_path) {
    RK_GUARD_BEGIN
    if (!ctx || !d_nodes || !h_rows || !h_path) return RK_ERR_INVALID;
    uint32_t H = 0;
    RK_TRY(check_mats(mats, n_mats, &H));
    if (index >= H) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t pos = 0;
    for (uint32_t m = 0; m < n_mats; m++) {
        const rk_matrix& M = mats[m];
        const uint32_t r = index / (H / M.height);
        if (M.row_major == 1) {
            RK_HIP_TRY(ctx, hipMemcpyAsync(h_rows + pos, M.d_values + (size_t)r * M.width, (size_t)M.width * 4, hipMemcpyDeviceToHost,
                                           ctx->stream));
        } else {
            const uint32_t at = M.row_major == 2 ? bb::bitrev(r, log2u(M.height)) : r;
            RK_HIP_TRY(ctx, hipMemcpy2DAsync(h_rows + pos, 4, M.d_values + at, (size_t)M.height * 4, 4, M.width, hipMemcpyDeviceToHost,
                                             ctx->stream));
        }
        pos += M.width;
    }
    size_t idx = (size_t)H + index, lvl = 0;
    while (idx > 1) {
        RK_HIP_TRY(ctx, hipMemcpyAsync(h_path + lvl * p2::OUT, d_nodes + (idx ^ 1) * p2::OUT, p2::OUT * 4, hipMemcpyDeviceToHost,
                                       ctx->stream));
        idx >>= 1;
        lvl++;
    }
    RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RK_OK;
    RK_GUARD_END
}

int rk_mmcs_verify(const rk_params* params, const uint32_t* heights, const uint32_t* widths, uint32_t n_mats, uint32_t index,
                   const uint32_t* rows, const uint32_t* path, const uint32_t root[8]) {
    RK_GUARD_BEGIN
    if (!heights || !widths || !rows || !path || !root || n_mats == 0 || n_mats > MAX_MATS) return RK_ERR_INVALID;
    rk_params def;
    rk::params_preset(&def, RK_PRESET_RISC0);
    rk::Sys sys;
    auto k = std::make_unique<p2::Any>();
    RK_TRY(rk::resolve_params(params ? params : &def, &sys, k.get()));
    uint32_t H = 0;
    for (uint32_t m = 0; m < n_mats; m++) {
        if (heights[m] == 0 || (heights[m] & (heights[m] - 1)) || widths[m] == 0) return RK_ERR_INVALID;
        H = std::max(H, heights[m]);
    }
    if (index >= H) return RK_ERR_INVALID;
    // the opened rows of the matrices of height h, concatenated in commit order
    auto level_hash = [&](uint32_t h, uint32_t* digest) -> bool {
        std::vector<uint32_t> cat;
        size_t pos = 0;
        for (uint32_t m = 0; m < n_mats; m++) {
            if (heights[m] == h) cat.insert(cat.end(), rows + pos, rows + pos + widths[m]);
            pos += widths[m];
        }
        if (cat.empty()) return false;
        for (uint32_t v : cat)
            if (v >= bb::P) return false;
        k->hash_elems(cat.data(), cat.size(), digest);
        return true;
    };
    uint32_t cur[8];
    if (!level_hash(H, cur)) return 1;
    uint32_t idx = index, lvl = 0;
    for (uint32_t size = H / 2; size >= 1; size /= 2, lvl++) {
        const uint32_t* sib = path + (size_t)lvl * 8;
        uint32_t nxt[8], extra[8];
        if (idx & 1) k->hash_pair(sib, cur, nxt);
        else k->hash_pair(cur, sib, nxt);
        idx >>= 1;
        bool joins = false;
        for (uint32_t m = 0; m < n_mats; m++) joins |= heights[m] == size;
        if (joins) {
            if (!level_hash(size, extra)) return 1;
            k->hash_pair(nxt, extra, cur);
        } else {
            std::memcpy(cur, nxt, 32);
        }
    }
    return std::memcmp(cur, root, 32) == 0 ? 0 : 1;
    RK_GUARD_END
}

}  // extern "C"
