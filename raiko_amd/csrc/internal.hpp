// Internal declarations shared by the translation units of libraiko_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/raiko_hip.h"
#include "taps.hpp"
#include "bb.hpp"
#include "ntt_core.hpp"
#include "poseidon2_core.hpp"
#include "params.hpp"

struct rk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string last_error;

    // twiddle / power tables (one device allocation)
    uint32_t* d_tables = nullptr;
    ntt::Tables tb{};

    // the parameter set (rk_set_params): field / protocol values and the Poseidon2 instance -- host copy
    // for the transcript, device copy (the active Core's Consts) for the kernels
    rk::Sys sys{};
    p2::Any h_p2{};
    void* d_p2 = nullptr;

    // caching allocator (exact-size free lists) so steady-state proving never calls hipMalloc
    std::multimap<size_t, void*> free_list;
    std::unordered_map<void*, size_t> live;
    size_t pooled_bytes = 0;                         // bytes parked in free_list
    size_t pool_limit = (size_t)24 << 30;            // parked bytes above this: the cache is dropped (varying shapes); several contexts share one GPU

    // small staging area for per-call parameter uploads
    void* d_scratch = nullptr;
    size_t scratch_bytes = 0;
    // page-locked ring the small uploads are staged in (rk::upload): no wait per upload
    void* h_ring = nullptr;
    size_t h_ring_at = 0;
    bool h_ring_failed = false;

    rk_timing timing{};
    rk_p3_timing p3_timing{};
    std::vector<hipEvent_t> stage_events;            // pool behind the per-stage brackets of prove_segment

    // optional per-kernel-class timing (hipEvent pairs on the ctx stream, resolved lazily)
    bool ktime_on = false;
    struct KRec {
        hipEvent_t a, b;
        int cls;
        double bytes;
    };
    std::vector<KRec> krecs;
    size_t krec_used = 0;
    double k_ms[RK_KCLASS_COUNT] = {0};
    double k_bytes[RK_KCLASS_COUNT] = {0};
    uint64_t k_launches[RK_KCLASS_COUNT] = {0};
};

#define RK_HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);              \
            return RK_ERR_HIP;                                                                  \
        }                                                                                       \
    } while (0)
// extern "C" bodies never let a C++ exception cross the ABI ("never aborts")
#define RK_GUARD_BEGIN try {
#define RK_GUARD_END                                   \
    }                                                  \
    catch (const std::bad_alloc&) { return RK_ERR_NOMEM; } \
    catch (...) { return RK_ERR_INTERNAL; }
#define RK_TRY(expr)                \
    do {                            \
        int _s = (expr);            \
        if (_s != RK_OK) return _s; \
    } while (0)

static inline bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
static inline unsigned log2u(size_t n) {
    unsigned k = 0;
    while (((size_t)1 << k) < n) k++;
    return k;
}

namespace rk {

// per-kernel-class timing (context.hip): bracket a launch; no-ops unless enabled
struct KTimer {
    rk_ctx* ctx;
    int idx;
    KTimer(rk_ctx* c, int cls, double algorithmic_bytes);
    ~KTimer();
};

// optional roctx ranges around the proof stages (context.hip): libroctx64 is looked up at run time and only
// when RK_ROCTX is set, so the library has no link-time dependency on it
bool trace_push(const char* name);
void trace_pop();

// memory (context.hip)
int dev_alloc(rk_ctx* ctx, size_t bytes, void** out);
int dev_free(rk_ctx* ctx, void* p);
int scratch(rk_ctx* ctx, size_t bytes, void** out);  // valid until the next scratch() call
// host -> device on the ctx stream without waiting: the bytes are staged in the context's page-locked ring, so the
// caller's buffer is free on return (large uploads fall back to copy + wait)
int upload(rk_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int post_launch(rk_ctx* ctx, const char* what);

// NTT (kernels_ntt.hip)
int ntt_reverse(rk_ctx* ctx, uint32_t* d_io, size_t size, size_t count, bool fuse_zk_shift);
int ntt_reverse_from(rk_ctx* ctx, uint32_t* d_dst, const uint32_t* d_src, size_t size, size_t count, bool fuse_zk_shift);
int ntt_forward(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t in_size, size_t count,
                unsigned expand_bits);
int zk_shift(rk_ctx* ctx, uint32_t* d_io, size_t size, size_t count);
int bit_reverse(rk_ctx* ctx, uint32_t* d_io, size_t size, size_t count);

// hashing (kernels_hash.hip)
int hash_rows(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_matrix, size_t rows, size_t cols);
int hash_fold(rk_ctx* ctx, uint32_t* d_nodes, size_t output_size);
int merkle_build(rk_ctx* ctx, uint32_t* d_nodes, const uint32_t* d_matrix, size_t rows, size_t cols);
// smallest proof-of-work nonce for the transcript generator state h_cells (host, p2 width words)
int pow_grind(rk_ctx* ctx, const uint32_t* h_cells, unsigned bits, uint32_t* nonce);
int duplex_grind(rk_ctx* ctx, const uint32_t* h_state, const uint32_t* h_input, unsigned n_input, unsigned bits, uint32_t* witness);

// polynomial / elementwise (kernels_poly.hip)
int eltwise_add(rk_ctx* ctx, uint32_t* d_out, const uint32_t* a, const uint32_t* b, size_t n);
int eltwise_sum_ext(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t count, size_t to_add);
int eltwise_zeroize(rk_ctx* ctx, uint32_t* d_io, size_t n);
int fri_fold(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t out_count, const bb::Ext& mix);
int fri_fold_evals(rk_ctx* ctx, uint32_t* d_out_ext, const uint32_t* d_in_ext, size_t n_out, const bb::Ext& beta);
int gather_sample(rk_ctx* ctx, uint32_t* d_dst, const uint32_t* d_src, size_t idx, size_t size, size_t stride);
// rows: d_dst[q*cols + c] = d_matrix[c*rows + h_idx[q]]
int gather_rows(rk_ctx* ctx, uint32_t* d_dst, const uint32_t* d_matrix, size_t rows, size_t cols,
                const uint32_t* d_idx, size_t n_idx);
// digests: d_dst[i] = d_nodes[d_idx[i]]
int gather_digests(rk_ctx* ctx, uint32_t* d_dst, const uint32_t* d_nodes, const uint32_t* d_idx, size_t n_idx);
// openings of several trees in two launches: job k gathers rows idx[idx_off .. +n) of its column-major matrix to
// dst[dst_off ..) (n x cols) and, behind them, the digests idx[idx_off + n .. + n * path_len) of its node heap
struct GatherJob {
    const uint32_t* matrix;
    const uint32_t* nodes;
    uint64_t rows, cols, idx_off, dst_off;
    uint32_t n, path_len;
};
int gather_many(rk_ctx* ctx, uint32_t* d_dst, const uint32_t* d_idx, const GatherJob* d_jobs, const GatherJob* h_jobs, size_t n_jobs);
int ext_powers(rk_ctx* ctx, uint32_t* d_pw_ext, const bb::Ext& x, size_t n, bool bit_reversed);
// n_pts tables of n powers each, one launch: table j (at d_pw_ext + j * n * 4) holds h_pts[j]^k
int ext_powers_many(rk_ctx* ctx, uint32_t* d_pw_ext, const bb::Ext* h_pts, size_t n_pts, size_t n, bool bit_reversed);
int bit_reverse_ext(rk_ctx* ctx, uint32_t* d_io_ext, size_t size, size_t count);
// levels of a Merkle tree with <= 1024 parents, fused in one launch (kernels_hash.hip)
// every level from the one with top_output_size (<= HASH_FOLD_TOP_MAX, a power of two) parents up to the root
constexpr size_t HASH_FOLD_TOP_MAX = 4096;
int hash_fold_top(rk_ctx* ctx, uint32_t* d_nodes, size_t top_output_size);
// d_out_ext[e] = sum_k coeffs[which[e]*size + k] * pw[pw_sel[e]*size + k]
int eval_dot(rk_ctx* ctx, uint32_t* d_out_ext, const uint32_t* d_coeffs, size_t size, const uint32_t* d_which,
             const uint32_t* d_pw_ext, const uint32_t* d_pw_sel, size_t eval_count);
int mix_poly_coeffs(rk_ctx* ctx, uint32_t* d_out_ext, const bb::Ext& mix_start, const bb::Ext& mix,
                    const uint32_t* d_in, const uint32_t* h_combos, size_t input_size, size_t count);
int poly_divide(rk_ctx* ctx, uint32_t* d_poly_ext, size_t count, const bb::Ext& z, bb::Ext* h_rem);
// n_items polynomials of `count` ext coefficients at d_base_ext + h_offsets[i] (in ext elements), each divided
// by (x - h_z[i]) in one batch of launches; remainders to h_rems (may be null)
int poly_divide_many(rk_ctx* ctx, uint32_t* d_base_ext, size_t count, const size_t* h_offsets, const bb::Ext* h_z,
                     size_t n_items, bb::Ext* h_rems, uint32_t* d_rems = nullptr);  // d_rems: n_items x 4 words, no wait
int prefix_products(rk_ctx* ctx, uint32_t* d_io_ext, size_t count);
int scatter(rk_ctx* ctx, uint32_t* d_into, size_t into_words, const uint32_t* h_index, size_t n_cycles,
            const uint32_t* h_offsets, const uint32_t* h_values);
// circuit_program.hip: the step program behind rk_circuit_hooks.program / rk_verify_opts.program
int program_eval_check(const rk_program* prog, const rk_circuit_view* view, const uint32_t poly_mix[4], uint32_t* d_check);
int program_poly_ext(const rk_program* prog, uint32_t wm, const uint32_t poly_mix[4], const uint32_t* eval_u_ext, size_t n_taps,
                     const uint32_t* globals, uint32_t n_globals, const uint32_t* mix, uint32_t n_mix, uint32_t out_ext[4]);
// d_ext[idx[i]] -= delta[i]
// Plonky3 two-adic PCS steps on row-major matrices (kernels_pcs.hip)
// keep_cols != nullptr: *keep_cols receives the column-major natural-order evaluations (w columns of h << blow-up
// words, a dev_alloc'd block the caller frees) the rows were transposed from
int pcs_coset_lde_rows(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t h, size_t w, uint32_t** keep_cols = nullptr);
// column-major w columns of H words (natural order) -> row-major H x w with row bitrev(j) = index j
int pcs_cols_to_rows_bitrev(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_cols, size_t H, size_t w);
// cols: d_lde is column-major in natural order (rk_matrix layout 2) instead of row-major with bit-reversed rows
int pcs_eval_at(rk_ctx* ctx, uint32_t* d_out_ext, const uint32_t* d_lde, size_t H, size_t w, const uint32_t* h_points, size_t n_points,
                bool cols = false);
int pcs_reduce_openings(rk_ctx* ctx, uint32_t* d_ro_ext, const uint32_t* d_lde, size_t H, size_t w, size_t n_points,
                        const uint32_t* h_points, const uint32_t* h_ys, const bb::Ext& alpha, uint64_t alpha_offset, bool cols = false);
int pcs_coset_lde_cols(rk_ctx* ctx, uint32_t* d_cols, const uint32_t* d_in, size_t h, size_t w);
int ext_sub_at(rk_ctx* ctx, uint32_t* d_ext, const uint32_t* h_idx, const bb::Ext* h_delta, size_t n);
// p3.hip: the contexts rk_p3_prove_shards keeps per device (freed by rk_session_release)
void p3_release_pools();

}  // namespace rk
