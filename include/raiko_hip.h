/* raiko_hip.h -- C ABI of libraiko_hip.so, the MI355X (gfx950) STARK proving
 * backend for raiko's risc0 block-proof path.
 *
 * Boundary being replaced (reference = Champii/raiko @ 2024-08-07):
 *   - plugin level:   `trait Prover { run, cancel }`        lib/src/prover.rs:52-62
 *                     `Risc0Prover::run`                    provers/risc0/driver/src/lib.rs:56-112
 *                     `prove_locally` -> `session.prove()`  provers/risc0/driver/src/bonsai.rs:230-272
 *   - operator level: `risc0_zkp::hal::Hal` of risc0-zkp 1.0.1 (Cargo.lock:7243), the trait
 *                     risc0's own CUDA/Metal backends implement; that crate is not vendored
 *                     in the reference, so each entry point cites the trait method by name.
 *
 * Rules of the ABI: plain pointers and sizes only; every function returns
 * RK_OK (0) or a negative rk_status and never aborts; `d_` pointers are device
 * memory of the ctx's GPU; work is asynchronous on the ctx stream unless the
 * function returns data to the host (those synchronise the stream).  A ctx is
 * used by one thread at a time; different ctxs, also of one GPU, run concurrently
 * (rk_prove_session does exactly that).
 * All field elements are BabyBear Montgomery residues (u32 < p = 15*2^27+1);
 * extension elements are 4 consecutive u32; digests are 8 consecutive u32.
 * Matrices are column-major: element (row r, column c) at c*rows + r.
 */
#ifndef RAIKO_HIP_H
#define RAIKO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RK_OK = 0,
    RK_ERR_INVALID = -1,   /* bad argument (size not a power of two, null pointer, ...) */
    RK_ERR_HIP = -2,       /* a HIP runtime call failed; see rk_last_error */
    RK_ERR_NOMEM = -3,
    RK_ERR_NODEVICE = -4,
    RK_ERR_CAPACITY = -5,  /* caller-provided output buffer too small */
    RK_ERR_INTERNAL = -6,  /* a prover invariant failed (non-zero remainder in the DEEP division) */
    RK_ERR_VERIFY = -7,    /* rk_prove_session: a produced seal did not pass rk_verify_segment */
    RK_ERR_CALLBACK = -8   /* a circuit hook (rk_circuit_hooks) returned non-zero */
} rk_status;

typedef struct rk_ctx rk_ctx;

/* ---- library / device management ---- */
int rk_abi_version(void);
const char* rk_strerror(int status);
const char* rk_last_error(rk_ctx* ctx);            /* detail text of the last RK_ERR_HIP */
int rk_device_count(int* count);
/* stream = a hipStream_t the caller owns (e.g. torch's current stream) or NULL for a private one */
int rk_ctx_create(int device, void* stream, rk_ctx** out);
int rk_ctx_destroy(rk_ctx* ctx);
int rk_sync(rk_ctx* ctx);
int rk_alloc(rk_ctx* ctx, size_t bytes, void** d_ptr);
int rk_free(rk_ctx* ctx, void* d_ptr);
int rk_h2d(rk_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int rk_d2h(rk_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);

/* Poseidon2 t=24 constants, Montgomery form: 192 external round constants (8 rounds x 24),
 * 21 internal round constants, 24 internal-diagonal entries.  Defaults are compiled in. */
int rk_set_poseidon2_params(rk_ctx* ctx, const uint32_t* rc_ext, const uint32_t* rc_int, const uint32_t* diag);

/* ---- the parameter blob: everything instance-specific about the proof system in one place ----
 * (the same kernels serve risc0's parameter set and SP1 / Plonky3's;
 * presets carry the RECALLED values of both, any field can be overridden.)  Field elements in the
 * blob are CANONICAL integers except the Poseidon2 tables, which are Montgomery residues like
 * every buffer of the ABI.  rk_set_params validates (ext_w a non-residue, root of exact order
 * 2^27, shapes in range), rebuilds the context's twiddle / shift tables and replaces its
 * Poseidon2 instance; RK_ERR_INVALID leaves the context unchanged. */
typedef enum { RK_PRESET_RISC0 = 0, RK_PRESET_SP1 = 1 } rk_preset;
typedef struct {
    uint32_t struct_size;        /* = sizeof(rk_params): guards against a caller built for another layout */
    /* field */
    uint32_t ext_w;              /* extension Fp[x]/(x^4 - ext_w): p - 11 (risc0: x^4 + 11), 11 (Plonky3: x^4 - 11) */
    uint32_t root_2_27;          /* generator of the 2^27 subgroup: 137 (risc0), 0x1a427a41 (Plonky3) */
    uint32_t coset_shift;        /* shift of the zk / LDE coset: 3 (risc0), 31 (Plonky3) */
    /* Poseidon2 */
    uint32_t p2_width;           /* 24 (rate 16, R_P 21) or 16 (rate 8, R_P 13); R_F = 8, x^7, digest = 8 cells */
    uint32_t p2_m4;              /* 4x4 block of the external layer: 0 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]]
                                  * (Poseidon2 paper, risc0), 1 = circ(2,3,1,1) (Plonky3 MDSMat4) */
    uint32_t p2_pad_free;        /* sponge: 0 = zero-pad the last partial block (risc0), 1 = padding-free overwrite
                                  * (Plonky3 PaddingFreeSponge: cells beyond the input keep their values) */
    const uint32_t* p2_rc_ext;   /* 8 * p2_width external round constants; NULL = derived defaults */
    const uint32_t* p2_rc_int;   /* 21 or 13 internal round constants; NULL = derived defaults */
    const uint32_t* p2_diag;     /* p2_width internal-diagonal entries (matrix 1 1^T + diag); NULL = defaults */
    /* protocol */
    uint32_t queries;            /* 50 (risc0), 100 (SP1 core) */
    uint32_t blowup_log2;        /* 2 (risc0), 1 (SP1 core); 1..4.  The LDE domain of a segment has 2^(po2 + blowup_log2)
                                  * points, the check polynomial 4 * 2^blowup_log2 columns, hooks see that domain */
    uint32_t fri_fold_log2;      /* 4 (risc0), 1 (Plonky3): FRI folds by 2^fri_fold_log2 per round (1..4) */
    uint32_t fri_min_degree;     /* 256 (risc0), 1 (Plonky3: down to a constant): folding stops at this many coefficients */
    uint32_t pow_bits;           /* 0 (risc0), 16 (SP1 core): proof-of-work bits ground before the queries are drawn
                                  * (rk_pow_grind), 0..24 */
} rk_params;
int rk_params_preset(rk_params* out, int preset);
int rk_set_params(rk_ctx* ctx, const rk_params* params);
int rk_get_params(rk_ctx* ctx, rk_params* out);    /* pointers in *out refer to the context's own copies */

/* ---- Hal trait operators (risc0-zkp 1.0.1 hal/mod.rs `trait Hal`) ---- */
/* Hal::batch_interpolate_ntt: `count` columns of `size` natural-order evaluations ->
 * bit-reversed coefficients, in place. */
int rk_batch_interpolate_ntt(rk_ctx* ctx, uint32_t* d_io, size_t size, size_t count);
/* Hal::batch_evaluate_ntt: bit-reversed coefficients -> natural-order evaluations, in place. */
int rk_batch_evaluate_ntt(rk_ctx* ctx, uint32_t* d_io, size_t size, size_t count, uint32_t expand_bits);
/* Hal::zk_shift: io[col][i] *= 3^bitrev(i). */
int rk_zk_shift(rk_ctx* ctx, uint32_t* d_io, size_t size, size_t count);
/* Hal::batch_expand_into_evaluate_ntt: zero-extend `in_size` bit-reversed coefficients to
 * in_size << expand_bits and evaluate (natural order out). */
int rk_batch_expand_into_evaluate_ntt(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t in_size,
                                      size_t count, uint32_t expand_bits);
/* Hal::batch_bit_reverse */
int rk_batch_bit_reverse(rk_ctx* ctx, uint32_t* d_io, size_t size, size_t count);
/* Hal::hash_rows: one Poseidon2 sponge digest per row of a column-major rows x cols matrix. */
int rk_hash_rows(rk_ctx* ctx, uint32_t* d_out_digests, const uint32_t* d_matrix, size_t rows, size_t cols);
/* Hal::hash_fold: nodes[out_size + i] = H(nodes[2*(out_size+i)], nodes[2*(out_size+i)+1]), i < out_size. */
int rk_hash_fold(rk_ctx* ctx, uint32_t* d_nodes, size_t input_size, size_t output_size);
/* Hal::batch_evaluate_any: out[e] = sum_k coeffs[which[e]*size + k] * xs[e]^k (host arrays for
 * which/xs/out: they are transcript-sized). */
int rk_batch_evaluate_any(rk_ctx* ctx, const uint32_t* d_coeffs, size_t poly_count, size_t size,
                          const uint32_t* h_which, const uint32_t* h_xs, size_t eval_count, uint32_t* h_out);
/* Hal::mix_poly_coeffs: out[combos[i]*count + idx] += mix_start * mix^i * in[i*count + idx]. */
int rk_mix_poly_coeffs(rk_ctx* ctx, uint32_t* d_out_ext, const uint32_t mix_start[4], const uint32_t mix[4],
                       const uint32_t* d_in, const uint32_t* h_combos, size_t input_size, size_t count);
/* Hal::eltwise_* */
int rk_eltwise_add_elem(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_a, const uint32_t* d_b, size_t n);
int rk_eltwise_sum_extelem(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in_ext, size_t count, size_t to_add);
int rk_eltwise_copy_elem(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t n);
int rk_eltwise_zeroize_elem(rk_ctx* ctx, uint32_t* d_io, size_t n);
/* Hal::fri_fold: fold of 4 coefficient planes by 2^fri_fold_log2 of the context's parameters
 * (16 by default; bit-reversed order in and out). */
int rk_fri_fold(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t out_count, const uint32_t mix[4]);
/* Plonky3's FRI fold (p3-fri `fold_even_odd`, RECALLED; SP1's arity-2 fold works on evaluations, risc0's
 * rk_fri_fold on coefficients): d_in_ext = 2 * n_out extension elements (4 consecutive words each), the
 * evaluations of p over the subgroup of order 2 * n_out in bit-reversed order; d_out_ext = the n_out evaluations
 * of p_even + beta * p_odd over the squared subgroup, bit-reversed:
 *   out[i] = (p(x) + p(-x)) / 2 + beta * (p(x) - p(-x)) / (2 x),   x = g^bitrev(i). */
int rk_fri_fold_evals(rk_ctx* ctx, uint32_t* d_out_ext, const uint32_t* d_in_ext, size_t n_out, const uint32_t beta[4]);
/* Hal::gather_sample: dst[g] = src[g*stride + idx], g < size. */
int rk_gather_sample(rk_ctx* ctx, uint32_t* d_dst, const uint32_t* d_src, size_t idx, size_t size, size_t stride);

/* Hal::prefix_products: in-place inclusive running product of `count` extension elements
 * (io[i] = io[0] * ... * io[i]); the grand-product accumulator of the accum group. */
int rk_prefix_products(rk_ctx* ctx, uint32_t* d_io_ext, size_t count);
/* Hal::scatter: for cycle c < n_cycles and k in [h_index[c], h_index[c+1]):
 * d_into[h_offsets[k]] = h_values[k] (witness-generation helper; index has n_cycles + 1 entries,
 * the host arrays are copied before the call returns; a later entry wins over an earlier one
 * with the same offset).  Offsets >= into_words are rejected with RK_ERR_INVALID. */
int rk_scatter(rk_ctx* ctx, uint32_t* d_into, size_t into_words, const uint32_t* h_index, size_t n_cycles,
               const uint32_t* h_offsets, const uint32_t* h_values);

/* Proof of work on the transcript (Plonky3 challenger `grind` / `check_witness`, restated on this
 * library's Poseidon2 transcript; risc0 has none): the smallest nonce w (a field element written as the
 * u32 < p it is stored as) such that after absorbing hash([w]) the next random_bits(bits) are all
 * zero.  sponge_cells = the transcript generator's p2_width state words right after a commit.  One
 * lane per candidate, two permutations each. */
int rk_pow_grind(rk_ctx* ctx, const uint32_t* sponge_cells, uint32_t bits, uint32_t* nonce);

/* ---- fused building blocks (no single Hal counterpart) ---- */
/* MerkleTreeProver::new (risc0-zkp prove/merkle.rs): hash_rows + every hash_fold level into
 * d_nodes (2*rows digests, heap order, root at index 1). */
int rk_merkle_build(rk_ctx* ctx, uint32_t* d_nodes, const uint32_t* d_matrix, size_t rows, size_t cols);
/* ---- mixed-matrix commitment: Plonky3's MerkleTreeMmcs (p3-merkle-tree `MerkleTree::new`, RECALLED; what
 * SP1 commits a shard's per-chip traces with -- provers/sp1/driver/src/lib.rs:48-57 reaches it through
 * sp1-sdk) on this library's Poseidon2 sponge and 2-to-1 compression ----
 * Matrices of different power-of-two heights, row-major (element (r, c) at r * width + c, Plonky3's
 * RowMajorMatrix) or column-major, in one tree: the leaves are the hashes of the concatenated rows of
 * the tallest matrices (given order); going up, a level whose size equals the height of further
 * matrices takes them in -- node = compress(compress(left, right), hash(concatenated rows of those
 * matrices)).  d_nodes: 2 * H digests in heap order (H = largest height; leaves at H + i, root at 1). */
typedef struct {
    const uint32_t* d_values;      /* device */
    uint32_t height;               /* power of two */
    uint32_t width;
    uint32_t row_major;            /* 1: (r, c) at r * width + c; 0: at c * height + r; 2: at c * height + bitrev(r) -- a column-major
                                    * matrix whose columns are in NATURAL order while the committed rows are the bit-reversed
                                    * ones (Plonky3 commits `lde.bit_reverse_rows()`): the evaluations exactly as the NTT leaves
                                    * them, no reordering pass */
} rk_matrix;
int rk_mmcs_commit(rk_ctx* ctx, const rk_matrix* mats, uint32_t n_mats, uint32_t* d_nodes, uint32_t h_root[8]);
/* Mmcs::open_batch at leaf `index` of the tallest matrices: row (index >> log2(H / height)) of every
 * matrix, concatenated in the given order into h_rows (sum of widths words), and the log2(H) sibling
 * digests from the leaf level up into h_path. */
int rk_mmcs_open(rk_ctx* ctx, const rk_matrix* mats, uint32_t n_mats, const uint32_t* d_nodes, uint32_t index,
                 uint32_t* h_rows, uint32_t* h_path);
/* Mmcs::verify_batch on the host (no GPU): heights / widths of the matrices in commit order, the opened rows
 * and path, against `root`.  params NULL = risc0's Poseidon2 instance.  0 = accepted, 1 = rejected. */
int rk_mmcs_verify(const rk_params* params, const uint32_t* heights, const uint32_t* widths, uint32_t n_mats,
                   uint32_t index, const uint32_t* rows, const uint32_t* path, const uint32_t root[8]);

/* ---- Plonky3 two-adic FRI PCS, the data-parallel steps (SP1: provers/sp1/driver/src/lib.rs:48-57 -> sp1-core ->
 * p3-fri TwoAdicFriPcs::commit / open; RECALLED, the crates are outside the reference tree).  Matrices are
 * row-major height x width; coset shift, 2-adic generator, extension and blow-up come from rk_set_params. ----
 * commit: `coset_lde_batch(evals, log_blowup, shift).bit_reverse_rows()`: d_in = evaluations of `width` polynomials
 * over the subgroup of order `height` (natural order); d_out = (height << blowup_log2) x width, row r = their values
 * at shift * g^bitrev(r), g generating the larger subgroup.  Feed d_out to rk_mmcs_commit. */
int rk_pcs_coset_lde_rows(rk_ctx* ctx, uint32_t* d_out, const uint32_t* d_in, size_t height, size_t width);
/* open: the opened values of one committed matrix at the extension point z (`interpolate_coset` on the LDE's low
 * coset, i.e. its first lde_height >> blowup_log2 rows): d_out_ext = width extension elements p_c(z). */
int rk_pcs_eval_at(rk_ctx* ctx, uint32_t* d_out_ext, const uint32_t* d_lde, size_t lde_height, size_t width, const uint32_t z[4]);
/* the same for n_points (<= 4) points in one pass over the low coset (a trace is opened at zeta and zeta * g):
 * h_points = n_points x 4 words, d_out_ext = n_points x width extension elements, point-major. */
int rk_pcs_eval_at_many(rk_ctx* ctx, uint32_t* d_out_ext, const uint32_t* d_lde, size_t lde_height, size_t width, uint32_t n_points,
                        const uint32_t* h_points);
/* open, "reduce rows": for the n_points (<= 8) opening points of one matrix, h_points = n_points x 4 words and
 * h_opened = n_points x width x 4 words (the values rk_pcs_eval_at returned),
 *   d_ro_ext[r] += alpha^(alpha_offset + j * width) * (sum_c alpha^c M[r][c] - sum_c alpha^c opened_j[c]) / (x_r - z_j)
 * over all rows r and points j, x_r = shift * g^bitrev(r).  d_ro_ext (lde_height extension elements, one vector per
 * matrix height, zeroed by the caller) is what the FRI commit phase folds with rk_fri_fold_evals. */
int rk_pcs_reduce_openings(rk_ctx* ctx, uint32_t* d_ro_ext, const uint32_t* d_lde, size_t lde_height, size_t width, uint32_t n_points,
                           const uint32_t* h_points, const uint32_t* h_opened, const uint32_t alpha[4], uint64_t alpha_offset);
/* The same three steps for hosts that can keep the LDE the way the NTT leaves it -- `width` columns of
 * height << blowup_log2 evaluations in NATURAL order, i.e. rk_matrix layout 2 (row_major = 2: committed row r at index
 * bitrev(r) of every column).  rk_mmcs_commit / rk_mmcs_open take that layout as it is; the values returned here are those
 * of the row-major forms above (same d_out_ext, same d_ro_ext indexed by committed row).  No pass exists only to reorder:
 * a 2^20 x 256 LDE costs 3.4 ms instead of 5.0 (rk_p3_prove works this way). */
int rk_pcs_coset_lde_cols(rk_ctx* ctx, uint32_t* d_cols, const uint32_t* d_in_rows, size_t height, size_t width);
int rk_pcs_eval_at_many_cols(rk_ctx* ctx, uint32_t* d_out_ext, const uint32_t* d_lde_cols, size_t lde_height, size_t width, uint32_t n_points,
                             const uint32_t* h_points);
int rk_pcs_reduce_openings_cols(rk_ctx* ctx, uint32_t* d_ro_ext, const uint32_t* d_lde_cols, size_t lde_height, size_t width, uint32_t n_points,
                                const uint32_t* h_points, const uint32_t* h_opened, const uint32_t alpha[4], uint64_t alpha_offset);
/* `challenger.grind(bits)` of Plonky3's DuplexChallenger (p3-challenger, RECALLED): sponge_state = the challenger's
 * `width` state cells (Montgomery words), input_buffer = its n_input (< rate = width - 8) buffered observations.
 * *witness = the smallest field element w (canonical integer) for which observing w and then sample_bits(bits) gives
 * zero: w joins the inputs, they overwrite the first cells, one permutation, the last rate cell is the sample.
 * Plonky3 accepts any such w (`find_any`); the smallest is returned so that the result is reproducible. */
int rk_duplex_grind(rk_ctx* ctx, const uint32_t* sponge_state, const uint32_t* input_buffer, uint32_t n_input, uint32_t bits,
                    uint32_t* witness);

/* synthetic division of one extension polynomial (count coefficients, natural order) by (x - z),
 * in place (core/poly.rs poly_divide); the remainder f(z) goes to h_rem (4 words, may be NULL). */
int rk_poly_divide(rk_ctx* ctx, uint32_t* d_polys_ext, size_t count, const uint32_t z[4], uint32_t* h_rem);

/* ---- whole-segment prover (Prover::commit_group/finalize + fri_prove of risc0-zkp 1.0.1) ---- */
typedef struct {
    uint32_t group_size[3];        /* columns per register group: 0 accum, 1 code, 2 data */
    uint32_t n_regs;               /* registers sorted by (group, offset); must cover every column once */
    const uint32_t* reg_group;
    const uint32_t* reg_offset;
    const uint32_t* reg_combo;
    uint32_t n_combos;
    const uint32_t* combo_off;     /* n_combos + 1 prefix offsets into combo_backs */
    const uint32_t* combo_backs;
} rk_taps;

/* ---- circuit hooks: the two places where a segment proof depends on Fiat-Shamir randomness that
 * exists only after earlier groups are committed (risc0-circuit-rv32im 1.0.1 prove/mod.rs
 * prove_segment + risc0-zkp prove/prover.rs finalize; CircuitHal::accumulate / eval_check) ----
 *   1. after code and data are committed the prover draws n_accum_mix field elements and calls
 *      `accumulate`, which fills the accum group from them and the witness;
 *   2. after accum is committed it draws poly_mix and calls `eval_check`, which evaluates the
 *      circuit's mixed constraint polynomial over the 4N-point LDE domain.
 * Hooks run on the thread that called rk_prove_segment (a prover thread of rk_prove_session),
 * may call rk_* operators on view->ctx, must enqueue their device work on view->stream and must
 * not synchronise other streams; they return 0 or non-zero (-> RK_ERR_CALLBACK). */
typedef struct {
    rk_ctx* ctx;
    void* stream;                  /* hipStream_t of ctx */
    uint32_t po2;
    uint32_t group_size[3];        /* 0 accum, 1 code, 2 data */
    const uint32_t* d_trace[3];    /* device, column-major 2^po2 x group_size[g]: the witness as handed in
                                    * ([1], [2]; valid in `accumulate` only, NULL otherwise) */
    const uint32_t* d_lde[3];      /* device, column-major 4*2^po2 x group_size[g]: evaluations on the coset
                                    * 3*w^i, natural order (PolyGroup::evaluated); NULL until committed.
                                    * (4 = 2^blowup_log2, 3 = coset_shift of the context's rk_params) */
    const uint32_t* globals;       /* host */
    uint32_t n_globals;
    const uint32_t* mix;           /* host: the n_accum_mix elements drawn before the accum commit */
    uint32_t n_mix;
} rk_circuit_view;
typedef struct rk_program rk_program;   /* a circuit's constraint polynomial as data, see below */
typedef struct {
    void* user;
    /* CircuitHal::accumulate: write the accum group, column-major 2^po2 x group_size[0], to d_accum */
    int (*accumulate)(void* user, const rk_circuit_view* view, uint32_t* d_accum);
    /* CircuitHal::eval_check: write 4 x 4*2^po2 values (component e of point i at e*4*2^po2 + i):
     * sum_k poly_mix^k * constraint_k at x_i = 3*w^i, divided by (x_i^(2^po2) - 1) */
    int (*eval_check)(void* user, const rk_circuit_view* view, const uint32_t poly_mix[4], uint32_t* d_check);
    /* used when eval_check is NULL: the library evaluates the step program on the GPU
     * (rk_program_eval_check) -- no circuit-specific kernel needed */
    const rk_program* program;
} rk_circuit_hooks;

/* ---- the constraint polynomial as data: risc0-zkp 1.0.1 adapter.rs `PolyExtStepDef` ----
 * risc0 ships every circuit's mixed constraint polynomial as a list of steps over two growing
 * value lists (RECALLED from the crate; it is outside the reference tree): field values
 * (CONST / GET / GET_GLOBAL / ADD / SUB / MUL push one) and mix states {tot, mul} (TRUE /
 * AND_EQZ / AND_COND push one); operands are positions in the list of their kind:
 *   CONST a            value a (a canonical integer)
 *   GET a              tap a: eval_u[a] (rk_poly_ext_fn's order: registers by (group, offset), backs in combo order)
 *   GET_GLOBAL a b     args[a][b]: a = 0 the segment's globals, a = 1 the accum mix
 *   ADD|SUB|MUL a b    field values a, b
 *   TRUE               {tot 0, mul 1}
 *   AND_EQZ a b        x = mix state a, v = field value b:   {x.tot + x.mul * v,  x.mul * poly_mix}
 *   AND_COND a b c     x = mix state a, cond = field value b, inner = mix state c:
 *                      {x.tot + cond * inner.tot * x.mul,  x.mul * inner.mul}
 * and `ret` names the mix state whose tot is the result.  The verifier interprets the list on the
 * tap openings (CircuitDef::poly_ext); risc0's eval_check kernels are the same list turned into
 * straight-line code by its build.  Here the list is an operand: rk_program_create validates it,
 * drops dead steps, folds every `mul` into a compile-time power of poly_mix, assigns the live
 * intermediate values to reusable slots, and the result serves both sides --
 * rk_program_eval_check runs it for all 4 * 2^po2 points of the LDE domain on the GPU (one point per
 * lane, slots in LDS, the step list read through the scalar cache), rk_program_poly_ext runs it on
 * extension elements on the host.  tools/circuit_gen.py turns the same list into straight-line HIP
 * (what risc0's build does) when the interpreter's per-step overhead matters. */
typedef enum {
    RK_STEP_CONST = 0, RK_STEP_GET = 1, RK_STEP_GET_GLOBAL = 2, RK_STEP_ADD = 3, RK_STEP_SUB = 4, RK_STEP_MUL = 5,
    RK_STEP_TRUE = 6, RK_STEP_AND_EQZ = 7, RK_STEP_AND_COND = 8
} rk_step_op;
typedef struct { uint32_t op, a, b, c; } rk_poly_step;
typedef struct {
    uint64_t n_steps;          /* as given */
    uint64_t n_ops;            /* arithmetic steps left after dead-code elimination (what a point costs) */
    uint32_t n_fp_slots;       /* slots the live field values need at the same time */
    uint32_t n_mix_slots;      /* ... and the live mix states (4 words each) */
    uint32_t n_consts;         /* distinct constants */
    uint32_t n_mix_powers;     /* distinct powers of poly_mix */
    uint32_t max_power;        /* the highest of them = number of constraints on the longest path */
    uint32_t n_taps;           /* taps of the tap set it was created against */
} rk_program_info;
/* taps: the tap set the GET indices refer to (copied).  RK_ERR_INVALID for an operand that names a
 * value not yet pushed, a tap / argument list that does not exist, or an unknown op. */
int rk_program_create(const rk_poly_step* steps, size_t n_steps, uint32_t ret, const rk_taps* taps, rk_program** out);
int rk_program_destroy(rk_program* prog);
int rk_program_get_info(const rk_program* prog, rk_program_info* out);
/* CircuitHal::eval_check from the program, on view->ctx / view->stream (what the prover calls for
 * hooks with `program` set; a hook of the caller's may call it too) */
int rk_program_eval_check(const rk_program* prog, const rk_circuit_view* view, const uint32_t poly_mix[4],
                          uint32_t* d_check);
/* Optional: turn the list into a gfx950 code object for ctx's GPU now (straight-line HIP generated from the list
 * -- what tools/circuit_gen.py emits at build time -- compiled with hiprtc; seconds for 10^4 steps).  From then on
 * rk_program_eval_check on that GPU runs the generated kernel (about 3x the interpreter's speed) instead of the
 * interpreter; results are identical.  RK_ERR_HIP with the compiler's log in rk_last_error if it cannot be
 * built -- the interpreter stays in charge.
 * With RK_JIT_CACHE_DIR set, the code object is kept there under a name derived from the generated source, the
 * architecture and the hiprtc version, and loaded instead of compiled the next time (a host restart then costs
 * milliseconds, not the ~20 s of a 30 k-step list). */
int rk_program_compile(rk_program* prog, rk_ctx* ctx);
/* the HIP source rk_program_compile hands to the compiler (NUL-terminated; *length without the NUL;
 * RK_ERR_CAPACITY with *length set when `out` is too small) */
int rk_program_source(const rk_program* prog, char* out, size_t capacity, size_t* length);
/* CircuitDef::poly_ext from the program (host): the value rk_poly_ext_fn returns.  ext_w: canonical W
 * of the extension (rk_params.ext_w; 0 = risc0's) */
int rk_program_poly_ext(const rk_program* prog, uint32_t ext_w, const uint32_t poly_mix[4], const uint32_t* eval_u_ext,
                        size_t n_taps, const uint32_t* globals, uint32_t n_globals, const uint32_t* mix, uint32_t n_mix,
                        uint32_t out_ext[4]);

typedef struct {
    uint32_t po2;                  /* segment has 2^po2 rows */
    uint32_t on_device;            /* 0: group[] and check are host pointers; 1: device pointers, left untouched
                                    * (the prover works on a copy); 2: device pointers the prover may overwrite */
    rk_taps taps;
    const uint32_t* group[3];      /* trace evaluations, column-major 2^po2 x group_size[g] */
    const uint32_t* check;         /* eval_check output: 4 x 4*2^po2 evaluations (CircuitHal::eval_check) */
    const uint32_t* globals;       /* host, n_globals elements */
    uint32_t n_globals;
    uint32_t n_accum_mix;          /* Fiat-Shamir elements drawn before the accum group is committed */
    uint8_t proof_system_info[16];
    uint8_t circuit_info[16];
    const rk_circuit_hooks* hooks; /* NULL: group[0] and check are taken as given (pre-computed stand-ins).
                                    * A non-NULL `accumulate` replaces group[0], a non-NULL `eval_check`
                                    * replaces check (the replaced pointer is ignored and may be NULL) */
} rk_segment;

/* Produces the seal (the u32 Fiat-Shamir transcript) of one segment. */
int rk_prove_segment(rk_ctx* ctx, const rk_segment* seg, uint32_t* h_seal, size_t seal_capacity_words,
                     size_t* seal_words);
/* Host-side verifier (no GPU needed): the counterpart of `receipt.verify()` the reference calls
 * after proving (provers/risc0/driver/src/lib.rs:136, benchmark.rs:19); restates risc0-zkp
 * verify/{mod,fri,merkle}.rs for the flow of rk_prove_segment.  Reads only the public part of
 * `pub` (po2, taps, globals, n_accum_mix, infos).  Returns 0 for a valid seal, RK_ERR_INVALID for
 * malformed arguments, a positive reason code otherwise (see verify.hip).  The circuit's
 * constraint identity is NOT checked (no rv32im circuit in this repo). */
int rk_verify_segment(const rk_segment* pub, const uint32_t* seal, size_t seal_words);
/* The same with options.  p2_*: the Poseidon2 instance the seal was produced under (all three or
 * none; none = the compiled-in defaults, which is what rk_verify_segment assumes).  poly_ext:
 * CircuitDef::poly_ext -- the circuit's mixed constraint polynomial on the tap openings; when
 * given, the verifier also checks the constraint identity
 *     poly_ext(poly_mix, eval_u, globals, mix) == check(z) * ((3z)^(2^po2) - 1)
 * (reason code 70 on mismatch; the same with `program` in place of the callback).  eval_u holds one extension element per tap, registers in
 * (group, offset) order, each register's backs in combo order: the value of the register's
 * polynomial at 3*z*w^-back. */
typedef int (*rk_poly_ext_fn)(void* user, const rk_segment* pub, const uint32_t poly_mix[4], const uint32_t* eval_u_ext,
                              size_t n_taps, const uint32_t* mix, uint32_t n_mix, uint32_t out_ext[4]);
typedef struct {
    const uint32_t* p2_rc_ext;     /* 192 */
    const uint32_t* p2_rc_int;     /* 21 */
    const uint32_t* p2_diag;       /* 24 */
    rk_poly_ext_fn poly_ext;
    void* user;
    const rk_program* program;     /* used when poly_ext is NULL: the constraint identity from the step program */
    const rk_params* params;       /* optional: the parameter blob the seal was produced under (field, Poseidon2
                                    * instance incl. width 16, queries); overrides the three p2_* pointers */
} rk_verify_opts;
int rk_verify_segment_ex(const rk_segment* pub, const rk_verify_opts* opts, const uint32_t* seal, size_t seal_words);
/* Upper bound on the seal size for a given shape (0 for a shape rk_prove_segment rejects), for up to
 * RK_MAX_QUERIES queries' worth of openings when called without parameters. */
size_t rk_seal_bound_words(const rk_segment* seg);
size_t rk_seal_bound_words_for(const rk_segment* seg, uint32_t queries);
/* the same for a whole parameter set (queries, blow-up, fold arity, final degree, proof of work) */
size_t rk_seal_bound_words_params(const rk_segment* seg, const rk_params* params);
#define RK_MAX_QUERIES 256

/* ---- whole-session prover: what replaces `session.prove()` (provers/risc0/driver/src/bonsai.rs:271,
 * which proves the segments one after the other) ----
 * Proves segs[0..n) on one GPU with `inflight` segments in flight (one prover context, HIP stream
 * and host thread each, taken from a shared index) and, for host-resident segments
 * (on_device == 0), one more context that uploads `upload_ahead` segments ahead into a ring of
 * device buffers, so the PCIe transfer runs under the previous proofs.  With `verify` != 0 every
 * seal is checked with rk_verify_segment on one more host thread while the GPU goes on.  Seal i goes to
 * h_seals[i] (capacity seal_capacity_words[i], e.g. rk_seal_bound_words), its length to
 * seal_words[i].  Returns RK_OK or the first failure (RK_ERR_VERIFY for a seal that does not
 * verify) with the segment's index in *failed_index; rk_session_last_error gives the detail.
 * Contexts and staging buffers persist per device for the life of the process (the reference's
 * `Prover` has no `self`: lib/src/prover.rs:52-62); concurrent calls for one device are
 * serialised.  rk_session_release frees them. */
typedef struct {
    int device;        /* the GPU, when n_devices == 0 */
    int inflight;      /* per GPU, 1..16; 3 is where an MI355X saturates at 2^20-cycle segments */
    int upload_ahead;  /* per GPU, 0..16 staged segments waiting for a prover; 2 hides a 20 ms upload */
    int verify;        /* 0: no verification; 1: verify every seal (the library picks up to 4 host threads);
                        * 2..16: that many verifier threads */
    const int* devices; /* optional list of distinct GPUs of this node: every GPU's prover contexts take
                         * segments from ONE shared index (a work queue: a short last segment or a slower
                         * GPU does not stall the others), seals land in the caller's host buffers, so a
                         * single-process host needs no collective */
    int n_devices;      /* 0: use `device` */
    const rk_verify_opts* verify_opts; /* optional, for verify != 0 */
    const rk_params* params;    /* optional: the parameter set of this session's proofs (and of their verification,
                                 * unless verify_opts carries its own); NULL = risc0's.  The device's contexts are
                                 * re-parameterised only when the set differs from the previous session's */
} rk_session_opts;
int rk_prove_session(const rk_session_opts* opts, const rk_segment* segs, size_t n, uint32_t* const* h_seals,
                     const size_t* seal_capacity_words, size_t* seal_words, size_t* failed_index);
/* The same for a session whose segments arrive while it runs (risc0's executor can yield segments one by one:
 * `run_with_callback`): segment k is proven while the executor is still producing segment k + 1.  A worker thread
 * proves whatever has been submitted since its last look as one rk_prove_session batch.  The rk_segment is copied at
 * submit; what it points at, the seal buffer and *seal_words must stay valid until rk_stream_close returns -- which
 * waits for everything submitted, frees the stream, and returns RK_OK or the first failure with the index (in
 * submission order) in *failed_index.  opts as for rk_prove_session (devices / params / verify_opts are copied;
 * tables a params blob points at stay the caller's). */
typedef struct rk_stream rk_stream;
int rk_stream_open(const rk_session_opts* opts, rk_stream** out);
int rk_stream_submit(rk_stream* stream, const rk_segment* seg, uint32_t* h_seal, size_t seal_capacity_words, size_t* seal_words);
/* Back-pressure for hosts whose sessions are larger than memory (the reference keeps segments on disk for that reason:
 * segment_path, bonsai.rs:261-266): blocks until at most max_pending submitted segments are unfinished (or one has
 * failed: the status so far is returned), and reports in *finished_prefix how many segments from the start of the
 * submission order are done -- their inputs and everything their rk_segment pointed at may be freed, their seals and
 * seal_words are final.  A host submits segment k, waits with max_pending = inflight + upload_ahead, frees what the
 * prefix has released, and only then generates the witness of segment k + 1. */
int rk_stream_wait(rk_stream* stream, size_t max_pending, size_t* finished_prefix);
int rk_stream_close(rk_stream* stream, size_t* failed_index);
const char* rk_session_last_error(int device);
/* how many segments `device` proved in the last session it took part in (the balance of the work queue) */
int rk_session_last_proven(int device, size_t* count);
int rk_session_release(void);
/* Test switch: with RK_TEST_LOGICAL_DEVICES=k in the environment the session entry points see k devices, logical
 * device d running on physical GPU d mod (number of GPUs) -- the multi-device path (a pool, a feeder and `inflight`
 * provers per device, session-wide claim flags, pinning of device-resident segments to the first device of the GPU
 * that holds them) on a box with one GPU.  Not for production: the devices share one GPU's memory and time. */

/* ---- the one collective of the path, for hosts that run one process per GPU (raiko's own single-process
 * host needs none: rk_prove_session writes every seal into the caller's buffers) ----
 * Rank r proves global segments r, r + world, ... and the ranks exchange the variable-length seals with two
 * ncclAllGather calls over RCCL (xGMI inside a node): a length table, then padded payloads.  RCCL is looked up at run
 * time (librccl.so.1); RK_ERR_NODEVICE when it is not there.  rk_comm_unique_id is called by one rank, the 128 bytes go to
 * the others out of band (a file, MPI, the launcher's store), every rank then calls rk_comm_create. */
typedef struct rk_comm rk_comm;
#define RK_COMM_ID_BYTES 128
int rk_comm_unique_id(uint8_t id[RK_COMM_ID_BYTES]);
int rk_comm_create(const uint8_t id[RK_COMM_ID_BYTES], int rank, int world, int device, rk_comm** out);
int rk_comm_destroy(rk_comm* comm);
const char* rk_comm_last_error(rk_comm* comm);
/* h_local_seals / local_words: this rank's n_local seals in the order it proved them (segments rank, rank + world, ...;
 * n_local must be that count for n_total).  On return EVERY rank has out_words[i] for all i < n_total and, where h_out and
 * h_out[i] are given, seal i copied into h_out[i] (RK_ERR_CAPACITY if out_capacity[i] is too small; the others are
 * still delivered). */
int rk_gather_seals(rk_comm* comm, const uint32_t* const* h_local_seals, const size_t* local_words, size_t n_local, size_t n_total,
                    uint32_t* const* h_out, const size_t* out_capacity, size_t* out_words);
/* the host half of the gather: the gathered length table (world x per_rank words) and padded payloads
 * (world x per_rank x max_len words) into segment order */
int rk_gather_unpack(const uint32_t* all_lens, const uint32_t* all_payload, int world, size_t per_rank, size_t max_len, size_t n_total,
                     uint32_t* const* h_out, const size_t* out_capacity, size_t* out_words);

/* ---- RV32IM executor + segmenter: the step before the path ----
 * `ExecutorImpl::from_elf(env, elf).run()` (provers/risc0/driver/src/bonsai.rs:267-269): interprets a
 * 32-bit RISC-V ELF (RV32I + M) and cuts the run into segments of at most 2^segment_limit_po2
 * cycles (bonsai.rs:249), each with the machine-state digests before and after it.  Host code.
 * NOT risc0's: the cycle model (one cycle per instruction), the ecall table (t0 selects:
 * RK_ECALL_HALT a0 = exit code; RK_ECALL_READ a0 = word-aligned destination, a1 = capacity in
 * words -> a0 = words taken from input_words; RK_ECALL_COMMIT a0 = source, a1 = bytes appended to
 * the journal) and the state digest (Poseidon2 over pc, registers and touched pages).  The
 * rv32im witness layout is outside this repo: a segment carries bounds and digests, not columns. */
typedef struct rk_exec rk_exec;
enum { RK_ECALL_HALT = 0, RK_ECALL_READ = 1, RK_ECALL_COMMIT = 2 };
enum { RK_EXIT_HALTED = 0, RK_EXIT_SYSTEM_SPLIT = 2 };
typedef struct {
    uint32_t struct_size;          /* = sizeof(rk_exec_opts) */
    uint32_t segment_limit_po2;    /* 13..24 */
    uint64_t session_limit;        /* total cycles allowed, 0 = no limit (session_limit(None), bonsai.rs:248) */
    const uint32_t* input_words;   /* env.write_slice(&encoded_input), bonsai.rs:250 */
    size_t n_input_words;
    uint32_t record_trace;         /* keep every executed cycle (28 bytes each) for rk_exec_witness */
    uint32_t profile;              /* count executed cycles per pc (rk_exec_profile): what the reference switches on with
                                    * `profile: true` -> env_builder.enable_profiler(..), bonsai.rs:252-255 */
} rk_exec_opts;
typedef struct {
    uint64_t total_cycles;
    uint32_t n_segments;
    uint32_t exit_code;            /* a0 of the halting ecall */
    size_t journal_bytes;
    size_t input_words_read;
    int status;                    /* RK_OK, or why the run stopped (the segments before it stay readable) */
} rk_exec_summary;
typedef struct {
    uint32_t index;
    uint32_t po2;                  /* smallest power of two >= cycles, at least 13 */
    uint64_t cycles;
    uint32_t start_pc, end_pc;
    uint32_t exit;                 /* RK_EXIT_SYSTEM_SPLIT for every segment but the last */
    uint32_t pre_state[8], post_state[8];   /* post_state of segment i == pre_state of segment i + 1 */
} rk_exec_segment;
/* *out is set also when the run traps (status < 0): read rk_exec_error, then rk_exec_free */
int rk_exec_elf(const uint8_t* elf, size_t elf_bytes, const rk_exec_opts* opts, rk_exec** out);
/* the same one segment at a time (risc0's `run_with_callback` shape): a host can hand segment k to the prover
 * while segment k + 1 executes.  rk_exec_open loads the ELF (input words are copied), every
 * rk_exec_next_segment runs up to the next boundary; *more = 0 once the guest has halted. */
int rk_exec_open(const uint8_t* elf, size_t elf_bytes, const rk_exec_opts* opts, rk_exec** out);
int rk_exec_next_segment(rk_exec* ex, int* more);
int rk_exec_summary_get(const rk_exec* ex, rk_exec_summary* out);
int rk_exec_segment_get(const rk_exec* ex, uint32_t index, rk_exec_segment* out);
int rk_exec_journal(const rk_exec* ex, uint8_t* out, size_t capacity, size_t* len);
/* the cycle profile of a run made with rk_exec_opts.profile: distinct program counters and the cycles spent at each,
 * most expensive first; *n = how many there are (RK_ERR_CAPACITY when more than `capacity`: call again) */
int rk_exec_profile(const rk_exec* ex, uint32_t* pcs, uint64_t* cycles, size_t capacity, size_t* n);
/* Witness generation for the STAND-IN trace circuit (not rv32im's: that layout is in a crate outside the
 * reference tree).  Columns of executed segment `index`, column-major 2^po2 rows, Montgomery form, every 32-bit
 * machine word as two 16-bit field elements; needs rk_exec_opts.record_trace:
 *   code (RK_TRACE_CODE_COLS = 2):   0 first-row selector, 1 last-row selector
 *   data (RK_TRACE_DATA_COLS = 16):  0/1 pc lo/hi, 2/3 next pc lo/hi, 4/5 instruction lo/hi, 6 seq (next = pc + 4),
 *                                    7 carry of pc_lo + 4, 8/9 rs1 value, 10/11 rs2 value, 12/13 value written to rd,
 *                                    14 rd written, 15 active (0 on the padding rows after the last cycle)
 * The constraints a proof over them checks (raiko_amd/circuit_program.py trace_program): flags are bits, a seq row
 * advances pc by 4 with the stated carry, every row starts where the previous one went, padding is final, the
 * segment's first and last pc are the public ones.  Limb ranges, instruction decoding, registers and memory are
 * NOT constrained: this is the witness path exercised end to end, not a zkVM. */
#define RK_TRACE_CODE_COLS 2
#define RK_TRACE_DATA_COLS 16
int rk_exec_witness(const rk_exec* ex, uint32_t index, uint32_t* code, uint32_t* data);
/* The two tables the data columns look values up in when the segment is proven as a uni-stark shard with lookups
 * (rk_air_create_lookup; raiko_amd/executor.py p3_trace_air(lookups=True)) -- what SP1's cpu chip has in its program and
 * range chips.  Row-major Montgomery words, ready to be an rk_p3_table's trace:
 *   range_table    65536 x 2: (v, how often v occurs among the ten 16-bit limbs -- pc, next pc, rs1, rs2, rd value -- of
 *                  the executed cycles)
 *   program_table  rows x 5: (pc lo, pc hi, instruction lo, instruction hi, cycles spent there), the distinct pairs in
 *                  ascending order, zero-padded to a power of two >= 2
 * *program_rows: in = the capacity of program_table in rows, out = the rows needed (RK_ERR_CAPACITY when it did not fit,
 * nothing written: call again).  Needs rk_exec_opts.record_trace. */
int rk_exec_lookup_tables(const rk_exec* ex, uint32_t index, uint32_t* range_table, uint32_t* program_table, size_t* program_rows);
/* the same columns written on the GPU into device buffers (2 and 16 columns of 2^po2 words), asynchronously on the
 * ctx stream: only the executed cycles (28 bytes each) cross PCIe; the trace has been copied when the call returns.
 * Hand the buffers to rk_prove_segment / a session as on_device inputs after rk_sync(ctx). */
int rk_exec_witness_device(rk_ctx* ctx, const rk_exec* ex, uint32_t index, uint32_t* d_code, uint32_t* d_data);
/* the 16 data columns as ONE row-major matrix (2^po2 rows x RK_TRACE_DATA_COLS words) in device memory: the form an
 * on_device rk_p3_table takes (the execution proven as uni-stark shards: raiko_amd/executor.py p3_trace_air) */
int rk_exec_witness_device_rows(rk_ctx* ctx, const rk_exec* ex, uint32_t index, uint32_t* d_rows);
const char* rk_exec_error(const rk_exec* ex);
int rk_exec_free(rk_exec* ex);

/* ---- Plonky3-style STARK for AIRs handed over as data: the proof system behind SP1's `client.setup(ELF)` /
 * `client.prove(&pk, stdin)` (provers/sp1/driver/src/lib.rs:44-57; shard knobs docs/README_Sp1.md:19-32) as far as it
 * exists without SP1's chips -- p3-uni-stark prove / verify over p3-fri's TwoAdicFriPcs with a DuplexChallenger
 * (Plonky3@88ea2b8, Cargo.lock:4889-5127; RECALLED, the crates are outside the reference tree), for one or several
 * tables under shared challenges, the way sp1-core proves the chips of a shard, including the permutation (LogUp)
 * argument that ties the tables together (rk_air_create_lookup below).  NOT here: SP1's chips, its recursion / compress VM.
 *
 * An AIR is a step list, the shape `Air::eval` leaves in a symbolic builder: every step except ASSERT_ZERO pushes one
 * value; a, b name earlier values by their position in that list.
 *   CONST a              the canonical integer a
 *   LOCAL a / NEXT a     column a of the current / the next row (cyclic)
 *   PUBLIC a             public value a
 *   IS_FIRST_ROW, IS_LAST_ROW, IS_TRANSITION     the Lagrange selectors (unnormalised, as p3-commit domain.rs has them)
 *   ADD | SUB | MUL a b, NEG a
 *   ASSERT_ZERO a        ConstraintFolder::assert_zero: accumulator = accumulator * alpha + a
 *   PERM_LOCAL a / PERM_NEXT a   base column a of the table's permutation trace (rk_air_create_lookup), current / next row
 *   CHALLENGE a          base component a of the permutation challenges [alpha | beta^0 | beta^1 | ...] (4 words each)
 *   CUMSUM a             base component a of the table's cumulative sum
 * rk_air_create validates, derives the quotient degree from the symbolic degrees (get_log_quotient_degree: a cell and
 * is_first_row / is_last_row count 1, is_transition and constants 0) and translates the list into an rk_program whose
 * taps are the columns of the LDE, so the quotient is evaluated by the same GPU evaluator as risc0's eval_check;
 * rk_air_compile builds the straight-line kernel with hiprtc (optional, about 3x the interpreter). */
typedef enum {
    RK_AIR_CONST = 0, RK_AIR_LOCAL = 1, RK_AIR_NEXT = 2, RK_AIR_PUBLIC = 3, RK_AIR_IS_FIRST_ROW = 4, RK_AIR_IS_LAST_ROW = 5,
    RK_AIR_IS_TRANSITION = 6, RK_AIR_ADD = 7, RK_AIR_SUB = 8, RK_AIR_MUL = 9, RK_AIR_NEG = 10, RK_AIR_ASSERT_ZERO = 11,
    RK_AIR_PERM_LOCAL = 12, RK_AIR_PERM_NEXT = 13, RK_AIR_CHALLENGE = 14, RK_AIR_CUMSUM = 15
} rk_air_op;
typedef struct { uint32_t op, a, b; } rk_air_step;
typedef struct rk_air rk_air;
typedef struct {
    uint64_t n_steps;              /* as given */
    uint64_t n_ops;                /* arithmetic steps a point of the quotient domain costs */
    uint32_t n_constraints;
    uint32_t max_degree;           /* symbolic degree of the highest constraint */
    uint32_t log_quotient_degree;  /* log2_ceil(max(max_degree, 2) - 1): the quotient has 2^this chunks */
    uint32_t n_fp_slots;           /* live intermediate values (rk_program_info) */
} rk_air_info;
int rk_air_create(const rk_air_step* steps, size_t n_steps, uint32_t width, uint32_t n_public, rk_air** out);
/* An AIR whose table takes part in lookups, in the shape sp1-core gives them (stark/permutation.rs, lookup/interaction.rs,
 * air/builder.rs send / receive; sp1-core is pulled by the reference's Cargo.lock, RECALLED): interaction i says "every
 * row sends (kind 0) or receives (kind 1) the tuple (bus, local[value_cols]...) `mult` times", mult a main-trace column
 * or -- mult_is_const -- a canonical constant.  Flat form: per interaction the words kind, bus, mult_is_const, mult,
 * n_values followed by its n_values column numbers (n_words in all); at most 64 values per tuple and 120 distinct
 * columns over all interactions of a table.
 * The prover (rk_p3_prove) then, after the main traces are committed, draws two extension challenges alpha, beta,
 * fills the table's permutation trace on the GPU -- one extension column per batch of two interactions holding
 * sum +-mult / (alpha + beta^0 bus + sum_j beta^(j+1) value_j), and a last column with the running sum of the row totals
 * (4 base columns each: 4 * (ceil(n / 2) + 1) in all) --, commits those traces as a second batch and observes root and
 * cumulative sums before the constraint challenge; the verifier also checks that the cumulative sums of all tables add
 * up to zero (reason 8): every tuple sent is received as often.  The constraints tying the permutation trace to the
 * main trace are AIR steps over PERM_LOCAL / PERM_NEXT / CHALLENGE / CUMSUM -- sp1-core's eval_permutation_constraints,
 * every extension identity as four base asserts: per batch entry * prod rlc_i = sum_i +-mult_i * prod_(j != i) rlc_j,
 * phi[0] = sum entries[0], phi' = phi + sum entries' on transitions, phi[last] = cumulative sum.  ext_w != 0: `steps`
 * holds the table's own constraints only (what a chip's Air::eval emits) and the library appends those steps for the
 * extension x^4 - ext_w (the canonical W of the parameter set the proofs will be made under: 11 for SP1's); ext_w = 0:
 * `steps` already contains them (raiko_amd/p3.py AirBuilder can write them itself).  rk_air_get_steps reads the full
 * list back.  A proof without any interaction keeps the bytes it had before this entry point existed. */
int rk_air_create_lookup(const rk_air_step* steps, size_t n_steps, uint32_t width, uint32_t n_public,
                         const uint32_t* interaction_words, uint32_t n_interactions, size_t n_words, uint32_t ext_w, rk_air** out);
/* the AIR's complete step list (with the appended lookup constraints); RK_ERR_CAPACITY with *n_steps set when it does not fit */
int rk_air_get_steps(const rk_air* air, rk_air_step* out, size_t capacity, size_t* n_steps);
int rk_air_destroy(rk_air* air);
int rk_air_get_info(const rk_air* air, rk_air_info* out);
int rk_air_compile(rk_air* air, rk_ctx* ctx);
/* The Poseidon2 chip: a table whose every row is one permutation of the parameter set's instance (width 16 or 24, either
 * 4x4 block) with the intermediate values a degree-3 AIR needs in columns -- the shape of sp1-recursion-core's Poseidon2
 * wide chip (RECALLED; outside the reference tree), the table a recursion / compress layer spends most of its rows on:
 * every Merkle path step and sponge block of a proof it verifies is one lookup (bus: in[0..W), out[0..8)) into it.
 * Columns: in W | external rounds 0..3: cube W, state after the round W | internal rounds: cube of cell 0 (R_P), cell 0
 * entering rounds 1.. (R_P - 1) | state after the internal rounds W | external rounds 4..7 | multiplicity; x^7 = cube *
 * cube * x, so every constraint has degree 3 (two quotient chunks).  314 columns for width 16, 474 for width 24.
 * rk_p2_chip_air: the AIR (step list written by the library, one receive interaction on `bus` with the multiplicity
 * column); rk_p2_chip_trace: the rows on the GPU under the context's parameter set, one lane per permutation -- d_inputs
 * n x W row-major Montgomery words, d_mult n multiplicities or NULL for ones, d_trace n x rk_p2_chip_width row-major:
 * ready to be an on_device rk_p3_table.  params NULL = the SP1 preset. */
uint32_t rk_p2_chip_width(const rk_params* params);
int rk_p2_chip_air(const rk_params* params, uint32_t bus, rk_air** out);
int rk_p2_chip_trace(rk_ctx* ctx, const uint32_t* d_inputs, const uint32_t* d_mult, size_t n, uint32_t* d_trace);
/* One table of a proof.  trace: row-major 2^log_height x width Montgomery words (Plonky3's RowMajorMatrix), in host
 * memory or -- on_device = 1 -- in the memory of the context's GPU (left untouched). */
typedef struct {
    const uint32_t* trace;
    uint32_t log_height;           /* 1 .. 24 - blowup_log2; verifier: 0 = read it from the proof, else the height required */
    uint32_t width;                /* = the AIR's */
    const rk_air* air;
    const uint32_t* public_values; /* host, Montgomery words */
    uint32_t n_public;             /* = the AIR's */
    uint32_t on_device;
} rk_p3_table;
/* p3-uni-stark `prove` under the context's parameter set (rk_set_params: field, coset shift = Val::generator(), Poseidon2
 * instance, blowup_log2 = FriConfig::log_blowup, queries, pow_bits; fri_fold_log2 / fri_min_degree are not used -- the
 * PCS folds by two down to a constant).  Every table's quotient degree must fit the blow-up.  Transcript:
 * observe(init_words) -- whatever binds the statement: SP1 observes the verifying key and pc_start there --,
 * observe(trace root), observe(public values of every table), [lookups: permutation alpha, beta, observe(permutation
 * root), observe(cumulative sums)], alpha, observe(quotient root), zeta, then the PCS's own challenges.
 * Proof = u32 words (field elements as Montgomery words):
 *   n_tables | log_height per table | trace root 8 | [permutation root 8 | cumulative sum 4 per table with lookups] |
 *   quotient root 8 |
 *   per table: opened trace row at zeta (4 words per column), at zeta * g, [the permutation trace's, likewise],
 *   2^log_quotient_degree chunks x 4 x 4 |
 *   n_rounds | n_rounds x 8 commit-phase roots | final polynomial 4 | proof-of-work witness (canonical integer) |
 *   per query: the trace batch (every table's LDE row, then the Merkle path), [the permutation batch], the quotient
 *   batch (every chunk's row, then the path), then per FRI round the sibling value (4) and its path.
 * RK_ERR_CAPACITY with *proof_words = the exact size when the buffer is too small.  A trace that breaks its AIR still
 * yields a proof (as in Plonky3's release builds); rk_p3_verify rejects it with reason 3. */
int rk_p3_prove(rk_ctx* ctx, const rk_p3_table* tables, uint32_t n_tables, const uint32_t* init_words, size_t n_init,
                uint32_t* h_proof, size_t capacity_words, size_t* proof_words);
/* p3-uni-stark `verify` on the host (no GPU): params NULL = the SP1 preset; trace / on_device of the tables are ignored;
 * log_height = 0 takes the table's height from the proof, any other value PINS it (the proof must carry that height: what
 * a statement with a fixed-size table -- all 2^16 values of a range check -- needs).  0 = accepted, RK_ERR_INVALID for malformed arguments, otherwise a reason: 1 malformed proof (short,
 * trailing or non-canonical words), 2 shape mismatch, 3 constraint identity (OodEvaluationMismatch), 4 proof of work,
 * 5 input opening, 6 commit-phase opening, 7 final polynomial, 8 the cumulative sums of the lookups do not cancel. */
int rk_p3_verify(const rk_params* params, const rk_p3_table* tables, uint32_t n_tables, const uint32_t* init_words, size_t n_init,
                 const uint32_t* proof, size_t proof_words);
/* rk_p3_verify that also hands back EVERY Poseidon2 permutation the check performed -- the transcript's, the leaf sponges
 * over the opened rows, the Merkle compressions, the proof of work's -- as input states (p2_width Montgomery words each,
 * in the order they happened): what the Poseidon2 chip of a recursion / compress layer has to prove for this proof
 * (rk_p2_chip_trace turns them into that chip's rows).  Returns the verifier's verdict (0 accepted, 1..8 the reason a
 * sequential check gives; the log then ends where the check stopped) or a negative status; RK_ERR_CAPACITY with
 * *n_permutations set when `states` holds fewer than that many: call again. */
int rk_p3_verify_hashes(const rk_params* params, const rk_p3_table* tables, uint32_t n_tables, const uint32_t* init_words, size_t n_init,
                        const uint32_t* proof, size_t proof_words, uint32_t* states, size_t capacity_permutations, size_t* n_permutations);
/* exact proof size for the tables' shapes (log_height, width, air); 0 for shapes rk_p3_prove rejects */
size_t rk_p3_proof_bound_words(const rk_params* params, const rk_p3_table* tables, uint32_t n_tables);
/* Many independent proofs -- the shards of one SP1 execution -- with `batch` of them in flight per GPU: what SP1's
 * SHARD_BATCH_SIZE bounds (docs/README_Sp1.md:27-32; shards are proven from one work queue, the latency-bound parts of
 * one proof -- transcript round trips, the small FRI layers -- under the throughput-bound parts of the others).  One
 * worker thread and prover context per slot, kept per device for the life of the process like rk_prove_session's
 * (rk_session_release frees them too); proof i lands in shards[i].h_proof.  verify != 0: every proof is checked with
 * rk_p3_verify on host threads of their own while the GPU goes on.  Returns RK_OK or the first failure (RK_ERR_VERIFY for a
 * proof that does not verify) with the shard's index in *failed_index. */
typedef struct {
    const rk_p3_table* tables;
    uint32_t n_tables;
    const uint32_t* init_words;
    size_t n_init;
    uint32_t* h_proof;
    size_t capacity_words;
    size_t proof_words;            /* out */
} rk_p3_shard;
typedef struct {
    int device;                    /* the GPU, when n_devices == 0 */
    int batch;                     /* proofs in flight per GPU, 1..16 (SHARD_BATCH_SIZE) */
    int verify;
    const int* devices;            /* optional list of distinct GPUs sharing the work queue */
    int n_devices;
    const rk_params* params;       /* NULL = the SP1 preset */
} rk_p3_session_opts;
int rk_p3_prove_shards(const rk_p3_session_opts* opts, rk_p3_shard* shards, size_t n, size_t* failed_index);
/* wall-clock per stage of the last rk_p3_prove on this ctx, milliseconds (the stream is drained at every boundary) */
typedef struct { float lde, commit, quotient, open, fri, query, total, perm /* permutation traces + their LDE */; } rk_p3_timing;
int rk_p3_last_timing(rk_ctx* ctx, rk_p3_timing* out);

/* per-stage device time of the last rk_prove_segment on this ctx, milliseconds (hipEvent) */
typedef struct { float ntt, hash, deep, fri, query, total, circuit /* time inside rk_circuit_hooks */; } rk_timing;
int rk_last_timing(rk_ctx* ctx, rk_timing* out);

/* ---- per-kernel-class timing (for bench.py's roofline object) ----
 * When enabled every launch of a class is bracketed by a hipEvent pair on the ctx stream;
 * rk_kernel_stats synchronises the stream and returns launches, total device milliseconds and
 * total ALGORITHMIC bytes (what the launch must read + write once) since the last reset. */
typedef enum {
    RK_KCLASS_HASH_ROWS = 0,   /* hash_rows_kernel */
    RK_KCLASS_HASH_FOLD = 1,   /* hash_fold_kernel */
    RK_KCLASS_NTT_PASS = 2,    /* ntt_pass_kernel<fwd|rev> */
    RK_KCLASS_BIT_REVERSE = 3, /* bit_reverse_kernel */
    RK_KCLASS_POLY = 4,        /* mix / eval / divide / fold / sum kernels */
    RK_KCLASS_COUNT = 5
} rk_kclass;
typedef struct { uint64_t launches; double ms; double bytes; } rk_kernel_stat;
int rk_set_kernel_timing(rk_ctx* ctx, int enabled);   /* also resets the counters */
int rk_kernel_stats(rk_ctx* ctx, int kclass, rk_kernel_stat* out);
const char* rk_kernel_class_name(int kclass);
/* the same for the prover contexts rk_prove_session keeps for `device`: enable / reset, then the
 * totals over all of them (what bench.py reads after timing the drop-in entry point) */
int rk_session_set_kernel_timing(int device, int enabled);
int rk_session_kernel_stats(int device, int kclass, rk_kernel_stat* out);

#ifdef __cplusplus
}
#endif
#endif /* RAIKO_HIP_H */
