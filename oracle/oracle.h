/* TEST INFRASTRUCTURE -- CPU oracle for the segment-proof hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (raiko_amd/, libraiko_hip.so) never does.
 *
 * PARITY UNPINNED (byte level).  The path is `session.prove()` called at
 * /root/reference provers/risc0/driver/src/bonsai.rs:271; the algorithm lives in
 * the crates.io dependency risc0-zkp 1.0.1 (+ risc0-core 1.0.1,
 * risc0-circuit-rv32im 1.0.1; reference Cargo.lock:7243,:7171,:7129) whose
 * source is absent from /root/reference, and the reference holds no golden
 * vector for it.  Every function below restates the
 * published algorithm of the named risc0-zkp module and is pinned by
 * algebraic known-answer tests (tests/test_oracle_*.py): exact big-int field
 * arithmetic, O(n^2) DFT, iNTT(NTT)=id, Merkle openings that verify, FRI folds
 * equal to direct evaluation, and a verifier that accepts the produced seal.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include "or_field.h"

#ifdef __cplusplus
extern "C" {
#endif

/* protocol shape: risc0's values (4 / 16 / 256 / no proof of work) unless or_set_params says otherwise */
#define OR_INV_RATE ((size_t)1 << g_or.blowup_log2)
#define OR_INV_RATE_PO2 (g_or.blowup_log2)
#define OR_QUERIES (g_or.queries)
#define OR_MAX_QUERIES 256
#define OR_FRI_FOLD ((size_t)1 << g_or.fri_fold_log2)
#define OR_FRI_FOLD_PO2 (g_or.fri_fold_log2)
#define OR_FRI_MIN_DEGREE ((size_t)g_or.fri_min_degree)
#define OR_MAX_FRI_FOLD 16
#define OR_MAX_CHECK_SIZE 64
#define OR_DIGEST_WORDS 8
#define OR_MAX_CELLS 24
/* sponge width / rate of the configured Poseidon2 instance (24 / 16 by default) */
#define OR_CELLS (g_or.p2_width)
#define OR_CELLS_RATE (g_or.p2_width - 8)
#define OR_CELLS_OUT 8
#define OR_EXT 4
#define OR_CHECK_SIZE ((size_t)OR_EXT << g_or.blowup_log2)

/* ---- the parameter set (the oracle is configured process-wide, one set at a time) ----
 * Mirrors rk_params of include/raiko_hip.h: canonical field values, Montgomery Poseidon2 tables.
 * Presets: 0 = risc0 (x^4+11, 137, shift 3, Poseidon2 t=24 / M4 of the paper / zero-padded sponge,
 * 50 queries), 1 = SP1 / Plonky3 shape (x^4-11, 0x1a427a41, shift 31, t=16 / circ(2,3,1,1) /
 * padding-free sponge, 100 queries, fold 2).  RECALLED values, parity unpinned like the rest. */
typedef struct {
    uint32_t ext_w, root_2_27, coset_shift;
    uint32_t p2_width, p2_m4, p2_pad_free;
    const fp *p2_rc_ext, *p2_rc_int, *p2_diag;   /* NULL = the derived defaults of the width */
    uint32_t queries, blowup_log2, fri_fold_log2, fri_min_degree, pow_bits;
} or_params;
extern or_params g_or;
void or_params_preset(or_params* out, int preset);
int or_set_params(const or_params* p);   /* 0, or -1 for an unusable set (left unchanged) */

/* ---- field helpers exported for ctypes tests ---- */
uint32_t or_fp_mul(uint32_t a, uint32_t b);
uint32_t or_fp_add(uint32_t a, uint32_t b);
uint32_t or_fp_sub(uint32_t a, uint32_t b);
uint32_t or_fp_inv(uint32_t a);
uint32_t or_fp_encode(uint32_t canon);
uint32_t or_fp_decode(uint32_t mont);
void or_fp4_mul(const uint32_t* a, const uint32_t* b, uint32_t* out);
void or_fp4_inv(const uint32_t* a, uint32_t* out);
uint32_t or_rou_fwd(unsigned k); /* Montgomery form of 137^(2^(27-k)) */
uint32_t or_rou_rev(unsigned k);

/* ---- risc0-zkp core/ntt.rs ---- */
void or_bit_reverse(fp* io, size_t n);
void or_interpolate_ntt(fp* io, size_t n);                   /* natural evals -> bit-reversed coeffs */
void or_evaluate_ntt(fp* io, size_t n, unsigned expand_bits);/* bit-reversed coeffs -> natural evals */

/* ---- risc0-zkp core/hash/poseidon2 ---- */
void or_poseidon2_mix(fp* cells /*24*/);
void or_hash_elem_slice(const fp* in, size_t n, size_t stride, uint32_t* digest /*8*/);
void or_hash_pair(const uint32_t* a, const uint32_t* b, uint32_t* out);

/* ---- risc0-zkp hal/cpu.rs: the Hal trait ---- */
void or_batch_interpolate_ntt(fp* io, size_t size, size_t count);
void or_batch_evaluate_ntt(fp* io, size_t size, size_t count, unsigned expand_bits);
void or_zk_shift(fp* io, size_t size, size_t count);
void or_batch_expand_into_evaluate_ntt(fp* out, const fp* in, size_t in_size, size_t count, unsigned expand_bits);
void or_batch_bit_reverse(fp* io, size_t size, size_t count);
void or_hash_rows(uint32_t* out, const fp* matrix, size_t rows, size_t cols);
void or_hash_fold(uint32_t* nodes, size_t input_size, size_t output_size);
void or_batch_evaluate_any(const fp* coeffs, size_t size, const uint32_t* which, const fp4* xs,
                           size_t eval_count, fp4* out);
void or_mix_poly_coeffs(fp4* out, const uint32_t* mix_start, const uint32_t* mix, const fp* in,
                        const uint32_t* combos, size_t input_size, size_t count);
void or_eltwise_add_elem(fp* out, const fp* a, const fp* b, size_t n);
void or_eltwise_sum_extelem(fp* out, const fp4* in, size_t count, size_t to_add);
void or_eltwise_copy_elem(fp* out, const fp* in, size_t n);
void or_eltwise_zeroize_elem(fp* io, size_t n);
void or_fri_fold(fp* out, const fp* in, size_t out_count, const uint32_t* mix); /* by 2^g_or.fri_fold_log2 */
void or_fri_fold_evals(fp4* out, const fp4* in, size_t n_out, const uint32_t* beta); /* Plonky3's, on evaluations */
void or_gather_sample(fp* dst, const fp* src, size_t idx, size_t size, size_t stride);
void or_prefix_products(fp4* io, size_t count);              /* io[i] *= io[i-1], sequential */
void or_scatter(fp* into, const uint32_t* index, size_t n_cycles, const uint32_t* offsets, const fp* values);

/* ---- Plonky3 two-adic FRI PCS, data-parallel steps (or_pcs.c; row-major matrices) ---- */
void or_pcs_coset_lde_rows(fp* out, const fp* in, size_t h, size_t w);
void or_pcs_eval_at(fp4* out, const fp* lde, size_t H, size_t w, const uint32_t* z);
void or_pcs_reduce_openings(fp4* ro, const fp* lde, size_t H, size_t w, size_t n_points, const uint32_t* points,
                            const uint32_t* ys, const uint32_t* alpha, uint64_t alpha_offset);

uint32_t or_duplex_grind(const fp* state, const fp* input, size_t n_input, unsigned bits);

/* ---- risc0-zkp core/poly.rs ---- */
void or_poly_interpolate(fp4* out, const fp4* x, const fp4* fx, size_t n);
void or_poly_divide(fp4* p, size_t n, const uint32_t* z, uint32_t* remainder);
void or_poly_eval(const fp4* coeffs, size_t n, const uint32_t* x, uint32_t* out);

/* ---- risc0-zkp prove/merkle.rs ---- */
typedef struct {
    size_t rows, cols, queries, layers, top_layer, top_size;
    uint32_t* nodes;   /* 2*rows digests, heap order, nodes[1] = root */
    const fp* matrix;  /* column-major rows x cols */
} or_merkle;
void or_merkle_build(or_merkle* m, const fp* matrix, size_t rows, size_t cols, size_t queries);
void or_merkle_free(or_merkle* m);

/* ---- risc0-zkp prove/write_iop.rs + hash/poseidon2/rng.rs ---- */
typedef struct {
    uint32_t* proof; size_t len, cap;
    fp cells[OR_MAX_CELLS]; size_t pool_used;
} or_iop;
void or_iop_init(or_iop* iop);
void or_iop_free(or_iop* iop);
void or_iop_commit(or_iop* iop, const uint32_t* digest);
void or_iop_write(or_iop* iop, const uint32_t* words, size_t n);
uint32_t or_iop_random_bits(or_iop* iop, unsigned bits);
fp or_iop_random_elem(or_iop* iop);
fp4 or_iop_random_ext(or_iop* iop);

/* proof of work on the transcript (Plonky3 `grind`, restated on this transcript; risc0 has none):
 * the smallest nonce w such that after absorbing hash([w]) the next random_bits(bits) are zero */
uint32_t or_pow_grind(const or_iop* iop, unsigned bits);

/* ---- tap set (risc0-zkp taps.rs), supplied by the caller ---- */
typedef struct {
    uint32_t group_size[3];        /* columns per group: 0 accum, 1 code, 2 data */
    uint32_t n_regs;               /* sorted by (group, offset) */
    const uint32_t* reg_group;
    const uint32_t* reg_offset;
    const uint32_t* reg_combo;
    uint32_t n_combos;
    const uint32_t* combo_off;     /* n_combos+1 prefix offsets into combo_backs */
    const uint32_t* combo_backs;
} or_taps;

/* CircuitHal::accumulate / eval_check (risc0-circuit-rv32im prove/mod.rs, risc0-zkp prove/prover.rs
 * finalize) as host callbacks: all pointers are host memory.  trace = the witness (N rows), lde =
 * PolyGroup::evaluated (4N rows, natural order), both column-major and indexed by group id. */
typedef struct {
    uint32_t po2;
    uint32_t group_size[3];
    const fp* trace[3];
    const fp* lde[3];
    const fp* globals; uint32_t n_globals;
    const fp* mix; uint32_t n_mix;
} or_circuit_view;
typedef struct {
    void* user;
    int (*accumulate)(void* user, const or_circuit_view* v, fp* accum /* N x group_size[0] */);
    int (*eval_check)(void* user, const or_circuit_view* v, const fp* poly_mix /*4*/, fp* check /* 4 x 4N */);
} or_circuit_hooks;

typedef struct {
    uint32_t po2;
    or_taps taps;
    const fp* group[3];            /* trace evaluations, column-major N x group_size, by group id */
    const fp* check;               /* stand-in for eval_check output: 4 x 4N evaluations */
    const fp* globals; uint32_t n_globals;
    uint32_t n_accum_mix;          /* elements drawn before the accum commit */
    uint8_t proof_system_info[16];
    uint8_t circuit_info[16];
    const or_circuit_hooks* hooks; /* NULL: group[0] / check are taken as given */
} or_segment;
/* CircuitDef::poly_ext for the verifier: the mixed constraint polynomial on the tap openings */
typedef int (*or_poly_ext_fn)(void* user, const or_segment* pub, const fp* poly_mix, const fp4* eval_u, size_t n_taps,
                              const fp* mix, uint32_t n_mix, fp* out /*4*/);

/* risc0-zkp prove/prover.rs + circuit/rv32im prove/mod.rs (prove_segment):
 * returns malloc'd seal (u32 transcript). threads>0 sets the OpenMP team size. */
int or_prove_segment(const or_segment* seg, uint32_t** seal, size_t* seal_words, int threads);
/* risc0-zkp verify/mod.rs restated for the same flow, minus the circuit's
 * constraint identity (no rv32im circuit available): 0 = accept. */
int or_verify_segment(const or_segment* pub_only, const uint32_t* seal, size_t seal_words);
/* the same plus the constraint identity poly_ext(...) == check(z) * ((3z)^N - 1) (rc 70) */
int or_verify_segment_circuit(const or_segment* pub_only, const uint32_t* seal, size_t seal_words,
                              or_poly_ext_fn poly_ext, void* user);

/* ---- toy circuit (oracle/or_toy.c): CPU restatement of examples/toy_circuit ---- */
const or_circuit_hooks* or_toy_hooks(void);
int or_toy_poly_ext(void* user, const or_segment* pub, const fp* poly_mix, const fp4* eval_u, size_t n_taps,
                    const fp* mix, uint32_t n_mix, fp* out);
/* ---- the constraint polynomial as a step list (oracle/or_program.c): literal interpretation of
 * risc0-zkp adapter.rs PolyExtStepDef, the checker of raiko_amd/csrc/circuit_program.hip ---- */
typedef struct { uint32_t op, a, b, c; } or_step;
typedef struct { const or_step* steps; size_t n_steps; uint32_t ret; const or_taps* taps; } or_program;
/* or_circuit_hooks.eval_check / or_poly_ext_fn with user = an or_program */
int or_program_eval_check(void* user, const or_circuit_view* v, const fp* poly_mix, fp* check);
int or_program_poly_ext(void* user, const or_segment* pub, const fp* poly_mix, const fp4* eval_u, size_t n_taps,
                        const fp* mix, uint32_t n_mix, fp* out);
/* ---- Plonky3 MerkleTreeMmcs restated (oracle/or_mmcs.c): the checker of rk_mmcs_commit / open / verify ---- */
typedef struct { const fp* values; uint32_t height, width, row_major; } or_matrix;
void or_mmcs_commit(const or_matrix* mats, uint32_t n, uint32_t* nodes /* 2 * H digests, heap order */);
int or_mmcs_verify(const uint32_t* heights, const uint32_t* widths, uint32_t n, uint32_t index, const fp* rows, const uint32_t* path,
                   const uint32_t* root);
/* ---- Plonky3 uni-stark over the two-adic FRI PCS, one or several tables (oracle/or_p3.c): the checker of
 * rk_p3_prove / rk_p3_verify.  An AIR is a step list: every step but ASSERT_ZERO pushes one value; a, b name
 * earlier values (CONST: a = canonical integer; LOCAL / NEXT: a = column; PUBLIC: a = index). ---- */
enum { OR_AIR_CONST = 0, OR_AIR_LOCAL = 1, OR_AIR_NEXT = 2, OR_AIR_PUBLIC = 3, OR_AIR_IS_FIRST_ROW = 4, OR_AIR_IS_LAST_ROW = 5,
       OR_AIR_IS_TRANSITION = 6, OR_AIR_ADD = 7, OR_AIR_SUB = 8, OR_AIR_MUL = 9, OR_AIR_NEG = 10, OR_AIR_ASSERT_ZERO = 11,
       /* the permutation (LogUp) trace of a table with interactions: base column a of its flattened extension columns on the
        * current / next row; base component a of the challenge vector [alpha | beta^0 | beta^1 | ...]; component a of the
        * table's cumulative sum */
       OR_AIR_PERM_LOCAL = 12, OR_AIR_PERM_NEXT = 13, OR_AIR_CHALLENGE = 14, OR_AIR_CUMSUM = 15 };
typedef struct { uint32_t op, a, b; } or_air_step;
/* one interaction of a table with a bus (sp1-core lookup/interaction.rs, RECALLED): the tuple (bus, values...) is sent
 * (kind 0) or received (kind 1) `multiplicity` times on every row; values are main-trace columns, the multiplicity a
 * column or a constant */
typedef struct { uint32_t kind, bus, mult_is_const, mult, n_values; const uint32_t* value_cols; } or_interaction;
typedef struct { const or_air_step* steps; size_t n_steps; const or_interaction* interactions; uint32_t n_interactions; } or_air;
typedef struct {
    const fp* trace;               /* row-major 2^log_height x width (prover only) */
    uint32_t log_height, width;    /* the verifier takes log_height from the proof when 0, else requires it */
    const or_air* air;
    const fp* public_values; uint32_t n_public;
} or_p3_table;
int or_air_log_quotient_degree(const or_air* air);
/* 0 and a malloc'd proof (or_free), or < 0: -1 malformed input, -2 quotient degree above the blow-up,
 * -3 the witness does not satisfy the AIR (the folded quotient is not low-degree), -4 proof of work */
int or_p3_prove(const or_p3_table* tables, uint32_t n_tables, const fp* init, size_t n_init, uint32_t** proof, size_t* words);
int or_p3_verify(const or_p3_table* tables, uint32_t n_tables, const fp* init, size_t n_init, const uint32_t* proof, size_t words);
void or_free(void* p);
int or_max_threads(void);
void or_set_threads(int n);
/* or_fast.c: route the hot operators (NTT, zk_shift, Poseidon2 rows/folds, tap evaluation, DEEP mix)
 * through their AVX2 / table-driven forms -- the timed cpu_baseline; results are bit-identical */
void or_set_fast(int on);
int or_get_fast(void);

/* stage timing of the last or_prove_segment call, seconds */
typedef struct { double ntt, hash, deep, fri, query, total; } or_timing;
void or_last_timing(or_timing* t);

#ifdef __cplusplus
}
#endif
#endif
