/* Plain-C caller of the constraint-list API (rk_program): a circuit defined by data only -- no kernel, no
 * callback -- proven and verified through libraiko_hip.so.  This is what a host does with a circuit's
 * `PolyExtStepDef` (risc0-zkp adapter.rs; the Rust crate converts it in provers/hip/driver/src/lib.rs
 * circuit::program): create the program once, point rk_circuit_hooks.program and
 * rk_verify_opts.program at it.
 *
 *   gcc -O2 -I include examples/program_demo.c -o program_demo -L raiko_amd -lraiko_hip -Wl,-rpath,$PWD/raiko_amd
 *   ./program_demo [po2]
 *
 * The circuit: data column d0 is boolean and d1 is its running XOR with the previous row's d1 once the
 * first row is past (code column c0 = 1 on row 0 only):
 *   K0 = d0 * (d0 - 1)
 *   K1 = (1 - c0) * (d1 - (d0 + d1[-1] - 2 * d0 * d1[-1]))
 * written as six / fourteen steps below.  Exit code 0: a valid witness proves and verifies including the
 * constraint identity, a witness with one wrong cell proves (the commitments are consistent) but fails
 * the identity with reason code 70. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "raiko_hip.h"

#define P 2013265921u
static uint32_t to_mont(uint32_t x) { return (uint32_t)(((uint64_t)x << 32) % P); }

static uint64_t rng_state = 0x243F6A8885A308D3ull;
static uint32_t next_u32(void) {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 20);
}

enum { W_ACCUM = 1, W_CODE = 1, W_DATA = 2, N_REGS = 4 };

int main(int argc, char** argv) {
    unsigned po2 = argc > 1 ? (unsigned)atoi(argv[1]) : 10;
    size_t rows = (size_t)1 << po2, dom = rows * 4;

    /* tap set: accum a0 {0}, code c0 {0}, data d0 {0}, d1 {0, 1}; combos sorted: {0} -> 0, {0,1} -> 1 */
    static uint32_t combo_off[3] = {0, 1, 3}, combo_backs[3] = {0, 0, 1};
    static uint32_t reg_group[N_REGS] = {0, 1, 2, 2}, reg_offset[N_REGS] = {0, 0, 0, 1}, reg_combo[N_REGS] = {0, 0, 0, 1};
    rk_taps taps;
    memset(&taps, 0, sizeof taps);
    taps.group_size[0] = W_ACCUM; taps.group_size[1] = W_CODE; taps.group_size[2] = W_DATA;
    taps.n_regs = N_REGS; taps.reg_group = reg_group; taps.reg_offset = reg_offset; taps.reg_combo = reg_combo;
    taps.n_combos = 2; taps.combo_off = combo_off; taps.combo_backs = combo_backs;
    /* eval_u order: a0@0 | c0@0 | d0@0 | d1@0, d1@1  ->  tap numbers 0 | 1 | 2 | 3, 4 */
    enum { T_C0 = 1, T_D0 = 2, T_D1 = 3, T_D1_PREV = 4 };
    static const rk_poly_step steps[] = {
        /* field values (numbered in push order) */
        {RK_STEP_CONST, 1, 0, 0},          /* f0 = 1 */
        {RK_STEP_CONST, 2, 0, 0},          /* f1 = 2 */
        {RK_STEP_GET, T_C0, 0, 0},         /* f2 = c0 */
        {RK_STEP_GET, T_D0, 0, 0},         /* f3 = d0 */
        {RK_STEP_GET, T_D1, 0, 0},         /* f4 = d1 */
        {RK_STEP_GET, T_D1_PREV, 0, 0},    /* f5 = d1[-1] */
        {RK_STEP_SUB, 3, 0, 0},            /* f6 = d0 - 1 */
        {RK_STEP_MUL, 3, 6, 0},            /* f7 = d0 (d0 - 1)                 K0 */
        {RK_STEP_MUL, 3, 5, 0},            /* f8 = d0 d1[-1] */
        {RK_STEP_MUL, 1, 8, 0},            /* f9 = 2 d0 d1[-1] */
        {RK_STEP_ADD, 3, 5, 0},            /* f10 = d0 + d1[-1] */
        {RK_STEP_SUB, 10, 9, 0},           /* f11 = xor */
        {RK_STEP_SUB, 4, 11, 0},           /* f12 = d1 - xor */
        {RK_STEP_SUB, 0, 2, 0},            /* f13 = 1 - c0 */
        /* mix states */
        {RK_STEP_TRUE, 0, 0, 0},           /* m0 */
        {RK_STEP_AND_EQZ, 0, 7, 0},        /* m1 = m0 and K0 = 0 */
        {RK_STEP_TRUE, 0, 0, 0},           /* m2 */
        {RK_STEP_AND_EQZ, 2, 12, 0},       /* m3 = [d1 - xor = 0] */
        {RK_STEP_AND_COND, 1, 13, 3},      /* m4 = m1 and (1 - c0) * m3 */
    };
    rk_program* prog = NULL;
    int st = rk_program_create(steps, sizeof steps / sizeof steps[0], 4, &taps, &prog);
    if (st != RK_OK) { fprintf(stderr, "rk_program_create: %s\n", rk_strerror(st)); return 1; }
    rk_program_info info;
    rk_program_get_info(prog, &info);
    printf("program: %llu steps, %llu ops after compilation, %u field slots, %u mix slots, %u powers of poly_mix\n",
           (unsigned long long)info.n_steps, (unsigned long long)info.n_ops, info.n_fp_slots, info.n_mix_slots, info.n_mix_powers);

    rk_ctx* ctx = NULL;
    st = rk_ctx_create(0, NULL, &ctx);
    if (st != RK_OK) { fprintf(stderr, "rk_ctx_create: %s\n", rk_strerror(st)); rk_program_destroy(prog); return 1; }

    uint32_t* accum = malloc(rows * 4);
    uint32_t* code = calloc(rows, 4);
    uint32_t* data = malloc(2 * rows * 4);
    uint32_t globals[4];
    for (int i = 0; i < 4; i++) globals[i] = next_u32() % P;
    for (size_t i = 0; i < rows; i++) accum[i] = next_u32() % P;   /* unconstrained */
    code[0] = to_mont(1);
    uint32_t acc = 0;
    for (size_t i = 0; i < rows; i++) {
        uint32_t bit = next_u32() & 1;
        acc = i == 0 ? (next_u32() & 1) : (acc ^ bit);            /* row 0 is free (c0 = 1 switches K1 off there) */
        data[i] = to_mont(bit);
        data[rows + i] = to_mont(acc);
    }
    (void)dom;

    rk_circuit_hooks hooks;
    memset(&hooks, 0, sizeof hooks);
    hooks.program = prog;                                        /* eval_check: the library, from the list */
    rk_verify_opts vopts;
    memset(&vopts, 0, sizeof vopts);
    vopts.program = prog;                                        /* the verifier's constraint identity, same list */

    rk_segment seg;
    memset(&seg, 0, sizeof seg);
    seg.po2 = po2;
    seg.taps = taps;
    seg.group[0] = accum; seg.group[1] = code; seg.group[2] = data;
    seg.globals = globals; seg.n_globals = 4;
    seg.n_accum_mix = 4;
    memcpy(seg.proof_system_info, "RISC0_STARK:v1__", 16);
    memcpy(seg.circuit_info, "XOR_DEMO:v1_____", 16);
    seg.hooks = &hooks;

    size_t cap = rk_seal_bound_words(&seg), words = 0;
    uint32_t* seal = malloc(cap * 4);
    int rc = 1;
    st = rk_prove_segment(ctx, &seg, seal, cap, &words);
    if (st != RK_OK) { fprintf(stderr, "rk_prove_segment: %s (%s)\n", rk_strerror(st), rk_last_error(ctx)); goto done; }
    int v = rk_verify_segment_ex(&seg, &vopts, seal, words);
    printf("valid witness: seal of %zu words, verifier says %d\n", words, v);
    if (v != 0) goto done;

    /* one wrong cell: d1 of a middle row */
    size_t bad_row = rows / 2;
    data[rows + bad_row] = to_mont(1) == data[rows + bad_row] ? 0 : to_mont(1);
    st = rk_prove_segment(ctx, &seg, seal, cap, &words);
    if (st != RK_OK) { fprintf(stderr, "rk_prove_segment (bad witness): %s\n", rk_strerror(st)); goto done; }
    int v_plain = rk_verify_segment(&seg, seal, words);           /* commitments, DEEP, FRI: consistent */
    int v_id = rk_verify_segment_ex(&seg, &vopts, seal, words);   /* the circuit says no */
    printf("broken witness: without the identity %d, with it %d\n", v_plain, v_id);
    rc = (v_plain == 0 && v_id == 70) ? 0 : 1;
done:
    free(seal); free(accum); free(code); free(data);
    rk_ctx_destroy(ctx);
    rk_program_destroy(prog);
    return rc;
}
