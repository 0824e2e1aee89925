/* Plain-C caller of rk_air_* / rk_p3_prove / rk_p3_verify: Plonky3's own uni-stark test -- the Fibonacci AIR
 * (columns left, right; public values a, b, x; first row = (a, b), next.left = right, next.right = left + right,
 * last row's right = x) -- handed over as a step list, proven on the GPU under SP1's parameter set and checked by the
 * host verifier; then the same with a wrong public value (the verifier refuses: reason 3, the constraint identity) and
 * with a changed proof word.  What a host on the SP1 side of raiko (provers/sp1/driver/src/lib.rs:44-57) would call
 * per shard, with its chips' AIRs in place of this one.
 *
 *   gcc -O2 -I include examples/p3_demo.c -o p3_demo -L raiko_amd -lraiko_hip -Wl,-rpath,$PWD/raiko_amd
 *   ./p3_demo [log2 rows]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "raiko_hip.h"

#define P 2013265921ull
static uint32_t mont(uint64_t x) { return (uint32_t)((x % P) * ((1ull << 32) % P) % P); }   /* canonical -> Montgomery word */

#define CHECK(call)                                                                           \
    do {                                                                                      \
        int st_ = (call);                                                                     \
        if (st_ != RK_OK) {                                                                   \
            fprintf(stderr, "%s -> %s (%s)\n", #call, rk_strerror(st_), ctx ? rk_last_error(ctx) : ""); \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

int main(int argc, char** argv) {
    const unsigned k = argc > 1 ? (unsigned)atoi(argv[1]) : 10;
    rk_ctx* ctx = NULL;
    /* the AIR as `Air::eval` leaves it in a symbolic builder: values are numbered in push order */
    enum { L = 0, R, NL, NR, A, B, X, FIRST, TRANS, LAST };   /* the first ten values: leaves */
    const rk_air_step steps[] = {
        {RK_AIR_LOCAL, 0, 0}, {RK_AIR_LOCAL, 1, 0}, {RK_AIR_NEXT, 0, 0}, {RK_AIR_NEXT, 1, 0},
        {RK_AIR_PUBLIC, 0, 0}, {RK_AIR_PUBLIC, 1, 0}, {RK_AIR_PUBLIC, 2, 0},
        {RK_AIR_IS_FIRST_ROW, 0, 0}, {RK_AIR_IS_TRANSITION, 0, 0}, {RK_AIR_IS_LAST_ROW, 0, 0},
        {RK_AIR_SUB, L, A}, {RK_AIR_MUL, FIRST, 10}, {RK_AIR_ASSERT_ZERO, 11, 0},        /* first row: left = a      */
        {RK_AIR_SUB, R, B}, {RK_AIR_MUL, FIRST, 12}, {RK_AIR_ASSERT_ZERO, 13, 0},        /* first row: right = b     */
        {RK_AIR_SUB, NL, R}, {RK_AIR_MUL, TRANS, 14}, {RK_AIR_ASSERT_ZERO, 15, 0},       /* next.left = right        */
        {RK_AIR_ADD, L, R}, {RK_AIR_SUB, NR, 16}, {RK_AIR_MUL, TRANS, 17}, {RK_AIR_ASSERT_ZERO, 18, 0},   /* next.right = l + r */
        {RK_AIR_SUB, R, X}, {RK_AIR_MUL, LAST, 19}, {RK_AIR_ASSERT_ZERO, 20, 0},         /* last row: right = x      */
    };
    rk_air* air = NULL;
    int st = rk_air_create(steps, sizeof steps / sizeof steps[0], 2, 3, &air);
    if (st != RK_OK) {
        fprintf(stderr, "rk_air_create -> %s\n", rk_strerror(st));
        return 1;
    }
    rk_air_info info;
    rk_air_get_info(air, &info);
    printf("fibonacci AIR: %u constraints, degree %u, %u quotient chunk(s), %llu ops per point\n", info.n_constraints, info.max_degree,
           1u << info.log_quotient_degree, (unsigned long long)info.n_ops);

    int n_dev = 0;
    if (rk_device_count(&n_dev) != RK_OK || n_dev < 1) {
        fprintf(stderr, "no usable GPU\n");
        return 1;
    }
    CHECK(rk_ctx_create(0, NULL, &ctx));
    rk_params par;
    CHECK(rk_params_preset(&par, RK_PRESET_SP1));
    CHECK(rk_set_params(ctx, &par));

    const size_t n = (size_t)1 << k;
    uint32_t* trace = (uint32_t*)malloc(n * 2 * 4);
    uint64_t l = 0, r = 1;
    for (size_t i = 0; i < n; i++) {
        trace[2 * i] = mont(l);
        trace[2 * i + 1] = mont(r);
        const uint64_t t = (l + r) % P;
        l = r;
        r = t;
    }
    uint32_t pub[3] = {mont(0), mont(1), trace[2 * (n - 1) + 1]};
    const uint32_t statement[2] = {mont(167009), mont(10)};   /* whatever binds the proof to its context: observed first */
    rk_p3_table table;
    memset(&table, 0, sizeof table);
    table.trace = trace;
    table.log_height = k;
    table.width = 2;
    table.air = air;
    table.public_values = pub;
    table.n_public = 3;
    const size_t cap = rk_p3_proof_bound_words(&par, &table, 1);
    uint32_t* proof = (uint32_t*)malloc(cap * 4);
    size_t words = 0;
    CHECK(rk_p3_prove(ctx, &table, 1, statement, 2, proof, cap, &words));
    rk_p3_timing tm;
    CHECK(rk_p3_last_timing(ctx, &tm));
    int v = rk_p3_verify(&par, &table, 1, statement, 2, proof, words);
    printf("2^%u rows: proof of %zu words in %.2f ms (lde %.2f commit %.2f quotient %.2f open %.2f fri %.2f query %.2f), verifier: %d\n", k, words,
           tm.total, tm.lde, tm.commit, tm.quotient, tm.open, tm.fri, tm.query, v);
    if (v != 0 || words != cap) return 1;
    pub[2] = mont(12345);   /* a claim about the last row that the trace does not support */
    v = rk_p3_verify(&par, &table, 1, statement, 2, proof, words);
    printf("wrong public value: verifier %d\n", v);
    if (v != 3) return 1;
    pub[2] = trace[2 * (n - 1) + 1];
    proof[words / 2] = (uint32_t)(((uint64_t)proof[words / 2] + 1) % P);
    v = rk_p3_verify(&par, &table, 1, statement, 2, proof, words);
    printf("changed proof word: verifier %d\n", v);
    if (v == 0) return 1;
    printf("fibonacci proven and verified\n");
    rk_air_destroy(air);
    rk_ctx_destroy(ctx);
    free(proof);
    free(trace);
    return 0;
}
