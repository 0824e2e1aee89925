/* Plain-C caller of rk_air_* / rk_p3_prove / rk_p3_verify: Plonky3's own uni-stark test -- the Fibonacci AIR
 * (columns left, right; public values a, b, x; first row = (a, b), next.left = right, next.right = left + right,
 * last row's right = x) -- handed over as a step list, proven on the GPU under SP1's parameter set and checked by the
 * host verifier; then the same with a wrong public value (the verifier refuses: reason 3, the constraint identity) and
 * with a changed proof word.  Then two tables tied by a lookup (rk_air_create_lookup): a user table sends (x, y) pairs, a
 * table of squares (v, v * v, multiplicity) receives them -- the library writes the permutation constraints, fills the
 * permutation traces on the GPU and the verifier checks that the cumulative sums cancel; one multiplicity off by one and
 * it answers 8.  What a host on the SP1 side of raiko (provers/sp1/driver/src/lib.rs:44-57) would call per shard, with
 * its chips' AIRs and interactions in place of these.
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

    /* ---- a lookup between two tables */
    const rk_air_step sq_steps[] = {   /* squares table: v counts up from 0, sq = v * v */
        {RK_AIR_LOCAL, 0, 0}, {RK_AIR_LOCAL, 1, 0}, {RK_AIR_NEXT, 0, 0}, {RK_AIR_IS_FIRST_ROW, 0, 0}, {RK_AIR_IS_TRANSITION, 0, 0},
        {RK_AIR_MUL, 3, 0}, {RK_AIR_ASSERT_ZERO, 5, 0},                                           /* first row: v = 0   */
        {RK_AIR_CONST, 1, 0}, {RK_AIR_ADD, 0, 6}, {RK_AIR_SUB, 2, 7}, {RK_AIR_MUL, 4, 8}, {RK_AIR_ASSERT_ZERO, 9, 0},   /* next v = v + 1 */
        {RK_AIR_MUL, 0, 0}, {RK_AIR_SUB, 1, 10}, {RK_AIR_ASSERT_ZERO, 11, 0},                      /* sq = v * v         */
    };
    /* interactions: kind (0 send / 1 receive), bus, mult_is_const, mult, n_values, columns */
    const uint32_t user_ix[] = {0, 9, 1, 1, 2, 0, 1};     /* every row sends (bus 9: x, y) once          */
    const uint32_t sq_ix[] = {1, 9, 0, 2, 2, 0, 1};       /* receives (bus 9: v, sq) column-2 times      */
    rk_air *user_air = NULL, *sq_air = NULL;
    CHECK(rk_air_create_lookup(NULL, 0, 2, 0, user_ix, 1, 7, par.ext_w, &user_air));   /* no constraints of its own */
    CHECK(rk_air_create_lookup(sq_steps, sizeof sq_steps / sizeof sq_steps[0], 3, 0, sq_ix, 1, 7, par.ext_w, &sq_air));
    enum { LOG_USER = 9, LOG_SQ = 6 };
    uint32_t* user = (uint32_t*)malloc(((size_t)2 << LOG_USER) * 4);
    uint32_t* sq = (uint32_t*)calloc((size_t)3 << LOG_SQ, 4);
    uint32_t count[1 << LOG_SQ] = {0};
    for (size_t i = 0; i < ((size_t)1 << LOG_USER); i++) {
        const uint32_t x = (uint32_t)((i * 2654435761u) >> 7) & ((1u << LOG_SQ) - 1);
        user[2 * i] = mont(x);
        user[2 * i + 1] = mont((uint64_t)x * x);
        count[x]++;
    }
    for (uint32_t vv = 0; vv < (1u << LOG_SQ); vv++) {
        sq[3 * vv] = mont(vv);
        sq[3 * vv + 1] = mont((uint64_t)vv * vv);
        sq[3 * vv + 2] = mont(count[vv]);
    }
    rk_p3_table pair[2];
    memset(pair, 0, sizeof pair);
    pair[0].trace = user, pair[0].log_height = LOG_USER, pair[0].width = 2, pair[0].air = user_air;
    pair[1].trace = sq, pair[1].log_height = LOG_SQ, pair[1].width = 3, pair[1].air = sq_air;
    const size_t cap2 = rk_p3_proof_bound_words(&par, pair, 2);
    uint32_t* proof2 = (uint32_t*)malloc(cap2 * 4);
    CHECK(rk_p3_prove(ctx, pair, 2, statement, 2, proof2, cap2, &words));
    v = rk_p3_verify(&par, pair, 2, statement, 2, proof2, words);
    printf("lookup: 2^%d pairs looked up in a table of 2^%d squares, proof of %zu words, verifier: %d\n", LOG_USER, LOG_SQ, words, v);
    if (v != 0) return 1;
    sq[3 * 5 + 2] = mont(count[5] + 1);   /* the table claims one more lookup of 5 than was made */
    CHECK(rk_p3_prove(ctx, pair, 2, statement, 2, proof2, cap2, &words));
    v = rk_p3_verify(&par, pair, 2, statement, 2, proof2, words);
    printf("one multiplicity off by one: verifier %d\n", v);
    if (v != 8) return 1;
    printf("lookup proven and verified\n");
    rk_air_destroy(user_air);
    rk_air_destroy(sq_air);
    free(proof2);
    free(sq);
    free(user);
    rk_air_destroy(air);
    rk_ctx_destroy(ctx);
    free(proof);
    free(trace);
    return 0;
}
