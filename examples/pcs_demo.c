/* Plain-C caller of the Plonky3 PCS steps (rk_pcs_*, rk_mmcs_*, rk_fri_fold_evals, rk_duplex_grind) under SP1's
 * parameter set: what a `Pcs` implementation on the SP1 side (provers/sp1/driver/src/lib.rs:48-57 -> sp1-core ->
 * p3-fri TwoAdicFriPcs) hands to the GPU, in the order it would -- commit one trace, open it at a point, reduce the
 * rows, fold the reduced opening down.  The check is Plonky3's own assertion at the end of the commit phase: the
 * folds leave `blowup` evaluations of a constant polynomial.
 *
 *   gcc -O2 -I include examples/pcs_demo.c -o pcs_demo -L raiko_amd -lraiko_hip -Wl,-rpath,$PWD/raiko_amd
 *   ./pcs_demo [log2 height] [width]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "raiko_hip.h"

#define P 2013265921u
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t next_elem(void) {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)((rng_state >> 16) % P);   /* any value below p is a valid Montgomery word */
}
#define CHECK(call)                                                                           \
    do {                                                                                      \
        int st_ = (call);                                                                     \
        if (st_ != RK_OK) {                                                                   \
            fprintf(stderr, "%s -> %s (%s)\n", #call, rk_strerror(st_), ctx ? rk_last_error(ctx) : ""); \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

int main(int argc, char** argv) {
    unsigned k = argc > 1 ? (unsigned)atoi(argv[1]) : 12, w = argc > 2 ? (unsigned)atoi(argv[2]) : 6;
    rk_ctx* ctx = NULL;
    int n_dev = 0;
    if (rk_device_count(&n_dev) != RK_OK || n_dev < 1) {
        fprintf(stderr, "no usable GPU\n");
        return 1;
    }
    CHECK(rk_ctx_create(0, NULL, &ctx));
    rk_params par;
    CHECK(rk_params_preset(&par, RK_PRESET_SP1));
    CHECK(rk_set_params(ctx, &par));
    const size_t h = (size_t)1 << k, H = h << par.blowup_log2;

    /* commit: LDE with bit-reversed rows, one tree over it */
    uint32_t* trace = (uint32_t*)malloc(h * w * 4);
    for (size_t i = 0; i < h * w; i++) trace[i] = next_elem();
    void *d_trace, *d_lde, *d_nodes, *d_ro, *d_ys, *d_next;
    CHECK(rk_alloc(ctx, h * w * 4, &d_trace));
    CHECK(rk_alloc(ctx, H * w * 4, &d_lde));
    CHECK(rk_alloc(ctx, 2 * H * 8 * 4, &d_nodes));
    CHECK(rk_h2d(ctx, d_trace, trace, h * w * 4));
    CHECK(rk_pcs_coset_lde_rows(ctx, (uint32_t*)d_lde, (const uint32_t*)d_trace, h, w));
    rk_matrix mat;
    memset(&mat, 0, sizeof mat);
    mat.d_values = (const uint32_t*)d_lde;
    mat.height = (uint32_t)H;
    mat.width = w;
    mat.row_major = 1;
    uint32_t root[8];
    CHECK(rk_mmcs_commit(ctx, &mat, 1, (uint32_t*)d_nodes, root));
    printf("committed a 2^%u x %u trace (LDE 2^%u rows): root %08x %08x ...\n", k, w, k + par.blowup_log2, root[0], root[1]);

    /* open at zeta: the opened values, then the reduced opening over the whole coset */
    uint32_t zeta[4], alpha[4];
    for (int i = 0; i < 4; i++) { zeta[i] = next_elem(); alpha[i] = next_elem(); }
    uint32_t* ys = (uint32_t*)malloc(w * 16);
    CHECK(rk_alloc(ctx, w * 16, &d_ys));
    CHECK(rk_pcs_eval_at(ctx, (uint32_t*)d_ys, (const uint32_t*)d_lde, H, w, zeta));
    CHECK(rk_d2h(ctx, ys, d_ys, w * 16));
    uint32_t* zero = (uint32_t*)calloc(H * 4, 4);
    CHECK(rk_alloc(ctx, H * 16, &d_ro));
    CHECK(rk_h2d(ctx, d_ro, zero, H * 16));
    CHECK(rk_pcs_reduce_openings(ctx, (uint32_t*)d_ro, (const uint32_t*)d_lde, H, w, 1, zeta, ys, alpha, 0));

    /* FRI commit phase without the hashing: fold with "challenges" until `blowup` values are left */
    CHECK(rk_alloc(ctx, H * 8, &d_next));
    void *cur = d_ro, *nxt = d_next;
    size_t n = H;
    while (n > ((size_t)1 << par.blowup_log2)) {
        uint32_t beta[4];
        for (int i = 0; i < 4; i++) beta[i] = next_elem();
        CHECK(rk_fri_fold_evals(ctx, (uint32_t*)nxt, (const uint32_t*)cur, n / 2, beta));
        void* t = cur; cur = nxt; nxt = t;
        n /= 2;
    }
    uint32_t fin[16 * 4];
    CHECK(rk_d2h(ctx, fin, cur, n * 16));
    int constant = 1, nonzero = 0;
    for (size_t i = 0; i < n; i++) {
        if (memcmp(fin + 4 * i, fin, 16) != 0) constant = 0;
        for (int c = 0; c < 4; c++) nonzero |= fin[4 * i + c] != 0;
    }
    printf("reduced opening folded %u times: %zu values, %s\n", k, n, constant && nonzero ? "constant" : "NOT constant");

    /* the challenger's proof of work on an arbitrary sponge state */
    uint32_t state[16], witness = 0;
    for (int i = 0; i < 16; i++) state[i] = next_elem();
    CHECK(rk_duplex_grind(ctx, state, fin, 4, 12, &witness));
    printf("grind(12 bits) -> witness %u\n", witness);

    rk_free(ctx, d_trace); rk_free(ctx, d_lde); rk_free(ctx, d_nodes); rk_free(ctx, d_ro); rk_free(ctx, d_ys); rk_free(ctx, d_next);
    rk_ctx_destroy(ctx);
    free(trace); free(ys); free(zero);
    return constant && nonzero ? 0 : 2;
}
