/* TEST INFRASTRUCTURE -- CPU oracle (see oracle.h header: parity unpinned).
 *
 * A literal restatement of Plonky3's univariate STARK over the two-adic FRI PCS, for one or several tables
 * proven under shared challenges the way SP1 core proves the chips of a shard (reference call site:
 * provers/sp1/driver/src/lib.rs:44-57 `client.setup(ELF)` / `client.prove(&pk, stdin)`, shard knobs
 * docs/README_Sp1.md:19-32; the crates -- p3-uni-stark prover.rs / verifier.rs / folder.rs / symbolic_*.rs,
 * p3-commit domain.rs, p3-fri two_adic_pcs.rs / prover.rs / verifier.rs, p3-challenger duplex_challenger.rs at
 * Plonky3@88ea2b8, reference Cargo.lock:4889-5127 -- are outside the reference tree: RECALLED).
 *
 * What is NOT here: SP1's chips and its recursion VM.  The permutation (LogUp) argument that ties the tables of a shard
 * together IS here in the shape sp1-core gives it (stark/permutation.rs, lookup/interaction.rs, RECALLED): a table lists
 * interactions -- tuples (bus, values...) sent or received with a multiplicity per row --; after the main traces are
 * committed two extension challenges alpha, beta are drawn, every table with interactions gets a permutation trace of
 * ceil(n / 2) + 1 extension columns (one per batch of two interactions: sum of +-mult / (alpha + sum_j beta^j x_j), x_0
 * the bus; the last column the running sum of the row totals), committed as a second batch; the constraints that tie
 * it to the main trace are ordinary AIR steps over PERM_LOCAL / PERM_NEXT / CHALLENGE / CUMSUM (the front end writes
 * them); the verifier additionally checks that the tables' cumulative sums add up to zero.  An AIR is data -- a list
 * of steps over the local row, the next row, public values and the three selectors (or_air_step), the shape an
 * `Air::eval` call leaves in a symbolic builder.
 *
 * Flow (uni-stark prove, per table, with the commitments batched over tables):
 *   observe(init words); commit every trace's coset LDE (bit-reversed rows) in one MMCS; observe(root),
 *   observe(public values); [with interactions: pa, pb <- sample_ext; permutation traces; commit; observe(root),
 *   observe(cumulative sums)]; alpha <- sample_ext;
 *   per table: quotient values on the disjoint coset s * H_(N * qd) -- acc = acc * alpha + constraint over the
 *   asserts in order, times 1 / Z_H(x) -- split into qd chunks (rows j, j + qd, ...), each flattened to 4 base
 *   columns; commit every chunk's LDE in one MMCS; observe(root); zeta <- sample_ext;
 *   PCS open: rounds = [traces at {zeta, zeta * g_N}], [chunks at {zeta}]; alpha' <- sample_ext; barycentric opened
 *   values; reduced openings per LDE height; FRI commit phase (arity 2 on evaluations, shorter inputs join when the
 *   sizes meet) down to `blowup` equal values; observe(final); proof of work; queries.
 * Proof = u32 words (field elements as Montgomery words, the form every buffer of this repo uses):
 *   n_tables | log_height per table | trace root 8 | [permutation root 8 | cumulative sum 4 per table with interactions] |
 *   quotient root 8 | per table: trace_local 4w, trace_next 4w, [perm_local 4 * 4P, perm_next 4 * 4P], quotient chunks qd x 4 x 4 |
 *   n_rounds | n_rounds x 8 commit-phase roots | final_poly 4 | pow witness 1 (canonical integer) |
 *   per query: [trace batch: every table's opened LDE row, then the path] [permutation batch] [quotient batch: every
 *   chunk's row, then the path] then per FRI round: sibling value 4, path. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

static unsigned p3_log2(size_t n) { unsigned k = 0; while (((size_t)1 << k) < n) k++; return k; }
static size_t p3_bitrev(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

/* ---------------------------------------------------------------- p3-challenger DuplexChallenger */
typedef struct { fp state[OR_MAX_CELLS], in[OR_MAX_CELLS], out[OR_MAX_CELLS]; size_t n_in, n_out; } chal;
static void ch_duplex(chal* c) {
    for (size_t i = 0; i < c->n_in; i++) c->state[i] = c->in[i];
    c->n_in = 0;
    or_poseidon2_mix(c->state);
    for (size_t i = 0; i < OR_CELLS_RATE; i++) c->out[i] = c->state[i];
    c->n_out = OR_CELLS_RATE;
}
static void ch_observe(chal* c, fp v) {
    c->n_out = 0;
    c->in[c->n_in++] = v;
    if (c->n_in == OR_CELLS_RATE) ch_duplex(c);
}
static void ch_observe_n(chal* c, const fp* v, size_t n) { for (size_t i = 0; i < n; i++) ch_observe(c, v[i]); }
static fp ch_sample(chal* c) {
    if (c->n_in != 0 || c->n_out == 0) ch_duplex(c);
    return c->out[--c->n_out];
}
static fp4 ch_sample_ext(chal* c) { fp4 r; for (int i = 0; i < 4; i++) r.c[i] = ch_sample(c); return r; }
static uint32_t ch_sample_bits(chal* c, unsigned bits) { return fp_to_u32(ch_sample(c)) & (uint32_t)(((uint64_t)1 << bits) - 1); }
static int ch_check_witness(chal* c, unsigned bits, uint32_t w) {
    ch_observe(c, fp_from_u32(w));
    return ch_sample_bits(c, bits) == 0;
}

/* ---------------------------------------------------------------- the AIR as data */
/* p3-uni-stark symbolic_expression.rs degree_multiple: a trace cell 1, is_first_row / is_last_row 1,
 * is_transition 0, constants and public values 0; add / sub max, mul sum.  get_log_quotient_degree:
 * log2_ceil(max(constraint degree, 2) - 1). */
int or_air_log_quotient_degree(const or_air* air) {
    uint32_t* deg = (uint32_t*)calloc(air->n_steps + 1, sizeof(uint32_t));
    size_t nv = 0;
    uint32_t max_deg = 0;
    for (size_t s = 0; s < air->n_steps; s++) {
        const or_air_step* st = &air->steps[s];
        switch (st->op) {
            case OR_AIR_CONST: case OR_AIR_PUBLIC: case OR_AIR_IS_TRANSITION: case OR_AIR_CHALLENGE: case OR_AIR_CUMSUM: deg[nv++] = 0; break;
            case OR_AIR_LOCAL: case OR_AIR_NEXT: case OR_AIR_IS_FIRST_ROW: case OR_AIR_IS_LAST_ROW:
            case OR_AIR_PERM_LOCAL: case OR_AIR_PERM_NEXT: deg[nv++] = 1; break;
            case OR_AIR_ADD: case OR_AIR_SUB: deg[nv] = deg[st->a] > deg[st->b] ? deg[st->a] : deg[st->b]; nv++; break;
            case OR_AIR_MUL: deg[nv] = deg[st->a] + deg[st->b]; nv++; break;
            case OR_AIR_NEG: deg[nv] = deg[st->a]; nv++; break;
            case OR_AIR_ASSERT_ZERO: if (deg[st->a] > max_deg) max_deg = deg[st->a]; break;
            default: free(deg); return -1;
        }
    }
    free(deg);
    if (max_deg < 2) max_deg = 2;
    return (int)p3_log2(max_deg - 1);
}
/* extension columns of the permutation trace: one per batch of two interactions + the running sum */
static uint32_t perm_ext_cols(const or_air* air) { return air->n_interactions ? (air->n_interactions + 1) / 2 + 1 : 0; }
/* powers of beta the challenge vector carries: beta^0 .. beta^K, K = the longest value tuple */
static uint32_t perm_max_values(const or_air* air) {
    uint32_t k = 0;
    for (uint32_t i = 0; i < air->n_interactions; i++) if (air->interactions[i].n_values > k) k = air->interactions[i].n_values;
    return k;
}
static int air_check(const or_air* air, uint32_t width, uint32_t n_public) {
    size_t nv = 0;
    const uint32_t pw = 4 * perm_ext_cols(air), nch = air->n_interactions ? 4 * (perm_max_values(air) + 2) : 0;
    for (uint32_t i = 0; i < air->n_interactions; i++) {
        const or_interaction* it = &air->interactions[i];
        if (it->kind > 1 || it->bus >= OR_P || it->n_values > 64 || (it->n_values && !it->value_cols)) return -1;
        if (it->mult_is_const ? it->mult >= OR_P : it->mult >= width) return -1;
        for (uint32_t j = 0; j < it->n_values; j++) if (it->value_cols[j] >= width) return -1;
    }
    for (size_t s = 0; s < air->n_steps; s++) {
        const or_air_step* st = &air->steps[s];
        switch (st->op) {
            case OR_AIR_CONST: if (st->a >= OR_P) return -1; nv++; break;
            case OR_AIR_LOCAL: case OR_AIR_NEXT: if (st->a >= width) return -1; nv++; break;
            case OR_AIR_PUBLIC: if (st->a >= n_public) return -1; nv++; break;
            case OR_AIR_IS_FIRST_ROW: case OR_AIR_IS_LAST_ROW: case OR_AIR_IS_TRANSITION: nv++; break;
            case OR_AIR_PERM_LOCAL: case OR_AIR_PERM_NEXT: if (st->a >= pw) return -1; nv++; break;
            case OR_AIR_CHALLENGE: if (st->a >= nch) return -1; nv++; break;
            case OR_AIR_CUMSUM: if (st->a >= 4 || !air->n_interactions) return -1; nv++; break;
            case OR_AIR_ADD: case OR_AIR_SUB: case OR_AIR_MUL: if (st->a >= nv || st->b >= nv) return -1; nv++; break;
            case OR_AIR_NEG: if (st->a >= nv) return -1; nv++; break;
            case OR_AIR_ASSERT_ZERO: if (st->a >= nv) return -1; break;
            default: return -1;
        }
    }
    return 0;
}
/* folder.rs: `assert_zero(x)`: accumulator = accumulator * alpha + x, in the order the AIR asserts.  One evaluator
 * for both sides: the prover's rows are base-field values embedded in the extension, the verifier's are openings. */
typedef struct { const fp4 *local, *next; const fp* chal; const fp* cumsum; } perm_view;   /* NULLs for a table without interactions */
static fp4 air_fold(const or_air* air, const fp4* local, const fp4* next, const fp* pub, fp4 is_first, fp4 is_last,
                    fp4 is_trans, fp4 alpha, fp4* vals, const perm_view* pv) {
    size_t nv = 0;
    fp4 acc = fp4_zero();
    for (size_t s = 0; s < air->n_steps; s++) {
        const or_air_step* st = &air->steps[s];
        switch (st->op) {
            case OR_AIR_CONST: vals[nv++] = fp4_from_fp(fp_from_u32(st->a)); break;
            case OR_AIR_LOCAL: vals[nv++] = local[st->a]; break;
            case OR_AIR_NEXT: vals[nv++] = next[st->a]; break;
            case OR_AIR_PUBLIC: vals[nv++] = fp4_from_fp(pub[st->a]); break;
            case OR_AIR_IS_FIRST_ROW: vals[nv++] = is_first; break;
            case OR_AIR_IS_LAST_ROW: vals[nv++] = is_last; break;
            case OR_AIR_IS_TRANSITION: vals[nv++] = is_trans; break;
            case OR_AIR_PERM_LOCAL: vals[nv++] = pv->local[st->a]; break;
            case OR_AIR_PERM_NEXT: vals[nv++] = pv->next[st->a]; break;
            case OR_AIR_CHALLENGE: vals[nv++] = fp4_from_fp(pv->chal[st->a]); break;
            case OR_AIR_CUMSUM: vals[nv++] = fp4_from_fp(pv->cumsum[st->a]); break;
            case OR_AIR_ADD: vals[nv] = fp4_add(vals[st->a], vals[st->b]); nv++; break;
            case OR_AIR_SUB: vals[nv] = fp4_sub(vals[st->a], vals[st->b]); nv++; break;
            case OR_AIR_MUL: vals[nv] = fp4_mul(vals[st->a], vals[st->b]); nv++; break;
            case OR_AIR_NEG: vals[nv] = fp4_sub(fp4_zero(), vals[st->a]); nv++; break;
            default: acc = fp4_add(fp4_mul(acc, alpha), vals[st->a]); break;   /* ASSERT_ZERO */
        }
    }
    return acc;
}

/* ---------------------------------------------------------------- helpers */
typedef struct { uint32_t* p; size_t n, cap; } wvec;
static void wv_push(wvec* v, const uint32_t* w, size_t n) {
    if (v->n + n > v->cap) {
        v->cap = (v->n + n) * 2 + 64;
        v->p = (uint32_t*)realloc(v->p, v->cap * 4);
    }
    memcpy(v->p + v->n, w, n * 4);
    v->n += n;
}
static void wv_push1(wvec* v, uint32_t w) { wv_push(v, &w, 1); }

/* Radix2Dit::coset_lde_batch(evals, log_blowup, shift).bit_reverse_rows(): row r of out = the columns' interpolants
 * (as if the evaluations were over the subgroup) at shift * g_K^bitrev(r) */
static void lde_rows_shift(fp* out, const fp* in, size_t h, size_t w, fp shift) {
    const unsigned k = p3_log2(h), kb = k + g_or.blowup_log2;
    const size_t H = (size_t)1 << kb;
    fp* col = (fp*)malloc(H * sizeof(fp));
    fp* tmp = (fp*)malloc(h * sizeof(fp));
    for (size_t c = 0; c < w; c++) {
        memset(col, 0, H * sizeof(fp));
        for (size_t i = 0; i < h; i++) tmp[i] = in[i * w + c];
        or_interpolate_ntt(tmp, h);
        fp s = fp_from_u32(1);
        for (size_t i = 0; i < h; i++) {
            col[p3_bitrev(i, kb)] = fp_mul(tmp[p3_bitrev(i, k)], s);
            s = fp_mul(s, shift);
        }
        or_evaluate_ntt(col, H, 0);
        for (size_t j = 0; j < H; j++) out[p3_bitrev(j, kb) * w + c] = col[j];
    }
    free(tmp);
    free(col);
}
/* Mmcs::open_batch: the row index >> log2(H / height) of every matrix, then the siblings from the leaves up */
static void mmcs_open(const or_matrix* mats, uint32_t n, const uint32_t* nodes, uint32_t H, uint32_t index, wvec* out) {
    for (uint32_t m = 0; m < n; m++) {
        size_t r = index / (H / mats[m].height);
        wv_push(out, mats[m].values + r * mats[m].width, mats[m].width);
    }
    for (size_t idx = (size_t)H + index; idx > 1; idx >>= 1) wv_push(out, nodes + (idx ^ 1) * 8, 8);
}

typedef struct {
    uint32_t log_n, width, lqd;   /* trace height, width, log2 of the quotient degree */
    fp* lde;                      /* (N << blowup) x width */
    fp* chunk_lde[16];            /* qd matrices (N << blowup) x 4 */
    fp4 *y_local, *y_next;        /* opened values */
    fp4 y_chunk[16][4];
    uint32_t pw;                  /* base columns of the permutation trace (0: none) */
    fp* perm;                     /* N x pw row-major */
    fp* perm_lde;                 /* (N << blowup) x pw */
    fp4 *yp_local, *yp_next;
    fp cumsum[4];
} tstate;

/* the challenge vector of the permutation argument: [alpha | beta^0 | ... | beta^K] as base components */
static void perm_challenges(fp4 pa, fp4 pb, uint32_t K, fp* out) {
    memcpy(out, pa.c, 16);
    fp4 cur = fp4_one();
    for (uint32_t j = 0; j <= K; j++) { memcpy(out + 4 * (j + 1), cur.c, 16); cur = fp4_mul(cur, pb); }
}
/* sp1-core generate_permutation_trace (RECALLED): per row and batch of two interactions the sum of +-mult / rlc,
 * rlc = alpha + sum_j beta^j x_j (x_0 = the bus, then the values); last column = inclusive running sum of the row totals */
static void perm_trace(const or_air* air, const fp* main, size_t n, size_t w, const fp* chal, fp* perm, fp* cumsum) {
    const uint32_t nb = (air->n_interactions + 1) / 2, pw = 4 * (nb + 1);
    const fp4* C = (const fp4*)chal;   /* C[0] = alpha, C[1 + j] = beta^j */
    fp4 phi = fp4_zero();
    for (size_t r = 0; r < n; r++) {
        fp4 row_sum = fp4_zero();
        for (uint32_t b = 0; b < nb; b++) {
            fp4 entry = fp4_zero();
            for (uint32_t i = 2 * b; i < 2 * b + 2 && i < air->n_interactions; i++) {
                const or_interaction* it = &air->interactions[i];
                fp4 rlc = fp4_add(C[0], fp4_scale(C[1], fp_from_u32(it->bus)));
                for (uint32_t j = 0; j < it->n_values; j++) rlc = fp4_add(rlc, fp4_scale(C[2 + j], main[r * w + it->value_cols[j]]));
                fp m = it->mult_is_const ? fp_from_u32(it->mult) : main[r * w + it->mult];
                fp4 term = fp4_scale(fp4_inv(rlc), m);
                entry = it->kind == 0 ? fp4_add(entry, term) : fp4_sub(entry, term);
            }
            memcpy(perm + r * pw + 4 * b, entry.c, 16);
            row_sum = fp4_add(row_sum, entry);
        }
        phi = fp4_add(phi, row_sum);
        memcpy(perm + r * pw + 4 * nb, phi.c, 16);
    }
    memcpy(cumsum, phi.c, 16);
}

/* LagrangeSelectors at a point of the extension (domain.rs selectors_at_point; the prover's selectors_on_coset are
 * the same expressions at base-field points): the trace domain is the subgroup H of order 2^log_n */
static void selectors(fp4 x, unsigned log_n, fp4* is_first, fp4* is_last, fp4* is_trans, fp4* inv_zeroifier) {
    fp4 z_h = fp4_sub(fp4_pow(x, (uint64_t)1 << log_n), fp4_one());
    fp g_inv = or_rou_rev(log_n);
    *is_first = fp4_mul(z_h, fp4_inv(fp4_sub(x, fp4_one())));
    *is_last = fp4_mul(z_h, fp4_inv(fp4_sub(x, fp4_from_fp(g_inv))));
    *is_trans = fp4_sub(x, fp4_from_fp(g_inv));
    *inv_zeroifier = fp4_inv(z_h);
}

int or_p3_prove(const or_p3_table* tables, uint32_t n_tables, const fp* init, size_t n_init, uint32_t** proof_out, size_t* words_out) {
    const unsigned blow = g_or.blowup_log2;
    const fp shift = fp_from_u32(g_or.coset_shift);
    if (n_tables == 0 || n_tables > 32) return -1;
    for (uint32_t t = 0; t < n_tables; t++) {
        const or_p3_table* tb = &tables[t];
        if (tb->log_height < 1 || tb->log_height + blow > 24 || tb->width == 0) return -1;
        if (air_check(tb->air, tb->width, tb->n_public) != 0) return -1;
        int lqd = or_air_log_quotient_degree(tb->air);
        if (lqd < 0 || (unsigned)lqd > blow) return -2;   /* the LDE must cover the quotient domain */
    }
    tstate* ts = (tstate*)calloc(n_tables, sizeof(tstate));
    wvec pf = {0, 0, 0};
    chal ch; memset(&ch, 0, sizeof ch);
    ch_observe_n(&ch, init, n_init);

    /* ---- trace commitment */
    wv_push1(&pf, n_tables);
    or_matrix* tmats = (or_matrix*)calloc(n_tables, sizeof(or_matrix));
    uint32_t Ht = 0;
    for (uint32_t t = 0; t < n_tables; t++) {
        const or_p3_table* tb = &tables[t];
        ts[t].log_n = tb->log_height; ts[t].width = tb->width; ts[t].lqd = (uint32_t)or_air_log_quotient_degree(tb->air);
        size_t n = (size_t)1 << tb->log_height, H = n << blow;
        ts[t].lde = (fp*)malloc(H * tb->width * sizeof(fp));
        or_pcs_coset_lde_rows(ts[t].lde, tb->trace, n, tb->width);
        tmats[t].values = ts[t].lde; tmats[t].height = (uint32_t)H; tmats[t].width = tb->width; tmats[t].row_major = 1;
        if (H > Ht) Ht = (uint32_t)H;
        wv_push1(&pf, tb->log_height);
    }
    uint32_t* tnodes = (uint32_t*)malloc((size_t)2 * Ht * 8 * 4);
    or_mmcs_commit(tmats, n_tables, tnodes);
    wv_push(&pf, tnodes + 8, 8);
    ch_observe_n(&ch, tnodes + 8, 8);
    for (uint32_t t = 0; t < n_tables; t++) ch_observe_n(&ch, tables[t].public_values, tables[t].n_public);

    /* ---- permutation traces (only when some table lists interactions: proofs without any keep their bytes) */
    uint32_t n_perm = 0, Hp = 0, Kmax = 0;
    for (uint32_t t = 0; t < n_tables; t++) if (tables[t].air->n_interactions) {
        n_perm++;
        if (perm_max_values(tables[t].air) > Kmax) Kmax = perm_max_values(tables[t].air);
    }
    or_matrix* pmats = (or_matrix*)calloc(n_perm + 1, sizeof(or_matrix));
    uint32_t* pnodes = NULL;
    fp* pchal = (fp*)calloc(4 * (Kmax + 2), sizeof(fp));
    if (n_perm) {
        const fp4 pa = ch_sample_ext(&ch), pb = ch_sample_ext(&ch);
        perm_challenges(pa, pb, Kmax, pchal);
        uint32_t pm = 0;
        for (uint32_t t = 0; t < n_tables; t++) {
            const or_p3_table* tb = &tables[t];
            if (!tb->air->n_interactions) continue;
            size_t n = (size_t)1 << tb->log_height, H = n << blow;
            ts[t].pw = 4 * perm_ext_cols(tb->air);
            ts[t].perm = (fp*)malloc(n * ts[t].pw * sizeof(fp));
            perm_trace(tb->air, tb->trace, n, tb->width, pchal, ts[t].perm, ts[t].cumsum);
            ts[t].perm_lde = (fp*)malloc(H * ts[t].pw * sizeof(fp));
            or_pcs_coset_lde_rows(ts[t].perm_lde, ts[t].perm, n, ts[t].pw);
            pmats[pm].values = ts[t].perm_lde; pmats[pm].height = (uint32_t)H; pmats[pm].width = ts[t].pw; pmats[pm].row_major = 1;
            pm++;
            if (H > Hp) Hp = (uint32_t)H;
        }
        pnodes = (uint32_t*)malloc((size_t)2 * Hp * 8 * 4);
        or_mmcs_commit(pmats, n_perm, pnodes);
        wv_push(&pf, pnodes + 8, 8);
        ch_observe_n(&ch, pnodes + 8, 8);
        for (uint32_t t = 0; t < n_tables; t++) if (ts[t].pw) { wv_push(&pf, ts[t].cumsum, 4); ch_observe_n(&ch, ts[t].cumsum, 4); }
    }
    const fp4 alpha = ch_sample_ext(&ch);

    /* ---- quotient: values on s * H_(N qd), split, chunk LDEs, commitment */
    uint32_t n_chunks = 0, Hq = 0;
    for (uint32_t t = 0; t < n_tables; t++) n_chunks += 1u << ts[t].lqd;
    or_matrix* qmats = (or_matrix*)calloc(n_chunks, sizeof(or_matrix));
    uint32_t qm = 0;
    for (uint32_t t = 0; t < n_tables; t++) {
        const or_p3_table* tb = &tables[t];
        const unsigned k = ts[t].log_n, lqd = ts[t].lqd, kq = k + lqd, kb = k + blow;
        const size_t n = (size_t)1 << k, qd = (size_t)1 << lqd, nq = n << lqd, H = n << blow, w = tb->width;
        fp** chunk = (fp**)malloc(qd * sizeof(fp*));
        for (size_t j = 0; j < qd; j++) chunk[j] = (fp*)malloc(n * 4 * sizeof(fp));
        const fp gq = or_rou_fwd(kq);
#pragma omp parallel
        {
            fp4* local = (fp4*)malloc(w * sizeof(fp4));
            fp4* next = (fp4*)malloc(w * sizeof(fp4));
            fp4* vals = (fp4*)malloc((tb->air->n_steps + 1) * sizeof(fp4));
            const size_t pw = ts[t].pw;
            fp4* pl = (fp4*)malloc((pw + 1) * sizeof(fp4));
            fp4* pn = (fp4*)malloc((pw + 1) * sizeof(fp4));
            perm_view pv = {pl, pn, pchal, ts[t].cumsum};
#pragma omp for schedule(static)
            for (size_t i = 0; i < nq; i++) {
                /* get_evaluations_on_domain: the quotient domain's point i is LDE row bitrev(i * (2^blow / qd)) */
                const size_t r0 = p3_bitrev(i << (blow - lqd), kb), r1 = p3_bitrev(((i + qd) & (nq - 1)) << (blow - lqd), kb);
                for (size_t c = 0; c < w; c++) {
                    local[c] = fp4_from_fp(ts[t].lde[r0 * w + c]);
                    next[c] = fp4_from_fp(ts[t].lde[r1 * w + c]);
                }
                for (size_t c = 0; c < pw; c++) {
                    pl[c] = fp4_from_fp(ts[t].perm_lde[r0 * pw + c]);
                    pn[c] = fp4_from_fp(ts[t].perm_lde[r1 * pw + c]);
                }
                fp4 x = fp4_from_fp(fp_mul(shift, fp_pow(gq, i))), f, l, tr, iz;
                selectors(x, k, &f, &l, &tr, &iz);
                fp4 q = fp4_mul(air_fold(tb->air, local, next, tb->public_values, f, l, tr, alpha, vals, &pv), iz);
                memcpy(chunk[i & (qd - 1)] + (i >> lqd) * 4, q.c, 16);   /* split_evals: chunk j = rows j, j + qd, ... */
            }
            free(pn); free(pl); free(vals); free(next); free(local);
        }
        for (size_t j = 0; j < qd; j++) {
            /* commit(domain = s g_q^j H_N, evals): shift of the LDE = generator / domain.shift = g_q^-j */
            ts[t].chunk_lde[j] = (fp*)malloc(H * 4 * sizeof(fp));
            lde_rows_shift(ts[t].chunk_lde[j], chunk[j], n, 4, fp_pow(or_rou_rev(kq), j));
            qmats[qm].values = ts[t].chunk_lde[j]; qmats[qm].height = (uint32_t)H; qmats[qm].width = 4; qmats[qm].row_major = 1;
            qm++;
            free(chunk[j]);
        }
        free(chunk);
        if (H > Hq) Hq = (uint32_t)H;
    }
    uint32_t* qnodes = (uint32_t*)malloc((size_t)2 * Hq * 8 * 4);
    or_mmcs_commit(qmats, n_chunks, qnodes);
    wv_push(&pf, qnodes + 8, 8);
    ch_observe_n(&ch, qnodes + 8, 8);
    const fp4 zeta = ch_sample_ext(&ch);

    /* ---- PCS open */
    const fp4 alpha2 = ch_sample_ext(&ch);
    fp4* ro[32]; uint64_t num_reduced[32];
    memset(ro, 0, sizeof ro); memset(num_reduced, 0, sizeof num_reduced);
    unsigned log_max = 0;
    for (uint32_t t = 0; t < n_tables; t++) {          /* round 0: the traces at zeta and zeta * g */
        const unsigned lh = ts[t].log_n + blow;
        const size_t H = (size_t)1 << lh, w = ts[t].width;
        if (lh > log_max) log_max = lh;
        if (!ro[lh]) ro[lh] = (fp4*)calloc(H, sizeof(fp4));
        fp4 pts[2]; pts[0] = zeta; pts[1] = fp4_scale(zeta, or_rou_fwd(ts[t].log_n));
        fp4* ys = (fp4*)malloc(2 * w * sizeof(fp4));
        for (int j = 0; j < 2; j++) or_pcs_eval_at(ys + j * w, ts[t].lde, H, w, pts[j].c);
        or_pcs_reduce_openings(ro[lh], ts[t].lde, H, w, 2, pts[0].c, ys[0].c, alpha2.c, num_reduced[lh]);
        num_reduced[lh] += 2 * w;
        ts[t].y_local = ys; ts[t].y_next = ys + w;
    }
    for (uint32_t t = 0; t < n_tables; t++) {          /* round 1 (with interactions): the permutation traces at zeta and zeta * g */
        if (!ts[t].pw) continue;
        const unsigned lh = ts[t].log_n + blow;
        const size_t H = (size_t)1 << lh, w = ts[t].pw;
        fp4 pts[2]; pts[0] = zeta; pts[1] = fp4_scale(zeta, or_rou_fwd(ts[t].log_n));
        fp4* ys = (fp4*)malloc(2 * w * sizeof(fp4));
        for (int j = 0; j < 2; j++) or_pcs_eval_at(ys + j * w, ts[t].perm_lde, H, w, pts[j].c);
        or_pcs_reduce_openings(ro[lh], ts[t].perm_lde, H, w, 2, pts[0].c, ys[0].c, alpha2.c, num_reduced[lh]);
        num_reduced[lh] += 2 * w;
        ts[t].yp_local = ys; ts[t].yp_next = ys + w;
    }
    for (uint32_t t = 0; t < n_tables; t++) {          /* last round: the quotient chunks at zeta */
        const unsigned lh = ts[t].log_n + blow;
        const size_t H = (size_t)1 << lh;
        for (size_t j = 0; j < ((size_t)1 << ts[t].lqd); j++) {
            or_pcs_eval_at(ts[t].y_chunk[j], ts[t].chunk_lde[j], H, 4, zeta.c);
            or_pcs_reduce_openings(ro[lh], ts[t].chunk_lde[j], H, 4, 1, zeta.c, ts[t].y_chunk[j][0].c, alpha2.c, num_reduced[lh]);
            num_reduced[lh] += 4;
        }
    }
    for (uint32_t t = 0; t < n_tables; t++) {
        wv_push(&pf, ts[t].y_local[0].c, 4 * ts[t].width);
        wv_push(&pf, ts[t].y_next[0].c, 4 * ts[t].width);
        if (ts[t].pw) { wv_push(&pf, ts[t].yp_local[0].c, 4 * ts[t].pw); wv_push(&pf, ts[t].yp_next[0].c, 4 * ts[t].pw); }
        for (size_t j = 0; j < ((size_t)1 << ts[t].lqd); j++) wv_push(&pf, ts[t].y_chunk[j][0].c, 16);
    }

    /* ---- FRI commit phase (p3-fri prover.rs commit_phase) */
    const unsigned n_rounds = log_max - blow;
    fp4** layer = (fp4**)calloc(n_rounds + 1, sizeof(fp4*));
    uint32_t** lnodes = (uint32_t**)calloc(n_rounds + 1, sizeof(uint32_t*));
    fp4* folded = ro[log_max];
    size_t len = (size_t)1 << log_max;
    wv_push1(&pf, n_rounds);
    for (unsigned rd = 0; rd < n_rounds; rd++) {
        or_matrix lm; lm.values = (const fp*)folded; lm.height = (uint32_t)(len / 2); lm.width = 8; lm.row_major = 1;
        lnodes[rd] = (uint32_t*)malloc(len * 8 * 4);
        or_mmcs_commit(&lm, 1, lnodes[rd]);
        wv_push(&pf, lnodes[rd] + 8, 8);
        ch_observe_n(&ch, lnodes[rd] + 8, 8);
        fp4 beta = ch_sample_ext(&ch);
        fp4* nxt = (fp4*)malloc(len / 2 * sizeof(fp4));
        or_fri_fold_evals(nxt, folded, len / 2, beta.c);
        layer[rd] = folded;
        folded = nxt; len /= 2;
        unsigned lg = p3_log2(len);
        if (lg != log_max && lg < 32 && ro[lg])
            for (size_t i = 0; i < len; i++) folded[i] = fp4_add(folded[i], ro[lg][i]);
    }
    int rc = 0;
    for (size_t i = 1; i < len; i++) if (!fp4_eq(folded[i], folded[0])) rc = -3;   /* `blowup` values of a constant */
    wv_push(&pf, folded[0].c, 4);
    ch_observe_n(&ch, folded[0].c, 4);
    uint32_t witness = g_or.pow_bits ? or_duplex_grind(ch.state, ch.in, ch.n_in, g_or.pow_bits) : 0;
    if (!ch_check_witness(&ch, g_or.pow_bits, witness)) rc = -4;
    wv_push1(&pf, witness);

    /* ---- queries */
    for (unsigned q = 0; q < g_or.queries && rc == 0; q++) {
        uint32_t index = ch_sample_bits(&ch, log_max);
        mmcs_open(tmats, n_tables, tnodes, Ht, index >> (log_max - p3_log2(Ht)), &pf);
        if (n_perm) mmcs_open(pmats, n_perm, pnodes, Hp, index >> (log_max - p3_log2(Hp)), &pf);
        mmcs_open(qmats, n_chunks, qnodes, Hq, index >> (log_max - p3_log2(Hq)), &pf);
        for (unsigned rd = 0; rd < n_rounds; rd++) {
            uint32_t idx = index >> rd, pair = idx >> 1;
            size_t height = ((size_t)1 << (log_max - rd)) / 2;
            wv_push(&pf, layer[rd][2 * (size_t)pair + ((idx ^ 1) & 1)].c, 4);
            for (size_t at = height + pair; at > 1; at >>= 1) wv_push(&pf, lnodes[rd] + (at ^ 1) * 8, 8);
        }
    }
    /* release */
    for (unsigned rd = 0; rd < n_rounds; rd++) { free(lnodes[rd]); if (layer[rd] != ro[log_max]) free(layer[rd]); }
    if (n_rounds == 0 || folded != ro[log_max]) free(folded);
    free(layer); free(lnodes);
    for (int i = 0; i < 32; i++) free(ro[i]);
    for (uint32_t t = 0; t < n_tables; t++) {
        free(ts[t].lde); free(ts[t].y_local);
        free(ts[t].perm); free(ts[t].perm_lde); free(ts[t].yp_local);
        for (int j = 0; j < 16; j++) free(ts[t].chunk_lde[j]);
    }
    free(pchal); free(pnodes); free(pmats);
    free(qnodes); free(tnodes); free(qmats); free(tmats); free(ts);
    if (rc != 0) { free(pf.p); return rc; }
    *proof_out = pf.p; *words_out = pf.n;
    return 0;
}

/* ---------------------------------------------------------------- verifier (uni-stark verifier.rs, two_adic_pcs.rs
 * verify, p3-fri verifier.rs).  0 = accept; reason codes: 1 malformed / short / non-canonical word, 2 shape
 * mismatch, 3 constraint identity (OodEvaluationMismatch), 4 proof of work, 5 input opening, 6 commit-phase
 * opening, 7 final polynomial, 8 the tables' cumulative sums do not add up to zero */
typedef struct { const uint32_t* p; size_t n, pos; int bad; } rd_t;
static const uint32_t* rd_take(rd_t* r, size_t n) {
    static const uint32_t zeros[64] = {0};
    if (r->pos + n > r->n) { r->bad = 1; return n <= 64 ? zeros : NULL; }
    const uint32_t* q = r->p + r->pos;
    r->pos += n;
    return q;
}
int or_p3_verify(const or_p3_table* tables, uint32_t n_tables, const fp* init, size_t n_init, const uint32_t* proof, size_t words) {
    const unsigned blow = g_or.blowup_log2;
    const fp shift = fp_from_u32(g_or.coset_shift);
    if (n_tables == 0 || n_tables > 32) return -1;
    for (size_t i = 0; i < words; i++) if (proof[i] >= OR_P) return 1;
    rd_t r = {proof, words, 0, 0};
    if (*rd_take(&r, 1) != n_tables || r.bad) return 2;
    unsigned log_n[32], lqd[32], log_max = 0;
    for (uint32_t t = 0; t < n_tables; t++) {
        log_n[t] = *rd_take(&r, 1);
        if (r.bad || log_n[t] < 1 || log_n[t] + blow > 24) return 2;
        if (tables[t].log_height && tables[t].log_height != log_n[t]) return 2;   /* a height the statement pins */
        if (air_check(tables[t].air, tables[t].width, tables[t].n_public) != 0) return -1;
        int d = or_air_log_quotient_degree(tables[t].air);
        if (d < 0 || (unsigned)d > blow) return 2;
        lqd[t] = (unsigned)d;
        if (log_n[t] + blow > log_max) log_max = log_n[t] + blow;
    }
    chal ch; memset(&ch, 0, sizeof ch);
    ch_observe_n(&ch, init, n_init);
    const uint32_t* troot = rd_take(&r, 8);
    if (r.bad) return 1;
    ch_observe_n(&ch, troot, 8);
    for (uint32_t t = 0; t < n_tables; t++) ch_observe_n(&ch, tables[t].public_values, tables[t].n_public);
    uint32_t n_perm = 0, Kmax = 0, pwid[32];
    unsigned log_pmax = 0;
    for (uint32_t t = 0; t < n_tables; t++) {
        pwid[t] = 4 * perm_ext_cols(tables[t].air);
        if (!pwid[t]) continue;
        n_perm++;
        if (perm_max_values(tables[t].air) > Kmax) Kmax = perm_max_values(tables[t].air);
        if (log_n[t] + blow > log_pmax) log_pmax = log_n[t] + blow;
    }
    const uint32_t* proot = NULL;
    const fp* cumsum[32];
    fp pchal[4 * 66];
    memset(pchal, 0, sizeof pchal);
    memset(cumsum, 0, sizeof cumsum);
    if (n_perm) {
        const fp4 pa = ch_sample_ext(&ch), pb = ch_sample_ext(&ch);
        perm_challenges(pa, pb, Kmax, pchal);
        proot = rd_take(&r, 8);
        if (r.bad) return 1;
        ch_observe_n(&ch, proot, 8);
        fp4 total = fp4_zero();
        for (uint32_t t = 0; t < n_tables; t++) if (pwid[t]) {
            cumsum[t] = rd_take(&r, 4);
            if (r.bad) return 1;
            ch_observe_n(&ch, cumsum[t], 4);
            total = fp4_add(total, *(const fp4*)cumsum[t]);
        }
        if (!fp4_eq(total, fp4_zero())) return 8;
    }
    const fp4 alpha = ch_sample_ext(&ch);
    const uint32_t* qroot = rd_take(&r, 8);
    if (r.bad) return 1;
    ch_observe_n(&ch, qroot, 8);
    const fp4 zeta = ch_sample_ext(&ch);
    const fp4 *y_local[32], *y_next[32], *y_chunk[32], *yp_local[32], *yp_next[32];
    for (uint32_t t = 0; t < n_tables; t++) {
        y_local[t] = (const fp4*)rd_take(&r, 4 * (size_t)tables[t].width);
        y_next[t] = (const fp4*)rd_take(&r, 4 * (size_t)tables[t].width);
        yp_local[t] = yp_next[t] = NULL;
        if (pwid[t]) {
            yp_local[t] = (const fp4*)rd_take(&r, 4 * (size_t)pwid[t]);
            yp_next[t] = (const fp4*)rd_take(&r, 4 * (size_t)pwid[t]);
            if (r.bad || !yp_local[t] || !yp_next[t]) return 1;
        }
        y_chunk[t] = (const fp4*)rd_take(&r, (size_t)16 << lqd[t]);
        if (r.bad || !y_local[t] || !y_next[t] || !y_chunk[t]) return 1;
    }
    /* the constraint identity per table: folded_constraints(zeta) / Z_H(zeta) == quotient(zeta) */
    for (uint32_t t = 0; t < n_tables; t++) {
        const unsigned k = log_n[t], kq = k + lqd[t];
        const size_t qd = (size_t)1 << lqd[t], n = (size_t)1 << k;
        /* zps[i] = prod_{j != i} Z_j(zeta) / Z_j(first point of chunk domain i), Z_j(x) = (x / shift_j)^n - 1 */
        fp4 quotient = fp4_zero();
        for (size_t i = 0; i < qd; i++) {
            fp4 zp = fp4_one();
            fp first_i = fp_mul(shift, fp_pow(or_rou_fwd(kq), i));
            for (size_t j = 0; j < qd; j++) {
                if (j == i) continue;
                fp sj_inv = fp_inv(fp_mul(shift, fp_pow(or_rou_fwd(kq), j)));
                fp4 a = fp4_sub(fp4_pow(fp4_scale(zeta, sj_inv), n), fp4_one());
                fp b = fp_sub(fp_pow(fp_mul(first_i, sj_inv), n), fp_from_u32(1));
                zp = fp4_mul(zp, fp4_scale(a, fp_inv(b)));
            }
            for (int e = 0; e < 4; e++) {
                fp4 mono = fp4_zero(); mono.c[e] = fp_from_u32(1);
                quotient = fp4_add(quotient, fp4_mul(fp4_mul(zp, mono), y_chunk[t][i * 4 + e]));
            }
        }
        fp4 f, l, tr, iz;
        selectors(zeta, k, &f, &l, &tr, &iz);
        fp4* vals = (fp4*)malloc((tables[t].air->n_steps + 1) * sizeof(fp4));
        perm_view pv = {yp_local[t], yp_next[t], pchal, cumsum[t]};
        fp4 folded = air_fold(tables[t].air, y_local[t], y_next[t], tables[t].public_values, f, l, tr, alpha, vals, &pv);
        free(vals);
        if (!fp4_eq(fp4_mul(folded, iz), quotient)) return 3;
    }
    /* PCS */
    const fp4 alpha2 = ch_sample_ext(&ch);
    const uint32_t n_rounds = *rd_take(&r, 1);
    if (r.bad || n_rounds != log_max - blow) return 2;
    const uint32_t* commits = rd_take(&r, (size_t)8 * n_rounds);
    if (r.bad || (n_rounds && !commits)) return 1;
    fp4 betas[32];
    for (uint32_t rd = 0; rd < n_rounds; rd++) {
        ch_observe_n(&ch, commits + 8 * rd, 8);
        betas[rd] = ch_sample_ext(&ch);
    }
    const fp4* final_poly = (const fp4*)rd_take(&r, 4);
    if (r.bad) return 1;
    ch_observe_n(&ch, final_poly->c, 4);
    const uint32_t witness = *rd_take(&r, 1);
    if (r.bad) return 1;
    if (!ch_check_witness(&ch, g_or.pow_bits, witness)) return 4;

    uint32_t n_chunks = 0;
    for (uint32_t t = 0; t < n_tables; t++) n_chunks += 1u << lqd[t];
    uint32_t *th = (uint32_t*)malloc(n_tables * 4), *tw = (uint32_t*)malloc(n_tables * 4);
    uint32_t *qh = (uint32_t*)malloc(n_chunks * 4), *qw = (uint32_t*)malloc(n_chunks * 4);
    uint32_t ph[32], pwd[32];
    size_t trow = 0, qrow = 4 * (size_t)n_chunks, prow = 0;
    for (uint32_t t = 0, m = 0, pm = 0; t < n_tables; t++) {
        th[t] = 1u << (log_n[t] + blow); tw[t] = tables[t].width; trow += tables[t].width;
        if (pwid[t]) { ph[pm] = th[t]; pwd[pm] = pwid[t]; pm++; prow += pwid[t]; }
        for (uint32_t j = 0; j < (1u << lqd[t]); j++, m++) { qh[m] = th[t]; qw[m] = 4; }
    }
    int rc = 0;
    for (unsigned q = 0; q < g_or.queries && rc == 0; q++) {
        const uint32_t index = ch_sample_bits(&ch, log_max);
        /* the trace and quotient batches have the global maximum height (every table is in both); the permutation
         * batch only holds the tables with interactions */
        const uint32_t* trows = rd_take(&r, trow); const uint32_t* tpath = rd_take(&r, (size_t)8 * log_max);
        const uint32_t *prows = NULL, *ppath = NULL;
        if (n_perm) { prows = rd_take(&r, prow); ppath = rd_take(&r, (size_t)8 * log_pmax); if (r.bad || !prows || !ppath) { rc = 1; break; } }
        const uint32_t* qrows = rd_take(&r, qrow); const uint32_t* qpath = rd_take(&r, (size_t)8 * log_max);
        if (r.bad || !trows || !tpath || !qrows || !qpath) { rc = 1; break; }
        if (or_mmcs_verify(th, tw, n_tables, index, trows, tpath, troot) != 0) { rc = 5; break; }
        if (n_perm && or_mmcs_verify(ph, pwd, n_perm, index >> (log_max - log_pmax), prows, ppath, proot) != 0) { rc = 5; break; }
        if (or_mmcs_verify(qh, qw, n_chunks, index, qrows, qpath, qroot) != 0) { rc = 5; break; }
        fp4 rop[32], apow[32]; int used[32];
        for (int i = 0; i < 32; i++) { rop[i] = fp4_zero(); apow[i] = fp4_one(); used[i] = 0; }
#define P3_REDUCE(LH, X, PT, PZ, PX) do { \
        fp4 den_ = fp4_sub(fp4_from_fp(X), (PT)); \
        fp4 quot_ = fp4_mul(fp4_sub(fp4_from_fp(PX), (PZ)), fp4_inv(den_)); \
        rop[LH] = fp4_add(rop[LH], fp4_mul(apow[LH], quot_)); apow[LH] = fp4_mul(apow[LH], alpha2); } while (0)
        size_t at = 0;
        for (uint32_t t = 0; t < n_tables; t++) {
            const unsigned lh = log_n[t] + blow;
            const fp x = fp_mul(shift, fp_pow(or_rou_fwd(lh), p3_bitrev(index >> (log_max - lh), lh)));
            used[lh] = 1;
            const fp4 zn = fp4_scale(zeta, or_rou_fwd(log_n[t]));
            for (uint32_t c = 0; c < tables[t].width; c++) P3_REDUCE(lh, x, zeta, y_local[t][c], trows[at + c]);
            for (uint32_t c = 0; c < tables[t].width; c++) P3_REDUCE(lh, x, zn, y_next[t][c], trows[at + c]);
            at += tables[t].width;
        }
        at = 0;
        for (uint32_t t = 0; t < n_tables; t++) {
            if (!pwid[t]) continue;
            const unsigned lh = log_n[t] + blow;
            const fp x = fp_mul(shift, fp_pow(or_rou_fwd(lh), p3_bitrev(index >> (log_max - lh), lh)));
            const fp4 zn = fp4_scale(zeta, or_rou_fwd(log_n[t]));
            for (uint32_t c = 0; c < pwid[t]; c++) P3_REDUCE(lh, x, zeta, yp_local[t][c], prows[at + c]);
            for (uint32_t c = 0; c < pwid[t]; c++) P3_REDUCE(lh, x, zn, yp_next[t][c], prows[at + c]);
            at += pwid[t];
        }
        at = 0;
        for (uint32_t t = 0; t < n_tables; t++) {
            const unsigned lh = log_n[t] + blow;
            const fp x = fp_mul(shift, fp_pow(or_rou_fwd(lh), p3_bitrev(index >> (log_max - lh), lh)));
            for (uint32_t j = 0; j < (1u << lqd[t]); j++, at += 4)
                for (int c = 0; c < 4; c++) P3_REDUCE(lh, x, zeta, y_chunk[t][j * 4 + c], qrows[at + c]);
        }
        /* verify_query */
        fp4 folded = fp4_zero();
        uint32_t idx = index;
        for (uint32_t rd = 0; rd < n_rounds; rd++) {
            const unsigned lfh = log_max - 1 - rd;
            if (used[lfh + 1]) folded = fp4_add(folded, rop[lfh + 1]);
            const fp4* sib = (const fp4*)rd_take(&r, 4);
            const uint32_t* path = rd_take(&r, (size_t)8 * lfh);
            if (r.bad || (lfh && !path)) { rc = 1; break; }
            fp4 evals[2];
            evals[idx & 1] = folded; evals[(idx ^ 1) & 1] = *sib;
            uint32_t dims_h = 1u << lfh, dims_w = 8;
            if (or_mmcs_verify(&dims_h, &dims_w, 1, idx >> 1, evals[0].c, path, commits + 8 * rd) != 0) { rc = 6; break; }
            idx >>= 1;
            /* fold_row: the line through (x0, e0), (-x0, e1) at beta, x0 = g^bitrev(idx) of the 2^(lfh+1) subgroup */
            fp x0 = fp_pow(or_rou_fwd(lfh + 1), p3_bitrev(idx, lfh));
            fp4 slope = fp4_scale(fp4_sub(evals[1], evals[0]), fp_inv(fp_sub(fp_neg(x0), x0)));
            folded = fp4_add(evals[0], fp4_mul(fp4_sub(betas[rd], fp4_from_fp(x0)), slope));
        }
        if (rc == 0 && !fp4_eq(folded, *final_poly)) rc = 7;
    }
    free(th); free(tw); free(qh); free(qw);
    if (rc == 0 && r.pos != r.n) rc = 1;
    return rc;
}
