/* TEST INFRASTRUCTURE -- CPU oracle (see oracle.h header: parity unpinned).
 * Restates risc0-zkp 1.0.1 prove/{merkle,write_iop,poly_group,prover,fri}.rs,
 * core/hash/poseidon2/rng.rs and the prove_segment driver of
 * risc0-circuit-rv32im 1.0.1 -- what `session.prove()` runs per segment at
 * /root/reference provers/risc0/driver/src/bonsai.rs:271.  The circuit-specific
 * steps (witness generation, accum construction, eval_check) have no source in
 * the container; their outputs are inputs of or_segment (SURVEY.md 8d "S20"). */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static or_timing g_timing;
void or_last_timing(or_timing* t) { *t = g_timing; }

static unsigned ilog2(size_t n) { unsigned k = 0; while (((size_t)1 << k) < n) k++; return k; }

/* ------------------------------------------------------------- merkle.rs */
void or_merkle_build(or_merkle* m, const fp* matrix, size_t rows, size_t cols, size_t queries) {
    m->rows = rows; m->cols = cols; m->queries = queries; m->matrix = matrix;
    m->layers = ilog2(rows);
    m->top_layer = 0;
    for (size_t i = 1; i < m->layers; i++) {
        if (((size_t)1 << i) > queries) break;
        m->top_layer = i;
    }
    m->top_size = (size_t)1 << m->top_layer;
    m->nodes = (uint32_t*)calloc(2 * rows * OR_DIGEST_WORDS, sizeof(uint32_t));
    or_hash_rows(m->nodes + rows * OR_DIGEST_WORDS, matrix, rows, cols);
    for (size_t i = m->layers; i-- > 0;) {
        size_t layer = (size_t)1 << i;
        or_hash_fold(m->nodes, layer * 2, layer);
    }
}
void or_merkle_free(or_merkle* m) { free(m->nodes); m->nodes = NULL; }
static const uint32_t* merkle_root(const or_merkle* m) { return m->nodes + OR_DIGEST_WORDS; }
static void merkle_commit(const or_merkle* m, or_iop* iop) {
    or_iop_write(iop, m->nodes + m->top_size * OR_DIGEST_WORDS, m->top_size * OR_DIGEST_WORDS);
    or_iop_commit(iop, merkle_root(m));
}
static void merkle_prove(const or_merkle* m, or_iop* iop, size_t idx) {
    fp* col = (fp*)malloc(m->cols * sizeof(fp));
    or_gather_sample(col, m->matrix, idx, m->cols, m->rows);
    or_iop_write(iop, col, m->cols);
    free(col);
    idx += m->rows;
    while (idx >= 2 * m->top_size) {
        size_t low = idx & 1;
        idx >>= 1;
        size_t other = 2 * idx + (1 - low);
        or_iop_write(iop, m->nodes + other * OR_DIGEST_WORDS, OR_DIGEST_WORDS);
    }
}

/* ------------------------------------------------- write_iop.rs + rng.rs */
void or_iop_init(or_iop* iop) {
    memset(iop, 0, sizeof *iop);
    iop->cap = 1 << 16;
    iop->proof = (uint32_t*)malloc(iop->cap * sizeof(uint32_t));
}
void or_iop_free(or_iop* iop) { free(iop->proof); iop->proof = NULL; }
void or_iop_write(or_iop* iop, const uint32_t* w, size_t n) {
    if (iop->len + n > iop->cap) {
        while (iop->len + n > iop->cap) iop->cap *= 2;
        iop->proof = (uint32_t*)realloc(iop->proof, iop->cap * sizeof(uint32_t));
    }
    memcpy(iop->proof + iop->len, w, n * sizeof(uint32_t));
    iop->len += n;
}
void or_iop_commit(or_iop* iop, const uint32_t* digest) {
    if (iop->pool_used != 0) { or_poseidon2_mix(iop->cells); iop->pool_used = 0; }
    for (int i = 0; i < OR_CELLS_OUT; i++) iop->cells[i] = fp_add(iop->cells[i], digest[i]);
    or_poseidon2_mix(iop->cells);
}
fp or_iop_random_elem(or_iop* iop) {
    if (iop->pool_used == OR_CELLS_RATE) { or_poseidon2_mix(iop->cells); iop->pool_used = 0; }
    return iop->cells[iop->pool_used++];
}
fp4 or_iop_random_ext(or_iop* iop) {
    fp4 r;
    for (int i = 0; i < 4; i++) r.c[i] = or_iop_random_elem(iop);
    return r;
}
uint32_t or_iop_random_bits(or_iop* iop, unsigned bits) {
    uint32_t val = fp_to_u32(or_iop_random_elem(iop));
    for (int i = 0; i < 3; i++) val ^= fp_to_u32(or_iop_random_elem(iop));
    return val & (uint32_t)(((uint64_t)1 << bits) - 1);
}

/* the literal definition: try w = 0, 1, 2, ... on a copy of the transcript generator -- write nothing,
 * absorb hash([w]) as or_iop_commit does, draw the bits as or_iop_random_bits does */
uint32_t or_pow_grind(const or_iop* iop, unsigned bits) {
    for (uint32_t w = 0; w < OR_P; w++) {
        or_iop t = *iop; /* shares `proof`, which the trial never touches */
        uint32_t digest[8];
        or_hash_elem_slice(&w, 1, 1, digest);
        or_iop_commit(&t, digest);
        if (or_iop_random_bits(&t, bits) == 0) return w;
    }
    return 0xffffffffu;
}

/* ---------------------------------------------------------- poly_group.rs */
typedef struct {
    fp* coeffs;      /* count x size, natural order after construction */
    fp* evaluated;   /* count x size*INV_RATE */
    size_t count, size;
    or_merkle merkle;
} polygroup;
/* coeffs arrive interpolated + zk-shifted, bit-reversed; ownership taken */
static void polygroup_new(polygroup* g, fp* coeffs, size_t count, size_t size) {
    double t0 = now_s();
    g->coeffs = coeffs; g->count = count; g->size = size;
    size_t domain = size * OR_INV_RATE;
    g->evaluated = (fp*)malloc(count * domain * sizeof(fp));
    or_batch_expand_into_evaluate_ntt(g->evaluated, coeffs, size, count, OR_INV_RATE_PO2);
    or_batch_bit_reverse(coeffs, size, count);
    double t1 = now_s();
    or_merkle_build(&g->merkle, g->evaluated, domain, count, OR_QUERIES);
    g_timing.ntt += t1 - t0;
    g_timing.hash += now_s() - t1;
}
static void polygroup_free(polygroup* g) {
    free(g->coeffs); free(g->evaluated); or_merkle_free(&g->merkle);
}
/* prover.rs commit_group: copy, interpolate, shift, LDE, Merkle, send root */
static void commit_group(polygroup* g, or_iop* iop, const fp* trace, size_t count, size_t size) {
    double t0 = now_s();
    fp* coeffs = (fp*)malloc(count * size * sizeof(fp));
    or_eltwise_copy_elem(coeffs, trace, count * size);
    or_batch_interpolate_ntt(coeffs, size, count);
    or_zk_shift(coeffs, size, count);
    g_timing.ntt += now_s() - t0;
    polygroup_new(g, coeffs, count, size);
    merkle_commit(&g->merkle, iop);
}

/* ------------------------------------------------------------------ fri.rs */
typedef struct { size_t domain; fp* coeffs; size_t coeffs_size; fp* evaluated; or_merkle merkle; } fri_round;
static void fri_round_new(fri_round* r, or_iop* iop, const fp* coeffs, size_t coeffs_size) {
    size_t size = coeffs_size / OR_EXT;
    size_t domain = size * OR_INV_RATE;
    r->domain = domain;
    r->evaluated = (fp*)malloc(domain * OR_EXT * sizeof(fp));
    or_batch_expand_into_evaluate_ntt(r->evaluated, coeffs, size, OR_EXT, OR_INV_RATE_PO2);
    or_merkle_build(&r->merkle, r->evaluated, domain / OR_FRI_FOLD, OR_FRI_FOLD * OR_EXT, OR_QUERIES);
    merkle_commit(&r->merkle, iop);
    fp4 fold_mix = or_iop_random_ext(iop);
    r->coeffs_size = size / OR_FRI_FOLD * OR_EXT;
    r->coeffs = (fp*)malloc(r->coeffs_size * sizeof(fp));
    or_fri_fold(r->coeffs, coeffs, size / OR_FRI_FOLD, fold_mix.c);
}
typedef void (*inner_fn)(void* ctx, or_iop* iop, size_t idx);
static void fri_prove(or_iop* iop, const fp* coeffs_in, size_t coeffs_size, inner_fn inner, void* ctx) {
    double t0 = now_s();
    size_t orig_domain = coeffs_size / OR_EXT * OR_INV_RATE;
    fri_round rounds[32];
    int n_rounds = 0;
    const fp* coeffs = coeffs_in;
    while (coeffs_size / OR_EXT > OR_FRI_MIN_DEGREE && coeffs_size / OR_EXT >= OR_FRI_FOLD) {
        fri_round_new(&rounds[n_rounds], iop, coeffs, coeffs_size);
        coeffs = rounds[n_rounds].coeffs;
        coeffs_size = rounds[n_rounds].coeffs_size;
        n_rounds++;
    }
    fp* final_coeffs = (fp*)malloc(coeffs_size * sizeof(fp));
    or_eltwise_copy_elem(final_coeffs, coeffs, coeffs_size);
    or_batch_bit_reverse(final_coeffs, coeffs_size / OR_EXT, OR_EXT);
    or_iop_write(iop, final_coeffs, coeffs_size);
    uint32_t digest[8];
    or_hash_elem_slice(final_coeffs, coeffs_size, 1, digest);
    or_iop_commit(iop, digest);
    free(final_coeffs);
    if (g_or.pow_bits) { /* proof of work before the query positions exist; the nonce is absorbed hashed */
        uint32_t nonce = or_pow_grind(iop, g_or.pow_bits);
        or_iop_write(iop, &nonce, 1);
        or_hash_elem_slice(&nonce, 1, 1, digest);
        or_iop_commit(iop, digest);
        (void)or_iop_random_bits(iop, g_or.pow_bits); /* zero by construction; the verifier draws them too */
    }
    double t1 = now_s();
    g_timing.fri += t1 - t0;
    for (uint32_t q = 0; q < OR_QUERIES; q++) {
        uint32_t rng = or_iop_random_bits(iop, ilog2(orig_domain));
        size_t pos = rng % orig_domain;
        inner(ctx, iop, pos);
        for (int r = 0; r < n_rounds; r++) {
            size_t group = pos % (rounds[r].domain / OR_FRI_FOLD);
            merkle_prove(&rounds[r].merkle, iop, group);
            pos = group;
        }
    }
    g_timing.query += now_s() - t1;
    for (int r = 0; r < n_rounds; r++) {
        free(rounds[r].coeffs); free(rounds[r].evaluated); or_merkle_free(&rounds[r].merkle);
    }
}

/* --------------------------------------------------------------- prover.rs */
typedef struct { polygroup* groups; polygroup* check; } inner_ctx;
static void inner_prove(void* vctx, or_iop* iop, size_t idx) {
    inner_ctx* c = (inner_ctx*)vctx;
    for (int g = 0; g < 3; g++) merkle_prove(&c->groups[g].merkle, iop, idx);
    merkle_prove(&c->check->merkle, iop, idx);
}
static void hash_info(const uint8_t* info, uint32_t* digest) {
    fp e[16];
    for (int i = 0; i < 16; i++) e[i] = fp_from_u32(info[i]);
    or_hash_elem_slice(e, 16, 1, digest);
}

int or_prove_segment(const or_segment* seg, uint32_t** seal, size_t* seal_words, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    memset(&g_timing, 0, sizeof g_timing);
    /* the segment flow is risc0's; blow-up, fold arity, final degree, queries and proof of work follow g_or */
    if (seg->po2 < 1 || seg->po2 + g_or.blowup_log2 > 24) return -4;
    double t_start = now_s();
    const or_taps* taps = &seg->taps;
    size_t N = (size_t)1 << seg->po2;
    size_t D = N * OR_INV_RATE;
    or_iop iop; or_iop_init(&iop);
    uint32_t digest[8];

    hash_info(seg->proof_system_info, digest); or_iop_commit(&iop, digest);
    hash_info(seg->circuit_info, digest); or_iop_commit(&iop, digest);
    {
        fp* vec = (fp*)malloc((seg->n_globals + 1) * sizeof(fp));
        memcpy(vec, seg->globals, seg->n_globals * sizeof(fp));
        vec[seg->n_globals] = fp_from_u32(seg->po2);
        or_hash_elem_slice(vec, seg->n_globals + 1, 1, digest);
        or_iop_commit(&iop, digest);
        free(vec);
        or_iop_write(&iop, seg->globals, seg->n_globals);
        or_iop_write(&iop, &seg->po2, 1);
    }
    polygroup groups[3];
    commit_group(&groups[1], &iop, seg->group[1], taps->group_size[1], N); /* code */
    commit_group(&groups[2], &iop, seg->group[2], taps->group_size[2], N); /* data */
    /* rv32im prove_segment: the accum mix is drawn once code and data are bound */
    fp* accum_mix = (fp*)malloc((seg->n_accum_mix + 1) * sizeof(fp));
    for (uint32_t i = 0; i < seg->n_accum_mix; i++) accum_mix[i] = or_iop_random_elem(&iop);
    or_circuit_view view;
    memset(&view, 0, sizeof view);
    view.po2 = seg->po2;
    for (int g = 0; g < 3; g++) view.group_size[g] = taps->group_size[g];
    view.globals = seg->globals; view.n_globals = seg->n_globals;
    view.mix = accum_mix; view.n_mix = seg->n_accum_mix;
    const or_circuit_hooks* hooks = seg->hooks;
    fp* accum_buf = NULL;
    const fp* accum_trace = seg->group[0];
    if (hooks && hooks->accumulate) { /* CircuitHal::accumulate */
        accum_buf = (fp*)malloc((size_t)taps->group_size[0] * N * sizeof(fp));
        view.trace[1] = seg->group[1]; view.trace[2] = seg->group[2];
        view.lde[1] = groups[1].evaluated; view.lde[2] = groups[2].evaluated;
        if (hooks->accumulate(hooks->user, &view, accum_buf) != 0) { or_iop_free(&iop); return -3; }
        accum_trace = accum_buf;
        view.trace[1] = view.trace[2] = NULL;
    }
    commit_group(&groups[0], &iop, accum_trace, taps->group_size[0], N); /* accum */
    free(accum_buf);

    /* finalize */
    fp4 poly_mix = or_iop_random_ext(&iop);
    double t0 = now_s();
    fp* check_poly = (fp*)malloc(OR_EXT * D * sizeof(fp));
    if (hooks && hooks->eval_check) { /* CircuitHal::eval_check over the LDE domain */
        for (int g = 0; g < 3; g++) view.lde[g] = groups[g].evaluated;
        if (hooks->eval_check(hooks->user, &view, poly_mix.c, check_poly) != 0) { or_iop_free(&iop); return -3; }
    } else {
        or_eltwise_copy_elem(check_poly, seg->check, OR_EXT * D); /* pre-computed stand-in */
    }
    free(accum_mix);
    /* 4 x D evaluations at 3*w^i -> 4 x D bit-reversed coefficients = 16 columns of N (quarter c of
     * component e = coefficients n with n mod 4 = bitrev2(c)).  No zk_shift: these already are the
     * coefficients of y -> check(3y), the form every PolyGroup is kept in. */
    or_batch_interpolate_ntt(check_poly, D, OR_EXT);
    g_timing.ntt += now_s() - t0;
    polygroup check;
    polygroup_new(&check, check_poly, OR_CHECK_SIZE, N);
    merkle_commit(&check.merkle, &iop);

    t0 = now_s();
    fp4 z = or_iop_random_ext(&iop);
    fp back_one = or_rou_rev(seg->po2);
    size_t tot_taps = 0;
    for (uint32_t r = 0; r < taps->n_regs; r++)
        tot_taps += taps->combo_off[taps->reg_combo[r] + 1] - taps->combo_off[taps->reg_combo[r]];
    fp4* all_xs = (fp4*)malloc(tot_taps * sizeof(fp4));
    fp4* eval_u = (fp4*)malloc(tot_taps * sizeof(fp4));
    uint32_t* which = (uint32_t*)malloc((tot_taps + OR_CHECK_SIZE) * sizeof(uint32_t));
    size_t pos = 0;
    uint32_t reg = 0;
    for (uint32_t gid = 0; gid < 3; gid++) {
        size_t start = pos;
        for (; reg < taps->n_regs && taps->reg_group[reg] == gid; reg++) {
            uint32_t cb = taps->reg_combo[reg];
            for (uint32_t b = taps->combo_off[cb]; b < taps->combo_off[cb + 1]; b++) {
                which[pos] = taps->reg_offset[reg];
                all_xs[pos] = fp4_scale(z, fp_pow(back_one, taps->combo_backs[b]));
                pos++;
            }
        }
        or_batch_evaluate_any(groups[gid].coeffs, N, which + start, all_xs + start, pos - start, eval_u + start);
    }
    size_t n_coeff_u = tot_taps + OR_CHECK_SIZE;
    fp4* coeff_u = (fp4*)malloc(n_coeff_u * sizeof(fp4));
    pos = 0;
    for (uint32_t r = 0; r < taps->n_regs; r++) {
        uint32_t cb = taps->reg_combo[r];
        size_t sz = taps->combo_off[cb + 1] - taps->combo_off[cb];
        or_poly_interpolate(coeff_u + pos, all_xs + pos, eval_u + pos, sz);
        pos += sz;
    }
    fp4 z_pow = fp4_pow(z, OR_INV_RATE);
    {
        fp4 xs[OR_MAX_CHECK_SIZE];
        for (uint32_t i = 0; i < OR_CHECK_SIZE; i++) { which[i] = i; xs[i] = z_pow; }
        or_batch_evaluate_any(check.coeffs, N, which, xs, OR_CHECK_SIZE, coeff_u + pos);
    }
    or_iop_write(&iop, (const uint32_t*)coeff_u, n_coeff_u * OR_EXT);
    or_hash_elem_slice((const fp*)coeff_u, n_coeff_u * OR_EXT, 1, digest);
    or_iop_commit(&iop, digest);

    fp4 mix = or_iop_random_ext(&iop);
    size_t combo_count = taps->n_combos;
    fp4* combos = (fp4*)calloc((combo_count + 1) * N, sizeof(fp4));
    fp4 cur_mix = fp4_one();
    reg = 0;
    for (uint32_t gid = 0; gid < 3; gid++) {
        uint32_t gs = taps->group_size[gid];
        for (uint32_t i = 0; i < gs; i++, reg++) which[i] = taps->reg_combo[reg];
        or_mix_poly_coeffs(combos, cur_mix.c, mix.c, groups[gid].coeffs, which, gs, N);
        cur_mix = fp4_mul(cur_mix, fp4_pow(mix, gs));
    }
    for (uint32_t i = 0; i < OR_CHECK_SIZE; i++) which[i] = (uint32_t)combo_count;
    or_mix_poly_coeffs(combos, cur_mix.c, mix.c, check.coeffs, which, OR_CHECK_SIZE, N);

    /* subtract the U polynomials, divide out the tap points */
    {
        size_t cur_pos = 0;
        fp4 cur = fp4_one();
        for (uint32_t r = 0; r < taps->n_regs; r++) {
            uint32_t cb = taps->reg_combo[r];
            size_t sz = taps->combo_off[cb + 1] - taps->combo_off[cb];
            for (size_t i = 0; i < sz; i++) {
                fp4* o = &combos[N * cb + i];
                *o = fp4_sub(*o, fp4_mul(cur, coeff_u[cur_pos + i]));
            }
            cur = fp4_mul(cur, mix);
            cur_pos += sz;
        }
        for (uint32_t i = 0; i < OR_CHECK_SIZE; i++) {
            fp4* o = &combos[N * combo_count];
            *o = fp4_sub(*o, fp4_mul(cur, coeff_u[cur_pos++]));
            cur = fp4_mul(cur, mix);
        }
        int bad = 0;
#pragma omp parallel for schedule(dynamic) reduction(| : bad)
        for (size_t c = 0; c < combo_count; c++) {
            for (uint32_t b = taps->combo_off[c]; b < taps->combo_off[c + 1]; b++) {
                fp4 pt = fp4_scale(z, fp_pow(back_one, taps->combo_backs[b]));
                fp4 rem;
                or_poly_divide(combos + c * N, N, pt.c, rem.c);
                if (!fp4_eq(rem, fp4_zero())) bad |= 1;
            }
        }
        fp4 rem;
        or_poly_divide(combos + combo_count * N, N, z_pow.c, rem.c);
        if (!fp4_eq(rem, fp4_zero())) bad |= 1;
        if (bad) { or_iop_free(&iop); return -2; }
    }
    fp* final_poly = (fp*)malloc(N * OR_EXT * sizeof(fp));
    or_eltwise_sum_extelem(final_poly, combos, N, combo_count + 1);
    or_batch_bit_reverse(final_poly, N, OR_EXT);
    free(combos);
    g_timing.deep += now_s() - t0;

    inner_ctx ictx = {groups, &check};
    fri_prove(&iop, final_poly, N * OR_EXT, inner_prove, &ictx);

    free(final_poly); free(coeff_u); free(which); free(eval_u); free(all_xs);
    for (int g = 0; g < 3; g++) polygroup_free(&groups[g]);
    polygroup_free(&check);
    *seal = iop.proof; *seal_words = iop.len;
    g_timing.total = now_s() - t_start;
    return 0;
}
