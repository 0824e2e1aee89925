/* TEST INFRASTRUCTURE -- CPU oracle (see oracle.h header: parity unpinned).
 * Restates risc0-zkp 1.0.1 verify/{mod,fri,merkle,read_iop}.rs for the flow of
 * or_prove_segment: transcript binding, Merkle openings, DEEP quotient
 * consistency, FRI folds, the final low-degree polynomial and, given the
 * circuit's poly_ext, the constraint identity on the tap openings (the rv32im
 * constraint system itself is absent from the container; or_toy.c is the toy one). */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

typedef struct {
    const uint32_t* p; size_t len, pos; int err;
    or_iop rng; /* only cells/pool_used are used */
} riop;
static void r_read(riop* r, uint32_t* out, size_t n) {
    if (r->pos + n > r->len) { r->err = 1; memset(out, 0, n * 4); return; }
    memcpy(out, r->p + r->pos, n * 4);
    r->pos += n;
}
static unsigned ilog2(size_t n) { unsigned k = 0; while (((size_t)1 << k) < n) k++; return k; }

typedef struct { size_t rows, cols, layers, top_layer, top_size; uint32_t* top; /* 2*top_size digests, heap */ } mverifier;
static void mv_new(mverifier* m, riop* r, size_t rows, size_t cols, size_t queries) {
    m->rows = rows; m->cols = cols; m->layers = ilog2(rows); m->top_layer = 0;
    for (size_t i = 1; i < m->layers; i++) { if (((size_t)1 << i) > queries) break; m->top_layer = i; }
    m->top_size = (size_t)1 << m->top_layer;
    m->top = (uint32_t*)calloc(2 * m->top_size * 8, 4);
    r_read(r, m->top + m->top_size * 8, m->top_size * 8);
    for (size_t i = m->top_size; i-- > 1;) or_hash_pair(m->top + 2 * i * 8, m->top + (2 * i + 1) * 8, m->top + i * 8);
    or_iop_commit(&r->rng, m->top + 8);
}
/* returns 0 if the opening is consistent; row values in out[cols] */
static int mv_verify(const mverifier* m, riop* r, size_t idx, fp* out) {
    if (idx >= m->rows) return 1;
    r_read(r, out, m->cols);
    uint32_t cur[8], other[8];
    or_hash_elem_slice(out, m->cols, 1, cur);
    idx += m->rows;
    while (idx >= 2 * m->top_size) {
        size_t low = idx & 1;
        r_read(r, other, 8);
        idx >>= 1;
        uint32_t nxt[8];
        if (low) or_hash_pair(other, cur, nxt); else or_hash_pair(cur, other, nxt);
        memcpy(cur, nxt, 32);
    }
    return memcmp(cur, m->top + idx * 8, 32) != 0;
}

int or_verify_segment(const or_segment* seg, const uint32_t* seal, size_t seal_words) {
    return or_verify_segment_circuit(seg, seal, seal_words, NULL, NULL);
}
int or_verify_segment_circuit(const or_segment* seg, const uint32_t* seal, size_t seal_words,
                              or_poly_ext_fn poly_ext, void* user) {
    const or_taps* taps = &seg->taps;
    if (seg->po2 < 1 || seg->po2 + g_or.blowup_log2 > 24) return -4;
    /* read_iop: a word that is not a canonical field element (>= p) is refused -- fp_add / fp_mul are arithmetic
     * mod p only for operands below p.  Every seal word is an element, a digest word or the small integer po2. */
    for (size_t i = 0; i < seal_words; i++) if (seal[i] >= OR_P) return 63;
    riop r; memset(&r, 0, sizeof r);
    r.p = seal; r.len = seal_words;
    uint32_t digest[8];
    fp e16[16];
    for (int i = 0; i < 16; i++) e16[i] = fp_from_u32(seg->proof_system_info[i]);
    or_hash_elem_slice(e16, 16, 1, digest); or_iop_commit(&r.rng, digest);
    for (int i = 0; i < 16; i++) e16[i] = fp_from_u32(seg->circuit_info[i]);
    or_hash_elem_slice(e16, 16, 1, digest); or_iop_commit(&r.rng, digest);

    fp* io = (fp*)malloc((seg->n_globals + 1) * 4);
    r_read(&r, io, seg->n_globals);
    uint32_t po2; r_read(&r, &po2, 1);
    if (r.err || po2 > 24 || po2 != seg->po2) { free(io); return 10; }
    if (memcmp(io, seg->globals, seg->n_globals * 4) != 0) { free(io); return 11; }
    io[seg->n_globals] = fp_from_u32(po2);
    or_hash_elem_slice(io, seg->n_globals + 1, 1, digest); or_iop_commit(&r.rng, digest);
    free(io);

    size_t N = (size_t)1 << po2, D = N * OR_INV_RATE;
    mverifier mg[3], mcheck;
    mv_new(&mg[1], &r, D, taps->group_size[1], OR_QUERIES);
    mv_new(&mg[2], &r, D, taps->group_size[2], OR_QUERIES);
    fp* accum_mix = (fp*)malloc((seg->n_accum_mix + 1) * sizeof(fp));
    for (uint32_t i = 0; i < seg->n_accum_mix; i++) accum_mix[i] = or_iop_random_elem(&r.rng);
    mv_new(&mg[0], &r, D, taps->group_size[0], OR_QUERIES);
    fp4 poly_mix = or_iop_random_ext(&r.rng);
    mv_new(&mcheck, &r, D, OR_CHECK_SIZE, OR_QUERIES);
    fp4 z = or_iop_random_ext(&r.rng);
    fp back_one = or_rou_rev(po2);

    size_t tot_taps = 0, tot_combo_backs = taps->combo_off[taps->n_combos];
    for (uint32_t i = 0; i < taps->n_regs; i++)
        tot_taps += taps->combo_off[taps->reg_combo[i] + 1] - taps->combo_off[taps->reg_combo[i]];
    size_t n_coeff_u = tot_taps + OR_CHECK_SIZE;
    fp4* coeff_u = (fp4*)malloc(n_coeff_u * sizeof(fp4));
    r_read(&r, (uint32_t*)coeff_u, n_coeff_u * 4);
    or_hash_elem_slice((const fp*)coeff_u, n_coeff_u * 4, 1, digest); or_iop_commit(&r.rng, digest);
    int identity_rc = 0;
    if (poly_ext && !r.err) {
        /* verify/mod.rs: U polynomials -> evaluations, circuit polynomial on them, against
         * check(z) * ((3z)^N - 1), check(z) = sum_i z^i * (ext element from check columns remap[i] + 4e) */
        fp4* eval_u = (fp4*)malloc((tot_taps + 1) * sizeof(fp4));
        size_t pos = 0;
        for (uint32_t i = 0; i < taps->n_regs; i++) {
            uint32_t cb = taps->reg_combo[i];
            size_t sz = taps->combo_off[cb + 1] - taps->combo_off[cb];
            for (size_t j = 0; j < sz; j++) {
                fp4 x = fp4_scale(z, fp_pow(back_one, taps->combo_backs[taps->combo_off[cb] + j]));
                or_poly_eval(coeff_u + pos, sz, x.c, eval_u[pos + j].c);
            }
            pos += sz;
        }
        fp4 result;
        if (poly_ext(user, seg, poly_mix.c, eval_u, tot_taps, accum_mix, seg->n_accum_mix, result.c) != 0) identity_rc = 71;
        /* part i of the check polynomial is in column bitrev(i) of each component ({0,2,1,3} for blow-up 4) */
        size_t parts = OR_INV_RATE;
        fp4 check = fp4_zero(), zi = fp4_one();
        for (size_t i = 0; i < parts; i++) {
            size_t rev = 0;
            for (unsigned b = 0; b < OR_INV_RATE_PO2; b++) rev |= ((i >> b) & 1) << (OR_INV_RATE_PO2 - 1 - b);
            for (int e = 0; e < 4; e++) {
                fp4 basis = fp4_zero();
                basis.c[e] = fp_from_u32(1);
                check = fp4_add(check, fp4_mul(fp4_mul(coeff_u[tot_taps + rev + parts * e], zi), basis));
            }
            zi = fp4_mul(zi, z);
        }
        fp4 vanish = fp4_sub(fp4_pow(fp4_scale(z, fp_from_u32(g_or.coset_shift)), N), fp4_one());
        if (!identity_rc && !fp4_eq(fp4_mul(check, vanish), result)) identity_rc = 70;
        free(eval_u);
    }
    free(accum_mix);
    fp4 mix = or_iop_random_ext(&r.rng);
    fp4* combo_u = (fp4*)calloc(tot_combo_backs + 1, sizeof(fp4));
    {
        fp4 cur = fp4_one(); size_t pos = 0;
        for (uint32_t i = 0; i < taps->n_regs; i++) {
            uint32_t cb = taps->reg_combo[i];
            size_t sz = taps->combo_off[cb + 1] - taps->combo_off[cb];
            for (size_t k = 0; k < sz; k++) {
                fp4* o = &combo_u[taps->combo_off[cb] + k];
                *o = fp4_add(*o, fp4_mul(cur, coeff_u[pos + k]));
            }
            cur = fp4_mul(cur, mix); pos += sz;
        }
        for (size_t i = 0; i < OR_CHECK_SIZE; i++) {
            combo_u[tot_combo_backs] = fp4_add(combo_u[tot_combo_backs], fp4_mul(cur, coeff_u[pos++]));
            cur = fp4_mul(cur, mix);
        }
    }
    fp4 z_pow = fp4_pow(z, OR_INV_RATE);

    /* ---- fri_verify ---- */
    int rc = identity_rc;
    size_t degree = N, domain = D, orig_domain = D;
    struct { size_t domain; mverifier m; fp4 mix; } rounds[32];
    int n_rounds = 0;
    while (degree > OR_FRI_MIN_DEGREE && degree >= OR_FRI_FOLD) {
        rounds[n_rounds].domain = domain;
        mv_new(&rounds[n_rounds].m, &r, domain / OR_FRI_FOLD, OR_FRI_FOLD * OR_EXT, OR_QUERIES);
        rounds[n_rounds].mix = or_iop_random_ext(&r.rng);
        n_rounds++;
        domain /= OR_FRI_FOLD; degree /= OR_FRI_FOLD;
    }
    fp* final_coeffs = (fp*)malloc(OR_EXT * degree * 4);
    r_read(&r, final_coeffs, OR_EXT * degree);
    or_hash_elem_slice(final_coeffs, OR_EXT * degree, 1, digest); or_iop_commit(&r.rng, digest);
    if (g_or.pow_bits) { /* proof of work: the nonce, absorbed hashed, must zero the next pow_bits random bits */
        uint32_t nonce = 0;
        r_read(&r, &nonce, 1);
        or_hash_elem_slice(&nonce, 1, 1, digest); or_iop_commit(&r.rng, digest);
        if (!r.err && (nonce >= OR_P || or_iop_random_bits(&r.rng, g_or.pow_bits) != 0) && !rc) rc = 62;
    }
    fp gen = or_rou_fwd(ilog2(domain));
    fp gen0 = or_rou_fwd(ilog2(orig_domain));
    size_t maxw = taps->group_size[0];
    for (int g = 1; g < 3; g++) if (taps->group_size[g] > maxw) maxw = taps->group_size[g];
    fp* rows[3];
    for (int g = 0; g < 3; g++) rows[g] = (fp*)malloc((taps->group_size[g] + 1) * 4);
    fp check_row[OR_MAX_CHECK_SIZE];
    fp4* tot = (fp4*)malloc((taps->n_combos + 1) * sizeof(fp4));

    for (uint32_t q = 0; q < OR_QUERIES && !rc && !r.err; q++) {
        uint32_t rng = or_iop_random_bits(&r.rng, ilog2(orig_domain));
        size_t pos = rng % orig_domain;
        /* inner: open every group at pos, recompute the DEEP quotient there */
        fp x = fp_pow(gen0, pos);
        for (int g = 0; g < 3; g++) if (mv_verify(&mg[g], &r, pos, rows[g])) { rc = 20 + g; break; }
        if (rc) break;
        if (mv_verify(&mcheck, &r, pos, check_row)) { rc = 23; break; }
        for (uint32_t c = 0; c <= taps->n_combos; c++) tot[c] = fp4_zero();
        fp4 cur = fp4_one();
        for (uint32_t i = 0; i < taps->n_regs; i++) {
            fp v = rows[taps->reg_group[i]][taps->reg_offset[i]];
            tot[taps->reg_combo[i]] = fp4_add(tot[taps->reg_combo[i]], fp4_scale(cur, v));
            cur = fp4_mul(cur, mix);
        }
        for (size_t i = 0; i < OR_CHECK_SIZE; i++) {
            tot[taps->n_combos] = fp4_add(tot[taps->n_combos], fp4_scale(cur, check_row[i]));
            cur = fp4_mul(cur, mix);
        }
        fp4 goal = fp4_zero();
        fp4 xe = fp4_from_fp(x);
        for (uint32_t c = 0; c < taps->n_combos; c++) {
            size_t b0 = taps->combo_off[c], b1 = taps->combo_off[c + 1];
            fp4 ev;
            or_poly_eval(combo_u + b0, b1 - b0, xe.c, ev.c);
            fp4 num = fp4_sub(tot[c], ev);
            fp4 divisor = fp4_one();
            for (size_t b = b0; b < b1; b++)
                divisor = fp4_mul(divisor, fp4_sub(xe, fp4_scale(z, fp_pow(back_one, taps->combo_backs[b]))));
            goal = fp4_add(goal, fp4_mul(num, fp4_inv(divisor)));
        }
        {
            fp4 num = fp4_sub(tot[taps->n_combos], combo_u[tot_combo_backs]);
            fp4 divisor = fp4_sub(xe, z_pow);
            goal = fp4_add(goal, fp4_mul(num, fp4_inv(divisor)));
        }
        /* per-round fold checks */
        for (int k = 0; k < n_rounds && !rc; k++) {
            size_t dom = rounds[k].domain;
            size_t quot = pos / (dom / OR_FRI_FOLD), group = pos % (dom / OR_FRI_FOLD);
            fp data[OR_MAX_FRI_FOLD * OR_EXT];
            if (mv_verify(&rounds[k].m, &r, group, data)) { rc = 30 + (k < 9 ? k : 9); break; }
            fp4 de[OR_MAX_FRI_FOLD];
            for (size_t i = 0; i < OR_FRI_FOLD; i++)
                for (int c = 0; c < 4; c++) de[i].c[c] = data[c * OR_FRI_FOLD + i];
            if (!fp4_eq(de[quot], goal)) { rc = 40 + (k < 9 ? k : 9); break; }
            /* fold_eval: inverse DFT of the 16 coset values, then evaluate at mix * w^-group */
            fp w16_inv = or_rou_rev(OR_FRI_FOLD_PO2);
            fp inv16 = fp_inv(fp_from_u32(OR_FRI_FOLD));
            fp4 gco[OR_MAX_FRI_FOLD];
            for (size_t i = 0; i < OR_FRI_FOLD; i++) {
                fp4 acc = fp4_zero();
                for (size_t j = 0; j < OR_FRI_FOLD; j++)
                    acc = fp4_add(acc, fp4_scale(de[j], fp_pow(w16_inv, (uint64_t)((i * j) % OR_FRI_FOLD))));
                gco[i] = fp4_scale(acc, inv16);
            }
            fp inv_wk = fp_pow(or_rou_rev(ilog2(dom)), group);
            fp4 pt = fp4_scale(rounds[k].mix, inv_wk);
            or_poly_eval(gco, OR_FRI_FOLD, pt.c, goal.c);
            pos = group;
        }
        if (rc) break;
        /* final polynomial */
        fp4 xf = fp4_from_fp(fp_pow(gen, pos));
        fp4 fx = fp4_zero();
        for (size_t i = degree; i-- > 0;) {
            fp4 c;
            for (int k = 0; k < 4; k++) c.c[k] = final_coeffs[k * degree + i];
            fx = fp4_add(fp4_mul(fx, xf), c);
        }
        if (!fp4_eq(fx, goal)) { rc = 50; break; }
    }
    if (!rc && r.err) rc = 60;
    if (!rc && r.pos != r.len) rc = 61;

    for (int g = 0; g < 3; g++) { free(rows[g]); free(mg[g].top); }
    free(mcheck.top); free(tot); free(final_coeffs); free(combo_u); free(coeff_u);
    for (int k = 0; k < n_rounds; k++) free(rounds[k].m.top);
    (void)maxw;
    return rc;
}
