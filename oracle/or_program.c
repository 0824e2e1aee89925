/* TEST INFRASTRUCTURE -- CPU oracle (see oracle.h header: parity unpinned).
 *
 * The circuit's constraint polynomial as a step list: risc0-zkp 1.0.1 adapter.rs
 * `PolyExtStepDef::step` (RECALLED; the crate is outside the reference tree -- the calls that
 * reach it are `session.prove()` at reference provers/risc0/driver/src/bonsai.rs:271 for
 * eval_check and `receipt.verify()` at provers/risc0/driver/src/lib.rs:136 for poly_ext).
 * Two growing lists, field values and mix states {tot, mul}; every step pushes onto one:
 *   CONST a | GET tap | GET_GLOBAL base off | ADD a b | SUB a b | MUL a b          -> value
 *   TRUE | AND_EQZ x v : {x.tot + x.mul*v, x.mul*mix}
 *        | AND_COND x cond inner : {x.tot + cond*inner.tot*x.mul, x.mul*inner.mul}  -> mix state
 * This file runs the list literally -- every step, both halves of every mix state, nothing
 * eliminated or precomputed -- once per LDE point over base-field values (eval_check) and once
 * over extension elements (poly_ext).  raiko_amd/csrc/circuit_program.hip is the compiled form
 * it checks. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

enum { ST_CONST = 0, ST_GET = 1, ST_GET_GLOBAL = 2, ST_ADD = 3, ST_SUB = 4, ST_MUL = 5, ST_TRUE = 6, ST_AND_EQZ = 7,
       ST_AND_COND = 8 };

typedef struct { fp4 tot, mul; } mix_state;

static size_t count_taps(const or_taps* t) {
    size_t n = 0;
    for (uint32_t r = 0; r < t->n_regs; r++) n += t->combo_off[t->reg_combo[r] + 1] - t->combo_off[t->reg_combo[r]];
    return n;
}

/* 0 or the index of the first malformed step + 1 */
static size_t program_check(const or_program* pg, size_t n_taps) {
    size_t nf = 0, nm = 0;
    for (size_t s = 0; s < pg->n_steps; s++) {
        const or_step* st = &pg->steps[s];
        switch (st->op) {
            case ST_CONST: nf++; break;
            case ST_GET: if (st->a >= n_taps) return s + 1; nf++; break;
            case ST_GET_GLOBAL: if (st->a > 1) return s + 1; nf++; break;
            case ST_ADD: case ST_SUB: case ST_MUL: if (st->a >= nf || st->b >= nf) return s + 1; nf++; break;
            case ST_TRUE: nm++; break;
            case ST_AND_EQZ: if (st->a >= nm || st->b >= nf) return s + 1; nm++; break;
            case ST_AND_COND: if (st->a >= nm || st->b >= nf || st->c >= nm) return s + 1; nm++; break;
            default: return s + 1;
        }
    }
    return pg->ret < nm ? 0 : pg->n_steps + 1;
}

/* the list on base-field values of one point; mix states in the extension */
static fp4 run_fp(const or_program* pg, const fp* u, const fp* globals, uint32_t n_globals, const fp* mix, uint32_t n_mix,
                  fp4 poly_mix, fp* vals, mix_state* ms) {
    size_t nf = 0, nm = 0;
    for (size_t s = 0; s < pg->n_steps; s++) {
        const or_step* st = &pg->steps[s];
        switch (st->op) {
            case ST_CONST: vals[nf++] = fp_from_u32(st->a % OR_P); break;
            case ST_GET: vals[nf++] = u[st->a]; break;
            case ST_GET_GLOBAL:
                vals[nf++] = st->a == 0 ? (st->b < n_globals ? globals[st->b] : 0) : (st->b < n_mix ? mix[st->b] : 0);
                break;
            case ST_ADD: vals[nf] = fp_add(vals[st->a], vals[st->b]); nf++; break;
            case ST_SUB: vals[nf] = fp_sub(vals[st->a], vals[st->b]); nf++; break;
            case ST_MUL: vals[nf] = fp_mul(vals[st->a], vals[st->b]); nf++; break;
            case ST_TRUE: ms[nm].tot = fp4_zero(); ms[nm].mul = fp4_one(); nm++; break;
            case ST_AND_EQZ: {
                mix_state x = ms[st->a];
                ms[nm].tot = fp4_add(x.tot, fp4_scale(x.mul, vals[st->b]));
                ms[nm].mul = fp4_mul(x.mul, poly_mix);
                nm++;
                break;
            }
            default: {
                mix_state x = ms[st->a], in = ms[st->c];
                ms[nm].tot = fp4_add(x.tot, fp4_mul(fp4_scale(in.tot, vals[st->b]), x.mul));
                ms[nm].mul = fp4_mul(x.mul, in.mul);
                nm++;
            }
        }
    }
    return ms[pg->ret].tot;
}

int or_program_eval_check(void* user, const or_circuit_view* v, const fp* poly_mix, fp* check) {
    const or_program* pg = (const or_program*)user;
    if (!pg || !pg->taps) return 1;
    const or_taps* t = pg->taps;
    size_t n_taps = count_taps(t);
    if (program_check(pg, n_taps)) return 2;
    size_t N = (size_t)1 << v->po2, D = N * OR_INV_RATE;
    /* tap -> (column base, shift on the LDE domain) */
    const fp** col = (const fp**)malloc((n_taps + 1) * sizeof(*col));
    size_t* shift = (size_t*)malloc((n_taps + 1) * sizeof(*shift));
    size_t pos = 0;
    for (uint32_t r = 0; r < t->n_regs; r++) {
        uint32_t cb = t->reg_combo[r], g = t->reg_group[r];
        for (uint32_t b = t->combo_off[cb]; b < t->combo_off[cb + 1]; b++, pos++) {
            col[pos] = (v->lde[g] && t->reg_offset[r] < v->group_size[g]) ? v->lde[g] + (size_t)t->reg_offset[r] * D : NULL;
            shift[pos] = ((size_t)t->combo_backs[b] * OR_INV_RATE) % D;
        }
    }
    fp4 pm;
    memcpy(&pm, poly_mix, 16);
    fp inv_den[16];
    fp shift_n = fp_pow(fp_from_u32(g_or.coset_shift), N), w4 = or_rou_fwd(OR_INV_RATE_PO2);
    for (size_t r = 0; r < OR_INV_RATE; r++) inv_den[r] = fp_inv(fp_sub(fp_mul(shift_n, fp_pow(w4, r)), fp_from_u32(1)));
    int bad = 0;
#pragma omp parallel
    {
        fp* u = (fp*)malloc((n_taps + 1) * sizeof(fp));
        fp* vals = (fp*)malloc((pg->n_steps + 1) * sizeof(fp));
        mix_state* ms = (mix_state*)malloc((pg->n_steps + 1) * sizeof(mix_state));
#pragma omp for schedule(static)
        for (size_t i = 0; i < D; i++) {
            for (size_t k = 0; k < n_taps; k++) u[k] = col[k] ? col[k][(i + D - shift[k]) % D] : 0;
            fp4 tot = run_fp(pg, u, v->globals, v->n_globals, v->mix, v->n_mix, pm, vals, ms);
            tot = fp4_scale(tot, inv_den[i & (OR_INV_RATE - 1)]);
            for (int e = 0; e < 4; e++) check[(size_t)e * D + i] = tot.c[e];
        }
        free(u); free(vals); free(ms);
    }
    free(col); free(shift);
    return bad;
}

/* or_poly_ext_fn with user = the or_program: the same list on extension elements */
int or_program_poly_ext(void* user, const or_segment* pub, const fp* poly_mix, const fp4* eval_u, size_t n_taps,
                        const fp* mix, uint32_t n_mix, fp* out) {
    const or_program* pg = (const or_program*)user;
    if (!pg || program_check(pg, n_taps)) return 1;
    fp4 pm;
    memcpy(&pm, poly_mix, 16);
    fp4* vals = (fp4*)malloc((pg->n_steps + 1) * sizeof(fp4));
    mix_state* ms = (mix_state*)malloc((pg->n_steps + 1) * sizeof(mix_state));
    size_t nf = 0, nm = 0;
    for (size_t s = 0; s < pg->n_steps; s++) {
        const or_step* st = &pg->steps[s];
        switch (st->op) {
            case ST_CONST: vals[nf++] = fp4_from_fp(fp_from_u32(st->a % OR_P)); break;
            case ST_GET: vals[nf++] = eval_u[st->a]; break;
            case ST_GET_GLOBAL:
                vals[nf++] = fp4_from_fp(st->a == 0 ? (st->b < pub->n_globals ? pub->globals[st->b] : 0)
                                                    : (st->b < n_mix ? mix[st->b] : 0));
                break;
            case ST_ADD: vals[nf] = fp4_add(vals[st->a], vals[st->b]); nf++; break;
            case ST_SUB: vals[nf] = fp4_sub(vals[st->a], vals[st->b]); nf++; break;
            case ST_MUL: vals[nf] = fp4_mul(vals[st->a], vals[st->b]); nf++; break;
            case ST_TRUE: ms[nm].tot = fp4_zero(); ms[nm].mul = fp4_one(); nm++; break;
            case ST_AND_EQZ: {
                mix_state x = ms[st->a];
                ms[nm].tot = fp4_add(x.tot, fp4_mul(x.mul, vals[st->b]));
                ms[nm].mul = fp4_mul(x.mul, pm);
                nm++;
                break;
            }
            default: {
                mix_state x = ms[st->a], in = ms[st->c];
                ms[nm].tot = fp4_add(x.tot, fp4_mul(fp4_mul(vals[st->b], in.tot), x.mul));
                ms[nm].mul = fp4_mul(x.mul, in.mul);
                nm++;
            }
        }
    }
    memcpy(out, &ms[pg->ret].tot, 16);
    free(vals); free(ms);
    return 0;
}
