/* TEST INFRASTRUCTURE -- CPU oracle (see oracle.h header: parity unpinned).
 *
 * The timed CPU baseline: the same operators as or_ops.c (risc0-zkp 1.0.1 hal/cpu.rs behind
 * `session.prove()`, reference provers/risc0/driver/src/bonsai.rs:271), written the way an
 * optimised CPU prover writes them -- 8-lane AVX2 Montgomery arithmetic, Poseidon2 over 8 rows /
 * 8 parents at a time, table-driven vectorised NTT butterflies, tap evaluation and DEEP mixing
 * through shared power tables -- and threaded with OpenMP.  or_set_fast(1) routes the hot
 * entry points of or_ops.c here; results are bit-identical to the plain restatement
 * (tests/test_oracle_fast.py), which stays the reference the GPU is compared with. */
#include "oracle.h"
#include <immintrin.h>
#define FAST_CELLS 24 /* the AVX2 permutation serves the width-24 instances only (or_ops.c dispatches) */
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int g_or_fast = 0;
void or_set_fast(int on) { g_or_fast = on; }
int or_get_fast(void) { return g_or_fast; }

/* ------------------------------------------------------------ 8-lane field */
typedef __m256i v8;
static inline v8 v_p(void) { return _mm256_set1_epi32((int)OR_P); }
static inline v8 v_add(v8 a, v8 b) {
    v8 r = _mm256_add_epi32(a, b);
    return _mm256_min_epu32(r, _mm256_sub_epi32(r, v_p()));
}
static inline v8 v_sub(v8 a, v8 b) {
    v8 r = _mm256_sub_epi32(a, b);
    return _mm256_min_epu32(r, _mm256_add_epi32(r, v_p()));
}
/* Montgomery product of canonical lanes: t = hi(a*b) - hi(q*p), q = lo(a*b) * p^-1; t in (-p, p) */
static inline v8 v_mul(v8 a, v8 b) {
    const v8 mu = _mm256_set1_epi32((int)OR_M), p = v_p();
    v8 a_odd = _mm256_srli_epi64(a, 32), b_odd = _mm256_srli_epi64(b, 32);
    v8 pe = _mm256_mul_epu32(a, b), po = _mm256_mul_epu32(a_odd, b_odd);
    v8 qe = _mm256_mul_epu32(pe, mu), qo = _mm256_mul_epu32(po, mu);
    v8 qpe = _mm256_mul_epu32(qe, p), qpo = _mm256_mul_epu32(qo, p);
    v8 hi = _mm256_blend_epi32(_mm256_srli_epi64(pe, 32), po, 0xaa);
    v8 qhi = _mm256_blend_epi32(_mm256_srli_epi64(qpe, 32), qpo, 0xaa);
    v8 t = _mm256_sub_epi32(hi, qhi);
    return _mm256_min_epu32(t, _mm256_add_epi32(t, p));
}
static inline v8 v_set1(fp x) { return _mm256_set1_epi32((int)x); }
static inline v8 v_load(const fp* p) { return _mm256_loadu_si256((const __m256i*)p); }
static inline void v_store(fp* p, v8 v) { _mm256_storeu_si256((__m256i*)p, v); }
static inline v8 v_sbox7(v8 x) {
    v8 x2 = v_mul(x, x), x4 = v_mul(x2, x2), x6 = v_mul(x4, x2);
    return v_mul(x6, x);
}

/* ------------------------------------------------------------- Poseidon2 x8 */
static void v_m_ext(v8* c) {
    v8 sums[4];
    for (int j = 0; j < 4; j++) sums[j] = _mm256_setzero_si256();
    for (int i = 0; i < FAST_CELLS; i += 4) {
        v8 a = c[i], b = c[i + 1], d = c[i + 2], e = c[i + 3];
        v8 t0 = v_add(a, b), t1 = v_add(d, e);
        v8 t2 = v_add(v_add(b, b), t1), t3 = v_add(v_add(e, e), t0);
        v8 t1_2 = v_add(t1, t1), t0_2 = v_add(t0, t0);
        v8 t4 = v_add(v_add(t1_2, t1_2), t3), t5 = v_add(v_add(t0_2, t0_2), t2);
        v8 t6 = v_add(t3, t5), t7 = v_add(t2, t4);
        c[i] = t6; c[i + 1] = t5; c[i + 2] = t7; c[i + 3] = t4;
        for (int j = 0; j < 4; j++) sums[j] = v_add(sums[j], c[i + j]);
    }
    for (int i = 0; i < FAST_CELLS; i++) c[i] = v_add(c[i], sums[i & 3]);
}
static void v_poseidon2(v8* c) {
    int r = 0;
    v_m_ext(c);
    for (int k = 0; k < 4; k++, r++) {
        for (int i = 0; i < FAST_CELLS; i++) c[i] = v_sbox7(v_add(c[i], v_set1(g_or.p2_rc_ext[r * FAST_CELLS + i])));
        v_m_ext(c);
    }
    for (int k = 0; k < 21; k++) {
        c[0] = v_sbox7(v_add(c[0], v_set1(g_or.p2_rc_int[k])));
        v8 sum = c[0];
        for (int i = 1; i < FAST_CELLS; i++) sum = v_add(sum, c[i]);
        for (int i = 0; i < FAST_CELLS; i++) c[i] = v_add(sum, v_mul(c[i], v_set1(g_or.p2_diag[i])));
    }
    for (int k = 0; k < 4; k++, r++) {
        for (int i = 0; i < FAST_CELLS; i++) c[i] = v_sbox7(v_add(c[i], v_set1(g_or.p2_rc_ext[r * FAST_CELLS + i])));
        v_m_ext(c);
    }
}

/* sponge over 8 consecutive rows of a column-major matrix: lane = row */
static void hash_rows8(uint32_t* out, const fp* matrix, size_t rows, size_t cols, size_t r0) {
    v8 st[FAST_CELLS];
    for (int i = 0; i < FAST_CELLS; i++) st[i] = _mm256_setzero_si256();
    size_t unmixed = 0;
    for (size_t c = 0; c < cols; c++) {
        st[unmixed++] = v_load(matrix + c * rows + r0);
        if (unmixed == 16) { v_poseidon2(st); unmixed = 0; }
    }
    if (unmixed != 0 || cols == 0) {
        for (size_t i = unmixed; i < 16; i++) st[i] = _mm256_setzero_si256();
        v_poseidon2(st);
    }
    uint32_t tmp[OR_CELLS_OUT][8];
    for (int w = 0; w < OR_CELLS_OUT; w++) v_store(tmp[w], st[w]);
    for (int l = 0; l < 8; l++)
        for (int w = 0; w < OR_CELLS_OUT; w++) out[(r0 + l) * OR_DIGEST_WORDS + w] = tmp[w][l];
}
void or_fast_hash_rows(uint32_t* out, const fp* matrix, size_t rows, size_t cols) {
    size_t blocks = rows / 8;
#pragma omp parallel for schedule(static)
    for (size_t b = 0; b < blocks; b++) hash_rows8(out, matrix, rows, cols, b * 8);
    for (size_t r = blocks * 8; r < rows; r++) or_hash_elem_slice(matrix + r, cols, rows, out + r * OR_DIGEST_WORDS);
}
/* 8 parents at a time: lane = parent, the 16 child words of a parent are contiguous */
void or_fast_hash_fold(uint32_t* nodes, size_t output_size) {
    size_t blocks = output_size / 8;
    const v8 lane_off = _mm256_setr_epi32(0, 16, 32, 48, 64, 80, 96, 112);
#pragma omp parallel for schedule(static)
    for (size_t b = 0; b < blocks; b++) {
        size_t idx = output_size + b * 8;
        const int* base = (const int*)(nodes + 2 * idx * OR_DIGEST_WORDS);
        v8 st[FAST_CELLS];
        for (int w = 0; w < 16; w++) st[w] = _mm256_i32gather_epi32(base + w, lane_off, 4);
        for (int w = 16; w < FAST_CELLS; w++) st[w] = _mm256_setzero_si256();
        v_poseidon2(st);
        uint32_t tmp[OR_CELLS_OUT][8];
        for (int w = 0; w < OR_CELLS_OUT; w++) v_store(tmp[w], st[w]);
        for (int l = 0; l < 8; l++)
            for (int w = 0; w < OR_CELLS_OUT; w++) nodes[(idx + l) * OR_DIGEST_WORDS + w] = tmp[w][l];
    }
    for (size_t i = blocks * 8; i < output_size; i++) {
        size_t idx = output_size + i;
        or_hash_pair(nodes + (2 * idx) * OR_DIGEST_WORDS, nodes + (2 * idx + 1) * OR_DIGEST_WORDS,
                     nodes + idx * OR_DIGEST_WORDS);
    }
}

/* -------------------------------------------------------------------- NTT */
/* twiddle tables: level n (butterflies of span 2^(n-1)) at offset 2^(n-1): w_n^(+-i), i < 2^(n-1) */
#define FAST_MAX_LOG 24
static fp* g_tw_fwd = NULL;
static fp* g_tw_rev = NULL;
static unsigned g_tw_log = 0;
void or_fast_reset_tables(void) { g_tw_log = 0; } /* after or_set_params: the old tables are abandoned */
static void tw_init(unsigned k) {
    if (k <= g_tw_log) return;
#pragma omp critical(or_fast_tw)
    {
        if (k > g_tw_log) {
            fp* f = (fp*)malloc(((size_t)1 << k) * sizeof(fp));
            fp* r = (fp*)malloc(((size_t)1 << k) * sizeof(fp));
            f[0] = r[0] = 0;
            for (unsigned n = 1; n <= k; n++) {
                size_t half = (size_t)1 << (n - 1);
                fp wf = or_rou_fwd(n), wr = or_rou_rev(n), cf = fp_from_u32(1), cr = cf;
                for (size_t i = 0; i < half; i++) {
                    f[half + i] = cf; r[half + i] = cr;
                    cf = fp_mul(cf, wf); cr = fp_mul(cr, wr);
                }
            }
            /* tables of an earlier, smaller size stay allocated: other threads may still read them */
            g_tw_fwd = f; g_tw_rev = r;
            __sync_synchronize();
            g_tw_log = k;
        }
    }
}
static void fast_rev(fp* io, unsigned n, const fp* tw) {
    if (n == 0) return;
    size_t half = (size_t)1 << (n - 1);
    const fp* t = tw + half;
    if (half >= 8) {
        for (size_t i = 0; i < half; i += 8) {
            v8 a = v_load(io + i), b = v_load(io + i + half);
            v_store(io + i, v_add(a, b));
            v_store(io + i + half, v_mul(v_sub(a, b), v_load(t + i)));
        }
    } else {
        for (size_t i = 0; i < half; i++) {
            fp a = io[i], b = io[i + half];
            io[i] = fp_add(a, b);
            io[i + half] = fp_mul(fp_sub(a, b), t[i]);
        }
    }
    fast_rev(io, n - 1, tw);
    fast_rev(io + half, n - 1, tw);
}
static void fast_fwd(fp* io, unsigned n, unsigned expand_bits, const fp* tw) {
    if (n == 0) return;
    if (n == expand_bits) {
        size_t sz = (size_t)1 << n;
        for (size_t i = 1; i < sz; i++) io[i] = io[0];
        return;
    }
    size_t half = (size_t)1 << (n - 1);
    fast_fwd(io, n - 1, expand_bits, tw);
    fast_fwd(io + half, n - 1, expand_bits, tw);
    const fp* t = tw + half;
    if (half >= 8) {
        for (size_t i = 0; i < half; i += 8) {
            v8 a = v_load(io + i), b = v_mul(v_load(io + i + half), v_load(t + i));
            v_store(io + i, v_add(a, b));
            v_store(io + i + half, v_sub(a, b));
        }
    } else {
        for (size_t i = 0; i < half; i++) {
            fp a = io[i], b = fp_mul(io[i + half], t[i]);
            io[i] = fp_add(a, b);
            io[i + half] = fp_sub(a, b);
        }
    }
}
static unsigned lg(size_t n) { unsigned k = 0; while (((size_t)1 << k) < n) k++; return k; }
static void scale_all(fp* io, size_t n, fp s) {
    size_t i = 0;
    v8 vs = v_set1(s);
    for (; i + 8 <= n; i += 8) v_store(io + i, v_mul(v_load(io + i), vs));
    for (; i < n; i++) io[i] = fp_mul(io[i], s);
}
void or_fast_batch_interpolate_ntt(fp* io, size_t size, size_t count) {
    unsigned k = lg(size);
    tw_init(k);
    const fp* tw = g_tw_rev;
    fp norm = fp_inv(fp_from_u32((uint32_t)size));
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < count; c++) {
        fast_rev(io + c * size, k, tw);
        scale_all(io + c * size, size, norm);
    }
}
void or_fast_batch_expand_into_evaluate_ntt(fp* out, const fp* in, size_t in_size, size_t count, unsigned expand_bits) {
    size_t out_size = in_size << expand_bits;
    unsigned k = lg(out_size);
    tw_init(k);
    const fp* tw = g_tw_fwd;
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < count; c++) {
        fp* o = out + c * out_size;
        const fp* s = in + c * in_size;
        for (size_t i = 0; i < out_size; i++) o[i] = s[i >> expand_bits];
        fast_fwd(o, k, expand_bits, tw);
    }
}
/* io[pos] *= 3^bitrev(pos): one table of the factors shared by all columns */
void or_fast_zk_shift(fp* io, size_t size, size_t count) {
    unsigned bits = lg(size);
    fp* f = (fp*)malloc(size * sizeof(fp));
    fp pw[32];
    pw[0] = fp_from_u32(g_or.coset_shift);
    for (int j = 1; j < 32; j++) pw[j] = fp_mul(pw[j - 1], pw[j - 1]);
    /* f[pos] for pos = b_{bits-1}..b_0 has exponent rev(pos): bit j of pos contributes 3^(2^(bits-1-j)) */
    f[0] = fp_from_u32(1);
    for (unsigned j = 0; j < bits; j++) {
        size_t span = (size_t)1 << j;
        fp m = pw[bits - 1 - j];
        for (size_t i = 0; i < span; i++) f[span + i] = fp_mul(f[i], m);
    }
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < count; c++) {
        fp* col = io + c * size;
        size_t i = 0;
        for (; i + 8 <= size; i += 8) v_store(col + i, v_mul(v_load(col + i), v_load(f + i)));
        for (; i < size; i++) col[i] = fp_mul(col[i], f[i]);
    }
    free(f);
}

/* ----------------------------------------------------- DEEP-side operators */
/* out[e] = sum_k coeffs[which[e]][k] * xs[e]^k through one power table per distinct point */
void or_fast_batch_evaluate_any(const fp* coeffs, size_t size, const uint32_t* which, const fp4* xs,
                                size_t eval_count, fp4* out) {
    if (eval_count == 0) return;
    uint32_t* sel = (uint32_t*)malloc(eval_count * sizeof(uint32_t));
    fp4* pts = (fp4*)malloc(eval_count * sizeof(fp4));
    size_t n_pts = 0;
    for (size_t e = 0; e < eval_count; e++) {
        size_t j = 0;
        for (; j < n_pts; j++) if (fp4_eq(pts[j], xs[e])) break;
        if (j == n_pts) pts[n_pts++] = xs[e];
        sel[e] = (uint32_t)j;
    }
    /* power tables as 4 planes per point (component-major) so the dot products vectorise */
    fp* pw = (fp*)malloc(n_pts * 4 * size * sizeof(fp));
    const size_t CH = 1024;
#pragma omp parallel for schedule(static) collapse(2)
    for (size_t j = 0; j < n_pts; j++)
        for (size_t c0 = 0; c0 < size; c0 += CH) {
            fp4 cur = fp4_pow(pts[j], c0);
            size_t c1 = c0 + CH < size ? c0 + CH : size;
            for (size_t k = c0; k < c1; k++) {
                for (int e = 0; e < 4; e++) pw[(j * 4 + e) * size + k] = cur.c[e];
                cur = fp4_mul(cur, pts[j]);
            }
        }
#pragma omp parallel for schedule(dynamic)
    for (size_t e = 0; e < eval_count; e++) {
        const fp* c = coeffs + (size_t)which[e] * size;
        fp4 tot = fp4_zero();
        for (int comp = 0; comp < 4; comp++) {
            const fp* p = pw + ((size_t)sel[e] * 4 + comp) * size;
            v8 acc = _mm256_setzero_si256();
            size_t k = 0;
            for (; k + 8 <= size; k += 8) acc = v_add(acc, v_mul(v_load(c + k), v_load(p + k)));
            fp lanes[8];
            v_store(lanes, acc);
            fp s = 0;
            for (int l = 0; l < 8; l++) s = fp_add(s, lanes[l]);
            for (; k < size; k++) s = fp_add(s, fp_mul(c[k], p[k]));
            tot.c[comp] = s;
        }
        out[e] = tot;
    }
    free(pw); free(pts); free(sel);
}
/* out[combos[i]][idx] += mix_start * mix^i * in[i][idx] with the powers computed once */
void or_fast_mix_poly_coeffs(fp4* out, const uint32_t* mix_start, const uint32_t* mix, const fp* in,
                             const uint32_t* combos, size_t input_size, size_t count) {
    fp4 ms, mx; memcpy(&ms, mix_start, 16); memcpy(&mx, mix, 16);
    fp4* pw = (fp4*)malloc((input_size + 1) * sizeof(fp4));
    fp4 cur = ms;
    for (size_t i = 0; i < input_size; i++) { pw[i] = cur; cur = fp4_mul(cur, mx); }
    const size_t CH = 4096;
#pragma omp parallel for schedule(static)
    for (size_t c0 = 0; c0 < count; c0 += CH) {
        size_t c1 = c0 + CH < count ? c0 + CH : count;
        for (size_t i = 0; i < input_size; i++) {
            fp4* o = out + (size_t)combos[i] * count;
            const fp* col = in + i * count;
            const fp4 w = pw[i];
            for (size_t idx = c0; idx < c1; idx++) {
                fp v = col[idx];
                for (int e = 0; e < 4; e++) o[idx].c[e] = fp_add(o[idx].c[e], fp_mul(w.c[e], v));
            }
        }
    }
    free(pw);
}
