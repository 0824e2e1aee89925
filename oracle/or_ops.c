/* TEST INFRASTRUCTURE -- CPU oracle (see oracle.h header: parity unpinned).
 * Restates risc0-zkp 1.0.1 core/ntt.rs, core/poly.rs, core/hash/poseidon2
 * and hal/cpu.rs (the Hal trait) -- the operators behind `session.prove()`
 * at /root/reference provers/risc0/driver/src/bonsai.rs:271. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* or_fast.c: the optimised forms used for the timed CPU baseline (bit-identical results) */
extern int g_or_fast;
void or_fast_hash_rows(uint32_t* out, const fp* matrix, size_t rows, size_t cols);
void or_fast_hash_fold(uint32_t* nodes, size_t output_size);
void or_fast_batch_interpolate_ntt(fp* io, size_t size, size_t count);
void or_fast_batch_expand_into_evaluate_ntt(fp* out, const fp* in, size_t in_size, size_t count, unsigned expand_bits);
void or_fast_zk_shift(fp* io, size_t size, size_t count);
void or_fast_batch_evaluate_any(const fp* coeffs, size_t size, const uint32_t* which, const fp4* xs, size_t eval_count,
                                fp4* out);
void or_fast_mix_poly_coeffs(fp4* out, const uint32_t* mix_start, const uint32_t* mix, const fp* in,
                             const uint32_t* combos, size_t input_size, size_t count);

/* ------------------------------------------------------------------ roots */
static fp g_rou_fwd[28], g_rou_rev[28];
static int g_rou_ready = 0;
void or_ops_reset_roots(void) { g_rou_ready = 0; }
static void rou_init(void) {
    if (g_rou_ready) return;
#pragma omp critical(or_rou)
    {
        if (!g_rou_ready) {
            g_rou_fwd[27] = fp_from_u32(g_or.root_2_27); /* generator of the 2^27 subgroup (137 for risc0) */
            for (int k = 26; k >= 0; k--) g_rou_fwd[k] = fp_mul(g_rou_fwd[k + 1], g_rou_fwd[k + 1]);
            for (int k = 0; k <= 27; k++) g_rou_rev[k] = fp_inv(g_rou_fwd[k]);
            g_rou_ready = 1;
        }
    }
}
uint32_t or_rou_fwd(unsigned k) { rou_init(); return g_rou_fwd[k]; }
uint32_t or_rou_rev(unsigned k) { rou_init(); return g_rou_rev[k]; }

uint32_t or_fp_mul(uint32_t a, uint32_t b) { return fp_mul(a, b); }
uint32_t or_fp_add(uint32_t a, uint32_t b) { return fp_add(a, b); }
uint32_t or_fp_sub(uint32_t a, uint32_t b) { return fp_sub(a, b); }
uint32_t or_fp_inv(uint32_t a) { return fp_inv(a); }
uint32_t or_fp_encode(uint32_t c) { return fp_from_u32(c); }
uint32_t or_fp_decode(uint32_t m) { return fp_to_u32(m); }
void or_fp4_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) {
    fp4 x, y; memcpy(&x, a, 16); memcpy(&y, b, 16);
    fp4 r = fp4_mul(x, y); memcpy(out, &r, 16);
}
void or_fp4_inv(const uint32_t* a, uint32_t* out) {
    fp4 x; memcpy(&x, a, 16); fp4 r = fp4_inv(x); memcpy(out, &r, 16);
}
/* team size of every later parallel region (the default is one thread per hardware thread of the host,
 * far more than a container's CPU share: oversubscribed teams spin) */
void or_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int or_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void or_free(void* p) { free(p); }

static unsigned log2_exact(size_t n) { unsigned k = 0; while (((size_t)1 << k) < n) k++; return k; }
static uint32_t bitrev32(uint32_t x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x >> 8) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8);
    return (x >> 16) | (x << 16);
}

/* ------------------------------------------------------------------ ntt.rs */
void or_bit_reverse(fp* io, size_t n) {
    unsigned k = log2_exact(n);
    if (k == 0) return;
    for (size_t i = 0; i < n; i++) {
        size_t r = bitrev32((uint32_t)i) >> (32 - k);
        if (i < r) { fp t = io[i]; io[i] = io[r]; io[r] = t; }
    }
}
/* decimation in frequency with inverse roots: out[i'] = sum_k in[k] w^-(k i) */
static void rev_butterfly(fp* io, unsigned n) {
    if (n == 0) return;
    size_t half = (size_t)1 << (n - 1);
    fp step = g_rou_rev[n], cur = fp_from_u32(1);
    for (size_t i = 0; i < half; i++) {
        fp a = io[i], b = io[i + half];
        io[i] = fp_add(a, b);
        io[i + half] = fp_mul(fp_sub(a, b), cur);
        cur = fp_mul(cur, step);
    }
    rev_butterfly(io, n - 1);
    rev_butterfly(io + half, n - 1);
}
/* decimation in time with forward roots; the lowest expand_bits levels see
 * zero-padded (bit-reversed) input and degenerate to a broadcast */
static void fwd_butterfly(fp* io, unsigned n, unsigned expand_bits) {
    if (n == 0) return;
    if (n == expand_bits) {
        size_t sz = (size_t)1 << n;
        for (size_t i = 1; i < sz; i++) io[i] = io[0];
        return;
    }
    size_t half = (size_t)1 << (n - 1);
    fwd_butterfly(io, n - 1, expand_bits);
    fwd_butterfly(io + half, n - 1, expand_bits);
    fp step = g_rou_fwd[n], cur = fp_from_u32(1);
    for (size_t i = 0; i < half; i++) {
        fp a = io[i], b = fp_mul(io[i + half], cur);
        io[i] = fp_add(a, b);
        io[i + half] = fp_sub(a, b);
        cur = fp_mul(cur, step);
    }
}
void or_interpolate_ntt(fp* io, size_t n) {
    rou_init();
    unsigned k = log2_exact(n);
    rev_butterfly(io, k);
    fp norm = fp_inv(fp_from_u32((uint32_t)n));
    for (size_t i = 0; i < n; i++) io[i] = fp_mul(io[i], norm);
}
void or_evaluate_ntt(fp* io, size_t n, unsigned expand_bits) {
    rou_init();
    fwd_butterfly(io, log2_exact(n), expand_bits);
}

/* ------------------------------------------------------- poseidon2/mod.rs */
static inline fp sbox7(fp x) {
    fp x2 = fp_mul(x, x), x4 = fp_mul(x2, x2), x6 = fp_mul(x4, x2);
    return fp_mul(x6, x);
}
/* external layer circ(2*M4, M4, ..., M4) over width/4 blocks; M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]]
 * (Poseidon2 paper, risc0) or circ(2,3,1,1) (Plonky3 MDSMat4) */
static void m_ext(fp* c) {
    const int W = (int)g_or.p2_width;
    fp sums[4] = {0, 0, 0, 0};
    for (int i = 0; i < W; i += 4) {
        fp a = c[i], b = c[i + 1], d = c[i + 2], e = c[i + 3];
        if (g_or.p2_m4 == 0) {
            fp t0 = fp_add(a, b), t1 = fp_add(d, e);
            fp t2 = fp_add(fp_add(b, b), t1), t3 = fp_add(fp_add(e, e), t0);
            fp t1_4 = fp_add(fp_add(t1, t1), fp_add(t1, t1));
            fp t0_4 = fp_add(fp_add(t0, t0), fp_add(t0, t0));
            fp t4 = fp_add(t1_4, t3), t5 = fp_add(t0_4, t2);
            fp t6 = fp_add(t3, t5), t7 = fp_add(t2, t4);
            c[i] = t6; c[i + 1] = t5; c[i + 2] = t7; c[i + 3] = t4;
        } else {
            fp s = fp_add(fp_add(a, b), fp_add(d, e));
            c[i] = fp_add(fp_add(s, a), fp_add(b, b));      /* 2a + 3b +  c +  d */
            c[i + 1] = fp_add(fp_add(s, b), fp_add(d, d));  /*  a + 2b + 3c +  d */
            c[i + 2] = fp_add(fp_add(s, d), fp_add(e, e));  /*  a +  b + 2c + 3d */
            c[i + 3] = fp_add(fp_add(s, e), fp_add(a, a));  /* 3a +  b +  c + 2d */
        }
        for (int j = 0; j < 4; j++) sums[j] = fp_add(sums[j], c[i + j]);
    }
    for (int i = 0; i < W; i++) c[i] = fp_add(c[i], sums[i & 3]);
}
static void m_int(fp* c) {
    const int W = (int)g_or.p2_width;
    fp sum = 0;
    for (int i = 0; i < W; i++) sum = fp_add(sum, c[i]);
    for (int i = 0; i < W; i++) c[i] = fp_add(sum, fp_mul(c[i], g_or.p2_diag[i]));
}
void or_poseidon2_mix(fp* c) {
    const int W = (int)g_or.p2_width, RP = W == 16 ? 13 : 21;
    int r = 0;
    m_ext(c);
    for (int k = 0; k < 4; k++, r++) {
        for (int i = 0; i < W; i++) c[i] = sbox7(fp_add(c[i], g_or.p2_rc_ext[r * W + i]));
        m_ext(c);
    }
    for (int k = 0; k < RP; k++) {
        c[0] = sbox7(fp_add(c[0], g_or.p2_rc_int[k]));
        m_int(c);
    }
    for (int k = 0; k < 4; k++, r++) {
        for (int i = 0; i < W; i++) c[i] = sbox7(fp_add(c[i], g_or.p2_rc_ext[r * W + i]));
        m_ext(c);
    }
}
/* overwrite-mode sponge over the configured rate; the last partial block is zero-padded (risc0) or
 * leaves the remaining rate cells as they are (Plonky3 PaddingFreeSponge); one permutation for an
 * empty input in the zero-padded mode */
void or_hash_elem_slice(const fp* in, size_t n, size_t stride, uint32_t* digest) {
    fp st[OR_MAX_CELLS];
    memset(st, 0, sizeof st);
    size_t unmixed = 0;
    for (size_t i = 0; i < n; i++) {
        st[unmixed++] = in[i * stride];
        if (unmixed == OR_CELLS_RATE) { or_poseidon2_mix(st); unmixed = 0; }
    }
    if (unmixed != 0 || (n == 0 && !g_or.p2_pad_free)) {
        if (!g_or.p2_pad_free)
            for (size_t i = unmixed; i < OR_CELLS_RATE; i++) st[i] = 0;
        or_poseidon2_mix(st);
    }
    memcpy(digest, st, OR_CELLS_OUT * sizeof(fp));
}
void or_hash_pair(const uint32_t* a, const uint32_t* b, uint32_t* out) {
    fp st[OR_MAX_CELLS];
    memcpy(st, a, 32); memcpy(st + 8, b, 32); memset(st + 16, 0, 32); /* width 16: the two digests fill the state */
    or_poseidon2_mix(st);
    memcpy(out, st, 32);
}

/* --------------------------------------------------------------- hal/cpu.rs */
void or_batch_interpolate_ntt(fp* io, size_t size, size_t count) {
    rou_init();
    if (g_or_fast) { or_fast_batch_interpolate_ntt(io, size, count); return; }
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < count; c++) or_interpolate_ntt(io + c * size, size);
}
void or_batch_evaluate_ntt(fp* io, size_t size, size_t count, unsigned expand_bits) {
    rou_init();
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < count; c++) or_evaluate_ntt(io + c * size, size, expand_bits);
}
void or_zk_shift(fp* io, size_t size, size_t count) {
    if (g_or_fast) { or_fast_zk_shift(io, size, count); return; }
    unsigned bits = log2_exact(size);
    fp three = fp_from_u32(g_or.coset_shift);
    /* shift^rev(pos): table of shift^(2^j) */
    fp pw[32];
    pw[0] = three;
    for (int j = 1; j < 32; j++) pw[j] = fp_mul(pw[j - 1], pw[j - 1]);
#pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < size * count; idx++) {
        size_t pos = idx & (size - 1);
        uint32_t rev = bits ? bitrev32((uint32_t)pos) >> (32 - bits) : 0;
        fp f = fp_from_u32(1);
        for (unsigned j = 0; j < bits; j++) if ((rev >> j) & 1) f = fp_mul(f, pw[j]);
        io[idx] = fp_mul(io[idx], f);
    }
}
void or_batch_expand_into_evaluate_ntt(fp* out, const fp* in, size_t in_size, size_t count, unsigned expand_bits) {
    rou_init();
    if (g_or_fast) { or_fast_batch_expand_into_evaluate_ntt(out, in, in_size, count, expand_bits); return; }
    size_t out_size = in_size << expand_bits;
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < count; c++) {
        fp* o = out + c * out_size;
        const fp* s = in + c * in_size;
        for (size_t i = 0; i < out_size; i++) o[i] = s[i >> expand_bits];
        or_evaluate_ntt(o, out_size, expand_bits);
    }
}
void or_batch_bit_reverse(fp* io, size_t size, size_t count) {
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < count; c++) or_bit_reverse(io + c * size, size);
}
void or_hash_rows(uint32_t* out, const fp* matrix, size_t rows, size_t cols) {
    if (g_or_fast && g_or.p2_width == 24 && g_or.p2_m4 == 0 && !g_or.p2_pad_free) { or_fast_hash_rows(out, matrix, rows, cols); return; }
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < rows; r++) or_hash_elem_slice(matrix + r, cols, rows, out + r * OR_DIGEST_WORDS);
}
void or_hash_fold(uint32_t* nodes, size_t input_size, size_t output_size) {
    (void)input_size; /* == 2*output_size; heap layout: children of i are 2i, 2i+1 */
    if (g_or_fast && g_or.p2_width == 24 && g_or.p2_m4 == 0) { or_fast_hash_fold(nodes, output_size); return; }
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < output_size; i++) {
        size_t idx = output_size + i;
        or_hash_pair(nodes + (2 * idx) * OR_DIGEST_WORDS, nodes + (2 * idx + 1) * OR_DIGEST_WORDS,
                     nodes + idx * OR_DIGEST_WORDS);
    }
}
void or_batch_evaluate_any(const fp* coeffs, size_t size, const uint32_t* which, const fp4* xs,
                           size_t eval_count, fp4* out) {
    if (g_or_fast) { or_fast_batch_evaluate_any(coeffs, size, which, xs, eval_count, out); return; }
#pragma omp parallel for schedule(dynamic)
    for (size_t e = 0; e < eval_count; e++) {
        const fp* c = coeffs + (size_t)which[e] * size;
        fp4 x = xs[e], tot = fp4_zero();
        for (size_t i = size; i-- > 0;) tot = fp4_add(fp4_mul(tot, x), fp4_from_fp(c[i]));
        out[e] = tot;
    }
}
void or_mix_poly_coeffs(fp4* out, const uint32_t* mix_start, const uint32_t* mix, const fp* in,
                        const uint32_t* combos, size_t input_size, size_t count) {
    if (g_or_fast) { or_fast_mix_poly_coeffs(out, mix_start, mix, in, combos, input_size, count); return; }
    fp4 ms, mx; memcpy(&ms, mix_start, 16); memcpy(&mx, mix, 16);
#pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < count; idx++) {
        fp4 cur = ms;
        for (size_t i = 0; i < input_size; i++) {
            fp4* o = &out[(size_t)combos[i] * count + idx];
            *o = fp4_add(*o, fp4_scale(cur, in[i * count + idx]));
            cur = fp4_mul(cur, mx);
        }
    }
}
void or_eltwise_add_elem(fp* out, const fp* a, const fp* b, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = fp_add(a[i], b[i]);
}
void or_eltwise_sum_extelem(fp* out, const fp4* in, size_t count, size_t to_add) {
#pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < count; idx++) {
        fp4 tot = fp4_zero();
        for (size_t j = 0; j < to_add; j++) tot = fp4_add(tot, in[j * count + idx]);
        for (int k = 0; k < 4; k++) out[k * count + idx] = tot.c[k];
    }
}
void or_eltwise_copy_elem(fp* out, const fp* in, size_t n) { memcpy(out, in, n * sizeof(fp)); }
void or_eltwise_zeroize_elem(fp* io, size_t n) {
    for (size_t i = 0; i < n; i++) if (io[i] == OR_INVALID) io[i] = 0;
}
void or_fri_fold(fp* out, const fp* in, size_t count, const uint32_t* mix) {
    fp4 mx; memcpy(&mx, mix, 16);
    const unsigned fold_po2 = g_or.fri_fold_log2, fold = 1u << fold_po2;
#pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < count; idx++) {
        fp4 tot = fp4_zero(), cur = fp4_one();
        for (unsigned i = 0; i < fold; i++) {
            size_t rev_i = bitrev32(i) >> (32 - fold_po2);
            size_t rev_idx = rev_i * count + idx;
            fp4 f;
            for (int k = 0; k < 4; k++) f.c[k] = in[(size_t)k * count * fold + rev_idx];
            tot = fp4_add(tot, fp4_mul(cur, f));
            cur = fp4_mul(cur, mx);
        }
        for (int k = 0; k < 4; k++) out[(size_t)k * count + idx] = tot.c[k];
    }
}
/* Plonky3 p3-fri fold_even_odd (RECALLED): evaluations of p over the subgroup of order 2 * n_out, bit-reversed,
 * interleaved extension elements -> evaluations of p_even + beta p_odd over the squared subgroup, bit-reversed */
void or_fri_fold_evals(fp4* out, const fp4* in, size_t n_out, const uint32_t* beta) {
    fp4 b; memcpy(&b, beta, 16);
    unsigned k = log2_exact(2 * n_out);
    fp half = fp_inv(fp_from_u32(2)), ginv = or_rou_rev(k);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n_out; i++) {
        size_t j = k > 1 ? (bitrev32((uint32_t)i) >> (32 - (k - 1))) : 0;
        fp xinv = fp_pow(ginv, j);
        fp4 even = fp4_scale(fp4_add(in[2 * i], in[2 * i + 1]), half);
        fp4 odd = fp4_scale(fp4_sub(in[2 * i], in[2 * i + 1]), fp_mul(half, xinv));
        out[i] = fp4_add(even, fp4_mul(b, odd));
    }
}
void or_gather_sample(fp* dst, const fp* src, size_t idx, size_t size, size_t stride) {
    for (size_t g = 0; g < size; g++) dst[g] = src[g * stride + idx];
}

/* hal/cpu.rs prefix_products: sequential running product */
void or_prefix_products(fp4* io, size_t count) {
    for (size_t i = 1; i < count; i++) io[i] = fp4_mul(io[i], io[i - 1]);
}
/* hal/cpu.rs scatter: per-cycle ranges [index[c], index[c+1]) of (offset, value) pairs, in order */
void or_scatter(fp* into, const uint32_t* index, size_t n_cycles, const uint32_t* offsets, const fp* values) {
    for (size_t c = 0; c < n_cycles; c++)
        for (uint32_t k = index[c]; k < index[c + 1]; k++) into[offsets[k]] = values[k];
}

/* ------------------------------------------------------------------ poly.rs */
void or_poly_eval(const fp4* coeffs, size_t n, const uint32_t* x, uint32_t* out) {
    fp4 xx; memcpy(&xx, x, 16);
    fp4 tot = fp4_zero();
    for (size_t i = n; i-- > 0;) tot = fp4_add(fp4_mul(tot, xx), coeffs[i]);
    memcpy(out, &tot, 16);
}
/* Lagrange interpolation through n points, coefficient output (n is tiny) */
void or_poly_interpolate(fp4* out, const fp4* x, const fp4* fx, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = fp4_zero();
    fp4* num = (fp4*)malloc((n + 1) * sizeof(fp4));
    for (size_t i = 0; i < n; i++) {
        /* num(X) = prod_{j!=i} (X - x_j), denom = prod_{j!=i} (x_i - x_j) */
        size_t deg = 0;
        fp4 denom = fp4_one();
        num[0] = fp4_one();
        for (size_t j = 0; j < n; j++) {
            if (j == i) continue;
            num[deg + 1] = num[deg];
            for (size_t k = deg; k > 0; k--) num[k] = fp4_sub(num[k - 1], fp4_mul(num[k], x[j]));
            num[0] = fp4_sub(fp4_zero(), fp4_mul(num[0], x[j]));
            deg++;
            denom = fp4_mul(denom, fp4_sub(x[i], x[j]));
        }
        fp4 scale = fp4_mul(fx[i], fp4_inv(denom));
        for (size_t k = 0; k < n; k++) out[k] = fp4_add(out[k], fp4_mul(num[k], scale));
    }
    free(num);
}
void or_poly_divide(fp4* p, size_t n, const uint32_t* z, uint32_t* remainder) {
    fp4 zz; memcpy(&zz, z, 16);
    fp4 cur = fp4_zero();
    for (size_t i = n; i-- > 0;) {
        fp4 next = fp4_add(fp4_mul(zz, cur), p[i]);
        p[i] = cur;
        cur = next;
    }
    memcpy(remainder, &cur, 16);
}
