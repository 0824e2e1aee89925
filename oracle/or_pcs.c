/* TEST INFRASTRUCTURE -- CPU restatement of the data-parallel steps of Plonky3's two-adic FRI PCS (p3-fri
 * two_adic_pcs.rs, p3-dft radix_2_dit*.rs, p3-interpolation lib.rs at the revision SP1 pins: Plonky3@88ea2b8,
 * reference Cargo.lock:4889-5127; reached from provers/sp1/driver/src/lib.rs:48-57).  The crates are outside the
 * reference tree: RECALLED, parity unpinned at byte level.  What the functions compute is fixed by algebra (values
 * of the interpolating polynomial, quotients by x - z), so tests/test_pcs.py pins them against big-integer
 * evaluation; the orderings (bit-reversed rows, low coset first) are the recalled part.
 *
 * Matrices are row-major, height x width, base-field Montgomery words; the coset shift, the 2-adic generator and
 * the blow-up come from g_or (or_set_params). */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static unsigned pcs_log2(size_t n) { unsigned k = 0; while (((size_t)1 << k) < n) k++; return k; }
static size_t pcs_bitrev(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

/* TwoAdicFriPcs::commit for one matrix: `Radix2Dit::coset_lde_batch(evals, log_blowup, shift).bit_reverse_rows()`.
 * in: evaluations over the subgroup of order h (natural order); out: (h << blowup) x w, row r = the values at
 * shift * g_K^bitrev(r) of the polynomials interpolating the columns. */
void or_pcs_coset_lde_rows(fp* out, const fp* in, size_t h, size_t w) {
    const unsigned k = pcs_log2(h), kb = k + g_or.blowup_log2;
    const size_t H = (size_t)1 << kb;
    const fp shift = fp_from_u32(g_or.coset_shift);
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < w; c++) {
        fp* col = (fp*)calloc(H, sizeof(fp));
        fp* tmp = (fp*)malloc(h * sizeof(fp));
        for (size_t i = 0; i < h; i++) tmp[i] = in[i * w + c];
        or_interpolate_ntt(tmp, h);                     /* natural evaluations -> bit-reversed coefficients */
        fp s = fp_from_u32(1);
        for (size_t i = 0; i < h; i++) {                /* coefficient i times shift^i, at its place in a size-H */
            col[pcs_bitrev(i, kb)] = fp_mul(tmp[pcs_bitrev(i, k)], s); /* bit-reversed coefficient vector */
            s = fp_mul(s, shift);
        }
        or_evaluate_ntt(col, H, 0);                     /* -> evaluations at g_K^j, j natural */
        for (size_t j = 0; j < H; j++) out[pcs_bitrev(j, kb) * w + c] = col[j];
        free(tmp);
        free(col);
    }
}

/* The opened values of one matrix at z (two_adic_pcs.rs open: interpolate_coset on the low coset of the LDE):
 *   p(z) = (z^h - s^h) / (h s^(h-1)) * sum_i g^i y_i / (z - s g^i),   y_i = row bitrev_k(i) of the LDE.
 * out: w extension elements. */
void or_pcs_eval_at(fp4* out, const fp* lde, size_t H, size_t w, const uint32_t* z) {
    const unsigned kb = pcs_log2(H), k = kb - g_or.blowup_log2;
    const size_t h = (size_t)1 << k;
    const fp shift = fp_from_u32(g_or.coset_shift), g = or_rou_fwd(k);
    fp4 zz; memcpy(&zz, z, 16);
    fp4* col_scale = (fp4*)malloc(h * sizeof(fp4));
    fp gi = fp_from_u32(1);
    for (size_t i = 0; i < h; i++) {
        fp4 diff = fp4_sub(zz, fp4_from_fp(fp_mul(shift, gi)));
        col_scale[i] = fp4_scale(fp4_inv(diff), gi);
        gi = fp_mul(gi, g);
    }
    fp4 zerofier = fp4_sub(fp4_pow(zz, h), fp4_from_fp(fp_pow(shift, h)));
    fp denom = fp_mul(fp_from_u32((uint32_t)(h % OR_P)), fp_pow(shift, h - 1));
    fp4 scaling = fp4_scale(zerofier, fp_inv(denom));
    for (size_t c = 0; c < w; c++) {
        fp4 sum = fp4_zero();
        for (size_t i = 0; i < h; i++) sum = fp4_add(sum, fp4_scale(col_scale[i], lde[pcs_bitrev(i, k) * w + c]));
        out[c] = fp4_mul(sum, scaling);
    }
    free(col_scale);
}

/* The "reduce rows" step of two_adic_pcs.rs open for one matrix and its opening points:
 *   ro[r] += alpha^(offset + j w) * (sum_c alpha^c M[r][c] - sum_c alpha^c ys_j[c]) / (x_r - z_j)   for every point j,
 * x_r = shift * g_K^bitrev(r) (the LDE's own row order).  ro: H extension elements, in/out. */
void or_pcs_reduce_openings(fp4* ro, const fp* lde, size_t H, size_t w, size_t n_points, const uint32_t* points,
                            const uint32_t* ys, const uint32_t* alpha, uint64_t alpha_offset) {
    const unsigned kb = pcs_log2(H);
    const fp shift = fp_from_u32(g_or.coset_shift), gK = or_rou_fwd(kb);
    fp4 a; memcpy(&a, alpha, 16);
    fp4* apow = (fp4*)malloc((w ? w : 1) * sizeof(fp4));
    fp4 cur = fp4_one();
    for (size_t c = 0; c < w; c++) { apow[c] = cur; cur = fp4_mul(cur, a); }
    for (size_t j = 0; j < n_points; j++) {
        fp4 zj; memcpy(&zj, points + 4 * j, 16);
        const fp4* y = (const fp4*)(ys + 4 * j * w);
        fp4 rys = fp4_zero();
        for (size_t c = 0; c < w; c++) rys = fp4_add(rys, fp4_mul(apow[c], y[c]));
        fp4 off = fp4_pow(a, alpha_offset + (uint64_t)j * w);
#pragma omp parallel for schedule(static)
        for (size_t r = 0; r < H; r++) {
            fp4 rr = fp4_zero();
            for (size_t c = 0; c < w; c++) rr = fp4_add(rr, fp4_scale(apow[c], lde[r * w + c]));
            fp x = fp_mul(shift, fp_pow(gK, pcs_bitrev(r, kb)));
            fp4 inv_denom = fp4_inv(fp4_sub(fp4_from_fp(x), zj));
            ro[r] = fp4_add(ro[r], fp4_mul(off, fp4_mul(fp4_sub(rr, rys), inv_denom)));
        }
    }
    free(apow);
}

/* DuplexChallenger::grind (p3-challenger, RECALLED): the literal definition -- for w = 0, 1, ... clone the challenger,
 * observe(w) (clear the outputs, push the input, duplex when the rate is full), sample_bits(bits) (duplex first if
 * inputs are buffered or no outputs are left, pop the LAST output, mask its canonical value) -- smallest w giving 0.
 * state: OR_CELLS sponge cells; input: the buffered observations (n_input < rate). */
uint32_t or_duplex_grind(const fp* state, const fp* input, size_t n_input, unsigned bits) {
    const size_t width = OR_CELLS, rate = OR_CELLS_RATE;
    for (uint32_t w = 0; w < OR_P; w++) {
        fp s[OR_MAX_CELLS], in[OR_MAX_CELLS], out[OR_MAX_CELLS];
        size_t n_in = n_input, n_out = 0;
        memcpy(s, state, width * sizeof(fp));
        memcpy(in, input, n_input * sizeof(fp));
        /* observe */
        n_out = 0;
        in[n_in++] = fp_from_u32(w);
        if (n_in == rate) {
            for (size_t i = 0; i < n_in; i++) s[i] = in[i];
            n_in = 0;
            or_poseidon2_mix(s);
            for (size_t i = 0; i < rate; i++) out[i] = s[i];
            n_out = rate;
        }
        /* sample_bits */
        if (n_in != 0 || n_out == 0) {
            for (size_t i = 0; i < n_in; i++) s[i] = in[i];
            n_in = 0;
            or_poseidon2_mix(s);
            for (size_t i = 0; i < rate; i++) out[i] = s[i];
            n_out = rate;
        }
        uint32_t v = or_fp_decode(out[--n_out]);
        if ((v & (uint32_t)(((uint64_t)1 << bits) - 1)) == 0) return w;
    }
    return 0xffffffffu;
}
