// Lane-level bodies of the uni-stark path's own kernels (p3.hip, p3_air.hip) as host/device functions: the GPU kernels are
// these phases with __syncthreads() between them; tests/emul runs the same code on the CPU one emulated lane at a time.
#pragma once
#include "bb.hpp"

namespace p3k {

// ---- lookups: the permutation trace (sp1-core generate_permutation_trace, RECALLED).  desc = the challenge vector
// [alpha | beta^0 | beta^1 ..] (4 words each, n_chal words), the flat interactions (kind, bus, mult_is_const, mult, n_values,
// slots...; constants as Montgomery words; columns renumbered to slots of the `used` list), then the n_used distinct
// main-trace columns the interactions read.  A workgroup takes PERM_ROWS rows of the row-major trace: first every wave
// stages the used columns of its rows in LDS -- one row per load instruction, the lanes along the used columns, so a row's
// cache lines are fetched once instead of once per interaction --, then one lane per row walks the interactions out of LDS
// (slot-major: conflict-free).  out = 4 (nb + 1) columns of n words: the nb batch entries, then the row totals (the
// prefix sums turn those into the running sum in place).
constexpr int PERM_ROWS = 256, PERM_LD = PERM_ROWS + 1;   // odd slot stride: the staging writes (lanes along slots) and the reads (lanes along rows) both spread over the banks
struct PermArgs {
    uint32_t* out;
    const uint32_t* trace;
    const uint32_t* desc;
    size_t n, w;
    uint32_t n_chal, n_lookups, wm, n_used, desc_words;
};
// phase 1: lane `tid` of workgroup `blk` stages its share of the tile (n_used x PERM_LD words).  A wave takes 64 rows, eight
// at a time: the loads of eight rows are issued before the first LDS store waits for any of them (one row per iteration
// would serialise 64 memory latencies per wave)
RK_HD void perm_stage(const PermArgs& a, size_t blk, unsigned tid, uint32_t* tile) {
    const size_t r0 = blk * PERM_ROWS;
    const uint32_t* used = a.desc + a.desc_words;
    const unsigned wave = tid >> 6, lane = tid & 63;
    constexpr int GROUP = 8, PER_LANE = 2;   // 2 x 64 lanes cover the 120 columns an AIR's interactions may read
    uint32_t col[PER_LANE];
    for (int q = 0; q < PER_LANE; q++) col[q] = lane + 64 * q < a.n_used ? used[lane + 64 * q] : 0;
    for (unsigned i = 0; i < 64; i += GROUP) {
        uint32_t v[GROUP][PER_LANE];
#pragma unroll
        for (int j = 0; j < GROUP; j++) {
            const size_t r = r0 + wave * 64 + i + j;
            const uint32_t* row = a.trace + (r < a.n ? r : a.n - 1) * a.w;    // past the end: a valid row, never stored
#pragma unroll
            for (int q = 0; q < PER_LANE; q++) v[j][q] = lane + 64 * q < a.n_used ? row[col[q]] : 0;
        }
#pragma unroll
        for (int j = 0; j < GROUP; j++) {
            const unsigned lr = wave * 64 + i + j;
            if (r0 + lr >= a.n) continue;
#pragma unroll
            for (int q = 0; q < PER_LANE; q++)
                if (lane + 64 * q < a.n_used) tile[(lane + 64 * q) * PERM_LD + lr] = v[j][q];
        }
    }
}
// phase 2: lane `tid` walks the interactions of its row.  (Inverting the denominators of eight interactions together --
// Montgomery's trick -- was measured and dropped: 0.50 ms against 0.45 at 2^20 rows x 16 interactions; the three arrays
// it keeps per lane cost more than the fourteen base-field powers it saves.)
RK_HD void perm_row(const PermArgs& a, size_t blk, unsigned tid, const uint32_t* tile) {
    const size_t r = blk * PERM_ROWS + tid;
    if (r >= a.n) return;
    const uint32_t* row = tile + tid;
    const uint32_t* ch = a.desc;
    const uint32_t* d = a.desc + a.n_chal;
    const bb::Ext alpha{{ch[0], ch[1], ch[2], ch[3]}};
    bb::Ext total = bb::ext_zero(), entry = bb::ext_zero();
    const uint32_t nb = (a.n_lookups + 1) / 2;
    for (uint32_t i = 0; i < a.n_lookups; i++) {
        const uint32_t kind = d[0], bus = d[1], is_const = d[2], mult = d[3], nv = d[4];
        bb::Ext rlc = bb::add(alpha, bb::scale(bb::Ext{{ch[4], ch[5], ch[6], ch[7]}}, bus));
        for (uint32_t j = 0; j < nv; j++) {
            const uint32_t* b = ch + 8 + 4 * j;
            rlc = bb::add(rlc, bb::scale(bb::Ext{{b[0], b[1], b[2], b[3]}}, row[d[5 + j] * PERM_LD]));
        }
        const uint32_t m = is_const ? mult : row[mult * PERM_LD];
        const bb::Ext term = bb::scale(bb::inv(rlc, a.wm), kind == 0 ? m : bb::neg(m));
        entry = bb::add(entry, term);
        d += 5 + nv;
        if ((i & 1u) || i + 1 == a.n_lookups) {
            const uint32_t b = i >> 1;
            for (int k = 0; k < 4; k++) a.out[(size_t)(4 * b + k) * a.n + r] = entry.c[k];
            total = bb::add(total, entry);
            entry = bb::ext_zero();
        }
    }
    for (int k = 0; k < 4; k++) a.out[(size_t)(4 * nb + k) * a.n + r] = total.c[k];
}

// ---- the Poseidon2 chip's rows (rk_p2_chip_trace): one lane per permutation; tab = rc_ext | rc_int | diag (Montgomery words)
struct P2ChipLayout {
    uint32_t W, RP, width;
    RK_HD uint32_t in() const { return 0; }
    RK_HD uint32_t x3(uint32_t r) const { return r < 4 ? W + 2 * W * r : W + 8 * W + 2 * RP - 1 + W + 2 * W * (r - 4); }
    RK_HD uint32_t post(uint32_t r) const { return x3(r) + W; }
    RK_HD uint32_t x3i(uint32_t k) const { return W + 8 * W + k; }
    RK_HD uint32_t s0(uint32_t k) const { return W + 8 * W + RP + (k - 1); }   // k >= 1
    RK_HD uint32_t int_out() const { return W + 8 * W + 2 * RP - 1; }
    RK_HD uint32_t mult() const { return width - 1; }
    RK_HD uint32_t out() const { return post(7); }
};
template <int W, int M4>
RK_HD void chip_m_ext(uint32_t (&c)[W]) {
    uint32_t sums[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < W; i += 4) {
        const uint32_t a = c[i], b = c[i + 1], d = c[i + 2], e = c[i + 3];
        if (M4 == 0) {
            const uint32_t t0 = bb::add(a, b), t1 = bb::add(d, e), t2 = bb::add(bb::dbl(b), t1), t3 = bb::add(bb::dbl(e), t0);
            const uint32_t t4 = bb::add(bb::dbl(bb::dbl(t1)), t3), t5 = bb::add(bb::dbl(bb::dbl(t0)), t2);
            c[i] = bb::add(t3, t5), c[i + 1] = t5, c[i + 2] = bb::add(t2, t4), c[i + 3] = t4;
        } else {
            const uint32_t s = bb::add(bb::add(a, b), bb::add(d, e));
            c[i] = bb::add(bb::add(s, a), bb::dbl(b));
            c[i + 1] = bb::add(bb::add(s, b), bb::dbl(d));
            c[i + 2] = bb::add(bb::add(s, d), bb::dbl(e));
            c[i + 3] = bb::add(bb::add(s, e), bb::dbl(a));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) sums[j] = bb::add(sums[j], c[i + j]);
    }
#pragma unroll
    for (int i = 0; i < W; i++) c[i] = bb::add(c[i], sums[i & 3]);
}
template <int W, int RP, int M4>
RK_HD void chip_row(uint32_t* row, const uint32_t* in, uint32_t mult, const uint32_t* tab, const P2ChipLayout& L) {
    const uint32_t *rc_ext = tab, *rc_int = tab + 8 * W, *diag = rc_int + RP;
    uint32_t c[W];
#pragma unroll
    for (int i = 0; i < W; i++) row[i] = c[i] = in[i];
    chip_m_ext<W, M4>(c);
    for (int rd = 0; rd < 8; rd++) {
        if (rd == 4) {
            for (int k = 0; k < RP; k++) {
                if (k > 0) row[L.s0(k)] = c[0];
                const uint32_t t = bb::add(c[0], rc_int[k]), x3 = bb::mul(bb::sqr(t), t);
                row[L.x3i(k)] = x3;
                c[0] = bb::mul(bb::sqr(x3), t);
                uint32_t sum = 0;
#pragma unroll
                for (int i = 0; i < W; i++) sum = bb::add(sum, c[i]);
#pragma unroll
                for (int i = 0; i < W; i++) c[i] = bb::add(sum, bb::mul(c[i], diag[i]));
            }
#pragma unroll
            for (int i = 0; i < W; i++) row[L.int_out() + i] = c[i];
        }
#pragma unroll
        for (int i = 0; i < W; i++) {
            const uint32_t s = bb::add(c[i], rc_ext[rd * W + i]), x3 = bb::mul(bb::sqr(s), s);
            row[L.x3(rd) + i] = x3;
            c[i] = bb::mul(bb::sqr(x3), s);
        }
        chip_m_ext<W, M4>(c);
#pragma unroll
        for (int i = 0; i < W; i++) row[L.post(rd) + i] = c[i];
    }
    row[L.mult()] = mult;
}

}  // namespace p3k
