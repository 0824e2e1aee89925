// Host-only shape checks of the C ABI: TapSet validation and the seal-size bound.  No HIP
// dependency, so tests/asan/ can build exactly this code with -fsanitize=address on the CPU.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/raiko_hip.h"

namespace rk {

constexpr unsigned MAX_PO2_PLUS_2 = 24;  // = ntt::LAMBDA: the 4x domain of the largest segment

// TapSet validation shared by the prover, the seal bound and the verifier: registers sorted by
// (group, offset) covering every column once, combo ids in range, combo_off strictly increasing
// from 0, backs of a combo increasing (as risc0-zkp taps.rs builds them) and <= 64.
inline int check_taps(const rk_taps& t) {
    if (!t.reg_group || !t.reg_offset || !t.reg_combo || !t.combo_off || !t.combo_backs) return RK_ERR_INVALID;
    if ((size_t)t.group_size[0] + t.group_size[1] + t.group_size[2] != t.n_regs) return RK_ERR_INVALID;
    if (t.n_combos == 0 || t.n_combos > (1u << 16)) return RK_ERR_INVALID;
    uint32_t r = 0;
    for (uint32_t g = 0; g < 3; g++)
        for (uint32_t o = 0; o < t.group_size[g]; o++, r++) {
            if (t.reg_group[r] != g || t.reg_offset[r] != o) return RK_ERR_INVALID;
            if (t.reg_combo[r] >= t.n_combos) return RK_ERR_INVALID;
        }
    if (t.combo_off[0] != 0) return RK_ERR_INVALID;
    for (uint32_t c = 0; c < t.n_combos; c++) {
        if (t.combo_off[c + 1] <= t.combo_off[c]) return RK_ERR_INVALID;
        if (t.combo_backs[t.combo_off[c]] > 64) return RK_ERR_INVALID;
        // backs of a combo are distinct (each is divided out once) and increasing, as TapSet builds them
        for (uint32_t b = t.combo_off[c] + 1; b < t.combo_off[c + 1]; b++)
            if (t.combo_backs[b] <= t.combo_backs[b - 1] || t.combo_backs[b] > 64) return RK_ERR_INVALID;
    }
    return RK_OK;
}


// the protocol shape of a segment proof (rk_params): risc0's by default
struct Shape {
    uint32_t queries = 50, blowup_log2 = 2, fold_log2 = 4, min_degree = 256, pow_bits = 0;
};
inline bool shape_ok(const Shape& s) {
    return s.queries >= 1 && s.queries <= RK_MAX_QUERIES && s.blowup_log2 >= 1 && s.blowup_log2 <= 4 && s.fold_log2 >= 1 &&
           s.fold_log2 <= 4 && s.min_degree >= 1 && !(s.min_degree & (s.min_degree - 1)) && s.pow_bits <= 24;
}

// upper bound on the seal words of a segment under a protocol shape; 0 for one rk_prove_segment would reject
inline size_t seal_bound_words(const rk_segment* seg, const Shape& sh = Shape()) {
    if (!seg || !shape_ok(sh) || seg->po2 < 1 || seg->po2 + sh.blowup_log2 > MAX_PO2_PLUS_2) return 0;
    const rk_taps& t = seg->taps;
    if (check_taps(t) != RK_OK) return 0;
    auto lg = [](size_t n) {
        size_t k = 0;
        while (((size_t)1 << k) < n) k++;
        return k;
    };
    const size_t queries = sh.queries, blow = (size_t)1 << sh.blowup_log2, fold = (size_t)1 << sh.fold_log2;
    const size_t check_size = 4 * blow;
    size_t N = (size_t)1 << seg->po2, D = blow * N;
    size_t layers = lg(D);
    size_t words = (size_t)seg->n_globals + 1;
    size_t tot_taps = 0;
    for (uint32_t r = 0; r < t.n_regs; r++) tot_taps += t.combo_off[t.reg_combo[r] + 1] - t.combo_off[t.reg_combo[r]];
    size_t w_all = (size_t)t.group_size[0] + t.group_size[1] + t.group_size[2] + check_size;
    size_t top = 1;                           // Merkle cap: the largest power of two <= queries
    while (top * 2 <= queries) top *= 2;
    words += 4 * top * 8;                     // top layers of the four trace trees
    words += (tot_taps + check_size) * 4;     // coeff_u
    words += queries * (w_all + 4 * layers * 8);   // trace openings
    size_t size = N;
    while (size > sh.min_degree && size >= fold) {   // fri_prove's loop
        size_t domain = size * blow;
        words += top * 8 + queries * (fold * 4 + lg(domain / fold) * 8);
        size /= fold;
    }
    words += size * 4;
    return words + 64;                        // + the proof-of-work nonce and slack
}
inline size_t seal_bound_words(const rk_segment* seg, size_t queries) {
    Shape sh;
    sh.queries = (uint32_t)queries;
    return queries > RK_MAX_QUERIES ? 0 : seal_bound_words(seg, sh);
}

}  // namespace rk
