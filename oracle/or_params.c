/* TEST INFRASTRUCTURE -- CPU oracle (see oracle.h header: parity unpinned).
 * The oracle's parameter set: the counterpart of rk_params / rk_set_params (include/raiko_hip.h).
 * risc0's values are the defaults; the SP1 / Plonky3 preset re-parameterises the same operators
 * (reference provers/sp1/driver/src/lib.rs:48-57 reaches them through sp1-sdk / p3-* crates that
 * are not in the reference tree: every number below is RECALLED). */
#include "oracle.h"
#include "poseidon2_consts.inc"
#include <string.h>

void or_ops_reset_roots(void);
void or_fast_reset_tables(void);

uint32_t g_or_wm = 1073741848u; /* Montgomery form of p - 11 */
or_params g_or = {OR_P - 11u, 137u, 3u, 24u, 0u, 0u, P2_RC_EXT_MONT, P2_RC_INT_MONT, P2_INT_DIAG_MONT, 50u, 2u, 4u, 256u, 0u};

void or_params_preset(or_params* o, int preset) {
    memset(o, 0, sizeof *o);
    if (preset == 1) { /* SP1 core / Plonky3 BabyBear */
        o->ext_w = 11u; o->root_2_27 = 0x1a427a41u; o->coset_shift = 31u;
        o->p2_width = 16; o->p2_m4 = 1; o->p2_pad_free = 1;
        o->queries = 100; o->blowup_log2 = 1; o->fri_fold_log2 = 1; o->fri_min_degree = 1; o->pow_bits = 16;
    } else {
        o->ext_w = OR_P - 11u; o->root_2_27 = 137u; o->coset_shift = 3u;
        o->p2_width = 24; o->p2_m4 = 0; o->p2_pad_free = 0;
        o->queries = 50; o->blowup_log2 = 2; o->fri_fold_log2 = 4; o->fri_min_degree = 256;
    }
}

int or_set_params(const or_params* p) {
    if (p->p2_width != 24 && p->p2_width != 16) return -1;
    if (p->p2_m4 > 1 || p->p2_pad_free > 1) return -1;
    if (p->ext_w == 0 || p->ext_w >= OR_P || p->root_2_27 == 0 || p->root_2_27 >= OR_P) return -1;
    if (p->coset_shift == 0 || p->coset_shift >= OR_P) return -1;
    if (p->queries == 0 || p->queries > OR_MAX_QUERIES || p->fri_fold_log2 < 1 || p->fri_fold_log2 > 4) return -1;
    if (p->blowup_log2 < 1 || p->blowup_log2 > 4 || p->pow_bits > 24) return -1;
    if (p->fri_min_degree == 0 || (p->fri_min_degree & (p->fri_min_degree - 1))) return -1;
    /* W must be a non-residue (x^4 - W irreducible), the root of exact order 2^27 */
    if (fp_pow(fp_from_u32(p->ext_w), (OR_P - 1) / 2) != fp_from_u32(OR_P - 1)) return -1;
    fp r = fp_from_u32(p->root_2_27);
    if (fp_pow(r, (uint64_t)1 << 27) != fp_from_u32(1) || fp_pow(r, (uint64_t)1 << 26) == fp_from_u32(1)) return -1;
    g_or = *p;
    if (!g_or.p2_rc_ext) g_or.p2_rc_ext = p->p2_width == 24 ? P2_RC_EXT_MONT : P2W16_RC_EXT_MONT;
    if (!g_or.p2_rc_int) g_or.p2_rc_int = p->p2_width == 24 ? P2_RC_INT_MONT : P2W16_RC_INT_MONT;
    if (!g_or.p2_diag) g_or.p2_diag = p->p2_width == 24 ? P2_INT_DIAG_MONT : P2W16_INT_DIAG_MONT;
    g_or_wm = fp_from_u32(p->ext_w);
    or_ops_reset_roots();
    or_fast_reset_tables();
    return 0;
}
