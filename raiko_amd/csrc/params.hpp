// rk_params -> the resolved parameter set host code works with: validation, derived Montgomery
// forms, the Poseidon2 instance.  Shared by the context (rk_set_params) and the host-side verifier
// (rk_verify_opts.params).  Presets carry the RECALLED values of risc0 1.0.1 and of SP1 / Plonky3
// (reference call sites provers/risc0/driver/src/bonsai.rs:271 and
// provers/sp1/driver/src/lib.rs:48-57 -- both crates families are outside the reference tree).
#pragma once
#include "../../include/raiko_hip.h"
#include "taps.hpp"
#include "poseidon2_any.hpp"
#include "poseidon2_consts.inc"

namespace rk {

struct Sys {
    uint32_t wm = bb::WM_RISC0;  // Montgomery form of W in Fp[x]/(x^4 - W)
    uint32_t root27m = 0;        // Montgomery form of the 2^27-subgroup generator
    uint32_t shiftm = 0;         // Montgomery form of the coset shift
    uint32_t ext_w = bb::P - 11, root_2_27 = 137, coset_shift = 3;
    uint32_t queries = 50, blowup_log2 = 2, fri_fold_log2 = 4, fri_min_degree = 256, pow_bits = 0;
    Sys() : root27m(bb::encode(137)), shiftm(bb::encode(3)) {}
    Shape shape() const { return Shape{queries, blowup_log2, fri_fold_log2, fri_min_degree, pow_bits}; }
    bool is_default_field() const { return ext_w == bb::P - 11 && root_2_27 == 137 && coset_shift == 3; }
};

inline void params_preset(rk_params* o, int preset) {
    *o = rk_params{};
    o->struct_size = (uint32_t)sizeof(rk_params);
    if (preset == RK_PRESET_SP1) {
        o->ext_w = 11u;
        o->root_2_27 = 0x1a427a41u;
        o->coset_shift = 31u;
        o->p2_width = 16;
        o->p2_m4 = 1;
        o->p2_pad_free = 1;
        o->queries = 100;
        o->blowup_log2 = 1;
        o->fri_fold_log2 = 1;
        o->fri_min_degree = 1;
        o->pow_bits = 16;
    } else {
        o->ext_w = bb::P - 11u;
        o->root_2_27 = 137u;
        o->coset_shift = 3u;
        o->p2_width = 24;
        o->p2_m4 = 0;
        o->p2_pad_free = 0;
        o->queries = 50;
        o->blowup_log2 = 2;
        o->fri_fold_log2 = 4;
        o->fri_min_degree = 256;
        o->pow_bits = 0;
    }
}

// RK_OK and *sys / *p2any filled, or RK_ERR_INVALID with both untouched
inline int resolve_params(const rk_params* in, Sys* sys, p2::Any* p2any) {
    if (!in || in->struct_size != sizeof(rk_params)) return RK_ERR_INVALID;
    if (in->p2_width != 24 && in->p2_width != 16) return RK_ERR_INVALID;
    if (in->p2_m4 > 1 || in->p2_pad_free > 1) return RK_ERR_INVALID;
    if (in->ext_w == 0 || in->ext_w >= bb::P || in->root_2_27 == 0 || in->root_2_27 >= bb::P) return RK_ERR_INVALID;
    if (in->coset_shift == 0 || in->coset_shift >= bb::P) return RK_ERR_INVALID;
    if (in->queries == 0 || in->queries > RK_MAX_QUERIES) return RK_ERR_INVALID;
    if (in->fri_fold_log2 < 1 || in->fri_fold_log2 > 4 || in->blowup_log2 < 1 || in->blowup_log2 > 4) return RK_ERR_INVALID;
    if (in->fri_min_degree == 0 || (in->fri_min_degree & (in->fri_min_degree - 1))) return RK_ERR_INVALID;
    if (in->pow_bits > 24) return RK_ERR_INVALID;
    // x^4 - W irreducible over Fp (p = 1 mod 4)  <=>  W is not a square
    if (bb::pow(bb::encode(in->ext_w), (bb::P - 1) / 2) != bb::encode(bb::P - 1)) return RK_ERR_INVALID;
    const uint32_t r = bb::encode(in->root_2_27);
    if (bb::pow(r, (uint64_t)1 << 27) != bb::ONE || bb::pow(r, (uint64_t)1 << 26) == bb::ONE) return RK_ERR_INVALID;
    const bool w24 = in->p2_width == 24;
    const uint32_t* rc_ext = in->p2_rc_ext ? in->p2_rc_ext : (w24 ? P2_RC_EXT_MONT : P2W16_RC_EXT_MONT);
    const uint32_t* rc_int = in->p2_rc_int ? in->p2_rc_int : (w24 ? P2_RC_INT_MONT : P2W16_RC_INT_MONT);
    const uint32_t* diag = in->p2_diag ? in->p2_diag : (w24 ? P2_INT_DIAG_MONT : P2W16_INT_DIAG_MONT);
    const unsigned width = in->p2_width, rp = w24 ? 21u : 13u;
    for (unsigned i = 0; i < 8 * width; i++)
        if (rc_ext[i] >= bb::P) return RK_ERR_INVALID;
    for (unsigned i = 0; i < rp; i++)
        if (rc_int[i] >= bb::P) return RK_ERR_INVALID;
    for (unsigned i = 0; i < width; i++)
        if (diag[i] >= bb::P) return RK_ERR_INVALID;
    p2any->set((int)width, (int)in->p2_m4, in->p2_pad_free != 0, rc_ext, rc_int, diag);
    sys->ext_w = in->ext_w;
    sys->root_2_27 = in->root_2_27;
    sys->coset_shift = in->coset_shift;
    sys->wm = bb::encode(in->ext_w);
    sys->root27m = r;
    sys->shiftm = bb::encode(in->coset_shift);
    sys->queries = in->queries;
    sys->blowup_log2 = in->blowup_log2;
    sys->fri_fold_log2 = in->fri_fold_log2;
    sys->fri_min_degree = in->fri_min_degree;
    sys->pow_bits = in->pow_bits;
    return RK_OK;
}

}  // namespace rk
