// Poseidon2 over BabyBear, width 24, rate 16, x^7, R_F = 8, R_P = 21 -- the
// permutation of risc0's default "poseidon2" hash suite (risc0-zkp
// core/hash/poseidon2, un-vendored; reached from the reference through
// `session.prove()` at provers/risc0/driver/src/bonsai.rs:271).  Constants come
// from tools/gen_poseidon2_consts.py (Grain LFSR + published diagonal).
//
// Host and device share this code: the device keeps the 24-word state in VGPRs
// (all indices are compile-time after unrolling) and reads round constants
// through wave-uniform scalar loads from __constant__ memory.
#pragma once
#include "bb.hpp"

namespace p2 {

constexpr int CELLS = 24;
constexpr int RATE = 16;
constexpr int OUT = 8;
constexpr int ROUNDS_HALF_FULL = 4;
constexpr int ROUNDS_PARTIAL = 21;

struct Consts {
    uint32_t rc_ext[2 * ROUNDS_HALF_FULL * CELLS];  // Montgomery form
    uint32_t rc_int[ROUNDS_PARTIAL];
    uint32_t diag[CELLS];
    // derived by derive(): rc - p (so `x + rc` needs no reduction before the S-box) and
    // diag * p^-1 mod 2^32 (companion of the constant multiplier, bb::smul_const)
    uint32_t rc_ext_mp[2 * ROUNDS_HALF_FULL * CELLS];
    uint32_t rc_int_mp[ROUNDS_PARTIAL];
    uint32_t diag_q[CELLS];   // diag * (-p^-1) mod 2^32 (bb::umul_const companion)
    uint32_t r2_q;            // (2^64 mod p) * (-p^-1) mod 2^32: companion of bb::R2
};
inline void derive(Consts& k) {
    for (int i = 0; i < 2 * ROUNDS_HALF_FULL * CELLS; i++) k.rc_ext_mp[i] = k.rc_ext[i] - bb::P;
    for (int i = 0; i < ROUNDS_PARTIAL; i++) k.rc_int_mp[i] = k.rc_int[i] - bb::P;
    for (int i = 0; i < CELLS; i++) k.diag_q[i] = k.diag[i] * (0u - bb::MPRIME);
    k.r2_q = bb::R2 * (0u - bb::MPRIME);
}

// circ(2*M4, M4, ..., M4) with M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]]
RK_HD void m_ext(uint32_t* s) {
#pragma unroll
    for (int i = 0; i < CELLS; i += 4) {
        uint32_t a = s[i], b = s[i + 1], c = s[i + 2], d = s[i + 3];
        uint32_t t0 = bb::add(a, b), t1 = bb::add(c, d);
        uint32_t t2 = bb::add(bb::dbl(b), t1), t3 = bb::add(bb::dbl(d), t0);
        uint32_t t4 = bb::add(bb::dbl(bb::dbl(t1)), t3), t5 = bb::add(bb::dbl(bb::dbl(t0)), t2);
        uint32_t t6 = bb::add(t3, t5), t7 = bb::add(t2, t4);
        s[i] = t6; s[i + 1] = t5; s[i + 2] = t7; s[i + 3] = t4;
    }
    // column sums over the six 4-cell chunks, as trees
    uint32_t t[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
        t[j] = bb::add(bb::add(bb::add(s[j], s[4 + j]), bb::add(s[8 + j], s[12 + j])), bb::add(s[16 + j], s[20 + j]));
#pragma unroll
    for (int i = 0; i < CELLS; i++) s[i] = bb::add(s[i], t[i & 3]);
}
// One partial round.  Cell 0 is canonical, cells 1..23 are "unsigned-lazy" representatives in
// [0, 2p):   cell0 <- sbox(cell0 + rc);  S = sum of all cells;  cell_i <- d_i * cell_i + S.
// S is accumulated exactly in 64 bits (one v_mad_u64_u32 per cell, overlapping the power chain
// of cell 0) and brought to [0, p) by REDC and a multiplication by 2^64 mod p; each product
// d_i * cell_i is reduced to [0, p) with one conditional subtraction and S is added without
// reduction (result < 2p fits a u32 because 2p < 2^32).
RK_HD void partial_round(uint32_t* s, const Consts& k, int r) {
    uint64_t acc = 0;
#pragma unroll
    for (int i = 1; i < CELLS; i++) acc = bb::acc_u32(acc, s[i]);
    s[0] = bb::sbox7_add(s[0], k.rc_int_mp[r]);
    acc = bb::acc_u32(acc, s[0]);
    uint32_t S = bb::ucanon(bb::umul_const(bb::uredc64(acc), bb::R2, k.r2_q));
    s[0] = bb::add(S, bb::ucanon(bb::umul_const(s[0], k.diag[0], k.diag_q[0])));
#pragma unroll
    for (int i = 1; i < CELLS; i++) s[i] = bb::ucanon(bb::umul_const(s[i], k.diag[i], k.diag_q[i])) + S;
}
RK_HD void full_round(uint32_t* s, const Consts& k, int r) {
#pragma unroll
    for (int i = 0; i < CELLS; i++) s[i] = bb::sbox7_add(s[i], k.rc_ext_mp[r * CELLS + i]);
    m_ext(s);
}
RK_HD void permute(uint32_t* s, const Consts& k) {
    m_ext(s);
#pragma unroll 1
    for (int r = 0; r < ROUNDS_HALF_FULL; r++) full_round(s, k, r);
#pragma unroll 1
    for (int r = 0; r < ROUNDS_PARTIAL; r++) partial_round(s, k, r);
    // cells 1..23 are representatives in [0, 2p): back to [0, p) for the external layers
#pragma unroll
    for (int i = 1; i < CELLS; i++) s[i] = bb::ucanon(s[i]);
#pragma unroll 1
    for (int r = ROUNDS_HALF_FULL; r < 2 * ROUNDS_HALF_FULL; r++) full_round(s, k, r);
}

}  // namespace p2
