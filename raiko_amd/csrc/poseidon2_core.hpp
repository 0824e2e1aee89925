// Poseidon2 over BabyBear, width 24, rate 16, x^7, R_F = 8, R_P = 21 -- the
// permutation of risc0's default "poseidon2" hash suite (risc0-zkp
// core/hash/poseidon2, un-vendored; reached from the reference through
// `session.prove()` at provers/risc0/driver/src/bonsai.rs:271).  Constants come
// from tools/gen_poseidon2_consts.py (Grain LFSR + published diagonal).
//
// Host and device share this code: the device keeps the 24-word state in VGPRs
// (all indices are compile-time after unrolling) and reads round constants
// through wave-uniform scalar loads.
//
// gfx950 cost model (profiles/r01_ubench_isa.txt): every VALU instruction -- 32-bit
// multiplies and 64-bit mads included -- issues at 16 lanes/clk/SIMD, only plain VGPR add/sub
// is twice as fast.  The permutation is therefore written to minimise instruction count:
//   * S-box: signed Montgomery products (3 instructions each, no reduction inside the chain);
//   * external layer: exact 64-bit accumulation (v_mad_u64_u32 with small literal multipliers,
//     64-bit adds) followed by one REDC per cell instead of ~5.5 modular additions per cell.
//     REDC divides by 2^32, so inside a block of four full rounds the state carries a known
//     scale factor 2^(32 e) (e = 0, -7, -56, -399, ...): the round constants are pre-scaled
//     per round, and one constant multiplication in the last round of the block brings the
//     state back to Montgomery form;
//   * internal layer: 64-bit exact sum, one constant product + one conditional subtraction per
//     cell, unsigned-lazy cells in [0, 2p).
#pragma once
#include "bb.hpp"

namespace p2 {

constexpr int CELLS = 24;
constexpr int RATE = 16;
constexpr int OUT = 8;
constexpr int ROUNDS_HALF_FULL = 4;
constexpr int ROUNDS_PARTIAL = 21;

struct Consts {
    // the instance (Montgomery form), set by the caller
    uint32_t rc_ext[2 * ROUNDS_HALF_FULL * CELLS];
    uint32_t rc_int[ROUNDS_PARTIAL];
    uint32_t diag[CELLS];
    // derived by derive()
    uint32_t rc_ext_in[2 * ROUNDS_HALF_FULL * CELLS];  // rc * scale(round) - p: added to the S-box input
    uint32_t rc_int_mp[ROUNDS_PARTIAL];                // rc - p
    uint32_t diag_q[CELLS];                            // diag * (-p^-1) mod 2^32 (bb::umul_const companion)
    uint32_t r2_q;                                     // companion of bb::R2 for bb::umul_const
    uint32_t fix[2], fix_q[2];                         // block-end rescale constants and bb::smul_const companions
};

// exponent e of the scale 2^(32 e) carried by the state at the S-box input of each full round
// (first block follows the initial external layer + REDC; second block starts from Montgomery form)
constexpr int SCALE_EXP[2 * ROUNDS_HALF_FULL] = {0, -7, -56, -399, 1, 0, -7, -56};

inline void derive(Consts& k) {
    const uint32_t Rm = bb::encode(bb::ONE);  // Montgomery form of the field element 2^32
    const uint32_t Rinv_m = bb::inv(Rm);
    auto rpow = [&](long e) {  // Montgomery form of 2^(32 e)
        return e >= 0 ? bb::pow(Rm, (uint64_t)e) : bb::pow(Rinv_m, (uint64_t)(-e));
    };
    for (int r = 0; r < 2 * ROUNDS_HALF_FULL; r++) {
        // stored residue v = rc * 2^32 (Montgomery form).  The S-box input of round r holds the
        // residue a * 2^(32 e) for the true state a, so the residue to add is rc * 2^(32 e):
        // bb::mul(v, f) = v * f / 2^32 with the residue f = 2^(32 e) = rpow(e - 1).
        uint32_t f = rpow((long)SCALE_EXP[r] - 1);
        for (int i = 0; i < CELLS; i++) {
            uint32_t c = bb::mul(k.rc_ext[r * CELLS + i], f);
            k.rc_ext_in[r * CELLS + i] = c - bb::P;
        }
    }
    for (int i = 0; i < ROUNDS_PARTIAL; i++) k.rc_int_mp[i] = k.rc_int[i] - bb::P;
    for (int i = 0; i < CELLS; i++) k.diag_q[i] = k.diag[i] * (0u - bb::MPRIME);
    k.r2_q = bb::R2 * (0u - bb::MPRIME);
    // last round of a block: S-box output carries 2^(32 (7 e - 6)); after x -> x * K / 2^32,
    // the external layer and REDC (another / 2^32) the state must carry 2^32 (Montgomery form):
    // K = 2^(32 (3 - (7 e - 6))) = 2^(32 (9 - 7 e)), as a plain residue.
    for (int b = 0; b < 2; b++) {
        long e = SCALE_EXP[b * ROUNDS_HALF_FULL + ROUNDS_HALF_FULL - 1];
        k.fix[b] = bb::decode(rpow(9 - 7 * e));
        k.fix_q[b] = k.fix[b] * bb::MPRIME;
    }
}

// acc + x * K with a literal multiplier: one v_mad_u64_u32
template <int K>
RK_HD uint64_t madk(uint32_t x, uint64_t acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "n"(K) : "vcc");
    return acc;
#else
    return acc + (uint64_t)x * (uint64_t)K;
#endif
}
template <int K>
RK_HD uint64_t mulk(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t r;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(x), "n"(K) : "vcc");
    return r;
#else
    return (uint64_t)x * (uint64_t)K;
#endif
}

// External layer circ(2*M4, M4, ..., M4), M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]], on
// canonical cells, exact: w[i] < 112 p < 2^38.  Then s[i] = w[i] * 2^-32 (mod p) in [0, p + 53).
RK_HD void m_ext_redc(uint32_t* s) {
    uint64_t w[CELLS];
#pragma unroll
    for (int i = 0; i < CELLS; i += 4) {
        uint32_t a = s[i], b = s[i + 1], c = s[i + 2], d = s[i + 3];
        uint32_t t0 = a + b, t1 = c + d;                     // < 2p < 2^32
        uint64_t u1 = madk<1>(t1, madk<6>(b, mulk<4>(a)));  // 4a + 6b +  c +  d
        uint64_t u0 = madk<2>(d, madk<1>(t0, u1));          // 5a + 7b +  c + 3d
        uint64_t u3 = madk<1>(t0, madk<6>(d, mulk<4>(c)));  //  a +  b + 4c + 6d
        uint64_t u2 = madk<2>(b, madk<1>(t1, u3));          //  a + 3b + 5c + 7d
        w[i] = u0; w[i + 1] = u1; w[i + 2] = u2; w[i + 3] = u3;
    }
    uint64_t t[4];
#pragma unroll
    for (int j = 0; j < 4; j++) t[j] = ((w[j] + w[4 + j]) + (w[8 + j] + w[12 + j])) + (w[16 + j] + w[20 + j]);
#pragma unroll
    for (int i = 0; i < CELLS; i++) s[i] = bb::uredc64(w[i] + t[i & 3]);
}

// (x + c)^7 * 2^(-6*32) as a signed-lazy value; x in [0, p + 53), c_mp = c - p
RK_HD int32_t sbox7_lazy(uint32_t x, uint32_t c_mp) {
    int32_t s = (int32_t)(x + c_mp);
    int32_t s2 = bb::smul(s, s);
    int32_t s3 = bb::smul(s2, s);
    int32_t s6 = bb::smul(s3, s3);
    return bb::smul(s6, s);
}
template <bool LAST>
RK_HD void full_round(uint32_t* s, const Consts& k, int r, int block) {
#pragma unroll
    for (int i = 0; i < CELLS; i++) {
        int32_t y = sbox7_lazy(s[i], k.rc_ext_in[r * CELLS + i]);
        if (LAST) y = bb::smul_const(y, (int32_t)k.fix[block], k.fix_q[block]);
        s[i] = bb::canon(y);
    }
    m_ext_redc(s);
}

// One partial round.  Cells are "unsigned-lazy" representatives in [0, 2p) (cell 0 < p + 53):
//   cell0 <- sbox(cell0 + rc);  S = sum of all cells;  cell_i <- d_i * cell_i + S.
// S is accumulated exactly in 64 bits (one v_mad_u64_u32 per cell, overlapping the power chain
// of cell 0) and brought to [0, p) by REDC and a multiplication by 2^64 mod p; each product
// d_i * cell_i is reduced to [0, p) with one conditional subtraction and S is added without
// reduction (result < 2p fits a u32 because 2p < 2^32).
RK_HD void partial_round(uint32_t* s, const Consts& k, int r) {
    uint64_t acc = 0;
#pragma unroll
    for (int i = 1; i < CELLS; i++) acc = bb::acc_u32(acc, s[i]);
    s[0] = bb::canon(sbox7_lazy(s[0], k.rc_int_mp[r]));
    acc = bb::acc_u32(acc, s[0]);
    uint32_t S = bb::ucanon(bb::umul_const(bb::uredc64(acc), bb::R2, k.r2_q));
    s[0] = bb::add(S, bb::ucanon(bb::umul_const(s[0], k.diag[0], k.diag_q[0])));
#pragma unroll
    for (int i = 1; i < CELLS; i++) s[i] = bb::ucanon(bb::umul_const(s[i], k.diag[i], k.diag_q[i])) + S;
}

RK_HD void permute(uint32_t* s, const Consts& k) {
    m_ext_redc(s);  // canonical Montgomery input -> plain residues (scale 2^0)
#pragma unroll 1
    for (int r = 0; r < ROUNDS_HALF_FULL - 1; r++) full_round<false>(s, k, r, 0);
    full_round<true>(s, k, ROUNDS_HALF_FULL - 1, 0);  // back to Montgomery form, cells in [0, p + 53)
#pragma unroll 1
    for (int r = 0; r < ROUNDS_PARTIAL; r++) partial_round(s, k, r);
    // cells 1..23 are representatives in [0, 2p): back to [0, p) before the S-box input offset
#pragma unroll
    for (int i = 1; i < CELLS; i++) s[i] = bb::ucanon(s[i]);
#pragma unroll 1
    for (int r = ROUNDS_HALF_FULL; r < 2 * ROUNDS_HALF_FULL - 1; r++) full_round<false>(s, k, r, 1);
    full_round<true>(s, k, 2 * ROUNDS_HALF_FULL - 1, 1);
#pragma unroll
    for (int i = 0; i < CELLS; i++) s[i] = bb::ucanon(s[i]);
}

}  // namespace p2
