// BabyBear field core shared by device kernels and the host-side prover.
//
// p = 15*2^27 + 1, elements are Montgomery residues (R = 2^32) in [0, p), the
// storage format of every buffer crossing the C ABI.  Extension: Fp[x]/(x^4+11).
// This restates the published field used by `session.prove()` (reference call
// site: provers/risc0/driver/src/bonsai.rs:271; the implementation is in the
// un-vendored crate risc0-core 1.0.1, reference Cargo.lock:7171).
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RK_HD __host__ __device__ __forceinline__
#define RK_D __device__ __forceinline__
#else
#define RK_HD inline
#define RK_D inline
#endif

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E).  For code whose indices
// must be constants after inlining (register arrays, scalar-register chunks): `#pragma unroll`
// gives up on loops that contain volatile asm or whose bounds depend on an outer unrolled loop.
template <int B, int E, class F>
RK_HD void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

namespace bb {

constexpr uint32_t P = 2013265921u;
constexpr uint32_t MPRIME = 0x88000001u;  // p^-1 mod 2^32
constexpr uint32_t R2 = 1172168163u;      // 2^64 mod p
constexpr uint32_t ONE = 268435454u;      // 2^32 mod p
constexpr uint32_t INVALID = 0xffffffffu;

RK_HD uint32_t add(uint32_t a, uint32_t b) {
    uint32_t r = a + b;  // < 2p < 2^32
    uint32_t s = r - P;
    return s < r ? s : r;  // min(r, r-p) in unsigned arithmetic
}
RK_HD uint32_t sub(uint32_t a, uint32_t b) {
    uint32_t r = a - b;
    uint32_t s = r + P;
    return s < r ? s : r;  // if a<b, r wrapped and r+p is the small value
}
RK_HD uint32_t neg(uint32_t a) { return a ? P - a : 0u; }
RK_HD uint32_t dbl(uint32_t a) { return add(a, a); }

// Montgomery reduction of a 64-bit value t < 2^32 * p: returns t * 2^-32 mod p in [0, p)
RK_HD uint32_t mont_reduce(uint64_t t) {
    uint32_t q = (uint32_t)t * (0u - MPRIME);  // q = -t * p^-1 mod 2^32
    uint64_t u = t + (uint64_t)q * P;          // low 32 bits cancel; u < 2^64
    uint32_t r = (uint32_t)(u >> 32);          // < 2p
    uint32_t s = r - P;
    return s < r ? s : r;
}
RK_HD uint32_t mul(uint32_t a, uint32_t b) { return mont_reduce((uint64_t)a * b); }
RK_HD uint32_t sqr(uint32_t a) { return mul(a, a); }
RK_HD uint32_t encode(uint32_t canon) { return mul(canon % P, R2); }
RK_HD uint32_t decode(uint32_t m) { return mont_reduce((uint64_t)m); }
RK_HD uint32_t pow(uint32_t a, uint64_t e) {
    uint32_t r = ONE;
    while (e) {
        if (e & 1) r = mul(r, a);
        a = mul(a, a);
        e >>= 1;
    }
    return r;
}
// a^(p - 2), p - 2 = 0b111 0 1^27: a^7, one squaring, then nine windows "cube of squarings, times a^7" -- 33 squarings and
// 12 products in a straight line instead of the 31 + 30 of square-and-multiply with its 64-bit exponent loop
RK_HD uint32_t inv(uint32_t a) {
    const uint32_t a2 = sqr(a), a3 = mul(a2, a), a7 = mul(sqr(a3), a);
    uint32_t r = sqr(a7);
#pragma unroll
    for (int w = 0; w < 9; w++) r = mul(sqr(sqr(sqr(r))), a7);
    return r;
}
RK_HD uint32_t sbox7(uint32_t x) {
    uint32_t x2 = mul(x, x);
    uint32_t x3 = mul(x2, x);
    uint32_t x6 = mul(x3, x3);
    return mul(x6, x);
}

// Signed Montgomery product.  For ANY int32 a, b it returns r == a*b*2^-32 (mod p) with
// |r| <= |a*b| / 2^32 + p/2 < 2^31, so chains of products need no reduction in between:
//   t = a*b (signed 64-bit), q = lo32(t) * p^-1 (as int32), u = t - q*p (divisible by 2^32),
//   r = hi32(u).
// gfx950: v_mad_i64_i32 + v_mul_lo_u32 + v_mad_i64_i32.  Every VALU instruction except plain
// VGPR/literal v_add/v_sub issues at 16 lanes/clk/SIMD on MI355X -- a 64-bit mad costs the
// same as a v_min_u32 (profiles/r01_ubench_isa.txt) -- so instruction count is what matters.
constexpr int32_t NEG_P = (int32_t)(0u - P);
RK_HD int32_t redc64(int64_t t) {
    int32_t q = (int32_t)((uint32_t)t * MPRIME);
    int64_t u = t + (int64_t)q * (int64_t)NEG_P;
    return (int32_t)(u >> 32);
}
RK_HD int32_t smul(int32_t a, int32_t b) { return redc64((int64_t)a * (int64_t)b); }
// product with a constant c whose companion c_q = c * p^-1 mod 2^32 is precomputed: q does not
// wait for the low half of the product
RK_HD int32_t smul_const(int32_t a, int32_t c, uint32_t c_q) {
    int64_t t = (int64_t)a * (int64_t)c;
    int32_t q = (int32_t)((uint32_t)a * c_q);
    int64_t u = t + (int64_t)q * (int64_t)NEG_P;
    return (int32_t)(u >> 32);
}
// Unsigned Montgomery product with a constant c < p whose companion c_nq = c * (-p^-1) mod 2^32
// is precomputed.  For any u with u*c < 2^32 * p + ... (here u < 2p): returns
// r == u*c*2^-32 (mod p) with r < u*c / 2^32 + p  (< 1.94 p for u < 2p), one subtraction
// from canonical.  v_mad_u64_u32 + v_mul_lo_u32 + v_mad_u64_u32.
RK_HD uint32_t umul_const(uint32_t u, uint32_t c, uint32_t c_nq) {
    uint64_t t = (uint64_t)u * c;
    uint32_t q = u * c_nq;
    uint64_t w = t + (uint64_t)q * P;  // low word cancels, no overflow: t, q*p < 2^63
    return (uint32_t)(w >> 32);
}
// t * 2^-32 (mod p) for an unsigned 64-bit t < 2^63: result < t / 2^32 + p
RK_HD uint32_t uredc64(uint64_t t) {
    uint32_t q = (uint32_t)t * (0u - MPRIME);
    uint64_t w = t + (uint64_t)q * P;
    return (uint32_t)(w >> 32);
}
// acc + x as one 64-bit v_mad_u64_u32 (x * 1 + acc): hipcc otherwise zero-extends x with a
// v_mov and adds with v_lshl_add_u64, two 4-cycle instructions per term
RK_HD uint64_t acc_u32(uint64_t acc, uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(acc) : "v"(x) : "vcc");
    return acc;
#else
    return acc + x;
#endif
}
// [0, 2p) -> [0, p)
RK_HD uint32_t ucanon(uint32_t r) {
    uint32_t s = r - P;
    return s < r ? s : r;
}
// signed-lazy value in (-p, p) -> canonical [0, p)
RK_HD uint32_t canon(int32_t r) {
    uint32_t u = (uint32_t)r, v = u + P;
    return v < u ? v : u;  // negative values wrap to small after +p
}
// (x + rc)^7 for canonical x; rc_minus_p = rc - p (two's complement), so x + rc_minus_p is a
// representative of x + rc in [-p, p): the addition needs no reduction and the whole power
// chain runs on signed-lazy values; one conditional +p at the end restores [0, p).
RK_HD uint32_t sbox7_add(uint32_t x, uint32_t rc_minus_p) {
    int32_t s = (int32_t)(x + rc_minus_p);
    int32_t s2 = smul(s, s);
    int32_t s3 = smul(s2, s);
    int32_t s6 = smul(s3, s3);
    return canon(smul(s6, s));
}

struct Ext {
    uint32_t c[4];
};
RK_HD Ext ext_zero() { return Ext{{0, 0, 0, 0}}; }
RK_HD Ext ext_one() { return Ext{{ONE, 0, 0, 0}}; }
RK_HD Ext ext_from(uint32_t a) { return Ext{{a, 0, 0, 0}}; }
RK_HD Ext add(const Ext& a, const Ext& b) {
    return Ext{{add(a.c[0], b.c[0]), add(a.c[1], b.c[1]), add(a.c[2], b.c[2]), add(a.c[3], b.c[3])}};
}
RK_HD Ext sub(const Ext& a, const Ext& b) {
    return Ext{{sub(a.c[0], b.c[0]), sub(a.c[1], b.c[1]), sub(a.c[2], b.c[2]), sub(a.c[3], b.c[3])}};
}
RK_HD Ext scale(const Ext& a, uint32_t s) {
    return Ext{{mul(a.c[0], s), mul(a.c[1], s), mul(a.c[2], s), mul(a.c[3], s)}};
}
RK_HD bool eq(const Ext& a, const Ext& b) {
    return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.c[3] == b.c[3];
}
// The extension is Fp[x]/(x^4 - W).  W travels as its Montgomery form `wm`: risc0's field has
// W = -11 (the default), Plonky3 / SP1's BabyBear quartic has W = +11 (rk_params.ext_w).
constexpr uint32_t WM_RISC0 = 1073741848u;  // Montgomery form of p - 11
// schoolbook product folded through x^4 = W
RK_HD Ext mul(const Ext& a, const Ext& b, uint32_t wm = WM_RISC0) {
    const uint32_t NBETA = wm;
    uint32_t h0 = add(add(mul(a.c[1], b.c[3]), mul(a.c[2], b.c[2])), mul(a.c[3], b.c[1]));
    uint32_t h1 = add(mul(a.c[2], b.c[3]), mul(a.c[3], b.c[2]));
    uint32_t h2 = mul(a.c[3], b.c[3]);
    Ext r;
    r.c[0] = add(mul(a.c[0], b.c[0]), mul(NBETA, h0));
    r.c[1] = add(add(mul(a.c[0], b.c[1]), mul(a.c[1], b.c[0])), mul(NBETA, h1));
    r.c[2] = add(add(add(mul(a.c[0], b.c[2]), mul(a.c[1], b.c[1])), mul(a.c[2], b.c[0])), mul(NBETA, h2));
    r.c[3] = add(add(mul(a.c[0], b.c[3]), mul(a.c[1], b.c[2])), add(mul(a.c[2], b.c[1]), mul(a.c[3], b.c[0])));
    return r;
}
RK_HD Ext pow(Ext a, uint64_t e, uint32_t wm = WM_RISC0) {
    Ext r = ext_one();
    while (e) {
        if (e & 1) r = mul(r, a, wm);
        a = mul(a, a, wm);
        e >>= 1;
    }
    return r;
}
// a^-1 via the norm to the quadratic subfield Fp[y], y = x^2, y^2 = W: for a = a0 + a1 x with
// a0 = c0 + c2 y, a1 = c1 + c3 y:
//   a * (a0 - a1 x) = a0^2 - a1^2 y =: n = n0 + n1 y,
//       n0 = c0^2 + W c2^2 - 2 W c1 c3,   n1 = 2 c0 c2 - c1^2 - W c3^2
//   n * (n0 - n1 y) = n0^2 - W n1^2 in Fp
RK_HD Ext inv(const Ext& a, uint32_t wm = WM_RISC0) {
    uint32_t n0 = sub(add(mul(a.c[0], a.c[0]), mul(wm, mul(a.c[2], a.c[2]))), mul(wm, dbl(mul(a.c[1], a.c[3]))));
    uint32_t n1 = sub(sub(dbl(mul(a.c[0], a.c[2])), mul(a.c[1], a.c[1])), mul(wm, mul(a.c[3], a.c[3])));
    uint32_t d = sub(mul(n0, n0), mul(wm, mul(n1, n1)));
    uint32_t di = inv(d);
    // n^-1 = (n0 - n1 y) / d ; a^-1 = (a0 - a1 x) * n^-1
    uint32_t m0 = mul(n0, di), m1 = neg(mul(n1, di));
    // (a0 - a1 x) as Ext: (c0, -c1, c2, -c3); times (m0 + m1 y) = (m0, 0, m1, 0)
    Ext conj{{a.c[0], neg(a.c[1]), a.c[2], neg(a.c[3])}};
    Ext m{{m0, 0, m1, 0}};
    return mul(conj, m, wm);
}

RK_HD uint32_t bitrev(uint32_t x, unsigned bits) {
    if (bits == 0) return 0;
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x) >> (32 - bits);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0f0f0f0fu) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x >> 8) & 0x00ff00ffu) | ((x & 0x00ff00ffu) << 8);
    x = (x >> 16) | (x << 16);
    return x >> (32 - bits);
#endif
}

}  // namespace bb
