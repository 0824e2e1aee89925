// Poseidon2 over BabyBear, x^7, R_F = 8 -- width 24 / rate 16 / R_P = 21 is the
// permutation of risc0's default "poseidon2" hash suite; width 16 / R_P = 13 is SP1's shape (risc0-zkp
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
//   * external layer: exact signed 64-bit accumulation of the signed-lazy S-box outputs
//     (v_mad_i64_i32 with small literal multipliers, 64-bit adds) followed by one signed REDC per
//     cell -- no modular addition and no canonicalisation anywhere in a full round.
//     REDC divides by 2^32, so inside a block of four full rounds the state carries a known
//     scale factor 2^(32 e) (e = 0, -7, -56, -399, then -2800): the round constants are
//     pre-scaled per round.  The first block's factor is absorbed by the constants of the
//     partial rounds (one product for cell 0), the second block's by one product per OUTPUT cell
//     after the last layer, so a caller that keeps only some cells (digest, sponge capacity)
//     pays only for those;
//   * internal layers: three instructions per cell per partial round, the addition of the cell sum
//     riding inside the Montgomery reduction (see partial_rounds()).
#pragma once
#include "bb.hpp"

namespace p2 {

constexpr int OUT = 8;              // digest cells
constexpr int ROUNDS_HALF_FULL = 4;  // R_F = 8

// acc + x * c for a wave-uniform constant c (scalar register operand): one v_mad_u64_u32.
// Plain C on purpose: an inline-asm mad makes the hazard recogniser pad every product with
// s_nop 1 (its vcc write followed by an opaque SGPR read).
RK_HD uint64_t mad_sc(uint64_t acc, uint32_t x, uint32_t c) { return acc + (uint64_t)x * (uint64_t)c; }
RK_HD uint64_t mul_sc(uint32_t x, uint32_t c) { return (uint64_t)x * (uint64_t)c; }

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

// signed forms: acc + x * K on signed-lazy values (v_mad_i64_i32)
template <int K>
RK_HD int64_t smadk(int32_t x, int64_t acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "n"(K) : "vcc");
    return acc;
#else
    return acc + (int64_t)x * (int64_t)K;
#endif
}
template <int K>
RK_HD int64_t smulk(int32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    int64_t r;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(x), "n"(K) : "vcc");
    return r;
#else
    return (int64_t)x * (int64_t)K;
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
typedef uint32_t sgpr16 __attribute__((ext_vector_type(16)));
struct KStream {
    const uint32_t* p;
    sgpr16 cur, nxt;
    __device__ __forceinline__ explicit KStream(const uint32_t* base) : p(base) {
        asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cur) : "s"(p));
        asm volatile("s_load_dwordx16 %0, %1, 0x40" : "=s"(nxt) : "s"(p));
    }
    // `acc` (the running sum the constant is about to be multiplied into) is threaded through the
    // wait so that the scheduler keeps it between the products of two chunks: without that it
    // bunches several wait + load pairs together and nothing overlaps the load latency
    template <int POS>
    __device__ __forceinline__ uint32_t get(uint64_t& acc) {
        uint32_t c = cur[POS & 15];
        if constexpr ((POS & 15) == 15) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(nxt), "+v"(acc));
            cur = nxt;
            // the chunk after next (byte offset as an immediate: no pointer arithmetic to hoist)
            asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(nxt) : "s"(p), "n"((POS / 16 + 2) * 64));
        }
        return c;
    }
    __device__ __forceinline__ void drain() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(nxt)); }
};
#else
struct KStream {
    const uint32_t* p;
    explicit KStream(const uint32_t* base) : p(base) {}
    template <int POS>
    uint32_t get(uint64_t&) { return p[POS]; }
    void drain() {}
};
#endif

// acc += x * (constant at stream position POS)
template <int POS>
RK_HD void pr_fma(KStream& ks, uint64_t& acc, uint32_t x) {
    uint32_t c = ks.template get<POS>(acc);
    acc = mad_sc(acc, x, c);
}

// One Poseidon2 instance family: width W (rate W - 8), RP partial rounds, 4x4 block M4K of the
// external layer (0: the Poseidon2 paper's / risc0's, 1: circ(2,3,1,1), Plonky3's MDSMat4).
// risc0's suite is Core<24, 21, 0>; SP1 / Plonky3's BabyBear permutation is Core<16, 13, 1>.
template <int W, int RP, int M4K>
struct Core {
static constexpr int CELLS = W;
static constexpr int RATE = W - 8;
static constexpr int ROUNDS_PARTIAL = RP;
static constexpr int M4_KIND = M4K;

// per round r: 23 pairs d_i^r, r pairs c_(r-1-j), then d_0 + 1; at the end 23 x (d_i^21, d_i^(20-j))
#if defined(RK_P2_CLOSED_FORM)
static constexpr int PR_STREAM_USED = ROUNDS_PARTIAL * (2 * (CELLS - 1) + 1) + ROUNDS_PARTIAL * (ROUNDS_PARTIAL - 1) +
                               (CELLS - 1) * 2 * (ROUNDS_PARTIAL + 1);
#else
static constexpr int PR_STREAM_USED = ROUNDS_PARTIAL * CELLS;  // one multiplier per cell per round
#endif
static constexpr int PR_STREAM_WORDS = (PR_STREAM_USED + 15) / 16 * 16 + 16;

struct Consts {
    // the instance (Montgomery form), set by the caller
    uint32_t rc_ext[2 * ROUNDS_HALF_FULL * CELLS];
    uint32_t rc_int[ROUNDS_PARTIAL];
    uint32_t diag[CELLS];
    // derived by derive()
    uint32_t rc_ext_in[2 * ROUNDS_HALF_FULL * CELLS];  // rc * scale(round) as the S-box input offset: rc - p after an
                                                       // unsigned state (rounds 0, 4), centred in (-p/2, p/2] otherwise
    uint32_t rc_int_mp[ROUNDS_PARTIAL];                // rc - p
    uint32_t fix[2];                                   // block-end rescale constants 2^(32 (2 - e_end)), plain residues
    uint32_t fix0_nq;                                  // fix[0] * (-p^-1) mod 2^32 (bb::umul_const companion)
    uint32_t fix1_q;                                   // fix[1] * p^-1 mod 2^32 (bb::smul_const companion)
    uint32_t sig0_c, sig0_nq;                          // fix[0] * 2^32 and its companion: entry sum -> Montgomery form
    uint32_t r3_c, r3_nq;                              // 2^96 mod p and its companion: S -> S * 2^32 (see partial_rounds())
    // partial rounds in closed form (see partial_rounds()): the constants in the order the
    // code consumes them, each as a pair {c, c * 2^16 mod p} for the low / high 16-bit halves
    // of the variable it multiplies (+ one chunk of padding for the read-ahead)
    uint32_t pr_stream[PR_STREAM_WORDS];
};

// exponent e of the scale 2^(32 e) carried by the state at the S-box input of each full round
// (first block follows the initial external layer + REDC; second block starts from Montgomery form)
static constexpr int scale_exp(int r) {
    constexpr int t[2 * ROUNDS_HALF_FULL] = {0, -7, -56, -399, 1, 0, -7, -56};
    return t[r];
}

static inline void derive(Consts& k) {
    const uint32_t Rm = bb::encode(bb::ONE);  // Montgomery form of the field element 2^32
    const uint32_t Rinv_m = bb::inv(Rm);
    auto rpow = [&](long e) {  // Montgomery form of 2^(32 e)
        return e >= 0 ? bb::pow(Rm, (uint64_t)e) : bb::pow(Rinv_m, (uint64_t)(-e));
    };
    for (int r = 0; r < 2 * ROUNDS_HALF_FULL; r++) {
        // stored residue v = rc * 2^32 (Montgomery form).  The S-box input of round r holds the
        // residue a * 2^(32 e) for the true state a, so the residue to add is rc * 2^(32 e):
        // bb::mul(v, f) = v * f / 2^32 with the residue f = 2^(32 e) = rpow(e - 1).
        uint32_t f = rpow((long)scale_exp(r) - 1);
        const bool after_unsigned = r % ROUNDS_HALF_FULL == 0;  // input in [0, p + 2^22): offset rc - p
        for (int i = 0; i < CELLS; i++) {
            uint32_t c = bb::mul(k.rc_ext[r * CELLS + i], f);
            // otherwise the input is a signed REDC output, |x| <= p/2 + 53: centred offset, |x + rc| < 2^31
            k.rc_ext_in[r * CELLS + i] = (after_unsigned || c > bb::P / 2) ? c - bb::P : c;
        }
    }
    for (int i = 0; i < ROUNDS_PARTIAL; i++) k.rc_int_mp[i] = k.rc_int[i] - bb::P;
    // a block ends with the state scaled by 2^(32 e_end), e_end = 7 e - 7 for the last round's e;
    // x -> x * K / 2^32 with K = 2^(32 (2 - e_end)) (plain residue) gives Montgomery form
    for (int b = 0; b < 2; b++) {
        long e = scale_exp(b * ROUNDS_HALF_FULL + ROUNDS_HALF_FULL - 1);
        k.fix[b] = bb::decode(rpow(2 - (7 * e - 7)));
    }
    k.fix0_nq = k.fix[0] * (0u - bb::MPRIME);
    k.fix1_q = k.fix[1] * bb::MPRIME;
    k.sig0_c = bb::encode(k.fix[0]);
    k.sig0_nq = k.sig0_c * (0u - bb::MPRIME);
    k.r3_c = bb::encode(bb::encode(bb::ONE));
    k.r3_nq = k.r3_c * (0u - bb::MPRIME);
#if defined(RK_P2_CLOSED_FORM)
    // closed-form partial rounds: powers of the diagonal (Montgomery residues: they multiply a
    // Montgomery-form variable and the sum goes through one REDC).  The constants that multiply
    // the ENTRY cells also carry fix[0]: those cells arrive scaled by the first block.
    const uint32_t two16 = bb::encode(65536u);
    uint32_t pw[CELLS - 1][ROUNDS_PARTIAL + 1];
    for (int i = 1; i < CELLS; i++) {
        pw[i - 1][0] = bb::ONE;
        for (int m = 1; m <= ROUNDS_PARTIAL; m++) pw[i - 1][m] = bb::mul(pw[i - 1][m - 1], k.diag[i]);
    }
    uint32_t csum[ROUNDS_PARTIAL];
    for (int m = 0; m < ROUNDS_PARTIAL; m++) {
        csum[m] = 0;
        for (int i = 0; i < CELLS - 1; i++) csum[m] = bb::add(csum[m], pw[i][m]);
    }
    int n = 0;
    auto put = [&](uint32_t c) { k.pr_stream[n++] = c; k.pr_stream[n++] = bb::mul(c, two16); };
    for (int r = 0; r < ROUNDS_PARTIAL; r++) {
        for (int i = 0; i < CELLS - 1; i++) put(bb::mul(pw[i][r], k.fix[0]));
        for (int j = 0; j < r; j++) put(csum[r - 1 - j]);
        k.pr_stream[n++] = bb::add(k.diag[0], bb::ONE);
    }
    for (int i = 0; i < CELLS - 1; i++) {
        put(bb::mul(pw[i][ROUNDS_PARTIAL], k.fix[0]));
        for (int j = 0; j < ROUNDS_PARTIAL; j++) put(pw[i][ROUNDS_PARTIAL - 1 - j]);
    }
#else
    // direct partial rounds: round r, cell i multiplies by d_i (Montgomery residue); in round 0 the cells
    // other than 0 still carry the first block's scale, so their multiplier is d_i * fix[0] / 2^32
    int n = 0;
    for (int r = 0; r < ROUNDS_PARTIAL; r++)
        for (int i = 0; i < CELLS; i++) k.pr_stream[n++] = (r == 0 && i > 0) ? bb::mul(k.diag[i], k.fix[0]) : k.diag[i];
#endif
    while (n < PR_STREAM_WORDS) k.pr_stream[n++] = 0;
}

// External layer circ(2*M4, M4, ..., M4), M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]], on
// canonical cells, exact: w[i] < 112 p < 2^38.  Then s[i] = w[i] * 2^-32 (mod p) in [0, p + 53).
static RK_HD void m_ext_redc(uint32_t* s) {
    uint64_t w[CELLS];
#pragma unroll
    for (int i = 0; i < CELLS; i += 4) {
        uint32_t a = s[i], b = s[i + 1], c = s[i + 2], d = s[i + 3];
        uint32_t t0 = a + b, t1 = c + d;                     // < 2p < 2^32
        if constexpr (M4K == 0) {
            uint64_t u1 = madk<1>(t1, madk<6>(b, mulk<4>(a)));  // 4a + 6b +  c +  d
            uint64_t u0 = madk<2>(d, madk<1>(t0, u1));          // 5a + 7b +  c + 3d
            uint64_t u3 = madk<1>(t0, madk<6>(d, mulk<4>(c)));  //  a +  b + 4c + 6d
            uint64_t u2 = madk<2>(b, madk<1>(t1, u3));          //  a + 3b + 5c + 7d
            w[i] = u0; w[i + 1] = u1; w[i + 2] = u2; w[i + 3] = u3;
        } else {  // circ(2, 3, 1, 1): every output is the block sum plus one cell plus twice the next
            uint64_t sum = madk<1>(t1, mulk<1>(t0));
            w[i] = madk<2>(b, madk<1>(a, sum));      // 2a + 3b +  c +  d
            w[i + 1] = madk<2>(c, madk<1>(b, sum));  //  a + 2b + 3c +  d
            w[i + 2] = madk<2>(d, madk<1>(c, sum));  //  a +  b + 2c + 3d
            w[i + 3] = madk<2>(a, madk<1>(d, sum));  // 3a +  b +  c + 2d
        }
    }
    uint64_t t[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        t[j] = w[j];
#pragma unroll
        for (int b = 4; b < CELLS; b += 4) t[j] += w[b + j];
    }
#pragma unroll
    for (int i = 0; i < CELLS; i++) s[i] = bb::uredc64(w[i] + t[i & 3]);
}

// The same layer on signed-lazy cells |y| < p (S-box outputs), exact in int64 (|w| < 112 p), then
// r[i] = w[i] * 2^-32 (mod p) as a signed value, |r| <= p/2 + 53.  M4 follows the Poseidon2 paper's
// add / double schedule: the first two sums and the two doublings take the 32-bit cells through
// v_mad_i64_i32 (sums like a + b do not fit 32 bits here), the rest are 64-bit shift-adds
// (v_lshl_add_u64) -- ten instructions per four cells, against two per CELL for a canonicalisation
// that would let the unsigned form above be used.
static RK_HD void m_ext_redc_s(const int32_t* y, int32_t* r) {
    int64_t w[CELLS];
#pragma unroll
    for (int i = 0; i < CELLS; i += 4) {
        int32_t a = y[i], b = y[i + 1], c = y[i + 2], d = y[i + 3];
        int64_t t0 = smadk<1>(b, smulk<1>(a));  //  a +  b
        int64_t t1 = smadk<1>(d, smulk<1>(c));  //  c +  d
        if constexpr (M4K == 0) {
            int64_t t2 = smadk<2>(b, t1);           // 2b +  c +  d
            int64_t t3 = smadk<2>(d, t0);           //  a +  b + 2d
            int64_t t4 = (t1 << 2) + t3;            //  a +  b + 4c + 6d
            int64_t t5 = (t0 << 2) + t2;            // 4a + 6b +  c +  d
            w[i] = t3 + t5;                         // 5a + 7b +  c + 3d
            w[i + 1] = t5;
            w[i + 2] = t2 + t4;                     //  a + 3b + 5c + 7d
            w[i + 3] = t4;
        } else {  // circ(2, 3, 1, 1)
            int64_t sum = t0 + t1;
            w[i] = smadk<2>(b, smadk<1>(a, sum));
            w[i + 1] = smadk<2>(c, smadk<1>(b, sum));
            w[i + 2] = smadk<2>(d, smadk<1>(c, sum));
            w[i + 3] = smadk<2>(a, smadk<1>(d, sum));
        }
    }
    int64_t t[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        t[j] = w[j];
#pragma unroll
        for (int b = 4; b < CELLS; b += 4) t[j] += w[b + j];
    }
#pragma unroll
    for (int i = 0; i < CELLS; i++) r[i] = bb::redc64(w[i] + t[i & 3]);
}

// (x + c)^7 * 2^(-6*32) as a signed-lazy value, |result| < p.  x + c must fit an int32: x unsigned in
// [0, p + 2^22) with c = rc - p, or x a signed REDC output (|x| <= p/2 + 53) with c centred.
static RK_HD int32_t sbox7_lazy(uint32_t x, uint32_t c_mp) {
    int32_t s = (int32_t)(x + c_mp);
    int32_t s2 = bb::smul(s, s);
    int32_t s3 = bb::smul(s2, s);
    int32_t s6 = bb::smul(s3, s3);
    return bb::smul(s6, s);
}
// One full round on the 32-bit patterns of the cells (unsigned after the first layer / the partial
// rounds, signed otherwise: the offset form of the round constants follows, see derive()).
static RK_HD void full_round(uint32_t* s, const Consts& k, int r) {
    int32_t y[CELLS], o[CELLS];
#pragma unroll
    for (int i = 0; i < CELLS; i++) y[i] = sbox7_lazy(s[i], k.rc_ext_in[r * CELLS + i]);
    m_ext_redc_s(y, o);
#pragma unroll
    for (int i = 0; i < CELLS; i++) s[i] = (uint32_t)o[i];
}

// All 21 partial rounds in closed form.  Entry: any 32-bit representatives of the cells, scaled by
// the first block of full rounds (see permute()); exit: Montgomery form, cells in [0, p + 2^22).
// With v = cells 1..23 at entry, y_k the S-box output of
// round k and S_k = y_k + sum(cells 1..23 before round k):
//     cells_i before round k   = d_i^k v_i + sum_{j<k} d_i^(k-1-j) S_j
//     sum of them              = sum_i d_i^k v_i + sum_{j<k} c_(k-1-j) S_j,     c_m = sum_i d_i^m
//     cell 0 after round k     = (d_0 + 1) y_k + that sum
//     cells_i after round 20   = d_i^21 v_i + sum_j d_i^(20-j) S_j
// so no cell is touched between entry and exit: each round is one dot product with constant
// vectors, and every product is accumulated exactly in 64 bits -- variables are split into
// 16-bit halves, constants come as {c, c 2^16}, so a term is < 2^47 and one v_mad_u64_u32 --
// with one REDC per sum.  ~3.0 k instructions instead of ~4.1 k for 21 rounds of
// multiply / reduce / add on every cell.
// Reader of Consts::pr_stream.  On the device the constants live in scalar registers: chunks of
// 16 are fetched with s_load_dwordx16 one chunk ahead of their use (the compiler's own scheduling
// of ~2400 scalar loads spills SGPRs), and the wait is attached to the chunk's registers so that
// no use can move above it.  Positions are consumed strictly in order.
#if defined(RK_P2_CLOSED_FORM)
static RK_HD void partial_rounds(uint32_t* s, const Consts& k) {
    constexpr int NV = CELLS - 1;
    uint32_t vlo[NV], vhi[NV], slo[ROUNDS_PARTIAL], shi[ROUNDS_PARTIAL];
#pragma unroll
    for (int i = 0; i < NV; i++) {
        vlo[i] = s[i + 1] & 0xffffu;
        vhi[i] = s[i + 1] >> 16;
    }
    KStream ks(k.pr_stream);
    // cell 0 feeds an S-box, so it needs its true (Montgomery) value: one product by fix[0]; the other
    // cells only enter linear forms, whose constants carry fix[0] (derive())
    uint32_t x0 = bb::ucanon(bb::umul_const(s[0], k.fix[0], k.fix0_nq));
    static_for<0, ROUNDS_PARTIAL>([&](auto rc) __attribute__((always_inline)) {
        constexpr int R = decltype(rc)::value;
        constexpr int BASE = (2 * NV + 1) * R + R * (R - 1);
        uint32_t y = bb::canon(sbox7_lazy(x0, k.rc_int_mp[R]));
        uint64_t acc = 0;
        pr_fma<BASE>(ks, acc, vlo[0]);
        pr_fma<BASE + 1>(ks, acc, vhi[0]);
        static_for<1, NV>([&](auto ic) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value;
            pr_fma<BASE + 2 * I>(ks, acc, vlo[I]);
            pr_fma<BASE + 2 * I + 1>(ks, acc, vhi[I]);
        });
        static_for<0, R>([&](auto jc) __attribute__((always_inline)) {
            constexpr int J = decltype(jc)::value;
            pr_fma<BASE + 2 * NV + 2 * J>(ks, acc, slo[J]);
            pr_fma<BASE + 2 * NV + 2 * J + 1>(ks, acc, shi[J]);
        });
        // acc < 88 * 2^47: sigma < p + 2^22, S = y + sigma < 2p + 2^22 < 2^32 (any representative
        // serves: only its halves are used)
        uint32_t S = y + bb::uredc64(acc);
        slo[R] = S & 0xffffu;
        shi[R] = S >> 16;
        // acc + (d_0 + 1) y < 2^54 + 2^62: REDC < 2p
        pr_fma<BASE + 2 * NV + 2 * R>(ks, acc, y);
        x0 = bb::ucanon(bb::uredc64(acc));
    });
    s[0] = x0;
    constexpr int FIN = (2 * NV + 1) * ROUNDS_PARTIAL + ROUNDS_PARTIAL * (ROUNDS_PARTIAL - 1);
    static_for<0, NV>([&](auto ic) __attribute__((always_inline)) {
        constexpr int I = decltype(ic)::value;
        constexpr int BASE = FIN + I * 2 * (ROUNDS_PARTIAL + 1);
        uint64_t acc = 0;
        pr_fma<BASE>(ks, acc, vlo[I]);
        pr_fma<BASE + 1>(ks, acc, vhi[I]);
        static_for<0, ROUNDS_PARTIAL>([&](auto jc) __attribute__((always_inline)) {
            constexpr int J = decltype(jc)::value;
            pr_fma<BASE + 2 + 2 * J>(ks, acc, slo[J]);
            pr_fma<BASE + 2 + 2 * J + 1>(ks, acc, shi[J]);
        });
        s[I + 1] = bb::uredc64(acc);  // < p + 2^22
    });
    ks.drain();
}

#else
// The partial rounds, one at a time, at three instructions per cell per round.  With S the sum of
// the cells (after the S-box on cell 0) the layer is y_i = d_i x_i + S.  In Montgomery form
// (everything times 2^32) and with SM2 = S * 2^32 (mod p) as a plain 32-bit addend,
//     y_i = REDC(d_i * x_i + SM2) = (d_i x_i + S 2^32) / 2^32
// is one v_mad_u64_u32 whose 64-bit addend is SM2 zero-extended, plus the two instructions of the
// REDC: the addition of S rides inside the reduction, and no canonicalisation is needed because
// the product tolerates any 32-bit x (d_i * x + SM2 + q p < 2^64 for x, SM2 < 2^32 - p).  Per round:
// 17 for the S-box chain of cell 0, 6 to turn the 64-bit sum into SM2 (REDC, then a product by
// 2^96), 3 x CELLS for the cells and CELLS - 1 to accumulate the next sum: 118 at width 24 against
// ~144 for the closed form this replaces (all rounds as dot products with constant vectors:
// ~3.0 k instructions for 21 rounds, kept behind RK_P2_CLOSED_FORM).
// Entry: any 32-bit representatives scaled by the first block (see permute()); exit: Montgomery
// form, canonical.
static RK_HD void partial_rounds(uint32_t* s, const Consts& k) {
    KStream ks(k.pr_stream);
    // cell 0 feeds an S-box: its true (Montgomery) value, one product by fix[0]
    uint32_t x0 = bb::ucanon(bb::umul_const(s[0], k.fix[0], k.fix0_nq));
    // sum of the other cells: their representatives add up in 64 bits; REDC and one product by
    // fix[0] * 2^32 give the sum in Montgomery form (< 2p)
    uint64_t sig = 0;
#pragma unroll
    for (int i = 1; i < CELLS; i++) sig = bb::acc_u32(sig, s[i]);
    sig = bb::umul_const(bb::uredc64(sig), k.sig0_c, k.sig0_nq);
    static_for<0, ROUNDS_PARTIAL>([&](auto rc) __attribute__((always_inline)) {
        constexpr int R = decltype(rc)::value;
        constexpr int BASE = R * CELLS;
        const uint32_t y0 = bb::canon(sbox7_lazy(x0, k.rc_int_mp[R]));
        const uint64_t S = bb::acc_u32(sig, y0);                                       // < 2^38
        const uint32_t sm2 = bb::umul_const(bb::uredc64(S), k.r3_c, k.r3_nq);          // S * 2^32 mod p, < 2p
        uint64_t t0 = sm2;
        pr_fma<BASE>(ks, t0, y0);
        x0 = bb::ucanon(bb::uredc64(t0));
        uint64_t nsig = 0;
        static_for<1, CELLS>([&](auto ic) __attribute__((always_inline)) {
            constexpr int I = decltype(ic)::value;
            uint64_t t = sm2;
            pr_fma<BASE + I>(ks, t, s[I]);
            s[I] = bb::uredc64(t);  // <= 2p
            nsig = bb::acc_u32(nsig, s[I]);
        });
        sig = nsig;
    });
    s[0] = x0;
#pragma unroll
    for (int i = 1; i < CELLS; i++) s[i] = bb::ucanon(s[i]);
    ks.drain();
}
#endif

static RK_HD void permute(uint32_t* s, const Consts& k) {
    m_ext_redc(s);  // canonical Montgomery input -> plain residues (scale 2^0), cells in [0, p + 53)
#pragma unroll 1
    for (int r = 0; r < ROUNDS_HALF_FULL; r++) full_round(s, k, r);
    // signed cells (|x| <= p/2 + 53) scaled by 2^(32 * -2800): + p gives non-negative representatives,
    // which is all partial_rounds() needs (it splits them into halves and owns the scale)
#pragma unroll
    for (int i = 0; i < CELLS; i++) s[i] += bb::P;
    partial_rounds(s, k);  // Montgomery form again, cells in [0, p + 2^22)
#pragma unroll 1
    for (int r = ROUNDS_HALF_FULL; r < 2 * ROUNDS_HALF_FULL; r++) full_round(s, k, r);
    // per-cell rescale to Montgomery form + canonical range: cells the caller never reads cost nothing
#pragma unroll
    for (int i = 0; i < CELLS; i++) s[i] = bb::canon(bb::smul_const((int32_t)s[i], (int32_t)k.fix[1], k.fix1_q));
}

};  // struct Core

// risc0's instance under the names the rest of the library grew up with
using C24 = Core<24, 21, 0>;
constexpr int CELLS = C24::CELLS;
constexpr int RATE = C24::RATE;
constexpr int ROUNDS_PARTIAL = C24::ROUNDS_PARTIAL;
using Consts = C24::Consts;
inline void derive(Consts& k) { C24::derive(k); }
RK_HD void permute(uint32_t* s, const Consts& k) { C24::permute(s, k); }

}  // namespace p2
