// Two-pass NTTs for 2^18 .. 2^22 points (the LDE shapes of a 2^16 .. 2^20-cycle segment) with
// every shape parameter fixed at compile time.
//
// A strided pass of G = k - 14 stages over 2^G x T tiles (T = 2^(14-G)) and a contiguous pass of
// 14 stages over 2^14-element blocks; one workgroup of 1024 lanes per 2^14-element tile, 16
// values per lane, rounds of up to four radix-2 stages in registers (on gfx950 nearly every VALU
// instruction costs the same 4 cycles per wave, only plain VGPR add/sub is cheaper --
// profiles/r01_ubench_isa.txt -- so instruction count is the budget):
//   * the first round of a pass takes its 16 values straight from global memory and the last
//     round stores straight back (both are 256-byte coalesced per wave instruction), so a
//     strided pass makes ONE trip through LDS and the forward contiguous pass two (round 1's
//     register-blocked kernels made one per round plus a load and a store phase);
//   * LDS addresses are base + compile-time immediate (the 4-per-64 skew is linear in the
//     register index for every round used here), no per-access address arithmetic;
//   * the four-step twiddle w_n^(+-e * bitrev(sp)) and the inverse pass's 1/n * 3^bitrev(pos)
//     come from per-size tables (one coalesced load + one product per element instead of
//     products of three partial factors); tiles of one sub-problem index run back to back
//     (column is the fastest block index) so a 64 KiB table row is shared through L2.
//
// LDS bank check (ds_read/write_b32: 32 banks, lanes 0-31 / 32-63): every round touches, per
// register index, either 32 consecutive words (L >= 6) or eight runs of 4 words at stride 68
// (L = 2: banks 4a + b, a < 8, b < 4) -- conflict-free; the b128 accesses of the inverse pass's
// last round: reads conflict-free, writes 2-way.
//
// Phase functions are RK_HD: tests/emul/emul.cpp runs them lane by lane on the CPU.
#pragma once
#include "ntt_core.hpp"

namespace nf {

// 16-byte vector with the alignment that lets the compiler pick ds_read/write_b128 and dwordx4
struct alignas(16) V4 {
    uint32_t x, y, z, w;
};

constexpr unsigned TILE_LOG = 14;
constexpr unsigned NTHR = 1024;
// LDS skew: 4 words every 64, so the stride-4 lane pattern of the L = 2 rounds spreads over the banks
RK_HD unsigned phys(unsigned e) { return e + ((e >> 6) << 2); }
constexpr unsigned LDS_WORDS = (1u << TILE_LOG) + ((1u << TILE_LOG) >> 6) * 4;

struct Args {
    uint32_t* dst;
    const uint32_t* src;
    size_t n;              // elements per column (dst)
    size_t n_src;          // elements per column (src): n >> 2 for the expanding pass
    unsigned k;            // log2 n
    unsigned count;        // columns
    const uint32_t* fs;    // contiguous passes: four-step twiddles, fs[sp << 14 | e] = w_n^(+-e * bitrev(sp))
    const uint32_t* zk;    // inverse contiguous pass: zk[pos] = 3^bitrev_k(pos) / n, or null
    uint32_t scale;        // inverse contiguous pass without zk: Montgomery 1/n
};

// ---- twiddles of one stage, compile-time stage position ----------------------------------
template <int LS, int B>
RK_HD void stage_tw(uint32_t* w, const uint32_t* tw, unsigned rlow) {
#pragma unroll
    for (int j = 0; j < (1 << B); j++) w[j] = tw[(1u << (LS + B)) + ((unsigned)j << LS) + rlow];
}
// forward (DIT) stages LS .. LS+NST-1 on 16 registers: (x, y) -> (x + y w^j, x - y w^j).
// Values stay "lazy" in [0, 2p) between stages, between rounds (LDS) and until the store:
//   t = y * w via an unsigned REDC (any u32 y, result < 2p), t and x brought to [0, p) with one
//   conditional subtraction each, then x + t and x - t + p need no reduction (both < 2p < 2^32).
// CANON_IN: the 16 inputs are canonical (fresh from global memory): the first stage skips its ucanon of x
// B0: first stage of the round to run (the earlier ones were done some other way: a broadcast in the 2x expanding pass)
template <int LS, int NST, bool CANON_IN = false, int B0 = 0>
RK_HD void dit(uint32_t* v, const uint32_t* tw, unsigned rlow) {
    static_for<B0, NST>([&](auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
        uint32_t w[1 << b];
        stage_tw<LS, b>(w, tw, rlow);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            if (m & (1 << b)) continue;
            uint32_t x = (CANON_IN && b == B0) ? v[m] : bb::ucanon(v[m]);
            uint32_t t;
            if (LS == 0 && (m & ((1 << b) - 1)) == 0)  // position 0 of its half: the twiddle is w^0 = 1
                t = (CANON_IN && b == B0) ? v[m | (1 << b)] : bb::ucanon(v[m | (1 << b)]);
            else
                t = bb::ucanon(bb::uredc64((uint64_t)v[m | (1 << b)] * w[m & ((1 << b) - 1)]));
            v[m] = x + t;
            v[m | (1 << b)] = x - t + bb::P;
        }
    });
}
// inverse (DIF) stages LS+NST-1 .. LS on 16 registers: (x, y) -> (x + y, (x - y) w^-j), canonical values;
// x - y in (-p, p) needs no reduction before the signed Montgomery product
template <int LS, int NST>
RK_HD void dif(uint32_t* v, const uint32_t* tw, unsigned rlow) {
    static_for<0, NST>([&](auto bc) __attribute__((always_inline)) {
        constexpr int b = NST - 1 - decltype(bc)::value;
        uint32_t w[1 << b];
        stage_tw<LS, b>(w, tw, rlow);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            if (m & (1 << b)) continue;
            uint32_t x = v[m], y = v[m | (1 << b)];
            v[m] = bb::add(x, y);
            if (LS == 0 && (m & ((1 << b) - 1)) == 0)  // w^-0 = 1
                v[m | (1 << b)] = bb::sub(x, y);
            else
                v[m | (1 << b)] = bb::canon(bb::smul((int32_t)(x - y), (int32_t)w[m & ((1 << b) - 1)]));
        }
    });
}

// ---- LDS access of one round: registers m = 0..15 at tile index e0 | m << L ---------------
// phys(e0 + (m << L)) == phys(e0) + m * ((1 << L) + (1 << L >> 4 & ~3...)): linear for L >= 6 and L <= 2
template <int L>
constexpr unsigned lds_step() {
    static_assert(L >= 6 || L + 4 <= 6, "the skew is not linear in the register index for this round");
    return (1u << L) + (((1u << L) >> 6) << 2);
}
template <int L>
RK_HD unsigned round_e0(unsigned tid) {
    if constexpr (L >= 10) return tid;  // e = m << 10 | tid
    else return ((tid >> L) << (L + 4)) | (tid & ((1u << L) - 1));
}
template <int L>
RK_HD void lds_read(uint32_t* v, const uint32_t* lds, unsigned tid) {
    const uint32_t* p = lds + phys(round_e0<L>(tid));
#pragma unroll
    for (int m = 0; m < 16; m++) v[m] = p[m * lds_step<L>()];
}
template <int L>
RK_HD void lds_write(const uint32_t* v, uint32_t* lds, unsigned tid) {
    uint32_t* p = lds + phys(round_e0<L>(tid));
#pragma unroll
    for (int m = 0; m < 16; m++) p[m * lds_step<L>()] = v[m];
}

// ---- tiles -------------------------------------------------------------------------------
// contiguous passes: block = sp * count + col (column fastest: one fs row serves `count` blocks)
struct CTile {
    size_t base;      // dst element offset of tile element 0
    size_t src_base;  // src element offset (expanding pass: of source word 0)
    unsigned sp;
};
RK_HD CTile ctile_of(const Args& a, size_t block, unsigned expand_bits) {
    CTile t;
    size_t col = block % a.count;
    t.sp = (unsigned)(block / a.count);
    t.base = col * a.n + ((size_t)t.sp << TILE_LOG);
    t.src_base = col * a.n_src + (((size_t)t.sp << TILE_LOG) >> expand_bits);
    return t;
}
// strided passes: block = col * (n >> 14) + b; tile = 2^G rows (stride S = n >> G) x T columns
template <int G>
struct STile {
    size_t base;
    size_t S;
};
template <int G>
RK_HD STile<G> stile_of(const Args& a, size_t block) {
    size_t per_col = a.n >> TILE_LOG;
    size_t col = block / per_col, b = block % per_col;
    STile<G> t;
    t.S = a.n >> G;
    t.base = col * a.n + (b << (TILE_LOG - G));
    return t;
}

// =========================== forward, contiguous, 4x expanding =============================
// stages 2..13 of the 2^14-point sub-transform (stages 0, 1 of a 4x zero-padded input are a
// broadcast), then the four-step twiddle on the way out
RK_HD void fwd_contig_a(const Args& a, const ntt::Tables& tb, const CTile& t, uint32_t* lds, unsigned tid) {
    // round L = 2: e = (tid >> 2) << 6 | m << 2 | (tid & 3), value = src[e >> 2] = src[16 * (tid >> 2) + m]
    uint32_t v[16];
    const V4* s = reinterpret_cast<const V4*>(a.src + t.src_base + 16 * (size_t)(tid >> 2));
#pragma unroll
    for (int q = 0; q < 4; q++) {
        V4 u = s[q];
        v[4 * q] = u.x; v[4 * q + 1] = u.y; v[4 * q + 2] = u.z; v[4 * q + 3] = u.w;
    }
    dit<2, 4, true>(v, tb.small[0], tid & 3);
    lds_write<2>(v, lds, tid);
}
RK_HD void fwd_contig_b(const ntt::Tables& tb, uint32_t* lds, unsigned tid) {
    uint32_t v[16];
    lds_read<6>(v, lds, tid);
    dit<6, 4>(v, tb.small[0], tid & 63);
    lds_write<6>(v, lds, tid);
}
RK_HD void fwd_contig_c(const Args& a, const ntt::Tables& tb, const CTile& t, const uint32_t* lds, unsigned tid) {
    uint32_t v[16], f[16];
    const uint32_t* fs = a.fs + ((size_t)t.sp << TILE_LOG) + tid;
#pragma unroll
    for (int m = 0; m < 16; m++) f[m] = fs[m << 10];
    lds_read<10>(v, lds, tid);
    dit<10, 4>(v, tb.small[0], tid);
    uint32_t* d = a.dst + t.base + tid;
#pragma unroll
    for (int m = 0; m < 16; m++) d[m << 10] = bb::mul(v[m], f[m]);  // lazy in, canonical out
}

// =========================== forward, contiguous, not expanding ============================
// 14 stages: (0,1) | 2..5 | 6..9 | 10..13; the first round reads 16 consecutive words per lane
RK_HD void fwd_contig0_a(const Args& a, const ntt::Tables& tb, const CTile& t, uint32_t* lds, unsigned tid) {
    uint32_t v[16];
    const V4* s = reinterpret_cast<const V4*>(a.src + t.src_base + 16 * (size_t)tid);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        V4 u = s[q];
        v[4 * q] = u.x; v[4 * q + 1] = u.y; v[4 * q + 2] = u.z; v[4 * q + 3] = u.w;
    }
    dit<0, 2, true>(v, tb.small[0], 0);
    V4* p = reinterpret_cast<V4*>(lds + phys(16 * tid));
#pragma unroll
    for (int q = 0; q < 4; q++) p[q] = V4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}
// 2x expanding: stage 0 of a zero-interleaved input is a broadcast (x, 0) -> (x, x): the lane reads 8 source words
// and runs stage 1 only; the other rounds are the non-expanding ones
RK_HD void fwd_contig1_a(const Args& a, const ntt::Tables& tb, const CTile& t, uint32_t* lds, unsigned tid) {
    uint32_t v[16];
    const V4* s = reinterpret_cast<const V4*>(a.src + t.src_base + 8 * (size_t)tid);
#pragma unroll
    for (int q = 0; q < 2; q++) {
        V4 u = s[q];
        v[8 * q] = v[8 * q + 1] = u.x;
        v[8 * q + 2] = v[8 * q + 3] = u.y;
        v[8 * q + 4] = v[8 * q + 5] = u.z;
        v[8 * q + 6] = v[8 * q + 7] = u.w;
    }
    dit<0, 2, true, 1>(v, tb.small[0], 0);
    V4* p = reinterpret_cast<V4*>(lds + phys(16 * tid));
#pragma unroll
    for (int q = 0; q < 4; q++) p[q] = V4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}
RK_HD void fwd_contig0_b(const ntt::Tables& tb, uint32_t* lds, unsigned tid) {
    uint32_t v[16];
    lds_read<2>(v, lds, tid);
    dit<2, 4>(v, tb.small[0], tid & 3);
    lds_write<2>(v, lds, tid);
}

// =========================== inverse, contiguous ===========================================
// four-step twiddle on the way in, stages 13..0 as 4 | 4 | 4 | 2, then 1/n (and the zk shift)
RK_HD void inv_contig_a(const Args& a, const ntt::Tables& tb, const CTile& t, uint32_t* lds, unsigned tid) {
    uint32_t v[16], f[16];
    const uint32_t* s = a.src + t.src_base + tid;
    const uint32_t* fs = a.fs + ((size_t)t.sp << TILE_LOG) + tid;
#pragma unroll
    for (int m = 0; m < 16; m++) {
        v[m] = s[m << 10];
        f[m] = fs[m << 10];
    }
#pragma unroll
    for (int m = 0; m < 16; m++) v[m] = bb::mul(v[m], f[m]);
    dif<10, 4>(v, tb.small[1], tid);
    lds_write<10>(v, lds, tid);
}
template <int L>
RK_HD void inv_contig_mid(const ntt::Tables& tb, uint32_t* lds, unsigned tid) {
    uint32_t v[16];
    lds_read<L>(v, lds, tid);
    dif<L, 4>(v, tb.small[1], tid & ((1u << L) - 1));
    lds_write<L>(v, lds, tid);
}
RK_HD void inv_contig_d(const ntt::Tables& tb, uint32_t* lds, unsigned tid) {
    // stages 1, 0 on the lane's 16 consecutive words
    uint32_t v[16];
    V4* p = reinterpret_cast<V4*>(lds + phys(16 * tid));
#pragma unroll
    for (int q = 0; q < 4; q++) {
        V4 u = p[q];
        v[4 * q] = u.x; v[4 * q + 1] = u.y; v[4 * q + 2] = u.z; v[4 * q + 3] = u.w;
    }
    dif<0, 2>(v, tb.small[1], 0);
#pragma unroll
    for (int q = 0; q < 4; q++) p[q] = V4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}
RK_HD void inv_contig_e(const Args& a, const CTile& t, const uint32_t* lds, unsigned tid) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        unsigned e = 4 * (tid + i * NTHR);
        V4 u = *reinterpret_cast<const V4*>(lds + phys(e));
        if (a.zk) {
            size_t pos = ((size_t)t.sp << TILE_LOG) + e;  // position inside the column
            V4 z = *reinterpret_cast<const V4*>(a.zk + pos);
            u = V4{bb::mul(u.x, z.x), bb::mul(u.y, z.y), bb::mul(u.z, z.z), bb::mul(u.w, z.w)};
        } else {
            u = V4{bb::mul(u.x, a.scale), bb::mul(u.y, a.scale), bb::mul(u.z, a.scale), bb::mul(u.w, a.scale)};
        }
        *reinterpret_cast<V4*>(a.dst + t.base + e) = u;
    }
}

// =========================== strided passes (G = 4 .. 8 stages over rows) ===================
// lane: lo = tid & (T - 1) (column of the tile), r = tid >> logT (G - 4 bits)
// round at row-stage 0 (G - 4 stages): rows 16 r + m      (tile index layout L = logT)
// round at row-stage G - 4 (4 stages): rows m << (G-4) | r (tile index m << 10 | tid, L = 10)
template <int G>
RK_HD void fwd_strided_a(const Args& a, const ntt::Tables& tb, const STile<G>& t, uint32_t* lds, unsigned tid) {
    constexpr int LOGT = TILE_LOG - G;
    unsigned lo = tid & ((1u << LOGT) - 1), r = tid >> LOGT;
    uint32_t v[16];
    const uint32_t* s = a.src + t.base + (size_t)(16 * r) * t.S + lo;
#pragma unroll
    for (int m = 0; m < 16; m++) v[m] = s[(size_t)m * t.S];
    dit<0, G - 4, true>(v, tb.small[0], 0);
    lds_write<LOGT>(v, lds, tid);  // e = (16 r + m) << logT | lo = round_e0<LOGT>(tid) | m << logT
}
template <int G>
RK_HD void fwd_strided_b(const Args& a, const ntt::Tables& tb, const STile<G>& t, const uint32_t* lds, unsigned tid) {
    constexpr int LOGT = TILE_LOG - G;
    unsigned lo = tid & ((1u << LOGT) - 1), r = tid >> LOGT;
    uint32_t v[16];
    if constexpr (G > 4) {
        lds_read<10>(v, lds, tid);
    } else {
        const uint32_t* s = a.src + t.base + lo;
#pragma unroll
        for (int m = 0; m < 16; m++) v[m] = s[(size_t)m * t.S];
    }
    dit<G - 4, 4, G == 4>(v, tb.small[0], r);
    uint32_t* d = a.dst + t.base + (size_t)r * t.S + lo;
#pragma unroll
    for (int m = 0; m < 16; m++) d[((size_t)m << (G - 4)) * t.S] = bb::ucanon(v[m]);
}
template <int G>
RK_HD void inv_strided_a(const Args& a, const ntt::Tables& tb, const STile<G>& t, uint32_t* lds, unsigned tid) {
    constexpr int LOGT = TILE_LOG - G;
    unsigned lo = tid & ((1u << LOGT) - 1), r = tid >> LOGT;
    uint32_t v[16];
    const uint32_t* s = a.src + t.base + (size_t)r * t.S + lo;
#pragma unroll
    for (int m = 0; m < 16; m++) v[m] = s[((size_t)m << (G - 4)) * t.S];
    dif<G - 4, 4>(v, tb.small[1], r);
    if constexpr (G > 4) {
        lds_write<10>(v, lds, tid);
    } else {
        uint32_t* d = a.dst + t.base + lo;
#pragma unroll
        for (int m = 0; m < 16; m++) d[(size_t)m * t.S] = v[m];
    }
}
template <int G>
RK_HD void inv_strided_b(const Args& a, const ntt::Tables& tb, const STile<G>& t, const uint32_t* lds, unsigned tid) {
    constexpr int LOGT = TILE_LOG - G;
    unsigned lo = tid & ((1u << LOGT) - 1), r = tid >> LOGT;
    uint32_t v[16];
    lds_read<LOGT>(v, lds, tid);
    dif<0, G - 4>(v, tb.small[1], 0);
    uint32_t* d = a.dst + t.base + (size_t)(16 * r) * t.S + lo;
#pragma unroll
    for (int m = 0; m < 16; m++) d[(size_t)m * t.S] = v[m];
}

// ---- table entries (generated on the device once per size, on the host by the emulator) -----
// fs[sp << 14 | e] = w_n^(+-e * bitrev_{k-14}(sp))
RK_HD uint32_t fs_entry(const ntt::Tables& tb, unsigned k, int dir, size_t idx) {
    unsigned sp = (unsigned)(idx >> TILE_LOG), e = (unsigned)(idx & ((1u << TILE_LOG) - 1));
    uint32_t k1 = bb::bitrev(sp, k - TILE_LOG);
    return ntt::root_pow(tb, dir, (e * k1) << (ntt::LAMBDA - k));
}
// zk[pos] = 3^bitrev_k(pos) / n
RK_HD uint32_t zk_entry(const ntt::Tables& tb, unsigned k, uint32_t scale, size_t pos) {
    return bb::mul(scale, ntt::pow3(tb, bb::bitrev((uint32_t)pos, k)));
}

// shapes served: 2^18 .. 2^22 points, forward with 4x or 2x expansion or none, 16-byte aligned buffers
inline bool usable(unsigned k, unsigned expand_bits, bool aligned16) {
    if (!aligned16 || k < TILE_LOG + 4 || k > TILE_LOG + 8) return false;
    return expand_bits <= 2;
}

}  // namespace nf
