// Register-blocked NTT passes for full 2^14-element tiles (the hot shapes of a 2^20-cycle
// segment: 2^20 = 2^6 x 2^14, 2^22 = 2^8 x 2^14).
//
// Same tiling as ntt_core.hpp (one workgroup per 2^g x T tile, T = 2^(14-g), 64 KiB + skew of
// LDS, 1024 lanes x 16 elements) but the g radix-2 stages run in rounds of up to four stages
// on 16 values held in registers, so the tile makes ceil(g/4) LDS round trips instead of g,
// index arithmetic is paid once per round, and each lane needs 15 twiddle words per round.
// On gfx950 nearly every VALU instruction costs the same 4 cycles per wave (only plain VGPR
// add/sub is cheaper, profiles/r01_ubench_isa.txt): instruction count is the budget.
//
// Two-pass plans only (strided pass + contiguous pass).  The four-step twiddle
// w_n^(+-position * bitrev(sub-problem)) lives in the CONTIGUOUS pass (on its load for the
// inverse transform, on its store for the forward one) where it is a per-lane geometric
// progression: one table lookup per lane plus products with block-uniform factors.
//
// Phase functions are RK_HD and are emulated lane by lane in tests/emul/emul.cpp.
#pragma once
#include "ntt_core.hpp"

namespace r16 {

constexpr unsigned TILE_LOG = 14;
constexpr unsigned NTHR = 1024;
constexpr unsigned EPT = 16;
// LDS skew: 4 words every 64 so stride-4 / stride-16 lane patterns spread over the banks
RK_HD unsigned phys(unsigned e) { return e + ((e >> 6) << 2); }
constexpr unsigned LDS_WORDS = (1u << TILE_LOG) + ((1u << TILE_LOG) >> 6) * 4;

struct Args {
    uint32_t* dst;
    const uint32_t* src;
    size_t n;              // elements per column (dst)
    size_t n_src;          // elements per column (src): n >> expand_bits for the expanding pass
    unsigned k;            // log2 n
    unsigned g;            // stages of this pass; logT = 14 - g
    unsigned g_outer;      // contiguous pass of a two-pass plan: stages of the strided pass, else 0
    unsigned expand_bits;  // forward contiguous pass only
    uint32_t scale;        // inverse contiguous pass: Montgomery 1/n (always set there)
    unsigned zk;           // inverse contiguous pass: 1 = fuse the zk shift 3^bitrev(pos)
};

struct Tile {
    size_t base;      // element offset of tile element 0 in dst
    size_t src_base;  // same in src
    size_t S;         // element stride between consecutive hi (1 for the contiguous pass)
    unsigned sp;      // index of the 2^14 sub-problem inside the column (contiguous pass)
};
// strided pass: tile = all 2^g values of hi for T consecutive positions; contiguous: one block of 2^14
RK_HD Tile tile_of(const Args& a, size_t block) {
    Tile t;
    unsigned logT = TILE_LOG - a.g;
    size_t per_col = a.n >> TILE_LOG;
    size_t col = block / per_col, b = block % per_col;
    if (logT == 0) {
        t.S = 1;
        t.sp = (unsigned)b;
        t.base = col * a.n + (b << TILE_LOG);
        t.src_base = col * a.n_src + ((b << TILE_LOG) >> a.expand_bits);
    } else {
        t.S = a.n >> a.g;
        t.sp = 0;
        t.base = col * a.n + (b << logT);
        t.src_base = t.base;
    }
    return t;
}

// ---- one round of NST stages on 16 registers --------------------------------------------
// Lane `tid` owns tile elements (hi(m), lo), m = 0..15, hi(m) = [rhi | m | rlow] with the
// 4-bit field m at bit `ls` of hi.  Stage halves are 2^(ls+b) (in units of hi).
struct RoundIdx {
    unsigned e0;     // logical tile index of m = 0
    unsigned estep;  // logical index step per m
    unsigned rlow;   // hi & (2^ls - 1)
    unsigned p0b;    // byte offset of phys(e0)
    unsigned pstepb; // byte step of phys() per m when `linear`
    bool linear;     // phys(e0 + m * estep) == phys(e0) + m * (estep + skew(estep)) for m < 16
};
// keep a wave-uniform value in a vector register: on gfx950 a v_add_u32 with only VGPR / literal
// operands issues twice as fast as one with a scalar operand (profiles/r01_ubench_isa.txt)
RK_HD unsigned in_vgpr(unsigned x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(x));
#endif
    return x;
}
RK_HD RoundIdx round_idx(unsigned tid, unsigned g, unsigned ls) {
    unsigned logT = TILE_LOG - g;
    unsigned lo = tid & ((1u << logT) - 1), r = tid >> logT;
    unsigned rlow = r & ((1u << ls) - 1), rhi = r >> ls;
    unsigned hi0 = (rhi << (ls + 4)) | rlow;
    RoundIdx x;
    x.e0 = (hi0 << logT) + lo;
    x.estep = 1u << (ls + logT);
    x.rlow = rlow;
    // the 4-bit field m sits at bit ls + logT of the tile index: the skew (4 words per 64) is linear
    // in m when the field lies entirely above or entirely below bit 6 -- every round of the
    // schedules below; the per-element form stays for the remaining case
    x.linear = ls + logT >= 6 || ls + logT + 4 <= 6;
    x.p0b = phys(x.e0) << 2;
    x.pstepb = in_vgpr((x.estep + ((x.estep >> 6) << 2)) << 2);
    return x;
}
RK_HD uint32_t& lds_at(uint32_t* lds, unsigned byte_off) {
    return *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(lds) + byte_off);
}
RK_HD void round_read(uint32_t* v, const uint32_t* lds, const RoundIdx& x) {
    if (x.linear) {
        unsigned o = x.p0b;
#pragma unroll
        for (int m = 0; m < 16; m++) {
            v[m] = lds_at(const_cast<uint32_t*>(lds), o);
            o += x.pstepb;
        }
    } else {
#pragma unroll
        for (int m = 0; m < 16; m++) v[m] = lds[phys(x.e0 + m * x.estep)];
    }
}
RK_HD void round_write(const uint32_t* v, uint32_t* lds, const RoundIdx& x) {
    if (x.linear) {
        unsigned o = x.p0b;
#pragma unroll
        for (int m = 0; m < 16; m++) {
            lds_at(lds, o) = v[m];
            o += x.pstepb;
        }
    } else {
#pragma unroll
        for (int m = 0; m < 16; m++) lds[phys(x.e0 + m * x.estep)] = v[m];
    }
}
// the 2^b twiddles of one stage: tw[H + (j << ls) + rlow], j < 2^b, addressed with 32-bit byte
// offsets from the (scalar) table base and a VGPR step
template <int B>
RK_HD void stage_twiddles(uint32_t* w, const uint32_t* tw, unsigned ls, unsigned rlow) {
    unsigned o = ((1u << (ls + B)) + rlow) << 2;
    const unsigned step = in_vgpr(4u << ls);
#pragma unroll
    for (int j = 0; j < (1 << B); j++) {
        w[j] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(tw) + o);
        o += step;
    }
}
// inverse (DIF): (x, y) -> (x + y, (x - y) * w^-j), largest half first
template <int NST>
RK_HD void round_dif(uint32_t* v, const uint32_t* tw /* heap-ordered inverse roots */, unsigned ls, unsigned rlow) {
    static_for<0, NST>([&](auto bc) __attribute__((always_inline)) {
        constexpr int b = NST - 1 - decltype(bc)::value;
        uint32_t w[1 << b];
        stage_twiddles<b>(w, tw, ls, rlow);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            if (m & (1 << b)) continue;
            uint32_t x = v[m], y = v[m | (1 << b)];
            v[m] = bb::add(x, y);
            // x - y in (-p, p) needs no reduction before the signed product
            v[m | (1 << b)] = bb::canon(bb::smul((int32_t)(x - y), (int32_t)w[m & ((1 << b) - 1)]));
        }
    });
}
// forward (DIT): (x, y) -> (x + y * w^j, x - y * w^j), smallest half first.
// Values stay "lazy" in [0, 2p) between stages, between rounds (LDS) and until the store phase:
//   t = y * w via an unsigned REDC (any u32 y, result < 2p), t and x brought to [0, p) with one
//   conditional subtraction each, then x + t and x - t + p need no reduction (both < 2p < 2^32).
// One v_min less per butterfly than reducing the product and both outputs.
template <int NST>
RK_HD void round_dit(uint32_t* v, const uint32_t* tw /* heap-ordered forward roots */, unsigned ls, unsigned rlow) {
    static_for<0, NST>([&](auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
        uint32_t w[1 << b];
        stage_twiddles<b>(w, tw, ls, rlow);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            if (m & (1 << b)) continue;
            uint32_t x = bb::ucanon(v[m]);
            uint32_t t = bb::ucanon(bb::uredc64((uint64_t)v[m | (1 << b)] * w[m & ((1 << b) - 1)]));
            v[m] = x + t;
            v[m | (1 << b)] = x - t + bb::P;
        }
    });
}

// ---- 16 factors A * C[c] * D[i] for the lane's elements e = 4*(tid + 1024*i) + c -----------
// (C[0] = D[0] = 1 implicitly.)  A depends on the lane, C and D only on the block.
RK_HD void apply_factors(uint32_t* r /*[i][c] = r[4*i + c]*/, uint32_t A, const uint32_t* C /*3*/, const uint32_t* D /*3*/) {
    uint32_t ac[4];
    ac[0] = A;
#pragma unroll
    for (int c = 1; c < 4; c++) ac[c] = bb::mul(A, C[c - 1]);
#pragma unroll
    for (int c = 0; c < 4; c++) r[c] = bb::mul(r[c], ac[c]);
#pragma unroll
    for (int i = 1; i < 4; i++)
#pragma unroll
        for (int c = 0; c < 4; c++) r[4 * i + c] = bb::mul(r[4 * i + c], bb::mul(ac[c], D[i - 1]));
}
// four-step twiddle of tile element e of sub-problem sp: W^(dir)(e * bitrev(sp)) scaled to order n
RK_HD void twiddle_parts(const Args& a, const ntt::Tables& tb, int dir, unsigned sp, unsigned tid, uint32_t& A,
                         uint32_t* C, uint32_t* D) {
    uint32_t k1 = bb::bitrev(sp, a.g_outer);
    unsigned sh = ntt::LAMBDA - a.k;
    A = ntt::root_pow(tb, dir, ((4u * tid) * k1) << sh);
#pragma unroll
    for (int c = 1; c < 4; c++) C[c - 1] = ntt::root_pow(tb, dir, ((unsigned)c * k1) << sh);
#pragma unroll
    for (int i = 1; i < 4; i++) D[i - 1] = ntt::root_pow(tb, dir, ((4096u * i) * k1) << sh);
}

// ---- global <-> LDS phases (16-byte accesses; 4 consecutive tile elements are consecutive in
// memory because T >= 64 or the pass is contiguous) -----------------------------------------
// strided tile rows: element e = 4 * (tid + 1024 i) lies in row (e >> logT) at column (e & tmask); the
// four rows of a lane are 4096 >> logT rows apart, i.e. n / 4 elements: 32-bit byte offsets from the
// (scalar) tile base, one add per access
RK_HD unsigned plain_off(const Args& a, const Tile& t, unsigned tid) {
    unsigned logT = TILE_LOG - a.g, tmask = (1u << logT) - 1;
    unsigned e = 4 * tid;
    return ((e >> logT) * (unsigned)t.S + (e & tmask)) << 2;
}
RK_HD void load_plain(const Args& a, const Tile& t, uint32_t* lds, unsigned tid) {
    const char* base = reinterpret_cast<const char*>(a.src + t.src_base);
    unsigned off = plain_off(a, t, tid);
    const unsigned step = in_vgpr((unsigned)(a.n >> 2) << 2);  // (4096 >> logT) * S elements = n / 4
    ntt::U4 v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        v[i] = *reinterpret_cast<const ntt::U4*>(base + off);
        off += step;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) *reinterpret_cast<ntt::U4*>(lds + phys(4 * (tid + i * NTHR))) = v[i];
}
// LAZY: the tile holds forward-transform values in [0, 2p) (round_dit): reduce on the way out
template <bool LAZY>
RK_HD void store_plain(const Args& a, const Tile& t, const uint32_t* lds, unsigned tid) {
    char* base = reinterpret_cast<char*>(a.dst + t.base);
    unsigned off = plain_off(a, t, tid);
    const unsigned step = in_vgpr((unsigned)(a.n >> 2) << 2);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ntt::U4 v = *reinterpret_cast<const ntt::U4*>(lds + phys(4 * (tid + i * NTHR)));
        if (LAZY) v = ntt::U4{bb::ucanon(v.x), bb::ucanon(v.y), bb::ucanon(v.z), bb::ucanon(v.w)};
        *reinterpret_cast<ntt::U4*>(base + off) = v;
        off += step;
    }
}
// inverse contiguous pass: load with the four-step twiddle w_n^-(e * bitrev(sp))
RK_HD void load_rev_contig(const Args& a, const ntt::Tables& tb, const Tile& t, uint32_t* lds, unsigned tid) {
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ntt::U4 v = *reinterpret_cast<const ntt::U4*>(a.src + t.src_base + 4 * (tid + i * NTHR));
        r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
    }
    if (a.g_outer && t.sp != 0) {
        uint32_t A, C[3], D[3];
        twiddle_parts(a, tb, 1, t.sp, tid, A, C, D);
        apply_factors(r, A, C, D);
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
        *reinterpret_cast<ntt::U4*>(lds + phys(4 * (tid + i * NTHR))) = ntt::U4{r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]};
}
// inverse contiguous pass: store with 1/n and (optionally) the zk shift 3^bitrev_k(position)
RK_HD void store_rev_contig(const Args& a, const ntt::Tables& tb, const Tile& t, const uint32_t* lds, unsigned tid) {
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ntt::U4 v = *reinterpret_cast<const ntt::U4*>(lds + phys(4 * (tid + i * NTHR)));
        r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
    }
    if (a.zk) {
        // position = sp * 2^14 + e, e = [i:2][tid:10][c:2]  =>  bitrev_k(position) =
        //   bitrev2(c) << (k-2) | bitrev10(tid) << (k-12) | bitrev2(i) << (k-14) | bitrev_{k-14}(sp)
        unsigned up = a.k - TILE_LOG;
        uint32_t A = bb::mul(a.scale, ntt::pow3(tb, (bb::bitrev(tid, 10) << (up + 2)) | bb::bitrev(t.sp, up)));
        uint32_t C[3], D[3];
#pragma unroll
        for (int c = 1; c < 4; c++) C[c - 1] = ntt::pow3(tb, bb::bitrev((unsigned)c, 2) << (up + 12));
#pragma unroll
        for (int i = 1; i < 4; i++) D[i - 1] = ntt::pow3(tb, bb::bitrev((unsigned)i, 2) << up);
        apply_factors(r, A, C, D);
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) r[j] = bb::mul(r[j], a.scale);
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
        *reinterpret_cast<ntt::U4*>(a.dst + t.base + 4 * (tid + i * NTHR)) = ntt::U4{r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]};
}
// forward contiguous pass: load (every input word feeds 2^expand_bits consecutive tile slots)
RK_HD void load_fwd_contig(const Args& a, const Tile& t, uint32_t* lds, unsigned tid) {
    if (a.expand_bits == 2) {
        // one source word per 4-slot vector
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; i++) w[i] = a.src[t.src_base + tid + i * NTHR];
#pragma unroll
        for (int i = 0; i < 4; i++)
            *reinterpret_cast<ntt::U4*>(lds + phys(4 * (tid + i * NTHR))) = ntt::U4{w[i], w[i], w[i], w[i]};
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
            *reinterpret_cast<ntt::U4*>(lds + phys(4 * (tid + i * NTHR))) =
                *reinterpret_cast<const ntt::U4*>(a.src + t.src_base + 4 * (tid + i * NTHR));
    }
}
// forward contiguous pass: store with the four-step twiddle w_n^(+e * bitrev(sp))
RK_HD void store_fwd_contig(const Args& a, const ntt::Tables& tb, const Tile& t, const uint32_t* lds, unsigned tid) {
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ntt::U4 v = *reinterpret_cast<const ntt::U4*>(lds + phys(4 * (tid + i * NTHR)));
        r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
    }
    // r[] is lazy (< 2p, round_dit): bb::mul accepts that and returns canonical values
    if (a.g_outer && t.sp != 0) {
        uint32_t A, C[3], D[3];
        twiddle_parts(a, tb, 0, t.sp, tid, A, C, D);
        apply_factors(r, A, C, D);
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) r[j] = bb::ucanon(r[j]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
        *reinterpret_cast<ntt::U4*>(a.dst + t.base + 4 * (tid + i * NTHR)) = ntt::U4{r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]};
}

// ---- round schedule ---------------------------------------------------------------------
// inverse: stages 0..g-1 from the largest half; forward: stages t0..g-1 from the smallest.
// Rounds of 4 stages, the odd remainder (g or g - t0 mod 4) goes to the round at ls == 0 / t0.
struct Sched {
    unsigned n;
    unsigned ls[4], nst[4];
};
inline Sched sched_dif(unsigned g) {
    Sched s{};
    unsigned rem = g;
    while (rem >= 4) {
        rem -= 4;
        s.ls[s.n] = rem;
        s.nst[s.n++] = 4;
    }
    if (rem) {
        s.ls[s.n] = 0;
        s.nst[s.n++] = rem;
    }
    return s;
}
inline Sched sched_dit(unsigned g, unsigned t0) {
    Sched s{};
    unsigned cnt = g - t0, pos = t0;
    if (cnt % 4) {
        s.ls[s.n] = pos;
        s.nst[s.n++] = cnt % 4;
        pos += cnt % 4;
    }
    while (pos < g) {
        s.ls[s.n] = pos;
        s.nst[s.n++] = 4;
        pos += 4;
    }
    return s;
}
// can a transform of 2^k points (expand_bits on the forward side) use these kernels?
// plan: k <= 14 is handled by ntt_core; 15 <= k <= 22 -> strided g = k - 14 >= ... and contiguous 14
inline bool usable(unsigned k, unsigned expand_bits, bool aligned16) {
    if (!aligned16 || k < TILE_LOG + 4 || k > TILE_LOG + 8) return false;  // strided pass needs 4 <= g <= 8
    return expand_bits == 0 || expand_bits == 2;
}

}  // namespace r16
