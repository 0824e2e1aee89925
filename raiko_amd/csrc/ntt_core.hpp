// Tiled radix-2 NTT passes for BabyBear columns (gfx950: one workgroup stages a
// 2^g x T tile of one column in LDS, runs g butterfly stages there, and applies
// the four-step twiddle on the way in/out, so each pass reads and writes every
// element exactly once from HBM).
//
// Conventions follow the published risc0 NTT (risc0-zkp core/ntt.rs, behind the
// call at reference provers/risc0/driver/src/bonsai.rs:271):
//   interpolate: natural-order evaluations -> bit-reversed coefficients,
//                decimation in frequency with inverse roots, scaled by 1/n;
//   evaluate:    bit-reversed coefficients -> natural-order evaluations,
//                decimation in time with forward roots; the lowest expand_bits
//                stages of a zero-padded input are a broadcast.
//
// The phase functions are RK_HD so tests can run them lane by lane on the CPU
// (tests/emul/emul.cpp) -- the GPU kernel in kernels_ntt.hip is the same code
// with __syncthreads() between phases.
#pragma once
#include "bb.hpp"

namespace ntt {

constexpr unsigned KS = 14;       // log2 of the in-LDS twiddle table order
constexpr unsigned LAMBDA = 24;   // log2 of the largest supported transform
constexpr unsigned TW_SPLIT = 12; // two-level table split for order-2^LAMBDA roots
constexpr unsigned MAX_TILE_LOG = 14;  // 2^14 elements = 64 KiB of LDS

struct Tables {
    // [0] = forward roots, [1] = inverse roots
    // per-stage twiddles in heap order: small[d][half + j] = w_{2*half}^{+-j}, j < half, half = 1 .. 2^(KS-1);
    // consecutive butterflies of a stage read consecutive words (one coalesced line per wave)
    const uint32_t* small[2];
    const uint32_t* hi[2];     // W^{+-(a << TW_SPLIT)}, W = w_{2^LAMBDA}, a < 2^(LAMBDA-TW_SPLIT)
    const uint32_t* lo[2];     // W^{+-b}, b < 2^TW_SPLIT
    const uint32_t* pow3_hi;   // shift^(a << TW_SPLIT), shift = the coset shift (3 for risc0)
    const uint32_t* pow3_lo;   // shift^b
};

struct PassArgs {
    uint32_t* dst;
    const uint32_t* src;   // == dst except for the expanding first forward pass
    size_t n;              // elements per column in dst
    size_t n_src;          // elements per column in src
    unsigned mu;           // log2 sub-problem size
    unsigned g;            // stages in this pass
    unsigned logT;         // log2 tile width (adjacent sub-problem positions)
    unsigned expand_bits;  // forward first pass only
    uint32_t scale;        // reverse last pass: Montgomery form of 1/n, else 0 (= skip)
    unsigned zk_bits;      // reverse last pass fused zk-shift: log2 n, else 0 (= skip)
};

RK_HD uint32_t root_pow(const Tables& t, int dir, uint32_t e24) {
    // W^(+-e24), e24 < 2^LAMBDA
    uint32_t h = t.hi[dir][e24 >> TW_SPLIT];
    uint32_t l = t.lo[dir][e24 & ((1u << TW_SPLIT) - 1)];
    return bb::mul(h, l);
}
RK_HD uint32_t pow3(const Tables& t, uint32_t e) {
    return bb::mul(t.pow3_hi[e >> TW_SPLIT], t.pow3_lo[e & ((1u << TW_SPLIT) - 1)]);
}

struct Tile {
    size_t base;     // element offset of (hi=0, lo=0) inside dst
    size_t src_base; // element offset inside src (expand pass)
    uint32_t lo0;    // first sub-problem position covered by the tile
    size_t S;        // stride between consecutive hi
};
// blocks are numbered column-major: block = col * blocks_per_col + b
RK_HD Tile tile_of(const PassArgs& a, size_t block) {
    size_t tile_elems = (size_t)1 << (a.g + a.logT);
    size_t per_col = a.n >> (a.g + a.logT);
    size_t col = block / per_col, b = block % per_col;
    size_t S = ((size_t)1 << a.mu) >> a.g;
    size_t tiles_per_sub = S >> a.logT;
    size_t sp = b / tiles_per_sub, ti = b % tiles_per_sub;
    Tile t;
    t.S = S;
    t.lo0 = (uint32_t)(ti << a.logT);
    t.base = col * a.n + (sp << a.mu) + t.lo0;
    t.src_base = col * a.n_src + ((b * tile_elems) >> a.expand_bits);
    return t;
}

// ---------------------------------------------------------------- reverse (DIF)
RK_HD void rev_load(const PassArgs& a, const Tile& t, uint32_t* lds, unsigned tid, unsigned nthr) {
    unsigned elems = 1u << (a.g + a.logT), tmask = (1u << a.logT) - 1;
    for (unsigned e = tid; e < elems; e += nthr) {
        unsigned hi = e >> a.logT, lo = e & tmask;
        lds[e] = a.src[t.base + (size_t)hi * t.S + lo];
    }
}
RK_HD void rev_stage(const PassArgs& a, const Tables& tb, uint32_t* lds, unsigned tid, unsigned nthr, unsigned s) {
    unsigned nb = 1u << (a.g - 1 + a.logT), tmask = (1u << a.logT) - 1;
    unsigned hlog = a.g - 1 - s, half = 1u << hlog;
    for (unsigned b = tid; b < nb; b += nthr) {
        unsigned lo = b & tmask, hb = b >> a.logT;
        unsigned j = hb & (half - 1), blk = hb >> hlog;
        unsigned i0 = (((blk << (hlog + 1)) + j) << a.logT) + lo;
        unsigned i1 = i0 + (half << a.logT);
        uint32_t x = lds[i0], y = lds[i1];
        lds[i0] = bb::add(x, y);
        lds[i1] = bb::mul(bb::sub(x, y), tb.small[1][half + j]);
    }
}
RK_HD void rev_store(const PassArgs& a, const Tables& tb, const Tile& t, const uint32_t* lds, unsigned tid,
                     unsigned nthr) {
    unsigned elems = 1u << (a.g + a.logT), tmask = (1u << a.logT) - 1;
    bool tw = t.S > 1;
    for (unsigned e = tid; e < elems; e += nthr) {
        unsigned p1 = e >> a.logT, lo = e & tmask;
        uint32_t v = lds[e];
        if (tw) {
            uint32_t k1 = bb::bitrev(p1, a.g);
            uint32_t ex = ((t.lo0 + lo) * k1) << (LAMBDA - a.mu);
            v = bb::mul(v, root_pow(tb, 1, ex));
        }
        size_t pos = t.base + (size_t)p1 * t.S + lo;
        if (a.scale) {
            uint32_t f = a.scale;
            if (a.zk_bits) {
                uint32_t within = (uint32_t)(pos & (a.n - 1));
                f = bb::mul(f, pow3(tb, bb::bitrev(within, a.zk_bits)));
            }
            v = bb::mul(v, f);
        }
        a.dst[pos] = v;
    }
}

// ---------------------------------------------------------------- forward (DIT)
RK_HD void fwd_load(const PassArgs& a, const Tables& tb, const Tile& t, uint32_t* lds, unsigned tid, unsigned nthr) {
    unsigned elems = 1u << (a.g + a.logT), tmask = (1u << a.logT) - 1;
    bool tw = t.S > 1;
    for (unsigned e = tid; e < elems; e += nthr) {
        unsigned hi = e >> a.logT, lo = e & tmask;
        uint32_t v;
        if (a.expand_bits) {
            v = a.src[t.src_base + (e >> a.expand_bits)];  // T == 1, S == 1 here
        } else {
            v = a.src[t.base + (size_t)hi * t.S + lo];
            if (tw) {
                uint32_t r = bb::bitrev(hi, a.g);
                uint32_t ex = (r * (t.lo0 + lo)) << (LAMBDA - a.mu);
                v = bb::mul(v, root_pow(tb, 0, ex));
            }
        }
        lds[e] = v;
    }
}
RK_HD void fwd_stage(const PassArgs& a, const Tables& tb, uint32_t* lds, unsigned tid, unsigned nthr, unsigned tt) {
    unsigned nb = 1u << (a.g - 1 + a.logT), tmask = (1u << a.logT) - 1;
    unsigned half = 1u << tt;
    for (unsigned b = tid; b < nb; b += nthr) {
        unsigned lo = b & tmask, hb = b >> a.logT;
        unsigned j = hb & (half - 1), blk = hb >> tt;
        unsigned i0 = (((blk << (tt + 1)) + j) << a.logT) + lo;
        unsigned i1 = i0 + (half << a.logT);
        uint32_t x = lds[i0];
        uint32_t y = bb::mul(lds[i1], tb.small[0][half + j]);
        lds[i0] = bb::add(x, y);
        lds[i1] = bb::sub(x, y);
    }
}
RK_HD void fwd_store(const PassArgs& a, const Tile& t, const uint32_t* lds, unsigned tid, unsigned nthr) {
    unsigned elems = 1u << (a.g + a.logT), tmask = (1u << a.logT) - 1;
    for (unsigned e = tid; e < elems; e += nthr) {
        unsigned hi = e >> a.logT, lo = e & tmask;
        a.dst[t.base + (size_t)hi * t.S + lo] = lds[e];
    }
}

// ---------------------------------------------------------------- unrolled variants
// Same arithmetic as the loops above with the per-thread trip count fixed at compile time
// (EPT elements = EPT/2 butterflies per thread, tile == EPT * nthr) so every LDS / table /
// global access of a phase is issued before the first use: the generic loops expose one
// memory round trip per butterfly, these expose one per phase.  VEC: 16-byte global accesses
// (4 consecutive tile elements are consecutive in memory when T >= 4 or the pass is contiguous).
struct U4 {
    uint32_t x, y, z, w;
};

template <int EPT, bool VEC>
RK_HD void rev_load_t(const PassArgs& a, const Tile& t, uint32_t* lds, unsigned tid, unsigned nthr) {
    unsigned tmask = (1u << a.logT) - 1;
    if (VEC) {
        U4 v[EPT / 4];
#pragma unroll
        for (int i = 0; i < EPT / 4; i++) {
            unsigned e = 4 * (tid + i * nthr);
            unsigned hi = e >> a.logT, lo = e & tmask;
            v[i] = *reinterpret_cast<const U4*>(a.src + t.base + (size_t)hi * t.S + lo);
        }
#pragma unroll
        for (int i = 0; i < EPT / 4; i++) *reinterpret_cast<U4*>(lds + 4 * (tid + i * nthr)) = v[i];
    } else {
        uint32_t v[EPT];
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            unsigned e = tid + i * nthr;
            v[i] = a.src[t.base + (size_t)(e >> a.logT) * t.S + (e & tmask)];
        }
#pragma unroll
        for (int i = 0; i < EPT; i++) lds[tid + i * nthr] = v[i];
    }
}
template <int BPT>
RK_HD void rev_stage_t(const PassArgs& a, const Tables& tb, uint32_t* lds, unsigned tid, unsigned nthr, unsigned s) {
    unsigned tmask = (1u << a.logT) - 1;
    unsigned hlog = a.g - 1 - s, half = 1u << hlog;
    unsigned i0[BPT];
    uint32_t x[BPT], y[BPT], w[BPT];
#pragma unroll
    for (int i = 0; i < BPT; i++) {
        unsigned b = tid + i * nthr;
        unsigned lo = b & tmask, hb = b >> a.logT;
        unsigned j = hb & (half - 1), blk = hb >> hlog;
        i0[i] = (((blk << (hlog + 1)) + j) << a.logT) + lo;
        w[i] = tb.small[1][half + j];
        x[i] = lds[i0[i]];
        y[i] = lds[i0[i] + (half << a.logT)];
    }
#pragma unroll
    for (int i = 0; i < BPT; i++) {
        lds[i0[i]] = bb::add(x[i], y[i]);
        lds[i0[i] + (half << a.logT)] = bb::mul(bb::sub(x[i], y[i]), w[i]);
    }
}
RK_HD uint32_t rev_out_factor(const PassArgs& a, const Tables& tb, const Tile& t, unsigned p1, unsigned lo, size_t pos,
                              bool& has) {
    uint32_t f = bb::ONE;
    has = false;
    if (t.S > 1) {
        uint32_t k1 = bb::bitrev(p1, a.g);
        f = root_pow(tb, 1, ((t.lo0 + lo) * k1) << (LAMBDA - a.mu));
        has = true;
    }
    if (a.scale) {
        uint32_t g = a.scale;
        if (a.zk_bits) g = bb::mul(g, pow3(tb, bb::bitrev((uint32_t)(pos & (a.n - 1)), a.zk_bits)));
        f = has ? bb::mul(f, g) : g;
        has = true;
    }
    return f;
}
template <int EPT, bool VEC>
RK_HD void rev_store_t(const PassArgs& a, const Tables& tb, const Tile& t, const uint32_t* lds, unsigned tid,
                       unsigned nthr) {
    unsigned tmask = (1u << a.logT) - 1;
    if (VEC) {
#pragma unroll
        for (int i = 0; i < EPT / 4; i++) {
            unsigned e = 4 * (tid + i * nthr);
            unsigned p1 = e >> a.logT, lo = e & tmask;
            size_t pos = t.base + (size_t)p1 * t.S + lo;
            U4 v = *reinterpret_cast<const U4*>(lds + e);
            uint32_t r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                // T >= 4 keeps the 4 elements in one tile row; the contiguous pass (T == 1) has S == 1
                unsigned pc = a.logT >= 2 ? p1 : p1 + c, lc = a.logT >= 2 ? lo + c : lo;
                bool has;
                uint32_t f = rev_out_factor(a, tb, t, pc, lc, pos + c, has);
                if (has) r[c] = bb::mul(r[c], f);
            }
            *reinterpret_cast<U4*>(a.dst + pos) = U4{r[0], r[1], r[2], r[3]};
        }
    } else {
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            unsigned e = tid + i * nthr;
            unsigned p1 = e >> a.logT, lo = e & tmask;
            size_t pos = t.base + (size_t)p1 * t.S + lo;
            bool has;
            uint32_t f = rev_out_factor(a, tb, t, p1, lo, pos, has);
            uint32_t v = lds[e];
            a.dst[pos] = has ? bb::mul(v, f) : v;
        }
    }
}

template <int EPT, bool VEC>
RK_HD void fwd_load_t(const PassArgs& a, const Tables& tb, const Tile& t, uint32_t* lds, unsigned tid, unsigned nthr) {
    unsigned tmask = (1u << a.logT) - 1;
    bool tw = t.S > 1;
    if (a.expand_bits) {
        // contiguous first pass of an LDE: 2^expand_bits consecutive outputs share one input
        // (T == 1, S == 1).  One source word feeds a group of EX tile slots.
        unsigned ex = 1u << a.expand_bits;
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            unsigned e = tid + i * nthr;
            lds[e] = a.src[t.src_base + (e / ex)];
        }
        return;
    }
    if (VEC) {
#pragma unroll
        for (int i = 0; i < EPT / 4; i++) {
            unsigned e = 4 * (tid + i * nthr);
            unsigned hi = e >> a.logT, lo = e & tmask;
            U4 v = *reinterpret_cast<const U4*>(a.src + t.base + (size_t)hi * t.S + lo);
            uint32_t r[4] = {v.x, v.y, v.z, v.w};
            if (tw) {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    unsigned hc = a.logT >= 2 ? hi : hi + c, lc = a.logT >= 2 ? lo + c : lo;
                    uint32_t rr = bb::bitrev(hc, a.g);
                    r[c] = bb::mul(r[c], root_pow(tb, 0, (rr * (t.lo0 + lc)) << (LAMBDA - a.mu)));
                }
            }
            *reinterpret_cast<U4*>(lds + e) = U4{r[0], r[1], r[2], r[3]};
        }
    } else {
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            unsigned e = tid + i * nthr;
            unsigned hi = e >> a.logT, lo = e & tmask;
            uint32_t v = a.src[t.base + (size_t)hi * t.S + lo];
            if (tw) v = bb::mul(v, root_pow(tb, 0, (bb::bitrev(hi, a.g) * (t.lo0 + lo)) << (LAMBDA - a.mu)));
            lds[e] = v;
        }
    }
}
template <int BPT>
RK_HD void fwd_stage_t(const PassArgs& a, const Tables& tb, uint32_t* lds, unsigned tid, unsigned nthr, unsigned tt) {
    unsigned tmask = (1u << a.logT) - 1;
    unsigned half = 1u << tt;
    unsigned i0[BPT];
    uint32_t x[BPT], y[BPT], w[BPT];
#pragma unroll
    for (int i = 0; i < BPT; i++) {
        unsigned b = tid + i * nthr;
        unsigned lo = b & tmask, hb = b >> a.logT;
        unsigned j = hb & (half - 1), blk = hb >> tt;
        i0[i] = (((blk << (tt + 1)) + j) << a.logT) + lo;
        w[i] = tb.small[0][half + j];
        x[i] = lds[i0[i]];
        y[i] = lds[i0[i] + (half << a.logT)];
    }
#pragma unroll
    for (int i = 0; i < BPT; i++) {
        uint32_t yy = bb::mul(y[i], w[i]);
        lds[i0[i]] = bb::add(x[i], yy);
        lds[i0[i] + (half << a.logT)] = bb::sub(x[i], yy);
    }
}
template <int EPT, bool VEC>
RK_HD void fwd_store_t(const PassArgs& a, const Tile& t, const uint32_t* lds, unsigned tid, unsigned nthr) {
    unsigned tmask = (1u << a.logT) - 1;
    if (VEC) {
#pragma unroll
        for (int i = 0; i < EPT / 4; i++) {
            unsigned e = 4 * (tid + i * nthr);
            unsigned hi = e >> a.logT, lo = e & tmask;
            *reinterpret_cast<U4*>(a.dst + t.base + (size_t)hi * t.S + lo) = *reinterpret_cast<const U4*>(lds + e);
        }
    } else {
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            unsigned e = tid + i * nthr;
            a.dst[t.base + (size_t)(e >> a.logT) * t.S + (e & tmask)] = lds[e];
        }
    }
}
// may the unrolled/vector form be used for this pass?
inline bool can_unroll(const PassArgs& a, int ept) { return ((size_t)1 << (a.g + a.logT)) >= (size_t)ept * 64; }
inline bool can_vec(const PassArgs& a) {
    size_t S = ((size_t)1 << a.mu) >> a.g;
    bool contiguous4 = a.logT >= 2 || (a.logT == 0 && S == 1);
    return contiguous4 && (a.n % 4 == 0) && (((uintptr_t)a.dst | (uintptr_t)a.src) & 15) == 0;
}

// ---------------------------------------------------------------- pass planning
struct Plan {
    unsigned npass;
    unsigned g[4], logT[4], mu[4];
};
// split k stages into passes, outermost (largest sub-problem) first; the host
// runs them in this order for the reverse transform and in the opposite order
// for the forward one.
inline Plan make_plan(unsigned k, unsigned max_tile_log = MAX_TILE_LOG) {
    Plan p{};
    if (k == 0) return p;
    unsigned rem = k, mu = k;
    // contiguous innermost pass takes as many stages as fit in one tile
    unsigned inner = k < max_tile_log ? k : max_tile_log;
    rem -= inner;
    // strided passes of at most 8 stages each (tile width >= 64 elements)
    unsigned outer[3], no = 0;
    while (rem > 0) {
        unsigned npieces = (rem + 7) / 8;
        unsigned gg = (rem + npieces - 1) / npieces;
        outer[no++] = gg;
        rem -= gg;
    }
    for (unsigned i = 0; i < no; i++) {
        p.g[p.npass] = outer[i];
        p.mu[p.npass] = mu;
        unsigned s_log = mu - outer[i];
        unsigned t = max_tile_log - outer[i];
        p.logT[p.npass] = t < s_log ? t : s_log;
        p.npass++;
        mu -= outer[i];
    }
    p.g[p.npass] = inner;
    p.mu[p.npass] = mu;
    p.logT[p.npass] = 0;
    p.npass++;
    return p;
}

// Host-side generation of every table behind `Tables` (one flat array + offsets).
struct TableLayout {
    size_t small[2], hi[2], lo[2], pow3_hi, pow3_lo, total;
};
inline TableLayout table_layout() {
    const size_t n_small = (size_t)1 << KS;
    const size_t n_hi = (size_t)1 << (LAMBDA - TW_SPLIT);
    const size_t n_lo = (size_t)1 << TW_SPLIT;
    TableLayout l{};
    size_t off = 0;
    for (int d = 0; d < 2; d++) { l.small[d] = off; off += n_small; }
    for (int d = 0; d < 2; d++) { l.hi[d] = off; off += n_hi; l.lo[d] = off; off += n_lo; }
    l.pow3_hi = off; off += n_hi;
    l.pow3_lo = off; off += n_lo;
    l.total = off;
    return l;
}
inline void fill_pow(uint32_t* v, size_t n, uint32_t base) {
    uint32_t cur = bb::ONE;
    for (size_t i = 0; i < n; i++) {
        v[i] = cur;
        cur = bb::mul(cur, base);
    }
}
// h must hold table_layout().total words.  root27 = a generator of the 2^27 subgroup, shift = the
// coset shift of the zk / LDE domain, both Montgomery form (risc0: 137 and 3; rk_params)
inline void fill_tables(uint32_t* h, uint32_t root27 = 0, uint32_t shift = 0) {
    if (root27 == 0) root27 = bb::encode(137);  // 137 generates the 2^27 subgroup of BabyBear
    if (shift == 0) shift = bb::encode(3);
    const TableLayout l = table_layout();
    const size_t n_hi = (size_t)1 << (LAMBDA - TW_SPLIT);
    const size_t n_lo = (size_t)1 << TW_SPLIT;
    uint32_t W = bb::pow(root27, (uint64_t)1 << (27 - LAMBDA));
    uint32_t Winv = bb::inv(W);
    for (int d = 0; d < 2; d++) {
        h[l.small[d]] = bb::ONE;  // slot 0 unused
        for (unsigned lg = 0; lg < KS; lg++) {
            size_t half = (size_t)1 << lg;
            uint32_t w = bb::pow(d == 0 ? W : Winv, (uint64_t)1 << (LAMBDA - lg - 1));  // order 2*half
            fill_pow(h + l.small[d] + half, half, w);
        }
    }
    for (int d = 0; d < 2; d++) {
        uint32_t g = d == 0 ? W : Winv;
        fill_pow(h + l.hi[d], n_hi, bb::pow(g, (uint64_t)1 << TW_SPLIT));
        fill_pow(h + l.lo[d], n_lo, g);
    }
    fill_pow(h + l.pow3_hi, n_hi, bb::pow(shift, (uint64_t)1 << TW_SPLIT));
    fill_pow(h + l.pow3_lo, n_lo, shift);
}
inline Tables tables_at(const uint32_t* base) {
    const TableLayout l = table_layout();
    Tables t{};
    for (int d = 0; d < 2; d++) {
        t.small[d] = base + l.small[d];
        t.hi[d] = base + l.hi[d];
        t.lo[d] = base + l.lo[d];
    }
    t.pow3_hi = base + l.pow3_hi;
    t.pow3_lo = base + l.pow3_lo;
    return t;
}

}  // namespace ntt
