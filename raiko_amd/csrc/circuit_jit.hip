// Run-time code generation for a constraint list (rk_program_compile): the list is turned into
// straight-line HIP -- the form risc0's build gives its CUDA / Metal eval_check kernels and
// tools/circuit_gen.py gives a circuit at build time -- and compiled for gfx950 with hiprtc, so a host
// that hands the list over at run time (the Rust crate does, provers/hip/driver/src/lib.rs) gets the
// generated kernel's speed (3x the interpreter's) without a build step.  One code object per device,
// kept inside the rk_program; rk_program_eval_check uses it when it is there.
//
// The generator reads the validated step list itself (its own liveness pass, independent of the
// interpreter's compiler -- the two are checked against each other and against the oracle):
//   * dead steps dropped, every mix state's `mul` folded to a power of poly_mix (table per proof);
//   * constants as Montgomery literals, taps as (group, column, back) immediates, arguments from the table;
//   * statements cut into __noinline__ functions of CHUNK statements (one basic block of 10^4 statements
//     costs the compiler tens of minutes), values that cross a cut travel through per-lane carry arrays,
//     leaves are re-read where they are used.
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cstring>
#include <set>
#include <sstream>

#include "circuit_program.hpp"

namespace {

using bb::Ext;
constexpr size_t CHUNK = 128;
constexpr uint32_t NONE = rk::PROGRAM_NONE;

// SHA-256 (FIPS 180-4) for the code-object cache: the files are executable content, a 64-bit FNV name is not an identity
struct Sha256 {
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    uint8_t buf[64];
    size_t fill = 0;
    uint64_t total = 0;
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
            0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
            0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
            0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
            0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
            0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const void* data, size_t n) {
        const uint8_t* p = (const uint8_t*)data;
        total += n;
        while (n) {
            const size_t k = std::min(n, 64 - fill);
            std::memcpy(buf + fill, p, k);
            fill += k; p += k; n -= k;
            if (fill == 64) {
                block(buf);
                fill = 0;
            }
        }
    }
    void finish(uint8_t out[32]) {
        const uint64_t bits = total * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t len[8];
        for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(len, 8);
        for (int i = 0; i < 8; i++)
            for (int j = 0; j < 4; j++) out[4 * i + j] = (uint8_t)(h[i] >> (24 - 8 * j));
    }
};

// the cache directory must be this user's and writable by nobody else (mode & 022 == 0)
bool cache_dir_is_private(const char* dir) {
    struct stat st;
    if (::stat(dir, &st) != 0 || !S_ISDIR(st.st_mode)) return false;
    return st.st_uid == ::geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
// file = "RKJIT2\0\0" | key digest 32 | body digest 32 | body length u64 LE | body
constexpr char CACHE_MAGIC[8] = {'R', 'K', 'J', 'I', 'T', '2', 0, 0};
std::vector<char> cache_read(const std::string& path, const uint8_t key[32]) {
    std::vector<char> code;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return code;
    uint8_t head[8 + 32 + 32 + 8];
    if (std::fread(head, 1, sizeof head, f) == sizeof head && std::memcmp(head, CACHE_MAGIC, 8) == 0 && std::memcmp(head + 8, key, 32) == 0) {
        uint64_t n = 0;
        for (int i = 0; i < 8; i++) n |= (uint64_t)head[72 + i] << (8 * i);
        if (n > 0 && n < ((uint64_t)1 << 31)) {
            code.resize((size_t)n);
            uint8_t d[32];
            Sha256 h;
            if (std::fread(code.data(), 1, (size_t)n, f) == (size_t)n && std::fgetc(f) == EOF) {
                h.update(code.data(), code.size());
                h.finish(d);
                if (std::memcmp(d, head + 40, 32) != 0) code.clear();
            } else {
                code.clear();
            }
        }
    }
    std::fclose(f);
    return code;
}
void cache_write(const std::string& path, const uint8_t key[32], const std::vector<char>& code) {
    uint8_t head[8 + 32 + 32 + 8];
    std::memcpy(head, CACHE_MAGIC, 8);
    std::memcpy(head + 8, key, 32);
    Sha256 h;
    h.update(code.data(), code.size());
    h.finish(head + 40);
    for (int i = 0; i < 8; i++) head[72 + i] = (uint8_t)((uint64_t)code.size() >> (8 * i));
    // written under a temporary name, renamed when complete: a reader never sees half a file
    const std::string tmp = path + ".tmp" + std::to_string((unsigned long long)::getpid()) + "_" + std::to_string((unsigned long long)(uintptr_t)&head);
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return;
    const bool ok = std::fwrite(head, 1, sizeof head, f) == sizeof head && std::fwrite(code.data(), 1, code.size(), f) == code.size();
    std::fclose(f);
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) (void)std::remove(tmp.c_str());
}

// what the generated kernel receives (same layout as `struct Args` in the source below)
struct JitArgs {
    const uint32_t* lde[3];
    uint64_t len[3];     // words per column of each group
    uint64_t tab;        // globals | accum mix | powers (4 words each)
    uint32_t* check;
    uint64_t d;
    uint32_t glob_base, mix_base, pw_base, wm;
    uint32_t blow, split;   // log2 (domain / trace rows); the result leaves in 2^split chunks (rk::EvalDomain)
    uint32_t str[3], one;   // point i of the domain is element i << str[g] of a column of group g; one = 1 (see below)
    uint32_t inv_base, pad; // the 1 / (x^n - 1) values sit in the table too: indexing an ARRAY inside the kernel arguments by
                            // the lane's point makes the compiler copy the whole argument block to scratch and read
                            // every argument back from there with vector loads
};

// Shape of the generated kernel (measured on the ISA of an 8 k-op list, tools/jit_census.sh):
//   * ONE function.  The statements are cut into blocks of CHUNK, each inside `if (flag) { ... }` with `flag` = 1
//     laundered through an empty asm, so every block is its own basic block (instruction selection and scheduling
//     work per basic block: one block of 10^4 statements costs the compiler tens of minutes) yet nothing is a call:
//     the kernel arguments stay in SGPRs (column bases are scalar arithmetic, a tap is one `global_load v, voff, s[base]`),
//     the power table is read through the scalar cache, values that cross a block boundary stay in registers.
//     The earlier form -- one __noinline__ function per block -- received its arguments through a pointer in VGPRs:
//     every argument read was a flat load, every tap address a 64-bit VALU multiply-add, every carried value a
//     scratch access: 10.9 VALU instructions per list op against 5 here.
//   * Mix states are kept unreduced.  tot = sum_k mont(pw_k, c_k) = (sum_k pw_k c_k) R^-1, so a state is four 64-bit
//     sums of plain products (one v_mad_u64_u32 per component and constraint), folded back below 2^60 by
//     hi * (2^32 mod p) + lo (one more v_mad_u64_u32) whenever the next product could overflow -- the bound of every
//     sum is tracked here, at generation time -- and Montgomery-reduced once, where its value is needed (an AND_COND's
//     inner block, the result).  32 instructions per constraint became 5 to 6; the field values are the same.
const char* PRELUDE = R"SRC(
#define RK_FI __device__ inline __attribute__((always_inline))
typedef unsigned int u32;
typedef unsigned long long u64;
typedef const __attribute__((address_space(4))) u32* const_u32;
namespace bb {
constexpr u32 P = 2013265921u;
constexpr u32 MPRIME = 0x88000001u;
constexpr u32 ONE = 268435454u;   // 2^32 mod p
RK_FI u32 add(u32 a, u32 b) { u32 r = a + b, s = r - P; return s < r ? s : r; }
RK_FI u32 sub(u32 a, u32 b) { u32 r = a - b, s = r + P; return s < r ? s : r; }
RK_FI u32 redc(u64 t) {   // t < 2^32 p  ->  t 2^-32 mod p, canonical
    u32 q = (u32)t * (0u - MPRIME);
    u64 w = t + (u64)q * P;
    u32 r = (u32)(w >> 32), s = r - P;
    return s < r ? s : r;
}
RK_FI u32 mul(u32 a, u32 b) { return redc((u64)a * b); }
struct Ext { u32 c[4]; };
struct Acc { u64 c[4]; };   // an unreduced mix state: component e = sum of products, congruent to tot_e 2^32
RK_FI Ext ext_zero() { return Ext{{0, 0, 0, 0}}; }
RK_FI Acc acc_zero() { return Acc{{0, 0, 0, 0}}; }
RK_FI Ext scale(const Ext& a, u32 s) { return Ext{{mul(a.c[0], s), mul(a.c[1], s), mul(a.c[2], s), mul(a.c[3], s)}}; }
RK_FI u64 fold(u64 t) { return (u64)(u32)(t >> 32) * ONE + (u32)t; }   // same residue, below 2^60.1
RK_FI Acc fold(const Acc& a) { return Acc{{fold(a.c[0]), fold(a.c[1]), fold(a.c[2]), fold(a.c[3])}}; }
RK_FI Ext fin(const Acc& a) { return Ext{{redc(a.c[0]), redc(a.c[1]), redc(a.c[2]), redc(a.c[3])}}; }   // components < 2^32 p
// x + pw * v, component-wise plain products
RK_FI Acc eqz(const Acc& x, const Ext& pw, u32 v) {
    return Acc{{x.c[0] + (u64)pw.c[0] * v, x.c[1] + (u64)pw.c[1] * v, x.c[2] + (u64)pw.c[2] * v, x.c[3] + (u64)pw.c[3] * v}};
}
}  // namespace bb
using bb::Ext;
using bb::Acc;
struct Args {
    const u32* lde[3];
    u64 len[3];
    u64 tab;
    u32* check;
    u64 d;
    u32 glob_base, mix_base, pw_base, wm;
    u32 blow, split;
    u32 str[3], one;
    u32 inv_base, pad;
};
RK_FI Ext load_pw(const_u32 tab, u32 base, u32 j) {
    const_u32 p = tab + base + 4 * j;
    return Ext{{p[0], p[1], p[2], p[3]}};
}
RK_FI u32 ld(const char* base, u32 byte_off) { return *(const u32*)(base + byte_off); }
)SRC";

struct Fp {
    uint32_t op, a, b;
    bool live = false;
};
struct Mx {
    uint32_t op, x, v, inner;
    uint64_t k = 0;
    bool zero = false, live = false;
};

// bounds of the unreduced sums, in units the 64-bit accumulators hold exactly
typedef unsigned __int128 u128;
constexpr u128 LIMIT = ((u128)1 << 64) - 1;
constexpr u128 PROD = (u128)(bb::P - 1) * (bb::P - 1);                               // one product of canonical words
constexpr u128 FOLDED = (u128)0xffffffffu * bb::ONE + 0xffffffffu;                   // after bb::fold
constexpr u128 REDC_OK = ((u128)bb::P << 32) - 1;                                    // bb::redc's precondition

// the source of the kernel; *powers: the exponents its table holds
std::string generate(const rk_program& pg, std::vector<uint32_t>* powers_out) {
    const auto& steps = pg.steps;
    std::vector<Fp> fp;
    std::vector<Mx> mx;
    std::vector<std::pair<bool, uint32_t>> where;
    where.reserve(steps.size());
    for (const rk_poly_step& st : steps) {  // validated by rk_program_create
        if (st.op <= RK_STEP_MUL) {
            where.push_back({false, (uint32_t)fp.size()});
            fp.push_back(Fp{st.op, st.a, st.b});
        } else {
            Mx m{st.op, st.a, st.b, st.c};
            if (st.op == RK_STEP_TRUE) {
                m.zero = true;
            } else if (st.op == RK_STEP_AND_COND) {
                m.k = mx[st.a].k + mx[st.c].k;
                m.zero = mx[st.a].zero && mx[st.c].zero;
            } else {
                m.k = mx[st.a].k + 1;
            }
            where.push_back({true, (uint32_t)mx.size()});
            mx.push_back(m);
        }
    }
    auto is_leaf = [&](uint32_t i) { return fp[i].op == RK_STEP_CONST || fp[i].op == RK_STEP_GET || fp[i].op == RK_STEP_GET_GLOBAL; };
    mx[pg.ret].live = true;
    for (size_t s = where.size(); s-- > 0;) {
        if (where[s].first) {
            const Mx& m = mx[where[s].second];
            if (!m.live || m.op == RK_STEP_TRUE) continue;
            mx[m.x].live = true;
            if (m.op == RK_STEP_AND_EQZ) {
                fp[m.v].live = true;
            } else if (!mx[m.inner].zero) {
                mx[m.inner].live = true;
                fp[m.v].live = true;
            }
        } else {
            const Fp& v = fp[where[s].second];
            if (v.live && !is_leaf(where[s].second)) fp[v.a].live = fp[v.b].live = true;
        }
    }
    std::set<uint64_t> pw_set;
    for (const Mx& m : mx)
        if (m.live && m.op != RK_STEP_TRUE && !m.zero) pw_set.insert(mx[m.x].k);
    std::vector<uint32_t> powers(pw_set.begin(), pw_set.end());
    auto pw_idx = [&](uint64_t k) { return (uint32_t)(std::lower_bound(powers.begin(), powers.end(), (uint32_t)k) - powers.begin()); };
    *powers_out = powers;

    // statements in list order; an AND_COND over an identically-zero block is its x under another name
    std::vector<uint32_t> alias(mx.size(), NONE);
    auto name_mx = [&](uint32_t i) {
        while (alias[i] != NONE) i = alias[i];
        return i;
    };
    struct Item {
        bool is_mix;
        uint32_t idx;
    };
    std::vector<Item> items;
    for (const auto& w : where) {
        if (!w.first) {
            if (fp[w.second].live && !is_leaf(w.second)) items.push_back(Item{false, w.second});
        } else {
            const Mx& m = mx[w.second];
            if (!m.live || m.op == RK_STEP_TRUE || m.zero) continue;
            if (m.op == RK_STEP_AND_COND && mx[m.inner].zero) {
                alias[w.second] = m.x;
                continue;
            }
            items.push_back(Item{true, w.second});
        }
    }
    const size_t n_chunks = std::max<size_t>(1, (items.size() + CHUNK - 1) / CHUNK);
    // the last block that reads every statement's result
    std::vector<uint32_t> fp_last(fp.size(), 0), mx_last(mx.size(), 0);
    auto note = [&](bool is_mix, uint32_t j, uint32_t c) {
        uint32_t& last = is_mix ? mx_last[j] : fp_last[j];
        last = std::max(last, c);
    };
    for (size_t t = 0; t < items.size(); t++) {
        const uint32_t c = (uint32_t)(t / CHUNK);
        const Item& it = items[t];
        if (it.is_mix) {
            const Mx& m = mx[it.idx];
            if (!is_leaf(m.v)) note(false, m.v, c);
            if (!mx[m.x].zero) note(true, name_mx(m.x), c);
            if (m.op == RK_STEP_AND_COND) note(true, name_mx(m.inner), c);
        } else {
            const Fp& v = fp[it.idx];
            if (!is_leaf(v.a)) note(false, v.a, c);
            if (!is_leaf(v.b)) note(false, v.b, c);
        }
    }
    const bool ret_zero = mx[pg.ret].zero;
    const uint32_t ret_name = ret_zero ? NONE : name_mx(pg.ret);
    if (!ret_zero) note(true, ret_name, (uint32_t)n_chunks);  // read by the kernel after the last block

    // which (back, group) byte offsets the list needs: computed once, ahead of the blocks
    std::set<std::pair<uint32_t, uint32_t>> offs;
    for (size_t j = 0; j < fp.size(); j++)
        if (fp[j].live && fp[j].op == RK_STEP_GET) offs.insert({pg.taps[fp[j].a].back, pg.taps[fp[j].a].group});
    auto off_name = [](uint32_t back, uint32_t group) {
        return "o" + (back == 0xffffffffu ? std::string("n") : std::to_string(back)) + "_" + std::to_string(group);
    };

    std::vector<uint32_t> fp_slot(fp.size(), NONE), mx_slot(mx.size(), NONE);
    std::vector<u128> mx_bound(mx.size(), 0);   // of the value named x<idx> (or its carried copy)
    std::vector<uint32_t> free_fp, free_mx;
    uint32_t next_fp = 0, next_mx = 0;
    std::vector<std::vector<std::pair<bool, uint32_t>>> release(n_chunks + 1);
    std::ostringstream blocks;
    for (size_t c = 0; c < n_chunks; c++) {
        std::ostringstream body;
        std::set<uint32_t> have_fp, have_mx;
        uint32_t tmp = 0;
        auto use_fp = [&](uint32_t j) -> std::string {
            std::string nm = "f" + std::to_string(j);
            if (have_fp.count(j)) return nm;
            have_fp.insert(j);
            const Fp& v = fp[j];
            if (v.op == RK_STEP_CONST) {
                body << "        const u32 " << nm << " = " << bb::encode(v.a) << "u;\n";
            } else if (v.op == RK_STEP_GET) {
                const rk::Tap& t = pg.taps[v.a];
                // the column's base through the block's opaque scalar sf (= 1): scalar arithmetic that no pass can share
                // between blocks -- shared, the hundreds of 64-bit bases of a long list outgrow the scalar registers
                // and the compiler moves all of them into VGPRs (256 VGPRs, scratch spills)
                body << "        const u32 " << nm << " = ld(gb" << t.group << " + (u64)(" << t.offset << "u * sf) * lb" << t.group << ", "
                     << off_name(t.back, t.group) << ");\n";
            } else if (v.op == RK_STEP_GET_GLOBAL) {
                body << "        const u32 " << nm << " = tab[a." << (v.a == 0 ? "glob_base" : "mix_base") << " + " << v.b << "u];\n";
            } else {
                body << "        const u32 " << nm << " = c" << fp_slot[j] << ";\n";
            }
            return nm;
        };
        // the name of mix state j as an Acc of this block (its carried copy when it was made in an earlier one)
        auto use_mx = [&](uint32_t j) -> std::string {
            std::string nm = "x" + std::to_string(j);
            if (have_mx.count(j)) return nm;
            have_mx.insert(j);
            const uint32_t k = mx_slot[j];
            body << "        const Acc " << nm << " = Acc{{m" << k << "_0, m" << k << "_1, m" << k << "_2, m" << k << "_3}};\n";
            return nm;
        };
        // an Acc expression of value `nm` (bound *b) that leaves room for `extra` more: folded first when it must be
        auto room = [&](std::string nm, u128* b, u128 extra) -> std::string {
            if (*b + extra <= LIMIT) return nm;
            const std::string t = "t" + std::to_string(tmp++);
            body << "        const Acc " << t << " = bb::fold(" << nm << ");\n";
            *b = FOLDED;
            return t;
        };
        for (size_t t = c * CHUNK; t < std::min(items.size(), (c + 1) * CHUNK); t++) {
            const Item& it = items[t];
            if (!it.is_mix) {
                const Fp& v = fp[it.idx];
                const char* fn = v.op == RK_STEP_ADD ? "add" : v.op == RK_STEP_SUB ? "sub" : "mul";
                std::string a_ = use_fp(v.a), b_ = use_fp(v.b);
                body << "        const u32 f" << it.idx << " = bb::" << fn << "(" << a_ << ", " << b_ << ");\n";
                have_fp.insert(it.idx);
                if (fp_last[it.idx] > c) {
                    uint32_t k;
                    if (!free_fp.empty()) {
                        k = free_fp.back();
                        free_fp.pop_back();
                    } else {
                        k = next_fp++;
                    }
                    fp_slot[it.idx] = k;
                    release[fp_last[it.idx]].push_back({false, it.idx});
                    body << "        c" << k << " = f" << it.idx << ";\n";
                }
            } else {
                const Mx& m = mx[it.idx];
                const std::string me = "x" + std::to_string(it.idx);
                const std::string pw = "load_pw(tab, a.pw_base, " + std::to_string(pw_idx(mx[m.x].k)) + ")";
                // the state this one extends (absent when it is identically zero)
                std::string xs = "bb::acc_zero()";
                u128 bound = 0;
                if (!mx[m.x].zero) {
                    const uint32_t xn = name_mx(m.x);
                    xs = use_mx(xn);
                    bound = mx_bound[xn];
                }
                const std::string vs = use_fp(m.v);   // may emit the leaf's own statement: before anything of this one
                if (m.op == RK_STEP_AND_EQZ) {
                    xs = room(xs, &bound, PROD);
                    body << "        const Acc " << me << " = bb::eqz(" << xs << ", " << pw << ", " << vs << ");\n";
                    bound += PROD;
                } else {
                    // x.tot + cond * inner.tot * x.mul: u = pw * cond (canonical), T = the inner block's total (canonical),
                    // added as the plain ext product u * T -- its x^4 wrap terms go through one Montgomery product so
                    // that they, too, enter as plain products wm * h
                    const uint32_t in = name_mx(m.inner);
                    u128 ib = mx_bound[in];
                    std::string is = use_mx(in);
                    if (ib > REDC_OK) {
                        const std::string t2 = "t" + std::to_string(tmp++);
                        body << "        const Acc " << t2 << " = bb::fold(" << is << ");\n";
                        is = t2;
                    }
                    // (names that cannot collide with a type: "u32" / "u64" would)
                    const std::string u = "cu_" + std::to_string(it.idx), T = "cT_" + std::to_string(it.idx), h = "ch_" + std::to_string(it.idx) + "_";
                    body << "        const Ext " << u << " = bb::scale(" << pw << ", " << vs << ");\n";
                    body << "        const Ext " << T << " = bb::fin(" << is << ");\n";
                    body << "        const u32 " << h << "0 = bb::add(bb::add(bb::mul(" << u << ".c[1], " << T << ".c[3]), bb::mul(" << u << ".c[2], " << T
                         << ".c[2])), bb::mul(" << u << ".c[3], " << T << ".c[1]));\n";
                    body << "        const u32 " << h << "1 = bb::add(bb::mul(" << u << ".c[2], " << T << ".c[3]), bb::mul(" << u << ".c[3], " << T << ".c[2]));\n";
                    body << "        const u32 " << h << "2 = bb::mul(" << u << ".c[3], " << T << ".c[3]);\n";
                    // products per component: 2, 3, 4, 4 -- added in two steps so that a fold in between always suffices
                    xs = room(xs, &bound, 2 * PROD);
                    const std::string s1 = "t" + std::to_string(tmp++);
                    body << "        const Acc " << s1 << " = Acc{{" << xs << ".c[0] + (u64)" << u << ".c[0] * " << T << ".c[0] + (u64)a.wm * " << h << "0, "
                         << xs << ".c[1] + (u64)" << u << ".c[0] * " << T << ".c[1] + (u64)" << u << ".c[1] * " << T << ".c[0], "
                         << xs << ".c[2] + (u64)" << u << ".c[0] * " << T << ".c[2] + (u64)" << u << ".c[1] * " << T << ".c[1], "
                         << xs << ".c[3] + (u64)" << u << ".c[0] * " << T << ".c[3] + (u64)" << u << ".c[1] * " << T << ".c[2]}};\n";
                    bound += 2 * PROD;
                    std::string s1n = room(s1, &bound, 2 * PROD);
                    body << "        const Acc " << me << " = Acc{{" << s1n << ".c[0], " << s1n << ".c[1] + (u64)a.wm * " << h << "1, "
                         << s1n << ".c[2] + (u64)" << u << ".c[2] * " << T << ".c[0] + (u64)a.wm * " << h << "2, "
                         << s1n << ".c[3] + (u64)" << u << ".c[2] * " << T << ".c[1] + (u64)" << u << ".c[3] * " << T << ".c[0]}};\n";
                    bound += 2 * PROD;
                }
                mx_bound[it.idx] = bound;
                have_mx.insert(it.idx);
                if (mx_last[it.idx] > c) {
                    uint32_t k;
                    if (!free_mx.empty()) {
                        k = free_mx.back();
                        free_mx.pop_back();
                    } else {
                        k = next_mx++;
                    }
                    mx_slot[it.idx] = k;
                    release[mx_last[it.idx]].push_back({true, it.idx});
                    for (int e = 0; e < 4; e++) body << "        m" << k << "_" << e << " = " << me << ".c[" << e << "];\n";
                }
            }
        }
        for (const auto& r : release[c]) (r.first ? free_mx : free_fp).push_back(r.first ? mx_slot[r.second] : fp_slot[r.second]);
        // a fresh flag per block: a.one = 1 made opaque in a VGPR (an "s" constraint fails with `illegal VGPR to SGPR
        // copy` once scalar registers run short in a long kernel), read back as a scalar so that the branch is uniform
        blocks << "    {\n    u32 flag = a.one;\n    asm volatile(\"\" : \"+v\"(flag));\n    const u32 sf = __builtin_amdgcn_readfirstlane(flag);\n"
               << "    if (sf) {\n" << body.str() << "    }\n    }\n";
    }
    std::ostringstream src;
    src << PRELUDE;
    src << "extern \"C\" __global__ __attribute__((amdgpu_flat_work_group_size(1, 256))) void rk_jit_eval_check(Args a) {\n"
        << "    const u32 i = blockIdx.x * 256u + threadIdx.x, d = (u32)a.d;\n"
        << "    if (i >= d) return;\n"
        << "    const const_u32 tab = (const_u32)a.tab;\n";
    for (const auto& o : offs) {
        src << "    const u32 " << off_name(o.first, o.second) << " = ((";
        if (o.first == 0) src << "i";
        else src << "((i + d - (" << o.first << "u << a.blow)) & (d - 1u))";   // back = 2^32 - 1: one row ahead, modulo d
        src << ") << a.str[" << o.second << "]) << 2;\n";
    }
    for (uint32_t g = 0; g < 3; g++)
        src << "    const char* const gb" << g << " = (const char*)a.lde[" << g << "];\n    const u64 lb" << g << " = a.len[" << g << "] * 4;\n";
    for (uint32_t k = 0; k < next_fp; k++) src << "    u32 c" << k << " = 0;\n";
    for (uint32_t k = 0; k < next_mx; k++)
        for (int e = 0; e < 4; e++) src << "    u64 m" << k << "_" << e << " = 0;\n";
    src << blocks.str();
    if (ret_zero) {
        src << "    const Ext tot = bb::ext_zero();\n";
    } else {
        const uint32_t k = mx_slot[ret_name];
        src << "    const Acc r0 = Acc{{m" << k << "_0, m" << k << "_1, m" << k << "_2, m" << k << "_3}};\n"
            << "    const Ext tot = bb::scale(bb::fin(" << (mx_bound[ret_name] > REDC_OK ? "bb::fold(r0)" : "r0") << "), tab[a.inv_base + (i & ((1u << a.blow) - 1u))]);\n";
    }
    src << "    const u64 rows = a.d >> a.split, at = (u64)(i & ((1u << a.split) - 1)) * 4 * rows + (i >> a.split);\n"
        << "    for (int e = 0; e < 4; e++) a.check[at + (u64)e * rows] = tot.c[e];\n"
        << "}\n";
    return src.str();
}

}  // namespace

namespace rk {

// the generated kernel for the context's device, or nullptr when rk_program_compile has not run for it
const JitEntry* program_jit(rk_program* pg, int device) {
    std::lock_guard<std::mutex> lk(pg->mu);
    auto it = pg->jit.find(device);
    return it == pg->jit.end() ? nullptr : &it->second;
}

int program_jit_launch(rk_ctx* ctx, const JitEntry& je, const EvalDomain& v, const uint32_t* d_tab, uint32_t glob_base,
                       uint32_t mix_base, uint32_t pw_base, uint32_t* d_check, uint32_t inv_base) {
    const unsigned blow = v.ratio_log2;
    JitArgs a{};
    for (int g = 0; g < 3; g++) {
        a.lde[g] = v.d_cols[g];
        a.len[g] = v.col_len[g];
        a.str[g] = v.stride_log2[g];
    }
    a.split = v.split_log2;
    a.one = 1;
    a.tab = (uint64_t)(uintptr_t)d_tab;
    a.check = d_check;
    a.d = (uint64_t)1 << (v.po2 + blow);
    a.glob_base = glob_base;
    a.mix_base = mix_base;
    a.pw_base = pw_base;
    a.wm = ctx->sys.wm;
    a.blow = blow;
    a.inv_base = inv_base;
    size_t sz = sizeof a;
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    RK_HIP_TRY(ctx, hipModuleLaunchKernel(je.kernel, (unsigned)((a.d + 255) / 256), 1, 1, 256, 1, 1, 0, ctx->stream, nullptr, cfg));
    return RK_OK;
}

}  // namespace rk

extern "C" {

int rk_program_compile(rk_program* pg, rk_ctx* ctx) {
    RK_GUARD_BEGIN
    if (!pg || !ctx) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        std::lock_guard<std::mutex> lk(pg->mu);
        if (pg->jit.count(ctx->device)) return RK_OK;
    }
    rk::JitEntry je;
    const std::string src = generate(*pg, &je.powers);
    je.n_powers = (uint32_t)je.powers.size();
    hipDeviceProp_t props;
    RK_HIP_TRY(ctx, hipGetDeviceProperties(&props, ctx->device));
    // RK_JIT_CACHE_DIR (opt-in): a circuit of 30 k steps costs ~30 s to compile, a host restart should not pay it again.
    // A cache file is code that will run on the GPU, so it is only taken from a directory that belongs to this user and
    // that nobody else can write to, and only when the SHA-256 digests stored in it match: `key` = the generated source,
    // the architecture string and the hiprtc version (so a file can never stand for another list), `body` = the code
    // object itself (a truncated or edited file is recompiled, not loaded).
    int rtc_major = 0, rtc_minor = 0;
    (void)hiprtcVersion(&rtc_major, &rtc_minor);
    std::string cache_path;
    uint8_t key[32] = {0};
    if (const char* dir = std::getenv("RK_JIT_CACHE_DIR")) {
        if (*dir && cache_dir_is_private(dir)) {
            Sha256 h;
            h.update(src.data(), src.size());
            h.update(props.gcnArchName, std::strlen(props.gcnArchName));
            char tail[96];
            std::snprintf(tail, sizeof tail, "|hiprtc %d.%d|%zu", rtc_major, rtc_minor, src.size());
            h.update(tail, std::strlen(tail));
            h.finish(key);
            char name[80];
            std::snprintf(name, sizeof name, "/rkjit_%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x.hsaco", key[0], key[1], key[2], key[3], key[4],
                          key[5], key[6], key[7], key[8], key[9], key[10], key[11]);
            cache_path = std::string(dir) + name;
        }
    }
    std::vector<char> code;
    if (!cache_path.empty()) code = cache_read(cache_path, key);
    if (code.empty()) {
        hiprtcProgram prog = nullptr;
        if (hiprtcCreateProgram(&prog, src.c_str(), "rk_program.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
            ctx->last_error = "hiprtcCreateProgram failed";
            return RK_ERR_HIP;
        }
        const std::string arch = std::string("--offload-arch=") + props.gcnArchName;
        const char* opts[] = {arch.c_str(), "-O3", "-std=c++17"};
        const hiprtcResult cr = hiprtcCompileProgram(prog, 3, opts);
        if (cr != HIPRTC_SUCCESS) {
            size_t n = 0;
            (void)hiprtcGetProgramLogSize(prog, &n);
            std::string log(n, '\0');
            if (n) (void)hiprtcGetProgramLog(prog, &log[0]);
            ctx->last_error = "hiprtcCompileProgram: " + log.substr(0, 2000);
            (void)hiprtcDestroyProgram(&prog);
            return RK_ERR_HIP;
        }
        size_t code_size = 0;
        (void)hiprtcGetCodeSize(prog, &code_size);
        code.resize(code_size);
        (void)hiprtcGetCode(prog, code.data());
        (void)hiprtcDestroyProgram(&prog);
        if (!cache_path.empty()) cache_write(cache_path, key, code);   // best effort
    }
    RK_HIP_TRY(ctx, hipModuleLoadData(&je.module, code.data()));
    hipError_t e = hipModuleGetFunction(&je.kernel, je.module, "rk_jit_eval_check");
    if (e != hipSuccess) {
        (void)hipModuleUnload(je.module);
        ctx->last_error = std::string("hipModuleGetFunction: ") + hipGetErrorString(e);
        return RK_ERR_HIP;
    }
    std::lock_guard<std::mutex> lk(pg->mu);
    if (pg->jit.count(ctx->device)) {  // another thread got there first
        (void)hipModuleUnload(je.module);
        return RK_OK;
    }
    pg->jit.emplace(ctx->device, std::move(je));
    return RK_OK;
    RK_GUARD_END
}

// the HIP source rk_program_compile would hand to hiprtc (for inspection and for building it ahead of time)
int rk_program_source(const rk_program* pg, char* out, size_t capacity, size_t* length) {
    RK_GUARD_BEGIN
    if (!pg || !length) return RK_ERR_INVALID;
    std::vector<uint32_t> powers;
    const std::string src = generate(*pg, &powers);
    *length = src.size();
    if (!out || capacity < src.size() + 1) return RK_ERR_CAPACITY;
    std::memcpy(out, src.c_str(), src.size() + 1);
    return RK_OK;
    RK_GUARD_END
}

}  // extern "C"
