// One configured Poseidon2 instance chosen at run time (rk_params): the four instantiations of
// p2::Core the library ships -- width 24 or 16, external 4x4 block of the Poseidon2 paper or
// circ(2,3,1,1) -- behind one host-side object, plus the sponge / compression / Fiat-Shamir
// generator built on it (risc0-zkp core/hash/poseidon2/{mod,rng}.rs).  Host code only: the
// kernels are instantiated per Core type in kernels_hash.hip and pick theirs from `kind`.
#pragma once
#include <cstring>
#include <vector>

#include "poseidon2_core.hpp"

namespace p2 {

using K0 = Core<24, 21, 0>;  // risc0
using K1 = Core<24, 21, 1>;
using K2 = Core<16, 13, 0>;
using K3 = Core<16, 13, 1>;  // SP1 / Plonky3 shape
constexpr int MAX_CELLS = 24;

// rk_p3_verify_hashes: while set on the calling thread, every host permutation appends its input state here (what a
// recursion layer's Poseidon2 chip has to prove for the proof being checked).  Host only.
inline thread_local std::vector<uint32_t>* g_permute_log = nullptr;

struct Any {
    int kind = 0;
    bool pad_free = false;  // sponge: last partial block keeps the remaining rate cells instead of zero-padding
    K0::Consts k0;
    K1::Consts k1;
    K2::Consts k2;
    K3::Consts k3;

    int cells() const { return kind < 2 ? 24 : 16; }
    int rate() const { return cells() - OUT; }
    int rounds_partial() const { return kind < 2 ? 21 : 13; }
    int m4() const { return kind & 1; }
    const void* raw() const { return kind == 0 ? (const void*)&k0 : kind == 1 ? (const void*)&k1 : kind == 2 ? (const void*)&k2 : (const void*)&k3; }
    size_t raw_size() const { return kind < 2 ? sizeof(K0::Consts) : sizeof(K2::Consts); }
    static constexpr size_t max_raw_size() { return sizeof(K0::Consts) > sizeof(K2::Consts) ? sizeof(K0::Consts) : sizeof(K2::Consts); }
    const uint32_t* rc_ext() const { return kind == 0 ? k0.rc_ext : kind == 1 ? k1.rc_ext : kind == 2 ? k2.rc_ext : k3.rc_ext; }
    const uint32_t* rc_int() const { return kind == 0 ? k0.rc_int : kind == 1 ? k1.rc_int : kind == 2 ? k2.rc_int : k3.rc_int; }
    const uint32_t* diag() const { return kind == 0 ? k0.diag : kind == 1 ? k1.diag : kind == 2 ? k2.diag : k3.diag; }

    template <class C>
    static void load(typename C::Consts& k, const uint32_t* rc_ext, const uint32_t* rc_int, const uint32_t* diag) {
        std::memcpy(k.rc_ext, rc_ext, sizeof k.rc_ext);
        std::memcpy(k.rc_int, rc_int, sizeof k.rc_int);
        std::memcpy(k.diag, diag, sizeof k.diag);
        C::derive(k);
    }
    // width in {24, 16}, m4 in {0, 1}; tables in Montgomery form, every entry < p (checked by the caller)
    void set(int width, int m4, bool pad_free_, const uint32_t* rc_ext, const uint32_t* rc_int, const uint32_t* diag) {
        kind = (width == 16 ? 2 : 0) + (m4 ? 1 : 0);
        pad_free = pad_free_;
        switch (kind) {
            case 0: load<K0>(k0, rc_ext, rc_int, diag); break;
            case 1: load<K1>(k1, rc_ext, rc_int, diag); break;
            case 2: load<K2>(k2, rc_ext, rc_int, diag); break;
            default: load<K3>(k3, rc_ext, rc_int, diag); break;
        }
    }
    void permute(uint32_t* s) const {
        if (g_permute_log) g_permute_log->insert(g_permute_log->end(), s, s + cells());
        switch (kind) {
            case 0: K0::permute(s, k0); break;
            case 1: K1::permute(s, k1); break;
            case 2: K2::permute(s, k2); break;
            default: K3::permute(s, k3); break;
        }
    }
    // Poseidon2HashFn::hash_elem_slice: overwrite-mode sponge over a contiguous element slice
    void hash_elems(const uint32_t* in, size_t n, uint32_t* digest) const {
        uint32_t s[MAX_CELLS];
        std::memset(s, 0, sizeof s);
        const size_t r = (size_t)rate();
        size_t unmixed = 0;
        for (size_t i = 0; i < n; i++) {
            s[unmixed++] = in[i];
            if (unmixed == r) {
                permute(s);
                unmixed = 0;
            }
        }
        if (unmixed != 0 || (n == 0 && !pad_free)) {
            if (!pad_free)
                for (size_t i = unmixed; i < r; i++) s[i] = 0;
            permute(s);
        }
        std::memcpy(digest, s, OUT * 4);
    }
    // Poseidon2HashFn::hash_pair: state = left || right (|| zeros for width 24), digest = first 8 cells
    void hash_pair(const uint32_t* a, const uint32_t* b, uint32_t* out) const {
        uint32_t s[MAX_CELLS];
        std::memcpy(s, a, 32);
        std::memcpy(s + 8, b, 32);
        std::memset(s + 16, 0, 32);
        permute(s);
        std::memcpy(out, s, 32);
    }
};

// Poseidon2Rng (core/hash/poseidon2/rng.rs): the Fiat-Shamir generator of the transcript
struct Rng {
    const Any* k;
    uint32_t cells[MAX_CELLS];
    unsigned pool_used = 0;
    explicit Rng(const Any* kc) : k(kc) { std::memset(cells, 0, sizeof cells); }
    void mix(const uint32_t* digest) {
        if (pool_used != 0) {
            k->permute(cells);
            pool_used = 0;
        }
        for (int i = 0; i < OUT; i++) cells[i] = bb::add(cells[i], digest[i]);
        k->permute(cells);
    }
    uint32_t random_elem() {
        if (pool_used == (unsigned)k->rate()) {
            k->permute(cells);
            pool_used = 0;
        }
        return cells[pool_used++];
    }
    bb::Ext random_ext() {
        bb::Ext r;
        for (int i = 0; i < 4; i++) r.c[i] = random_elem();
        return r;
    }
    uint32_t random_bits(unsigned bits) {
        uint32_t v = bb::decode(random_elem());
        for (int i = 0; i < 3; i++) v ^= bb::decode(random_elem());
        return v & (uint32_t)(((uint64_t)1 << bits) - 1);
    }
};

}  // namespace p2
