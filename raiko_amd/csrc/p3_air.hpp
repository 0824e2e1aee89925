// Shared by p3_air.hip (the AIR front end: rk_air_*, the lookup constraints, the Poseidon2 chip) and p3.hip (prover,
// verifier, shards): what an rk_air holds, and the scoped device buffer both use.
#pragma once
#include "internal.hpp"
#include "circuit_program.hpp"

#include <vector>

struct rk_air {
    std::vector<rk_air_step> steps;
    std::vector<uint32_t> lookups;   // flat interactions (rk_air_create_lookup): constants as Montgomery words, columns as slots of `used`
    std::vector<uint32_t> used;      // the distinct main-trace columns the interactions read
    uint32_t n_lookups = 0, perm_width = 0, n_chal = 0;   // base columns of the permutation trace, words of the challenge vector
    uint32_t width = 0, n_public = 0;
    rk_air_info info{};
    uint32_t sel_mask = 0;   // bit c: selector column c (is_first_row, is_last_row, is_transition) is named by the list
    rk_program* prog = nullptr;
};

namespace rk {

constexpr uint32_t NEXT_BACK = 0xffffffffu;  // a tap "one row ahead": back = -1 modulo any power-of-two domain

struct DevBuf {  // dev_alloc'd block released with the scope
    rk_ctx* ctx = nullptr;
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : ctx(o.ctx), p(o.p) { o.p = nullptr; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        reset();
        ctx = o.ctx;
        p = o.p;
        o.p = nullptr;
        return *this;
    }
    ~DevBuf() { reset(); }
    int alloc(rk_ctx* c, size_t bytes) {
        reset();
        ctx = c;
        return rk::dev_alloc(c, bytes, &p);
    }
    void reset() {
        if (p) (void)rk::dev_free(ctx, p);
        p = nullptr;
    }
    uint32_t* u32() const { return (uint32_t*)p; }
};

}  // namespace rk
