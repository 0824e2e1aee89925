// rk_program (include/raiko_hip.h): the object behind the handle, shared by the compiler / interpreter
// (circuit_program.hip) and the run-time code generator (circuit_jit.hip).
#pragma once
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "internal.hpp"

namespace rk {
constexpr uint32_t PROGRAM_NONE = 0xffffffffu;
struct Tap {
    uint32_t group, offset, back;
};
// a list compiled to a gfx950 code object for one device (circuit_jit.hip)
struct JitEntry {
    hipModule_t module = nullptr;
    hipFunction_t kernel = nullptr;
    uint32_t n_powers = 0;
    std::vector<uint32_t> powers;   // exponents of poly_mix the kernel's table holds, ascending
};
}  // namespace rk

struct rk_program;
namespace rk {
// Where a step list is evaluated: the points x_i = shift * w_d^i, i < d = n << ratio_log2, of a coset that contains the
// trace domain's n-th roots `ratio` times over.  risc0's eval_check has ratio = the blow-up and reads the committed
// LDE itself (col_len = d, stride 0); Plonky3's quotient domain is a sub-coset of the LDE (ratio = log2 of the
// quotient degree <= blow-up: element i << stride of a column of col_len words) and its result leaves split into
// 2^split_log2 chunks (point i = row i >> split of chunk i & (2^split - 1)), column-major 4 columns per chunk.
struct EvalDomain {
    rk_ctx* ctx = nullptr;
    unsigned po2 = 0, ratio_log2 = 0, split_log2 = 0;
    const uint32_t* d_cols[3] = {nullptr, nullptr, nullptr};
    uint32_t group_size[3] = {0, 0, 0};
    uint64_t col_len[3] = {0, 0, 0};
    uint32_t stride_log2[3] = {0, 0, 0};
    const uint32_t* globals = nullptr;
    uint32_t n_globals = 0;
    const uint32_t* mix = nullptr;
    uint32_t n_mix = 0;
};
// circuit_program.hip
int program_create_raw(const rk_poly_step* steps, size_t n_steps, uint32_t ret, std::vector<Tap>&& taps, bool horner, rk_program** out);
int program_eval_domain(const rk_program* prog, const EvalDomain& dom, const uint32_t poly_mix[4], uint32_t* d_out);
// the table of powers of poly_mix a proof's evaluation indexes: entry j = poly_mix^powers[j] (ascending exponents), or
// poly_mix^(chain - 1 - powers[j]) for a Horner-ordered list
void program_power_table(const rk_program* pg, const std::vector<uint32_t>& powers, const uint32_t poly_mix[4], uint32_t wm, uint32_t* out);
// circuit_jit.hip
const JitEntry* program_jit(rk_program* pg, int device);
int program_jit_launch(rk_ctx* ctx, const JitEntry& je, const EvalDomain& dom, const uint32_t* d_tab, uint32_t glob_base,
                       uint32_t mix_base, uint32_t pw_base, uint32_t* d_check, uint32_t inv_base);
}  // namespace rk

struct rk_program {
    std::vector<rk_poly_step> steps;
    uint32_t ret = 0;
    // false: constraint k carries poly_mix^k (risc0's AND_EQZ).  true: the list is one chain of AND_EQZ folded the
    // other way round -- acc = acc * mix + c_k, Plonky3's ConstraintFolder -- so constraint k carries mix^(K - 1 - k)
    bool horner = false;
    std::vector<rk::Tap> taps;
    uint32_t group_min[3] = {0, 0, 0};  // columns a view must have per group

    std::vector<uint4> code;
    std::vector<uint32_t> consts;   // Montgomery
    std::vector<uint32_t> powers;   // distinct exponents of poly_mix, ascending
    uint32_t n_fp_slots = 0, n_mix_slots = 0;
    uint32_t lds_fp = 0, lds_mix = 0;  // how many of them live in LDS (the rest in the HBM scratch matrix)
    uint32_t ret_slot = rk::PROGRAM_NONE;  // NONE: the result is identically zero
    uint32_t need_globals = 0, need_mix = 0;
    rk_program_info info{};

    std::mutex mu;
    std::map<int, void*> d_code;        // per device: the interpreter's op list
    std::map<int, rk::JitEntry> jit;    // per device: the generated kernel, once rk_program_compile has run
};
