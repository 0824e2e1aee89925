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
// circuit_jit.hip
const JitEntry* program_jit(rk_program* pg, int device);
int program_jit_launch(rk_ctx* ctx, const JitEntry& je, const rk_circuit_view* v, const uint32_t* d_tab, uint32_t glob_base,
                       uint32_t mix_base, uint32_t pw_base, uint32_t* d_check, const uint32_t inv_den[16]);
}  // namespace rk

struct rk_program {
    std::vector<rk_poly_step> steps;
    uint32_t ret = 0;
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
