// RV32IM executor + segmenter: the step BEFORE the proving path --
// `ExecutorImpl::from_elf(env, elf).run()` at reference provers/risc0/driver/src/bonsai.rs:267-269,
// which interprets the guest and cuts the run into segments of at most 2^segment_limit_po2 cycles
// (bonsai.rs:249) that `session.prove()` (bonsai.rs:271) then proves one by one.
//
// Host code (no GPU work: the reference's executor is CPU code too, single-threaded).  What is
// restated is the public part: the RV32IM instruction set (RISC-V unprivileged spec 2.2: RV32I +
// M), a little-endian paged memory, an ELF32 loader, and the cut into power-of-two segments.
// What is NOT risc0's (its executor lives in risc0-zkvm / risc0-circuit-rv32im 1.0.1, outside the
// reference tree): the cycle model (one cycle per instruction here; risc0 charges paging and
// multi-cycle ecalls), the ecall table (a three-call stand-in: halt / read input words / commit to
// the journal) and the state digest.  The witness layout of the rv32im circuit is not available
// either, so a segment here carries its bounds and state digests, not trace columns.
#include <algorithm>
#include <cstring>
#include <map>
#include <thread>
#include <memory>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/raiko_hip.h"
#include "internal.hpp"

namespace {

constexpr uint32_t PAGE_WORDS = 1024;  // 4 KiB pages
constexpr uint32_t MIN_PO2 = 13;       // risc0's MIN_CYCLES_PO2 (RECALLED): the smallest segment it proves

struct Machine {
    uint32_t pc = 0;
    uint32_t x[32] = {0};
    std::map<uint32_t, std::vector<uint32_t>> pages;  // page index -> 1024 words (ordered: digests walk it)
    // the last page touched by an instruction fetch and by a data access: the map is searched only when an
    // access leaves them (map nodes and the vectors inside them never move once created)
    uint32_t fetch_idx = 0xffffffffu, data_idx = 0xffffffffu;
    uint32_t* fetch_page = nullptr;
    uint32_t* data_page = nullptr;

    uint32_t* page_for_write(uint32_t idx) {
        if (idx == data_idx && data_page) return data_page;
        auto& pg = pages[idx];
        if (pg.empty()) {
            pg.assign(PAGE_WORDS, 0);
            if (idx == fetch_idx) fetch_page = pg.data();  // a page first seen absent by the fetch path now exists
        }
        data_idx = idx;
        data_page = pg.data();
        return data_page;
    }
    uint32_t* word_ptr(uint32_t addr) { return page_for_write(addr >> 12) + ((addr >> 2) & (PAGE_WORDS - 1)); }
    uint32_t load_word(uint32_t addr) {
        const uint32_t idx = addr >> 12;
        if (idx == data_idx && data_page) return data_page[(addr >> 2) & (PAGE_WORDS - 1)];
        auto it = pages.find(idx);
        if (it == pages.end()) return 0u;
        data_idx = idx;
        data_page = it->second.data();
        return data_page[(addr >> 2) & (PAGE_WORDS - 1)];
    }
    uint32_t fetch(uint32_t addr) {
        const uint32_t idx = addr >> 12;
        if (idx != fetch_idx) {
            auto it = pages.find(idx);
            fetch_idx = idx;
            fetch_page = it == pages.end() ? nullptr : it->second.data();
        }
        return fetch_page ? fetch_page[(addr >> 2) & (PAGE_WORDS - 1)] : 0u;
    }
    uint8_t load_byte(uint32_t addr) { return (uint8_t)(load_word(addr & ~3u) >> (8 * (addr & 3))); }
    void store_byte(uint32_t addr, uint8_t v) {
        uint32_t* w = word_ptr(addr & ~3u);
        unsigned sh = 8 * (addr & 3);
        *w = (*w & ~(0xffu << sh)) | ((uint32_t)v << sh);
    }
};

// digest of the machine state: Poseidon2 sponge (the default instance) over pc, the registers and
// every touched page (index, then its words), each 32-bit word as two 16-bit field elements so the
// encoding is injective.  Stands where risc0's SystemState { pc, merkle_root } stands.
void state_digest(const p2::Any& k, const Machine& m, uint32_t* out8) {
    std::vector<uint32_t> e;
    auto put = [&](uint32_t w) {
        e.push_back(bb::encode(w & 0xffffu));
        e.push_back(bb::encode(w >> 16));
    };
    put(m.pc);
    for (int i = 0; i < 32; i++) put(m.x[i]);
    for (const auto& kv : m.pages) {
        bool any = false;
        for (uint32_t w : kv.second) any |= w != 0;
        if (!any) continue;  // an all-zero page is the same as an absent one
        put(kv.first);
        for (uint32_t w : kv.second) put(w);
    }
    k.hash_elems(e.data(), e.size(), out8);
}

}  // namespace

// one executed cycle as the witness generator needs it
struct TraceRow {
    uint32_t pc, ins, a, b, res, next, wr;
};

struct rk_exec {
    std::vector<rk_exec_segment> segments;
    std::vector<std::vector<TraceRow>> traces;   // per segment, when rk_exec_opts.record_trace is set
    // the machine between segments (rk_exec_open / rk_exec_next_segment run it one segment at a time)
    Machine m;
    std::unique_ptr<p2::Any> k;                  // the default Poseidon2 instance: state digests at the boundaries
    rk_exec_opts o{};
    std::vector<uint32_t> input;                 // copy of the caller's input words
    size_t in_pos = 0;
    uint64_t total = 0;
    bool halted = false;
    int st = RK_OK;
    std::vector<uint8_t> journal;
    std::unordered_map<uint32_t, uint64_t> pc_cycles;   // rk_exec_opts.profile
    rk_exec_summary summary{};
    std::string error;
};

namespace {

int load_elf(Machine& m, const uint8_t* elf, size_t n, std::string& err) {
    auto rd16 = [&](size_t o) { return (uint32_t)elf[o] | (uint32_t)elf[o + 1] << 8; };
    auto rd32 = [&](size_t o) { return rd16(o) | rd16(o + 2) << 16; };
    if (n < 52 || std::memcmp(elf, "\x7f" "ELF", 4) != 0) { err = "not an ELF file"; return RK_ERR_INVALID; }
    if (elf[4] != 1 || elf[5] != 1) { err = "not a 32-bit little-endian ELF"; return RK_ERR_INVALID; }
    if (rd16(18) != 243) { err = "not a RISC-V ELF (e_machine != 243)"; return RK_ERR_INVALID; }
    m.pc = rd32(24);
    uint32_t phoff = rd32(28), phentsize = rd16(42), phnum = rd16(44);
    if (phentsize < 32 || (uint64_t)phoff + (uint64_t)phentsize * phnum > n) { err = "program headers out of range"; return RK_ERR_INVALID; }
    for (uint32_t i = 0; i < phnum; i++) {
        size_t ph = phoff + (size_t)i * phentsize;
        if (rd32(ph) != 1) continue;  // PT_LOAD
        uint32_t off = rd32(ph + 4), vaddr = rd32(ph + 8), filesz = rd32(ph + 16), memsz = rd32(ph + 20);
        if ((uint64_t)off + filesz > n || filesz > memsz || (uint64_t)vaddr + memsz > 0x100000000ull) {
            err = "PT_LOAD segment out of range";
            return RK_ERR_INVALID;
        }
        for (uint32_t b = 0; b < filesz; b++) m.store_byte(vaddr + b, elf[off + b]);
    }
    if (m.pc & 3) { err = "misaligned entry point"; return RK_ERR_INVALID; }
    return RK_OK;
}

inline int32_t sext(uint32_t v, unsigned bits) { return (int32_t)(v << (32 - bits)) >> (32 - bits); }

// one instruction; returns 0 to go on, 1 halted, negative rk_status on a trap
int step(Machine& m, rk_exec& ex, const rk_exec_opts& o, size_t& in_pos, std::string& err, TraceRow* row) {
    const uint32_t pc = m.pc, ins = m.fetch(pc);
    const uint32_t opc = ins & 0x7f, rd = (ins >> 7) & 31, f3 = (ins >> 12) & 7, rs1 = (ins >> 15) & 31, rs2 = (ins >> 20) & 31,
                   f7 = ins >> 25;
    const uint32_t a = m.x[rs1], b = m.x[rs2];
    uint32_t next = pc + 4, res = 0;
    bool wr = false;
    auto trap = [&](const char* what) {
        char buf[96];
        std::snprintf(buf, sizeof buf, "%s at pc 0x%08x (instruction 0x%08x)", what, pc, ins);
        err = buf;
        return RK_ERR_INVALID;
    };
    switch (opc) {
        case 0x37: res = ins & 0xfffff000u; wr = true; break;                       // LUI
        case 0x17: res = pc + (ins & 0xfffff000u); wr = true; break;                // AUIPC
        case 0x6f: {                                                                 // JAL
            uint32_t imm = ((ins >> 31) << 20) | (((ins >> 12) & 0xff) << 12) | (((ins >> 20) & 1) << 11) | (((ins >> 21) & 0x3ff) << 1);
            res = pc + 4; wr = true; next = pc + (uint32_t)sext(imm, 21);
            break;
        }
        case 0x67:                                                                   // JALR
            if (f3 != 0) return trap("illegal instruction");
            res = pc + 4; wr = true; next = (a + (uint32_t)sext(ins >> 20, 12)) & ~1u;
            break;
        case 0x63: {                                                                 // branches
            uint32_t imm = ((ins >> 31) << 12) | (((ins >> 7) & 1) << 11) | (((ins >> 25) & 0x3f) << 5) | (((ins >> 8) & 0xf) << 1);
            bool t;
            switch (f3) {
                case 0: t = a == b; break;
                case 1: t = a != b; break;
                case 4: t = (int32_t)a < (int32_t)b; break;
                case 5: t = (int32_t)a >= (int32_t)b; break;
                case 6: t = a < b; break;
                case 7: t = a >= b; break;
                default: return trap("illegal instruction");
            }
            if (t) next = pc + (uint32_t)sext(imm, 13);
            break;
        }
        case 0x03: {                                                                 // loads
            uint32_t addr = a + (uint32_t)sext(ins >> 20, 12);
            switch (f3) {
                case 0: res = (uint32_t)(int32_t)(int8_t)m.load_byte(addr); break;
                case 4: res = m.load_byte(addr); break;
                case 1: case 5: {
                    if (addr & 1) return trap("misaligned halfword load");
                    uint32_t h = m.load_byte(addr) | (uint32_t)m.load_byte(addr + 1) << 8;
                    res = f3 == 1 ? (uint32_t)sext(h, 16) : h;
                    break;
                }
                case 2:
                    if (addr & 3) return trap("misaligned word load");
                    res = m.load_word(addr);
                    break;
                default: return trap("illegal instruction");
            }
            wr = true;
            break;
        }
        case 0x23: {                                                                 // stores
            uint32_t addr = a + (uint32_t)sext(((ins >> 25) << 5) | ((ins >> 7) & 31), 12);
            switch (f3) {
                case 0: m.store_byte(addr, (uint8_t)b); break;
                case 1:
                    if (addr & 1) return trap("misaligned halfword store");
                    m.store_byte(addr, (uint8_t)b); m.store_byte(addr + 1, (uint8_t)(b >> 8));
                    break;
                case 2:
                    if (addr & 3) return trap("misaligned word store");
                    *m.word_ptr(addr) = b;
                    break;
                default: return trap("illegal instruction");
            }
            break;
        }
        case 0x13: {                                                                 // OP-IMM
            uint32_t imm = (uint32_t)sext(ins >> 20, 12), sh = rs2;
            switch (f3) {
                case 0: res = a + imm; break;
                case 2: res = (int32_t)a < (int32_t)imm; break;
                case 3: res = a < imm; break;
                case 4: res = a ^ imm; break;
                case 6: res = a | imm; break;
                case 7: res = a & imm; break;
                case 1: if (f7 != 0) return trap("illegal instruction"); res = a << sh; break;
                case 5:
                    if (f7 == 0) res = a >> sh;
                    else if (f7 == 0x20) res = (uint32_t)((int32_t)a >> sh);
                    else return trap("illegal instruction");
                    break;
            }
            wr = true;
            break;
        }
        case 0x33: {                                                                 // OP (RV32I + M)
            if (f7 == 1) {
                const int64_t sa = (int32_t)a, sb = (int32_t)b;
                switch (f3) {
                    case 0: res = a * b; break;                                                   // MUL
                    case 1: res = (uint32_t)((uint64_t)(sa * sb) >> 32); break;                   // MULH
                    case 2: res = (uint32_t)((uint64_t)(sa * (int64_t)(uint64_t)b) >> 32); break; // MULHSU
                    case 3: res = (uint32_t)(((uint64_t)a * b) >> 32); break;                     // MULHU
                    case 4: res = b == 0 ? 0xffffffffu : (a == 0x80000000u && b == 0xffffffffu) ? a : (uint32_t)((int32_t)a / (int32_t)b); break;
                    case 5: res = b == 0 ? 0xffffffffu : a / b; break;
                    case 6: res = b == 0 ? a : (a == 0x80000000u && b == 0xffffffffu) ? 0 : (uint32_t)((int32_t)a % (int32_t)b); break;
                    case 7: res = b == 0 ? a : a % b; break;
                }
            } else if (f7 == 0 || f7 == 0x20) {
                const bool alt = f7 == 0x20;
                switch (f3) {
                    case 0: res = alt ? a - b : a + b; break;
                    case 5: res = alt ? (uint32_t)((int32_t)a >> (b & 31)) : a >> (b & 31); break;
                    default:
                        if (alt) return trap("illegal instruction");
                        switch (f3) {
                            case 1: res = a << (b & 31); break;
                            case 2: res = (int32_t)a < (int32_t)b; break;
                            case 3: res = a < b; break;
                            case 4: res = a ^ b; break;
                            case 6: res = a | b; break;
                            case 7: res = a & b; break;
                        }
                }
            } else {
                return trap("illegal instruction");
            }
            wr = true;
            break;
        }
        case 0x0f: break;                                                            // FENCE: no-op
        case 0x73: {                                                                 // SYSTEM
            if (ins != 0x00000073u) return trap(ins == 0x00100073u ? "ebreak" : "illegal instruction");
            // the stand-in ecall table: t0 selects the call
            switch (m.x[5]) {
                case RK_ECALL_HALT:
                    ex.summary.exit_code = m.x[10];
                    if (row) *row = TraceRow{pc, ins, a, b, 0u, next, 0u};
                    m.pc = next;
                    return 1;
                case RK_ECALL_READ: {  // a0 = destination (word aligned), a1 = capacity in words -> a0 = words read
                    uint32_t dst = m.x[10], cap = m.x[11], got = 0;
                    if (dst & 3) return trap("misaligned read destination");
                    while (got < cap && in_pos < o.n_input_words) {
                        *m.word_ptr(dst + 4 * got) = o.input_words[in_pos++];
                        got++;
                    }
                    m.x[10] = got;
                    break;
                }
                case RK_ECALL_COMMIT: {  // a0 = source, a1 = bytes: appended to the journal
                    uint32_t src = m.x[10], len = m.x[11];
                    if (ex.journal.size() + (size_t)len > ((size_t)1 << 24)) return trap("journal larger than 16 MiB");
                    for (uint32_t i = 0; i < len; i++) ex.journal.push_back(m.load_byte(src + i));
                    break;
                }
                default: return trap("unknown ecall");
            }
            break;
        }
        default: return trap("illegal instruction");
    }
    if (wr && rd != 0) m.x[rd] = res;
    if (next & 3) return trap("misaligned jump target");
    if (row) *row = TraceRow{pc, ins, a, b, (wr && rd != 0) ? res : 0u, next, (wr && rd != 0) ? 1u : 0u};
    m.pc = next;
    return 0;
}

void refresh_summary(rk_exec* ex) {
    ex->summary.total_cycles = ex->total;
    ex->summary.n_segments = (uint32_t)ex->segments.size();
    ex->summary.journal_bytes = ex->journal.size();
    ex->summary.input_words_read = ex->in_pos;
    ex->summary.status = ex->st;
}

int exec_open(const uint8_t* elf, size_t elf_bytes, const rk_exec_opts* o, rk_exec** out) {
    if (!out) return RK_ERR_INVALID;
    *out = nullptr;
    if (!elf || !o || o->struct_size != sizeof(rk_exec_opts)) return RK_ERR_INVALID;
    if (o->segment_limit_po2 < MIN_PO2 || o->segment_limit_po2 > 24) return RK_ERR_INVALID;
    if (o->n_input_words && !o->input_words) return RK_ERR_INVALID;
    auto ex = std::make_unique<rk_exec>();
    ex->o = *o;
    if (o->n_input_words) ex->input.assign(o->input_words, o->input_words + o->n_input_words);
    ex->o.input_words = ex->input.data();
    ex->st = load_elf(ex->m, elf, elf_bytes, ex->error);
    rk::Sys sys;
    ex->k = std::make_unique<p2::Any>();
    rk_params def;
    rk::params_preset(&def, RK_PRESET_RISC0);
    if (ex->st == RK_OK) ex->st = rk::resolve_params(&def, &sys, ex->k.get());
    refresh_summary(ex.get());
    const int st = ex->st;
    *out = ex.release();  // also on failure: the caller reads the error text, then frees
    return st;
}

// one more segment; *more = 0 once the guest has halted (or the run has failed)
int exec_next(rk_exec* ex, int* more) {
    if (more) *more = 0;
    if (ex->st != RK_OK || ex->halted) return ex->st;
    Machine& m = ex->m;
    const rk_exec_opts& o = ex->o;
    const uint64_t limit = (uint64_t)1 << o.segment_limit_po2;
    rk_exec_segment seg{};
    seg.index = (uint32_t)ex->segments.size();
    seg.start_pc = m.pc;
    state_digest(*ex->k, m, seg.pre_state);
    uint64_t cycles = 0;
    std::vector<TraceRow> trace;
    if (o.record_trace) trace.reserve((size_t)std::min<uint64_t>(limit, (uint64_t)1 << 22));  // 28 bytes per cycle, no regrowth copies
    while (cycles < limit) {
        if (o.session_limit && ex->total >= o.session_limit) {
            ex->error = "session limit reached";
            ex->st = RK_ERR_CAPACITY;
            break;
        }
        TraceRow row{};
        if (o.profile) ex->pc_cycles[m.pc]++;
        int r = step(m, *ex, o, ex->in_pos, ex->error, o.record_trace ? &row : nullptr);
        if (r < 0) { ex->st = r; break; }
        if (o.record_trace) trace.push_back(row);
        cycles++;
        ex->total++;
        if (r == 1) { ex->halted = true; break; }
    }
    if (ex->st == RK_OK) {
        seg.cycles = cycles;
        uint32_t po2 = MIN_PO2;
        while (((uint64_t)1 << po2) < cycles) po2++;
        seg.po2 = po2;
        seg.end_pc = m.pc;
        seg.exit = ex->halted ? RK_EXIT_HALTED : RK_EXIT_SYSTEM_SPLIT;
        state_digest(*ex->k, m, seg.post_state);
        ex->segments.push_back(seg);
        if (o.record_trace) ex->traces.push_back(std::move(trace));
        if (ex->segments.size() > (1u << 20)) { ex->error = "more than 2^20 segments"; ex->st = RK_ERR_CAPACITY; }
    }
    refresh_summary(ex);
    if (more) *more = (ex->st == RK_OK && !ex->halted) ? 1 : 0;
    return ex->st;
}

int exec_elf(const uint8_t* elf, size_t elf_bytes, const rk_exec_opts* o, rk_exec** out) {
    int st = exec_open(elf, elf_bytes, o, out);
    if (st != RK_OK) return st;
    int more = 1;
    while (more) st = exec_next(*out, &more);
    return st;
}

}  // namespace

// The same columns written by the GPU: one lane per row, the trace rows (28 bytes per cycle) are the only upload --
// 2.5x less over PCIe than the 18 finished columns and none of the host's time (rk_exec_witness_device).
__global__ void exec_witness_kernel(uint32_t* __restrict__ code, uint32_t* __restrict__ data, const TraceRow* __restrict__ tr,
                                    size_t cycles, size_t n, uint32_t end_pc, uint32_t rows_only) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool active = i < cycles;
    TraceRow r{};
    if (active) r = tr[i];
    else r.pc = r.next = end_pc;
    const uint32_t lo = r.pc & 0xffffu, carry = (active && lo + 4 > 0xffffu) ? 1u : 0u;
    const uint32_t seq = (active && r.next == r.pc + 4 && r.pc <= 0xfffffffbu) ? 1u : 0u;
    // rows_only: the 16 data columns as one row-major row per lane (64 contiguous bytes: an rk_p3_table), no code columns
    auto put = [&](uint32_t* base, unsigned col, uint32_t canon) {
        base[rows_only ? i * RK_TRACE_DATA_COLS + col : (size_t)col * n + i] = bb::mul(canon, bb::R2);
    };
    if (!rows_only) {
        put(code, 0, i == 0 ? 1u : 0u);
        put(code, 1, i + 1 == n ? 1u : 0u);
    }
    put(data, 0, lo);
    put(data, 1, r.pc >> 16);
    put(data, 2, r.next & 0xffffu);
    put(data, 3, r.next >> 16);
    put(data, 4, r.ins & 0xffffu);
    put(data, 5, r.ins >> 16);
    put(data, 6, seq);
    put(data, 7, seq ? carry : 0u);
    put(data, 8, r.a & 0xffffu);
    put(data, 9, r.a >> 16);
    put(data, 10, r.b & 0xffffu);
    put(data, 11, r.b >> 16);
    put(data, 12, r.res & 0xffffu);
    put(data, 13, r.res >> 16);
    put(data, 14, r.wr);
    put(data, 15, active ? 1u : 0u);
}

extern "C" {

int rk_exec_elf(const uint8_t* elf, size_t elf_bytes, const rk_exec_opts* opts, rk_exec** out) {
    RK_GUARD_BEGIN
    return exec_elf(elf, elf_bytes, opts, out);
    RK_GUARD_END
}
int rk_exec_open(const uint8_t* elf, size_t elf_bytes, const rk_exec_opts* opts, rk_exec** out) {
    RK_GUARD_BEGIN
    return exec_open(elf, elf_bytes, opts, out);
    RK_GUARD_END
}
int rk_exec_next_segment(rk_exec* ex, int* more) {
    RK_GUARD_BEGIN
    if (!ex) return RK_ERR_INVALID;
    return exec_next(ex, more);
    RK_GUARD_END
}
int rk_exec_summary_get(const rk_exec* ex, rk_exec_summary* out) {
    if (!ex || !out) return RK_ERR_INVALID;
    *out = ex->summary;
    return RK_OK;
}
int rk_exec_segment_get(const rk_exec* ex, uint32_t index, rk_exec_segment* out) {
    if (!ex || !out || index >= ex->segments.size()) return RK_ERR_INVALID;
    *out = ex->segments[index];
    return RK_OK;
}
int rk_exec_profile(const rk_exec* ex, uint32_t* pcs, uint64_t* cycles, size_t capacity, size_t* n) {
    RK_GUARD_BEGIN
    if (!ex || !n) return RK_ERR_INVALID;
    *n = ex->pc_cycles.size();
    if (*n > capacity || (*n && (!pcs || !cycles))) return RK_ERR_CAPACITY;
    std::vector<std::pair<uint64_t, uint32_t>> v;
    v.reserve(*n);
    for (const auto& kv : ex->pc_cycles) v.push_back({kv.second, kv.first});
    std::sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
    for (size_t i = 0; i < v.size(); i++) {
        pcs[i] = v[i].second;
        cycles[i] = v[i].first;
    }
    return RK_OK;
    RK_GUARD_END
}
int rk_exec_journal(const rk_exec* ex, uint8_t* out, size_t capacity, size_t* len) {
    if (!ex || !len) return RK_ERR_INVALID;
    *len = ex->journal.size();
    if (ex->journal.size() > capacity || (!out && !ex->journal.empty())) return RK_ERR_CAPACITY;
    if (!ex->journal.empty()) std::memcpy(out, ex->journal.data(), ex->journal.size());
    return RK_OK;
}
// Witness of one executed segment for the stand-in trace circuit (include/raiko_hip.h): every 32-bit word as two
// 16-bit field elements; rows beyond the executed cycles repeat the final pc with active = seq = 0.
int rk_exec_witness(const rk_exec* ex, uint32_t index, uint32_t* code, uint32_t* data) {
    RK_GUARD_BEGIN
    if (!ex || !code || !data || index >= ex->segments.size() || index >= ex->traces.size()) return RK_ERR_INVALID;
    const rk_exec_segment& seg = ex->segments[index];
    const std::vector<TraceRow>& tr = ex->traces[index];
    const size_t n = (size_t)1 << seg.po2;
    if (tr.size() != seg.cycles || tr.size() > n) return RK_ERR_INTERNAL;
    // every cell is below 2^16 (or a bit): one Montgomery product each; rows are independent, four threads split them
    auto put = [&](uint32_t* base, unsigned col, size_t i, uint32_t canon) { base[(size_t)col * n + i] = bb::mul(canon, bb::R2); };
    auto fill = [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) {
            put(code, 0, i, i == 0 ? 1u : 0u);         // first row
            put(code, 1, i, i + 1 == n ? 1u : 0u);     // last row
            const bool active = i < tr.size();
            TraceRow r{};
            if (active) r = tr[i];
            else r.pc = r.next = seg.end_pc;           // padding: stay where the segment ended
            const uint32_t lo = r.pc & 0xffffu, carry = (active && lo + 4 > 0xffffu) ? 1u : 0u;
            const uint32_t seq = (active && r.next == r.pc + 4 && r.pc <= 0xfffffffbu) ? 1u : 0u;  // no wrap of the 32-bit pc
            put(data, 0, i, lo);
            put(data, 1, i, r.pc >> 16);
            put(data, 2, i, r.next & 0xffffu);
            put(data, 3, i, r.next >> 16);
            put(data, 4, i, r.ins & 0xffffu);
            put(data, 5, i, r.ins >> 16);
            put(data, 6, i, seq);
            put(data, 7, i, seq ? carry : 0u);
            put(data, 8, i, r.a & 0xffffu);
            put(data, 9, i, r.a >> 16);
            put(data, 10, i, r.b & 0xffffu);
            put(data, 11, i, r.b >> 16);
            put(data, 12, i, r.res & 0xffffu);
            put(data, 13, i, r.res >> 16);
            put(data, 14, i, r.wr);
            put(data, 15, i, active ? 1u : 0u);
        }
    };
    const size_t n_threads = n >= (1u << 16) ? 4 : 1;
    std::vector<std::thread> pool;
    for (size_t t = 1; t < n_threads; t++) pool.emplace_back(fill, n * t / n_threads, n * (t + 1) / n_threads);
    fill(0, n / n_threads);
    for (auto& th : pool) th.join();
    return RK_OK;
    RK_GUARD_END
}
int rk_exec_lookup_tables(const rk_exec* ex, uint32_t index, uint32_t* range_table, uint32_t* program_table, size_t* program_rows) {
    RK_GUARD_BEGIN
    if (!ex || !range_table || !program_rows || index >= ex->segments.size() || index >= ex->traces.size()) return RK_ERR_INVALID;
    const std::vector<TraceRow>& tr = ex->traces[index];
    if (tr.size() != ex->segments[index].cycles) return RK_ERR_INTERNAL;
    // one pass over the executed cycles: how often each (pc, instruction) pair ran, how often each 16-bit value occurs
    // among the ten limbs a row sends to the range table
    std::vector<uint32_t> hist((size_t)1 << 16, 0);
    std::unordered_map<uint64_t, uint32_t> seen;
    // programs run from a few KiB of text: count per word of the executed pc range; the map only for what does not fit that
    // picture (a range above 16 MiB, an address executed with two different instruction words)
    uint32_t pc_lo = 0xffffffffu, pc_hi = 0;
    for (const TraceRow& r : tr) pc_lo = std::min(pc_lo, r.pc), pc_hi = std::max(pc_hi, r.pc);
    struct Slot {
        uint32_t ins, count;
    };
    std::vector<Slot> direct;
    const bool use_direct = !tr.empty() && (pc_hi - pc_lo) / 4 < (1u << 22);
    if (use_direct) direct.assign((size_t)(pc_hi - pc_lo) / 4 + 1, Slot{0, 0});
    uint64_t last_key = ~(uint64_t)0;
    uint32_t* last = nullptr;
    for (const TraceRow& r : tr) {
        Slot* sl = use_direct && (r.pc & 3u) == 0 ? &direct[(r.pc - pc_lo) / 4] : nullptr;
        if (sl && (sl->count == 0 || sl->ins == r.ins)) {
            sl->ins = r.ins;
            sl->count++;
        } else {
            const uint64_t key = (uint64_t)r.pc << 32 | r.ins;
            if (key != last_key) {
                last = &seen[key];     // references into an unordered_map stay valid across rehashing
                last_key = key;
            }
            ++*last;
        }
        for (uint32_t v : {r.pc, r.next, r.a, r.b, r.res}) {
            hist[v & 0xffffu]++;
            hist[v >> 16]++;
        }
    }
    for (size_t i = 0; i < direct.size(); i++)
        if (direct[i].count) seen[(uint64_t)(pc_lo + 4 * (uint32_t)i) << 32 | direct[i].ins] += direct[i].count;
    size_t rows = 2;
    while (rows < seen.size()) rows <<= 1;
    const size_t capacity = *program_rows;
    *program_rows = rows;
    if (!program_table || capacity < rows) return RK_ERR_CAPACITY;
    auto mont = [](uint32_t canon) { return bb::mul(canon, bb::R2); };
    for (uint32_t v = 0; v < (1u << 16); v++) {
        range_table[2 * (size_t)v] = mont(v);
        range_table[2 * (size_t)v + 1] = mont(hist[v]);
    }
    std::vector<std::pair<uint64_t, uint32_t>> sorted(seen.begin(), seen.end());
    std::sort(sorted.begin(), sorted.end());
    std::memset(program_table, 0, rows * 5 * sizeof(uint32_t));
    for (size_t i = 0; i < sorted.size(); i++) {
        const uint32_t pc = (uint32_t)(sorted[i].first >> 32), ins = (uint32_t)sorted[i].first;
        uint32_t* row = program_table + 5 * i;
        row[0] = mont(pc & 0xffffu);
        row[1] = mont(pc >> 16);
        row[2] = mont(ins & 0xffffu);
        row[3] = mont(ins >> 16);
        row[4] = mont(sorted[i].second);
    }
    return RK_OK;
    RK_GUARD_END
}
static int witness_device(rk_ctx* ctx, const rk_exec* ex, uint32_t index, uint32_t* d_code, uint32_t* d_data, bool rows_only);
int rk_exec_witness_device(rk_ctx* ctx, const rk_exec* ex, uint32_t index, uint32_t* d_code, uint32_t* d_data) {
    RK_GUARD_BEGIN
    if (!d_code) return RK_ERR_INVALID;
    return witness_device(ctx, ex, index, d_code, d_data, false);
    RK_GUARD_END
}
int rk_exec_witness_device_rows(rk_ctx* ctx, const rk_exec* ex, uint32_t index, uint32_t* d_rows) {
    RK_GUARD_BEGIN
    return witness_device(ctx, ex, index, nullptr, d_rows, true);
    RK_GUARD_END
}
static int witness_device(rk_ctx* ctx, const rk_exec* ex, uint32_t index, uint32_t* d_code, uint32_t* d_data, bool rows_only) {
    {
    if (!ctx || !ex || !d_data || index >= ex->segments.size() || index >= ex->traces.size()) return RK_ERR_INVALID;
    const rk_exec_segment& seg = ex->segments[index];
    const std::vector<TraceRow>& tr = ex->traces[index];
    const size_t n = (size_t)1 << seg.po2;
    if (tr.size() != seg.cycles || tr.size() > n) return RK_ERR_INTERNAL;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    void* d_tr = nullptr;
    RK_TRY(rk::dev_alloc(ctx, std::max<size_t>(tr.size(), 1) * sizeof(TraceRow), &d_tr));
    int st = RK_OK;
    if (!tr.empty()) {
        // the trace stays valid while `ex` lives, but the caller may free `ex` right after this call: wait for the copy
        hipError_t e = hipMemcpyAsync(d_tr, tr.data(), tr.size() * sizeof(TraceRow), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            ctx->last_error = std::string("rk_exec_witness_device h2d: ") + hipGetErrorString(e);
            st = RK_ERR_HIP;
        }
    }
    if (st == RK_OK) {
        hipLaunchKernelGGL(exec_witness_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_code, d_data,
                           (const TraceRow*)d_tr, tr.size(), n, seg.end_pc, rows_only ? 1u : 0u);
        st = rk::post_launch(ctx, "exec_witness_kernel");
    }
    rk::dev_free(ctx, d_tr);
    return st;
    }
}
const char* rk_exec_error(const rk_exec* ex) { return ex ? ex->error.c_str() : ""; }
int rk_exec_free(rk_exec* ex) {
    delete ex;
    return RK_OK;
}

}  // extern "C"
