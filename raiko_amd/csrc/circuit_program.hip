// The constraint polynomial as data (include/raiko_hip.h, rk_program): compiler, GPU evaluator
// and host evaluator for risc0-zkp 1.0.1's `PolyExtStepDef` step lists (adapter.rs; RECALLED -- the
// crate is outside the reference tree, the call that reaches it is `session.prove()` at reference
// provers/risc0/driver/src/bonsai.rs:271 -> CircuitHal::eval_check, and `receipt.verify()` at
// provers/risc0/driver/src/lib.rs:136 -> CircuitDef::poly_ext).
//
// Compile (host, once per circuit):
//   * validate operands, mark what `ret` depends on, drop the rest;
//   * every mix state's `mul` is a power of poly_mix that does not depend on the data
//     (TRUE: 0, AND_EQZ: +1, AND_COND: k(x) + k(inner)), so only `tot` is computed per point and
//     the powers become one small table per proof;
//   * CONST / GET / GET_GLOBAL are operands, not steps: a tap is re-read from the LDE where it is
//     used (L2 / L1 hits) instead of occupying a slot for the rest of the program;
//   * live ADD / SUB / MUL results and mix tots get slots by a linear scan over last uses, lowest
//     free number first, so the busiest slots are the low ones.
// Evaluate (GPU): a workgroup covers 256 LDE points with 128 lanes (two points per lane: the scalar
// decode of an op is paid once per wave whatever it computes), no synchronisation at all (a lane
// only touches its own columns of the slot array).  Slots below the LDS budget live in LDS as
// [slot][point] (consecutive lanes, consecutive banks), the rest in a global scratch matrix
// [slot][point]; which is which is settled at compile time.  The op list, constants, powers and the
// per-proof tap table are wave-uniform and come through the scalar cache; every branch on an opcode
// or operand kind is a scalar branch.  Bound by scalar instructions (profiles/r02_pmc_program.json).
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <queue>

#include "internal.hpp"
#include "circuit_program.hpp"

namespace {

using bb::Ext;
using rk::Tap;

constexpr uint32_t NONE = rk::PROGRAM_NONE;
// operand = kind << 29 | index.  K_SLOT: a slot kept in LDS; K_SPILL: one in the HBM scratch matrix (which is
// which is settled when the list is compiled); K_TAP: index into the per-proof tap table {column base, shift}.
// The evaluator is bound by scalar instructions (profiles/r02_pmc_program.json), so what can be a table
// lookup (one SMEM instruction) or a compile-time decision is not computed per op
enum : uint32_t { K_SLOT = 0, K_TAP = 1, K_CONST = 2, K_GLOBAL = 3, K_MIX = 4, K_SPILL = 5 };
constexpr uint32_t IDX_MASK = (1u << 29) - 1;
inline uint32_t operand(uint32_t kind, uint32_t idx) { return (kind << 29) | idx; }
// op.x = opcode | dst << 8; DST_SPILL in the opcode byte: the result goes to the HBM scratch matrix
constexpr uint32_t DST_SPILL = 0x80u;
enum : uint32_t { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_EQZ = 3, OP_COND = 4 };
constexpr uint32_t MAX_SLOTS = 1u << 24, MAX_MIX_SLOTS = 1u << 12, MAX_POWERS = 1u << 20;

constexpr int WG = 256;
constexpr uint32_t LDS_WORDS_PER_LANE = 60;  // 60 KB per workgroup at most: two workgroups per CU
constexpr uint32_t LDS_MIX_SLOTS = 6;

}  // namespace

namespace {

struct FpVar {
    uint32_t op = 0, a = 0, b = 0;
    uint32_t opnd = NONE;  // operand encoding once materialised (leaf kinds at once, slots when emitted)
    bool live = false;
    size_t last_use = 0;
};
struct MixVar {
    uint32_t op = 0, x = 0, v = 0, inner = 0;
    uint64_t k = 0;        // mul = poly_mix^k
    bool zero = false;     // tot is identically zero
    bool live = false;
    size_t last_use = 0;
    uint32_t slot = NONE;
};

int compile(rk_program* pg) {
    const size_t n = pg->steps.size();
    std::vector<FpVar> fp;
    std::vector<MixVar> mx;
    std::vector<std::pair<bool, uint32_t>> where(n);  // step -> (is_mix, index in its list)
    fp.reserve(n);
    std::map<uint32_t, uint32_t> const_idx;
    auto intern_const = [&](uint32_t canon) {
        uint32_t m = bb::encode(canon);
        auto it = const_idx.find(m);
        if (it != const_idx.end()) return it->second;
        uint32_t i = (uint32_t)pg->consts.size();
        pg->consts.push_back(m);
        const_idx[m] = i;
        return i;
    };
    for (size_t s = 0; s < n; s++) {
        const rk_poly_step& st = pg->steps[s];
        switch (st.op) {
            case RK_STEP_CONST:
            case RK_STEP_GET:
            case RK_STEP_GET_GLOBAL:
            case RK_STEP_ADD:
            case RK_STEP_SUB:
            case RK_STEP_MUL: {
                FpVar v;
                v.op = st.op;
                v.a = st.a;
                v.b = st.b;
                if (st.op == RK_STEP_GET) {
                    if (st.a >= pg->taps.size()) return RK_ERR_INVALID;
                } else if (st.op == RK_STEP_GET_GLOBAL) {
                    if (st.a > 1 || st.b > IDX_MASK) return RK_ERR_INVALID;
                } else if (st.op != RK_STEP_CONST) {
                    if (st.a >= fp.size() || st.b >= fp.size()) return RK_ERR_INVALID;
                }
                where[s] = {false, (uint32_t)fp.size()};
                fp.push_back(v);
                break;
            }
            case RK_STEP_TRUE:
            case RK_STEP_AND_EQZ:
            case RK_STEP_AND_COND: {
                MixVar m;
                m.op = st.op;
                if (st.op == RK_STEP_TRUE) {
                    m.zero = true;
                } else {
                    if (st.a >= mx.size() || st.b >= fp.size()) return RK_ERR_INVALID;
                    m.x = st.a;
                    m.v = st.b;
                    if (st.op == RK_STEP_AND_COND) {
                        if (st.c >= mx.size()) return RK_ERR_INVALID;
                        m.inner = st.c;
                        m.k = mx[m.x].k + mx[m.inner].k;
                        m.zero = mx[m.x].zero && mx[m.inner].zero;
                    } else {
                        m.k = mx[m.x].k + 1;
                    }
                    if (m.k >= ((uint64_t)1 << 32)) return RK_ERR_INVALID;
                }
                where[s] = {true, (uint32_t)mx.size()};
                mx.push_back(m);
                break;
            }
            default:
                return RK_ERR_INVALID;
        }
    }
    if (pg->ret >= mx.size()) return RK_ERR_INVALID;

    // what the result depends on (operands always precede their users: one backward sweep)
    mx[pg->ret].live = true;
    for (size_t s = n; s-- > 0;) {
        if (where[s].first) {
            const MixVar& m = mx[where[s].second];
            if (!m.live || m.op == RK_STEP_TRUE) continue;
            mx[m.x].live = true;
            if (m.op == RK_STEP_AND_EQZ) {
                fp[m.v].live = true;
            } else if (!mx[m.inner].zero) {  // cond * 0: neither cond nor inner is needed
                mx[m.inner].live = true;
                fp[m.v].live = true;
            }
        } else {
            const FpVar& v = fp[where[s].second];
            if (!v.live) continue;
            if (v.op == RK_STEP_ADD || v.op == RK_STEP_SUB || v.op == RK_STEP_MUL) fp[v.a].live = fp[v.b].live = true;
        }
    }
    // last uses, in step positions
    for (size_t s = 0; s < n; s++) {
        if (where[s].first) {
            const MixVar& m = mx[where[s].second];
            if (!m.live || m.op == RK_STEP_TRUE) continue;
            mx[m.x].last_use = s;
            if (m.op == RK_STEP_AND_EQZ || !mx[m.inner].zero) fp[m.v].last_use = s;
            if (m.op == RK_STEP_AND_COND && !mx[m.inner].zero) mx[m.inner].last_use = s;
        } else {
            const FpVar& v = fp[where[s].second];
            if (!v.live) continue;
            if (v.op == RK_STEP_ADD || v.op == RK_STEP_SUB || v.op == RK_STEP_MUL) fp[v.a].last_use = fp[v.b].last_use = s;
        }
    }
    mx[pg->ret].last_use = n;  // read after the last step

    std::map<uint64_t, uint32_t> power_idx;
    for (size_t s = 0; s < n; s++)
        if (where[s].first) {
            const MixVar& m = mx[where[s].second];
            if (m.live && m.op != RK_STEP_TRUE && !m.zero) power_idx[mx[m.x].k] = 0;
        }
    if (power_idx.size() > MAX_POWERS) return RK_ERR_CAPACITY;
    for (auto& kv : power_idx) {
        kv.second = (uint32_t)pg->powers.size();
        pg->powers.push_back((uint32_t)kv.first);
    }

    // slots: lowest free number first
    struct Pool {
        std::priority_queue<uint32_t, std::vector<uint32_t>, std::greater<uint32_t>> free_list;
        uint32_t next = 0;
        uint32_t take() {
            if (!free_list.empty()) {
                uint32_t s = free_list.top();
                free_list.pop();
                return s;
            }
            return next++;
        }
        void give(uint32_t s) { free_list.push(s); }
    } fp_pool, mx_pool;
    auto fp_operand = [&](uint32_t i) { return fp[i].opnd; };
    auto release_fp = [&](uint32_t i, size_t s) {
        if (fp[i].last_use == s && (fp[i].opnd >> 29) == K_SLOT && fp[i].opnd != NONE) {
            fp_pool.give(fp[i].opnd & IDX_MASK);
            fp[i].last_use = (size_t)-1;  // an operand named twice by one step is released once
        }
    };
    auto release_mx = [&](uint32_t i, size_t s) {
        if (mx[i].last_use == s && mx[i].slot != NONE) {
            mx_pool.give(mx[i].slot);
            mx[i].last_use = (size_t)-1;
        }
    };
    uint64_t n_ops = 0;
    for (size_t s = 0; s < n; s++) {
        if (!where[s].first) {
            FpVar& v = fp[where[s].second];
            if (!v.live) continue;
            switch (v.op) {
                case RK_STEP_CONST:
                    v.opnd = operand(K_CONST, intern_const(v.a));
                    break;
                case RK_STEP_GET: {
                    const Tap& t = pg->taps[v.a];
                    v.opnd = operand(K_TAP, v.a);
                    pg->group_min[t.group] = std::max(pg->group_min[t.group], t.offset + 1);
                    break;
                }
                case RK_STEP_GET_GLOBAL:
                    v.opnd = operand(v.a == 0 ? K_GLOBAL : K_MIX, v.b);
                    if (v.a == 0) pg->need_globals = std::max(pg->need_globals, v.b + 1);
                    else pg->need_mix = std::max(pg->need_mix, v.b + 1);
                    break;
                default: {
                    uint32_t oa = fp_operand(v.a), ob = fp_operand(v.b);
                    release_fp(v.a, s);
                    release_fp(v.b, s);
                    uint32_t dst = fp_pool.take();
                    if (dst >= MAX_SLOTS) return RK_ERR_CAPACITY;
                    v.opnd = operand(K_SLOT, dst);
                    uint32_t oc = v.op == RK_STEP_ADD ? OP_ADD : v.op == RK_STEP_SUB ? OP_SUB : OP_MUL;
                    pg->code.push_back(make_uint4(oc | (dst << 8), oa, ob, 0));
                    n_ops++;
                }
            }
        } else {
            MixVar& m = mx[where[s].second];
            if (!m.live || m.op == RK_STEP_TRUE || m.zero) continue;
            const MixVar& x = mx[m.x];
            uint32_t xs = x.zero ? NONE : x.slot;
            const uint32_t pw = power_idx[x.k];
            // AND_COND with an identically-zero inner state: x.tot + cond * 0 * x.mul, the state of x under
            // a new name -- emitted as x.tot + power * 0 so that slot lifetimes stay one-to-one
            const bool copy_x = m.op == RK_STEP_AND_COND && mx[m.inner].zero;
            const uint32_t ov = copy_x ? operand(K_CONST, intern_const(0)) : fp_operand(m.v);
            if (m.op == RK_STEP_AND_EQZ || copy_x) {
                if (!copy_x) release_fp(m.v, s);
                release_mx(m.x, s);
                uint32_t dst = mx_pool.take();
                if (dst >= MAX_MIX_SLOTS) return RK_ERR_CAPACITY;
                m.slot = dst;
                pg->code.push_back(make_uint4(OP_EQZ | (dst << 8), xs, ov, pw));
            } else {
                uint32_t is = mx[m.inner].slot;
                release_fp(m.v, s);
                release_mx(m.x, s);
                release_mx(m.inner, s);
                uint32_t dst = mx_pool.take();
                if (dst >= MAX_MIX_SLOTS) return RK_ERR_CAPACITY;
                m.slot = dst;
                pg->code.push_back(make_uint4(OP_COND | (dst << 8), xs, ov, pw | (is << 20)));
            }
            n_ops++;
        }
    }
    pg->n_fp_slots = fp_pool.next;
    pg->n_mix_slots = mx_pool.next;
    // which slots live in LDS is a property of the compiled list: the lowest-numbered (busiest) ones
    pg->lds_mix = std::min(pg->n_mix_slots, LDS_MIX_SLOTS);
    pg->lds_fp = std::min(pg->n_fp_slots, LDS_WORDS_PER_LANE - 4 * pg->lds_mix);
    auto place = [&](uint32_t opnd) {
        if (opnd == NONE || (opnd >> 29) != K_SLOT) return opnd;
        const uint32_t slot = opnd & IDX_MASK;
        return slot < pg->lds_fp ? opnd : operand(K_SPILL, slot - pg->lds_fp);
    };
    for (uint4& op : pg->code) {
        const uint32_t oc = op.x & 0xffu, dst = op.x >> 8;
        if (oc <= OP_MUL) {
            op.y = place(op.y);
            op.z = place(op.z);
            if (dst >= pg->lds_fp) op.x = (oc | DST_SPILL) | ((dst - pg->lds_fp) << 8);
        } else {
            op.z = place(op.z);
        }
    }
    pg->ret_slot = mx[pg->ret].zero ? NONE : mx[pg->ret].slot;
    pg->info.n_steps = n;
    pg->info.n_ops = n_ops;
    pg->info.n_fp_slots = pg->n_fp_slots;
    pg->info.n_mix_slots = pg->n_mix_slots;
    pg->info.n_consts = (uint32_t)pg->consts.size();
    pg->info.n_mix_powers = (uint32_t)pg->powers.size();
    pg->info.max_power = (uint32_t)mx[pg->ret].k;
    pg->info.n_taps = (uint32_t)pg->taps.size();
    return RK_OK;
}

// ---------------------------------------------------------------------------------------------
// wave-uniform read-only tables: read through the scalar cache (address space 4 = constant: the
// backend then always selects s_load, whatever it can prove about the kernel's own stores)
#if defined(__HIP_DEVICE_COMPILE__)
#define RK_CONST_AS __attribute__((address_space(4)))
#else
#define RK_CONST_AS
#endif
// the slot array: an explicit LDS pointer.  Through a generic pointer the compiler merges the LDS, HBM-spill
// and tap cases of an operand fetch into one flat_load (the vector-memory path, hundreds of cycles);
// with the address space in the type each case keeps its own instruction and slots are ds_read / ds_write
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) uint32_t lds_u32;
#else
typedef uint32_t lds_u32;
#endif
typedef const RK_CONST_AS uint4* const_u4;
typedef const RK_CONST_AS uint32_t* const_u32;

struct EvalArgs {
    uint64_t code;       // uint4 per op
    uint64_t consts;     // program constants | globals | accum mix
    uint64_t powers;     // 4 words per distinct power of poly_mix
    uint64_t taps;       // uint4 per tap: column base (64-bit), shift in points, log2 of the column's stride per point
    uint32_t* spill;     // [slot][point of the tile]
    uint32_t* check;
    size_t d;
    size_t base, tile;   // the launch covers points [base, base + tile): the spill matrix is sized for one tile
    uint32_t n_ops, glob_base, mix_base;
    uint32_t lds_fp, lds_mix, n_fp_slots;
    uint32_t ret_slot, wm;
    uint32_t blow;       // log2 of (domain size / trace rows)
    uint32_t split;      // the result leaves in 2^split chunks (EvalDomain::split_log2)
    uint32_t inv_den[16];
};

// P points per lane: the evaluator is bound by scalar work per op (operand decode, kind branches), which a
// wave pays once whatever it computes -- with two points per lane that cost is spread over 128 points
// instead of 64.  A workgroup still covers WG = 256 points (its LDS tile is the same), with WG / P lanes;
// point p of lane t is tile point t + p * (WG / P), so every access stays coalesced / conflict-free.
template <int P>
struct Lane {
    static constexpr int NL = WG / P;  // lanes per workgroup
    const EvalArgs& a;
    lds_u32* lds;      // + tid
    size_t i;          // first point of the lane (the others: + NL, ...)
    const_u32 consts;
    const_u4 taps;
    __device__ __forceinline__ void fetch(uint32_t opnd, uint32_t (&out)[P]) const {
        const uint32_t kind = opnd >> 29, idx = opnd & IDX_MASK;
        if (kind == K_SLOT) {
#pragma unroll
            for (int p = 0; p < P; p++) out[p] = lds[idx * WG + p * NL];
        } else if (kind == K_TAP) {
            const uint4 t = taps[idx];
            const char* col = reinterpret_cast<const char*>(((uint64_t)t.y << 32) | t.x);
#pragma unroll
            for (int p = 0; p < P; p++) {
                const uint32_t off = (((uint32_t)i + p * NL + (uint32_t)a.d - t.z) & ((uint32_t)a.d - 1)) << t.w;
                out[p] = *reinterpret_cast<const uint32_t*>(col + ((size_t)off << 2));
            }
        } else if (kind == K_SPILL) {
#pragma unroll
            for (int p = 0; p < P; p++) out[p] = a.spill[(size_t)idx * a.tile + (i - a.base) + p * NL];
        } else {
            const uint32_t c = consts[(kind == K_CONST ? 0u : kind == K_GLOBAL ? a.glob_base : a.mix_base) + idx];
#pragma unroll
            for (int p = 0; p < P; p++) out[p] = c;
        }
    }
    __device__ __forceinline__ void store(uint32_t slot, bool spilled, const uint32_t (&v)[P]) const {
#pragma unroll
        for (int p = 0; p < P; p++) {
            if (!spilled) lds[slot * WG + p * NL] = v[p];
            else a.spill[(size_t)slot * a.tile + (i - a.base) + p * NL] = v[p];
        }
    }
    __device__ __forceinline__ void load_mix(uint32_t slot, Ext (&r)[P]) const {
        if (slot < a.lds_mix) {
            const lds_u32* q = lds + (a.lds_fp + 4 * slot) * WG;
#pragma unroll
            for (int p = 0; p < P; p++)
#pragma unroll
                for (int e = 0; e < 4; e++) r[p].c[e] = q[e * WG + p * NL];
        } else {
            const uint32_t* q = a.spill + ((size_t)(a.n_fp_slots - a.lds_fp) + 4 * (size_t)(slot - a.lds_mix)) * a.tile + (i - a.base);
#pragma unroll
            for (int p = 0; p < P; p++)
#pragma unroll
                for (int e = 0; e < 4; e++) r[p].c[e] = q[(size_t)e * a.tile + p * NL];
        }
    }
    __device__ __forceinline__ void store_mix(uint32_t slot, const Ext (&v)[P]) const {
        if (slot < a.lds_mix) {
            lds_u32* q = lds + (a.lds_fp + 4 * slot) * WG;
#pragma unroll
            for (int p = 0; p < P; p++)
#pragma unroll
                for (int e = 0; e < 4; e++) q[e * WG + p * NL] = v[p].c[e];
        } else {
            uint32_t* q = a.spill + ((size_t)(a.n_fp_slots - a.lds_fp) + 4 * (size_t)(slot - a.lds_mix)) * a.tile + (i - a.base);
#pragma unroll
            for (int p = 0; p < P; p++)
#pragma unroll
                for (int e = 0; e < 4; e++) q[(size_t)e * a.tile + p * NL] = v[p].c[e];
        }
    }
};

// the domain is a multiple of WG when P > 1 (the host picks P = 1 for smaller ones): every lane's points exist
template <int P>
__global__ __launch_bounds__(WG / P) void program_kernel(EvalArgs a) {
    extern __shared__ uint32_t lds_all[];
    constexpr int NL = WG / P;
    const size_t i = a.base + (size_t)blockIdx.x * WG + threadIdx.x;
    if (i >= a.d) return;  // P == 1 only; no barrier anywhere below
    Lane<P> ln{a, (lds_u32*)lds_all + threadIdx.x, i, (const_u32)a.consts, (const_u4)a.taps};
    const const_u4 code = (const_u4)a.code;
    const const_u32 powers = (const_u32)a.powers;
    uint4 next = code[0];  // the list has a spare entry at the end: the fetch of op pc + 1 runs under op pc
    for (uint32_t pc = 0; pc < a.n_ops; pc++) {
        const uint4 op = next;
        next = code[pc + 1];
        const uint32_t oc = op.x & 0x7fu, dst = op.x >> 8;
        if (oc <= OP_MUL) {
            uint32_t x[P], y[P], r[P];
            ln.fetch(op.y, x);
            ln.fetch(op.z, y);
#pragma unroll
            for (int p = 0; p < P; p++) r[p] = oc == OP_ADD ? bb::add(x[p], y[p]) : oc == OP_SUB ? bb::sub(x[p], y[p]) : bb::mul(x[p], y[p]);
            ln.store(dst, (op.x & DST_SPILL) != 0, r);
        } else {
            const const_u32 pw = powers + (size_t)(op.w & 0xfffffu) * 4;
            const Ext pwe{{pw[0], pw[1], pw[2], pw[3]}};
            uint32_t v[P];
            ln.fetch(op.z, v);
            Ext t[P];
#pragma unroll
            for (int p = 0; p < P; p++) t[p] = bb::scale(pwe, v[p]);
            if (oc == OP_COND) {
                Ext in[P];
                ln.load_mix(op.w >> 20, in);
#pragma unroll
                for (int p = 0; p < P; p++) t[p] = bb::mul(t[p], in[p], a.wm);
            }
            if (op.y != NONE) {
                Ext x[P];
                ln.load_mix(op.y, x);
#pragma unroll
                for (int p = 0; p < P; p++) t[p] = bb::add(t[p], x[p]);
            }
            ln.store_mix(dst, t);
        }
    }
    Ext tot[P];
    if (a.ret_slot == NONE) {
#pragma unroll
        for (int p = 0; p < P; p++) tot[p] = bb::ext_zero();
    } else {
        ln.load_mix(a.ret_slot, tot);
    }
#pragma unroll
    for (int p = 0; p < P; p++) {
        const size_t pt = i + p * NL;
        const Ext r = bb::scale(tot[p], a.inv_den[pt & ((1u << a.blow) - 1)]);
        const size_t rows = a.d >> a.split, at = (pt & ((1u << a.split) - 1)) * 4 * rows + (pt >> a.split);
#pragma unroll
        for (int e = 0; e < 4; e++) a.check[at + (size_t)e * rows] = r.c[e];
    }
}

int device_code(rk_program* pg, rk_ctx* ctx, const uint4** out) {
    std::lock_guard<std::mutex> lk(pg->mu);
    auto it = pg->d_code.find(ctx->device);
    if (it == pg->d_code.end()) {
        void* d = nullptr;
        size_t bytes = (pg->code.size() + 1) * sizeof(uint4);  // one spare entry: the evaluator fetches one op ahead
        RK_HIP_TRY(ctx, hipMalloc(&d, bytes));
        hipError_t e = hipMemset(d, 0, bytes);
        if (e == hipSuccess) e = hipMemcpy(d, pg->code.data(), pg->code.size() * sizeof(uint4), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(d);
            ctx->last_error = std::string("program upload: ") + hipGetErrorString(e);
            return RK_ERR_HIP;
        }
        it = pg->d_code.emplace(ctx->device, d).first;
    }
    *out = (const uint4*)it->second;
    return RK_OK;
}

}  // namespace

namespace rk {

void program_power_table(const rk_program* pg, const std::vector<uint32_t>& powers, const uint32_t poly_mix[4], uint32_t wm, uint32_t* out) {
    Ext pm, cur = bb::ext_one();
    std::memcpy(pm.c, poly_mix, 16);
    const size_t npw = powers.size();
    const uint32_t top = pg->horner ? pg->info.max_power - 1 : 0;   // Horner: exponent e stands for top - e
    uint32_t at = 0;
    for (size_t t = 0; t < npw; t++) {  // ascending effective exponents: one running power
        const size_t j = pg->horner ? npw - 1 - t : t;
        const uint32_t e = pg->horner ? top - powers[j] : powers[j];
        cur = bb::mul(cur, bb::pow(pm, e - at, wm), wm);
        at = e;
        std::memcpy(out + 4 * j, cur.c, 16);
    }
}

int program_eval_check(const rk_program* cprog, const rk_circuit_view* v, const uint32_t poly_mix[4], uint32_t* d_check) {
    if (!cprog || !v || !v->ctx || !poly_mix || !d_check) return RK_ERR_INVALID;
    const unsigned blow = v->ctx->sys.blowup_log2;
    if (v->po2 < 1 || v->po2 + blow > ntt::LAMBDA) return RK_ERR_INVALID;
    EvalDomain dom;
    dom.ctx = v->ctx;
    dom.po2 = v->po2;
    dom.ratio_log2 = blow;
    for (int g = 0; g < 3; g++) {
        dom.d_cols[g] = v->d_lde[g];
        dom.group_size[g] = v->group_size[g];
        dom.col_len[g] = (uint64_t)1 << (v->po2 + blow);
    }
    dom.globals = v->globals;
    dom.n_globals = v->n_globals;
    dom.mix = v->mix;
    dom.n_mix = v->n_mix;
    return program_eval_domain(cprog, dom, poly_mix, d_check);
}

int program_eval_domain(const rk_program* cprog, const EvalDomain& v, const uint32_t poly_mix[4], uint32_t* d_check) {
    rk_program* pg = const_cast<rk_program*>(cprog);  // the device copy of the op list is cached inside
    if (!pg || !v.ctx || !poly_mix || !d_check) return RK_ERR_INVALID;
    rk_ctx* ctx = v.ctx;
    const unsigned blow = v.ratio_log2;
    if (v.po2 < 1 || blow > 4 || v.po2 + blow > ntt::LAMBDA || v.split_log2 > blow) return RK_ERR_INVALID;
    if (v.n_globals < pg->need_globals || v.n_mix < pg->need_mix) return RK_ERR_INVALID;
    if ((v.n_globals && !v.globals) || (v.n_mix && !v.mix)) return RK_ERR_INVALID;
    const size_t n = (size_t)1 << v.po2, d = n << blow;
    for (int g = 0; g < 3; g++)
        if (pg->group_min[g] && (!v.d_cols[g] || v.group_size[g] < pg->group_min[g] || v.stride_log2[g] > 8 ||
                                 v.col_len[g] < (d << v.stride_log2[g])))
            return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t wm = ctx->sys.wm;
    // x_i^N for x_i = shift * w_d^i takes d/N values: shift^N * w_(d/N)^(i mod d/N)
    uint32_t inv_den[16] = {0};
    {
        const uint32_t sn = bb::pow(ctx->sys.shiftm, n), wb = bb::pow(ctx->sys.root27m, (uint64_t)1 << (27 - blow));
        for (unsigned r = 0; r < (1u << blow); r++) inv_den[r] = bb::inv(bb::sub(bb::mul(sn, bb::pow(wb, r)), bb::ONE));
    }

    if (const JitEntry* je = program_jit(pg, ctx->device)) {
        // the generated kernel (rk_program_compile): table = globals | mix | the powers its code indexes | 1 / (x^n - 1)
        const size_t o_mix = v.n_globals, o_pw = (o_mix + v.n_mix + 3) & ~(size_t)3, o_inv = o_pw + 4 * (size_t)je->n_powers, words = o_inv + 16;
        std::vector<uint32_t> tab(words + 4, 0);
        if (v.n_globals) std::memcpy(tab.data(), v.globals, (size_t)v.n_globals * 4);
        if (v.n_mix) std::memcpy(&tab[o_mix], v.mix, (size_t)v.n_mix * 4);
        program_power_table(pg, je->powers, poly_mix, wm, &tab[o_pw]);
        std::memcpy(&tab[o_inv], inv_den, sizeof inv_den);
        void* d_tab = nullptr;
        RK_TRY(scratch(ctx, words * 4 + 16, &d_tab));
        RK_TRY(upload(ctx, d_tab, tab.data(), words * 4));  // through the page-locked ring (no wait) when it fits
        return program_jit_launch(ctx, *je, v, (const uint32_t*)d_tab, 0, (uint32_t)o_mix, (uint32_t)o_pw, d_check, (uint32_t)o_inv);
    }
    EvalArgs a{};
    const uint4* d_ops = nullptr;
    RK_TRY(device_code(pg, ctx, &d_ops));
    a.code = (uint64_t)(uintptr_t)d_ops;
    // per-proof tables in one upload: constants | globals | mix | powers | taps
    const size_t nc = pg->consts.size(), npw = pg->powers.size(), ntap = pg->taps.size();
    const size_t o_glob = nc, o_mix = o_glob + v.n_globals, o_pw = (o_mix + v.n_mix + 3) & ~(size_t)3;
    const size_t o_tap = o_pw + 4 * npw, words = o_tap + 4 * ntap;
    std::vector<uint32_t> pack(words + 4, 0);
    std::memcpy(pack.data(), pg->consts.data(), nc * 4);
    if (v.n_globals) std::memcpy(&pack[o_glob], v.globals, (size_t)v.n_globals * 4);
    if (v.n_mix) std::memcpy(&pack[o_mix], v.mix, (size_t)v.n_mix * 4);
    program_power_table(pg, pg->powers, poly_mix, wm, &pack[o_pw]);
    for (size_t t = 0; t < ntap; t++) {  // a tap `back` rows behind is back << blow points behind
        const Tap& tp = pg->taps[t];
        uint64_t col = 0;
        if (tp.group < 3 && v.d_cols[tp.group] && tp.offset < v.group_size[tp.group])
            col = (uint64_t)(uintptr_t)(v.d_cols[tp.group] + (size_t)tp.offset * v.col_len[tp.group]);
        pack[o_tap + 4 * t] = (uint32_t)col;
        pack[o_tap + 4 * t + 1] = (uint32_t)(col >> 32);
        pack[o_tap + 4 * t + 2] = (uint32_t)((((size_t)tp.back) << blow) & (d - 1));
        pack[o_tap + 4 * t + 3] = tp.group < 3 ? v.stride_log2[tp.group] : 0u;
    }
    void* d_pack = nullptr;
    RK_TRY(scratch(ctx, words * 4 + 16, &d_pack));
    RK_TRY(upload(ctx, d_pack, pack.data(), words * 4));  // through the page-locked ring (no wait) when it fits, copy + wait otherwise
    const uint32_t* dp = (const uint32_t*)d_pack;
    a.consts = (uint64_t)(uintptr_t)dp;
    a.glob_base = (uint32_t)o_glob;
    a.mix_base = (uint32_t)o_mix;
    a.powers = (uint64_t)(uintptr_t)(dp + o_pw);
    a.taps = (uint64_t)(uintptr_t)(dp + o_tap);
    a.check = d_check;
    a.d = d;
    a.n_ops = (uint32_t)pg->code.size();
    a.lds_mix = pg->lds_mix;
    a.lds_fp = pg->lds_fp;
    a.n_fp_slots = pg->n_fp_slots;
    a.ret_slot = pg->ret_slot;
    a.wm = wm;
    a.blow = blow;
    a.split = v.split_log2;
    std::memcpy(a.inv_den, inv_den, sizeof a.inv_den);
    // The values that do not fit the LDS budget live in an HBM matrix [slot][point].  Sized for the whole domain it
    // would be live_slots * d words (a list with 2 k live values at 2^22 points: 37 GB per proof, times the proofs in
    // flight): the domain is walked in tiles instead, every tile reusing one matrix of at most SPILL_CAP bytes -- small
    // enough to stay in the 256 MB of Infinity Cache for typical lists.
    const size_t spill_rows = (size_t)(pg->n_fp_slots - a.lds_fp) + 4 * (size_t)(pg->n_mix_slots - a.lds_mix);
    constexpr size_t SPILL_CAP = (size_t)1 << 30;
    size_t tile = d;
    while (spill_rows * tile * 4 > SPILL_CAP && tile > (size_t)WG * 256 && tile % 2 == 0) tile /= 2;
    void* d_spill = nullptr;
    if (spill_rows) RK_TRY(dev_alloc(ctx, spill_rows * tile * 4, &d_spill));
    a.spill = (uint32_t*)d_spill;
    a.tile = tile;
    const size_t lds_bytes = (size_t)(a.lds_fp + 4 * a.lds_mix) * WG * 4;
    int rc = RK_OK;
    for (size_t base = 0; base < d && rc == RK_OK; base += tile) {
        a.base = base;
        const size_t n_pts = std::min(tile, d - base);
        if (n_pts % WG == 0)  // two points per lane: the per-op scalar work is paid once per 128 points
            hipLaunchKernelGGL(program_kernel<2>, dim3((unsigned)(n_pts / WG)), dim3(WG / 2), lds_bytes, ctx->stream, a);
        else
            hipLaunchKernelGGL(program_kernel<1>, dim3((unsigned)((n_pts + WG - 1) / WG)), dim3(WG), lds_bytes, ctx->stream, a);
        rc = post_launch(ctx, "program_kernel");
    }
    if (d_spill) {
        int fr = dev_free(ctx, d_spill);  // stream-ordered: the block is reused only by later work of this stream
        if (rc == RK_OK) rc = fr;
    }
    return rc;
}

// CircuitDef::poly_ext: the step list as written, on extension elements (no compile products used
// except the validated list itself, so this doubles as a check of the compiler)
int program_poly_ext(const rk_program* pg, uint32_t wm, const uint32_t poly_mix[4], const uint32_t* eval_u_ext, size_t n_taps,
                     const uint32_t* globals, uint32_t n_globals, const uint32_t* mix, uint32_t n_mix, uint32_t out_ext[4]) {
    if (!pg || !poly_mix || !out_ext || (n_taps && !eval_u_ext)) return RK_ERR_INVALID;
    if (n_taps != pg->taps.size() || n_globals < pg->need_globals || n_mix < pg->need_mix) return RK_ERR_INVALID;
    const Ext* u = reinterpret_cast<const Ext*>(eval_u_ext);
    Ext pm;
    std::memcpy(pm.c, poly_mix, 16);
    const Ext pm_inv = pg->horner ? bb::inv(pm, wm) : bb::ext_one();
    struct Mix {
        Ext tot, mul;
    };
    std::vector<Ext> fp;
    std::vector<Mix> mx;
    fp.reserve(pg->steps.size());
    for (const rk_poly_step& st : pg->steps) {
        switch (st.op) {
            case RK_STEP_CONST:
                fp.push_back(bb::ext_from(bb::encode(st.a)));
                break;
            case RK_STEP_GET:
                fp.push_back(u[st.a]);
                break;
            case RK_STEP_GET_GLOBAL: {
                // an argument the result does not depend on may lie beyond what the caller has
                uint32_t val = st.a == 0 ? (st.b < n_globals ? globals[st.b] : 0) : (st.b < n_mix ? mix[st.b] : 0);
                fp.push_back(bb::ext_from(val));
                break;
            }
            case RK_STEP_ADD:
                fp.push_back(bb::add(fp[st.a], fp[st.b]));
                break;
            case RK_STEP_SUB:
                fp.push_back(bb::sub(fp[st.a], fp[st.b]));
                break;
            case RK_STEP_MUL:
                fp.push_back(bb::mul(fp[st.a], fp[st.b], wm));
                break;
            case RK_STEP_TRUE:  // Horner-ordered chain: mul starts at mix^(K-1) and goes down
                mx.push_back(Mix{bb::ext_zero(), pg->horner ? bb::pow(pm, pg->info.max_power - 1, wm) : bb::ext_one()});
                break;
            case RK_STEP_AND_EQZ: {
                const Mix x = mx[st.a];
                mx.push_back(Mix{bb::add(x.tot, bb::mul(x.mul, fp[st.b], wm)), bb::mul(x.mul, pg->horner ? pm_inv : pm, wm)});
                break;
            }
            case RK_STEP_AND_COND: {
                const Mix x = mx[st.a], in = mx[st.c];
                mx.push_back(Mix{bb::add(x.tot, bb::mul(bb::mul(fp[st.b], in.tot, wm), x.mul, wm)), bb::mul(x.mul, in.mul, wm)});
                break;
            }
            default:
                return RK_ERR_INVALID;
        }
    }
    std::memcpy(out_ext, mx[pg->ret].tot.c, 16);
    return RK_OK;
}

}  // namespace rk

namespace rk {
// a program over an explicit tap list (no rk_taps behind it): what the AIR front end of p3.hip builds.  A Horner-ordered
// list must be a single AND_EQZ chain (no AND_COND): the exponent reversal is defined on that shape only.
int program_create_raw(const rk_poly_step* steps, size_t n_steps, uint32_t ret, std::vector<Tap>&& taps, bool horner, rk_program** out) {
    if (!out || !steps || n_steps == 0 || n_steps > ((size_t)1 << 28) || taps.size() > IDX_MASK) return RK_ERR_INVALID;
    if (horner)
        for (size_t s = 0; s < n_steps; s++)
            if (steps[s].op == RK_STEP_AND_COND) return RK_ERR_INVALID;
    std::unique_ptr<rk_program> pg(new rk_program);
    pg->steps.assign(steps, steps + n_steps);
    pg->ret = ret;
    pg->horner = horner;
    pg->taps = std::move(taps);
    RK_TRY(compile(pg.get()));
    if (horner && pg->info.max_power == 0) pg->horner = false;  // no constraint at all: nothing to reverse
    *out = pg.release();
    return RK_OK;
}
}  // namespace rk

extern "C" {

int rk_program_create(const rk_poly_step* steps, size_t n_steps, uint32_t ret, const rk_taps* taps, rk_program** out) {
    RK_GUARD_BEGIN
    if (!out) return RK_ERR_INVALID;
    *out = nullptr;
    if (!steps || n_steps == 0 || n_steps > ((size_t)1 << 28) || !taps) return RK_ERR_INVALID;
    RK_TRY(rk::check_taps(*taps));
    std::unique_ptr<rk_program> pg(new rk_program);
    pg->steps.assign(steps, steps + n_steps);
    pg->ret = ret;
    for (uint32_t r = 0; r < taps->n_regs; r++) {
        const uint32_t cb = taps->reg_combo[r];
        for (uint32_t b = taps->combo_off[cb]; b < taps->combo_off[cb + 1]; b++)
            pg->taps.push_back(Tap{taps->reg_group[r], taps->reg_offset[r], taps->combo_backs[b]});
    }
    if (pg->taps.size() > IDX_MASK) return RK_ERR_INVALID;
    RK_TRY(compile(pg.get()));
    *out = pg.release();
    return RK_OK;
    RK_GUARD_END
}

int rk_program_destroy(rk_program* pg) {
    RK_GUARD_BEGIN
    if (!pg) return RK_OK;
    for (auto& kv : pg->d_code) {
        if (hipSetDevice(kv.first) == hipSuccess) (void)hipFree(kv.second);
    }
    for (auto& kv : pg->jit) {
        if (hipSetDevice(kv.first) == hipSuccess && kv.second.module) (void)hipModuleUnload(kv.second.module);
    }
    delete pg;
    return RK_OK;
    RK_GUARD_END
}

int rk_program_get_info(const rk_program* pg, rk_program_info* out) {
    if (!pg || !out) return RK_ERR_INVALID;
    *out = pg->info;
    return RK_OK;
}

int rk_program_eval_check(const rk_program* pg, const rk_circuit_view* view, const uint32_t poly_mix[4], uint32_t* d_check) {
    RK_GUARD_BEGIN
    return rk::program_eval_check(pg, view, poly_mix, d_check);
    RK_GUARD_END
}

int rk_program_poly_ext(const rk_program* pg, uint32_t ext_w, const uint32_t poly_mix[4], const uint32_t* eval_u_ext,
                        size_t n_taps, const uint32_t* globals, uint32_t n_globals, const uint32_t* mix, uint32_t n_mix,
                        uint32_t out_ext[4]) {
    RK_GUARD_BEGIN
    if (ext_w >= bb::P) return RK_ERR_INVALID;
    return rk::program_poly_ext(pg, ext_w ? bb::encode(ext_w) : bb::WM_RISC0, poly_mix, eval_u_ext, n_taps, globals, n_globals,
                                mix, n_mix, out_ext);
    RK_GUARD_END
}

}  // extern "C"
