// The AIR front end of rk_p3_* (include/raiko_hip.h): rk_air_create / rk_air_create_lookup -- validation, symbolic degree,
// translation of the step list into an rk_program for the GPU evaluator, sp1-core's eval_permutation_constraints written
// as steps (PermStepGen) -- and the Poseidon2 chip: its AIR written from the configured instance's constants and its rows
// written on the GPU (rk_p2_chip_*).  Prover and verifier are in p3.hip.  Reference call site of the path:
// provers/sp1/driver/src/lib.rs:44-57; p3-uni-stark symbolic_builder.rs / symbolic_expression.rs, sp1-core
// stark/permutation.rs, sp1-recursion-core's Poseidon2 wide chip: outside the reference tree, RECALLED.
#include "p3_air.hpp"
#include "p3_kernels.hpp"

#include <array>
#include <cstring>
#include <map>
#include <memory>
#include <tuple>

namespace {

using rk::DevBuf;
using rk::NEXT_BACK;

// ---------------------------------------------------------------- AIR: checks, symbolic degree, host evaluation
int air_scan(const rk_air_step* steps, size_t n, uint32_t width, uint32_t n_public, uint32_t perm_width, uint32_t n_chal, rk_air_info* info) {
    std::vector<uint32_t> deg;
    deg.reserve(n);
    uint32_t max_deg = 0, n_con = 0;
    for (size_t s = 0; s < n; s++) {
        const rk_air_step& st = steps[s];
        const size_t nv = deg.size();
        switch (st.op) {
            case RK_AIR_CONST: if (st.a >= bb::P) return RK_ERR_INVALID; deg.push_back(0); break;
            case RK_AIR_LOCAL: case RK_AIR_NEXT: if (st.a >= width) return RK_ERR_INVALID; deg.push_back(1); break;
            case RK_AIR_PUBLIC: if (st.a >= n_public) return RK_ERR_INVALID; deg.push_back(0); break;
            case RK_AIR_IS_FIRST_ROW: case RK_AIR_IS_LAST_ROW: deg.push_back(1); break;
            case RK_AIR_IS_TRANSITION: deg.push_back(0); break;
            case RK_AIR_PERM_LOCAL: case RK_AIR_PERM_NEXT: if (st.a >= perm_width) return RK_ERR_INVALID; deg.push_back(1); break;
            case RK_AIR_CHALLENGE: if (st.a >= n_chal) return RK_ERR_INVALID; deg.push_back(0); break;
            case RK_AIR_CUMSUM: if (st.a >= 4 || perm_width == 0) return RK_ERR_INVALID; deg.push_back(0); break;
            case RK_AIR_ADD: case RK_AIR_SUB:
                if (st.a >= nv || st.b >= nv) return RK_ERR_INVALID;
                deg.push_back(std::max(deg[st.a], deg[st.b]));
                break;
            case RK_AIR_MUL:
                if (st.a >= nv || st.b >= nv) return RK_ERR_INVALID;
                deg.push_back(std::min<uint32_t>(deg[st.a] + deg[st.b], 1u << 20));
                break;
            case RK_AIR_NEG: if (st.a >= nv) return RK_ERR_INVALID; deg.push_back(deg[st.a]); break;
            case RK_AIR_ASSERT_ZERO:
                if (st.a >= nv) return RK_ERR_INVALID;
                max_deg = std::max(max_deg, deg[st.a]);
                n_con++;
                break;
            default: return RK_ERR_INVALID;
        }
    }
    // p3-uni-stark get_log_quotient_degree: log2_ceil(max(constraint degree, 2) - 1)
    info->n_steps = n;
    info->n_constraints = n_con;
    info->max_degree = max_deg;
    info->log_quotient_degree = log2u(std::max(max_deg, 2u) - 1);
    return RK_OK;
}

// sp1-core eval_permutation_constraints (RECALLED) written into a step list over base values: every extension identity
// is four base asserts (the evaluator works on base columns; W is baked in).  `raw` = the caller's interactions as given
// (canonical constants, column numbers).  Appended to `steps`, whose values so far number `nv`.  Equal values are shared.
struct PermStepGen {
    std::vector<rk_air_step>& steps;
    uint32_t nv, w;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> memo;
    using E = std::array<uint32_t, 4>;
    uint32_t push(uint32_t op, uint32_t a = 0, uint32_t b = 0) {
        const auto key = std::make_tuple(op, a, b);
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
        steps.push_back(rk_air_step{op, a, b});
        memo.emplace(key, nv);
        return nv++;
    }
    void assert_zero(uint32_t v) { steps.push_back(rk_air_step{RK_AIR_ASSERT_ZERO, v, 0}); }
    uint32_t add(uint32_t a, uint32_t b) { return push(RK_AIR_ADD, a, b); }
    uint32_t sub(uint32_t a, uint32_t b) { return push(RK_AIR_SUB, a, b); }
    uint32_t mul(uint32_t a, uint32_t b) { return push(RK_AIR_MUL, a, b); }
    uint32_t cst(uint32_t canon) { return push(RK_AIR_CONST, canon); }
    uint32_t col(uint32_t c) { return push(RK_AIR_LOCAL, c); }
    E leaf(uint32_t op, uint32_t at) { return E{push(op, 4 * at), push(op, 4 * at + 1), push(op, 4 * at + 2), push(op, 4 * at + 3)}; }
    E add(const E& x, const E& y) { return E{push(RK_AIR_ADD, x[0], y[0]), push(RK_AIR_ADD, x[1], y[1]), push(RK_AIR_ADD, x[2], y[2]), push(RK_AIR_ADD, x[3], y[3])}; }
    E sub(const E& x, const E& y) { return E{push(RK_AIR_SUB, x[0], y[0]), push(RK_AIR_SUB, x[1], y[1]), push(RK_AIR_SUB, x[2], y[2]), push(RK_AIR_SUB, x[3], y[3])}; }
    E scale(const E& x, uint32_t e) { return E{push(RK_AIR_MUL, x[0], e), push(RK_AIR_MUL, x[1], e), push(RK_AIR_MUL, x[2], e), push(RK_AIR_MUL, x[3], e)}; }
    E mul(const E& x, const E& y) {   // modulo t^4 - W
        E out;
        for (int k = 0; k < 4; k++) {
            uint32_t t = push(RK_AIR_MUL, x[0], y[k]);
            for (int i = 1; i <= k; i++) t = push(RK_AIR_ADD, t, push(RK_AIR_MUL, x[i], y[k - i]));
            if (k < 3) {
                uint32_t h = push(RK_AIR_MUL, x[k + 1], y[3]);
                for (int i = k + 2; i < 4; i++) h = push(RK_AIR_ADD, h, push(RK_AIR_MUL, x[i], y[k + 4 - i]));
                t = push(RK_AIR_ADD, t, push(RK_AIR_MUL, h, push(RK_AIR_CONST, w)));
            }
            out[k] = t;
        }
        return out;
    }
    void assert_ext_zero(int32_t cond, const E& x) {
        for (int k = 0; k < 4; k++) assert_zero(cond < 0 ? x[k] : push(RK_AIR_MUL, (uint32_t)cond, x[k]));
    }
    void run(const uint32_t* raw, uint32_t n_lookups) {
        struct Ix {
            uint32_t kind, bus, is_const, mult, nv;
            const uint32_t* cols;
        };
        std::vector<Ix> its;
        for (uint32_t i = 0; i < n_lookups; i++) {
            its.push_back(Ix{raw[0], raw[1], raw[2], raw[3], raw[4], raw + 5});
            raw += 5 + raw[4];
        }
        const uint32_t nb = (n_lookups + 1) / 2;
        const E alpha = leaf(RK_AIR_CHALLENGE, 0);
        auto rlc = [&](const Ix& it) {
            E acc = add(alpha, scale(leaf(RK_AIR_CHALLENGE, 1), push(RK_AIR_CONST, it.bus)));
            for (uint32_t j = 0; j < it.nv; j++) acc = add(acc, scale(leaf(RK_AIR_CHALLENGE, 2 + j), push(RK_AIR_LOCAL, it.cols[j])));
            return acc;
        };
        auto signed_mult = [&](const Ix& it) {
            const uint32_t m = it.is_const ? push(RK_AIR_CONST, it.mult) : push(RK_AIR_LOCAL, it.mult);
            return it.kind == 0 ? m : push(RK_AIR_NEG, m);
        };
        std::vector<E> el, en;
        for (uint32_t b = 0; b < nb; b++) el.push_back(leaf(RK_AIR_PERM_LOCAL, b));
        for (uint32_t b = 0; b < nb; b++) en.push_back(leaf(RK_AIR_PERM_NEXT, b));
        for (uint32_t b = 0; b < nb; b++) {
            if (2 * b + 1 < n_lookups) {   // entry * rlc0 * rlc1 = m0 * rlc1 + m1 * rlc0
                const E r0 = rlc(its[2 * b]), r1 = rlc(its[2 * b + 1]);
                const E lhs = mul(mul(el[b], r0), r1);
                const E rhs = add(scale(r1, signed_mult(its[2 * b])), scale(r0, signed_mult(its[2 * b + 1])));
                assert_ext_zero(-1, sub(lhs, rhs));
            } else {                       // entry * rlc = m
                const E lhs = mul(el[b], rlc(its[2 * b]));
                assert_zero(push(RK_AIR_SUB, lhs[0], signed_mult(its[2 * b])));
                for (int k = 1; k < 4; k++) assert_zero(lhs[k]);
            }
        }
        const E phi_l = leaf(RK_AIR_PERM_LOCAL, nb), phi_n = leaf(RK_AIR_PERM_NEXT, nb);
        E sum_l = el[0], sum_n = en[0];
        for (uint32_t b = 1; b < nb; b++) sum_l = add(sum_l, el[b]), sum_n = add(sum_n, en[b]);
        assert_ext_zero((int32_t)push(RK_AIR_IS_FIRST_ROW), sub(phi_l, sum_l));
        assert_ext_zero((int32_t)push(RK_AIR_IS_TRANSITION), sub(sub(phi_n, phi_l), sum_n));
        assert_ext_zero((int32_t)push(RK_AIR_IS_LAST_ROW), sub(phi_l, leaf(RK_AIR_CUMSUM, 0)));
    }
};

// ---------------------------------------------------------------- the Poseidon2 chip (rk_p2_chip_*)
// One row = one permutation of the configured instance with the values a degree-3 AIR needs in columns -- the shape of
// sp1-recursion-core's Poseidon2 wide chip (RECALLED), the table a recursion / compress layer spends most of its rows
// on (every Merkle path step and sponge block of the proofs it verifies is one lookup into it):
//   in W | per external round r = 0..3: x3_r W (the cube of state + rc), post_r W (the state after the round) |
//   x3i_k R_P (cube of cell 0 + rc in internal round k) | s0_k R_P - 1 (cell 0 entering internal round k >= 1) |
//   int_out W (the state after the internal rounds) | external rounds 4..7 likewise | multiplicity
// x^7 = x3 * x3 * x keeps every constraint at degree 3; between commitments the state is carried as expressions.
using p3k::P2ChipLayout;
P2ChipLayout p2_chip_layout(const p2::Any& k) {
    P2ChipLayout L;
    L.W = (uint32_t)k.cells();
    L.RP = (uint32_t)k.rounds_partial();
    L.width = L.W + 16 * L.W + 2 * L.RP - 1 + L.W + 1;
    return L;
}
void p2_chip_steps(const p2::Any& k, std::vector<rk_air_step>& steps) {
    const P2ChipLayout L = p2_chip_layout(k);
    const uint32_t W = L.W;
    PermStepGen g{steps, 0, 0, {}};
    auto m_ext = [&](std::vector<uint32_t>& c) {   // the external layer on expressions: the 4x4 block on every four cells, then the sums of the cells four apart
        uint32_t sums[4] = {0, 0, 0, 0};
        for (uint32_t i = 0; i < W; i += 4) {
            const uint32_t a = c[i], b = c[i + 1], d = c[i + 2], e = c[i + 3];
            if (!k.m4()) {
                const uint32_t t0 = g.add(a, b), t1 = g.add(d, e), t2 = g.add(g.add(b, b), t1), t3 = g.add(g.add(e, e), t0);
                const uint32_t t1_4 = g.add(g.add(t1, t1), g.add(t1, t1)), t0_4 = g.add(g.add(t0, t0), g.add(t0, t0));
                const uint32_t t4 = g.add(t1_4, t3), t5 = g.add(t0_4, t2);
                c[i] = g.add(t3, t5), c[i + 1] = t5, c[i + 2] = g.add(t2, t4), c[i + 3] = t4;
            } else {
                const uint32_t s = g.add(g.add(a, b), g.add(d, e));
                c[i] = g.add(g.add(s, a), g.add(b, b));
                c[i + 1] = g.add(g.add(s, b), g.add(d, d));
                c[i + 2] = g.add(g.add(s, d), g.add(e, e));
                c[i + 3] = g.add(g.add(s, e), g.add(a, a));
            }
            for (int j = 0; j < 4; j++) sums[j] = i == 0 ? c[j] : g.add(sums[j], c[i + j]);
        }
        for (uint32_t i = 0; i < W; i++) c[i] = g.add(c[i], sums[i & 3]);
    };
    auto ext_round = [&](std::vector<uint32_t>& st, uint32_t r) {
        std::vector<uint32_t> x7(W);
        for (uint32_t i = 0; i < W; i++) {
            const uint32_t s = g.add(st[i], g.cst(bb::decode(k.rc_ext()[r * W + i]))), x3 = g.col(L.x3(r) + i);
            g.assert_zero(g.sub(x3, g.mul(g.mul(s, s), s)));
            x7[i] = g.mul(g.mul(x3, x3), s);
        }
        m_ext(x7);
        for (uint32_t i = 0; i < W; i++) {
            st[i] = g.col(L.post(r) + i);
            g.assert_zero(g.sub(st[i], x7[i]));
        }
    };
    std::vector<uint32_t> st(W);
    for (uint32_t i = 0; i < W; i++) st[i] = g.col(L.in() + i);
    m_ext(st);
    for (uint32_t r = 0; r < 4; r++) ext_round(st, r);
    for (uint32_t kk = 0; kk < L.RP; kk++) {
        uint32_t s0 = st[0];
        if (kk > 0) {
            s0 = g.col(L.s0(kk));
            g.assert_zero(g.sub(s0, st[0]));
        }
        const uint32_t t = g.add(s0, g.cst(bb::decode(k.rc_int()[kk]))), x3 = g.col(L.x3i(kk));
        g.assert_zero(g.sub(x3, g.mul(g.mul(t, t), t)));
        st[0] = g.mul(g.mul(x3, x3), t);
        uint32_t sum = st[0];
        for (uint32_t i = 1; i < W; i++) sum = g.add(sum, st[i]);
        for (uint32_t i = 0; i < W; i++) st[i] = g.add(sum, g.mul(st[i], g.cst(bb::decode(k.diag()[i]))));
    }
    for (uint32_t i = 0; i < W; i++) {
        const uint32_t c = g.col(L.int_out() + i);
        g.assert_zero(g.sub(c, st[i]));
        st[i] = c;
    }
    for (uint32_t r = 4; r < 8; r++) ext_round(st, r);
}

// the chip's rows on the GPU: one lane per permutation (p3k::chip_row, p3_kernels.hpp)
template <int W, int RP, int M4>
__global__ void __launch_bounds__(128) p2_chip_trace_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ in, const uint32_t* __restrict__ mult,
                                                            const uint32_t* __restrict__ tab, size_t n, P2ChipLayout L) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    p3k::chip_row<W, RP, M4>(out + r * L.width, in + r * W, mult ? mult[r] : bb::ONE, tab, L);
}

}  // namespace

extern "C" {

int rk_air_create(const rk_air_step* steps, size_t n_steps, uint32_t width, uint32_t n_public, rk_air** out) {
    return rk_air_create_lookup(steps, n_steps, width, n_public, nullptr, 0, 0, 0, out);
}
int rk_air_create_lookup(const rk_air_step* steps_in, size_t n_steps_in, uint32_t width, uint32_t n_public, const uint32_t* iw,
                         uint32_t n_interactions, size_t n_words, uint32_t ext_w, rk_air** out) {
    RK_GUARD_BEGIN
    if (!out) return RK_ERR_INVALID;
    *out = nullptr;
    const rk_air_step* steps = steps_in;
    size_t n_steps = n_steps_in;
    // a table that only takes part in lookups has no constraints of its own: an empty list is fine when the library writes the rest
    if (((!steps || n_steps == 0) && !(ext_w && n_steps == 0)) || n_steps > ((size_t)1 << 27) || width == 0 || width > (1u << 16) || n_public > (1u << 20)) return RK_ERR_INVALID;
    if (n_interactions > 4096 || (n_interactions && !iw) || (!n_interactions && n_words)) return RK_ERR_INVALID;
    if (ext_w >= bb::P || (ext_w && !n_interactions)) return RK_ERR_INVALID;
    std::unique_ptr<rk_air> air(new rk_air);
    {   // the interactions: kind, bus, mult_is_const, mult, n_values, columns...
        size_t at = 0;
        uint32_t max_values = 0;
        for (uint32_t i = 0; i < n_interactions; i++) {
            if (at + 5 > n_words) return RK_ERR_INVALID;
            const uint32_t kind = iw[at], bus = iw[at + 1], is_const = iw[at + 2], mult = iw[at + 3], nv = iw[at + 4];
            if (kind > 1 || bus >= bb::P || is_const > 1 || nv > 64 || at + 5 + nv > n_words) return RK_ERR_INVALID;
            if (is_const ? mult >= bb::P : mult >= width) return RK_ERR_INVALID;
            auto slot = [&](uint32_t col) {
                auto it = std::find(air->used.begin(), air->used.end(), col);
                if (it == air->used.end()) {
                    air->used.push_back(col);
                    return (uint32_t)air->used.size() - 1;
                }
                return (uint32_t)(it - air->used.begin());
            };
            air->lookups.insert(air->lookups.end(), {kind, bb::encode(bus), is_const, is_const ? bb::encode(mult) : slot(mult), nv});
            for (uint32_t j = 0; j < nv; j++) {
                if (iw[at + 5 + j] >= width) return RK_ERR_INVALID;
                air->lookups.push_back(slot(iw[at + 5 + j]));
            }
            max_values = std::max(max_values, nv);
            at += 5 + nv;
        }
        if (at != n_words) return RK_ERR_INVALID;
        if (air->used.size() > 120) return RK_ERR_INVALID;   // the staged tile (120 x 257 words of LDS)
        air->n_lookups = n_interactions;
        if (n_interactions) {
            air->perm_width = 4 * ((n_interactions + 1) / 2 + 1);
            air->n_chal = 4 * (max_values + 2);
        }
    }
    const uint32_t pw = air->perm_width, n_chal = air->n_chal;
    std::vector<rk_air_step> extended;
    if (ext_w) {   // the caller's list holds the main constraints only: append eval_permutation_constraints for x^4 - ext_w
        if (n_steps) RK_TRY(air_scan(steps, n_steps, width, n_public, 0, 0, &air->info));   // ... and may not name the permutation trace itself
        if (n_steps) extended.assign(steps, steps + n_steps);
        uint32_t nv = 0;
        for (const rk_air_step& st : extended) nv += st.op != RK_AIR_ASSERT_ZERO;
        PermStepGen gen{extended, nv, ext_w, {}};
        gen.run(iw, n_interactions);
        steps = extended.data();
        n_steps = extended.size();
    }
    RK_TRY(air_scan(steps, n_steps, width, n_public, pw, n_chal, &air->info));
    air->steps.assign(steps, steps + n_steps);
    air->width = width;
    air->n_public = n_public;
    // the list as an rk_program: taps 0..2 = the selector columns (group 0), 3 + c = LOCAL c, 3 + width + c = NEXT c
    // (group 2), then PERM_LOCAL / PERM_NEXT c (group 1); PUBLIC / CHALLENGE / CUMSUM = GET_GLOBAL of the proof's globals
    // (public values | challenges | cumulative sum); NEG a = 0 - a; the asserts one AND_EQZ chain
    std::vector<rk::Tap> taps;
    for (uint32_t c = 0; c < 3; c++) taps.push_back(rk::Tap{0, c, 0});
    for (uint32_t c = 0; c < width; c++) taps.push_back(rk::Tap{2, c, 0});
    for (uint32_t c = 0; c < width; c++) taps.push_back(rk::Tap{2, c, NEXT_BACK});
    for (uint32_t c = 0; c < pw; c++) taps.push_back(rk::Tap{1, c, 0});
    for (uint32_t c = 0; c < pw; c++) taps.push_back(rk::Tap{1, c, NEXT_BACK});
    std::vector<rk_poly_step> ps;
    ps.reserve(n_steps + 2);
    std::vector<uint32_t> fp_of;   // AIR value -> position in the program's field-value list
    fp_of.reserve(n_steps);
    uint32_t n_fp = 0, n_mx = 0, zero = rk::PROGRAM_NONE;
    ps.push_back(rk_poly_step{RK_STEP_TRUE, 0, 0, 0});
    uint32_t chain = n_mx++;
    for (size_t s = 0; s < n_steps; s++) {
        const rk_air_step& st = steps[s];
        switch (st.op) {
            case RK_AIR_CONST: ps.push_back(rk_poly_step{RK_STEP_CONST, st.a, 0, 0}); fp_of.push_back(n_fp++); break;
            case RK_AIR_LOCAL: ps.push_back(rk_poly_step{RK_STEP_GET, 3 + st.a, 0, 0}); fp_of.push_back(n_fp++); break;
            case RK_AIR_NEXT: ps.push_back(rk_poly_step{RK_STEP_GET, 3 + width + st.a, 0, 0}); fp_of.push_back(n_fp++); break;
            case RK_AIR_PUBLIC: ps.push_back(rk_poly_step{RK_STEP_GET_GLOBAL, 0, st.a, 0}); fp_of.push_back(n_fp++); break;
            case RK_AIR_PERM_LOCAL: ps.push_back(rk_poly_step{RK_STEP_GET, 3 + 2 * width + st.a, 0, 0}); fp_of.push_back(n_fp++); break;
            case RK_AIR_PERM_NEXT: ps.push_back(rk_poly_step{RK_STEP_GET, 3 + 2 * width + pw + st.a, 0, 0}); fp_of.push_back(n_fp++); break;
            case RK_AIR_CHALLENGE: ps.push_back(rk_poly_step{RK_STEP_GET_GLOBAL, 0, n_public + st.a, 0}); fp_of.push_back(n_fp++); break;
            case RK_AIR_CUMSUM: ps.push_back(rk_poly_step{RK_STEP_GET_GLOBAL, 0, n_public + n_chal + st.a, 0}); fp_of.push_back(n_fp++); break;
            case RK_AIR_IS_FIRST_ROW: case RK_AIR_IS_LAST_ROW: case RK_AIR_IS_TRANSITION:
                air->sel_mask |= 1u << (st.op - RK_AIR_IS_FIRST_ROW);
                ps.push_back(rk_poly_step{RK_STEP_GET, st.op - RK_AIR_IS_FIRST_ROW, 0, 0});
                fp_of.push_back(n_fp++);
                break;
            case RK_AIR_ADD: ps.push_back(rk_poly_step{RK_STEP_ADD, fp_of[st.a], fp_of[st.b], 0}); fp_of.push_back(n_fp++); break;
            case RK_AIR_SUB: ps.push_back(rk_poly_step{RK_STEP_SUB, fp_of[st.a], fp_of[st.b], 0}); fp_of.push_back(n_fp++); break;
            case RK_AIR_MUL: ps.push_back(rk_poly_step{RK_STEP_MUL, fp_of[st.a], fp_of[st.b], 0}); fp_of.push_back(n_fp++); break;
            case RK_AIR_NEG:
                if (zero == rk::PROGRAM_NONE) {
                    ps.push_back(rk_poly_step{RK_STEP_CONST, 0, 0, 0});
                    zero = n_fp++;
                }
                ps.push_back(rk_poly_step{RK_STEP_SUB, zero, fp_of[st.a], 0});
                fp_of.push_back(n_fp++);
                break;
            default:  // ASSERT_ZERO
                ps.push_back(rk_poly_step{RK_STEP_AND_EQZ, chain, fp_of[st.a], 0});
                chain = n_mx++;
                break;
        }
    }
    RK_TRY(rk::program_create_raw(ps.data(), ps.size(), chain, std::move(taps), /*horner=*/true, &air->prog));
    rk_program_info pi;
    (void)rk_program_get_info(air->prog, &pi);
    air->info.n_ops = pi.n_ops;
    air->info.n_fp_slots = pi.n_fp_slots;
    *out = air.release();
    return RK_OK;
    RK_GUARD_END
}
uint32_t rk_p2_chip_width(const rk_params* params) {
    rk_params def;
    rk::params_preset(&def, RK_PRESET_SP1);
    rk::Sys sys;
    auto k = std::make_unique<p2::Any>();
    if (rk::resolve_params(params ? params : &def, &sys, k.get()) != RK_OK) return 0;
    return p2_chip_layout(*k).width;
}
int rk_p2_chip_air(const rk_params* params, uint32_t bus, rk_air** out) {
    RK_GUARD_BEGIN
    if (!out) return RK_ERR_INVALID;
    *out = nullptr;
    rk_params def;
    rk::params_preset(&def, RK_PRESET_SP1);
    const rk_params& par = params ? *params : def;
    rk::Sys sys;
    auto k = std::make_unique<p2::Any>();
    RK_TRY(rk::resolve_params(&par, &sys, k.get()));
    if (bus >= bb::P) return RK_ERR_INVALID;
    const P2ChipLayout L = p2_chip_layout(*k);
    std::vector<rk_air_step> steps;
    p2_chip_steps(*k, steps);
    // receives (bus: in[0..W), out[0..8)) `multiplicity` times per row
    std::vector<uint32_t> ix = {1, bus, 0, L.mult(), L.W + (uint32_t)p2::OUT};
    for (uint32_t i = 0; i < L.W; i++) ix.push_back(L.in() + i);
    for (uint32_t i = 0; i < (uint32_t)p2::OUT; i++) ix.push_back(L.out() + i);
    return rk_air_create_lookup(steps.data(), steps.size(), L.width, 0, ix.data(), 1, ix.size(), par.ext_w, out);
    RK_GUARD_END
}
int rk_p2_chip_trace(rk_ctx* ctx, const uint32_t* d_inputs, const uint32_t* d_mult, size_t n, uint32_t* d_trace) {
    RK_GUARD_BEGIN
    if (!ctx || !d_inputs || !d_trace || n == 0 || n > ((size_t)1 << 24)) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const p2::Any& k = ctx->h_p2;
    const P2ChipLayout L = p2_chip_layout(k);
    std::vector<uint32_t> tab(k.rc_ext(), k.rc_ext() + 8 * L.W);
    tab.insert(tab.end(), k.rc_int(), k.rc_int() + L.RP);
    tab.insert(tab.end(), k.diag(), k.diag() + L.W);
    DevBuf d_tab;
    RK_TRY(d_tab.alloc(ctx, tab.size() * 4));
    RK_TRY(rk::upload(ctx, d_tab.p, tab.data(), tab.size() * 4));
    const dim3 grid((unsigned)((n + 127) / 128)), block(128);
    switch (k.kind) {
        case 0: hipLaunchKernelGGL((p2_chip_trace_kernel<24, 21, 0>), grid, block, 0, ctx->stream, d_trace, d_inputs, d_mult, (const uint32_t*)d_tab.u32(), n, L); break;
        case 1: hipLaunchKernelGGL((p2_chip_trace_kernel<24, 21, 1>), grid, block, 0, ctx->stream, d_trace, d_inputs, d_mult, (const uint32_t*)d_tab.u32(), n, L); break;
        case 2: hipLaunchKernelGGL((p2_chip_trace_kernel<16, 13, 0>), grid, block, 0, ctx->stream, d_trace, d_inputs, d_mult, (const uint32_t*)d_tab.u32(), n, L); break;
        default: hipLaunchKernelGGL((p2_chip_trace_kernel<16, 13, 1>), grid, block, 0, ctx->stream, d_trace, d_inputs, d_mult, (const uint32_t*)d_tab.u32(), n, L); break;
    }
    return rk::post_launch(ctx, "p2_chip_trace_kernel");
    RK_GUARD_END
}
int rk_air_get_steps(const rk_air* air, rk_air_step* out, size_t capacity, size_t* n_steps) {
    if (!air || !n_steps) return RK_ERR_INVALID;
    *n_steps = air->steps.size();
    if (!out || capacity < air->steps.size()) return RK_ERR_CAPACITY;
    std::memcpy(out, air->steps.data(), air->steps.size() * sizeof(rk_air_step));
    return RK_OK;
}
int rk_air_destroy(rk_air* air) {
    RK_GUARD_BEGIN
    if (!air) return RK_OK;
    (void)rk_program_destroy(air->prog);
    delete air;
    return RK_OK;
    RK_GUARD_END
}
int rk_air_get_info(const rk_air* air, rk_air_info* out) {
    if (!air || !out) return RK_ERR_INVALID;
    *out = air->info;
    return RK_OK;
}
int rk_air_compile(rk_air* air, rk_ctx* ctx) {
    RK_GUARD_BEGIN
    if (!air || !ctx) return RK_ERR_INVALID;
    return rk_program_compile(air->prog, ctx);
    RK_GUARD_END
}

}  // extern "C"
