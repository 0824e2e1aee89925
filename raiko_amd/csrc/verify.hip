// Host-side seal verifier: the counterpart of `receipt.verify(image_id)` that the reference
// calls after proving (provers/risc0/driver/src/lib.rs:136, benchmark.rs:19, bonsai.rs:67),
// restating risc0-zkp 1.0.1 verify/{mod,fri,merkle,read_iop}.rs for the flow of
// rk_prove_segment.  Pure CPU code (no GPU needed): a host can check seals produced elsewhere.
//
// Checked: transcript binding, Merkle openings, the DEEP quotient at every query, FRI folds, the
// final low-degree polynomial and -- when the caller supplies the circuit's `poly_ext`
// (rk_verify_opts) -- the constraint identity on the tap openings.  The rv32im constraint system
// itself (risc0-circuit-rv32im) is not in this repo; examples/toy_circuit shows a complete one.
#include "internal.hpp"

#include <cstring>
#include <memory>

namespace {

using bb::Ext;

// the compiled-in risc0 parameter set (what rk_verify_segment assumes)
struct Defaults {
    rk::Sys sys;
    p2::Any p2any;
    Defaults() {
        rk_params def;
        rk::params_preset(&def, RK_PRESET_RISC0);
        (void)rk::resolve_params(&def, &sys, &p2any);
    }
};
const Defaults& defaults() {
    static Defaults d;
    return d;
}
using Sponge = p2::Rng;

struct Reader {
    const uint32_t* p;
    size_t len, pos = 0;
    bool short_read = false;
    void read(uint32_t* out, size_t n) {
        if (pos + n > len) {
            short_read = true;
            std::memset(out, 0, n * 4);
            return;
        }
        std::memcpy(out, p + pos, n * 4);
        pos += n;
    }
};

struct TreeVerifier {  // MerkleTreeVerifier
    size_t rows = 0, cols = 0, top_size = 1;
    std::vector<uint32_t> top;  // heap, index 1 = root
    void init(const p2::Any& k, Reader& r, Sponge& rng, size_t rows_, size_t cols_, size_t queries) {
        rows = rows_;
        cols = cols_;
        size_t layers = log2u(rows), top_layer = 0;
        for (size_t i = 1; i < layers; i++) {
            if (((size_t)1 << i) > queries) break;
            top_layer = i;
        }
        top_size = (size_t)1 << top_layer;
        top.assign(2 * top_size * 8, 0);
        r.read(top.data() + top_size * 8, top_size * 8);
        for (size_t i = top_size; i-- > 1;) k.hash_pair(&top[2 * i * 8], &top[(2 * i + 1) * 8], &top[i * 8]);
        rng.mix(&top[8]);
    }
    bool open(const p2::Any& k, Reader& r, size_t idx, uint32_t* row) const {
        if (idx >= rows) return false;
        r.read(row, cols);
        uint32_t cur[8], other[8], nxt[8];
        k.hash_elems(row, cols, cur);
        idx += rows;
        while (idx >= 2 * top_size) {
            bool right = idx & 1;
            r.read(other, 8);
            idx >>= 1;
            if (right) k.hash_pair(other, cur, nxt);
            else k.hash_pair(cur, other, nxt);
            std::memcpy(cur, nxt, 32);
        }
        return std::memcmp(cur, &top[idx * 8], 32) == 0;
    }
};

Ext poly_eval(const Ext* c, size_t n, const Ext& x, uint32_t wm) {
    Ext acc = bb::ext_zero();
    for (size_t i = n; i-- > 0;) acc = bb::add(bb::mul(acc, x, wm), c[i]);
    return acc;
}

// 0: the seal is a valid proof for the public data of `pub` (po2, taps, globals, infos);
// RK_ERR_INVALID: malformed arguments; otherwise a positive reason code:
//   10 header mismatch, 2x group opening failed (x = group id, 3 = check), 3x FRI round opening,
//   4x fold inconsistency, 50 final polynomial mismatch, 60 seal too short, 61 trailing words,
//   62 proof of work, 63 a seal word that is not a canonical field element (>= p),
//   70 constraint identity (only with opts->poly_ext), 71 poly_ext callback failed
int verify_segment(const rk_segment* pub, const rk_verify_opts* opts, const uint32_t* seal, size_t seal_words) {
    if (!pub || !seal) return RK_ERR_INVALID;
    const rk_taps& taps = pub->taps;
    if (rk::check_taps(taps) != RK_OK) return RK_ERR_INVALID;
    if (pub->n_globals && !pub->globals) return RK_ERR_INVALID;
    if (pub->n_accum_mix > (1u << 16)) return RK_ERR_INVALID;
    // parameter set: opts->params (the whole blob), else the three width-24 tables of ABI 1, else the defaults
    rk::Sys sys = defaults().sys;
    auto custom = std::make_unique<p2::Any>();
    const p2::Any* kp = &defaults().p2any;
    if (opts && opts->params) {
        if (rk::resolve_params(opts->params, &sys, custom.get()) != RK_OK) return RK_ERR_INVALID;
        kp = custom.get();
    } else if (opts && (opts->p2_rc_ext || opts->p2_rc_int || opts->p2_diag)) {
        if (!opts->p2_rc_ext || !opts->p2_rc_int || !opts->p2_diag) return RK_ERR_INVALID;
        rk_params pp;
        rk::params_preset(&pp, RK_PRESET_RISC0);
        pp.p2_rc_ext = opts->p2_rc_ext;
        pp.p2_rc_int = opts->p2_rc_int;
        pp.p2_diag = opts->p2_diag;
        if (rk::resolve_params(&pp, &sys, custom.get()) != RK_OK) return RK_ERR_INVALID;
        kp = custom.get();
    }
    const rk::Shape shape = sys.shape();
    if (!rk::shape_ok(shape) || pub->po2 < 1 || pub->po2 + shape.blowup_log2 > ntt::LAMBDA) return RK_ERR_INVALID;
    const p2::Any& k = *kp;
    const uint32_t wm = sys.wm;
    const unsigned BLOW = shape.blowup_log2, FOLD_LOG = shape.fold_log2;
    const size_t QUERIES = shape.queries, FOLD = (size_t)1 << FOLD_LOG, MIN_DEGREE = shape.min_degree, CHECK = (size_t)4 << BLOW;
    // Every word of a seal is a field element in Montgomery form (values, digests, the nonce) or the small integer
    // po2: the arithmetic below (bb::add / sub / mont_reduce) is arithmetic mod p only for operands < p, so a seal
    // from elsewhere carrying a + p in place of a is refused before anything is computed from it (risc0's read_iop
    // rejects invalid elements the same way; rk_mmcs_verify does it per opened row)
    for (size_t i = 0; i < seal_words; i++)
        if (seal[i] >= bb::P) return 63;
    Reader r{seal, seal_words};
    Sponge rng(&k);
    uint32_t digest[8], e16[16];
    for (int i = 0; i < 16; i++) e16[i] = bb::encode(pub->proof_system_info[i]);
    k.hash_elems(e16, 16, digest);
    rng.mix(digest);
    for (int i = 0; i < 16; i++) e16[i] = bb::encode(pub->circuit_info[i]);
    k.hash_elems(e16, 16, digest);
    rng.mix(digest);

    std::vector<uint32_t> io(pub->n_globals + 1);
    r.read(io.data(), pub->n_globals);
    uint32_t po2 = 0;
    r.read(&po2, 1);
    if (r.short_read || po2 != pub->po2) return 10;
    if (pub->n_globals && std::memcmp(io.data(), pub->globals, pub->n_globals * 4) != 0) return 10;
    io[pub->n_globals] = bb::encode(po2);
    k.hash_elems(io.data(), io.size(), digest);
    rng.mix(digest);

    const size_t N = (size_t)1 << po2, D = N << BLOW;
    TreeVerifier tg[3], tcheck;
    tg[1].init(k, r, rng, D, taps.group_size[1], QUERIES);
    tg[2].init(k, r, rng, D, taps.group_size[2], QUERIES);
    std::vector<uint32_t> accum_mix(pub->n_accum_mix);
    for (uint32_t i = 0; i < pub->n_accum_mix; i++) accum_mix[i] = rng.random_elem();
    tg[0].init(k, r, rng, D, taps.group_size[0], QUERIES);
    const Ext poly_mix = rng.random_ext();
    tcheck.init(k, r, rng, D, CHECK, QUERIES);
    const Ext z = rng.random_ext();
    const uint32_t w27 = sys.root27m;
    const uint32_t back_one = bb::inv(bb::pow(w27, (uint64_t)1 << (27 - po2)));

    size_t tot_taps = 0;
    const size_t tot_backs = taps.combo_off[taps.n_combos];
    for (uint32_t i = 0; i < taps.n_regs; i++)
        tot_taps += taps.combo_off[taps.reg_combo[i] + 1] - taps.combo_off[taps.reg_combo[i]];
    std::vector<Ext> coeff_u(tot_taps + CHECK);
    r.read((uint32_t*)coeff_u.data(), coeff_u.size() * 4);
    k.hash_elems((const uint32_t*)coeff_u.data(), coeff_u.size() * 4, digest);
    rng.mix(digest);
    if (r.short_read) return 60;
    if (opts && (opts->poly_ext || opts->program)) {
        // verify/mod.rs: U polynomials back to evaluation form, the circuit's mixed constraint
        // polynomial on them, against check(z) * ((3z)^N - 1) with check(z) = sum_i z^i * g_i(z^4),
        // g_i = the extension element whose component e is opened in check column 4e + remap[i]
        std::vector<Ext> eval_u(tot_taps);
        size_t pos = 0;
        for (uint32_t i = 0; i < taps.n_regs; i++) {
            uint32_t cb = taps.reg_combo[i];
            size_t sz = taps.combo_off[cb + 1] - taps.combo_off[cb];
            for (size_t j = 0; j < sz; j++) {
                Ext x = bb::scale(z, bb::pow(back_one, taps.combo_backs[taps.combo_off[cb] + j]));
                eval_u[pos + j] = poly_eval(&coeff_u[pos], sz, x, wm);
            }
            pos += sz;
        }
        Ext result;
        if (opts->poly_ext) {
            if (opts->poly_ext(opts->user, pub, poly_mix.c, (const uint32_t*)eval_u.data(), tot_taps, accum_mix.data(),
                               pub->n_accum_mix, result.c) != 0)
                return 71;
        } else if (rk::program_poly_ext(opts->program, wm, poly_mix.c, (const uint32_t*)eval_u.data(), tot_taps, pub->globals,
                                        pub->n_globals, accum_mix.data(), pub->n_accum_mix, result.c) != RK_OK) {
            return 71;
        }
        // part j of the check polynomial sits in column bitrev(j) of each component ([0,2,1,3] for blow-up 4)
        const size_t parts = (size_t)1 << BLOW;
        Ext check = bb::ext_zero(), zi = bb::ext_one();
        for (size_t i = 0; i < parts; i++) {
            for (int e = 0; e < 4; e++) {
                Ext basis = bb::ext_zero();
                basis.c[e] = bb::ONE;
                check = bb::add(check, bb::mul(bb::mul(coeff_u[tot_taps + bb::bitrev((uint32_t)i, BLOW) + parts * e], zi, wm), basis, wm));
            }
            zi = bb::mul(zi, z, wm);
        }
        Ext vanish = bb::sub(bb::pow(bb::scale(z, sys.shiftm), N, wm), bb::ext_one());
        if (!bb::eq(bb::mul(check, vanish, wm), result)) return 70;
    }
    const Ext mix = rng.random_ext();
    std::vector<Ext> combo_u(tot_backs + 1, bb::ext_zero());
    {
        Ext cur = bb::ext_one();
        size_t pos = 0;
        for (uint32_t i = 0; i < taps.n_regs; i++) {
            uint32_t cb = taps.reg_combo[i];
            size_t sz = taps.combo_off[cb + 1] - taps.combo_off[cb];
            for (size_t j = 0; j < sz; j++)
                combo_u[taps.combo_off[cb] + j] = bb::add(combo_u[taps.combo_off[cb] + j], bb::mul(cur, coeff_u[pos + j], wm));
            cur = bb::mul(cur, mix, wm);
            pos += sz;
        }
        for (size_t i = 0; i < CHECK; i++) {
            combo_u[tot_backs] = bb::add(combo_u[tot_backs], bb::mul(cur, coeff_u[pos++], wm));
            cur = bb::mul(cur, mix, wm);
        }
    }
    const Ext z_pow = bb::pow(z, (uint64_t)1 << BLOW, wm);

    // FRI commitments
    struct Round {
        size_t domain;
        TreeVerifier tree;
        Ext mix;
    };
    std::vector<Round> rounds;
    size_t degree = N, domain = D;
    while (degree > MIN_DEGREE && degree >= FOLD) {
        rounds.emplace_back();
        Round& rd = rounds.back();
        rd.domain = domain;
        rd.tree.init(k, r, rng, domain / FOLD, FOLD * 4, QUERIES);
        rd.mix = rng.random_ext();
        domain /= FOLD;
        degree /= FOLD;
    }
    std::vector<uint32_t> final_coeffs(4 * degree);
    r.read(final_coeffs.data(), final_coeffs.size());
    k.hash_elems(final_coeffs.data(), final_coeffs.size(), digest);
    rng.mix(digest);
    if (r.short_read) return 60;
    if (shape.pow_bits) {  // proof of work: the nonce, absorbed, must zero the next pow_bits random bits (reason 62)
        uint32_t nonce = 0;
        r.read(&nonce, 1);
        if (r.short_read) return 60;
        if (nonce >= bb::P) return 62;
        k.hash_elems(&nonce, 1, digest);
        rng.mix(digest);
        if (rng.random_bits(shape.pow_bits) != 0) return 62;
    }

    const uint32_t gen0 = bb::pow(w27, (uint64_t)1 << (27 - log2u(D)));
    const uint32_t gen_final = bb::pow(w27, (uint64_t)1 << (27 - log2u(domain)));
    const uint32_t w16_inv = bb::inv(bb::pow(w27, (uint64_t)1 << (27 - FOLD_LOG)));   // inverse of the FOLD-th root
    const uint32_t inv16 = bb::inv(bb::encode((uint32_t)FOLD));
    std::vector<uint32_t> row[3];
    for (int g = 0; g < 3; g++) row[g].resize(taps.group_size[g] + 1);
    uint32_t check_row[64];
    std::vector<Ext> tot(taps.n_combos + 1);

    for (size_t q = 0; q < QUERIES; q++) {
        size_t pos = rng.random_bits(log2u(D)) % D;
        const Ext x = bb::ext_from(bb::pow(gen0, pos));
        for (int g = 0; g < 3; g++)
            if (!tg[g].open(k, r, pos, row[g].data())) return r.short_read ? 60 : 20 + g;
        if (!tcheck.open(k, r, pos, check_row)) return r.short_read ? 60 : 23;
        for (auto& t : tot) t = bb::ext_zero();
        Ext cur = bb::ext_one();
        for (uint32_t i = 0; i < taps.n_regs; i++) {
            uint32_t v = row[taps.reg_group[i]][taps.reg_offset[i]];
            tot[taps.reg_combo[i]] = bb::add(tot[taps.reg_combo[i]], bb::scale(cur, v));
            cur = bb::mul(cur, mix, wm);
        }
        for (size_t i = 0; i < CHECK; i++) {
            tot[taps.n_combos] = bb::add(tot[taps.n_combos], bb::scale(cur, check_row[i]));
            cur = bb::mul(cur, mix, wm);
        }
        Ext goal = bb::ext_zero();
        for (uint32_t c = 0; c < taps.n_combos; c++) {
            size_t b0 = taps.combo_off[c], b1 = taps.combo_off[c + 1];
            Ext num = bb::sub(tot[c], poly_eval(&combo_u[b0], b1 - b0, x, wm));
            Ext den = bb::ext_one();
            for (size_t b = b0; b < b1; b++)
                den = bb::mul(den, bb::sub(x, bb::scale(z, bb::pow(back_one, taps.combo_backs[b]))), wm);
            goal = bb::add(goal, bb::mul(num, bb::inv(den, wm), wm));
        }
        goal = bb::add(goal, bb::mul(bb::sub(tot[taps.n_combos], combo_u[tot_backs]), bb::inv(bb::sub(x, z_pow), wm), wm));

        for (size_t kr = 0; kr < rounds.size(); kr++) {
            const Round& rd = rounds[kr];
            size_t rows = rd.domain / FOLD;
            size_t quot = pos / rows, group = pos % rows;
            uint32_t data[64];
            if (!rd.tree.open(k, r, group, data)) return r.short_read ? 60 : 30 + (int)(kr < 9 ? kr : 9);
            Ext de[16];
            for (size_t i = 0; i < FOLD; i++)
                for (int c = 0; c < 4; c++) de[i].c[c] = data[c * FOLD + i];
            if (!bb::eq(de[quot], goal)) return 40 + (int)(kr < 9 ? kr : 9);
            // interpolate the FOLD coset values and evaluate at mix * w^-group
            Ext co[16];
            for (size_t i = 0; i < FOLD; i++) {
                Ext acc = bb::ext_zero();
                for (size_t j = 0; j < FOLD; j++) acc = bb::add(acc, bb::scale(de[j], bb::pow(w16_inv, (uint64_t)((i * j) & (FOLD - 1)))));
                co[i] = bb::scale(acc, inv16);
            }
            uint32_t inv_wk = bb::pow(bb::inv(bb::pow(w27, (uint64_t)1 << (27 - log2u(rd.domain)))), group);
            goal = poly_eval(co, FOLD, bb::scale(rd.mix, inv_wk), wm);
            pos = group;
        }
        const Ext xf = bb::ext_from(bb::pow(gen_final, pos));
        Ext fx = bb::ext_zero();
        for (size_t i = degree; i-- > 0;) {
            Ext c{{final_coeffs[i], final_coeffs[degree + i], final_coeffs[2 * degree + i], final_coeffs[3 * degree + i]}};
            fx = bb::add(bb::mul(fx, xf, wm), c);
        }
        if (!bb::eq(fx, goal)) return 50;
    }
    if (r.short_read) return 60;
    if (r.pos != r.len) return 61;
    return RK_OK;
}

}  // namespace

extern "C" {

int rk_verify_segment_ex(const rk_segment* pub, const rk_verify_opts* opts, const uint32_t* seal, size_t seal_words) {
    RK_GUARD_BEGIN
    return verify_segment(pub, opts, seal, seal_words);
    RK_GUARD_END
}
// the compiled-in Poseidon2 instance, no constraint identity
int rk_verify_segment(const rk_segment* pub, const uint32_t* seal, size_t seal_words) {
    return rk_verify_segment_ex(pub, nullptr, seal, seal_words);
}

}  // extern "C"
