// Whole-segment STARK prover on one MI355X: the host-side C++ mirror of
// risc0-zkp 1.0.1 prove/{prover,poly_group,merkle,write_iop,fri}.rs and the
// prove_segment driver of risc0-circuit-rv32im 1.0.1 -- the work behind
// `session.prove()` at reference provers/risc0/driver/src/bonsai.rs:271.
//
// Everything sized O(trace) stays in HBM for the whole proof; the host only
// sees transcript-sized data (Merkle caps, tap openings, the final FRI
// polynomial, query openings).  The Fiat-Shamir transcript forces a handful
// of stream synchronisations per segment (one per commitment).
//
// Circuit-specific steps are the caller's: witness generation hands in code and data, and the
// two steps that depend on Fiat-Shamir randomness (accum construction after the accum mix is
// drawn, eval_check after poly_mix is drawn) are rk_circuit_hooks called back from the proof
// (CircuitHal::accumulate / eval_check).  Without hooks their outputs are taken as given
// (the synthetic S20 stand-in of SURVEY.md section 8d).
#include "internal.hpp"

#include <algorithm>
#include <cstring>
#include <memory>

namespace {

using bb::Ext;

// ---------------------------------------------------------------- transcript
// WriteIOP + Poseidon2Rng (risc0-zkp prove/write_iop.rs, core/hash/poseidon2/rng.rs) over the
// context's Poseidon2 instance (poseidon2_any.hpp)
struct Transcript {
    p2::Rng rng;
    std::vector<uint32_t> proof;
    explicit Transcript(const p2::Any* kc) : rng(kc) {}
    void write(const uint32_t* w, size_t n) { proof.insert(proof.end(), w, w + n); }
    void commit(const uint32_t* digest) { rng.mix(digest); }
    uint32_t random_elem() { return rng.random_elem(); }
    Ext random_ext() { return rng.random_ext(); }
    uint32_t random_bits(unsigned bits) { return rng.random_bits(bits); }
};

// ---------------------------------------------------------------- device helpers
struct DevBuf {
    rk_ctx* ctx = nullptr;
    void* p = nullptr;
    bool borrowed = false;  // caller-owned memory used in place (rk_segment.on_device == 2)
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    int alloc(rk_ctx* c, size_t bytes) {
        release();
        ctx = c;
        return rk::dev_alloc(c, bytes, &p);
    }
    void adopt(rk_ctx* c, void* ptr) {
        release();
        ctx = c;
        p = ptr;
        borrowed = true;
    }
    void release() {
        if (p && !borrowed) rk::dev_free(ctx, p);
        p = nullptr;
        borrowed = false;
    }
    uint32_t* u32() const { return (uint32_t*)p; }
};

int d2h_sync(rk_ctx* ctx, void* h, const void* d, size_t bytes) {
    RK_HIP_TRY(ctx, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RK_OK;
}
int h2d_sync(rk_ctx* ctx, void* d, const void* h, size_t bytes) {
    RK_HIP_TRY(ctx, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RK_OK;
}

// ---------------------------------------------------------------- Merkle prover
struct MerkleDev {
    size_t rows = 0, cols = 0, layers = 0, top_layer = 0, top_size = 1;
    DevBuf nodes;
    const uint32_t* matrix = nullptr;  // device, column-major rows x cols
    std::vector<uint32_t> top;         // host copy of nodes[1 .. 2*top_size)

    int build(rk_ctx* ctx, const uint32_t* d_matrix, size_t r, size_t c, size_t queries) {
        rows = r;
        cols = c;
        matrix = d_matrix;
        layers = log2u(r);
        top_layer = 0;
        for (size_t i = 1; i < layers; i++) {
            if (((size_t)1 << i) > queries) break;
            top_layer = i;
        }
        top_size = (size_t)1 << top_layer;
        RK_TRY(nodes.alloc(ctx, 2 * rows * p2::OUT * 4));
        return rk::merkle_build(ctx, nodes.u32(), d_matrix, rows, cols);
    }
    // MerkleTreeProver::commit: send the top layer, absorb the root
    int commit(rk_ctx* ctx, Transcript& iop) {
        top.assign(2 * top_size * p2::OUT, 0);
        // nodes[1 .. 2*top_size) in one copy (index 0 unused)
        RK_TRY(d2h_sync(ctx, top.data() + p2::OUT, nodes.u32() + p2::OUT, (2 * top_size - 1) * p2::OUT * 4));
        iop.write(top.data() + top_size * p2::OUT, top_size * p2::OUT);
        iop.commit(top.data() + p2::OUT);
        return RK_OK;
    }
    size_t path_len() const { return layers - top_layer; }
};

// Openings of several trees at n query positions each: one upload of every index, two gathers per tree, one
// download of everything (a proof opens 4 + FRI-round trees: one host round trip instead of two per tree).
struct Openings {
    std::vector<uint32_t> rows;   // n x cols
    std::vector<uint32_t> paths;  // n x path_len x 8
};
struct OpenJob {
    const MerkleDev* m;
    std::vector<uint32_t> pos;
    Openings* out;
};
int open_many(rk_ctx* ctx, std::vector<OpenJob>& jobs) {
    size_t idx_words = 0, out_words = 0;
    for (const OpenJob& j : jobs) {
        const size_t n = j.pos.size(), pl = j.m->path_len();
        idx_words += n + n * pl;
        out_words += n * j.m->cols + n * pl * p2::OUT;
    }
    if (out_words == 0) return RK_OK;
    std::vector<uint32_t> h_idx(idx_words), h_out(out_words);
    size_t at = 0;
    for (const OpenJob& j : jobs) {
        const MerkleDev& m = *j.m;
        const size_t n = j.pos.size(), pl = m.path_len();
        std::copy(j.pos.begin(), j.pos.end(), h_idx.begin() + at);
        uint32_t* node_idx = h_idx.data() + at + n;
        for (size_t q = 0; q < n; q++) {
            size_t idx = j.pos[q] + m.rows, k = 0;
            while (idx >= 2 * m.top_size) {
                size_t low = idx & 1;
                idx >>= 1;
                node_idx[q * pl + k++] = (uint32_t)(2 * idx + (1 - low));
            }
        }
        at += n + n * pl;
    }
    // [indices | job table]: one upload
    const size_t jobs_at = (idx_words + 3) & ~(size_t)3;
    std::vector<uint32_t> pack(jobs_at + jobs.size() * (sizeof(rk::GatherJob) / 4));
    std::copy(h_idx.begin(), h_idx.end(), pack.begin());
    std::vector<rk::GatherJob> h_jobs(jobs.size());
    size_t ia = 0, oa = 0;
    for (size_t k = 0; k < jobs.size(); k++) {
        const MerkleDev& m = *jobs[k].m;
        const size_t n = jobs[k].pos.size(), pl = m.path_len();
        h_jobs[k] = rk::GatherJob{m.matrix, m.nodes.u32(), m.rows, m.cols, ia, oa, (uint32_t)n, (uint32_t)pl};
        ia += n + n * pl;
        oa += n * m.cols + n * pl * p2::OUT;
    }
    std::memcpy(&pack[jobs_at], h_jobs.data(), h_jobs.size() * sizeof(rk::GatherJob));
    DevBuf d_idx, d_out;
    RK_TRY(d_idx.alloc(ctx, pack.size() * 4));
    RK_TRY(d_out.alloc(ctx, out_words * 4 + 16));
    RK_TRY(rk::upload(ctx, d_idx.p, pack.data(), pack.size() * 4));
    RK_TRY(rk::gather_many(ctx, d_out.u32(), d_idx.u32(), (const rk::GatherJob*)(d_idx.u32() + jobs_at), h_jobs.data(), jobs.size()));
    RK_TRY(d2h_sync(ctx, h_out.data(), d_out.u32(), out_words * 4));
    oa = 0;
    for (OpenJob& j : jobs) {
        const size_t n = j.pos.size(), pl = j.m->path_len(), rw = n * j.m->cols, pw = n * pl * p2::OUT;
        j.out->rows.assign(h_out.begin() + oa, h_out.begin() + oa + rw);
        j.out->paths.assign(h_out.begin() + oa + rw, h_out.begin() + oa + rw + pw);
        oa += rw + pw;
    }
    return RK_OK;
}
void write_opening(Transcript& iop, const MerkleDev& m, const Openings& o, size_t q) {
    iop.write(o.rows.data() + q * m.cols, m.cols);
    size_t pl = m.path_len();
    iop.write(o.paths.data() + q * pl * p2::OUT, pl * p2::OUT);
}

// ---------------------------------------------------------------- PolyGroup
struct PolyGroup {
    DevBuf coeffs;     // count x size, natural order once built
    DevBuf evaluated;  // count x size << blowup_log2
    size_t count = 0, size = 0;
    MerkleDev merkle;
    // coeffs must hold interpolated, zk-shifted, bit-reversed coefficients
    int build(rk_ctx* ctx, size_t cnt, size_t sz) {
        count = cnt;
        size = sz;
        const unsigned blow = ctx->sys.blowup_log2;
        size_t domain = sz << blow;
        RK_TRY(evaluated.alloc(ctx, cnt * domain * 4));
        RK_TRY(rk::ntt_forward(ctx, evaluated.u32(), coeffs.u32(), sz, cnt, blow));
        // risc0 bit-reverses the coefficients here (PolyGroup::new) because its later users
        // index them in natural order.  The device pipeline instead keeps them bit-reversed:
        // the tap evaluation multiplies by power tables stored in the same order and the
        // DEEP mix is pointwise, so only the (combo_count + 1) mixed polynomials are permuted
        // afterwards -- 1/50th of the data for the same field results.
        return RK_OK;
    }
};

// Per-stage device time without stalling the proof: every stage is bracketed by two events taken from a pool
// the context keeps, and the brackets are read once the proof has finished (a stop that waited for its event
// made the host notice every stage boundary before it could queue the next stage: ~11 bubbles per proof, which
// is what small segments are made of).  With RK_ROCTX=1 a stage is also a roctx range and does wait, so that
// the range covers the device work.
struct StageClock {
    rk_ctx* ctx;
    struct Bracket {
        hipEvent_t a, b;
        float* acc;
    };
    std::vector<Bracket> used;
    size_t open = (size_t)-1;
    bool ranged = false;
    explicit StageClock(rk_ctx* c) : ctx(c) {}
    void start(const char* stage = nullptr) {
        ranged = stage && rk::trace_push(stage);
        if (ctx->stage_events.size() < 2 * (used.size() + 1)) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
                if (e0) (void)hipEventDestroy(e0);
                open = (size_t)-1;
                return;
            }
            ctx->stage_events.push_back(e0);
            ctx->stage_events.push_back(e1);
        }
        Bracket br{ctx->stage_events[2 * used.size()], ctx->stage_events[2 * used.size() + 1], nullptr};
        (void)hipEventRecord(br.a, ctx->stream);
        used.push_back(br);
        open = used.size() - 1;
    }
    void stop(float* acc) {
        if (open != (size_t)-1) {
            used[open].acc = acc;
            (void)hipEventRecord(used[open].b, ctx->stream);
            if (ranged) (void)hipEventSynchronize(used[open].b);  // profiling mode: the range ends with the device work
        }
        if (ranged) rk::trace_pop();
        ranged = false;
        open = (size_t)-1;
    }
    // after the proof's last synchronisation: add every bracket to its accumulator
    void resolve() {
        for (const Bracket& br : used) {
            if (!br.acc) continue;
            (void)hipEventSynchronize(br.b);
            float ms = 0;
            if (hipEventElapsedTime(&ms, br.a, br.b) == hipSuccess) *br.acc += ms;
        }
        used.clear();
    }
};

// core/poly.rs poly_interpolate for the handful of taps of one register
void poly_interpolate(Ext* out, const Ext* x, const Ext* fx, size_t n, uint32_t wm) {
    std::vector<Ext> num(n + 1);
    for (size_t i = 0; i < n; i++) out[i] = bb::ext_zero();
    for (size_t i = 0; i < n; i++) {
        size_t deg = 0;
        Ext denom = bb::ext_one();
        num[0] = bb::ext_one();
        for (size_t j = 0; j < n; j++) {
            if (j == i) continue;
            num[deg + 1] = num[deg];
            for (size_t k = deg; k > 0; k--) num[k] = bb::sub(num[k - 1], bb::mul(num[k], x[j], wm));
            num[0] = bb::sub(bb::ext_zero(), bb::mul(num[0], x[j], wm));
            deg++;
            denom = bb::mul(denom, bb::sub(x[i], x[j]), wm);
        }
        Ext sc = bb::mul(fx[i], bb::inv(denom, wm), wm);
        for (size_t k = 0; k < n; k++) out[k] = bb::add(out[k], bb::mul(num[k], sc, wm));
    }
}

int prove_segment(rk_ctx* ctx, const rk_segment* seg, std::vector<uint32_t>& seal) {
    const rk_taps& taps = seg->taps;
    RK_TRY(rk::check_taps(taps));
    const rk::Shape shape = ctx->sys.shape();
    if (!rk::shape_ok(shape) || seg->po2 < 1 || seg->po2 + shape.blowup_log2 > ntt::LAMBDA) return RK_ERR_INVALID;
    const rk_circuit_hooks* hooks = seg->hooks;
    const bool hook_accum = hooks && hooks->accumulate, hook_check = hooks && (hooks->eval_check || hooks->program);
    for (int g = 0; g < 3; g++) {
        if (taps.group_size[g] == 0) return RK_ERR_INVALID;
        if (!seg->group[g] && !(g == 0 && hook_accum)) return RK_ERR_INVALID;
    }
    if ((!seg->check && !hook_check) || (seg->n_globals && !seg->globals)) return RK_ERR_INVALID;
    if (seg->n_accum_mix > (1u << 16)) return RK_ERR_INVALID;

    // the flow is risc0's; its shape (blow-up, fold arity, final degree, queries, proof of work) follows rk_params
    const unsigned BLOW = shape.blowup_log2;
    const size_t N = (size_t)1 << seg->po2, D = N << BLOW;
    const size_t QUERIES = shape.queries, FRI_FOLD = (size_t)1 << shape.fold_log2, FRI_MIN_DEGREE = shape.min_degree;
    const size_t CHECK_SIZE = (size_t)4 << BLOW;
    const uint32_t wm = ctx->sys.wm;
    const p2::Any& kc = ctx->h_p2;
    Transcript iop(&kc);
    uint32_t digest[8];
    ctx->timing = rk_timing{};
    StageClock sw(ctx);
    std::vector<Ext> rems;  // filled by a download that is only waited for later: must outlive `finish_on_exit`
    sw.start("segment");
    const size_t total_bracket = sw.used.size() - 1;
    struct Finish {  // also on the error returns
        StageClock& c;
        bool ranged;
        ~Finish() {
            (void)hipStreamSynchronize(c.ctx->stream);  // nothing of this proof is in flight once it returns
            c.resolve();
            if (ranged) rk::trace_pop();
        }
    } finish_on_exit{sw, sw.ranged};
    sw.ranged = false;

    {
        uint32_t e[16];
        for (int i = 0; i < 16; i++) e[i] = bb::encode(seg->proof_system_info[i]);
        kc.hash_elems(e, 16, digest);
        iop.commit(digest);
        for (int i = 0; i < 16; i++) e[i] = bb::encode(seg->circuit_info[i]);
        kc.hash_elems(e, 16, digest);
        iop.commit(digest);
        std::vector<uint32_t> vec(seg->globals, seg->globals + seg->n_globals);
        vec.push_back(bb::encode(seg->po2));
        kc.hash_elems(vec.data(), vec.size(), digest);
        iop.commit(digest);
        iop.write(seg->globals, seg->n_globals);
        iop.write(&seg->po2, 1);
    }

    auto load_trace = [&](DevBuf& dst, const uint32_t* src, size_t words) -> int {
        if (seg->on_device == 2) {  // the caller gave the buffer up: transform it in place
            dst.adopt(ctx, (void*)src);
            return RK_OK;
        }
        RK_TRY(dst.alloc(ctx, words * 4));
        RK_HIP_TRY(ctx, hipMemcpyAsync(dst.p, src, words * 4,
                                       seg->on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
        // host inputs stay valid until rk_prove_segment returns (it always drains the stream before
        // returning), so the upload needs no synchronisation here: with page-locked buffers it is a
        // true DMA that overlaps the kernels already queued
        return RK_OK;
    };
    // With an `accumulate` hook the witness of code and data must outlive their commitment (risc0's
    // commit_group works on a copy for the same reason): device inputs are then copied, not consumed,
    // and host inputs are uploaded once into `raw` and copied from there.
    DevBuf raw[3];
    const uint32_t* d_raw[3] = {nullptr, nullptr, nullptr};
    // `from`: where the interpolation's first pass reads the trace when that is not `dst` itself -- a device input that
    // must stay untouched is never copied, the transform's first pass reads it and writes the prover's own buffer
    auto load_group = [&](int g, DevBuf& dst, const uint32_t* src, size_t words, const uint32_t*& from) -> int {
        from = nullptr;
        if (!hook_accum || g == 0) {
            if (seg->on_device != 1) return load_trace(dst, src, words);   // given up (in place) or a host buffer (upload)
            from = src;
            return dst.alloc(ctx, words * 4);
        }
        if (seg->on_device) {
            d_raw[g] = src;
        } else {
            RK_TRY(raw[g].alloc(ctx, words * 4));
            RK_HIP_TRY(ctx, hipMemcpyAsync(raw[g].p, src, words * 4, hipMemcpyHostToDevice, ctx->stream));
            d_raw[g] = raw[g].u32();
        }
        from = d_raw[g];
        return dst.alloc(ctx, words * 4);
    };
    // Prover::commit_group; `preloaded`: pg.coeffs already holds the trace (written by a hook)
    auto commit_group = [&](int g, PolyGroup& pg, const uint32_t* trace, size_t count, bool preloaded = false) -> int {
        const uint32_t* from = nullptr;
        if (!preloaded) RK_TRY(load_group(g, pg.coeffs, trace, count * N, from));
        sw.start("ntt");
        RK_TRY(rk::ntt_reverse_from(ctx, pg.coeffs.u32(), from ? from : pg.coeffs.u32(), N, count, /*fuse_zk_shift=*/true));
        RK_TRY(pg.build(ctx, count, N));
        sw.stop(&ctx->timing.ntt);
        sw.start("hash");
        RK_TRY(pg.merkle.build(ctx, pg.evaluated.u32(), D, count, QUERIES));
        sw.stop(&ctx->timing.hash);
        return pg.merkle.commit(ctx, iop);
    };

    PolyGroup groups[3], check;
    RK_TRY(commit_group(1, groups[1], seg->group[1], taps.group_size[1]));  // code
    RK_TRY(commit_group(2, groups[2], seg->group[2], taps.group_size[2]));  // data
    // the accum mix: drawn once code and data are bound (rv32im prove_segment)
    std::vector<uint32_t> accum_mix(seg->n_accum_mix);
    for (uint32_t i = 0; i < seg->n_accum_mix; i++) accum_mix[i] = iop.random_elem();
    rk_circuit_view view{};
    view.ctx = ctx;
    view.stream = (void*)ctx->stream;
    view.po2 = seg->po2;
    for (int g = 0; g < 3; g++) view.group_size[g] = taps.group_size[g];
    view.globals = seg->globals;
    view.n_globals = seg->n_globals;
    view.mix = accum_mix.data();
    view.n_mix = seg->n_accum_mix;
    if (hook_accum) {
        // CircuitHal::accumulate: the hook writes the accum witness straight into the buffer the
        // interpolation then transforms in place
        RK_TRY(groups[0].coeffs.alloc(ctx, (size_t)taps.group_size[0] * N * 4));
        view.d_trace[1] = d_raw[1];
        view.d_trace[2] = d_raw[2];
        view.d_lde[1] = groups[1].evaluated.u32();
        view.d_lde[2] = groups[2].evaluated.u32();
        sw.start("circuit");
        if (hooks->accumulate(hooks->user, &view, groups[0].coeffs.u32()) != 0) {
            ctx->last_error = "circuit hook `accumulate` failed";
            return RK_ERR_CALLBACK;
        }
        sw.stop(&ctx->timing.circuit);
        RK_TRY(commit_group(0, groups[0], nullptr, taps.group_size[0], true));
        raw[1].release();
        raw[2].release();
        view.d_trace[1] = view.d_trace[2] = nullptr;
    } else {
        RK_TRY(commit_group(0, groups[0], seg->group[0], taps.group_size[0]));  // accum
    }

    // Prover::finalize
    const Ext poly_mix = iop.random_ext();
    const uint32_t* check_from = nullptr;
    if (hook_check) {
        // CircuitHal::eval_check over the LDE domain, into the buffer that becomes the check group
        RK_TRY(check.coeffs.alloc(ctx, 4 * D * 4));
        for (int g = 0; g < 3; g++) view.d_lde[g] = groups[g].evaluated.u32();
        sw.start("circuit");
        if (hooks->eval_check) {
            if (hooks->eval_check(hooks->user, &view, poly_mix.c, check.coeffs.u32()) != 0) {
                ctx->last_error = "circuit hook `eval_check` failed";
                return RK_ERR_CALLBACK;
            }
        } else {
            // the circuit's step program, evaluated by the library (circuit_program.hip)
            RK_TRY(rk::program_eval_check(hooks->program, &view, poly_mix.c, check.coeffs.u32()));
        }
        sw.stop(&ctx->timing.circuit);
    } else if (seg->on_device == 1) {
        RK_TRY(check.coeffs.alloc(ctx, 4 * D * 4));   // pre-computed stand-in, left untouched: read by the transform's first pass
        check_from = seg->check;
    } else {
        RK_TRY(load_trace(check.coeffs, seg->check, 4 * D));  // pre-computed stand-in
    }
    sw.start("ntt");
    // 4 x D evaluations -> 4 x D bit-reversed coefficients = 4 * D/N columns of N (16 for blow-up 4):
    // part c of plane e holds the coefficients n with n mod D/N = bitrev(c) of component e, i.e.
    // check(x) = sum_j x^j g_j(x^(D/N)) with g_j in column (D/N) e + bitrev(j) (for blow-up 4 the
    // verifier's remap [0,2,1,3]).
    // No zk_shift here: the hook evaluates at x_i = 3*w^i, so these already are the coefficients
    // of y -> check(3y), the form every PolyGroup is kept in (recalled items).
    RK_TRY(rk::ntt_reverse_from(ctx, check.coeffs.u32(), check_from ? check_from : check.coeffs.u32(), D, 4, false));
    RK_TRY(check.build(ctx, CHECK_SIZE, N));
    sw.stop(&ctx->timing.ntt);
    sw.start("hash");
    RK_TRY(check.merkle.build(ctx, check.evaluated.u32(), D, CHECK_SIZE, QUERIES));
    sw.stop(&ctx->timing.hash);
    RK_TRY(check.merkle.commit(ctx, iop));

    sw.start("deep");
    Ext z = iop.random_ext();
    uint32_t w27 = ctx->sys.root27m;
    uint32_t back_one = bb::inv(bb::pow(w27, (uint64_t)1 << (27 - seg->po2)));
    Ext z_pow = bb::pow(z, (uint64_t)1 << BLOW, wm);

    // tap openings: every register at z * back_one^back for each of its backs
    size_t tot_taps = 0;
    for (uint32_t r = 0; r < taps.n_regs; r++)
        tot_taps += taps.combo_off[taps.reg_combo[r] + 1] - taps.combo_off[taps.reg_combo[r]];
    uint32_t max_back = 0;
    for (uint32_t b = 0; b < taps.combo_off[taps.n_combos]; b++)
        if (taps.combo_backs[b] > max_back) max_back = taps.combo_backs[b];
    if (max_back > 64) return RK_ERR_INVALID;
    // device power tables: slot b = (z*back_one^b)^k, slot max_back+1 = (z^(D/N))^k
    size_t n_pts = (size_t)max_back + 2;
    std::vector<Ext> pts(n_pts);
    for (uint32_t b = 0; b <= max_back; b++) pts[b] = bb::scale(z, bb::pow(back_one, b));
    pts[max_back + 1] = z_pow;
    std::vector<Ext> all_xs(tot_taps), coeff_u(tot_taps + CHECK_SIZE);
    {
        DevBuf d_pw, d_small;
        RK_TRY(d_pw.alloc(ctx, n_pts * N * 16));
        RK_TRY(rk::ext_powers_many(ctx, d_pw.u32(), pts.data(), n_pts, N, true));
        // every evaluation of the four groups is queued before the one download: which polynomial / which power
        // table per evaluation depend on the tap set only
        std::vector<Ext> eval_u(tot_taps + CHECK_SIZE);
        std::vector<uint32_t> which, sel;
        std::vector<size_t> first(5, 0);
        size_t pos = 0;
        uint32_t reg = 0;
        for (uint32_t gid = 0; gid < 3; gid++) {
            for (; reg < taps.n_regs && taps.reg_group[reg] == gid; reg++) {
                uint32_t cb = taps.reg_combo[reg];
                for (uint32_t b = taps.combo_off[cb]; b < taps.combo_off[cb + 1]; b++) {
                    which.push_back(taps.reg_offset[reg]);
                    sel.push_back(taps.combo_backs[b]);
                    all_xs[pos++] = pts[taps.combo_backs[b]];
                }
            }
            first[gid + 1] = pos;
        }
        for (uint32_t i = 0; i < CHECK_SIZE; i++) {
            which.push_back(i);
            sel.push_back(max_back + 1);
        }
        first[4] = pos + CHECK_SIZE;
        const size_t n_ev = first[4];
        RK_TRY(d_small.alloc(ctx, n_ev * (4 + 4 + 16) + 32));
        uint32_t* d_which = d_small.u32();
        uint32_t* d_sel = d_which + n_ev;
        uint32_t* d_out = d_sel + n_ev;
        if (((uintptr_t)d_out & 15) != 0) d_out += (16 - ((uintptr_t)d_out & 15)) / 4;
        RK_TRY(rk::upload(ctx, d_which, which.data(), n_ev * 4));
        RK_TRY(rk::upload(ctx, d_sel, sel.data(), n_ev * 4));
        for (uint32_t gid = 0; gid < 4; gid++) {
            const PolyGroup& pg = gid < 3 ? groups[gid] : check;
            const size_t a = first[gid], n = first[gid + 1] - a;
            if (n) RK_TRY(rk::eval_dot(ctx, d_out + a * 4, pg.coeffs.u32(), N, d_which + a, d_pw.u32(), d_sel + a, n));
        }
        RK_TRY(d2h_sync(ctx, eval_u.data(), d_out, n_ev * 16));
        // registers -> coefficients of their interpolating polynomials
        size_t p = 0;
        for (uint32_t r = 0; r < taps.n_regs; r++) {
            uint32_t cb = taps.reg_combo[r];
            size_t sz = taps.combo_off[cb + 1] - taps.combo_off[cb];
            poly_interpolate(&coeff_u[p], &all_xs[p], &eval_u[p], sz, wm);
            p += sz;
        }
        for (uint32_t i = 0; i < CHECK_SIZE; i++) coeff_u[tot_taps + i] = eval_u[tot_taps + i];
    }
    iop.write((const uint32_t*)coeff_u.data(), coeff_u.size() * 4);
    kc.hash_elems((const uint32_t*)coeff_u.data(), coeff_u.size() * 4, digest);
    iop.commit(digest);

    // DEEP: mix all columns into one polynomial per combo, remove the openings, divide
    Ext mix = iop.random_ext();
    const size_t combo_count = taps.n_combos;
    DevBuf combos;
    RK_TRY(combos.alloc(ctx, (combo_count + 1) * N * 16));
    RK_HIP_TRY(ctx, hipMemsetAsync(combos.p, 0, (combo_count + 1) * N * 16, ctx->stream));
    {
        Ext cur_mix = bb::ext_one();
        uint32_t reg = 0;
        std::vector<uint32_t> which;
        for (uint32_t gid = 0; gid < 3; gid++) {
            uint32_t gs = taps.group_size[gid];
            which.assign(gs, 0);
            for (uint32_t i = 0; i < gs; i++, reg++) which[i] = taps.reg_combo[reg];
            RK_TRY(rk::mix_poly_coeffs(ctx, combos.u32(), cur_mix, mix, groups[gid].coeffs.u32(), which.data(), gs, N));
            cur_mix = bb::mul(cur_mix, bb::pow(mix, gs, wm), wm);
        }
        which.assign(CHECK_SIZE, (uint32_t)combo_count);
        RK_TRY(rk::mix_poly_coeffs(ctx, combos.u32(), cur_mix, mix, check.coeffs.u32(), which.data(), CHECK_SIZE, N));
        // inputs were bit-reversed, so are the mixed polynomials: natural order for the division
        RK_TRY(rk::bit_reverse_ext(ctx, combos.u32(), N, combo_count + 1));
    }
    DevBuf d_rems;
    {
        // combos[size*combo + i] -= cur * coeff_u[...]: accumulate per touched coefficient on the host
        std::vector<Ext> delta((combo_count + 1) * (max_back + 2), bb::ext_zero());
        std::vector<uint32_t> delta_idx;
        size_t stride = (size_t)max_back + 2;
        Ext cur = bb::ext_one();
        size_t cur_pos = 0;
        for (uint32_t r = 0; r < taps.n_regs; r++) {
            uint32_t cb = taps.reg_combo[r];
            size_t sz = taps.combo_off[cb + 1] - taps.combo_off[cb];
            if (sz > stride || sz > N) return RK_ERR_INVALID;
            for (size_t i = 0; i < sz; i++)
                delta[cb * stride + i] = bb::add(delta[cb * stride + i], bb::mul(cur, coeff_u[cur_pos + i], wm));
            cur = bb::mul(cur, mix, wm);
            cur_pos += sz;
        }
        for (uint32_t i = 0; i < CHECK_SIZE; i++) {
            delta[combo_count * stride] = bb::add(delta[combo_count * stride], bb::mul(cur, coeff_u[cur_pos++], wm));
            cur = bb::mul(cur, mix, wm);
        }
        std::vector<Ext> dl;
        for (size_t c = 0; c <= combo_count; c++) {
            size_t sz = c < combo_count ? taps.combo_off[c + 1] - taps.combo_off[c] : 1;
            for (size_t i = 0; i < sz; i++) {
                delta_idx.push_back((uint32_t)(c * N + i));
                dl.push_back(delta[c * stride + i]);
            }
        }
        RK_TRY(rk::ext_sub_at(ctx, combos.u32(), delta_idx.data(), dl.data(), dl.size()));
        // divide every combo by (x - z*w^-back) for each of its backs and the check combo by
        // (x - z^4): round j handles the j-th back of every combo that has one, in one batch
        size_t max_sz = 1;
        for (size_t c = 0; c < combo_count; c++) max_sz = std::max<size_t>(max_sz, taps.combo_off[c + 1] - taps.combo_off[c]);
        // every round's remainders, read with the next download the transcript needs anyway (FRI's last polynomial)
        RK_TRY(d_rems.alloc(ctx, max_sz * (combo_count + 1) * 16));
        size_t n_rems = 0;
        for (size_t j = 0; j < max_sz; j++) {
            std::vector<size_t> offs;
            std::vector<Ext> zs;
            for (size_t c = 0; c < combo_count; c++) {
                size_t sz = taps.combo_off[c + 1] - taps.combo_off[c];
                if (j < sz) {
                    offs.push_back(c * N);
                    zs.push_back(pts[taps.combo_backs[taps.combo_off[c] + j]]);
                }
            }
            if (j == 0) {
                offs.push_back(combo_count * N);
                zs.push_back(z_pow);
            }
            RK_TRY(rk::poly_divide_many(ctx, combos.u32(), N, offs.data(), zs.data(), offs.size(), nullptr,
                                        d_rems.u32() + n_rems * 4));
            n_rems += offs.size();
        }
        rems.resize(n_rems);
        RK_HIP_TRY(ctx, hipMemcpyAsync(rems.data(), d_rems.p, n_rems * 16, hipMemcpyDeviceToHost, ctx->stream));
    }
    DevBuf final_poly;
    RK_TRY(final_poly.alloc(ctx, N * 16));
    RK_TRY(rk::eltwise_sum_ext(ctx, final_poly.u32(), combos.u32(), N, combo_count + 1));
    RK_TRY(rk::bit_reverse(ctx, final_poly.u32(), N, 4));
    combos.release();
    sw.stop(&ctx->timing.deep);

    // ---- fri_prove ----
    sw.start("fri");
    struct Round {
        size_t domain;
        DevBuf coeffs, evaluated;
        size_t coeffs_words;
        MerkleDev merkle;
    };
    std::vector<std::unique_ptr<Round>> rounds;
    const uint32_t* cur_coeffs = final_poly.u32();
    size_t cur_words = N * 4;
    const size_t orig_domain = D;
    while (cur_words / 4 > FRI_MIN_DEGREE && cur_words / 4 >= FRI_FOLD) {
        std::unique_ptr<Round> r(new Round());
        size_t size = cur_words / 4, domain = size << BLOW;
        r->domain = domain;
        RK_TRY(r->evaluated.alloc(ctx, domain * 16));
        RK_TRY(rk::ntt_forward(ctx, r->evaluated.u32(), cur_coeffs, size, 4, BLOW));
        RK_TRY(r->merkle.build(ctx, r->evaluated.u32(), domain / FRI_FOLD, FRI_FOLD * 4, QUERIES));
        RK_TRY(r->merkle.commit(ctx, iop));
        Ext fold_mix = iop.random_ext();
        r->coeffs_words = size / FRI_FOLD * 4;
        RK_TRY(r->coeffs.alloc(ctx, r->coeffs_words * 4));
        RK_TRY(rk::fri_fold(ctx, r->coeffs.u32(), cur_coeffs, size / FRI_FOLD, fold_mix));
        cur_coeffs = r->coeffs.u32();
        cur_words = r->coeffs_words;
        rounds.push_back(std::move(r));
    }
    {
        DevBuf fin;
        RK_TRY(fin.alloc(ctx, cur_words * 4));
        RK_HIP_TRY(ctx, hipMemcpyAsync(fin.p, cur_coeffs, cur_words * 4, hipMemcpyDeviceToDevice, ctx->stream));
        RK_TRY(rk::bit_reverse(ctx, fin.u32(), cur_words / 4, 4));
        std::vector<uint32_t> h(cur_words);
        RK_TRY(d2h_sync(ctx, h.data(), fin.p, cur_words * 4));
        for (const Ext& rem : rems)  // queued after the DEEP divisions: every division was exact
            if (!bb::eq(rem, bb::ext_zero())) return RK_ERR_INTERNAL;
        iop.write(h.data(), h.size());
        kc.hash_elems(h.data(), h.size(), digest);
        iop.commit(digest);
    }
    if (shape.pow_bits) {
        // proof of work before the query positions exist (Plonky3 `grind`; risc0's parameter set has none):
        // the nonce goes into the seal and, hashed, into the transcript
        uint32_t nonce = 0;
        RK_TRY(rk::pow_grind(ctx, iop.rng.cells, shape.pow_bits, &nonce));
        iop.write(&nonce, 1);
        kc.hash_elems(&nonce, 1, digest);
        iop.commit(digest);
        if (iop.random_bits(shape.pow_bits) != 0) return RK_ERR_INTERNAL;
    }
    sw.stop(&ctx->timing.fri);

    // ---- queries: positions depend only on the sponge, so all openings are gathered in bulk ----
    sw.start("query");
    std::vector<uint32_t> pos0(QUERIES);
    for (size_t q = 0; q < QUERIES; q++) pos0[q] = iop.random_bits(log2u(orig_domain)) % (uint32_t)orig_domain;
    Openings og[3], ocheck;
    std::vector<Openings> oround(rounds.size());
    {
        std::vector<OpenJob> jobs;
        for (int g = 0; g < 3; g++) jobs.push_back(OpenJob{&groups[g].merkle, pos0, &og[g]});
        jobs.push_back(OpenJob{&check.merkle, pos0, &ocheck});
        std::vector<uint32_t> pos = pos0;
        for (size_t k = 0; k < rounds.size(); k++) {
            for (size_t q = 0; q < QUERIES; q++) pos[q] %= (uint32_t)(rounds[k]->domain / FRI_FOLD);
            jobs.push_back(OpenJob{&rounds[k]->merkle, pos, &oround[k]});
        }
        RK_TRY(open_many(ctx, jobs));
    }
    for (size_t q = 0; q < QUERIES; q++) {
        for (int g = 0; g < 3; g++) write_opening(iop, groups[g].merkle, og[g], q);
        write_opening(iop, check.merkle, ocheck, q);
        for (size_t k = 0; k < rounds.size(); k++) write_opening(iop, rounds[k]->merkle, oround[k], q);
    }
    sw.stop(&ctx->timing.query);
    if (total_bracket < sw.used.size()) {  // the outermost bracket was opened first and closes last
        sw.used[total_bracket].acc = &ctx->timing.total;
        (void)hipEventRecord(sw.used[total_bracket].b, ctx->stream);
    }
    seal.swap(iop.proof);
    return RK_OK;
}

}  // namespace

extern "C" {

size_t rk_seal_bound_words(const rk_segment* seg) { return rk::seal_bound_words(seg); }
size_t rk_seal_bound_words_for(const rk_segment* seg, uint32_t queries) { return rk::seal_bound_words(seg, (size_t)queries); }
size_t rk_seal_bound_words_params(const rk_segment* seg, const rk_params* params) {
    if (!params || params->struct_size != sizeof(rk_params)) return 0;
    return rk::seal_bound_words(seg, rk::Shape{params->queries, params->blowup_log2, params->fri_fold_log2, params->fri_min_degree,
                                               params->pow_bits});
}

int rk_prove_segment(rk_ctx* ctx, const rk_segment* seg, uint32_t* h_seal, size_t cap, size_t* seal_words) {
    RK_GUARD_BEGIN
    if (!ctx || !seg || !seal_words || seg->on_device > 2) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> seal;
    int st = prove_segment(ctx, seg, seal);
    if (st != RK_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return st;
    }
    *seal_words = seal.size();
    if (!h_seal || seal.size() > cap) return RK_ERR_CAPACITY;
    std::memcpy(h_seal, seal.data(), seal.size() * 4);
    return RK_OK;
    RK_GUARD_END
}

}  // extern "C"
