// rk_p3_prove / rk_p3_verify / rk_p3_prove_shards (include/raiko_hip.h; the AIR front end rk_air_* is in p3_air.hip): a univariate STARK over the two-adic FRI PCS for AIRs
// handed over as data -- the proof system behind SP1's `client.setup(ELF)` / `client.prove(&pk, stdin)` (reference
// provers/sp1/driver/src/lib.rs:44-57, shard knobs docs/README_Sp1.md:19-32) as far as it exists without SP1's chips:
// Plonky3's p3-uni-stark prover.rs / verifier.rs on p3-fri's TwoAdicFriPcs with a DuplexChallenger, several tables
// under shared challenges the way sp1-core proves the chips of a shard (Plonky3@88ea2b8, reference
// Cargo.lock:4889-5127; the crates are outside the reference tree: RECALLED), with sp1-core's permutation (LogUp)
// argument between the tables (rk_air_create_lookup).  SP1's chips and its recursion VM are NOT here.
//
// Device side of one proof:
//   trace LDE      rows -> columns, iNTT (coset shift fused), expanding NTT (kernels_pcs.hip): the LDE stays the way the
//                  NTT leaves it -- column-major, natural order -- and is committed as rk_matrix layout 2 (committed row r =
//                  natural index bitrev(r)): the hashing, the opened values, the reduced openings and the quotient all walk
//                  the natural index with coalesced column loads and only WRITE at bit-reversed positions, so no pass over
//                  the data exists just to reorder it (the round-2 operator form, rk_pcs_coset_lde_rows, spends 30 % of
//                  its time transposing back to rows)
//   commitments    rk_mmcs_commit (mmcs.hip)
//   lookups        perm_entries_kernel (one lane per row: the batches' sum of +-mult / (alpha + sum beta^j x_j), extension
//                  inverses in registers) writes the permutation trace column-major; the running-sum column is four base
//                  prefix sums (psum_* kernels: workgroup totals, one carry pass, apply); the same LDE / commit as a trace
//   quotient       the AIR is translated once into an rk_program (circuit_program.hip): LOCAL / NEXT are taps of the
//                  column-major LDE (NEXT = one trace row ahead, cyclic), the three selectors are taps of three
//                  columns written by selector_kernel, the asserts one AND_EQZ chain with Horner-ordered powers of
//                  alpha; rk::program_eval_domain runs the interpreter or the hiprtc-generated kernel over the
//                  quotient domain (a sub-coset of the LDE: stride 2^(blow-up - log quotient degree)) and leaves the
//                  result split into chunks, column-major, ready for the chunks' own LDE
//   chunk LDE      iNTT, coefficient i of chunk j times w_(N qd)^(-j i) (chunk_shift_kernel), expanding NTT, to rows
//   openings       rk_pcs_eval_at_many / rk_pcs_reduce_openings, FRI commit phase rk_fri_fold_evals + rk_mmcs_commit
//   queries        every opened row and sibling digest of the proof in ONE gather launch and one download
// The transcript (DuplexChallenger) runs on the host between those steps; the proof of work on the GPU.
#include "p3_air.hpp"
#include "p3_kernels.hpp"

#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <tuple>


namespace {

using bb::Ext;
using rk::DevBuf;
using rk::NEXT_BACK;
constexpr uint32_t MAX_TABLES = 32, MAX_QD_LOG = 4;

// ---------------------------------------------------------------- p3-challenger DuplexChallenger (host)
struct Challenger {
    const p2::Any* k;
    uint32_t state[p2::MAX_CELLS], in[p2::MAX_CELLS], out[p2::MAX_CELLS];
    unsigned n_in = 0, n_out = 0;
    explicit Challenger(const p2::Any* kk) : k(kk) { std::memset(state, 0, sizeof state); }
    unsigned rate() const { return (unsigned)k->rate(); }
    void duplex() {
        for (unsigned i = 0; i < n_in; i++) state[i] = in[i];
        n_in = 0;
        k->permute(state);
        for (unsigned i = 0; i < rate(); i++) out[i] = state[i];
        n_out = rate();
    }
    void observe(uint32_t v) {
        n_out = 0;
        in[n_in++] = v;
        if (n_in == rate()) duplex();
    }
    void observe(const uint32_t* v, size_t n) {
        for (size_t i = 0; i < n; i++) observe(v[i]);
    }
    uint32_t sample() {
        if (n_in != 0 || n_out == 0) duplex();
        return out[--n_out];
    }
    Ext sample_ext() {
        Ext r;
        for (int i = 0; i < 4; i++) r.c[i] = sample();
        return r;
    }
    uint32_t sample_bits(unsigned bits) { return bb::decode(sample()) & (uint32_t)(((uint64_t)1 << bits) - 1); }
    bool check_witness(unsigned bits, uint32_t w) {
        observe(bb::encode(w));
        return sample_bits(bits) == 0;
    }
};

// folder.rs on extension elements (the verifier's side): accumulator = accumulator * alpha + x per assert, in order
struct PermView {   // the verifier's view of a table's lookup argument (all null without one)
    const Ext *local = nullptr, *next = nullptr;
    const uint32_t *chal = nullptr, *cumsum = nullptr;
};
Ext air_fold(const rk_air& air, const Ext* local, const Ext* next, const uint32_t* pub, const Ext& is_first, const Ext& is_last,
             const Ext& is_trans, const Ext& alpha, uint32_t wm, const PermView& pv) {
    std::vector<Ext> v;
    v.reserve(air.steps.size());
    Ext acc = bb::ext_zero();
    for (const rk_air_step& st : air.steps) {
        switch (st.op) {
            case RK_AIR_CONST: v.push_back(bb::ext_from(bb::encode(st.a))); break;
            case RK_AIR_LOCAL: v.push_back(local[st.a]); break;
            case RK_AIR_NEXT: v.push_back(next[st.a]); break;
            case RK_AIR_PUBLIC: v.push_back(bb::ext_from(pub[st.a])); break;
            case RK_AIR_IS_FIRST_ROW: v.push_back(is_first); break;
            case RK_AIR_IS_LAST_ROW: v.push_back(is_last); break;
            case RK_AIR_IS_TRANSITION: v.push_back(is_trans); break;
            case RK_AIR_PERM_LOCAL: v.push_back(pv.local[st.a]); break;
            case RK_AIR_PERM_NEXT: v.push_back(pv.next[st.a]); break;
            case RK_AIR_CHALLENGE: v.push_back(bb::ext_from(pv.chal[st.a])); break;
            case RK_AIR_CUMSUM: v.push_back(bb::ext_from(pv.cumsum[st.a])); break;
            case RK_AIR_ADD: v.push_back(bb::add(v[st.a], v[st.b])); break;
            case RK_AIR_SUB: v.push_back(bb::sub(v[st.a], v[st.b])); break;
            case RK_AIR_MUL: v.push_back(bb::mul(v[st.a], v[st.b], wm)); break;
            case RK_AIR_NEG: v.push_back(bb::sub(bb::ext_zero(), v[st.a])); break;
            default: acc = bb::add(bb::mul(acc, alpha, wm), v[st.a]); break;
        }
    }
    return acc;
}

// ---------------------------------------------------------------- kernels
// LagrangeSelectors on the quotient coset (p3-commit domain.rs selectors_on_coset): point i is x = shift * w_d^i;
// Z_H(x) = x^n - 1 takes d / n values.  Columns: 0 is_first_row = Z_H / (x - 1), 1 is_last_row = Z_H / (x - g^-1),
// 2 is_transition = x - g^-1.  One lane per point; the two inversions are Fermat powers (a few hundred products per
// point against the hundreds of thousands a wide AIR costs).
struct SelArgs {
    uint32_t* out;       // 3 columns of d words
    size_t d;
    unsigned log_d;
    uint32_t shiftm, g_inv;
    uint32_t zh[16];     // shift^n * w_(d/n)^r - 1
    uint32_t ratio_mask, want;   // want: bit c set = column c is read by the AIR
};
__global__ void selector_kernel(SelArgs a, ntt::Tables tb) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.d) return;
    const uint32_t x = bb::mul(a.shiftm, ntt::root_pow(tb, 0, (uint32_t)(i << (ntt::LAMBDA - a.log_d))));
    const uint32_t zh = a.zh[i & a.ratio_mask];
    if (a.want & 1u) a.out[i] = bb::mul(zh, bb::inv(bb::sub(x, bb::ONE)));
    if (a.want & 2u) a.out[a.d + i] = bb::mul(zh, bb::inv(bb::sub(x, a.g_inv)));
    if (a.want & 4u) a.out[2 * a.d + i] = bb::sub(x, a.g_inv);
}

// bit-reversed coefficients of the chunk columns (4 per chunk, n words each): coefficient i of chunk j times
// w_(n qd)^(-j i) -- Pcs::commit's `shift = generator / domain.shift` for the chunk domain shift * w^j * H_n
__global__ void chunk_shift_kernel(uint32_t* __restrict__ q, size_t n, unsigned log_n, unsigned log_nq, unsigned n_cols, ntt::Tables tb) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * n_cols) return;
    const uint32_t col = (uint32_t)(t >> log_n), p = (uint32_t)(t & (n - 1)), j = col >> 2;
    if (j == 0) return;
    const uint32_t i = bb::bitrev(p, log_n);
    const uint32_t e = (uint32_t)(((uint64_t)j * i) & (((uint64_t)1 << log_nq) - 1));
    q[t] = bb::mul(q[t], ntt::root_pow(tb, 1, e << (ntt::LAMBDA - log_nq)));
}

// out[i] += in[i] (extension elements as 4 words): the shorter reduced opening joining the folded vector
__global__ void add_words_kernel(uint32_t* __restrict__ io, const uint32_t* __restrict__ in, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) io[i] = bb::add(io[i], in[i]);
}

// ---- lookups: the permutation trace.  The lane bodies are p3_kernels.hpp's (tests/emul runs them on the CPU too)
using p3k::PERM_LD;
using p3k::PERM_ROWS;
using p3k::PermArgs;
__global__ void __launch_bounds__(PERM_ROWS) perm_entries_kernel(PermArgs a) {
    extern __shared__ uint32_t tile[];   // n_used x PERM_LD
    p3k::perm_stage(a, blockIdx.x, threadIdx.x, tile);
    __syncthreads();
    p3k::perm_row(a, blockIdx.x, threadIdx.x, tile);
}

// inclusive prefix sums of `cols` columns of n words (blockIdx.y = column), in place: workgroup totals, one carry pass
// per column, apply
constexpr int PS_TPB = 256, PS_CH = 8, PS_BLOCK = PS_TPB * PS_CH;
__device__ uint32_t psum_block_scan(uint32_t v, uint32_t* sh) {   // inclusive, across the workgroup
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int d = 1; d < PS_TPB; d <<= 1) {
        const uint32_t o = t >= d ? sh[t - d] : 0u;
        __syncthreads();
        if (t >= d) {
            v = bb::add(o, v);
            sh[t] = v;
        }
        __syncthreads();
    }
    return v;
}
__global__ void psum_totals_kernel(uint32_t* __restrict__ totals, const uint32_t* __restrict__ io, size_t n, size_t n_blocks) {
    __shared__ uint32_t sh[PS_TPB];
    const uint32_t* col = io + (size_t)blockIdx.y * n;
    const size_t base = (size_t)blockIdx.x * PS_BLOCK;
    uint32_t acc = 0;
    for (int j = 0; j < PS_CH; j++) {   // lane t takes elements t, t + 256, ...: coalesced; only the sum matters here
        const size_t i = base + (size_t)j * PS_TPB + threadIdx.x;
        if (i < n) acc = bb::add(acc, col[i]);
    }
    acc = psum_block_scan(acc, sh);
    if (threadIdx.x == PS_TPB - 1) totals[(size_t)blockIdx.y * n_blocks + blockIdx.x] = acc;
}
__global__ void psum_carry_kernel(uint32_t* totals, size_t n_blocks) {   // totals[b] <- sum of the blocks before b
    __shared__ uint32_t sh[PS_TPB];
    uint32_t* t = totals + (size_t)blockIdx.x * n_blocks;
    const size_t per = (n_blocks + PS_TPB - 1) / PS_TPB;
    const size_t lo0 = (size_t)threadIdx.x * per, lo = lo0 < n_blocks ? lo0 : n_blocks, hi = lo + per < n_blocks ? lo + per : n_blocks;
    uint32_t acc = 0;
    for (size_t i = lo; i < hi; i++) acc = bb::add(acc, t[i]);
    const uint32_t incl = psum_block_scan(acc, sh);
    uint32_t run = bb::sub(incl, acc);
    for (size_t i = lo; i < hi; i++) {
        const uint32_t v = t[i];
        t[i] = run;
        run = bb::add(run, v);
    }
}
__global__ void psum_apply_kernel(uint32_t* __restrict__ io, const uint32_t* __restrict__ totals, size_t n, size_t n_blocks) {
    __shared__ uint32_t sh[PS_TPB];
    uint32_t* col = io + (size_t)blockIdx.y * n;
    const size_t base = (size_t)blockIdx.x * PS_BLOCK + (size_t)threadIdx.x * PS_CH;   // lane t: 8 consecutive words (two 16 B loads)
    uint32_t v[PS_CH];
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < PS_CH; j++) {
        v[j] = base + j < n ? col[base + j] : 0u;
        acc = bb::add(acc, v[j]);
        v[j] = acc;
    }
    const uint32_t incl = psum_block_scan(acc, sh);
    const uint32_t carry = bb::add(bb::sub(incl, acc), totals[(size_t)blockIdx.y * n_blocks + blockIdx.x]);
#pragma unroll
    for (int j = 0; j < PS_CH; j++)
        if (base + j < n) col[base + j] = bb::add(v[j], carry);
}

// every opened row / digest of a proof's query phase: job = (source address, words, destination offset); one
// 64-lane group per job
struct GatherJob {
    uint64_t src;
    uint64_t stride;     // in words: 1 for a digest or a row of a row-major matrix, the column length for a row of a column-major one
    uint32_t words, dst;
};
__global__ void gather_jobs_kernel(uint32_t* __restrict__ dst, const GatherJob* __restrict__ jobs, size_t n_jobs) {
    const size_t j = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if (j >= n_jobs) return;
    const GatherJob jb = jobs[j];
    const uint32_t* s = reinterpret_cast<const uint32_t*>(jb.src);
    for (uint32_t w = threadIdx.x & 63; w < jb.words; w += 64) dst[jb.dst + w] = s[(size_t)w * jb.stride];
}

// ---------------------------------------------------------------- prover state

struct TableState {
    unsigned k = 0, lqd = 0;      // log2 rows, log2 quotient degree
    size_t n = 0, H = 0, w = 0;
    DevBuf lde;                   // w columns of H natural-order evaluations (rk_matrix layout 2: committed row r at index bitrev(r))
    DevBuf chunks;                // 4 qd columns of H: the qd chunk LDEs side by side (as one matrix they hash, open and
                                  // reduce exactly like qd matrices of width 4 that follow each other in the batch)
    std::vector<uint32_t> y;      // opened values: local 4w | next 4w | [perm local 4pw | perm next 4pw] | chunks 16 each
    size_t pw = 0;                // base columns of the permutation trace (0: the table has no lookups)
    DevBuf staged;                // a host trace's copy in HBM, kept for the permutation trace
    const uint32_t* d_trace = nullptr;
    DevBuf perm;                  // pw columns of H, laid out like lde
    uint32_t cumsum[4] = {0, 0, 0, 0};
};

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

size_t proof_bound(const rk_params& p, const rk_p3_table* t, uint32_t n, const uint32_t* lqd) {
    size_t words = 1 + n + 16, log_max = 0, log_pmax = 0, row_t = 0, row_q = 0, row_p = 0;
    for (uint32_t i = 0; i < n; i++) {
        const size_t pw = t[i].air->perm_width;
        words += 8 * (size_t)t[i].width + 8 * pw + (pw ? 4 : 0) + ((size_t)16 << lqd[i]);
        log_max = std::max<size_t>(log_max, t[i].log_height + p.blowup_log2);
        if (pw) log_pmax = std::max<size_t>(log_pmax, t[i].log_height + p.blowup_log2);
        row_t += t[i].width;
        row_p += pw;
        row_q += (size_t)4 << lqd[i];
    }
    const size_t rounds = log_max - p.blowup_log2;
    words += 1 + 8 * rounds + 4 + 1 + (row_p ? 8 : 0);
    size_t per_query = row_t + row_q + 2 * 8 * log_max + (row_p ? row_p + 8 * log_pmax : 0);
    for (size_t r = 0; r < rounds; r++) per_query += 4 + 8 * (log_max - 1 - r);
    return words + per_query * p.queries;
}

int check_tables(const rk_params& par, const rk_p3_table* tables, uint32_t n_tables, bool prover, uint32_t* lqd) {
    if (!tables || n_tables == 0 || n_tables > MAX_TABLES) return RK_ERR_INVALID;
    for (uint32_t t = 0; t < n_tables; t++) {
        const rk_p3_table& tb = tables[t];
        if (!tb.air || tb.width != tb.air->width || tb.n_public != tb.air->n_public || (tb.n_public && !tb.public_values)) return RK_ERR_INVALID;
        for (uint32_t i = 0; i < tb.n_public; i++)
            if (tb.public_values[i] >= bb::P) return RK_ERR_INVALID;
        lqd[t] = tb.air->info.log_quotient_degree;
        if (lqd[t] > par.blowup_log2 || lqd[t] > MAX_QD_LOG) return RK_ERR_INVALID;  // the LDE must cover the quotient domain
        if (prover && (!tb.trace || tb.log_height < 1 || tb.log_height + par.blowup_log2 > ntt::LAMBDA || tb.on_device > 1)) return RK_ERR_INVALID;
    }
    return RK_OK;
}

int d2h(rk_ctx* ctx, void* h, const void* d, size_t bytes) {
    RK_HIP_TRY(ctx, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RK_OK;
}

// One rk_p3_prove: the state the stages share and the stages in the order the transcript imposes
struct ProofRun {
    rk_ctx* ctx;
    const rk_p3_table* tables;
    uint32_t n_tables;
    rk_params par;
    unsigned blow = 0;
    uint32_t lqd[MAX_TABLES];
    rk_p3_timing& tm;
    double t_mark = 0;
    std::vector<uint32_t> pf;            // the proof words before the query openings
    Challenger ch;
    std::vector<TableState> ts;
    std::vector<rk_matrix> tmats, pmats, qmats;   // the three input batches: traces, permutation traces, quotient chunks
    DevBuf tnodes, pnodes, qnodes;
    size_t Ht = 0, Hp = 0, Hq = 0;
    unsigned log_max = 0, n_rounds = 0;
    std::vector<uint32_t> pchal;         // lookups: [alpha | beta^0 | .. | beta^K], every table reads a prefix
    uint32_t root[8];
    Ext zeta;
    DevBuf ro[ntt::LAMBDA + 1];          // reduced openings per LDE height
    struct Layer {
        DevBuf values, nodes;            // 2^(log_max - round) extension elements; the tree over their pairs
    };
    std::vector<Layer> layers;

    ProofRun(rk_ctx* c, const rk_p3_table* t, uint32_t n) : ctx(c), tables(t), n_tables(n), tm(c->p3_timing), ch(&c->h_p2) {}
    void lap(float& slot) {
        (void)hipStreamSynchronize(ctx->stream);
        const double t = now_ms();
        slot += (float)(t - t_mark);
        t_mark = t;
    }
    void push(const uint32_t* w, size_t n) { pf.insert(pf.end(), w, w + n); }

    int commit_traces();
    int permutation_traces();
    int quotients(const Ext& alpha);
    int open();
    int fri();
    int queries(uint32_t* h_proof, size_t capacity, size_t* proof_words);
};

int ProofRun::commit_traces() {
    // ---- trace LDEs and their commitment
    ts = std::vector<TableState>(n_tables);
    tmats.resize(n_tables);
    pf.push_back(n_tables);
    for (uint32_t t = 0; t < n_tables; t++) {
        const rk_p3_table& tb = tables[t];
        TableState& s = ts[t];
        s.k = tb.log_height;
        s.lqd = lqd[t];
        s.n = (size_t)1 << s.k;
        s.H = s.n << blow;
        s.w = tb.width;
        pf.push_back(tb.log_height);
        s.pw = tb.air->perm_width;
        s.d_trace = tb.trace;
        if (!tb.on_device) {
            RK_TRY(s.staged.alloc(ctx, s.n * s.w * 4));
            RK_HIP_TRY(ctx, hipMemcpyAsync(s.staged.p, tb.trace, s.n * s.w * 4, hipMemcpyHostToDevice, ctx->stream));
            s.d_trace = s.staged.u32();
        }
        RK_TRY(s.lde.alloc(ctx, s.H * s.w * 4));
        RK_TRY(rk::pcs_coset_lde_cols(ctx, s.lde.u32(), s.d_trace, s.n, s.w));
        if (!tb.on_device) RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the caller's host buffer is free again
        if (!s.pw) s.staged.reset();   // only the lookups read the rows again
        tmats[t] = rk_matrix{s.lde.u32(), (uint32_t)s.H, (uint32_t)s.w, 2};
        Ht = std::max(Ht, s.H);
        log_max = std::max(log_max, s.k + blow);
    }
    lap(tm.lde);
    RK_TRY(tnodes.alloc(ctx, 2 * Ht * p2::OUT * 4));
    RK_TRY(rk_mmcs_commit(ctx, tmats.data(), n_tables, tnodes.u32(), root));
    push(root, 8);
    ch.observe(root, 8);
    for (uint32_t t = 0; t < n_tables; t++) ch.observe(tables[t].public_values, tables[t].n_public);
    lap(tm.commit);

    return RK_OK;
}

int ProofRun::permutation_traces() {
    // ---- lookups: permutation traces, their LDE and commitment (nothing of this without interactions)
    {
        uint32_t n_chal = 0;
        for (uint32_t t = 0; t < n_tables; t++) n_chal = std::max(n_chal, tables[t].air->n_chal);
        if (n_chal) {
            const Ext pa = ch.sample_ext(), pb = ch.sample_ext();
            pchal.resize(n_chal);
            std::memcpy(pchal.data(), pa.c, 16);
            Ext cur = bb::ext_one();
            for (uint32_t j = 1; 4 * j < n_chal; j++) {
                std::memcpy(&pchal[4 * j], cur.c, 16);
                cur = bb::mul(cur, pb, ctx->sys.wm);
            }
        }
        for (uint32_t t = 0; t < n_tables && n_chal; t++) {
            TableState& s = ts[t];
            if (!s.pw) continue;
            const rk_air& air = *tables[t].air;
            std::vector<uint32_t> desc(pchal.begin(), pchal.begin() + air.n_chal);
            desc.insert(desc.end(), air.lookups.begin(), air.lookups.end());
            const uint32_t desc_words = (uint32_t)desc.size();
            desc.insert(desc.end(), air.used.begin(), air.used.end());
            DevBuf d_desc, cols, totals;
            RK_TRY(d_desc.alloc(ctx, desc.size() * 4));
            RK_TRY(rk::upload(ctx, d_desc.p, desc.data(), desc.size() * 4));
            RK_TRY(cols.alloc(ctx, s.n * s.pw * 4));
            PermArgs a{cols.u32(), s.d_trace, d_desc.u32(), s.n, s.w, air.n_chal, air.n_lookups, ctx->sys.wm, (uint32_t)air.used.size(), desc_words};
            const size_t lds = std::max<size_t>(air.used.size(), 1) * PERM_LD * 4;
            if (lds > 64 * 1024)
                RK_HIP_TRY(ctx, hipFuncSetAttribute((const void*)perm_entries_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(perm_entries_kernel, dim3((unsigned)((s.n + PERM_ROWS - 1) / PERM_ROWS)), dim3(PERM_ROWS), lds, ctx->stream, a);
            RK_TRY(rk::post_launch(ctx, "perm_entries_kernel"));
            s.staged.reset();
            // the running sum: four base prefix sums over the row totals, in place
            const size_t nblk = (s.n + PS_BLOCK - 1) / PS_BLOCK;
            uint32_t* phi = cols.u32() + (s.pw - 4) * s.n;
            RK_TRY(totals.alloc(ctx, 4 * nblk * 4));
            hipLaunchKernelGGL(psum_totals_kernel, dim3((unsigned)nblk, 4), dim3(PS_TPB), 0, ctx->stream, totals.u32(), (const uint32_t*)phi, s.n, nblk);
            RK_TRY(rk::post_launch(ctx, "psum_totals_kernel"));
            hipLaunchKernelGGL(psum_carry_kernel, dim3(4), dim3(PS_TPB), 0, ctx->stream, totals.u32(), nblk);
            RK_TRY(rk::post_launch(ctx, "psum_carry_kernel"));
            hipLaunchKernelGGL(psum_apply_kernel, dim3((unsigned)nblk, 4), dim3(PS_TPB), 0, ctx->stream, phi, (const uint32_t*)totals.u32(), s.n, nblk);
            RK_TRY(rk::post_launch(ctx, "psum_apply_kernel"));
            for (int k = 0; k < 4; k++)
                RK_HIP_TRY(ctx, hipMemcpyAsync(&s.cumsum[k], phi + (size_t)k * s.n + (s.n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
            // the same LDE as a trace's, from columns
            RK_TRY(rk::ntt_reverse(ctx, cols.u32(), s.n, s.pw, /*fuse_zk_shift=*/true));
            RK_TRY(s.perm.alloc(ctx, s.H * s.pw * 4));
            RK_TRY(rk::ntt_forward(ctx, s.perm.u32(), cols.u32(), s.n, s.pw, blow));
            pmats.push_back(rk_matrix{s.perm.u32(), (uint32_t)s.H, (uint32_t)s.pw, 2});
            Hp = std::max(Hp, s.H);
        }
        if (!pmats.empty()) {
            lap(tm.perm);
            RK_TRY(pnodes.alloc(ctx, 2 * Hp * p2::OUT * 4));
            RK_TRY(rk_mmcs_commit(ctx, pmats.data(), (uint32_t)pmats.size(), pnodes.u32(), root));
            RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the cumulative sums have landed
            push(root, 8);
            ch.observe(root, 8);
            for (const TableState& s : ts)
                if (s.pw) {
                    push(s.cumsum, 4);
                    ch.observe(s.cumsum, 4);
                }
            lap(tm.commit);
        }
    }
    return RK_OK;
}

int ProofRun::quotients(const Ext& alpha) {
    // ---- quotients
    for (uint32_t t = 0; t < n_tables; t++) {
        const rk_p3_table& tb = tables[t];
        TableState& s = ts[t];
        const unsigned kq = s.k + s.lqd;
        const size_t qd = (size_t)1 << s.lqd, d = s.n << s.lqd, n_qcols = 4 * qd;
        rk_program* pg = tb.air->prog;
        DevBuf sel, q;
        if (pg->group_min[0]) {
            RK_TRY(sel.alloc(ctx, 3 * d * 4));
            SelArgs a{};
            a.out = sel.u32();
            a.d = d;
            a.log_d = kq;
            a.shiftm = ctx->sys.shiftm;
            a.g_inv = bb::inv(bb::pow(ctx->sys.root27m, (uint64_t)1 << (27 - s.k)));
            const uint32_t sn = bb::pow(ctx->sys.shiftm, s.n), wr = bb::pow(ctx->sys.root27m, (uint64_t)1 << (27 - s.lqd));
            for (unsigned r = 0; r < qd; r++) a.zh[r] = bb::sub(bb::mul(sn, bb::pow(wr, r)), bb::ONE);
            a.ratio_mask = (uint32_t)qd - 1;
            a.want = tb.air->sel_mask;
            hipLaunchKernelGGL(selector_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, ctx->stream, a, ctx->tb);
            RK_TRY(rk::post_launch(ctx, "selector_kernel"));
        }
        RK_TRY(q.alloc(ctx, n_qcols * s.n * 4));
        rk::EvalDomain dom;
        dom.ctx = ctx;
        dom.po2 = s.k;
        dom.ratio_log2 = dom.split_log2 = s.lqd;
        dom.d_cols[0] = sel.u32();
        dom.group_size[0] = 3;
        dom.col_len[0] = d;
        dom.d_cols[2] = s.lde.u32();
        dom.group_size[2] = (uint32_t)s.w;
        dom.col_len[2] = s.H;
        dom.stride_log2[2] = blow - s.lqd;
        dom.globals = tb.public_values;
        dom.n_globals = tb.n_public;
        std::vector<uint32_t> globals;   // lookups: public values | challenges | cumulative sum
        if (s.pw) {
            dom.d_cols[1] = s.perm.u32();
            dom.group_size[1] = (uint32_t)s.pw;
            dom.col_len[1] = s.H;
            dom.stride_log2[1] = blow - s.lqd;
            globals.assign(tb.public_values, tb.public_values + tb.n_public);
            globals.insert(globals.end(), pchal.begin(), pchal.begin() + tb.air->n_chal);
            globals.insert(globals.end(), s.cumsum, s.cumsum + 4);
            dom.globals = globals.data();
            dom.n_globals = (uint32_t)globals.size();
        }
        RK_TRY(rk::program_eval_domain(pg, dom, alpha.c, q.u32()));
        sel.reset();
        // the chunks' own LDE: interpolate over H_n, move to the chunk's coset, evaluate on the LDE coset
        RK_TRY(rk::ntt_reverse(ctx, q.u32(), s.n, n_qcols, /*fuse_zk_shift=*/false));
        if (qd > 1) {
            const size_t tot = s.n * n_qcols;
            hipLaunchKernelGGL(chunk_shift_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, q.u32(), s.n, s.k, kq,
                               (unsigned)n_qcols, ctx->tb);
            RK_TRY(rk::post_launch(ctx, "chunk_shift_kernel"));
        }
        RK_TRY(s.chunks.alloc(ctx, n_qcols * s.H * 4));
        RK_TRY(rk::ntt_forward(ctx, s.chunks.u32(), q.u32(), s.n, n_qcols, blow));
        qmats.push_back(rk_matrix{s.chunks.u32(), (uint32_t)s.H, (uint32_t)n_qcols, 2});
        Hq = std::max(Hq, s.H);
    }
    lap(tm.quotient);
    RK_TRY(qnodes.alloc(ctx, 2 * Hq * p2::OUT * 4));
    RK_TRY(rk_mmcs_commit(ctx, qmats.data(), (uint32_t)qmats.size(), qnodes.u32(), root));
    push(root, 8);
    ch.observe(root, 8);
    zeta = ch.sample_ext();
    lap(tm.commit);

    return RK_OK;
}

int ProofRun::open() {
    // ---- PCS open: opened values and reduced openings
    const Ext alpha2 = ch.sample_ext();
    uint64_t num_reduced[ntt::LAMBDA + 1] = {0};
    // every opened value first (no host round trip between the launches), ONE download, then the reduced openings, which
    // need the opened values on the host: a shard of a dozen tables waits once instead of three dozen times
    struct Opening {
        TableState* s;
        const uint32_t* mat;
        size_t w, n_points, y_at, d_at;   // y_at: words into s->y; d_at: words into the device buffer
    };
    std::vector<Opening> openings;
    size_t y_words = 0;
    for (TableState& s : ts) s.y.resize(8 * s.w + 8 * s.pw + ((size_t)16 << s.lqd));
    auto add_opening = [&](TableState& s, const uint32_t* mat, size_t w, size_t n_points, size_t y_at) {
        openings.push_back(Opening{&s, mat, w, n_points, y_at, y_words});
        y_words += 4 * w * n_points;
    };
    for (TableState& s : ts) add_opening(s, s.lde.u32(), s.w, 2, 0);                       // round 0: every trace at zeta and zeta * g
    for (TableState& s : ts)
        if (s.pw) add_opening(s, s.perm.u32(), s.pw, 2, 8 * s.w);                          // round 1 (lookups): the permutation traces, likewise
    for (TableState& s : ts) add_opening(s, s.chunks.u32(), (size_t)4 << s.lqd, 1, 8 * s.w + 8 * s.pw);   // last round: every quotient chunk at zeta
    DevBuf d_ys;
    RK_TRY(d_ys.alloc(ctx, y_words * 4));
    auto points_of = [&](const TableState& s, uint32_t pts[8]) {
        std::memcpy(pts, zeta.c, 16);
        const Ext zn = bb::scale(zeta, bb::pow(ctx->sys.root27m, (uint64_t)1 << (27 - s.k)));
        std::memcpy(pts + 4, zn.c, 16);
    };
    for (const Opening& o : openings) {
        uint32_t pts[8];
        points_of(*o.s, pts);
        RK_TRY(rk::pcs_eval_at(ctx, d_ys.u32() + o.d_at, o.mat, o.s->H, o.w, pts, o.n_points, /*cols=*/true));
    }
    {
        std::vector<uint32_t> h_ys(y_words);
        RK_TRY(d2h(ctx, h_ys.data(), d_ys.p, y_words * 4));
        for (const Opening& o : openings) std::memcpy(o.s->y.data() + o.y_at, &h_ys[o.d_at], 4 * o.w * o.n_points * 4);
    }
    for (const Opening& o : openings) {
        TableState& s = *o.s;
        const unsigned lh = s.k + blow;
        if (!ro[lh].p) {
            RK_TRY(ro[lh].alloc(ctx, s.H * 16));
            RK_HIP_TRY(ctx, hipMemsetAsync(ro[lh].p, 0, s.H * 16, ctx->stream));
        }
        uint32_t pts[8];
        points_of(s, pts);
        RK_TRY(rk::pcs_reduce_openings(ctx, ro[lh].u32(), o.mat, s.H, o.w, o.n_points, pts, s.y.data() + o.y_at, alpha2, num_reduced[lh], /*cols=*/true));
        num_reduced[lh] += o.n_points * o.w;
    }
    for (const TableState& s : ts) push(s.y.data(), s.y.size());
    lap(tm.open);

    return RK_OK;
}

int ProofRun::fri() {
    // ---- FRI commit phase
    n_rounds = log_max - blow;
    layers = std::vector<Layer>(n_rounds);
    DevBuf folded = std::move(ro[log_max]);
    size_t len = (size_t)1 << log_max;
    pf.push_back(n_rounds);
    for (unsigned rd = 0; rd < n_rounds; rd++) {
        Layer& L = layers[rd];
        RK_TRY(L.nodes.alloc(ctx, len * p2::OUT * 4));   // 2 * (len / 2) digests
        const rk_matrix lm{folded.u32(), (uint32_t)(len / 2), 8, 1};
        RK_TRY(rk_mmcs_commit(ctx, &lm, 1, L.nodes.u32(), root));
        push(root, 8);
        ch.observe(root, 8);
        const Ext beta = ch.sample_ext();
        DevBuf nxt;
        RK_TRY(nxt.alloc(ctx, len / 2 * 16));
        RK_TRY(rk::fri_fold_evals(ctx, nxt.u32(), folded.u32(), len / 2, beta));
        L.values = std::move(folded);
        folded = std::move(nxt);
        len /= 2;
        const unsigned lg = log2u(len);
        if (lg != log_max && ro[lg].p) {
            hipLaunchKernelGGL(add_words_kernel, dim3((unsigned)((len * 4 + 255) / 256)), dim3(256), 0, ctx->stream, folded.u32(),
                               (const uint32_t*)ro[lg].u32(), len * 4);
            RK_TRY(rk::post_launch(ctx, "add_words_kernel"));
        }
    }
    std::vector<uint32_t> fin(len * 4);
    RK_TRY(d2h(ctx, fin.data(), folded.p, len * 16));
    for (size_t i = 1; i < len; i++)
        if (std::memcmp(&fin[4 * i], &fin[0], 16) != 0) return RK_ERR_INTERNAL;   // `blowup` values of a constant
    push(fin.data(), 4);
    ch.observe(fin.data(), 4);
    uint32_t witness = 0;
    if (par.pow_bits) RK_TRY(rk::duplex_grind(ctx, ch.state, ch.in, ch.n_in, par.pow_bits, &witness));
    if (!ch.check_witness(par.pow_bits, witness)) return RK_ERR_INTERNAL;
    pf.push_back(witness);
    lap(tm.fri);

    return RK_OK;
}

int ProofRun::queries(uint32_t* h_proof, size_t capacity, size_t* proof_words) {
    // ---- queries: all indices first (they depend on the transcript only), then one gather
    std::vector<GatherJob> jobs;
    uint32_t at = 0;
    auto job = [&](const uint32_t* src, uint32_t words, uint64_t stride = 1) {
        jobs.push_back(GatherJob{(uint64_t)(uintptr_t)src, stride, words, at});
        at += words;
    };
    auto open_batch = [&](const std::vector<rk_matrix>& mats, const uint32_t* nodes, size_t H, uint32_t index) {
        for (const rk_matrix& m : mats) {   // layout 2: committed row r of a column-major matrix sits at index bitrev(r) of every column
            const uint32_t r = (uint32_t)(index / (H / m.height));
            job(m.d_values + bb::bitrev(r, log2u(m.height)), m.width, m.height);
        }
        for (size_t idx = H + index; idx > 1; idx >>= 1) job(nodes + (idx ^ 1) * p2::OUT, p2::OUT);
    };
    for (uint32_t qi = 0; qi < par.queries; qi++) {
        const uint32_t index = ch.sample_bits(log_max);
        open_batch(tmats, tnodes.u32(), Ht, index >> (log_max - log2u(Ht)));
        if (!pmats.empty()) open_batch(pmats, pnodes.u32(), Hp, index >> (log_max - log2u(Hp)));
        open_batch(qmats, qnodes.u32(), Hq, index >> (log_max - log2u(Hq)));
        for (unsigned rd = 0; rd < n_rounds; rd++) {
            const uint32_t idx = index >> rd, pair = idx >> 1;
            const size_t height = ((size_t)1 << (log_max - rd)) / 2;
            job(layers[rd].values.u32() + (2 * (size_t)pair + ((idx ^ 1) & 1)) * 4, 4);
            for (size_t a = height + pair; a > 1; a >>= 1) job(layers[rd].nodes.u32() + (a ^ 1) * p2::OUT, p2::OUT);
        }
    }
    if (pf.size() + at > capacity) return RK_ERR_INTERNAL;  // the bound is exact: cannot happen
    {
        DevBuf d_jobs, d_out;
        RK_TRY(d_jobs.alloc(ctx, jobs.size() * sizeof(GatherJob)));
        RK_TRY(d_out.alloc(ctx, (size_t)at * 4 + 16));
        RK_HIP_TRY(ctx, hipMemcpyAsync(d_jobs.p, jobs.data(), jobs.size() * sizeof(GatherJob), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(gather_jobs_kernel, dim3((unsigned)((jobs.size() + 3) / 4)), dim3(256), 0, ctx->stream, d_out.u32(),
                           (const GatherJob*)d_jobs.p, jobs.size());
        RK_TRY(rk::post_launch(ctx, "gather_jobs_kernel"));
        std::memcpy(h_proof, pf.data(), pf.size() * 4);
        RK_TRY(d2h(ctx, h_proof + pf.size(), d_out.p, (size_t)at * 4));
    }
    *proof_words = pf.size() + at;
    return RK_OK;
}

int p3_prove(rk_ctx* ctx, const rk_p3_table* tables, uint32_t n_tables, const uint32_t* init, size_t n_init, uint32_t* h_proof,
             size_t capacity, size_t* proof_words) {
    ProofRun run(ctx, tables, n_tables);
    RK_TRY(rk_get_params(ctx, &run.par));
    RK_TRY(check_tables(run.par, tables, n_tables, true, run.lqd));
    for (size_t i = 0; i < n_init; i++)
        if (init[i] >= bb::P) return RK_ERR_INVALID;
    const size_t bound = proof_bound(run.par, tables, n_tables, run.lqd);
    if (capacity < bound) {
        *proof_words = bound;
        return RK_ERR_CAPACITY;
    }
    run.blow = run.par.blowup_log2;
    run.tm = rk_p3_timing{};
    const double t_start = now_ms();
    run.t_mark = t_start;
    run.pf.reserve(bound);
    run.ch.observe(init, n_init);
    RK_TRY(run.commit_traces());
    RK_TRY(run.permutation_traces());     // nothing of it without interactions
    const Ext alpha = run.ch.sample_ext();
    RK_TRY(run.quotients(alpha));
    RK_TRY(run.open());
    RK_TRY(run.fri());
    RK_TRY(run.queries(h_proof, capacity, proof_words));
    run.lap(run.tm.query);
    run.tm.total = (float)(now_ms() - t_start);
    return RK_OK;
}

// ---------------------------------------------------------------- verifier (host)
struct Reader {
    const uint32_t* p;
    size_t n, pos = 0;
    bool bad = false;
    const uint32_t* take(size_t k) {
        if (pos + k > n) {
            bad = true;
            return nullptr;
        }
        const uint32_t* q = p + pos;
        pos += k;
        return q;
    }
};
struct Selectors {
    Ext is_first, is_last, is_trans, inv_zeroifier;
};
Selectors selectors_at(const Ext& x, unsigned log_n, uint32_t root27m, uint32_t wm) {
    const Ext z_h = bb::sub(bb::pow(x, (uint64_t)1 << log_n, wm), bb::ext_one());
    const uint32_t g_inv = bb::inv(bb::pow(root27m, (uint64_t)1 << (27 - log_n)));
    Selectors s;
    s.is_first = bb::mul(z_h, bb::inv(bb::sub(x, bb::ext_one()), wm), wm);
    s.is_last = bb::mul(z_h, bb::inv(bb::sub(x, bb::ext_from(g_inv)), wm), wm);
    s.is_trans = bb::sub(x, bb::ext_from(g_inv));
    s.inv_zeroifier = bb::inv(z_h, wm);
    return s;
}
Ext load_ext(const uint32_t* p) { return Ext{{p[0], p[1], p[2], p[3]}}; }

// 0 accept; 1 malformed / short / trailing / non-canonical word, 2 shape mismatch, 3 constraint identity
// (OodEvaluationMismatch), 4 proof of work, 5 input opening, 6 commit-phase opening, 7 final polynomial, 8 the
// lookups' cumulative sums do not cancel
int p3_verify(const rk_params* params, const rk_p3_table* tables, uint32_t n_tables, const uint32_t* init, size_t n_init,
              const uint32_t* proof, size_t words, bool one_thread = false) {
    rk_params def;
    rk::params_preset(&def, RK_PRESET_SP1);
    const rk_params& par = params ? *params : def;
    rk::Sys sys;
    auto k = std::make_unique<p2::Any>();
    RK_TRY(rk::resolve_params(&par, &sys, k.get()));
    uint32_t lqd[MAX_TABLES];
    RK_TRY(check_tables(par, tables, n_tables, false, lqd));
    if (!proof || (n_init && !init)) return RK_ERR_INVALID;
    for (size_t i = 0; i < n_init; i++)
        if (init[i] >= bb::P) return RK_ERR_INVALID;
    for (size_t i = 0; i < words; i++)
        if (proof[i] >= bb::P) return 1;
    const unsigned blow = sys.blowup_log2;
    const uint32_t wm = sys.wm, shiftm = sys.shiftm;
    auto gen = [&](unsigned bits) { return bb::pow(sys.root27m, (uint64_t)1 << (27 - bits)); };
    Reader r{proof, words};
    const uint32_t* hdr = r.take(1 + (size_t)n_tables);
    if (!hdr || hdr[0] != n_tables) return 2;
    unsigned log_n[MAX_TABLES], log_max = 0;
    for (uint32_t t = 0; t < n_tables; t++) {
        log_n[t] = hdr[1 + t];
        if (log_n[t] < 1 || log_n[t] + blow > ntt::LAMBDA) return 2;
        if (tables[t].log_height && tables[t].log_height != log_n[t]) return 2;   // a height the statement pins (a 2^16-row range table)
        log_max = std::max(log_max, log_n[t] + blow);
    }
    Challenger ch(k.get());
    ch.observe(init, n_init);
    const uint32_t* troot = r.take(8);
    if (!troot) return 1;
    ch.observe(troot, 8);
    for (uint32_t t = 0; t < n_tables; t++) ch.observe(tables[t].public_values, tables[t].n_public);
    // lookups: the permutation challenges, the second commitment, the cumulative sums (which must cancel)
    uint32_t n_chal = 0, n_perm = 0, pwid[MAX_TABLES];
    unsigned log_pmax = 0;
    for (uint32_t t = 0; t < n_tables; t++) {
        pwid[t] = tables[t].air->perm_width;
        n_chal = std::max(n_chal, tables[t].air->n_chal);
        if (!pwid[t]) continue;
        n_perm++;
        log_pmax = std::max(log_pmax, log_n[t] + blow);
    }
    std::vector<uint32_t> pchal(n_chal);
    const uint32_t* proot = nullptr;
    const uint32_t* cumsum[MAX_TABLES] = {nullptr};
    if (n_perm) {
        const Ext pa = ch.sample_ext(), pb = ch.sample_ext();
        std::memcpy(pchal.data(), pa.c, 16);
        Ext cur = bb::ext_one();
        for (uint32_t j = 1; 4 * j < n_chal; j++) {
            std::memcpy(&pchal[4 * j], cur.c, 16);
            cur = bb::mul(cur, pb, wm);
        }
        proot = r.take(8);
        if (!proot) return 1;
        ch.observe(proot, 8);
        Ext total = bb::ext_zero();
        for (uint32_t t = 0; t < n_tables; t++) {
            if (!pwid[t]) continue;
            cumsum[t] = r.take(4);
            if (!cumsum[t]) return 1;
            ch.observe(cumsum[t], 4);
            total = bb::add(total, load_ext(cumsum[t]));
        }
        if (!bb::eq(total, bb::ext_zero())) return 8;
    }
    const Ext alpha = ch.sample_ext();
    const uint32_t* qroot = r.take(8);
    if (!qroot) return 1;
    ch.observe(qroot, 8);
    const Ext zeta = ch.sample_ext();
    const uint32_t *y_local[MAX_TABLES], *y_next[MAX_TABLES], *y_chunk[MAX_TABLES], *yp_local[MAX_TABLES], *yp_next[MAX_TABLES];
    for (uint32_t t = 0; t < n_tables; t++) {
        y_local[t] = r.take(4 * (size_t)tables[t].width);
        y_next[t] = r.take(4 * (size_t)tables[t].width);
        yp_local[t] = yp_next[t] = nullptr;
        if (pwid[t]) {
            yp_local[t] = r.take(4 * (size_t)pwid[t]);
            yp_next[t] = r.take(4 * (size_t)pwid[t]);
        }
        y_chunk[t] = r.take((size_t)16 << lqd[t]);
        if (r.bad) return 1;
    }
    for (uint32_t t = 0; t < n_tables; t++) {
        const unsigned kq = log_n[t] + lqd[t];
        const size_t qd = (size_t)1 << lqd[t], n = (size_t)1 << log_n[t];
        // quotient(zeta) = sum_i zps_i * sum_e x^e * chunk_i[e], zps_i = prod_{j != i} Z_j(zeta) / Z_j(first point of domain i)
        Ext quotient = bb::ext_zero();
        for (size_t i = 0; i < qd; i++) {
            Ext zp = bb::ext_one();
            const uint32_t first_i = bb::mul(shiftm, bb::pow(gen(kq), i));
            for (size_t j = 0; j < qd; j++) {
                if (j == i) continue;
                const uint32_t sj_inv = bb::inv(bb::mul(shiftm, bb::pow(gen(kq), j)));
                const Ext a = bb::sub(bb::pow(bb::scale(zeta, sj_inv), n, wm), bb::ext_one());
                const uint32_t b = bb::sub(bb::pow(bb::mul(first_i, sj_inv), n), bb::ONE);
                zp = bb::mul(zp, bb::scale(a, bb::inv(b)), wm);
            }
            for (int e = 0; e < 4; e++) {
                Ext mono = bb::ext_zero();
                mono.c[e] = bb::ONE;
                quotient = bb::add(quotient, bb::mul(bb::mul(zp, mono, wm), load_ext(y_chunk[t] + (i * 4 + e) * 4), wm));
            }
        }
        const Selectors s = selectors_at(zeta, log_n[t], sys.root27m, wm);
        const PermView pv{(const Ext*)yp_local[t], (const Ext*)yp_next[t], pchal.data(), cumsum[t]};
        const Ext folded = air_fold(*tables[t].air, (const Ext*)y_local[t], (const Ext*)y_next[t], tables[t].public_values, s.is_first,
                                    s.is_last, s.is_trans, alpha, wm, pv);
        if (!bb::eq(bb::mul(folded, s.inv_zeroifier, wm), quotient)) return 3;
    }
    const Ext alpha2 = ch.sample_ext();
    const uint32_t* nr = r.take(1);
    if (!nr) return 1;
    const uint32_t n_rounds = *nr;
    if (n_rounds != log_max - blow) return 2;
    const uint32_t* commits = r.take(8 * (size_t)n_rounds);
    if (r.bad) return 1;
    std::vector<Ext> betas(n_rounds);
    for (uint32_t rd = 0; rd < n_rounds; rd++) {
        ch.observe(commits + 8 * rd, 8);
        betas[rd] = ch.sample_ext();
    }
    const uint32_t* fp = r.take(4);
    const uint32_t* wit = r.take(1);
    if (r.bad) return 1;
    const Ext final_poly = load_ext(fp);
    ch.observe(fp, 4);
    if (!ch.check_witness(sys.pow_bits, *wit)) return 4;

    std::vector<uint32_t> th(n_tables), tw(n_tables), qh, qw, ph, pwd;
    size_t trow = 0, prow = 0;
    for (uint32_t t = 0; t < n_tables; t++) {
        th[t] = 1u << (log_n[t] + blow);
        tw[t] = tables[t].width;
        trow += tables[t].width;
        if (pwid[t]) {
            ph.push_back(th[t]);
            pwd.push_back(pwid[t]);
            prow += pwid[t];
        }
        for (uint32_t j = 0; j < (1u << lqd[t]); j++) {
            qh.push_back(th[t]);
            qw.push_back(4);
        }
    }
    const size_t qrow = 4 * qh.size();
    auto check_query = [&](uint32_t index, Reader r) -> int {
        // every table is in the trace and the quotient batch: both trees have the global maximum height; the permutation
        // batch only holds the tables with lookups
        const uint32_t* trows = r.take(trow);
        const uint32_t* tpath = r.take(8 * (size_t)log_max);
        const uint32_t* prows = n_perm ? r.take(prow) : nullptr;
        const uint32_t* ppath = n_perm ? r.take(8 * (size_t)log_pmax) : nullptr;
        const uint32_t* qrows = r.take(qrow);
        const uint32_t* qpath = r.take(8 * (size_t)log_max);
        if (r.bad) return 1;
        if (rk_mmcs_verify(&par, th.data(), tw.data(), n_tables, index, trows, tpath, troot) != 0) return 5;
        if (n_perm && rk_mmcs_verify(&par, ph.data(), pwd.data(), n_perm, index >> (log_max - log_pmax), prows, ppath, proot) != 0) return 5;
        if (rk_mmcs_verify(&par, qh.data(), qw.data(), (uint32_t)qh.size(), index, qrows, qpath, qroot) != 0) return 5;
        Ext rop[ntt::LAMBDA + 1], apow[ntt::LAMBDA + 1];
        bool used[ntt::LAMBDA + 1] = {false};
        for (unsigned i = 0; i <= ntt::LAMBDA; i++) rop[i] = bb::ext_zero(), apow[i] = bb::ext_one();
        auto reduce = [&](unsigned lh, uint32_t x, const Ext& z, const Ext& p_at_z, uint32_t p_at_x) {
            const Ext den = bb::sub(bb::ext_from(x), z);
            const Ext quot = bb::mul(bb::sub(bb::ext_from(p_at_x), p_at_z), bb::inv(den, wm), wm);
            rop[lh] = bb::add(rop[lh], bb::mul(apow[lh], quot, wm));
            apow[lh] = bb::mul(apow[lh], alpha2, wm);
        };
        size_t at = 0;
        for (uint32_t t = 0; t < n_tables; t++) {
            const unsigned lh = log_n[t] + blow;
            const uint32_t x = bb::mul(shiftm, bb::pow(gen(lh), bb::bitrev(index >> (log_max - lh), lh)));
            used[lh] = true;
            const Ext zn = bb::scale(zeta, gen(log_n[t]));
            for (uint32_t c = 0; c < tables[t].width; c++) reduce(lh, x, zeta, load_ext(y_local[t] + 4 * c), trows[at + c]);
            for (uint32_t c = 0; c < tables[t].width; c++) reduce(lh, x, zn, load_ext(y_next[t] + 4 * c), trows[at + c]);
            at += tables[t].width;
        }
        at = 0;
        for (uint32_t t = 0; t < n_tables; t++) {
            if (!pwid[t]) continue;
            const unsigned lh = log_n[t] + blow;
            const uint32_t x = bb::mul(shiftm, bb::pow(gen(lh), bb::bitrev(index >> (log_max - lh), lh)));
            const Ext zn = bb::scale(zeta, gen(log_n[t]));
            for (uint32_t c = 0; c < pwid[t]; c++) reduce(lh, x, zeta, load_ext(yp_local[t] + 4 * c), prows[at + c]);
            for (uint32_t c = 0; c < pwid[t]; c++) reduce(lh, x, zn, load_ext(yp_next[t] + 4 * c), prows[at + c]);
            at += pwid[t];
        }
        at = 0;
        for (uint32_t t = 0; t < n_tables; t++) {
            const unsigned lh = log_n[t] + blow;
            const uint32_t x = bb::mul(shiftm, bb::pow(gen(lh), bb::bitrev(index >> (log_max - lh), lh)));
            for (uint32_t j = 0; j < (1u << lqd[t]); j++, at += 4)
                for (int c = 0; c < 4; c++) reduce(lh, x, zeta, load_ext(y_chunk[t] + (j * 4 + c) * 4), qrows[at + c]);
        }
        Ext folded = bb::ext_zero();
        uint32_t idx = index;
        for (uint32_t rd = 0; rd < n_rounds; rd++) {
            const unsigned lfh = log_max - 1 - rd;
            if (used[lfh + 1]) folded = bb::add(folded, rop[lfh + 1]);
            const uint32_t* sib = r.take(4);
            const uint32_t* path = r.take(8 * (size_t)lfh);
            if (r.bad) return 1;
            uint32_t pair[8];
            std::memcpy(pair + 4 * (idx & 1), folded.c, 16);
            std::memcpy(pair + 4 * ((idx ^ 1) & 1), sib, 16);
            const uint32_t dh = 1u << lfh, dw = 8;
            static const uint32_t no_path[8] = {0};
            if (rk_mmcs_verify(&par, &dh, &dw, 1, idx >> 1, pair, lfh ? path : no_path, commits + 8 * rd) != 0) return 6;
            idx >>= 1;
            // fold_row: the line through (x0, e0) and (-x0, e1) at beta; x0 = g^bitrev(idx) in the subgroup of order 2^(lfh+1)
            const uint32_t x0 = bb::pow(gen(lfh + 1), bb::bitrev(idx, lfh));
            const Ext e0 = load_ext(pair), e1 = load_ext(pair + 4);
            const Ext slope = bb::scale(bb::sub(e1, e0), bb::inv(bb::sub(bb::neg(x0), x0)));
            folded = bb::add(e0, bb::mul(bb::sub(betas[rd], bb::ext_from(x0)), slope, wm));
        }
        if (!bb::eq(folded, final_poly)) return 7;
        return 0;
    };
    // the query positions come from the transcript one after the other; the queries themselves are independent and of
    // one size, so they are checked on a few threads (100 queries cost ~40 ms of Poseidon2 on one core)
    size_t per_query = trow + qrow + 16 * (size_t)log_max + (n_perm ? prow + 8 * (size_t)log_pmax : 0);
    for (uint32_t rd = 0; rd < n_rounds; rd++) per_query += 4 + 8 * (size_t)(log_max - 1 - rd);
    const size_t q0 = r.pos;
    if (q0 + per_query * sys.queries != words) return 1;   // short or trailing words
    std::vector<uint32_t> indices(sys.queries);
    for (uint32_t qi = 0; qi < sys.queries; qi++) indices[qi] = ch.sample_bits(log_max);
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned n_thr = sys.queries >= 16 && !one_thread ? std::max(1u, std::min(4u, hw / 2)) : 1u;
    std::vector<int> first_bad(n_thr, 0);
    std::vector<uint32_t> first_at(n_thr, 0xffffffffu);
    auto run = [&](unsigned t) {
        for (uint32_t qi = t; qi < sys.queries; qi += n_thr) {
            Reader rq{proof, words};
            rq.pos = q0 + per_query * qi;
            const int rc = check_query(indices[qi], rq);
            if (rc != 0) {
                first_bad[t] = rc;
                first_at[t] = qi;
                return;
            }
        }
    };
    if (n_thr == 1) {
        run(0);
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_thr; t++) pool.emplace_back(run, t);
        for (auto& th_ : pool) th_.join();
    }
    uint32_t best = 0xffffffffu;
    int rc = 0;
    for (unsigned t = 0; t < n_thr; t++)   // the verdict of the first failing query, as a sequential check would give it
        if (first_at[t] < best) {
            best = first_at[t];
            rc = first_bad[t];
        }
    return rc;
}

// ---------------------------------------------------------------- shards in flight (rk_p3_prove_shards)
struct ShardPool {
    std::mutex busy;                 // one batch at a time per device
    std::vector<rk_ctx*> ctxs;
    std::vector<std::vector<uint32_t>> key;   // the parameter set each context carries
    rk_ctx* uploader = nullptr;      // stages host-resident traces ahead of the provers (its own stream)
    std::mutex up_mu;                // the uploader's allocator: the feeder allocates, the provers free
    ~ShardPool() {
        for (rk_ctx* c : ctxs) (void)rk_ctx_destroy(c);
        if (uploader) (void)rk_ctx_destroy(uploader);
    }
};
std::mutex g_shard_mu;
std::map<int, std::shared_ptr<ShardPool>> g_shard_pools;

std::vector<uint32_t> params_key(const rk_params& p) {
    std::vector<uint32_t> k = {p.ext_w, p.root_2_27, p.coset_shift, p.p2_width, p.p2_m4, p.p2_pad_free, p.queries, p.blowup_log2,
                               p.fri_fold_log2, p.fri_min_degree, p.pow_bits};
    rk::Sys sys;
    auto any = std::make_unique<p2::Any>();
    if (rk::resolve_params(&p, &sys, any.get()) != RK_OK) return {};
    k.insert(k.end(), any->rc_ext(), any->rc_ext() + 8 * any->cells());
    k.insert(k.end(), any->rc_int(), any->rc_int() + any->rounds_partial());
    k.insert(k.end(), any->diag(), any->diag() + any->cells());
    return k;
}

int p3_prove_shards(const rk_p3_session_opts* opts, rk_p3_shard* shards, size_t n, size_t* failed_index) {
    if (failed_index) *failed_index = (size_t)-1;
    if (!opts || (n && !shards) || opts->batch < 1 || opts->batch > 16) return RK_ERR_INVALID;
    if (opts->n_devices < 0 || opts->n_devices > 64 || (opts->n_devices > 0 && !opts->devices)) return RK_ERR_INVALID;
    if (n == 0) return RK_OK;
    std::vector<int> devices;
    if (opts->n_devices > 0) devices.assign(opts->devices, opts->devices + opts->n_devices);
    else devices.push_back(opts->device);
    std::sort(devices.begin(), devices.end());   // pools are locked in ascending order
    if (std::adjacent_find(devices.begin(), devices.end()) != devices.end()) return RK_ERR_INVALID;
    int n_gpus = 0;
    if (hipGetDeviceCount(&n_gpus) != hipSuccess || n_gpus <= 0) return RK_ERR_NODEVICE;
    for (int d : devices)
        if (d < 0 || d >= n_gpus) return RK_ERR_INVALID;
    rk_params par;
    rk::params_preset(&par, RK_PRESET_SP1);
    if (opts->params) par = *opts->params;
    const std::vector<uint32_t> key = params_key(par);
    if (key.empty()) return RK_ERR_INVALID;
    for (size_t i = 0; i < n; i++)
        if (!shards[i].h_proof || (shards[i].n_init && !shards[i].init_words)) return RK_ERR_INVALID;

    std::vector<std::shared_ptr<ShardPool>> pools;
    {
        std::lock_guard<std::mutex> l(g_shard_mu);
        for (int d : devices) {
            auto& sp = g_shard_pools[d];
            if (!sp) sp = std::make_shared<ShardPool>();
            pools.push_back(sp);
        }
    }
    std::vector<std::unique_lock<std::mutex>> held;
    for (auto& p : pools) held.emplace_back(p->busy);
    const size_t per_dev = std::min<size_t>((size_t)opts->batch, n);
    for (size_t d = 0; d < devices.size(); d++) {
        ShardPool& pool = *pools[d];
        while (pool.ctxs.size() < per_dev) {
            rk_ctx* c = nullptr;
            RK_TRY(rk_ctx_create(devices[d], nullptr, &c));
            pool.ctxs.push_back(c);
            pool.key.emplace_back();
        }
        for (size_t j = 0; j < per_dev; j++) {
            if (pool.key[j] == key) continue;
            RK_TRY(rk_set_params(pool.ctxs[j], &par));
            pool.key[j] = key;
        }
        if (!pool.uploader) RK_TRY(rk_ctx_create(devices[d], nullptr, &pool.uploader));
    }
    std::atomic<size_t> next{0};
    std::mutex mu;
    std::condition_variable cv;
    std::deque<size_t> to_verify;      // proven shards waiting for a verifier thread
    size_t workers_left = 0;
    int status = RK_OK;
    size_t failed = (size_t)-1;
    auto fail = [&](int st, size_t i) {
        std::lock_guard<std::mutex> l(mu);
        if (status == RK_OK) {
            status = st;
            failed = i;
        }
        cv.notify_all();
    };
    // One feeder per device claims shards from the common queue and, for traces in host memory, uploads them into device
    // buffers AHEAD of the provers (a single thread per device keeps the PCIe link busy with one stream of copies, and a
    // proof never waits for its own upload); the provers then see on_device tables.  At most `per_dev + 1` staged shards
    // per device.
    struct Staged {
        size_t idx = 0;
        std::vector<rk_p3_table> tables;
        std::vector<void*> bufs;
    };
    struct DevQueue {
        std::deque<std::unique_ptr<Staged>> ready;
        size_t outstanding = 0;     // staged or being proven
        bool feeder_done = false;
    };
    std::vector<DevQueue> dq(devices.size());
    auto release = [&](ShardPool& pool, Staged& st) {
        std::lock_guard<std::mutex> l(pool.up_mu);
        for (void* b : st.bufs) (void)rk_free(pool.uploader, b);
        st.bufs.clear();
    };
    auto feeder = [&](size_t d) {
        ShardPool& pool = *pools[d];
        for (;;) {
            {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [&] { return status != RK_OK || dq[d].outstanding < per_dev + 1; });
                if (status != RK_OK) break;
            }
            const size_t i = next.fetch_add(1);
            if (i >= n) break;
            auto st = std::make_unique<Staged>();
            st->idx = i;
            const rk_p3_shard& sh = shards[i];
            int rc = sh.tables && sh.n_tables ? RK_OK : RK_ERR_INVALID;
            if (rc == RK_OK) st->tables.assign(sh.tables, sh.tables + sh.n_tables);
            for (uint32_t t = 0; rc == RK_OK && t < sh.n_tables; t++) {
                rk_p3_table& tb = st->tables[t];
                if (tb.on_device || !tb.trace || tb.log_height < 1 || tb.log_height > ntt::LAMBDA || tb.width == 0) continue;   // the prover refuses what is malformed
                const size_t bytes = ((size_t)tb.width << tb.log_height) * 4;
                void* buf = nullptr;
                {
                    std::lock_guard<std::mutex> l(pool.up_mu);
                    rc = rk_alloc(pool.uploader, bytes, &buf);
                }
                if (rc != RK_OK) break;
                st->bufs.push_back(buf);
                rc = rk_h2d(pool.uploader, buf, tb.trace, bytes);      // copy + wait on the uploader's own stream
                tb.trace = (const uint32_t*)buf;
                tb.on_device = 1;
            }
            if (rc != RK_OK) {
                release(pool, *st);
                fail(rc, i);
                break;
            }
            std::lock_guard<std::mutex> l(mu);
            dq[d].outstanding++;
            dq[d].ready.push_back(std::move(st));
            cv.notify_all();
        }
        std::lock_guard<std::mutex> l(mu);
        dq[d].feeder_done = true;
        cv.notify_all();
    };
    auto worker = [&](rk_ctx* ctx, size_t d) {
        ShardPool& pool = *pools[d];
        for (;;) {
            std::unique_ptr<Staged> st;
            {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [&] { return status != RK_OK || !dq[d].ready.empty() || dq[d].feeder_done; });
                if (status != RK_OK || dq[d].ready.empty()) break;
                st = std::move(dq[d].ready.front());
                dq[d].ready.pop_front();
            }
            const size_t i = st->idx;
            rk_p3_shard& sh = shards[i];
            int rc = RK_ERR_INTERNAL;
            try {
                rc = rk_p3_prove(ctx, st->tables.data(), sh.n_tables, sh.init_words, sh.n_init, sh.h_proof, sh.capacity_words, &sh.proof_words);
            } catch (...) {
            }
            release(pool, *st);
            {
                std::lock_guard<std::mutex> l(mu);
                dq[d].outstanding--;
                cv.notify_all();
            }
            if (rc != RK_OK) {
                fail(rc, i);
                break;
            }
            if (opts->verify) {   // host work (~40 ms for 100 queries): never on the thread that feeds the GPU
                std::lock_guard<std::mutex> l(mu);
                to_verify.push_back(i);
                cv.notify_all();
            }
        }
        std::lock_guard<std::mutex> l(mu);
        workers_left--;
        cv.notify_all();
    };
    auto verifier = [&]() {
        for (;;) {
            size_t i;
            {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [&] { return !to_verify.empty() || workers_left == 0 || status != RK_OK; });
                if (status != RK_OK || to_verify.empty()) return;
                i = to_verify.front();
                to_verify.pop_front();
            }
            int v = RK_ERR_INTERNAL;
            try {
                const rk_p3_shard& sh = shards[i];
                v = rk_p3_verify(&par, sh.tables, sh.n_tables, sh.init_words, sh.n_init, sh.h_proof, sh.proof_words);
            } catch (...) {
            }
            if (v != 0) {
                fail(RK_ERR_VERIFY, i);
                return;
            }
        }
    };
    std::vector<std::thread> threads;
    workers_left = devices.size() * per_dev;
    for (size_t d = 0; d < devices.size(); d++) {
        threads.emplace_back(feeder, d);
        for (size_t j = 0; j < per_dev; j++) threads.emplace_back(worker, pools[d]->ctxs[j], d);
    }
    if (opts->verify) {
        const unsigned hw = std::thread::hardware_concurrency();
        size_t nv = std::min<size_t>(std::min<size_t>(16, 4 * devices.size()), std::max<unsigned>(1, hw / 4));
        nv = std::min(nv, n);
        for (size_t v = 0; v < nv; v++) threads.emplace_back(verifier);
    }
    for (auto& t : threads) t.join();
    for (size_t d = 0; d < devices.size(); d++)     // a failed run leaves staged shards nobody proved
        for (auto& st : dq[d].ready) release(*pools[d], *st);
    if (failed_index) *failed_index = failed;
    return status;
}

}  // namespace

namespace rk {
void p3_release_pools() {
    std::map<int, std::shared_ptr<ShardPool>> pools;
    {
        std::lock_guard<std::mutex> l(g_shard_mu);
        pools.swap(g_shard_pools);
    }
    for (auto& kv : pools) {
        std::lock_guard<std::mutex> l(kv.second->busy);   // wait for a running batch
        std::vector<rk_ctx*> ctxs;
        ctxs.swap(kv.second->ctxs);
        for (rk_ctx* c : ctxs) (void)rk_ctx_destroy(c);
    }
}
}  // namespace rk

extern "C" {

int rk_p3_prove_shards(const rk_p3_session_opts* opts, rk_p3_shard* shards, size_t n, size_t* failed_index) {
    RK_GUARD_BEGIN
    return p3_prove_shards(opts, shards, n, failed_index);
    RK_GUARD_END
}

size_t rk_p3_proof_bound_words(const rk_params* params, const rk_p3_table* tables, uint32_t n_tables) {
    rk_params def;
    rk::params_preset(&def, RK_PRESET_SP1);
    const rk_params& par = params ? *params : def;
    rk::Sys sys;
    auto k = std::make_unique<p2::Any>();
    if (rk::resolve_params(&par, &sys, k.get()) != RK_OK) return 0;
    uint32_t lqd[MAX_TABLES];
    if (check_tables(par, tables, n_tables, false, lqd) != RK_OK) return 0;
    for (uint32_t t = 0; t < n_tables; t++)
        if (tables[t].log_height < 1 || tables[t].log_height + par.blowup_log2 > ntt::LAMBDA) return 0;
    return proof_bound(par, tables, n_tables, lqd);
}

int rk_p3_prove(rk_ctx* ctx, const rk_p3_table* tables, uint32_t n_tables, const uint32_t* init_words, size_t n_init, uint32_t* h_proof,
                size_t capacity_words, size_t* proof_words) {
    RK_GUARD_BEGIN
    if (!ctx || !h_proof || !proof_words || (n_init && !init_words)) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rc = p3_prove(ctx, tables, n_tables, init_words, n_init, h_proof, capacity_words, proof_words);
    if (rc != RK_OK) (void)hipStreamSynchronize(ctx->stream);   // scoped buffers are back in the pool: nothing may still read them
    return rc;
    RK_GUARD_END
}

int rk_p3_verify(const rk_params* params, const rk_p3_table* tables, uint32_t n_tables, const uint32_t* init_words, size_t n_init,
                 const uint32_t* proof, size_t proof_words) {
    RK_GUARD_BEGIN
    return p3_verify(params, tables, n_tables, init_words, n_init, proof, proof_words);
    RK_GUARD_END
}

int rk_p3_verify_hashes(const rk_params* params, const rk_p3_table* tables, uint32_t n_tables, const uint32_t* init_words, size_t n_init,
                        const uint32_t* proof, size_t proof_words, uint32_t* states, size_t capacity, size_t* n_permutations) {
    RK_GUARD_BEGIN
    if (!n_permutations || (capacity && !states)) return RK_ERR_INVALID;
    rk_params def;
    rk::params_preset(&def, RK_PRESET_SP1);
    const size_t w = (params ? params : &def)->p2_width;
    std::vector<uint32_t> log;
    struct Scope {   // the log is this thread's for the duration of the check, also when the verifier throws
        explicit Scope(std::vector<uint32_t>* l) { p2::g_permute_log = l; }
        ~Scope() { p2::g_permute_log = nullptr; }
    };
    int verdict;
    {
        Scope scope(&log);
        verdict = p3_verify(params, tables, n_tables, init_words, n_init, proof, proof_words, /*one_thread=*/true);
    }
    if (verdict < 0 || (w != 16 && w != 24)) return verdict < 0 ? verdict : RK_ERR_INVALID;
    *n_permutations = log.size() / w;
    if (*n_permutations > capacity) return RK_ERR_CAPACITY;
    std::memcpy(states, log.data(), log.size() * 4);
    return verdict;
    RK_GUARD_END
}

int rk_p3_last_timing(rk_ctx* ctx, rk_p3_timing* out) {
    if (!ctx || !out) return RK_ERR_INVALID;
    *out = ctx->p3_timing;
    return RK_OK;
}

}  // extern "C"
