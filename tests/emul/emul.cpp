// TEST INFRASTRUCTURE: runs the product's host/device-shared code (bb.hpp,
// ntt_core.hpp, poseidon2_core.hpp) on the CPU, one emulated lane at a time, so
// the tile/index/twiddle logic of the HIP kernels can be checked against the
// oracle without a GPU.  The GPU kernels are these same phase functions with
// __syncthreads() between phases (raiko_amd/csrc/kernels_ntt.hip).
#include <algorithm>
#include <cstring>
#include <vector>

#include "bb.hpp"
#include "ntt_core.hpp"
#include "ntt_fused.hpp"
#include "poseidon2_any.hpp"
#include "poseidon2_consts.inc"
#include "p3_kernels.hpp"

namespace {
std::vector<uint32_t> g_tables;
ntt::Tables g_tb;
void ensure_tables() {
    if (!g_tables.empty()) return;
    g_tables.resize(ntt::table_layout().total);
    ntt::fill_tables(g_tables.data());
    g_tb = ntt::tables_at(g_tables.data());
}
p2::Consts consts() {
    p2::Consts k;
    std::memcpy(k.rc_ext, P2_RC_EXT_MONT, sizeof k.rc_ext);
    std::memcpy(k.rc_int, P2_RC_INT_MONT, sizeof k.rc_int);
    std::memcpy(k.diag, P2_INT_DIAG_MONT, sizeof k.diag);
    p2::derive(k);
    return k;
}
// mirrors launch_pass / ntt_pass_kernel(_u) of kernels_ntt.hip, one emulated lane at a time
template <bool FWD, bool VEC>
void run_pass_u(const ntt::PassArgs& a, size_t count) {
    constexpr int EPT = 16;
    size_t tile = (size_t)1 << (a.g + a.logT);
    size_t blocks = count * (a.n >> (a.g + a.logT));
    unsigned nthr = (unsigned)(tile / EPT);
    std::vector<uint32_t> lds(tile);
    for (size_t blk = 0; blk < blocks; blk++) {
        ntt::Tile t = ntt::tile_of(a, blk);
        if (FWD) {
            for (unsigned tid = 0; tid < nthr; tid++) ntt::fwd_load_t<EPT, VEC>(a, g_tb, t, lds.data(), tid, nthr);
            for (unsigned s = a.expand_bits; s < a.g; s++)
                for (unsigned tid = 0; tid < nthr; tid++) ntt::fwd_stage_t<EPT / 2>(a, g_tb, lds.data(), tid, nthr, s);
            for (unsigned tid = 0; tid < nthr; tid++) ntt::fwd_store_t<EPT, VEC>(a, t, lds.data(), tid, nthr);
        } else {
            for (unsigned tid = 0; tid < nthr; tid++) ntt::rev_load_t<EPT, VEC>(a, t, lds.data(), tid, nthr);
            for (unsigned s = 0; s < a.g; s++)
                for (unsigned tid = 0; tid < nthr; tid++) ntt::rev_stage_t<EPT / 2>(a, g_tb, lds.data(), tid, nthr, s);
            for (unsigned tid = 0; tid < nthr; tid++) ntt::rev_store_t<EPT, VEC>(a, g_tb, t, lds.data(), tid, nthr);
        }
    }
}
template <bool FWD>
void run_pass(const ntt::PassArgs& a, size_t count, unsigned nthr) {
    if (ntt::can_unroll(a, 16)) {
        if (ntt::can_vec(a)) run_pass_u<FWD, true>(a, count);
        else run_pass_u<FWD, false>(a, count);
        return;
    }
    size_t tile = (size_t)1 << (a.g + a.logT);
    size_t blocks = count * (a.n >> (a.g + a.logT));
    std::vector<uint32_t> lds(tile);
    for (size_t blk = 0; blk < blocks; blk++) {
        ntt::Tile t = ntt::tile_of(a, blk);
        if (FWD) {
            for (unsigned tid = 0; tid < nthr; tid++) ntt::fwd_load(a, g_tb, t, lds.data(), tid, nthr);
            for (unsigned s = a.expand_bits; s < a.g; s++)
                for (unsigned tid = 0; tid < nthr; tid++) ntt::fwd_stage(a, g_tb, lds.data(), tid, nthr, s);
            for (unsigned tid = 0; tid < nthr; tid++) ntt::fwd_store(a, t, lds.data(), tid, nthr);
        } else {
            for (unsigned tid = 0; tid < nthr; tid++) ntt::rev_load(a, t, lds.data(), tid, nthr);
            for (unsigned s = 0; s < a.g; s++)
                for (unsigned tid = 0; tid < nthr; tid++) ntt::rev_stage(a, g_tb, lds.data(), tid, nthr, s);
            for (unsigned tid = 0; tid < nthr; tid++) ntt::rev_store(a, g_tb, t, lds.data(), tid, nthr);
        }
    }
}
// mirrors the nf_* kernels of kernels_ntt.hip (ntt_fused.hpp): phases in program order, every lane
// of a phase before the next phase (= the barriers of the GPU kernels)
std::vector<uint32_t> nf_table(int kind, unsigned k) {
    size_t n = (size_t)1 << k;
    std::vector<uint32_t> t(n);
    uint32_t scale = bb::inv(bb::encode((uint32_t)n));
    for (size_t i = 0; i < n; i++) t[i] = kind == 2 ? nf::zk_entry(g_tb, k, scale, i) : nf::fs_entry(g_tb, k, kind, i);
    return t;
}
template <class F>
void all_lanes(F&& f) {
    for (unsigned tid = 0; tid < nf::NTHR; tid++) f(tid);
}
template <bool FWD, int G>
void run_nf_strided_g(const nf::Args& a) {
    size_t blocks = (size_t)a.count * (a.n >> nf::TILE_LOG);
    std::vector<uint32_t> lds(nf::LDS_WORDS);
    for (size_t blk = 0; blk < blocks; blk++) {
        nf::STile<G> t = nf::stile_of<G>(a, blk);
        if (FWD) {
            if (G > 4) all_lanes([&](unsigned tid) { nf::fwd_strided_a<G>(a, g_tb, t, lds.data(), tid); });
            // in-place pass: every lane's loads of a phase come before any lane's stores on the GPU only
            // because tiles are disjoint and phase b's loads (G == 4) read what nobody else writes
            all_lanes([&](unsigned tid) { nf::fwd_strided_b<G>(a, g_tb, t, lds.data(), tid); });
        } else {
            all_lanes([&](unsigned tid) { nf::inv_strided_a<G>(a, g_tb, t, lds.data(), tid); });
            if (G > 4) all_lanes([&](unsigned tid) { nf::inv_strided_b<G>(a, g_tb, t, lds.data(), tid); });
        }
    }
}
template <bool FWD>
void run_nf_strided(const nf::Args& a) {
    switch (a.k - nf::TILE_LOG) {
        case 4: run_nf_strided_g<FWD, 4>(a); break;
        case 5: run_nf_strided_g<FWD, 5>(a); break;
        case 6: run_nf_strided_g<FWD, 6>(a); break;
        case 7: run_nf_strided_g<FWD, 7>(a); break;
        default: run_nf_strided_g<FWD, 8>(a); break;
    }
}
void run_nf_fwd_contig(const nf::Args& a, unsigned expand_bits) {
    size_t blocks = (size_t)a.count * (a.n >> nf::TILE_LOG);
    std::vector<uint32_t> lds(nf::LDS_WORDS);
    for (size_t blk = 0; blk < blocks; blk++) {
        nf::CTile t = nf::ctile_of(a, blk, expand_bits);
        if (expand_bits == 2) {
            all_lanes([&](unsigned tid) { nf::fwd_contig_a(a, g_tb, t, lds.data(), tid); });
        } else {
            if (expand_bits == 1) all_lanes([&](unsigned tid) { nf::fwd_contig1_a(a, g_tb, t, lds.data(), tid); });
            else all_lanes([&](unsigned tid) { nf::fwd_contig0_a(a, g_tb, t, lds.data(), tid); });
            all_lanes([&](unsigned tid) { nf::fwd_contig0_b(g_tb, lds.data(), tid); });
        }
        all_lanes([&](unsigned tid) { nf::fwd_contig_b(g_tb, lds.data(), tid); });
        all_lanes([&](unsigned tid) { nf::fwd_contig_c(a, g_tb, t, lds.data(), tid); });
    }
}
void run_nf_inv_contig(const nf::Args& a) {
    size_t blocks = (size_t)a.count * (a.n >> nf::TILE_LOG);
    std::vector<uint32_t> lds(nf::LDS_WORDS);
    for (size_t blk = 0; blk < blocks; blk++) {
        nf::CTile t = nf::ctile_of(a, blk, 0);
        all_lanes([&](unsigned tid) { nf::inv_contig_a(a, g_tb, t, lds.data(), tid); });
        all_lanes([&](unsigned tid) { nf::inv_contig_mid<6>(g_tb, lds.data(), tid); });
        all_lanes([&](unsigned tid) { nf::inv_contig_mid<2>(g_tb, lds.data(), tid); });
        all_lanes([&](unsigned tid) { nf::inv_contig_d(g_tb, lds.data(), tid); });
        all_lanes([&](unsigned tid) { nf::inv_contig_e(a, t, lds.data(), tid); });
    }
}
inline bool aligned16(const void* p, const void* q) { return ((((uintptr_t)p) | ((uintptr_t)q)) & 15) == 0; }
unsigned log2u(size_t n) {
    unsigned k = 0;
    while (((size_t)1 << k) < n) k++;
    return k;
}
}  // namespace

extern "C" {

// mirrors rk::ntt_reverse (kernels_ntt.hip)
int emul_ntt_reverse(uint32_t* io, size_t size, size_t count, int fuse_zk, unsigned max_tile_log, unsigned nthr) {
    ensure_tables();
    unsigned k = log2u(size);
    if (k == 0) return 0;
    uint32_t scale = bb::inv(bb::encode((uint32_t)size));
    if (max_tile_log == ntt::MAX_TILE_LOG && nf::usable(k, 0, aligned16(io, io))) {
        nf::Args a{};
        a.dst = io; a.src = io; a.n = a.n_src = size; a.k = k; a.count = (unsigned)count;
        run_nf_strided<false>(a);
        std::vector<uint32_t> fs = nf_table(1, k), zk;
        a.fs = fs.data();
        if (fuse_zk) { zk = nf_table(2, k); a.zk = zk.data(); }
        a.scale = scale;
        run_nf_inv_contig(a);
        return 102;
    }
    ntt::Plan plan = ntt::make_plan(k, max_tile_log);
    for (unsigned p = 0; p < plan.npass; p++) {
        ntt::PassArgs a{};
        a.dst = io; a.src = io; a.n = size; a.n_src = size;
        a.mu = plan.mu[p]; a.g = plan.g[p]; a.logT = plan.logT[p];
        bool last = p + 1 == plan.npass;
        a.scale = last ? scale : 0;
        a.zk_bits = (last && fuse_zk) ? k : 0;
        run_pass<false>(a, count, nthr);
    }
    return (int)plan.npass;
}
// mirrors rk::ntt_forward
int emul_ntt_forward(uint32_t* out, const uint32_t* in, size_t in_size, size_t count, unsigned expand_bits,
                     unsigned max_tile_log, unsigned nthr) {
    ensure_tables();
    size_t size = in_size << expand_bits;
    unsigned k = log2u(size);
    if (k == 0) { std::memcpy(out, in, count * 4); return 0; }
    if (max_tile_log == ntt::MAX_TILE_LOG && nf::usable(k, expand_bits, aligned16(out, in))) {
        nf::Args a{};
        a.dst = out; a.src = in; a.n = size; a.n_src = in_size; a.k = k; a.count = (unsigned)count;
        std::vector<uint32_t> fs = nf_table(0, k);
        a.fs = fs.data();
        run_nf_fwd_contig(a, expand_bits);
        a.src = out; a.n_src = size;
        run_nf_strided<true>(a);
        return 102;
    }
    if (expand_bits == 0 && out != in) { std::memcpy(out, in, count * size * 4); in = out; }
    ntt::Plan plan = ntt::make_plan(k, max_tile_log);
    for (unsigned pi = plan.npass; pi-- > 0;) {
        ntt::PassArgs a{};
        bool first = pi + 1 == plan.npass;
        a.dst = out; a.src = first ? in : out; a.n = size; a.n_src = first ? in_size : size;
        a.mu = plan.mu[pi]; a.g = plan.g[pi]; a.logT = plan.logT[pi];
        a.expand_bits = first ? expand_bits : 0;
        if (first && expand_bits > a.g) return -1;
        run_pass<true>(a, count, nthr);
    }
    return (int)plan.npass;
}
// any shipped instance: width 24 / 16, external 4x4 block 0 / 1, caller's tables (Montgomery form)
void emul_poseidon2_permute_cfg(uint32_t* cells, int width, int m4, const uint32_t* rc_ext, const uint32_t* rc_int,
                                const uint32_t* diag) {
    static p2::Any k;
    k.set(width, m4, false, rc_ext, rc_int, diag);
    k.permute(cells);
}
// the extension product / inverse for a caller-chosen W (Montgomery form)
void emul_ext_mul_w(const uint32_t* a, const uint32_t* b, uint32_t wm, uint32_t* out) {
    bb::Ext x, y;
    std::memcpy(x.c, a, 16);
    std::memcpy(y.c, b, 16);
    bb::Ext r = bb::mul(x, y, wm);
    std::memcpy(out, r.c, 16);
}
void emul_ext_inv_w(const uint32_t* a, uint32_t wm, uint32_t* out) {
    bb::Ext x;
    std::memcpy(x.c, a, 16);
    bb::Ext r = bb::inv(x, wm);
    std::memcpy(out, r.c, 16);
}
void emul_poseidon2_permute(uint32_t* cells) {
    p2::Consts k = consts();
    p2::permute(cells, k);
}
// the same permutation with caller-supplied constants (Montgomery form): rk_set_poseidon2_params' path
void emul_poseidon2_permute_with(uint32_t* cells, const uint32_t* rc_ext, const uint32_t* rc_int, const uint32_t* diag) {
    static p2::Consts k;  // ~12 KiB
    std::memcpy(k.rc_ext, rc_ext, sizeof k.rc_ext);
    std::memcpy(k.rc_int, rc_int, sizeof k.rc_int);
    std::memcpy(k.diag, diag, sizeof k.diag);
    p2::derive(k);
    p2::permute(cells, k);
}
uint32_t emul_mul(uint32_t a, uint32_t b) { return bb::mul(a, b); }
uint32_t emul_add(uint32_t a, uint32_t b) { return bb::add(a, b); }
uint32_t emul_sub(uint32_t a, uint32_t b) { return bb::sub(a, b); }
uint32_t emul_inv(uint32_t a) { return bb::inv(a); }
uint32_t emul_encode(uint32_t a) { return bb::encode(a); }
uint32_t emul_decode(uint32_t a) { return bb::decode(a); }
void emul_ext_mul(const uint32_t* a, const uint32_t* b, uint32_t* o) {
    bb::Ext x, y;
    std::memcpy(x.c, a, 16); std::memcpy(y.c, b, 16);
    bb::Ext r = bb::mul(x, y);
    std::memcpy(o, r.c, 16);
}
void emul_ext_inv(const uint32_t* a, uint32_t* o) {
    bb::Ext x;
    std::memcpy(x.c, a, 16);
    bb::Ext r = bb::inv(x);
    std::memcpy(o, r.c, 16);
}
uint32_t emul_pow3(uint32_t e) { ensure_tables(); return ntt::pow3(g_tb, e); }

// mirrors perm_entries_kernel (p3.hip): per workgroup, phase 1 (staging) for every lane, then phase 2 (one row per lane)
void emul_perm_entries(uint32_t* out, const uint32_t* trace, const uint32_t* desc, size_t n, size_t w, uint32_t n_chal, uint32_t n_lookups,
                       uint32_t wm, uint32_t n_used, uint32_t desc_words) {
    p3k::PermArgs a{out, trace, desc, n, w, n_chal, n_lookups, wm, n_used, desc_words};
    std::vector<uint32_t> tile((size_t)(n_used ? n_used : 1) * p3k::PERM_LD, 0xdeadbeefu);
    for (size_t blk = 0; blk < (n + p3k::PERM_ROWS - 1) / p3k::PERM_ROWS; blk++) {
        std::fill(tile.begin(), tile.end(), 0xdeadbeefu);      // a read of a slot nobody staged shows
        for (unsigned tid = 0; tid < (unsigned)p3k::PERM_ROWS; tid++) p3k::perm_stage(a, blk, tid, tile.data());
        for (unsigned tid = 0; tid < (unsigned)p3k::PERM_ROWS; tid++) p3k::perm_row(a, blk, tid, tile.data());
    }
}
// mirrors p2_chip_trace_kernel (p3_air.hip)
int emul_p2_chip_rows(uint32_t* out, const uint32_t* in, const uint32_t* mult, const uint32_t* tab, size_t n, int width16, int m4) {
    p3k::P2ChipLayout L;
    L.W = width16 ? 16 : 24;
    L.RP = width16 ? 13 : 21;
    L.width = L.W + 16 * L.W + 2 * L.RP - 1 + L.W + 1;
    for (size_t r = 0; r < n; r++) {
        uint32_t* row = out + r * L.width;
        const uint32_t m = mult ? mult[r] : bb::ONE;
        if (width16 && m4) p3k::chip_row<16, 13, 1>(row, in + r * 16, m, tab, L);
        else if (width16) p3k::chip_row<16, 13, 0>(row, in + r * 16, m, tab, L);
        else if (m4) p3k::chip_row<24, 21, 1>(row, in + r * 24, m, tab, L);
        else p3k::chip_row<24, 21, 0>(row, in + r * 24, m, tab, L);
    }
    return (int)L.width;
}

}  // extern "C"
