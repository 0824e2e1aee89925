// TEST INFRASTRUCTURE: runs the product's host/device-shared code (bb.hpp,
// ntt_core.hpp, poseidon2_core.hpp) on the CPU, one emulated lane at a time, so
// the tile/index/twiddle logic of the HIP kernels can be checked against the
// oracle without a GPU.  The GPU kernels are these same phase functions with
// __syncthreads() between phases (raiko_amd/csrc/kernels_ntt.hip).
#include <cstring>
#include <vector>

#include "bb.hpp"
#include "ntt_core.hpp"
#include "ntt_r16.hpp"
#include "poseidon2_core.hpp"
#include "poseidon2_consts.inc"

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
// mirrors ntt_r16_kernel / launch_r16 of kernels_ntt.hip
template <bool FWD, bool CONTIG>
void run_r16(const r16::Args& a, size_t count) {
    size_t blocks = count * (a.n >> r16::TILE_LOG);
    r16::Sched sc = FWD ? r16::sched_dit(a.g, a.expand_bits) : r16::sched_dif(a.g);
    std::vector<uint32_t> lds(r16::LDS_WORDS);
    for (size_t blk = 0; blk < blocks; blk++) {
        r16::Tile t = r16::tile_of(a, blk);
        for (unsigned tid = 0; tid < r16::NTHR; tid++) {
            if (CONTIG) {
                if (FWD) r16::load_fwd_contig(a, t, lds.data(), tid);
                else r16::load_rev_contig(a, g_tb, t, lds.data(), tid);
            } else {
                r16::load_plain(a, t, lds.data(), tid);
            }
        }
        for (unsigned rd = 0; rd < sc.n; rd++) {
            unsigned ls = sc.ls[rd];
            // all lanes read, then all lanes write: a round is in place per lane, so lane order is free
            for (unsigned tid = 0; tid < r16::NTHR; tid++) {
                r16::RoundIdx x = r16::round_idx(tid, a.g, ls);
                uint32_t v[16];
                r16::round_read(v, lds.data(), x);
                const uint32_t* tw = g_tb.small[FWD ? 0 : 1];
                switch (sc.nst[rd]) {
                    case 4: FWD ? r16::round_dit<4>(v, tw, ls, x.rlow) : r16::round_dif<4>(v, tw, ls, x.rlow); break;
                    case 3: FWD ? r16::round_dit<3>(v, tw, ls, x.rlow) : r16::round_dif<3>(v, tw, ls, x.rlow); break;
                    case 2: FWD ? r16::round_dit<2>(v, tw, ls, x.rlow) : r16::round_dif<2>(v, tw, ls, x.rlow); break;
                    default: FWD ? r16::round_dit<1>(v, tw, ls, x.rlow) : r16::round_dif<1>(v, tw, ls, x.rlow); break;
                }
                r16::round_write(v, lds.data(), x);
            }
        }
        for (unsigned tid = 0; tid < r16::NTHR; tid++) {
            if (CONTIG) {
                if (FWD) r16::store_fwd_contig(a, g_tb, t, lds.data(), tid);
                else r16::store_rev_contig(a, g_tb, t, lds.data(), tid);
            } else {
                r16::store_plain<FWD>(a, t, lds.data(), tid);
            }
        }
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
    if (max_tile_log == ntt::MAX_TILE_LOG && r16::usable(k, 0, aligned16(io, io))) {
        r16::Args a{};
        a.dst = io; a.src = io; a.n = a.n_src = size; a.k = k; a.g = k - r16::TILE_LOG;
        run_r16<false, false>(a, count);
        a.g_outer = a.g; a.g = r16::TILE_LOG; a.scale = scale; a.zk = fuse_zk ? 1 : 0;
        run_r16<false, true>(a, count);
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
    if (max_tile_log == ntt::MAX_TILE_LOG && r16::usable(k, expand_bits, aligned16(out, in))) {
        r16::Args a{};
        a.dst = out; a.src = in; a.n = size; a.n_src = in_size; a.k = k; a.g = r16::TILE_LOG;
        a.g_outer = k - r16::TILE_LOG; a.expand_bits = expand_bits;
        run_r16<true, true>(a, count);
        a.src = out; a.n_src = size; a.g = k - r16::TILE_LOG; a.g_outer = 0; a.expand_bits = 0;
        run_r16<true, false>(a, count);
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

}  // extern "C"
