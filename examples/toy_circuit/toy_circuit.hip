// GPU side of the toy circuit (see toy_circuit.h): the two rk_circuit_hooks as HIP kernels on the
// prover's stream, and the verifier's poly_ext on the host.  Field arithmetic comes from the
// library's own host/device header (raiko_amd/csrc/bb.hpp).
#include <hip/hip_runtime.h>

#include <cstring>

#include "bb.hpp"
#include "toy_circuit.h"

namespace {

using bb::Ext;
constexpr int MAX_EXTRA = 64;  // accum columns beyond a0..a3

struct ColMix {
    uint32_t m[MAX_EXTRA];  // multiplier of accum column 4 + j
};

bool shape_ok(const uint32_t* gs, uint32_t n_mix) {
    return gs[0] >= 4 && gs[0] - 4 <= MAX_EXTRA && gs[1] >= 3 && gs[2] >= 4 && n_mix >= 4;
}

__device__ __forceinline__ Ext add_base(Ext a, uint32_t b) {
    a.c[0] = bb::add(a.c[0], b);
    return a;
}

// t[i] = (m + d2[i]) / (m + d3[i])
__global__ void ratio_kernel(uint32_t* t_ext, const uint32_t* data, size_t n, Ext m, uint32_t wm) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ext r = bb::mul(add_base(m, data[2 * n + i]), bb::inv(add_base(m, data[3 * n + i]), wm), wm);
    *reinterpret_cast<uint4*>(t_ext + i * 4) = make_uint4(r.c[0], r.c[1], r.c[2], r.c[3]);
}
// accum columns from the running product and the plain data columns
__global__ void fill_accum_kernel(uint32_t* accum, const uint32_t* prod_ext, const uint32_t* data, size_t n, uint32_t wa,
                                  uint32_t wd, ColMix cm) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 a = *reinterpret_cast<const uint4*>(prod_ext + i * 4);
    accum[i] = a.x;
    accum[n + i] = a.y;
    accum[2 * n + i] = a.z;
    accum[3 * n + i] = a.w;
    for (uint32_t k = 4; k < wa; k++) accum[(size_t)k * n + i] = bb::mul(cm.m[k - 4], data[(size_t)(k % wd) * n + i]);
}

struct CheckArgs {
    const uint32_t *acc, *code, *data;
    size_t d;
    uint32_t wa, wd;
    Ext poly_mix, m;
    uint32_t wm;         // Montgomery form of the extension's W (rk_params.ext_w)
    uint32_t blow;       // log2 of the blow-up (rk_params.blowup_log2): one row back = 1 << blow points back
    uint32_t inv_den[16];
    ColMix cm;
};
__global__ void eval_check_kernel(uint32_t* check, CheckArgs a) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.d) return;
    const size_t d = a.d;
    const size_t b1 = (i + d - ((size_t)1 << a.blow)) % d, b2 = (i + d - ((size_t)2 << a.blow)) % d;  // one / two rows back
    const uint32_t one = bb::ONE;
    uint32_t c0 = a.code[i], c1 = a.code[d + i], c2 = a.code[2 * d + i];
    uint32_t d0 = a.data[i], d0b1 = a.data[b1], d0b2 = a.data[b2];
    uint32_t d1 = a.data[d + i], d2 = a.data[2 * d + i], d3 = a.data[3 * d + i];
    Ext A{{a.acc[i], a.acc[d + i], a.acc[2 * d + i], a.acc[3 * d + i]}};
    Ext Ab{{a.acc[b1], a.acc[d + b1], a.acc[2 * d + b1], a.acc[3 * d + b1]}};
    // base-valued constraints first, then the extension-valued ones
    uint32_t k0 = bb::mul(bb::sub(bb::sub(one, c0), c1), bb::sub(bb::sub(d0, d0b1), d0b2));
    uint32_t k1 = bb::sub(d1, bb::mul(d0, d0b1));
    Ext prev = add_base(bb::scale(Ab, bb::sub(one, c0)), c0);
    Ext k2 = bb::sub(bb::mul(A, add_base(a.m, d3), a.wm), bb::mul(prev, add_base(a.m, d2), a.wm));
    Ext k3 = bb::scale(add_base(A, bb::neg(one)), c2);
    Ext pw = a.poly_mix;
    Ext tot = bb::ext_from(k0);
    tot = bb::add(tot, bb::scale(pw, k1));
    pw = bb::mul(pw, a.poly_mix, a.wm);
    tot = bb::add(tot, bb::mul(pw, k2, a.wm));
    pw = bb::mul(pw, a.poly_mix, a.wm);
    tot = bb::add(tot, bb::mul(pw, k3, a.wm));
    for (uint32_t k = 4; k < a.wa; k++) {
        pw = bb::mul(pw, a.poly_mix, a.wm);
        uint32_t kk = bb::sub(a.acc[(size_t)k * d + i], bb::mul(a.cm.m[k - 4], a.data[(size_t)(k % a.wd) * d + i]));
        tot = bb::add(tot, bb::scale(pw, kk));
    }
    tot = bb::scale(tot, a.inv_den[i & ((1u << a.blow) - 1)]);
#pragma unroll
    for (int e = 0; e < 4; e++) check[(size_t)e * d + i] = tot.c[e];
}

ColMix col_mix(const rk_circuit_view* v) {
    ColMix cm{};
    for (uint32_t k = 4; k < v->group_size[0]; k++) cm.m[k - 4] = v->mix[k % v->n_mix];
    return cm;
}

int accumulate(void*, const rk_circuit_view* v, uint32_t* d_accum) {
    if (!shape_ok(v->group_size, v->n_mix) || !v->d_trace[2]) return 1;
    const size_t n = (size_t)1 << v->po2;
    hipStream_t stream = (hipStream_t)v->stream;
    rk_params prm;
    if (rk_get_params(v->ctx, &prm) != RK_OK) return 2;
    void* t = nullptr;
    if (rk_alloc(v->ctx, n * 16, &t) != RK_OK) return 2;
    Ext m{{v->mix[0], v->mix[1], v->mix[2], v->mix[3]}};
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(ratio_kernel, dim3(blocks), dim3(256), 0, stream, (uint32_t*)t, v->d_trace[2], n, m, bb::encode(prm.ext_w));
    int rc = hipGetLastError() == hipSuccess ? 0 : 3;
    if (!rc && rk_prefix_products(v->ctx, (uint32_t*)t, n) != RK_OK) rc = 4;  // Hal::prefix_products
    if (!rc) {
        hipLaunchKernelGGL(fill_accum_kernel, dim3(blocks), dim3(256), 0, stream, d_accum, (const uint32_t*)t, v->d_trace[2],
                           n, v->group_size[0], v->group_size[2], col_mix(v));
        if (hipGetLastError() != hipSuccess) rc = 3;
    }
    (void)rk_free(v->ctx, t);  // drains the stream first
    return rc;
}

int eval_check(void*, const rk_circuit_view* v, const uint32_t poly_mix[4], uint32_t* d_check) {
    if (!shape_ok(v->group_size, v->n_mix) || !v->d_lde[0] || !v->d_lde[1] || !v->d_lde[2]) return 1;
    rk_params prm;
    if (rk_get_params(v->ctx, &prm) != RK_OK) return 2;
    const size_t n = (size_t)1 << v->po2, d = n << prm.blowup_log2;
    CheckArgs a{};
    a.acc = v->d_lde[0];
    a.code = v->d_lde[1];
    a.data = v->d_lde[2];
    a.d = d;
    a.blow = prm.blowup_log2;
    a.wm = bb::encode(prm.ext_w);
    a.wa = v->group_size[0];
    a.wd = v->group_size[2];
    std::memcpy(a.poly_mix.c, poly_mix, 16);
    std::memcpy(a.m.c, v->mix, 16);
    // x_i^N for x_i = shift * w_D^i takes D/N values: shift^N * w_(D/N)^(i mod D/N)
    const uint32_t shift_n = bb::pow(bb::encode(prm.coset_shift), n);
    const uint32_t wb = bb::pow(bb::encode(prm.root_2_27), (uint64_t)1 << (27 - prm.blowup_log2));
    for (uint32_t r = 0; r < (1u << prm.blowup_log2); r++) a.inv_den[r] = bb::inv(bb::sub(bb::mul(shift_n, bb::pow(wb, r)), bb::ONE));
    a.cm = col_mix(v);
    hipLaunchKernelGGL(eval_check_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, (hipStream_t)v->stream, d_check,
                       a);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}

// position of (group, offset, back) in eval_u: registers in (group, offset) order, backs in combo order
long tap_index(const rk_taps& t, uint32_t group, uint32_t offset, uint32_t back) {
    size_t pos = 0;
    for (uint32_t r = 0; r < t.n_regs; r++) {
        uint32_t cb = t.reg_combo[r], b0 = t.combo_off[cb], b1 = t.combo_off[cb + 1];
        if (t.reg_group[r] == group && t.reg_offset[r] == offset) {
            for (uint32_t b = b0; b < b1; b++)
                if (t.combo_backs[b] == back) return (long)(pos + (b - b0));
            return -1;
        }
        pos += b1 - b0;
    }
    return -1;
}

const rk_circuit_hooks g_hooks = {nullptr, accumulate, eval_check};

}  // namespace

extern "C" {

const rk_circuit_hooks* toy_circuit_hooks(void) { return &g_hooks; }

int toy_circuit_poly_ext(void*, const rk_segment* pub, const uint32_t poly_mix[4], const uint32_t* eval_u_ext, size_t,
                         const uint32_t* mix, uint32_t n_mix, uint32_t out_ext[4]) {
    const rk_taps& t = pub->taps;
    if (!shape_ok(t.group_size, n_mix)) return 1;
    const Ext* u = reinterpret_cast<const Ext*>(eval_u_ext);
    auto at = [&](uint32_t g, uint32_t o, uint32_t back, Ext* out) {
        long i = tap_index(t, g, o, back);
        if (i < 0) return false;
        *out = u[i];
        return true;
    };
    Ext c0, c1, c2, d0, d0b1, d0b2, d1, d2, d3;
    if (!at(1, 0, 0, &c0) || !at(1, 1, 0, &c1) || !at(1, 2, 0, &c2) || !at(2, 0, 0, &d0) || !at(2, 0, 1, &d0b1) ||
        !at(2, 0, 2, &d0b2) || !at(2, 1, 0, &d1) || !at(2, 2, 0, &d2) || !at(2, 3, 0, &d3))
        return 2;
    Ext A = bb::ext_zero(), Ab = bb::ext_zero();
    for (uint32_t e = 0; e < 4; e++) {  // off the trace domain each a_e opens to an extension element
        Ext v0, v1, basis = bb::ext_zero();
        if (!at(0, e, 0, &v0) || !at(0, e, 1, &v1)) return 2;
        basis.c[e] = bb::ONE;
        A = bb::add(A, bb::mul(v0, basis));
        Ab = bb::add(Ab, bb::mul(v1, basis));
    }
    Ext pm, m;
    std::memcpy(pm.c, poly_mix, 16);
    std::memcpy(m.c, mix, 16);
    const Ext one = bb::ext_one();
    Ext k0 = bb::mul(bb::sub(bb::sub(one, c0), c1), bb::sub(bb::sub(d0, d0b1), d0b2));
    Ext k1 = bb::sub(d1, bb::mul(d0, d0b1));
    Ext prev = bb::add(bb::mul(bb::sub(one, c0), Ab), c0);
    Ext k2 = bb::sub(bb::mul(A, bb::add(m, d3)), bb::mul(prev, bb::add(m, d2)));
    Ext k3 = bb::mul(c2, bb::sub(A, one));
    Ext tot = k0, pw = pm;
    tot = bb::add(tot, bb::mul(pw, k1));
    pw = bb::mul(pw, pm);
    tot = bb::add(tot, bb::mul(pw, k2));
    pw = bb::mul(pw, pm);
    tot = bb::add(tot, bb::mul(pw, k3));
    for (uint32_t k = 4; k < t.group_size[0]; k++) {
        Ext ak, dk;
        if (!at(0, k, 0, &ak) || !at(2, k % t.group_size[2], 0, &dk)) return 2;
        pw = bb::mul(pw, pm);
        tot = bb::add(tot, bb::mul(pw, bb::sub(ak, bb::scale(dk, mix[k % n_mix]))));
    }
    std::memcpy(out_ext, tot.c, 16);
    return 0;
}

}  // extern "C"
