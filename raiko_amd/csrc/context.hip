// Context, device memory and table management + the plain C entry points for them.
#include <dlfcn.h>

#include <cstdlib>

#include "internal.hpp"

#include <cstring>
#include <memory>

namespace rk {

namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* on = std::getenv("RK_ROCTX");
        if (!on || !*on || *on == '0') return;
        // rocprofv3 listens to the SDK's roctx; the older tools to libroctx64
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
const Roctx& roctx() {
    static Roctx r;
    return r;
}
}  // namespace

bool trace_push(const char* name) {
    const Roctx& r = roctx();
    if (!r.push) return false;
    (void)r.push(name);
    return true;
}
void trace_pop() {
    const Roctx& r = roctx();
    if (r.pop) (void)r.pop();
}

int post_launch(rk_ctx* ctx, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
        return RK_ERR_HIP;
    }
    return RK_OK;
}

int dev_alloc(rk_ctx* ctx, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) & ~(size_t)255;
    auto it = ctx->free_list.find(bytes);
    if (it != ctx->free_list.end()) {
        *out = it->second;
        ctx->free_list.erase(it);
        ctx->pooled_bytes -= bytes;
        ctx->live[*out] = bytes;
        return RK_OK;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        // drop the cache and retry once
        (void)hipStreamSynchronize(ctx->stream);
        for (auto& kv : ctx->free_list) (void)hipFree(kv.second);
        ctx->free_list.clear();
        ctx->pooled_bytes = 0;
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            ctx->last_error = std::string("hipMalloc: ") + hipGetErrorString(e);
            return RK_ERR_NOMEM;
        }
    }
    ctx->live[p] = bytes;
    *out = p;
    return RK_OK;
}
int dev_free(rk_ctx* ctx, void* p) {
    if (!p) return RK_OK;
    auto it = ctx->live.find(p);
    if (it == ctx->live.end()) return RK_ERR_INVALID;
    ctx->free_list.emplace(it->second, p);
    ctx->pooled_bytes += it->second;
    ctx->live.erase(it);
    if (ctx->pooled_bytes > ctx->pool_limit) {
        // shapes keep changing (e.g. a service proving different po2): hand the cache back.
        // Blocks may still be referenced by queued kernels, so drain the stream first.
        (void)hipStreamSynchronize(ctx->stream);
        for (auto& kv : ctx->free_list) (void)hipFree(kv.second);
        ctx->free_list.clear();
        ctx->pooled_bytes = 0;
    }
    return RK_OK;
}
int upload(rk_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    constexpr size_t RING = (size_t)4 << 20;
    if (bytes == 0) return RK_OK;
    if (!ctx->h_ring && !ctx->h_ring_failed && bytes <= RING / 4) {
        if (hipHostMalloc(&ctx->h_ring, RING, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();  // no page-locked memory to be had: every upload waits instead
            ctx->h_ring = nullptr;
            ctx->h_ring_failed = true;
        }
    }
    if (bytes > RING / 4 || !ctx->h_ring) {
        RK_HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
        RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return RK_OK;
    }
    size_t at = (ctx->h_ring_at + 63) & ~(size_t)63;
    if (at + bytes > RING) {
        // wrap: every copy out of the ring was queued on this stream, so after this wait none is pending
        RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        at = 0;
    }
    std::memcpy((char*)ctx->h_ring + at, h_src, bytes);
    RK_HIP_TRY(ctx, hipMemcpyAsync(d_dst, (char*)ctx->h_ring + at, bytes, hipMemcpyHostToDevice, ctx->stream));
    ctx->h_ring_at = at + bytes;
    return RK_OK;
}

int scratch(rk_ctx* ctx, size_t bytes, void** out) {
    if (bytes > ctx->scratch_bytes) {
        // stream-ordered users of the old buffer must finish before it is released
        RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_scratch) RK_HIP_TRY(ctx, hipFree(ctx->d_scratch));
        ctx->d_scratch = nullptr;
        size_t nb = bytes < (1u << 20) ? (1u << 20) : bytes * 2;
        RK_HIP_TRY(ctx, hipMalloc(&ctx->d_scratch, nb));
        ctx->scratch_bytes = nb;
    }
    *out = ctx->d_scratch;
    return RK_OK;
}

KTimer::KTimer(rk_ctx* c, int cls, double bytes) : ctx(c), idx(-1) {
    if (!c->ktime_on) return;
    if (c->krec_used == c->krecs.size()) {
        rk_ctx::KRec r{};
        if (hipEventCreate(&r.a) != hipSuccess) return;
        if (hipEventCreate(&r.b) != hipSuccess) {
            (void)hipEventDestroy(r.a);
            return;
        }
        c->krecs.push_back(r);
    }
    idx = (int)c->krec_used++;
    c->krecs[idx].cls = cls;
    c->krecs[idx].bytes = bytes;
    (void)hipEventRecord(c->krecs[idx].a, c->stream);
}
KTimer::~KTimer() {
    if (idx >= 0) (void)hipEventRecord(ctx->krecs[idx].b, ctx->stream);
}

// fold finished event pairs into the per-class totals (stream must be idle)
static void ktime_collect(rk_ctx* ctx) {
    for (size_t i = 0; i < ctx->krec_used; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx->krecs[i].a, ctx->krecs[i].b) == hipSuccess) {
            int c = ctx->krecs[i].cls;
            ctx->k_ms[c] += ms;
            ctx->k_bytes[c] += ctx->krecs[i].bytes;
            ctx->k_launches[c] += 1;
        }
    }
    ctx->krec_used = 0;
}

// twiddle / shift tables of the context's field parameters (allocated once, refilled by rk_set_params)
static int build_tables(rk_ctx* ctx) {
    const ntt::TableLayout l = ntt::table_layout();
    std::vector<uint32_t> h(l.total);
    ntt::fill_tables(h.data(), ctx->sys.root27m, ctx->sys.shiftm);
    if (!ctx->d_tables) RK_HIP_TRY(ctx, hipMalloc((void**)&ctx->d_tables, l.total * sizeof(uint32_t)));
    RK_HIP_TRY(ctx, hipMemcpy(ctx->d_tables, h.data(), l.total * sizeof(uint32_t), hipMemcpyHostToDevice));
    ctx->tb = ntt::tables_at(ctx->d_tables);
    return RK_OK;
}
static int upload_p2(rk_ctx* ctx) {
    if (!ctx->d_p2) RK_HIP_TRY(ctx, hipMalloc(&ctx->d_p2, p2::Any::max_raw_size()));
    RK_HIP_TRY(ctx, hipMemcpy(ctx->d_p2, ctx->h_p2.raw(), ctx->h_p2.raw_size(), hipMemcpyHostToDevice));
    return RK_OK;
}

}  // namespace rk

extern "C" {

int rk_abi_version(void) { return 4; }

const char* rk_strerror(int s) {
    switch (s) {
        case RK_OK: return "ok";
        case RK_ERR_INVALID: return "invalid argument";
        case RK_ERR_HIP: return "HIP runtime error";
        case RK_ERR_NOMEM: return "out of device memory";
        case RK_ERR_NODEVICE: return "no usable GPU";
        case RK_ERR_CAPACITY: return "output buffer too small";
        case RK_ERR_INTERNAL: return "prover invariant violated";
        case RK_ERR_VERIFY: return "a produced seal failed verification";
        case RK_ERR_CALLBACK: return "a circuit hook failed";
        default: return "unknown status";
    }
}
const char* rk_last_error(rk_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int rk_device_count(int* count) {
    RK_GUARD_BEGIN
    if (!count) return RK_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        *count = 0;
        return RK_ERR_NODEVICE;
    }
    *count = n;
    return RK_OK;
    RK_GUARD_END
}

int rk_ctx_create(int device, void* stream, rk_ctx** out) {
    RK_GUARD_BEGIN
    if (!out) return RK_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RK_ERR_NODEVICE;
    if (device < 0 || device >= n) return RK_ERR_INVALID;
    rk_ctx* ctx = new rk_ctx();
    ctx->device = device;
    int st = RK_OK;
    do {
        if (hipSetDevice(device) != hipSuccess) { st = RK_ERR_NODEVICE; break; }
        if (stream) {
            ctx->stream = (hipStream_t)stream;
        } else {
            if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { st = RK_ERR_HIP; break; }
            ctx->own_stream = true;
        }
        rk_params def;
        rk::params_preset(&def, RK_PRESET_RISC0);
        st = rk::resolve_params(&def, &ctx->sys, &ctx->h_p2);
        if (st != RK_OK) break;
        st = rk::build_tables(ctx);
        if (st != RK_OK) break;
        st = rk::upload_p2(ctx);
    } while (0);
    if (st != RK_OK) {
        rk_ctx_destroy(ctx);
        return st;
    }
    *out = ctx;
    return RK_OK;
    RK_GUARD_END
}

int rk_ctx_destroy(rk_ctx* ctx) {
    RK_GUARD_BEGIN
    if (!ctx) return RK_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->free_list) (void)hipFree(kv.second);
    for (auto& kv : ctx->live) (void)hipFree(kv.first);
    for (auto& r : ctx->krecs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (hipEvent_t e : ctx->stage_events) (void)hipEventDestroy(e);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->h_ring) (void)hipHostFree(ctx->h_ring);
    if (ctx->d_tables) (void)hipFree(ctx->d_tables);
    if (ctx->d_p2) (void)hipFree(ctx->d_p2);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RK_OK;
    RK_GUARD_END
}

int rk_sync(rk_ctx* ctx) {
    RK_GUARD_BEGIN
    if (!ctx) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RK_OK;
    RK_GUARD_END
}
int rk_alloc(rk_ctx* ctx, size_t bytes, void** d_ptr) {
    RK_GUARD_BEGIN
    if (!ctx || !d_ptr) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return rk::dev_alloc(ctx, bytes, d_ptr);
    RK_GUARD_END
}
int rk_free(rk_ctx* ctx, void* d_ptr) {
    RK_GUARD_BEGIN
    if (!ctx) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the block may still be in use by queued work on the stream
    RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return rk::dev_free(ctx, d_ptr);
    RK_GUARD_END
}
int rk_h2d(rk_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    RK_GUARD_BEGIN
    if (!ctx || (!d_dst && bytes) || (!h_src && bytes)) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    RK_HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RK_OK;
    RK_GUARD_END
}
int rk_d2h(rk_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    RK_GUARD_BEGIN
    if (!ctx || (!h_dst && bytes) || (!d_src && bytes)) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    RK_HIP_TRY(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RK_OK;
    RK_GUARD_END
}

// width-24 tables of the paper's instance (the ABI-1 entry point): the rest of the parameter set stays
int rk_set_poseidon2_params(rk_ctx* ctx, const uint32_t* rc_ext, const uint32_t* rc_int, const uint32_t* diag) {
    RK_GUARD_BEGIN
    if (!ctx || !rc_ext || !rc_int || !diag) return RK_ERR_INVALID;
    rk_params p;
    RK_TRY(rk_get_params(ctx, &p));
    if (p.p2_width != 24) return RK_ERR_INVALID;
    p.p2_rc_ext = rc_ext;
    p.p2_rc_int = rc_int;
    p.p2_diag = diag;
    return rk_set_params(ctx, &p);
    RK_GUARD_END
}

int rk_params_preset(rk_params* out, int preset) {
    RK_GUARD_BEGIN
    if (!out || (preset != RK_PRESET_RISC0 && preset != RK_PRESET_SP1)) return RK_ERR_INVALID;
    rk::params_preset(out, preset);
    return RK_OK;
    RK_GUARD_END
}
int rk_set_params(rk_ctx* ctx, const rk_params* params) {
    RK_GUARD_BEGIN
    if (!ctx || !params) return RK_ERR_INVALID;
    rk::Sys sys;
    auto p2any = std::make_unique<p2::Any>();  // ~50 KiB: not on the stack
    RK_TRY(rk::resolve_params(params, &sys, p2any.get()));
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // kernels in flight read the old tables
    const bool field_changed = sys.root27m != ctx->sys.root27m || sys.shiftm != ctx->sys.shiftm;
    ctx->sys = sys;
    ctx->h_p2 = *p2any;
    if (field_changed) RK_TRY(rk::build_tables(ctx));
    return rk::upload_p2(ctx);
    RK_GUARD_END
}
int rk_get_params(rk_ctx* ctx, rk_params* out) {
    RK_GUARD_BEGIN
    if (!ctx || !out) return RK_ERR_INVALID;
    *out = rk_params{};
    out->struct_size = (uint32_t)sizeof(rk_params);
    out->ext_w = ctx->sys.ext_w;
    out->root_2_27 = ctx->sys.root_2_27;
    out->coset_shift = ctx->sys.coset_shift;
    out->p2_width = (uint32_t)ctx->h_p2.cells();
    out->p2_m4 = (uint32_t)ctx->h_p2.m4();
    out->p2_pad_free = ctx->h_p2.pad_free ? 1u : 0u;
    out->p2_rc_ext = ctx->h_p2.rc_ext();
    out->p2_rc_int = ctx->h_p2.rc_int();
    out->p2_diag = ctx->h_p2.diag();
    out->queries = ctx->sys.queries;
    out->blowup_log2 = ctx->sys.blowup_log2;
    out->fri_fold_log2 = ctx->sys.fri_fold_log2;
    out->fri_min_degree = ctx->sys.fri_min_degree;
    out->pow_bits = ctx->sys.pow_bits;
    return RK_OK;
    RK_GUARD_END
}

int rk_set_kernel_timing(rk_ctx* ctx, int enabled) {
    RK_GUARD_BEGIN
    if (!ctx) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->krec_used = 0;
    for (int c = 0; c < RK_KCLASS_COUNT; c++) {
        ctx->k_ms[c] = 0;
        ctx->k_bytes[c] = 0;
        ctx->k_launches[c] = 0;
    }
    ctx->ktime_on = enabled != 0;
    return RK_OK;
    RK_GUARD_END
}
int rk_kernel_stats(rk_ctx* ctx, int kclass, rk_kernel_stat* out) {
    RK_GUARD_BEGIN
    if (!ctx || !out || kclass < 0 || kclass >= RK_KCLASS_COUNT) return RK_ERR_INVALID;
    RK_HIP_TRY(ctx, hipSetDevice(ctx->device));
    RK_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    rk::ktime_collect(ctx);
    out->launches = ctx->k_launches[kclass];
    out->ms = ctx->k_ms[kclass];
    out->bytes = ctx->k_bytes[kclass];
    return RK_OK;
    RK_GUARD_END
}
const char* rk_kernel_class_name(int kclass) {
    switch (kclass) {
        case RK_KCLASS_HASH_ROWS: return "hash_rows_kernel";
        case RK_KCLASS_HASH_FOLD: return "hash_fold_kernel";
        case RK_KCLASS_NTT_PASS: return "ntt_pass_kernel";
        case RK_KCLASS_BIT_REVERSE: return "bit_reverse_kernel";
        case RK_KCLASS_POLY: return "poly kernels (mix/eval/divide/fold/sum)";
        default: return "?";
    }
}

int rk_last_timing(rk_ctx* ctx, rk_timing* out) {
    RK_GUARD_BEGIN
    if (!ctx || !out) return RK_ERR_INVALID;
    *out = ctx->timing;
    return RK_OK;
    RK_GUARD_END
}

}  // extern "C"
