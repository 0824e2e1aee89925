// rk_comm_* / rk_gather_seals (include/raiko_hip.h): the one collective of the path -- the seal gather of a host that
// runs one process per GPU (`north_star`: "execution-trace segments shard embarrassingly across the 8 GPUs of one node
// with an RCCL-over-xGMI gather only for the final receipt tree").  raiko's own host is a single process
// (core/src/interfaces.rs:187-193) and needs no collective: rk_prove_session's work queue writes every seal into the
// caller's buffers.  This entry point is for the other deployment: rank r proves segments r, r + world, ... and
// the ranks exchange the variable-length seals with two ncclAllGather calls (a length table, then padded payloads:
// < 3 MB for 8 segments, latency-bound) over RCCL.
//
// RCCL is looked up at run time (dlopen of librccl.so.1): the library has no link-time dependency on it, a host that
// never calls rk_comm_* never loads it, and a process that already carries a librccl (PyTorch-ROCm does) shares it.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "internal.hpp"

namespace {

// the slice of rccl.h this file uses (ABI-stable NCCL 2 entry points)
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId {
    char internal[128];
};
typedef int ncclResult_t;
enum { ncclUint32 = 3 };
struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
        }
        if (!r.so) {
            r.error = "librccl.so.1 not found";
            return;
        }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.so, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.so, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))dlsym(r.so, "ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) r.error = "librccl lacks an NCCL 2 entry point";
    });
    return r;
}

}  // namespace

struct rk_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    std::string last_error;
};

namespace {
int nccl_fail(rk_comm* c, const char* what, ncclResult_t r) {
    c->last_error = std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error");
    return RK_ERR_HIP;
}
}  // namespace

extern "C" {

int rk_comm_unique_id(uint8_t id[RK_COMM_ID_BYTES]) {
    RK_GUARD_BEGIN
    if (!id) return RK_ERR_INVALID;
    Rccl& r = rccl();
    if (!r.error.empty()) return RK_ERR_NODEVICE;
    ncclUniqueId u;
    if (r.GetUniqueId(&u) != 0) return RK_ERR_HIP;
    static_assert(sizeof u == RK_COMM_ID_BYTES, "NCCL unique id size");
    std::memcpy(id, &u, sizeof u);
    return RK_OK;
    RK_GUARD_END
}

int rk_comm_create(const uint8_t id[RK_COMM_ID_BYTES], int rank, int world, int device, rk_comm** out) {
    RK_GUARD_BEGIN
    if (!out) return RK_ERR_INVALID;
    *out = nullptr;
    if (!id || world < 1 || world > 1024 || rank < 0 || rank >= world || device < 0) return RK_ERR_INVALID;
    Rccl& r = rccl();
    if (!r.error.empty()) return RK_ERR_NODEVICE;
    if (hipSetDevice(device) != hipSuccess) return RK_ERR_NODEVICE;
    std::unique_ptr<rk_comm> c(new rk_comm);
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    if (r.CommInitRank(&c->comm, world, u, rank) != 0) return RK_ERR_HIP;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        (void)r.CommDestroy(c->comm);
        return RK_ERR_HIP;
    }
    *out = c.release();
    return RK_OK;
    RK_GUARD_END
}

int rk_comm_destroy(rk_comm* c) {
    RK_GUARD_BEGIN
    if (!c) return RK_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamDestroy(c->stream);
    }
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    delete c;
    return RK_OK;
    RK_GUARD_END
}

const char* rk_comm_last_error(rk_comm* c) { return c ? c->last_error.c_str() : ""; }

// The host half of the gather, separately callable (and testable without a GPU): the gathered length table
// (world x per_rank words) and payload (world x per_rank x max_len words) into segment order -- global segment i is
// slot i / world of rank i mod world.
int rk_gather_unpack(const uint32_t* all_lens, const uint32_t* all_payload, int world, size_t per_rank, size_t max_len, size_t n_total,
                     uint32_t* const* h_out, const size_t* out_capacity, size_t* out_words) {
    RK_GUARD_BEGIN
    if (!all_lens || !out_words || world < 1 || per_rank * (size_t)world < n_total || (max_len && !all_payload)) return RK_ERR_INVALID;
    int rc = RK_OK;
    for (size_t i = 0; i < n_total; i++) {
        const size_t r = i % (size_t)world, j = i / (size_t)world;
        const size_t len = all_lens[r * per_rank + j];
        if (len > max_len) return RK_ERR_INTERNAL;
        out_words[i] = len;
        if (!h_out || !h_out[i]) continue;
        if (out_capacity && out_capacity[i] < len) {
            rc = RK_ERR_CAPACITY;
            continue;
        }
        std::memcpy(h_out[i], all_payload + (r * per_rank + j) * max_len, len * 4);
    }
    return rc;
    RK_GUARD_END
}

int rk_gather_seals(rk_comm* c, const uint32_t* const* h_local_seals, const size_t* local_words, size_t n_local, size_t n_total,
                    uint32_t* const* h_out, const size_t* out_capacity, size_t* out_words) {
    RK_GUARD_BEGIN
    if (!c || !out_words || (n_local && (!h_local_seals || !local_words))) return RK_ERR_INVALID;
    const size_t world = (size_t)c->world, rank = (size_t)c->rank;
    const size_t mine = n_total > rank ? (n_total - rank + world - 1) / world : 0;   // segments rank, rank + world, ...
    if (n_local != mine) return RK_ERR_INVALID;
    const size_t per_rank = (n_total + world - 1) / world;
    if (per_rank == 0) return RK_OK;
    for (size_t j = 0; j < n_local; j++)
        if (local_words[j] >= ((size_t)1 << 31) || (local_words[j] && !h_local_seals[j])) return RK_ERR_INVALID;
    Rccl& r = rccl();
    if (hipSetDevice(c->device) != hipSuccess) return RK_ERR_NODEVICE;
    struct Dev {
        void* p = nullptr;
        ~Dev() {
            if (p) (void)hipFree(p);
        }
    } d_lens, d_all_lens, d_pay, d_all_pay;
    auto hip_fail = [&](const char* what, hipError_t e) {
        c->last_error = std::string(what) + ": " + hipGetErrorString(e);
        return RK_ERR_HIP;
    };
    hipError_t e;
    // 1. the length table
    std::vector<uint32_t> lens(per_rank, 0), all_lens(per_rank * world, 0);
    for (size_t j = 0; j < n_local; j++) lens[j] = (uint32_t)local_words[j];
    if ((e = hipMalloc(&d_lens.p, per_rank * 4)) != hipSuccess) return hip_fail("hipMalloc", e);
    if ((e = hipMalloc(&d_all_lens.p, per_rank * world * 4)) != hipSuccess) return hip_fail("hipMalloc", e);
    if ((e = hipMemcpyAsync(d_lens.p, lens.data(), per_rank * 4, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return hip_fail("hipMemcpyAsync", e);
    ncclResult_t nr = r.AllGather(d_lens.p, d_all_lens.p, per_rank, ncclUint32, c->comm, c->stream);
    if (nr != 0) return nccl_fail(c, "ncclAllGather(lengths)", nr);
    if ((e = hipMemcpyAsync(all_lens.data(), d_all_lens.p, per_rank * world * 4, hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
        return hip_fail("hipMemcpyAsync", e);
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return hip_fail("hipStreamSynchronize", e);
    size_t max_len = 0;
    for (uint32_t v : all_lens) max_len = std::max<size_t>(max_len, v);
    // 2. the padded payloads
    std::vector<uint32_t> all_pay;
    if (max_len) {
        std::vector<uint32_t> pay(per_rank * max_len, 0);
        for (size_t j = 0; j < n_local; j++) std::memcpy(&pay[j * max_len], h_local_seals[j], local_words[j] * 4);
        all_pay.resize(per_rank * max_len * world);
        if ((e = hipMalloc(&d_pay.p, pay.size() * 4)) != hipSuccess) return hip_fail("hipMalloc", e);
        if ((e = hipMalloc(&d_all_pay.p, all_pay.size() * 4)) != hipSuccess) return hip_fail("hipMalloc", e);
        if ((e = hipMemcpyAsync(d_pay.p, pay.data(), pay.size() * 4, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return hip_fail("hipMemcpyAsync", e);
        nr = r.AllGather(d_pay.p, d_all_pay.p, pay.size(), ncclUint32, c->comm, c->stream);
        if (nr != 0) return nccl_fail(c, "ncclAllGather(payload)", nr);
        if ((e = hipMemcpyAsync(all_pay.data(), d_all_pay.p, all_pay.size() * 4, hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
            return hip_fail("hipMemcpyAsync", e);
        if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return hip_fail("hipStreamSynchronize", e);   // `pay` must outlive the copy
    }
    return rk_gather_unpack(all_lens.data(), all_pay.data(), c->world, per_rank, max_len, n_total, h_out, out_capacity, out_words);
    RK_GUARD_END
}

}  // extern "C"
