// rk_prove_session: all segments of a session, several in flight per GPU, over one or more GPUs.
//
// The reference proves a session's segments one after the other (`session.prove()`,
// provers/risc0/driver/src/bonsai.rs:271).  One proof is a chain of ~120 dependent launches with a
// host round trip at every Merkle root, so a lone proof leaves an MI355X partly idle; segments are
// independent, so this entry point runs `inflight` prover contexts per GPU (one HIP stream, scratch
// pool and host thread each).  Per GPU one more context + thread (the feeder) claims segments from
// the session-wide work queue -- one claim flag per segment, shared by all GPUs, so a slower GPU or a
// short last segment never stalls the others -- and stages host-resident traces up to
// `upload_ahead` segments ahead into a ring of device buffers: the PCIe upload of segment i+1 runs
// under the proof of segment i.  Seals land in the caller's host buffers (no inter-GPU traffic, no
// collective: a single-process host such as raiko's, core/src/interfaces.rs:187-193, needs none) and
// are verified (rk_verify_segment_ex, host code) by one more thread while the GPUs go on.
// Contexts and staging buffers are kept per device for the life of the process
// (rk_session_release frees them): the `Prover` trait of the reference has no `self`, a backend's
// state is process-global (lib/src/prover.rs:52-62).
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <cstdlib>

#include "internal.hpp"

namespace {

// RK_TEST_LOGICAL_DEVICES=k (test switch): the session code sees k devices, logical device d running on physical GPU
// d mod (number of GPUs).  It lets the multi-device path -- one pool, one feeder and `inflight` provers per device, the
// session-wide claim flags, the pinning of device-resident segments -- run on a box with a single GPU; every pool
// still has its own contexts, streams, staging ring and lock, so nothing is shared that real devices would not share.
int logical_devices() {
    const char* e = std::getenv("RK_TEST_LOGICAL_DEVICES");
    if (!e || !*e) return 0;
    const int k = std::atoi(e);
    return k > 0 && k <= 64 ? k : 0;
}
int physical_of(int device, int n_gpus) { return logical_devices() > 0 && n_gpus > 0 ? device % n_gpus : device; }

struct Slot {
    void* group[3] = {nullptr, nullptr, nullptr};
    void* check = nullptr;
    size_t words[4] = {0, 0, 0, 0};
};

void free_slot(rk_ctx* up, Slot* s) {
    for (int g = 0; g < 3; g++) {
        if (s->group[g]) (void)rk_free(up, s->group[g]);
        s->group[g] = nullptr;
    }
    if (s->check) (void)rk_free(up, s->check);
    s->check = nullptr;
    for (auto& w : s->words) w = 0;
}

struct DevicePool {
    int device = 0;
    std::vector<rk_ctx*> provers;
    rk_ctx* uploader = nullptr;
    std::vector<Slot*> ring;
    std::mutex busy;  // one session at a time per device (`run` may be entered from many threads)
    std::string last_error;
    int physical = 0;       // the GPU behind it (= device unless RK_TEST_LOGICAL_DEVICES is set)
    size_t last_proven = 0; // segments this device proved in the last session it took part in
    bool ktime_on = false;  // applied to contexts created later
    std::vector<std::vector<uint32_t>> prover_key;  // per prover: the parameter set it carries (empty: risc0's defaults)
    void clear() {
        if (uploader) {
            for (Slot* s : ring) {
                free_slot(uploader, s);
                delete s;
            }
            (void)rk_ctx_destroy(uploader);
        }
        ring.clear();
        uploader = nullptr;
        for (rk_ctx* c : provers) (void)rk_ctx_destroy(c);
        provers.clear();
        prover_key.clear();
    }
    ~DevicePool() { clear(); }
};

std::mutex g_mu;
// shared ownership: a session that looked its pool up keeps it alive even if rk_session_release
// drops it from the map before the session locks `busy`
std::map<int, std::shared_ptr<DevicePool>> g_pools;

struct Item {
    size_t idx;
    Slot* slot;
};

struct DevRun {
    DevicePool* pool = nullptr;
    std::deque<Item> ready;       // claimed (and staged) segments not yet taken by a prover
    std::deque<Slot*> free_slots;
    size_t outstanding = 0;       // claimed, proof not finished
    size_t max_outstanding = 1;
    size_t n_slots = 1;
    size_t cursor = 0;
    size_t provers_left = 0;
    size_t proven = 0;
    bool feeder_done = false;
};

// shared state of one rk_prove_session call
struct Run {
    const rk_segment* segs;
    size_t n;
    uint32_t* const* h_seals;
    const size_t* caps;
    size_t* words;
    int verify;
    const rk_verify_opts* verify_opts;
    unsigned blowup_log2 = 2;  // of the session's parameter set: the check evaluations span 4 << blowup_log2 columns of rows

    std::unique_ptr<std::atomic<unsigned char>[]> claimed;
    std::vector<int> owner;          // device a device-resident segment lives on (-1: host-resident, any GPU)
    std::vector<DevRun> devs;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<size_t> to_verify;    // finished seals waiting for the verifier thread
    size_t provers_left = 0;
    int status = RK_OK;              // first failure
    size_t failed = (size_t)-1;
    std::string detail;

    void fail(int st, size_t idx, const char* text) {
        std::lock_guard<std::mutex> l(mu);
        if (status == RK_OK) {
            status = st;
            failed = idx;
            detail = text ? text : "";
        }
        cv.notify_all();
    }
    // next unclaimed segment device `dev` may take, scanning from its cursor (mu held)
    bool claim(size_t d, size_t* out) {
        DevRun& dr = devs[d];
        for (size_t i = dr.cursor; i < n; i++) {
            if (owner[i] >= 0 && owner[i] != dr.pool->device) continue;
            if (claimed[i].exchange(1)) {
                if (i == dr.cursor) dr.cursor++;
                continue;
            }
            if (i == dr.cursor) dr.cursor++;
            *out = i;
            return true;
        }
        return false;
    }
};

bool needs_staging(const rk_segment& s) { return s.on_device == 0; }
bool hook_accum(const rk_segment& s) { return s.hooks && s.hooks->accumulate; }
bool hook_check(const rk_segment& s) { return s.hooks && (s.hooks->eval_check || s.hooks->program); }

// claims segments for one GPU and stages the host-resident ones
void feeder(Run* run, size_t d) {
    DevRun& dr = run->devs[d];
    rk_ctx* up = dr.pool->uploader;
    for (;;) {
        size_t i = 0;
        Slot* slot = nullptr;
        {
            std::unique_lock<std::mutex> l(run->mu);
            run->cv.wait(l, [&] { return run->status != RK_OK || dr.outstanding < dr.max_outstanding; });
            if (run->status != RK_OK || !run->claim(d, &i)) break;
            dr.outstanding++;
            if (needs_staging(run->segs[i])) {
                if (!dr.free_slots.empty()) {
                    slot = dr.free_slots.front();
                    dr.free_slots.pop_front();
                } else {  // outstanding < n_slots: the ring has room to grow
                    slot = new Slot();
                    dr.pool->ring.push_back(slot);
                }
            }
        }
        if (slot) {
            const rk_segment& seg = run->segs[i];
            const size_t rows = (size_t)1 << seg.po2;
            const size_t want[4] = {hook_accum(seg) ? 0 : rows * seg.taps.group_size[0], rows * seg.taps.group_size[1],
                                    rows * seg.taps.group_size[2], hook_check(seg) ? 0 : (rows * 4) << run->blowup_log2};
            int st = RK_OK;
            if (want[0] != slot->words[0] || want[1] != slot->words[1] || want[2] != slot->words[2] ||
                want[3] != slot->words[3]) {
                free_slot(up, slot);
                for (int g = 0; g < 3 && st == RK_OK; g++)
                    if (want[g]) st = rk_alloc(up, want[g] * 4, &slot->group[g]);
                if (st == RK_OK && want[3]) st = rk_alloc(up, want[3] * 4, &slot->check);
                if (st == RK_OK)
                    for (int k = 0; k < 4; k++) slot->words[k] = want[k];
            }
            for (int g = 0; g < 3 && st == RK_OK; g++) {
                if (!want[g]) continue;
                st = seg.group[g] ? rk_h2d(up, slot->group[g], seg.group[g], want[g] * 4) : RK_ERR_INVALID;
            }
            if (st == RK_OK && want[3]) st = seg.check ? rk_h2d(up, slot->check, seg.check, want[3] * 4) : RK_ERR_INVALID;
            if (st == RK_OK) st = rk_sync(up);  // the buffers change hands after this
            if (st != RK_OK) {
                run->fail(st, i, rk_last_error(up));
                break;
            }
        }
        std::lock_guard<std::mutex> l(run->mu);
        dr.ready.push_back(Item{i, slot});
        run->cv.notify_all();
    }
    std::lock_guard<std::mutex> l(run->mu);
    dr.feeder_done = true;
    run->cv.notify_all();
}

void prover(rk_ctx* ctx, Run* run, size_t d) {
    DevRun& dr = run->devs[d];
    for (;;) {
        Item it{};
        {
            std::unique_lock<std::mutex> l(run->mu);
            run->cv.wait(l, [&] { return run->status != RK_OK || !dr.ready.empty() || dr.feeder_done; });
            if (run->status != RK_OK || dr.ready.empty()) return;
            it = dr.ready.front();
            dr.ready.pop_front();
        }
        const size_t i = it.idx;
        rk_segment seg = run->segs[i];
        if (it.slot) {
            seg.on_device = 2;  // the staged copy is ours: no second copy inside the prover
            for (int g = 0; g < 3; g++) seg.group[g] = (const uint32_t*)it.slot->group[g];
            seg.check = (const uint32_t*)it.slot->check;
        }
        int st = rk_prove_segment(ctx, &seg, run->h_seals[i], run->caps[i], &run->words[i]);
        {
            std::lock_guard<std::mutex> l(run->mu);
            if (it.slot) dr.free_slots.push_back(it.slot);
            dr.outstanding--;
            if (st == RK_OK) dr.proven++;
            if (st == RK_OK && run->verify) run->to_verify.push_back(i);  // host work for the verifier thread
            run->cv.notify_all();
        }
        if (st != RK_OK) {
            run->fail(st, i, rk_last_error(ctx));
            return;
        }
    }
}

void prover_thread(rk_ctx* ctx, Run* run, size_t d) {
    try {
        prover(ctx, run, d);
    } catch (...) {
        run->fail(RK_ERR_INTERNAL, (size_t)-1, "exception in a prover thread");
    }
    std::lock_guard<std::mutex> l(run->mu);
    run->provers_left--;
    run->cv.notify_all();
}
void feeder_thread(Run* run, size_t d) {
    try {
        feeder(run, d);
    } catch (...) {
        run->fail(RK_ERR_INTERNAL, (size_t)-1, "exception in a feeder thread");
        std::lock_guard<std::mutex> l(run->mu);
        run->devs[d].feeder_done = true;
        run->cv.notify_all();
    }
}

// rk_verify_segment_ex of every finished seal (~9 ms of host time at S20, against ~26 ms per proof)
void verifier(Run* run) {
    for (;;) {
        size_t i;
        {
            std::unique_lock<std::mutex> l(run->mu);
            run->cv.wait(l, [&] { return !run->to_verify.empty() || run->provers_left == 0 || run->status != RK_OK; });
            if (run->status != RK_OK) return;
            if (run->to_verify.empty()) return;  // all provers done, nothing left
            i = run->to_verify.front();
            run->to_verify.pop_front();
        }
        int v = RK_ERR_INTERNAL;
        try {
            v = rk_verify_segment_ex(&run->segs[i], run->verify_opts, run->h_seals[i], run->words[i]);
        } catch (...) {
        }
        if (v != 0) {
            std::string why = "seal failed verification (reason " + std::to_string(v) + ")";
            run->fail(RK_ERR_VERIFY, i, why.c_str());
            return;
        }
    }
}

int prove_session(const rk_session_opts* opts, const rk_segment* segs, size_t n, uint32_t* const* h_seals,
                  const size_t* seal_capacity_words, size_t* seal_words, size_t* failed_index) {
    if (failed_index) *failed_index = (size_t)-1;
    if (!opts || (n && (!segs || !h_seals || !seal_capacity_words || !seal_words))) return RK_ERR_INVALID;
    if (opts->inflight < 1 || opts->inflight > 16 || opts->upload_ahead < 0 || opts->upload_ahead > 16) return RK_ERR_INVALID;
    if (opts->n_devices < 0 || opts->n_devices > 64 || (opts->n_devices > 0 && !opts->devices)) return RK_ERR_INVALID;
    std::vector<int> devices;
    if (opts->n_devices > 0) devices.assign(opts->devices, opts->devices + opts->n_devices);
    else devices.push_back(opts->device);
    std::sort(devices.begin(), devices.end());  // pools are locked in ascending order: no lock cycles
    if (std::adjacent_find(devices.begin(), devices.end()) != devices.end()) return RK_ERR_INVALID;
    if (n == 0) return RK_OK;
    int n_gpus = 0;
    if (hipGetDeviceCount(&n_gpus) != hipSuccess || n_gpus <= 0) return RK_ERR_NODEVICE;
    const int n_visible = logical_devices() > 0 ? logical_devices() : n_gpus;
    for (int d : devices)
        if (d < 0 || d >= n_visible) return RK_ERR_INVALID;

    std::vector<std::shared_ptr<DevicePool>> pools;
    {
        std::lock_guard<std::mutex> l(g_mu);
        for (int d : devices) {
            auto& sp = g_pools[d];
            if (!sp) {
                sp = std::make_shared<DevicePool>();
                sp->device = d;
            }
            sp->physical = physical_of(d, n_gpus);
            pools.push_back(sp);
        }
    }
    std::vector<std::unique_lock<std::mutex>> sessions;
    for (auto& p : pools) sessions.emplace_back(p->busy);

    Run run;
    run.segs = segs;
    run.n = n;
    run.h_seals = h_seals;
    run.caps = seal_capacity_words;
    run.words = seal_words;
    run.verify = opts->verify;
    run.verify_opts = opts->verify_opts;
    run.claimed.reset(new std::atomic<unsigned char>[n]);
    for (size_t i = 0; i < n; i++) run.claimed[i].store(0);
    run.owner.assign(n, -1);
    bool any_host = false;
    for (size_t i = 0; i < n; i++) {
        if (segs[i].on_device > 2) return RK_ERR_INVALID;
        if (needs_staging(segs[i])) {
            any_host = true;
        } else if (devices.size() == 1) {
            run.owner[i] = devices[0];
        } else {  // device-resident input: only the GPU holding it can prove it
            hipPointerAttribute_t attr{};
            if (!segs[i].group[1] || hipPointerGetAttributes(&attr, segs[i].group[1]) != hipSuccess) {
                (void)hipGetLastError();
                return RK_ERR_INVALID;
            }
            // the device of the session that runs on the GPU holding the data (with logical test devices: the first)
            for (int d : devices)
                if (run.owner[i] < 0 && physical_of(d, n_gpus) == attr.device) run.owner[i] = d;
            if (run.owner[i] < 0) {
                if (failed_index) *failed_index = i;
                return RK_ERR_INVALID;
            }
        }
    }
    const size_t workers = std::min<size_t>((size_t)opts->inflight, n);
    // the session's parameter set: every word that defines it, so that an equal set costs nothing next time
    rk_params prm;
    rk::params_preset(&prm, RK_PRESET_RISC0);
    // every word that defines a parameter set; empty for risc0's defaults
    auto key_of = [](const rk_params& p, std::vector<uint32_t>* out) -> int {
        rk::Sys sys;
        auto any = std::make_unique<p2::Any>();
        int st = rk::resolve_params(&p, &sys, any.get());
        if (st != RK_OK) return st;
        *out = {p.ext_w, p.root_2_27, p.coset_shift, p.p2_width, p.p2_m4, p.p2_pad_free, p.queries, p.blowup_log2, p.fri_fold_log2,
                p.fri_min_degree, p.pow_bits};
        out->insert(out->end(), any->rc_ext(), any->rc_ext() + 8 * any->cells());
        out->insert(out->end(), any->rc_int(), any->rc_int() + any->rounds_partial());
        out->insert(out->end(), any->diag(), any->diag() + any->cells());
        return RK_OK;
    };
    std::vector<uint32_t> key;
    if (opts->params) {
        std::vector<uint32_t> dkey;
        int st = key_of(*opts->params, &key);
        if (st != RK_OK) return st;
        (void)key_of(prm, &dkey);
        prm = *opts->params;
        if (key == dkey) key.clear();
    }
    rk_verify_opts vopts{};
    if (opts->verify_opts) vopts = *opts->verify_opts;
    if (opts->params && !vopts.params) vopts.params = opts->params;
    run.verify_opts = (opts->verify_opts || opts->params) ? &vopts : nullptr;
    run.blowup_log2 = prm.blowup_log2;
    run.devs.resize(devices.size());
    for (size_t d = 0; d < devices.size(); d++) {
        DevicePool* pool = pools[d].get();
        while (pool->provers.size() < workers) {
            rk_ctx* c = nullptr;
            int st = rk_ctx_create(pool->physical, nullptr, &c);
            if (st != RK_OK) return st;
            if (pool->ktime_on) (void)rk_set_kernel_timing(c, 1);
            pool->provers.push_back(c);
        }
        pool->prover_key.resize(pool->provers.size());
        for (size_t j = 0; j < workers; j++) {
            if (pool->prover_key[j] == key) continue;
            int st = rk_set_params(pool->provers[j], &prm);
            if (st != RK_OK) return st;
            pool->prover_key[j] = key;
        }
        if (any_host && !pool->uploader) {
            int st = rk_ctx_create(pool->physical, nullptr, &pool->uploader);
            if (st != RK_OK) return st;
        }
        DevRun& dr = run.devs[d];
        dr.pool = pool;
        // with nothing to hide behind (upload_ahead == 0) the ring still needs one slot per prover
        dr.n_slots = (size_t)opts->upload_ahead + workers;
        dr.max_outstanding = dr.n_slots;
        for (size_t k = 0; k < pool->ring.size() && k < dr.n_slots; k++) dr.free_slots.push_back(pool->ring[k]);
        dr.provers_left = workers;
    }
    run.provers_left = workers * devices.size();
    std::vector<std::thread> threads;
    for (size_t d = 0; d < devices.size(); d++) {
        threads.emplace_back(feeder_thread, &run, d);
        for (size_t w = 0; w < workers; w++) threads.emplace_back(prover_thread, pools[d]->provers[w], &run, d);
    }
    // a seal takes ~7 ms of host time to verify whatever the segment size (50 query paths of Poseidon2):
    // one thread keeps up with 2^20-cycle segments, smaller ones need more (2^16: 3.5 ms per proof)
    if (run.verify) {
        unsigned hw = std::thread::hardware_concurrency();
        size_t nv = run.verify == 1 ? std::min<size_t>(std::min<size_t>(16, 4 * devices.size()), std::max<unsigned>(1, hw / 4))
                                    : std::min<size_t>((size_t)run.verify, 16);
        nv = std::min(nv, n);
        for (size_t v = 0; v < nv; v++) threads.emplace_back(verifier, &run);
    }
    for (auto& t : threads) t.join();
    for (size_t d = 0; d < devices.size(); d++) pools[d]->last_proven = run.devs[d].proven;
    if (run.status != RK_OK) {
        if (failed_index) *failed_index = run.failed;
        std::lock_guard<std::mutex> l(g_mu);
        for (auto& p : pools) p->last_error = run.detail;
    }
    return run.status;
}

}  // namespace

extern "C" {

int rk_prove_session(const rk_session_opts* opts, const rk_segment* segs, size_t n, uint32_t* const* h_seals,
                     const size_t* seal_capacity_words, size_t* seal_words, size_t* failed_index) {
    RK_GUARD_BEGIN
    return prove_session(opts, segs, n, h_seals, seal_capacity_words, seal_words, failed_index);
    RK_GUARD_END
}

}  // extern "C"

// ---- sessions that grow while they run -------------------------------------------------------------
// A worker thread takes whatever has been submitted since its last look and proves it as one
// rk_prove_session batch (the device's contexts persist, so a batch costs nothing to start): the host can
// hand over segment k while its executor is still producing segment k + 1.
struct rk_stream {
    rk_session_opts opts{};
    std::vector<int> devices;
    rk_params params{};
    rk_verify_opts vopts{};
    struct Item {
        rk_segment seg;
        uint32_t* seal;
        size_t cap;
        size_t* words;
    };
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Item> items;
    size_t next = 0;
    size_t done = 0;     // items [0, done) are finished: batches complete in submission order
    bool closed = false;
    int status = RK_OK;
    size_t failed = (size_t)-1;
    std::thread worker;

    void run() {
        for (;;) {
            size_t b0, b1;
            {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [&] { return closed || next < items.size(); });
                if (next >= items.size()) return;  // closed and drained
                b0 = next;
                b1 = items.size();
                next = b1;
                if (status != RK_OK) {             // after a failure the rest is only drained
                    done = b1;
                    cv.notify_all();
                    continue;
                }
            }
            const size_t n = b1 - b0;
            std::vector<rk_segment> segs(n);
            std::vector<uint32_t*> seals(n);
            std::vector<size_t> caps(n), words(n, 0);
            {
                std::lock_guard<std::mutex> l(mu);
                for (size_t i = 0; i < n; i++) {
                    segs[i] = items[b0 + i].seg;
                    seals[i] = items[b0 + i].seal;
                    caps[i] = items[b0 + i].cap;
                }
            }
            size_t bad = (size_t)-1;
            int st = RK_ERR_INTERNAL;
            try {
                st = prove_session(&opts, segs.data(), n, seals.data(), caps.data(), words.data(), &bad);
            } catch (...) {
            }
            std::lock_guard<std::mutex> l(mu);
            for (size_t i = 0; i < n; i++)
                if (items[b0 + i].words) *items[b0 + i].words = words[i];
            if (st != RK_OK && status == RK_OK) {
                status = st;
                failed = bad == (size_t)-1 ? bad : b0 + bad;
            }
            done = b1;
            cv.notify_all();
        }
    }
};

extern "C" {

int rk_stream_open(const rk_session_opts* opts, rk_stream** out) {
    RK_GUARD_BEGIN
    if (!opts || !out) return RK_ERR_INVALID;
    *out = nullptr;
    if (opts->inflight < 1 || opts->inflight > 16 || opts->upload_ahead < 0 || opts->upload_ahead > 16) return RK_ERR_INVALID;
    if (opts->n_devices < 0 || opts->n_devices > 64 || (opts->n_devices > 0 && !opts->devices)) return RK_ERR_INVALID;
    std::unique_ptr<rk_stream> s(new rk_stream);
    s->opts = *opts;
    if (opts->n_devices > 0) {
        s->devices.assign(opts->devices, opts->devices + opts->n_devices);
        s->opts.devices = s->devices.data();
    }
    if (opts->params) {  // the blob is copied; tables it points at stay the caller's
        s->params = *opts->params;
        s->opts.params = &s->params;
    }
    if (opts->verify_opts) {
        s->vopts = *opts->verify_opts;
        s->opts.verify_opts = &s->vopts;
    }
    rk_stream* raw = s.get();
    s->worker = std::thread([raw] { raw->run(); });
    *out = s.release();
    return RK_OK;
    RK_GUARD_END
}

int rk_stream_submit(rk_stream* s, const rk_segment* seg, uint32_t* h_seal, size_t seal_capacity_words, size_t* seal_words) {
    RK_GUARD_BEGIN
    if (!s || !seg || !h_seal) return RK_ERR_INVALID;
    std::lock_guard<std::mutex> l(s->mu);
    if (s->closed) return RK_ERR_INVALID;
    s->items.push_back(rk_stream::Item{*seg, h_seal, seal_capacity_words, seal_words});
    s->cv.notify_all();
    return RK_OK;
    RK_GUARD_END
}

int rk_stream_wait(rk_stream* s, size_t max_pending, size_t* finished_prefix) {
    RK_GUARD_BEGIN
    if (!s) return RK_ERR_INVALID;
    std::unique_lock<std::mutex> l(s->mu);
    s->cv.wait(l, [&] { return s->status != RK_OK || s->items.size() - s->done <= max_pending; });
    if (finished_prefix) *finished_prefix = s->done;
    return s->status;
    RK_GUARD_END
}

int rk_stream_close(rk_stream* s, size_t* failed_index) {
    RK_GUARD_BEGIN
    if (failed_index) *failed_index = (size_t)-1;
    if (!s) return RK_ERR_INVALID;
    {
        std::lock_guard<std::mutex> l(s->mu);
        s->closed = true;
        s->cv.notify_all();
    }
    if (s->worker.joinable()) s->worker.join();
    const int st = s->status;
    if (failed_index) *failed_index = s->failed;
    delete s;
    return st;
    RK_GUARD_END
}

// the text is copied into storage of the calling thread: a later session cannot change it under the caller
const char* rk_session_last_error(int device) {
    thread_local std::string text;
    try {
        std::shared_ptr<DevicePool> p;
        {
            std::lock_guard<std::mutex> l(g_mu);
            auto it = g_pools.find(device);
            if (it != g_pools.end()) p = it->second;
        }
        if (!p) return "";
        std::lock_guard<std::mutex> l(g_mu);
        text = p->last_error;
        return text.c_str();
    } catch (...) {
        return "";
    }
}

static std::shared_ptr<DevicePool> pool_of(int device, bool create) {
    std::lock_guard<std::mutex> l(g_mu);
    auto it = g_pools.find(device);
    if (it != g_pools.end()) return it->second;
    if (!create) return nullptr;
    auto sp = std::make_shared<DevicePool>();
    sp->device = device;
    g_pools[device] = sp;
    return sp;
}
int rk_session_set_kernel_timing(int device, int enabled) {
    RK_GUARD_BEGIN
    if (device < 0) return RK_ERR_INVALID;
    auto p = pool_of(device, true);
    std::lock_guard<std::mutex> s(p->busy);
    p->ktime_on = enabled != 0;
    for (rk_ctx* c : p->provers) RK_TRY(rk_set_kernel_timing(c, enabled));
    return RK_OK;
    RK_GUARD_END
}
int rk_session_kernel_stats(int device, int kclass, rk_kernel_stat* out) {
    RK_GUARD_BEGIN
    if (!out || kclass < 0 || kclass >= RK_KCLASS_COUNT) return RK_ERR_INVALID;
    *out = rk_kernel_stat{0, 0.0, 0.0};
    auto p = pool_of(device, false);
    if (!p) return RK_OK;
    std::lock_guard<std::mutex> s(p->busy);
    for (rk_ctx* c : p->provers) {
        rk_kernel_stat st{};
        RK_TRY(rk_kernel_stats(c, kclass, &st));
        out->launches += st.launches;
        out->ms += st.ms;
        out->bytes += st.bytes;
    }
    return RK_OK;
    RK_GUARD_END
}

// segments `device` proved in the last session it took part in (a diagnosis of the work queue's balance)
int rk_session_last_proven(int device, size_t* count) {
    RK_GUARD_BEGIN
    if (!count) return RK_ERR_INVALID;
    *count = 0;
    auto p = pool_of(device, false);
    if (!p) return RK_OK;
    std::lock_guard<std::mutex> s(p->busy);
    *count = p->last_proven;
    return RK_OK;
    RK_GUARD_END
}

int rk_session_release(void) {
    RK_GUARD_BEGIN
    rk::p3_release_pools();
    std::map<int, std::shared_ptr<DevicePool>> pools;
    {
        std::lock_guard<std::mutex> l(g_mu);
        pools.swap(g_pools);
    }
    for (auto& kv : pools) {
        std::lock_guard<std::mutex> s(kv.second->busy);  // wait for a running session of this device
        kv.second->clear();
    }
    return RK_OK;
    RK_GUARD_END
}

}  // extern "C"
