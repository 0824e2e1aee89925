// rk_prove_session: all segments of a session, several in flight on one GPU.
//
// The reference proves a session's segments one after the other (`session.prove()`,
// provers/risc0/driver/src/bonsai.rs:271).  One proof is a chain of ~120 dependent launches with a
// host round trip at every Merkle root, so a lone proof leaves an MI355X partly idle; segments are
// independent, so this entry point runs `inflight` prover contexts (one HIP stream, scratch pool and
// host thread each) over a shared index, and one more context + thread stages host-resident traces
// `upload_ahead` segments ahead into a ring of device buffers, so the PCIe upload of segment i+1
// runs under the proof of segment i.  Each seal is verified (rk_verify_segment, host code) by one
// more thread while the GPU goes on.  Contexts and staging buffers are
// kept per device for the life of the process (rk_session_release frees them): the `Prover` trait
// of the reference has no `self`, a backend's state is process-global (lib/src/prover.rs:52-62).
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "internal.hpp"

namespace {

struct Slot {
    void* group[3] = {nullptr, nullptr, nullptr};
    void* check = nullptr;
    size_t words[4] = {0, 0, 0, 0};
};

struct DevicePool {
    int device = 0;
    std::vector<rk_ctx*> provers;
    rk_ctx* uploader = nullptr;
    std::vector<Slot*> ring;
    std::mutex busy;  // one session at a time per device (`run` may be entered from many threads)
    std::string last_error;
};

std::mutex g_mu;
std::map<int, DevicePool*> g_pools;

void free_slot(rk_ctx* up, Slot* s) {
    for (int g = 0; g < 3; g++) {
        if (s->group[g]) (void)rk_free(up, s->group[g]);
        s->group[g] = nullptr;
    }
    if (s->check) (void)rk_free(up, s->check);
    s->check = nullptr;
    for (auto& w : s->words) w = 0;
}

void destroy_pool(DevicePool* p) {
    if (p->uploader) {
        for (Slot* s : p->ring) {
            free_slot(p->uploader, s);
            delete s;
        }
        (void)rk_ctx_destroy(p->uploader);
    }
    for (rk_ctx* c : p->provers) (void)rk_ctx_destroy(c);
    delete p;
}

// shared state of one rk_prove_session call
struct Run {
    const rk_segment* segs;
    size_t n;
    uint32_t* const* h_seals;
    const size_t* caps;
    size_t* words;
    int verify;

    std::atomic<size_t> next{0};
    std::mutex mu;
    std::condition_variable cv;
    std::map<size_t, Slot*> ready;   // staged segments not yet taken by a prover
    std::deque<Slot*> free_slots;
    std::deque<size_t> to_verify;    // finished seals waiting for the verifier thread
    size_t provers_left = 0;
    bool stop = false;               // stager finished or the run is aborted
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
        stop = true;
        cv.notify_all();
    }
    bool aborted() {
        std::lock_guard<std::mutex> l(mu);
        return status != RK_OK;
    }
};

bool needs_staging(const rk_segment& s) { return s.on_device == 0; }

void stager(DevicePool* pool, Run* run, size_t n_slots) {
    rk_ctx* up = pool->uploader;
    for (size_t i = 0; i < run->n; i++) {
        const rk_segment& seg = run->segs[i];
        if (!needs_staging(seg)) continue;
        Slot* slot = nullptr;
        {
            std::unique_lock<std::mutex> l(run->mu);
            for (;;) {
                if (run->status != RK_OK) return;
                if (!run->free_slots.empty()) {
                    slot = run->free_slots.front();
                    run->free_slots.pop_front();
                    break;
                }
                if (pool->ring.size() < n_slots) {  // grow the ring before waiting for a proof to end
                    slot = new Slot();
                    pool->ring.push_back(slot);
                    break;
                }
                run->cv.wait(l);
            }
        }
        const size_t rows = (size_t)1 << seg.po2;
        const size_t want[4] = {rows * seg.taps.group_size[0], rows * seg.taps.group_size[1],
                                rows * seg.taps.group_size[2], rows * 16};
        if (want[0] != slot->words[0] || want[1] != slot->words[1] || want[2] != slot->words[2] ||
            want[3] != slot->words[3]) {
            free_slot(up, slot);
            int st = RK_OK;
            for (int g = 0; g < 3 && st == RK_OK; g++) st = rk_alloc(up, want[g] * 4, &slot->group[g]);
            if (st == RK_OK) st = rk_alloc(up, want[3] * 4, &slot->check);
            if (st != RK_OK) {
                run->fail(st, i, rk_last_error(up));
                return;
            }
            for (int k = 0; k < 4; k++) slot->words[k] = want[k];
        }
        int st = RK_OK;
        for (int g = 0; g < 3 && st == RK_OK; g++) {
            if (!seg.group[g] && want[g]) st = RK_ERR_INVALID;
            else if (want[g]) st = rk_h2d(up, slot->group[g], seg.group[g], want[g] * 4);
        }
        if (st == RK_OK) st = seg.check ? rk_h2d(up, slot->check, seg.check, want[3] * 4) : RK_ERR_INVALID;
        if (st == RK_OK) st = rk_sync(up);  // the buffers change hands after this
        if (st != RK_OK) {
            run->fail(st, i, rk_last_error(up));
            return;
        }
        std::lock_guard<std::mutex> l(run->mu);
        run->ready[i] = slot;
        run->cv.notify_all();
    }
}

void prover(rk_ctx* ctx, Run* run) {
    for (;;) {
        size_t i = run->next.fetch_add(1);
        if (i >= run->n || run->aborted()) return;
        rk_segment seg = run->segs[i];
        Slot* slot = nullptr;
        if (needs_staging(seg)) {
            std::unique_lock<std::mutex> l(run->mu);
            run->cv.wait(l, [&] { return run->ready.count(i) || run->status != RK_OK; });
            if (run->status != RK_OK) return;
            slot = run->ready[i];
            run->ready.erase(i);
            seg.on_device = 2;  // the staged copy is ours: no second copy inside the prover
            for (int g = 0; g < 3; g++) seg.group[g] = (const uint32_t*)slot->group[g];
            seg.check = (const uint32_t*)slot->check;
        }
        int st = rk_prove_segment(ctx, &seg, run->h_seals[i], run->caps[i], &run->words[i]);
        if (slot) {
            std::lock_guard<std::mutex> l(run->mu);
            run->free_slots.push_back(slot);
            run->cv.notify_all();
        }
        if (st != RK_OK) {
            run->fail(st, i, rk_last_error(ctx));
            return;
        }
        if (run->verify) {  // host work: handed to the verifier thread so this context goes straight on
            std::lock_guard<std::mutex> l(run->mu);
            run->to_verify.push_back(i);
            run->cv.notify_all();
        }
    }
}

void prover_thread(rk_ctx* ctx, Run* run) {
    prover(ctx, run);
    std::lock_guard<std::mutex> l(run->mu);
    run->provers_left--;
    run->cv.notify_all();
}

// rk_verify_segment of every finished seal (~9 ms of host time at S20, against ~26 ms per proof)
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
        if (rk_verify_segment(&run->segs[i], run->h_seals[i], run->words[i]) != 0) {
            run->fail(RK_ERR_VERIFY, i, "seal failed verification");
            return;
        }
    }
}

}  // namespace

extern "C" {

int rk_prove_session(const rk_session_opts* opts, const rk_segment* segs, size_t n, uint32_t* const* h_seals,
                     const size_t* seal_capacity_words, size_t* seal_words, size_t* failed_index) {
    if (failed_index) *failed_index = (size_t)-1;
    if (!opts || (n && (!segs || !h_seals || !seal_capacity_words || !seal_words))) return RK_ERR_INVALID;
    if (opts->inflight < 1 || opts->inflight > 16 || opts->upload_ahead < 0 || opts->upload_ahead > 16) return RK_ERR_INVALID;
    if (n == 0) return RK_OK;
    DevicePool* pool = nullptr;
    {
        std::lock_guard<std::mutex> l(g_mu);
        auto it = g_pools.find(opts->device);
        if (it == g_pools.end()) {
            pool = new DevicePool();
            pool->device = opts->device;
            g_pools[opts->device] = pool;
        } else {
            pool = it->second;
        }
    }
    std::lock_guard<std::mutex> session(pool->busy);
    while ((int)pool->provers.size() < opts->inflight) {
        rk_ctx* c = nullptr;
        int st = rk_ctx_create(opts->device, nullptr, &c);
        if (st != RK_OK) return st;
        pool->provers.push_back(c);
    }
    bool any_host = false;
    for (size_t i = 0; i < n; i++) any_host |= needs_staging(segs[i]);
    if (any_host && !pool->uploader) {
        int st = rk_ctx_create(opts->device, nullptr, &pool->uploader);
        if (st != RK_OK) return st;
    }
    Run run;
    run.segs = segs;
    run.n = n;
    run.h_seals = h_seals;
    run.caps = seal_capacity_words;
    run.words = seal_words;
    run.verify = opts->verify;
    const size_t workers = std::min<size_t>((size_t)opts->inflight, n);
    // with nothing to hide behind (upload_ahead == 0) the ring still needs one slot per prover
    const size_t n_slots = (size_t)opts->upload_ahead + workers;
    for (size_t k = 0; k < pool->ring.size() && k < n_slots; k++) run.free_slots.push_back(pool->ring[k]);
    run.provers_left = workers;
    std::vector<std::thread> threads;
    if (any_host) threads.emplace_back(stager, pool, &run, n_slots);
    for (size_t w = 0; w < workers; w++) threads.emplace_back(prover_thread, pool->provers[w], &run);
    if (run.verify) threads.emplace_back(verifier, &run);
    for (auto& t : threads) t.join();
    if (run.status != RK_OK) {
        if (failed_index) *failed_index = run.failed;
        pool->last_error = run.detail;
    }
    return run.status;
}

const char* rk_session_last_error(int device) {
    std::lock_guard<std::mutex> l(g_mu);
    auto it = g_pools.find(device);
    if (it == g_pools.end()) return "";
    return it->second->last_error.c_str();
}

int rk_session_release(void) {
    std::lock_guard<std::mutex> l(g_mu);
    for (auto& kv : g_pools) {
        kv.second->busy.lock();  // wait for a running session of this device
        kv.second->busy.unlock();
        destroy_pool(kv.second);
    }
    g_pools.clear();
    return RK_OK;
}

}  // extern "C"
