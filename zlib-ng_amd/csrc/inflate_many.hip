// inflate_many.hip -- many independent raw deflate streams: host decode on T threads, device resolution in batches.
//
// The sequential DEFLATE bitstream stays on the host (BASELINE.json north_star), and the host decode is where an
// inflate spends its time: 0.27 s of the 0.286 s a 256 MiB stream takes end to end, against 6 ms on the device
// (DESIGN.md section 5).  One stream cannot be decoded in parallel -- every code's position depends on the one before
// it (inffast_tpl.h:151-298) -- but independent streams can, which is how the reference is used at scale (pigz, one
// zlib stream per thread: test/pigz/CMakeLists.txt).
//
//   workers     T host threads; each takes the next stream and decodes it to tokens (inflate_host.cpp) into a pinned,
//               reusable token set, then hands the set to the dispatcher and goes on to its next stream
//   dispatcher  the calling thread; takes whatever sets are ready, lays the streams out one behind the other in symbol
//               space (inflate_resolve.hip: inflate_resolve_batch), copies tokens and literals to the device (plain
//               DMA from pinned memory) and runs ONE set of launches for the whole batch; two batches can be in flight
//
// Why batches: the device stage of one small stream is four short, latency-bound launches (a 4 MiB stream keeps 128
// wavefronts busy for ~2 ms), and a HIP process has a handful of hardware queues, so "one stream per worker, each on
// its own HIP stream" topped out at 4.9 GB/s of output however many threads decoded (7.9 GB/s with
// GPU_MAX_HW_QUEUES=16; the decode alone reaches 13.7 GB/s on 16 threads).  A batch fills the machine instead.
#include "context.h"
#include "inflate_threads.h"

#include <string.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

extern "C" int zng_rocm_inflate_resolve_window_dev(const uint32_t *d_tokens, size_t ntokens, const uint8_t *d_literals,
                                                   size_t nliterals, const uint64_t *d_segs, size_t nsegs,
                                                   uint16_t *d_symbols, uint8_t *d_out, uint64_t out_len,
                                                   const uint8_t *d_window, uint32_t window_len, void *stream);
int zr_inflate_decode_reuse(const uint8_t *src, size_t src_len, uint32_t window_len, zng_rocm_inflate_tokens *t,
                            size_t caps[3], void *(*re)(void *, size_t, size_t));

namespace zr {

// The token arrays live in PINNED host memory and are kept across streams and calls: the copy to the device is a
// plain DMA (a pageable source goes through the runtime's one staging buffer, which serialises the threads) and no
// stream pays page faults for fresh arrays.  Growth = new pinned block, copy, free.
static void *pinned_realloc(void *old, size_t old_bytes, size_t new_bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, new_bytes, hipHostMallocPortable) != hipSuccess) return nullptr;      // any thread, any device
    if (old) {
        memcpy(p, old, old_bytes < new_bytes ? old_bytes : new_bytes);
        (void)hipHostFree(old);
    }
    return p;
}

struct TokenSet {                             // one decoded stream on its way to the device
    zng_rocm_inflate_tokens tk;
    size_t caps[3];
    size_t job;
};

struct DevBuf {                               // growable device / pinned buffer
    void  *p = nullptr;
    size_t cap = 0;
    bool   pinned = false;
    int reserve(size_t bytes) {
        if (bytes <= cap) return ZNG_ROCM_OK;
        if (p) (void)(pinned ? hipHostFree(p) : hipFree(p));
        p = nullptr;
        cap = 0;
        const size_t want = bytes + (bytes >> 2) + 4096;
        hipError_t e = pinned ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
        if (e != hipSuccess) {
            set_error("inflate batch buffer of %zu bytes: %s", want, hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? ZNG_ROCM_ENOMEM : ZNG_ROCM_EHIP;
        }
        cap = want;
        return ZNG_ROCM_OK;
    }
    void release() {
        if (p) (void)(pinned ? hipHostFree(p) : hipFree(p));
        p = nullptr;
        cap = 0;
    }
};

struct BatchSlot {                            // one batch in flight
    DevBuf d_tokens, d_literals, d_meta, d_sym, h_meta;
    hipEvent_t done = nullptr;
    bool busy = false;
    std::vector<TokenSet *> sets;             // go back to the free list when `done` has fired
};

// Everything a call needs besides its threads; created once, pooled between calls (creating it costs milliseconds:
// pinning pages, device allocations -- and the driver serialises such calls across threads).
struct Engine {
    uint64_t    generation = 0;
    hipStream_t st = nullptr;
    BatchSlot   slot[2];
    std::vector<TokenSet *> all_sets;
    std::vector<ZrPart> parts;                // the parts of the multi-threaded single-stream decode (pinned arrays)
};

static std::mutex g_pool_mu;
static std::vector<Engine *> g_pool;

static void destroy_engine(Engine *e, bool device_alive) {
    if (device_alive) {
        for (BatchSlot &s : e->slot) {
            s.d_tokens.release();
            s.d_literals.release();
            s.d_meta.release();
            s.d_sym.release();
            s.h_meta.release();
            if (s.done) (void)hipEventDestroy(s.done);
        }
        for (TokenSet *ts : e->all_sets) {
            if (ts->tk.tokens) (void)hipHostFree(ts->tk.tokens);
            if (ts->tk.literals) (void)hipHostFree(ts->tk.literals);
            if (ts->tk.segs) (void)hipHostFree(ts->tk.segs);
        }
        for (ZrPart &p : e->parts) {
            if (p.tk.tokens) (void)hipHostFree(p.tk.tokens);
            if (p.tk.literals) (void)hipHostFree(p.tk.literals);
            if (p.tk.segs) (void)hipHostFree(p.tk.segs);
        }
        if (e->st) {
            (void)zng_rocm_stream_release(e->st);
            (void)hipStreamDestroy(e->st);
        }
    }
    for (TokenSet *ts : e->all_sets) delete ts;
    delete e;
}

static Engine *acquire_engine() {
    Context *c = ctx();
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        while (!g_pool.empty()) {
            Engine *e = g_pool.back();
            g_pool.pop_back();
            if (e->generation == c->generation) return e;
            destroy_engine(e, false);          // left over from a context that was shut down: its device state is gone
        }
    }
    Engine *e = new Engine();
    e->generation = c->generation;
    for (BatchSlot &s : e->slot) s.h_meta.pinned = true;
    if (hipStreamCreateWithFlags(&e->st, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&e->slot[0].done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->slot[1].done, hipEventDisableTiming) != hipSuccess) {
        set_error("stream / event creation failed for the batched inflate");
        destroy_engine(e, true);
        return nullptr;
    }
    return e;
}

static void release_engine(Engine *e) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool.push_back(e);
}

void inflate_pool_shutdown() {                // zng_rocm_shutdown: while the device is still alive
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (Engine *e : g_pool) destroy_engine(e, true);
    g_pool.clear();
}

struct Shared {                               // between the workers and the dispatcher of one call
    std::mutex mu;
    std::condition_variable cv_free, cv_ready;
    std::vector<TokenSet *> free_sets;
    std::deque<TokenSet *> ready;
    size_t workers_running = 0;
    bool abort = false;
};

static void inflate_worker(zng_rocm_inflate_job *jobs, size_t njobs, std::atomic<size_t> *next, Shared *sh) {
    DeviceGuard dev;                          // the pinned (re)allocations belong to the backend's device
    for (;;) {
        const size_t i = next->fetch_add(1, std::memory_order_relaxed);
        if (i >= njobs) break;
        TokenSet *ts = nullptr;
        {
            std::unique_lock<std::mutex> lk(sh->mu);
            sh->cv_free.wait(lk, [&] { return !sh->free_sets.empty() || sh->abort; });
            if (sh->abort) break;
            ts = sh->free_sets.back();
            sh->free_sets.pop_back();
        }
        zng_rocm_inflate_job &j = jobs[i];
        const int status = zr_inflate_decode_reuse(j.src, j.src_len, j.window_len, &ts->tk, ts->caps, pinned_realloc);
        j.out_len = ts->tk.out_len;
        j.in_used = ts->tk.in_used;
        j.status = status;
        j.msg = ts->tk.msg;
        bool to_device = status != -4 && ts->tk.out_len != 0;
        if (to_device && ts->tk.out_len > j.dst_cap) {
            j.status = -5;                    // Z_BUF_ERROR: the destination is too small
            j.msg = "output buffer full";
            to_device = false;
        }
        ts->job = i;
        std::lock_guard<std::mutex> lk(sh->mu);
        if (to_device) {
            sh->ready.push_back(ts);
            sh->cv_ready.notify_one();
        } else {
            sh->free_sets.push_back(ts);
            sh->cv_free.notify_one();
        }
    }
    std::lock_guard<std::mutex> lk(sh->mu);
    --sh->workers_running;
    sh->cv_ready.notify_one();
}

constexpr uint64_t kBatchSymbols = 96ull << 20;     // a batch holds at most this much output (its symbols take twice that)
constexpr size_t kBatchStreams = 1024;

struct BatchStreamHost {                      // = BatchStream of inflate_resolve.hip
    uint64_t       v_start;
    const uint8_t *d_window;
    uint64_t       window_len;
};

// lay out, copy and launch one batch; the sets stay with the slot until its event has fired
static int launch_batch(Engine *e, BatchSlot &slot, std::vector<TokenSet *> &sets, zng_rocm_inflate_job *jobs) {
    const size_t ns = sets.size();
    size_t ntok = 0, nlit = 0, nsegs = 0;
    uint64_t nsym = 0;
    for (TokenSet *ts : sets) {
        ntok += ts->tk.ntokens;
        nlit += ts->tk.nliterals;
        nsegs += ts->tk.nsegs;
        nsym += ts->tk.out_len + 32768u;                                // the stream and the window gap behind it
    }
    // meta: segs (nsegs + 1 triples) | seg_dst | seg_end | streams
    const size_t segs_w = 3 * (nsegs + 1), meta_w = segs_w + 2 * nsegs + 3 * ns;
    if (int rc = slot.h_meta.reserve(meta_w * 8)) return rc;
    if (int rc = slot.d_meta.reserve(meta_w * 8)) return rc;
    if (int rc = slot.d_tokens.reserve(ntok * 4 + 256)) return rc;
    if (int rc = slot.d_literals.reserve(nlit + 256)) return rc;
    if (int rc = slot.d_sym.reserve((nsym + 32768u) * 2)) return rc;
    uint64_t *segs = (uint64_t *)slot.h_meta.p, *seg_dst = segs + segs_w, *seg_end = seg_dst + nsegs;
    BatchStreamHost *streams = (BatchStreamHost *)(seg_end + nsegs);
    uint32_t *d_tokens = (uint32_t *)slot.d_tokens.p;
    uint8_t *d_literals = (uint8_t *)slot.d_literals.p;
    uint64_t tok0 = 0, lit0 = 0, v = 0;
    size_t seg0 = 0;
    for (size_t k = 0; k < ns; ++k) {
        const zng_rocm_inflate_tokens &tk = sets[k]->tk;
        const zng_rocm_inflate_job &j = jobs[sets[k]->job];
        for (size_t s = 0; s < tk.nsegs; ++s) {
            segs[3 * (seg0 + s)] = tk.segs[3 * s] + tok0;
            segs[3 * (seg0 + s) + 1] = tk.segs[3 * s + 1] + v;
            segs[3 * (seg0 + s) + 2] = tk.segs[3 * s + 2] + lit0;
            seg_dst[seg0 + s] = (uint64_t)(uintptr_t)(j.d_dst + tk.segs[3 * s + 1]);
            seg_end[seg0 + s] = tk.segs[3 * s + 4] + v;                  // the stream's own (nsegs + 1)-th triple ends its last segment
        }
        streams[k] = BatchStreamHost{v, j.d_window, j.window_len};
        ZR_HIP(hipMemcpyAsync(d_tokens + tok0, tk.tokens, tk.ntokens * 4, hipMemcpyHostToDevice, e->st));
        if (tk.nliterals) ZR_HIP(hipMemcpyAsync(d_literals + lit0, tk.literals, tk.nliterals, hipMemcpyHostToDevice, e->st));
        tok0 += tk.ntokens;
        lit0 += tk.nliterals;
        seg0 += tk.nsegs;
        v += tk.out_len + 32768u;             // the next stream starts behind this one's gap = ITS window
    }
    segs[3 * nsegs] = tok0;                   // the batch's terminal triple: behind the last gap
    segs[3 * nsegs + 1] = v;
    segs[3 * nsegs + 2] = lit0;
    ZR_HIP(hipMemcpyAsync(slot.d_meta.p, slot.h_meta.p, meta_w * 8, hipMemcpyHostToDevice, e->st));
    const uint64_t *d_segs = (const uint64_t *)slot.d_meta.p;
    uint16_t *sym = (uint16_t *)slot.d_sym.p + 32768;                    // the first stream's window sits in front
    if (int rc = inflate_resolve_batch(d_tokens, d_literals, (size_t)lit0, d_segs, nsegs, sym, d_segs + segs_w,
                                       d_segs + segs_w + nsegs, d_segs + segs_w + 2 * nsegs, ns, e->st))
        return rc;
    ZR_HIP(hipEventRecord(slot.done, e->st));
    slot.busy = true;
    slot.sets.swap(sets);
    return ZNG_ROCM_OK;
}

}  // namespace zr

using namespace zr;

extern "C" {

int zng_rocm_inflate_many(zng_rocm_inflate_job *jobs, size_t njobs, int nthreads) {
    if (!ctx()) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    if (!njobs) return ZNG_ROCM_OK;
    if (!jobs) return ZNG_ROCM_EINVAL;
    for (size_t i = 0; i < njobs; ++i) {
        if ((!jobs[i].src && jobs[i].src_len) || (!jobs[i].d_dst && jobs[i].dst_cap) || jobs[i].window_len > 32768u ||
            (jobs[i].window_len && !jobs[i].d_window)) {
            set_error("inflate job %zu: null buffer or window above 32768 bytes", i);
            return ZNG_ROCM_EINVAL;
        }
    }
    DeviceGuard dev;
    unsigned t = nthreads > 0 ? (unsigned)nthreads : zr_default_threads();
    if (t > njobs) t = (unsigned)njobs;
    Engine *e = acquire_engine();
    if (!e) return ZNG_ROCM_EHIP;
    while (e->all_sets.size() < 3 * (size_t)t + 2) {                     // decode runs ahead of the device by up to two sets per thread
        TokenSet *ts = new TokenSet();
        memset(ts, 0, sizeof(*ts));
        e->all_sets.push_back(ts);
    }
    Shared sh;
    sh.free_sets = e->all_sets;
    sh.workers_running = t;
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    pool.reserve(t);
    for (unsigned k = 0; k < t; ++k) pool.emplace_back(inflate_worker, jobs, njobs, &next, &sh);

    int rc = ZNG_ROCM_OK;
    int which = 0;
    auto retire = [&](BatchSlot &slot) {                                  // wait for a slot and hand its sets back
        if (!slot.busy) return;
        if (hipEventSynchronize(slot.done) != hipSuccess && rc == ZNG_ROCM_OK) rc = ZNG_ROCM_EHIP;
        slot.busy = false;
        std::lock_guard<std::mutex> lk(sh.mu);
        for (TokenSet *ts : slot.sets) sh.free_sets.push_back(ts);
        slot.sets.clear();
        sh.cv_free.notify_all();
    };
    for (;;) {
        std::vector<TokenSet *> batch;
        bool finished = false, idle = false;
        {
            std::unique_lock<std::mutex> lk(sh.mu);
            if (sh.ready.empty()) {
                if (sh.workers_running == 0) {
                    finished = true;
                } else if (e->slot[0].busy || e->slot[1].busy) {
                    idle = true;              // nothing to launch: use the time to take a finished batch's sets back
                } else {
                    sh.cv_ready.wait(lk, [&] { return !sh.ready.empty() || sh.workers_running == 0; });
                }
            }
            uint64_t sym = 0;
            while (!finished && !idle && !sh.ready.empty() && batch.size() < kBatchStreams) {
                TokenSet *ts = sh.ready.front();
                if (!batch.empty() && sym + ts->tk.out_len > kBatchSymbols) break;
                sym += ts->tk.out_len;
                batch.push_back(ts);
                sh.ready.pop_front();
            }
        }
        if (finished) break;
        if (idle) {                           // the workers may all be waiting for exactly these sets
            retire(e->slot[which].busy ? e->slot[which] : e->slot[which ^ 1]);
            continue;
        }
        if (batch.empty()) continue;
        BatchSlot &slot = e->slot[which];
        which ^= 1;
        retire(slot);                                                     // its buffers are about to be rewritten
        const int brc = launch_batch(e, slot, batch, jobs);
        if (brc != ZNG_ROCM_OK) {
            if (rc == ZNG_ROCM_OK) rc = brc;
            (void)hipStreamSynchronize(e->st);                            // copies of this batch may have been issued
            std::lock_guard<std::mutex> lk(sh.mu);
            for (TokenSet *ts : batch) {
                jobs[ts->job].status = brc;
                jobs[ts->job].msg = "device stage failed";
                sh.free_sets.push_back(ts);
            }
            sh.cv_free.notify_all();
        }
    }
    retire(e->slot[0]);
    retire(e->slot[1]);
    {
        std::lock_guard<std::mutex> lk(sh.mu);
        sh.abort = true;
        sh.cv_free.notify_all();
    }
    for (auto &th : pool) th.join();
    if (hipStreamSynchronize(e->st) != hipSuccess && rc == ZNG_ROCM_OK) rc = ZNG_ROCM_EHIP;
    release_engine(e);
    return rc;
}


// One raw stream, decoded on `nthreads` host threads (inflate_threads.cpp), resolved on the device in one pass.
int zng_rocm_inflate_raw_threads(const uint8_t *src, size_t src_len, const uint8_t *d_window, uint32_t window_len,
                                 uint8_t *d_dst, size_t dst_cap, uint64_t *out_len, size_t *in_used, int nthreads) {
    if (!ctx()) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    if ((!src && src_len) || window_len > 32768u || (window_len && !d_window)) return ZNG_ROCM_EINVAL;
    DeviceGuard dev;
    const unsigned t = nthreads > 0 ? (unsigned)nthreads : zr_default_threads();
    // one thread: nothing to cut the stream for -- the one-shot path (which takes a stream of 4 MiB and more up to the device
    // and decodes it there, inflate_large.hip; a shorter one on this thread)
    if (t == 1) return zng_rocm_inflate_raw_window(src, src_len, d_window, window_len, d_dst, dst_cap, out_len, in_used, nullptr);
    Engine *e = acquire_engine();
    if (!e) return ZNG_ROCM_EHIP;
    const size_t max_parts = 8u * (size_t)t;
    while (e->parts.size() < max_parts) {
        ZrPart p;
        memset(&p, 0, sizeof(p));
        e->parts.push_back(p);
    }
    std::vector<size_t> chain;
    ZrThreadsResult res;
    memset(&res, 0, sizeof(res));
    const int drc = zr_inflate_decode_threads(src, src_len, window_len, t, pinned_realloc, e->parts.data(), max_parts,
                                              &chain, &res);
    zr_inflate_note_parts(drc == ZR_THREADS_OK ? (int)chain.size() : 0);
    if (drc != ZR_THREADS_OK) {
        release_engine(e);
        if (drc == -4) return ZNG_ROCM_ENOMEM;
        // nothing to cut it at, or something irregular on the way: the one-thread path reports exactly what happened
        return zng_rocm_inflate_raw_window(src, src_len, d_window, window_len, d_dst, dst_cap, out_len, in_used, nullptr);
    }
    if (out_len) *out_len = res.out_len;
    if (in_used) *in_used = res.in_used;
    if (res.out_len > dst_cap) {
        set_error("inflate output (%llu bytes) exceeds dst_cap", (unsigned long long)res.out_len);
        release_engine(e);
        return -5;
    }
    int rc = ZNG_ROCM_OK;
    if (res.out_len) {
        // join the parts on the device: tokens and literals go up part by part (DMA from the parts' pinned arrays);
        // a part's first segment is merged into the last segment of the part before it (that one may hold less than
        // 32 KiB; every segment but the stream's last must hold more)
        size_t ntok = 0, nlit = 0, nseg = 0;
        for (size_t c = 0; c < chain.size(); ++c) {
            const zng_rocm_inflate_tokens &tk = e->parts[chain[c]].tk;
            ntok += tk.ntokens;
            nlit += tk.nliterals;
            nseg += tk.nsegs - (c ? 1 : 0);
        }
        BatchSlot &slot = e->slot[0];
        auto go = [&]() -> int {
            if (int r = slot.h_meta.reserve((nseg + 1) * 24)) return r;
            if (int r = slot.d_meta.reserve((nseg + 1) * 24)) return r;
            if (int r = slot.d_tokens.reserve(ntok * 4 + 256)) return r;
            if (int r = slot.d_literals.reserve(nlit + 256)) return r;
            if (int r = slot.d_sym.reserve(((size_t)res.out_len + 32768u) * 2)) return r;
            uint64_t *segs = (uint64_t *)slot.h_meta.p;
            uint32_t *d_tokens = (uint32_t *)slot.d_tokens.p;
            uint8_t *d_literals = (uint8_t *)slot.d_literals.p;
            uint64_t tok0 = 0, lit0 = 0, o0 = 0;
            size_t s0 = 0;
            for (size_t c = 0; c < chain.size(); ++c) {
                const zng_rocm_inflate_tokens &tk = e->parts[chain[c]].tk;
                if (tk.ntokens) ZR_HIP(hipMemcpyAsync(d_tokens + tok0, tk.tokens, tk.ntokens * 4, hipMemcpyHostToDevice, e->st));
                if (tk.nliterals) ZR_HIP(hipMemcpyAsync(d_literals + lit0, tk.literals, tk.nliterals, hipMemcpyHostToDevice, e->st));
                for (size_t sg = c ? 1 : 0; sg < tk.nsegs; ++sg, ++s0) {
                    segs[3 * s0] = tk.segs[3 * sg] + tok0;
                    segs[3 * s0 + 1] = tk.segs[3 * sg + 1] + o0;
                    segs[3 * s0 + 2] = tk.segs[3 * sg + 2] + lit0;
                }
                tok0 += tk.ntokens;
                lit0 += tk.nliterals;
                o0 += tk.out_len;
            }
            segs[3 * s0] = tok0;
            segs[3 * s0 + 1] = o0;
            segs[3 * s0 + 2] = lit0;
            ZR_HIP(hipMemcpyAsync(slot.d_meta.p, slot.h_meta.p, (nseg + 1) * 24, hipMemcpyHostToDevice, e->st));
            if (int r = zng_rocm_inflate_resolve_window_dev(d_tokens, ntok, d_literals, nlit, (const uint64_t *)slot.d_meta.p, nseg,
                                                            (uint16_t *)slot.d_sym.p, d_dst, res.out_len, d_window, window_len,
                                                            e->st))
                return r;
            return ZNG_ROCM_OK;
        };
        rc = go();
        if (hipStreamSynchronize(e->st) != hipSuccess && rc == ZNG_ROCM_OK) rc = ZNG_ROCM_EHIP;
    }
    release_engine(e);
    return rc != ZNG_ROCM_OK ? rc : res.status;
}

}  // extern "C"
