// inflate_many.hip -- many independent raw deflate streams, host decode on T threads, device resolution per stream.
//
// The sequential DEFLATE bitstream stays on the host (BASELINE.json north_star), and the host decode is where an
// inflate spends its time: 0.27 s of the 0.286 s a 256 MiB stream takes end to end, against 6 ms on the device
// (DESIGN.md section 5).  One stream cannot be decoded in parallel -- every code's position depends on the one before
// it (inffast_tpl.h:151-298) -- but independent streams can, which is how the reference is used at scale (pigz, one
// zlib stream per thread: test/pigz/CMakeLists.txt).  Each worker thread takes the next stream, decodes it to tokens
// (inflate_host.cpp), hands tokens and literals to the device on ITS OWN HIP stream and launches the resolution
// kernels there; while those run it is already decoding its next stream.  Nothing is shared between workers but the
// job counter.
#include "context.h"

#include <string.h>

#include <atomic>
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

// The token arrays of a worker live in PINNED host memory and are kept across streams: the copy to the device is a
// plain DMA (a pageable source would go through the runtime's one staging buffer, which serialises the workers:
// measured 2.9 GB/s of output on 16 threads against 13.7 GB/s for the decode alone), and no stream pays page faults
// for fresh arrays.  Growth = new pinned block, copy, free.
static void *pinned_realloc(void *old, size_t old_bytes, size_t new_bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, new_bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    if (old) {
        memcpy(p, old, old_bytes < new_bytes ? old_bytes : new_bytes);
        (void)hipHostFree(old);
    }
    return p;
}

struct TokenSet {                             // one of a worker's two decode targets
    zng_rocm_inflate_tokens tk;
    size_t caps[3];
    hipEvent_t copied;                        // recorded behind the copies that read this set
    bool in_flight;
};

// Everything a worker needs besides its thread: HIP stream, the two pinned token sets, the device work buffer.
// Creating these costs milliseconds (pinning pages, device allocation -- and the driver serialises such calls across
// threads), far more than decoding a few MiB, so they are kept in a pool between calls and reused.
struct WorkerCtx {
    uint64_t    generation;
    hipStream_t st;
    TokenSet    sets[2];
    uint8_t    *d_work;
    size_t      work_cap;
};

static std::mutex g_pool_mu;
static std::vector<WorkerCtx *> g_pool;

static void destroy_worker(WorkerCtx *w, bool device_alive) {
    if (device_alive) {
        if (w->d_work) (void)hipFree(w->d_work);
        for (TokenSet &ts : w->sets) {
            if (ts.tk.tokens) (void)hipHostFree(ts.tk.tokens);
            if (ts.tk.literals) (void)hipHostFree(ts.tk.literals);
            if (ts.tk.segs) (void)hipHostFree(ts.tk.segs);
            if (ts.copied) (void)hipEventDestroy(ts.copied);
        }
        if (w->st) {
            (void)zng_rocm_stream_release(w->st);
            (void)hipStreamDestroy(w->st);
        }
    }
    delete w;
}

static WorkerCtx *acquire_worker() {
    Context *c = ctx();
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        while (!g_pool.empty()) {
            WorkerCtx *w = g_pool.back();
            g_pool.pop_back();
            if (w->generation == c->generation) return w;
            destroy_worker(w, false);          // left over from a context that was shut down: its device state is gone
        }
    }
    WorkerCtx *w = new WorkerCtx();
    memset(w, 0, sizeof(*w));
    w->generation = c->generation;
    if (hipStreamCreateWithFlags(&w->st, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&w->sets[0].copied, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&w->sets[1].copied, hipEventDisableTiming) != hipSuccess) {
        set_error("stream / event creation failed in an inflate worker");
        destroy_worker(w, true);
        return nullptr;
    }
    return w;
}

static void release_worker(WorkerCtx *w) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool.push_back(w);
}

void inflate_pool_shutdown() {                // zng_rocm_shutdown: while the device is still alive
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (WorkerCtx *w : g_pool) destroy_worker(w, true);
    g_pool.clear();
}

static void inflate_worker(zng_rocm_inflate_job *jobs, size_t njobs, std::atomic<size_t> *next, int *first_error) {
    DeviceGuard dev;
    WorkerCtx *w = acquire_worker();
    if (!w) {
        *first_error = ZNG_ROCM_EHIP;
        return;
    }
    hipStream_t st = w->st;
    int which = 0;
    for (;;) {
        const size_t i = next->fetch_add(1, std::memory_order_relaxed);
        if (i >= njobs) break;
        zng_rocm_inflate_job &j = jobs[i];
        TokenSet &ts = w->sets[which];
        which ^= 1;
        if (ts.in_flight) {                   // the copies of two streams ago have long left this set
            (void)hipEventSynchronize(ts.copied);
            ts.in_flight = false;
        }
        zng_rocm_inflate_tokens &tk = ts.tk;
        int status = zr_inflate_decode_reuse(j.src, j.src_len, j.window_len, &tk, ts.caps, pinned_realloc);
        j.out_len = tk.out_len;
        j.in_used = tk.in_used;
        j.status = status;
        j.msg = tk.msg;
        if (status == -4) continue;
        if (tk.out_len > j.dst_cap) {
            j.status = -5;                    // Z_BUF_ERROR: the destination is too small
            j.msg = "output buffer full";
            continue;
        }
        if (!tk.out_len) continue;
        const size_t tok_b = (tk.ntokens * 4 + 255) & ~(size_t)255;
        const size_t seg_b = ((tk.nsegs + 1) * 24 + 255) & ~(size_t)255;
        const size_t lit_b = (tk.nliterals + 255) & ~(size_t)255;
        const size_t sym_b = ((size_t)tk.out_len + 32768) * 2;
        const size_t need = tok_b + seg_b + lit_b + sym_b;
        hipError_t e = hipSuccess;
        if (need > w->work_cap) {
            if (w->d_work) (void)hipFree(w->d_work);    // waits for the previous stream's kernels
            w->d_work = nullptr;
            w->work_cap = 0;
            e = hipMalloc(&w->d_work, need + (need >> 1));
            if (e == hipSuccess) w->work_cap = need + (need >> 1);
        }
        uint8_t *d_work = w->d_work;
        // the copies are ordered behind the previous stream's kernels on this worker's HIP stream, so the device work
        // buffer is reused without a host-side wait; the host goes straight on to decode its next stream
        if (e == hipSuccess) e = hipMemcpyAsync(d_work, tk.tokens, tk.ntokens * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d_work + tok_b, tk.segs, (tk.nsegs + 1) * 24, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && tk.nliterals)
            e = hipMemcpyAsync(d_work + tok_b + seg_b, tk.literals, tk.nliterals, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipEventRecord(ts.copied, st);
        ts.in_flight = e == hipSuccess;
        int rc = ZNG_ROCM_OK;
        if (e != hipSuccess) {
            set_error("inflate worker: %s", hipGetErrorString(e));
            rc = e == hipErrorOutOfMemory ? ZNG_ROCM_ENOMEM : ZNG_ROCM_EHIP;
        } else {
            rc = zng_rocm_inflate_resolve_window_dev((const uint32_t *)d_work, tk.ntokens, d_work + tok_b + seg_b,
                                                     tk.nliterals, (const uint64_t *)(d_work + tok_b), tk.nsegs,
                                                     (uint16_t *)(d_work + tok_b + seg_b + lit_b), j.d_dst, tk.out_len,
                                                     j.d_window, j.window_len, st);
        }
        if (rc != ZNG_ROCM_OK) {
            j.status = rc;
            j.msg = "device stage failed";
            if (!*first_error) *first_error = rc;
        }
    }
    if (hipStreamSynchronize(st) != hipSuccess && !*first_error) *first_error = ZNG_ROCM_EHIP;
    w->sets[0].in_flight = w->sets[1].in_flight = false;
    release_worker(w);
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
    unsigned t = nthreads > 0 ? (unsigned)nthreads : std::thread::hardware_concurrency();
    if (t == 0) t = 1;
    if (t > njobs) t = (unsigned)njobs;
    std::atomic<size_t> next{0};
    std::vector<int> errors(t, 0);
    std::vector<std::thread> pool;
    pool.reserve(t);
    for (unsigned k = 1; k < t; ++k) pool.emplace_back(inflate_worker, jobs, njobs, &next, &errors[k]);
    inflate_worker(jobs, njobs, &next, &errors[0]);          // the calling thread is worker 0
    for (auto &th : pool) th.join();
    for (int e : errors)
        if (e) return e;
    return ZNG_ROCM_OK;
}

}  // extern "C"
