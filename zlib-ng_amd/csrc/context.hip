// context.hip -- lifecycle of the arch/rocm backend (zng_rocm_init & friends).
//
// Plays the role cpu_features.c / x86_features.c play for the CPU tiers
// (cpu_features.h:23-37, x86_features.c:69-117): probe the device once, then
// the functable can point at the zng_rocm_* slots.
#include "context.h"
#include "tables.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <vector>

namespace zr {

static std::mutex g_mu;
static std::atomic<Context *> g_ctx{nullptr};
static std::map<hipStream_t, Workspace *> g_ws;
static uint64_t g_generation = 0;
static thread_local char g_err[512] = "";
static char g_err_global[512] = "";
static std::mutex g_err_mu;

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    std::lock_guard<std::mutex> lk(g_err_mu);
    memcpy(g_err_global, g_err, sizeof(g_err));
}

[[noreturn]] void die(const char *what) {
    fprintf(stderr, "libzng_rocm: fatal: %s: %s\n", what, g_err[0] ? g_err : g_err_global);
    abort();
}

Context *ctx() { return g_ctx.load(std::memory_order_acquire); }

DeviceGuard::DeviceGuard() {
    Context *c = ctx();
    if (!c) return;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != c->device) switched = hipSetDevice(c->device) == hipSuccess;
}

DeviceGuard::~DeviceGuard() {
    if (switched && prev >= 0) (void)hipSetDevice(prev);
}

// ---- tracing ---------------------------------------------------------------
// The event lists are only touched under g_tr_mu; g_tr_on is the cheap gate the launch paths test.
static std::mutex g_tr_mu;
static std::vector<hipEvent_t> g_tr_start, g_tr_stop;
static int g_tr_cap = 0, g_tr_n = 0, g_tr_stride = 1, g_tr_seen = 0;
static std::atomic<bool> g_tr_on{false};

void trace_pick(hipEvent_t *start, hipEvent_t *stop) {
    if (!g_tr_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_tr_mu);
    if (!g_tr_on.load(std::memory_order_relaxed) || g_tr_n >= g_tr_cap) return;
    if (g_tr_seen++ % g_tr_stride) return;
    *start = g_tr_start[g_tr_n];
    *stop = g_tr_stop[g_tr_n];
    ++g_tr_n;
}

static void free_workspace(Workspace *ws) {
    if (ws->partials) (void)hipFree(ws->partials);
    if (ws->result) (void)hipFree(ws->result);
    if (ws->acc) (void)hipFree(ws->acc);
    if (ws->pinned) (void)hipHostFree(ws->pinned);
    if (ws->stage) (void)hipFree(ws->stage);
    for (int i = 0; i < kScrCount; ++i) {
        Scratch &sc = ws->scratch[i];
        if (sc.p) (void)(sc.host ? hipHostFree(sc.p) : hipFree(sc.p));
    }
    if (ws->host_done) (void)hipEventDestroy(ws->host_done);
    delete ws;
}

Workspace *workspace_for(hipStream_t s) {
    Context *c = ctx();
    if (!c) {
        set_error("zng_rocm_init() has not succeeded");
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ws.find(s);
    if (it != g_ws.end()) return it->second;
    DeviceGuard dev;
    if (s) {
        hipDevice_t sd = -1;
        if (hipStreamGetDevice(s, &sd) == hipSuccess && (int)sd != c->device) {
            set_error("the stream belongs to device %d, the backend was initialised on device %d", (int)sd, c->device);
            return nullptr;
        }
    }
    Workspace *ws = new Workspace();
    ws->partials = nullptr;
    ws->acc = ws->result = ws->pinned = nullptr;
    ws->stage = nullptr;
    ws->stage_bytes = 0;
    memset(ws->scratch, 0, sizeof(ws->scratch));
    ws->host_done = nullptr;
    ws->host_busy = false;
    if (hipMalloc(&ws->partials, sizeof(Partial) * kMaxGroups) != hipSuccess ||
        hipMalloc(&ws->result, 64) != hipSuccess ||
        hipMalloc(&ws->acc, 64) != hipSuccess || hipMemset(ws->acc, 0, 64) != hipSuccess ||
        hipHostMalloc(&ws->pinned, 64, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&ws->host_done, hipEventDisableTiming) != hipSuccess) {
        set_error("workspace allocation failed: %s", hipGetErrorString(hipGetLastError()));
        free_workspace(ws);
        return nullptr;
    }
    g_ws[s] = ws;
    return ws;
}

int ensure_stage(Workspace *ws, size_t bytes) {
    if (bytes <= ws->stage_bytes) return ZNG_ROCM_OK;
    if (ws->stage) (void)hipFree(ws->stage);           // hipFree waits for the device: nothing is still reading it
    ws->stage = nullptr;
    ws->stage_bytes = 0;
    ZR_HIP(hipMalloc(&ws->stage, bytes));
    ws->stage_bytes = bytes;
    return ZNG_ROCM_OK;
}

int scratch_reserve(Workspace *ws, int slot, size_t bytes, bool pinned_host, void **out) {
    Scratch &sc = ws->scratch[slot];
    if (bytes > sc.cap || !sc.p) {
        const size_t want = bytes + (bytes >> 3) + 256;
        if (sc.p) {
            // hipFree / hipHostFree wait for outstanding work, so a kernel still running on this scratch finishes first
            (void)(sc.host ? hipHostFree(sc.p) : hipFree(sc.p));
            sc.p = nullptr;
            sc.cap = 0;
        }
        if (pinned_host) ZR_HIP(hipHostMalloc(&sc.p, want, hipHostMallocDefault));
        else ZR_HIP(hipMalloc(&sc.p, want));
        sc.cap = want;
        sc.host = pinned_host;
    }
    *out = sc.p;
    return ZNG_ROCM_OK;
}

int host_tables_acquire(Workspace *ws) {
    if (ws->host_busy) {
        ZR_HIP(hipEventSynchronize(ws->host_done));
        ws->host_busy = false;
    }
    return ZNG_ROCM_OK;
}

int host_tables_release(Workspace *ws, hipStream_t s) {
    ZR_HIP(hipEventRecord(ws->host_done, s));
    ws->host_busy = true;
    return ZNG_ROCM_OK;
}

}  // namespace zr

using namespace zr;

extern "C" {

int zng_rocm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int zng_rocm_available(void) { return ctx() != nullptr; }

const char *zng_rocm_last_error(void) { return g_err[0] ? g_err : g_err_global; }

int zng_rocm_init(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (ctx()) return ZNG_ROCM_OK;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device visible");
        return ZNG_ROCM_ENODEV;
    }
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (device < 0) device = prev >= 0 ? prev : 0;
    if (device >= n) {
        set_error("device %d out of range (%d visible)", device, n);
        return ZNG_ROCM_EINVAL;
    }
    hipDeviceProp_t prop;
    ZR_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library carries gfx950 code objects only", device, prop.gcnArchName);
        return ZNG_ROCM_ENODEV;
    }
    ZR_HIP(hipSetDevice(device));
    Context *c = new Context();
    c->generation = ++g_generation;
    c->device = device;
    c->cus = prop.multiProcessorCount;
    c->lds_bytes = (int)prop.maxSharedMemoryPerMultiProcessor;
    int xccs = 0;
    if (hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, device) != hipSuccess || xccs <= 0) xccs = 1;
    c->xcds = xccs;
    build_tables(c->host_tables);
    hipError_t e = hipMalloc(&c->tables, sizeof(DeviceTables));
    if (e == hipSuccess) e = hipMemcpy(c->tables, &c->host_tables, sizeof(DeviceTables), hipMemcpyHostToDevice);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);       // the caller's current device is left as it was
    if (e != hipSuccess) {
        set_error("constant tables: %s", hipGetErrorString(e));
        delete c;
        return ZNG_ROCM_EHIP;
    }
    g_ctx.store(c, std::memory_order_release);
    return ZNG_ROCM_OK;
}

int zng_rocm_device_info(int32_t out[4]) {
    Context *c = ctx();
    if (!c) return ZNG_ROCM_ENODEV;
    out[0] = c->cus;
    out[1] = c->lds_bytes;
    out[2] = 64;
    out[3] = c->xcds;
    return ZNG_ROCM_OK;
}

int zng_rocm_trace_begin(int max_launches) {
    if (!ctx()) return ZNG_ROCM_ENODEV;
    if (max_launches <= 0) return ZNG_ROCM_EINVAL;
    DeviceGuard dev;
    std::lock_guard<std::mutex> lk(g_tr_mu);
    while ((int)g_tr_start.size() < max_launches) {
        hipEvent_t a, b;
        ZR_HIP(hipEventCreate(&a));
        ZR_HIP(hipEventCreate(&b));
        g_tr_start.push_back(a);
        g_tr_stop.push_back(b);
    }
    g_tr_cap = max_launches;
    g_tr_n = g_tr_seen = 0;
    g_tr_on.store(true);
    return ZNG_ROCM_OK;
}

int zng_rocm_trace_end(float *ms_out, int cap) {
    if (!ctx()) return ZNG_ROCM_ENODEV;
    DeviceGuard dev;
    g_tr_on.store(false);
    ZR_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_tr_mu);
    int n = g_tr_n < cap ? g_tr_n : cap;
    for (int i = 0; i < n; ++i) ZR_HIP(hipEventElapsedTime(&ms_out[i], g_tr_start[i], g_tr_stop[i]));
    return n;
}

int zng_rocm_trace_stride(int n) {
    if (n < 1) return ZNG_ROCM_EINVAL;
    std::lock_guard<std::mutex> lk(g_tr_mu);
    g_tr_stride = n;
    return ZNG_ROCM_OK;
}

int zng_rocm_stream_release(void *stream) {
    Context *c = ctx();
    if (!c) return ZNG_ROCM_OK;
    DeviceGuard dev;
    Workspace *ws = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_ws.find((hipStream_t)stream);
        if (it == g_ws.end()) return ZNG_ROCM_OK;
        ws = it->second;
        g_ws.erase(it);
    }
    (void)hipStreamSynchronize((hipStream_t)stream);
    free_workspace(ws);
    return ZNG_ROCM_OK;
}

int zng_rocm_shutdown(void) {
    Context *c = ctx();
    if (!c) return ZNG_ROCM_OK;
    DeviceGuard dev;
    (void)hipDeviceSynchronize();
    inflate_pool_shutdown();
    {
        std::lock_guard<std::mutex> lk(g_tr_mu);
        g_tr_on.store(false);
        for (hipEvent_t e : g_tr_start) (void)hipEventDestroy(e);
        for (hipEvent_t e : g_tr_stop) (void)hipEventDestroy(e);
        g_tr_start.clear();
        g_tr_stop.clear();
        g_tr_cap = g_tr_n = 0;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : g_ws) free_workspace(kv.second);
    g_ws.clear();
    checksum_reset_reserved_cus();
    g_ctx.store(nullptr, std::memory_order_release);
    (void)hipFree(c->tables);
    delete c;
    return ZNG_ROCM_OK;
}

}  // extern "C"
