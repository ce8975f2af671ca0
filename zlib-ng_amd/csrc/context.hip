// context.hip -- lifecycle of the arch/rocm backend (zng_rocm_init & friends).
//
// Plays the role cpu_features.c / x86_features.c play for the CPU tiers
// (cpu_features.h:23-37, x86_features.c:69-117): probe the device once, then
// the functable can point at the zng_rocm_* slots.
#include "context.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <vector>
#include <mutex>

namespace zr {

static std::mutex g_mu;
static Context *g_ctx = nullptr;
static std::map<hipStream_t, Workspace *> g_ws;
static thread_local char g_err[512] = "";
static char g_err_global[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    memcpy(g_err_global, g_err, sizeof(g_err));
}

[[noreturn]] void die(const char *what) {
    fprintf(stderr, "libzng_rocm: fatal: %s: %s\n", what, g_err[0] ? g_err : g_err_global);
    abort();
}

Context *ctx() { return g_ctx; }

static void build_tables(DeviceTables &t) {
    // byte table: the shift-register construction of tools/makecrct.c:66-73
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t r = i;
        for (int k = 0; k < 8; ++k) r = (r >> 1) ^ (kCrcPoly & (0u - (r & 1u)));
        t.byte_tab[i] = r;
    }
    // stride tables: tools/makecrct.c:99-112 with the braid stride n*w replaced
    // by this kernel's stride (one 16 KiB workgroup row)
    for (int k = 0; k < 4; ++k) {
        uint32_t adv = xpow_bits(8ull * (uint64_t)(kUnitBytes + 3 - k));
        for (uint32_t b = 0; b < 256; ++b) t.stride_tab[k][b] = mulmod(b << 24, adv);
    }
    // lane weights: x^(8*(U - 16t - 4c)), built incrementally from the far end
    //   w(t,c) with distance d = U - 16t - 4c; d decreases by 4 per (c+1)
    {
        uint32_t x32 = xpow_bits(32);
        uint32_t cur = x32;  // d = 4 : t = kWgThreads-1, c = 3
        for (int lane = kWgThreads - 1; lane >= 0; --lane)
            for (int c = 3; c >= 0; --c) {
                t.lane_weight[lane][c] = cur;
                cur = mulmod(cur, x32);
            }
    }
    for (int lane = 0; lane < kWgThreads; ++lane) {
        uint32_t b = t.lane_weight[lane][3];
        for (int k = 0; k < 32; ++k) {
            t.lane_pow[k][lane] = b;
            b = (b >> 1) ^ (kCrcPoly & (0u - (b & 1u)));                  // times x, reflected representation
        }
    }
    for (int i = 0; i < 2; ++i) {
        uint32_t step = xpow_bits(8ull * (uint64_t)kUnitBytes << (10 * i));   // x^(8 * U * 1024^i)
        uint32_t cur = 0x80000000u;
        for (int d = 0; d < 1024; ++d) {
            t.unit_pow[i][d] = cur;
            cur = mulmod(cur, step);
        }
    }
    for (int i = 0; i < kPowDigits; ++i) {
        uint32_t step = xpow_bits(8ull << (7 * i));      // x^(8 * 128^i)
        uint32_t cur = 0x80000000u;                      // digit 0 -> x^0
        for (int d = 0; d < 128; ++d) {
            t.pow_tab[i * 128 + d] = cur;
            cur = mulmod(cur, step);
        }
    }
}

// ---- tracing ---------------------------------------------------------------
static std::vector<hipEvent_t> g_tr_start, g_tr_stop;
static int g_tr_cap = 0, g_tr_n = 0;
static bool g_tr_on = false;

void trace_mark(hipStream_t s, bool begin) {
    if (!g_tr_on) return;
    if (begin) {
        if (g_tr_n >= g_tr_cap) return;
        (void)hipEventRecord(g_tr_start[g_tr_n], s);
    } else {
        if (g_tr_n >= g_tr_cap) return;
        (void)hipEventRecord(g_tr_stop[g_tr_n], s);
        ++g_tr_n;
    }
}

Workspace *workspace_for(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ws.find(s);
    if (it != g_ws.end()) return it->second;
    Workspace *ws = new Workspace();
    memset(ws, 0, sizeof(*ws));
    if (hipMalloc(&ws->partials, sizeof(Partial) * kMaxGroups) != hipSuccess ||
        hipMalloc(&ws->result, 64) != hipSuccess ||
        hipMalloc(&ws->acc, 64) != hipSuccess || hipMemset(ws->acc, 0, 64) != hipSuccess ||
        hipHostMalloc(&ws->pinned, 64, hipHostMallocDefault) != hipSuccess) {
        set_error("workspace allocation failed");
        delete ws;
        return nullptr;
    }
    g_ws[s] = ws;
    return ws;
}

int ensure_stage(Workspace *ws, size_t bytes) {
    if (bytes <= ws->stage_bytes) return ZNG_ROCM_OK;
    size_t want = bytes + (bytes >> 2) + 4096;
    if (ws->stage) (void)hipFree(ws->stage);
    ws->stage = nullptr;
    ws->stage_bytes = 0;
    ZR_HIP(hipMalloc(&ws->stage, want));
    ws->stage_bytes = want;
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

int zng_rocm_available(void) { return g_ctx != nullptr; }

const char *zng_rocm_last_error(void) { return g_err[0] ? g_err : g_err_global; }

int zng_rocm_init(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ctx) return ZNG_ROCM_OK;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device visible");
        return ZNG_ROCM_ENODEV;
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= n) {
        set_error("device %d out of range (%d visible)", device, n);
        return ZNG_ROCM_EINVAL;
    }
    ZR_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    ZR_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library carries gfx950 code objects only", device, prop.gcnArchName);
        return ZNG_ROCM_ENODEV;
    }
    Context *c = new Context();
    c->device = device;
    c->cus = prop.multiProcessorCount;
    c->lds_bytes = (int)prop.maxSharedMemoryPerMultiProcessor;
    c->xcds = 8;
    build_tables(c->host_tables);
    ZR_HIP(hipMalloc(&c->tables, sizeof(DeviceTables)));
    ZR_HIP(hipMemcpy(c->tables, &c->host_tables, sizeof(DeviceTables), hipMemcpyHostToDevice));
    g_ctx = c;
    return ZNG_ROCM_OK;
}

int zng_rocm_device_info(int32_t out[4]) {
    if (!g_ctx) return ZNG_ROCM_ENODEV;
    out[0] = g_ctx->cus;
    out[1] = g_ctx->lds_bytes;
    out[2] = 64;
    out[3] = g_ctx->xcds;
    return ZNG_ROCM_OK;
}

int zng_rocm_trace_begin(int max_launches) {
    if (!g_ctx) return ZNG_ROCM_ENODEV;
    if (max_launches <= 0) return ZNG_ROCM_EINVAL;
    while ((int)g_tr_start.size() < max_launches) {
        hipEvent_t a, b;
        ZR_HIP(hipEventCreate(&a));
        ZR_HIP(hipEventCreate(&b));
        g_tr_start.push_back(a);
        g_tr_stop.push_back(b);
    }
    g_tr_cap = max_launches;
    g_tr_n = 0;
    g_tr_on = true;
    return ZNG_ROCM_OK;
}

int zng_rocm_trace_end(float *ms_out, int cap) {
    if (!g_ctx) return ZNG_ROCM_ENODEV;
    g_tr_on = false;
    ZR_HIP(hipDeviceSynchronize());
    int n = g_tr_n < cap ? g_tr_n : cap;
    for (int i = 0; i < n; ++i) ZR_HIP(hipEventElapsedTime(&ms_out[i], g_tr_start[i], g_tr_stop[i]));
    return n;
}

int zng_rocm_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : g_ws) {
        Workspace *ws = kv.second;
        (void)hipFree(ws->partials);
        (void)hipFree(ws->result);
        (void)hipHostFree(ws->pinned);
        if (ws->stage) (void)hipFree(ws->stage);
        if (ws->pinned_stage) (void)hipHostFree(ws->pinned_stage);
        delete ws;
    }
    g_ws.clear();
    if (g_ctx) {
        (void)hipFree(g_ctx->tables);
        delete g_ctx;
        g_ctx = nullptr;
    }
    return ZNG_ROCM_OK;
}

}  // extern "C"
