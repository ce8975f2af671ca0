// hook.hip -- the device side of the COARSE boundary: what an arch/rocm DEFLATE_HOOK / INFLATE_TYPEDO_HOOK backend holds
// in its arch_deflate_state / arch_inflate_state (deflate.h:319-321, inflate.h:160-162) and calls per deflate() /
// inflate() (deflate.c:1039-1083, inflate.c:728).  The model is IBM Z DFLTCC (arch/s390/dfltcc_deflate.c,
// dfltcc_inflate.c): the history lives with the accelerator (here: 32 KiB in front of the staging buffer in HBM), the
// host code keeps zng_stream's next_in / next_out bookkeeping and falls back to software when the accelerator says no.
// Host pointers in, host pointers out: staging is this file's business (bounded: one block).
#include "context.h"
#include "deflate_dev.h"

#include <string.h>

struct zng_rocm_hook {
    uint64_t     generation;          // of the context it was created under
    hipStream_t  st;
    uint8_t     *d_in;                // [32768 history][block]
    size_t       in_cap;              // bytes of block room
    uint32_t     hist_len;            // history is d_in[32768 - hist_len, 32768)
    uint8_t     *d_out;               // compressed block / plaintext
    size_t       out_cap;
    uint8_t     *h_out;               // pinned mirror of d_out (what the inflate side hands back)
    size_t       h_cap;
    uint32_t    *d_check;             // 2 words
    uint32_t    *h_check;             // pinned
};

namespace zr {
int inflate_large_device_only(const uint8_t *d_src, size_t src_len, const uint8_t *d_window, uint32_t window_len, uint8_t *d_dst,
                              size_t dst_cap, uint64_t *out_len, size_t *in_used, hipStream_t st);      // inflate_large.hip
}

namespace {
constexpr uint32_t kHist = 32768u;

int grow_device(uint8_t **p, size_t *cap, size_t want, size_t front) {
    if (*cap >= want && *p) return ZNG_ROCM_OK;
    uint8_t *n = nullptr;
    const size_t c = want + (want >> 2) + 4096;
    if (hipMalloc(&n, front + c + 64) != hipSuccess) {
        zr::set_error("device allocation of %zu bytes failed", front + c);
        return ZNG_ROCM_ENOMEM;
    }
    if (*p && front) (void)hipMemcpy(n, *p, front, hipMemcpyDeviceToDevice);      // the history survives a growth
    if (*p) (void)hipFree(*p);
    *p = n;
    *cap = c;
    return ZNG_ROCM_OK;
}

int grow_pinned(uint8_t **p, size_t *cap, size_t want) {
    if (*cap >= want && *p) return ZNG_ROCM_OK;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    const size_t c = want + (want >> 2) + 4096;
    if (hipHostMalloc((void **)p, c, hipHostMallocDefault) != hipSuccess) {
        zr::set_error("pinned allocation of %zu bytes failed", c);
        *cap = 0;
        return ZNG_ROCM_ENOMEM;
    }
    *cap = c;
    return ZNG_ROCM_OK;
}

// history <- the last min(32768, hist + n) bytes of [history][n bytes at d_in + 32768]; through d_out (the two may overlap)
int roll_history(zng_rocm_hook *h, size_t n) {
    const size_t have = (size_t)h->hist_len + n;
    const uint32_t keep = (uint32_t)(have < kHist ? have : kHist);
    if (n == 0) return ZNG_ROCM_OK;
    if (int rc = grow_device(&h->d_out, &h->out_cap, kHist, 0)) return rc;
    const uint8_t *src = h->d_in + kHist + n - keep;
    if (hipMemcpyAsync(h->d_out, src, keep, hipMemcpyDeviceToDevice, h->st) != hipSuccess ||
        hipMemcpyAsync(h->d_in + kHist - keep, h->d_out, keep, hipMemcpyDeviceToDevice, h->st) != hipSuccess) {
        zr::set_error("history copy failed");
        return ZNG_ROCM_EHIP;
    }
    h->hist_len = keep;
    return ZNG_ROCM_OK;
}

bool stale(const zng_rocm_hook *h) { return !zr::ctx() || !h || h->generation != zr::ctx()->generation; }

constexpr size_t kHookLargeMember = 4u << 20;         // compressed bytes from which a member is decoded on the device

// 1 = stream end (outputs set), 0 = leave it to the host decoder, negative = error
int hook_inflate_on_device(zng_rocm_hook *h, const uint8_t *in, size_t in_len, int check, uint32_t *check_value,
                           const uint8_t **out, size_t *out_len, size_t *in_used) {
    if (grow_device(&h->d_out, &h->out_cap, in_len, 0) != ZNG_ROCM_OK) return 0;          // the compressed bytes
    if (hipMemcpyAsync(h->d_out, in, in_len, hipMemcpyHostToDevice, h->st) != hipSuccess) return 0;
    size_t room = h->in_cap > 4 * in_len ? h->in_cap : 4 * in_len;                    // a guess; the call says what it needs
    uint64_t n = 0;
    size_t used = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (grow_device(&h->d_in, &h->in_cap, room, kHist) != ZNG_ROCM_OK) return 0;
        const int rc = zr::inflate_large_device_only(h->d_out, in_len, h->hist_len ? h->d_in + kHist - h->hist_len : nullptr,
                                                     h->hist_len, h->d_in + kHist, h->in_cap, &n, &used, h->st);
        if (rc == 1) break;
        if (rc == -5 && attempt == 0 && n > h->in_cap) {
            room = (size_t)n;
            continue;
        }
        return 0;
    }
    if (grow_pinned(&h->h_out, &h->h_cap, (size_t)n) != ZNG_ROCM_OK) return ZNG_ROCM_ENOMEM;
    if (check && n) {
        const int rc = check == 1 ? zng_rocm_adler32_dev(*check_value, h->d_in + kHist, (size_t)n, h->d_check, h->st)
                                  : zng_rocm_crc32_dev(*check_value, h->d_in + kHist, (size_t)n, h->d_check, h->st);
        if (rc != ZNG_ROCM_OK) return rc;
        ZR_HIP(hipMemcpyAsync(h->h_check, h->d_check, 4, hipMemcpyDeviceToHost, h->st));
    }
    if (n) ZR_HIP(hipMemcpyAsync(h->h_out, h->d_in + kHist, (size_t)n, hipMemcpyDeviceToHost, h->st));
    if (int r2 = roll_history(h, (size_t)n)) return r2;
    ZR_HIP(hipStreamSynchronize(h->st));
    if (check && n) *check_value = h->h_check[0];
    *out = h->h_out;
    *out_len = (size_t)n;
    *in_used = used;
    return 1;
}
}  // namespace

using namespace zr;

extern "C" {

int zng_rocm_hook_create(zng_rocm_hook **out, size_t block_bytes) {
    if (!out) return ZNG_ROCM_EINVAL;
    *out = nullptr;
    if (!ctx()) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    DeviceGuard dev;
    zng_rocm_hook *h = new (std::nothrow) zng_rocm_hook();
    if (!h) return ZNG_ROCM_ENOMEM;
    memset(h, 0, sizeof *h);
    h->generation = ctx()->generation;
    if (hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&h->d_check, 16) != hipSuccess ||
        hipHostMalloc((void **)&h->h_check, 16, hipHostMallocDefault) != hipSuccess ||
        grow_device(&h->d_in, &h->in_cap, block_bytes ? block_bytes : (1u << 20), kHist) != ZNG_ROCM_OK) {
        set_error("hook context: stream / buffer creation failed");
        zng_rocm_hook_destroy(h);
        return ZNG_ROCM_EHIP;
    }
    *out = h;
    return ZNG_ROCM_OK;
}

void zng_rocm_hook_destroy(zng_rocm_hook *h) {
    if (!h) return;
    if (!stale(h)) {                           // after a shutdown the device objects are gone with their context
        DeviceGuard dev;
        if (h->st) {
            (void)hipStreamSynchronize(h->st);
            (void)zng_rocm_stream_release(h->st);
            (void)hipStreamDestroy(h->st);
        }
        if (h->d_in) (void)hipFree(h->d_in);
        if (h->d_out) (void)hipFree(h->d_out);
        if (h->d_check) (void)hipFree(h->d_check);
        if (h->h_out) (void)hipHostFree(h->h_out);
        if (h->h_check) (void)hipHostFree(h->h_check);
    }
    delete h;
}

int zng_rocm_hook_reset(zng_rocm_hook *h) {
    if (stale(h)) return ZNG_ROCM_ENODEV;
    h->hist_len = 0;
    return ZNG_ROCM_OK;
}

int zng_rocm_hook_set_history(zng_rocm_hook *h, const uint8_t *dict, uint32_t len) {
    if (stale(h)) return ZNG_ROCM_ENODEV;
    if (len && !dict) return ZNG_ROCM_EINVAL;
    DeviceGuard dev;
    const uint32_t keep = len < kHist ? len : kHist;               // the last 32 KiB count (deflate.c:479-484)
    if (keep) {
        ZR_HIP(hipMemcpyAsync(h->d_in + kHist - keep, dict + (len - keep), keep, hipMemcpyHostToDevice, h->st));
        ZR_HIP(hipStreamSynchronize(h->st));
    }
    h->hist_len = keep;
    return ZNG_ROCM_OK;
}

int zng_rocm_hook_get_history(zng_rocm_hook *h, uint8_t *dict, uint32_t *len) {
    if (stale(h)) return ZNG_ROCM_ENODEV;
    DeviceGuard dev;
    if (dict && h->hist_len) {
        ZR_HIP(hipMemcpyAsync(dict, h->d_in + kHist - h->hist_len, h->hist_len, hipMemcpyDeviceToHost, h->st));
        ZR_HIP(hipStreamSynchronize(h->st));
    }
    if (len) *len = h->hist_len;
    return ZNG_ROCM_OK;
}

size_t zng_rocm_hook_deflate_bound(size_t in_len) { return zng_rocm_deflate_bound(in_len); }

int zng_rocm_hook_deflate_block(zng_rocm_hook *h, int level, const uint8_t *in, size_t in_len, uint32_t flags, int check,
                                uint32_t *check_value, uint8_t *out, size_t out_cap, size_t *out_len) {
    if (stale(h)) return ZNG_ROCM_ENODEV;
    if (!out || !out_len || (in_len && !in) || (check && !check_value) || check < 0 || check > 2) return ZNG_ROCM_EINVAL;
    DeviceGuard dev;
    if (int rc = grow_device(&h->d_in, &h->in_cap, in_len, kHist)) return rc;
    const size_t bound = zng_rocm_deflate_bound(in_len);
    if (int rc = grow_device(&h->d_out, &h->out_cap, bound, 0)) return rc;
    if (in_len) ZR_HIP(hipMemcpyAsync(h->d_in + kHist, in, in_len, hipMemcpyHostToDevice, h->st));
    if (check && in_len) {
        const int rc = check == 1 ? zng_rocm_adler32_dev(*check_value, h->d_in + kHist, in_len, h->d_check, h->st)
                                  : zng_rocm_crc32_dev(*check_value, h->d_in + kHist, in_len, h->d_check, h->st);
        if (rc != ZNG_ROCM_OK) return rc;
        ZR_HIP(hipMemcpyAsync(h->h_check, h->d_check, 4, hipMemcpyDeviceToHost, h->st));
    }
    size_t clen = 0;
    if (int rc = zng_rocm_deflate_block_dev(level, h->d_in + kHist, in_len, h->hist_len, flags, h->d_out, h->out_cap, &clen, h->st))
        return rc;
    if (clen > out_cap) {
        set_error("compressed block (%zu bytes) exceeds out_cap", clen);
        return -5;
    }
    ZR_HIP(hipMemcpyAsync(out, h->d_out, clen, hipMemcpyDeviceToHost, h->st));
    if (int rc = roll_history(h, in_len)) return rc;
    ZR_HIP(hipStreamSynchronize(h->st));
    if (check && in_len) *check_value = h->h_check[0];
    *out_len = clen;
    return ZNG_ROCM_OK;
}

int zng_rocm_hook_inflate(zng_rocm_hook *h, const uint8_t *in, size_t in_len, int check, uint32_t *check_value,
                          const uint8_t **out, size_t *out_len, size_t *in_used, const char **msg) {
    if (stale(h)) return ZNG_ROCM_ENODEV;
    if (!out || !out_len || !in_used || (in_len && !in) || (check && !check_value) || check < 0 || check > 2) return ZNG_ROCM_EINVAL;
    DeviceGuard dev;
    *out = nullptr;
    *out_len = 0;
    *in_used = 0;
    if (msg) *msg = nullptr;
    // A large member goes to the device whole (inflate_large.hip: block starts found there, one wavefront per part): the
    // host decoder below manages ~0.8 GB/s of output on its one thread.  Anything but a clean end of stream -- truncated
    // input, a data error, output larger than the estimate twice over -- is left to that decoder, whose status, message
    // and byte counts are the reference's.
    if (in_len >= kHookLargeMember) {
        const int rc = hook_inflate_on_device(h, in, in_len, check, check_value, out, out_len, in_used);
        if (rc != 0) return rc;
    }
    zng_rocm_inflate_tokens tk;
    const int status = zng_rocm_inflate_tokens_decode_window(in, in_len, h->hist_len, &tk);
    if (status != 1) {                                   // not a complete stream: nothing is consumed, nothing produced
        if (msg) *msg = tk.msg;
        zng_rocm_inflate_tokens_free(&tk);
        return status == -4 ? ZNG_ROCM_ENOMEM : status;
    }
    const size_t n = (size_t)tk.out_len;
    int rc = grow_device(&h->d_in, &h->in_cap, n, kHist);               // the plaintext lands behind the history
    if (rc == ZNG_ROCM_OK) rc = grow_pinned(&h->h_out, &h->h_cap, n);
    if (rc == ZNG_ROCM_OK)
        rc = inflate_tokens_to_device(&tk, h->hist_len ? h->d_in + kHist - h->hist_len : nullptr, h->hist_len, h->d_in + kHist, h->st);
    *in_used = tk.in_used;
    zng_rocm_inflate_tokens_free(&tk);
    if (rc != ZNG_ROCM_OK) return rc;
    if (check && n) {
        rc = check == 1 ? zng_rocm_adler32_dev(*check_value, h->d_in + kHist, n, h->d_check, h->st)
                        : zng_rocm_crc32_dev(*check_value, h->d_in + kHist, n, h->d_check, h->st);
        if (rc != ZNG_ROCM_OK) return rc;
        ZR_HIP(hipMemcpyAsync(h->h_check, h->d_check, 4, hipMemcpyDeviceToHost, h->st));
    }
    if (n) ZR_HIP(hipMemcpyAsync(h->h_out, h->d_in + kHist, n, hipMemcpyDeviceToHost, h->st));
    if (int r2 = roll_history(h, n)) return r2;
    ZR_HIP(hipStreamSynchronize(h->st));
    if (check && n) *check_value = h->h_check[0];
    *out = h->h_out;
    *out_len = n;
    return 1;                                            // Z_STREAM_END
}

}  // extern "C"
