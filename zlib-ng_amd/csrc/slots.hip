// slots.hip -- functable slots with the reference's host-pointer signatures, and the
// combine operators.
//
//   zng_rocm_adler32 / _crc32 / _adler32_fold_copy / crc32_fold quartet:
//       stage the caller's bytes into HBM, run the streaming kernels of checksum.hip,
//       return the value.  Argument meaning and corner cases follow
//       arch/generic/adler32_c.c:11-54, arch/generic/crc32_braid_c.c:62-216,
//       arch/generic/adler32_fold_c.c:11-15, arch/generic/crc32_fold_c.c:10-31.
//   zng_rocm_*_combine_dev: adler32.c:32-54 / crc32_braid_comb.c:16-24 evaluated on device
//       over an array of {check, len} blocks (order preserving; the CRC operator is not
//       commutative).
#include "context.h"

#include <string.h>

#include <mutex>

namespace zr {

// ---- the calling thread's HIP stream -------------------------------------------
// Per host thread, so independent zlib-ng streams used from different threads do not serialise (zlib-ng.h.in:157-159).
// The holder releases the stream and its workspace when the thread ends, and re-creates it when the context it was
// made under is gone (zng_rocm_shutdown followed by a new zng_rocm_init, possibly on another device).
struct SlotStream {
    hipStream_t stream = nullptr;
    uint64_t    generation = 0;
    ~SlotStream() { drop(); }
    void drop() {
        Context *c = ctx();
        if (stream && c && c->generation == generation) {
            (void)zng_rocm_stream_release(stream);
            DeviceGuard dev;
            (void)hipStreamDestroy(stream);
        }
        stream = nullptr;            // a stream of a context that was shut down died with its device state
    }
};
static thread_local SlotStream t_slot;

static int slot_stream(hipStream_t *out) {
    Context *c = ctx();
    if (!c) {
        // a functable slot may be the first thing a process calls (functable.c:269-360 lazy init)
        int rc = zng_rocm_init(-1);
        if (rc != ZNG_ROCM_OK) return rc;
        c = ctx();
    }
    if (!t_slot.stream || t_slot.generation != c->generation) {
        t_slot.drop();
        DeviceGuard dev;
        ZR_HIP(hipStreamCreateWithFlags(&t_slot.stream, hipStreamNonBlocking));
        t_slot.generation = c->generation;
    }
    *out = t_slot.stream;
    return ZNG_ROCM_OK;
}

// Host bytes -> bounded device staging -> kernel -> two result words.
// The message goes through ONE device chunk of at most kStageChunk bytes, however long it is: copy a chunk, run the
// streaming pass over it with the seed taken ON THE DEVICE from the previous chunk's result (chunks of one call are
// ordered on the thread's stream, so the seed chains without a host round trip), copy the next.  The pageable copy
// is staged by the HIP runtime through its own pinned buffers; the kernel of a 16 MiB chunk takes ~10 us against
// ~300 us of PCIe time, so nothing is gained by a second chunk in flight.  One synchronisation per call.
constexpr size_t kStageChunk = 16u << 20;

static int run_host(bool do_adler, bool do_crc, uint32_t adler, uint32_t crc, const uint8_t *src, uint8_t *dst,
                    size_t len, uint32_t out[2]) {
    hipStream_t st;
    if (int rc = slot_stream(&st)) return rc;
    DeviceGuard dev;
    Workspace *ws = workspace_for(st);
    if (!ws) return ZNG_ROCM_ENOMEM;
    std::lock_guard<std::mutex> use(ws->mu);
    const size_t chunk = len < kStageChunk ? len : kStageChunk;
    const size_t half = (chunk + 63) & ~(size_t)15;                  // src chunk | dst chunk, same 16-byte phase
    if (int rc = ensure_stage(ws, (dst ? 2 : 1) * half + 64)) return rc;
    uint8_t *d_src = ws->stage;
    uint8_t *d_dst = dst ? ws->stage + half + 16 : nullptr;
    uint32_t *res = ws->result;                                      // [0..1] result, [2..3] the chained seed
    size_t done = 0;
    bool first = true;
    do {
        const size_t m = len - done < kStageChunk ? len - done : kStageChunk;
        if (m) ZR_HIP(hipMemcpyAsync(d_src, src + done, m, hipMemcpyHostToDevice, st));
        if (!first) ZR_HIP(hipMemcpyAsync(res + 2, res, 8, hipMemcpyDeviceToDevice, st));
        if (int rc = launch_checksum(do_adler, do_crc, adler, crc, d_src, m ? d_dst : nullptr, m, res, res + 1, st,
                                     first ? nullptr : res + 2, first ? nullptr : res + 3))
            return rc;
        if (dst && m) ZR_HIP(hipMemcpyAsync(dst + done, d_dst, m, hipMemcpyDeviceToHost, st));
        done += m;
        first = false;
    } while (done < len);
    ZR_HIP(hipMemcpyAsync(ws->pinned, res, 8, hipMemcpyDeviceToHost, st));
    ZR_HIP(hipStreamSynchronize(st));
    out[0] = ws->pinned[0];
    out[1] = ws->pinned[1];
    return ZNG_ROCM_OK;
}

static void run_host_or_die(bool do_adler, bool do_crc, uint32_t adler, uint32_t crc, const uint8_t *src, uint8_t *dst,
                            size_t len, uint32_t out[2], const char *slot) {
    if (run_host(do_adler, do_crc, adler, crc, src, dst, len, out) != ZNG_ROCM_OK) die(slot);
}

// ---- combine kernels ---------------------------------------------------------
// One workgroup; tiles of 1024 blocks walked from the END of the array so that
// "bytes after block i" is a running suffix sum.
template <bool IS_CRC>
__global__ __launch_bounds__(1024)
void combine_kernel(const uint32_t *__restrict__ checks, const uint64_t *__restrict__ lens, long long count,
                    long long check_stride, long long len_stride,      // in elements: 1, 1 for plain arrays
                    const DeviceTables *__restrict__ tabs, uint32_t *__restrict__ out) {
    __shared__ unsigned long long scan[1024];
    __shared__ unsigned long long red_a[16], red_b[16];
    __shared__ uint32_t red_c[16];
    __shared__ unsigned long long carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();

    uint32_t c = 0;
    unsigned long long a = 0, b = 0;
    for (long long hi = count; hi > 0; hi -= 1024) {
        const long long lo = hi > 1024 ? hi - 1024 : 0;
        const int m = (int)(hi - lo);
        // reversed order inside the tile: slot j holds block hi-1-j, inclusive prefix = suffix sum
        const long long idx = hi - 1 - t;
        unsigned long long len = (t < m) ? lens[idx * len_stride] : 0;
        scan[t] = len;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            unsigned long long v = (t >= d) ? scan[t - d] : 0;
            __syncthreads();
            scan[t] += v;
            __syncthreads();
        }
        const unsigned long long base = carry;
        const unsigned long long after = base + scan[t] - len;
        if (t < m) {
            uint32_t chk = checks[idx * check_stride];
            if (IS_CRC) {
                c ^= mulmod(chk, xpow_bytes(tabs->pow_tab, after));
            } else {
                // linear form of the block: A = s1 - 1, B = s2 - len  (seed 1 implied), SURVEY.md 9.2
                unsigned long long s1 = chk & 0xffffu, s2 = chk >> 16;
                unsigned long long A = (s1 + kAdlerBase - 1) % kAdlerBase;
                unsigned long long B = (s2 + kAdlerBase - len % kAdlerBase) % kAdlerBase;
                a += A;
                b = (b + B + A * (after % kAdlerBase)) % kAdlerBase;
            }
        }
        __syncthreads();
        if (t == 1023) carry = base + scan[1023];
        __syncthreads();
    }
#pragma unroll
    for (int mm = 32; mm >= 1; mm >>= 1) {
        c ^= __shfl_xor(c, mm, 64);
        a += __shfl_xor(a, mm, 64);
        b += __shfl_xor(b, mm, 64);
    }
    if ((t & 63) == 0) {
        red_c[t >> 6] = c;
        red_a[t >> 6] = a;
        red_b[t >> 6] = b;
    }
    __syncthreads();
    if (t == 0) {
        uint32_t cc = 0;
        unsigned long long A = 0, B = 0;
        for (int w = 0; w < 16; ++w) {
            cc ^= red_c[w];
            A += red_a[w] % kAdlerBase;
            B += red_b[w] % kAdlerBase;
        }
        if (IS_CRC) {
            *out = cc;
        } else {
            const unsigned long long total = carry % kAdlerBase;
            unsigned long long r1 = (1 + A) % kAdlerBase;
            unsigned long long r2 = (total + B) % kAdlerBase;
            *out = (uint32_t)(r1 | (r2 << 16));
        }
    }
}

template <bool IS_CRC>
static int launch_combine(const uint32_t *d_checks, const uint64_t *d_lens, size_t count, uint32_t *d_out,
                          hipStream_t stream, long long check_stride = 1, long long len_stride = 1) {
    Context *c = ctx();
    if (!c) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    if (!d_out || (count && (!d_checks || !d_lens))) return ZNG_ROCM_EINVAL;
    DeviceGuard dev;
    hipLaunchKernelGGL((combine_kernel<IS_CRC>), dim3(1), dim3(1024), 0, stream, d_checks, d_lens, (long long)count,
                       check_stride, len_stride, c->tables, d_out);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

// host-side polynomial helpers for the scalar combine API
static uint32_t host_x2n_table[32];          // x2n_table of crc32_braid_tbl.h:9437-9444, built as tools/makecrct.c:75-79 does
static std::once_flag host_x2n_once;
static uint32_t host_x2nmodp(int64_t n, unsigned k) {
    std::call_once(host_x2n_once, [] {
        uint32_t p = 0x40000000u;
        host_x2n_table[0] = p;
        for (int i = 1; i < 32; ++i) host_x2n_table[i] = p = mulmod(p, p);
    });
    uint32_t p = 0x80000000u;
    while (n) {
        if (n & 1) p = mulmod(host_x2n_table[k & 31], p);
        n >>= 1;
        k++;
    }
    return p;
}

}  // namespace zr

using namespace zr;

extern "C" {

uint32_t zng_rocm_adler32(uint32_t adler, const uint8_t *buf, size_t len) {
    if (buf == nullptr) return 1u;                       // adler32_c.c:24-25
    uint32_t out[2];
    run_host_or_die(true, false, adler, 0, buf, nullptr, len, out, "zng_rocm_adler32");
    return out[0];
}

uint32_t zng_rocm_adler32_fold_copy(uint32_t adler, uint8_t *dst, const uint8_t *src, size_t len) {
    uint32_t out[2];
    run_host_or_die(true, false, adler, 0, src, dst, len, out, "zng_rocm_adler32_fold_copy");
    return out[0];
}

uint32_t zng_rocm_crc32(uint32_t crc, const uint8_t *buf, size_t len) {
    if (buf == nullptr) return 0u;                       // export layer, crc32.c:22,28
    uint32_t out[2];
    run_host_or_die(false, true, 0, crc, buf, nullptr, len, out, "zng_rocm_crc32");
    return out[1];
}

uint32_t zng_rocm_crc32_fold_reset(zng_rocm_crc32_fold_t *crc) {
    crc->value = 0;                                      // crc32_fold_c.c:10-13
    return crc->value;
}

void zng_rocm_crc32_fold(zng_rocm_crc32_fold_t *crc, const uint8_t *src, size_t len, uint32_t init_crc) {
    (void)init_crc;                                      // crc32_fold_c.c:20-27: ignored by the generic form
    uint32_t out[2];
    run_host_or_die(false, true, 0, crc->value, src, nullptr, len, out, "zng_rocm_crc32_fold");
    crc->value = out[1];
}

void zng_rocm_crc32_fold_copy(zng_rocm_crc32_fold_t *crc, uint8_t *dst, const uint8_t *src, size_t len) {
    uint32_t out[2];
    run_host_or_die(false, true, 0, crc->value, src, dst, len, out, "zng_rocm_crc32_fold_copy");
    crc->value = out[1];
}

uint32_t zng_rocm_crc32_fold_final(zng_rocm_crc32_fold_t *crc) { return crc->value; }

// ---- the same slots with an error channel: 0 and the value through *out, or a ZNG_ROCM_E* code and nothing
// written (SURVEY.md 8b "any HIP failure must degrade to the CPU implementation, never surface": the reference-side
// adapter calls these and falls back to the CPU tier it remembered) --------------------
int zng_rocm_adler32_try(uint32_t adler, const uint8_t *buf, size_t len, uint32_t *out) {
    if (!out) return ZNG_ROCM_EINVAL;
    if (buf == nullptr) { *out = 1u; return ZNG_ROCM_OK; }
    uint32_t r[2];
    if (int rc = run_host(true, false, adler, 0, buf, nullptr, len, r)) return rc;
    *out = r[0];
    return ZNG_ROCM_OK;
}

int zng_rocm_adler32_fold_copy_try(uint32_t adler, uint8_t *dst, const uint8_t *src, size_t len, uint32_t *out) {
    if (!out || (len && (!dst || !src))) return ZNG_ROCM_EINVAL;
    uint32_t r[2];
    if (int rc = run_host(true, false, adler, 0, src, dst, len, r)) return rc;
    *out = r[0];
    return ZNG_ROCM_OK;
}

int zng_rocm_crc32_try(uint32_t crc, const uint8_t *buf, size_t len, uint32_t *out) {
    if (!out) return ZNG_ROCM_EINVAL;
    if (buf == nullptr) { *out = 0u; return ZNG_ROCM_OK; }
    uint32_t r[2];
    if (int rc = run_host(false, true, 0, crc, buf, nullptr, len, r)) return rc;
    *out = r[1];
    return ZNG_ROCM_OK;
}

int zng_rocm_crc32_fold_try(zng_rocm_crc32_fold_t *crc, const uint8_t *src, size_t len, uint32_t init_crc) {
    (void)init_crc;
    if (!crc || (len && !src)) return ZNG_ROCM_EINVAL;
    uint32_t r[2];
    if (int rc = run_host(false, true, 0, crc->value, src, nullptr, len, r)) return rc;
    crc->value = r[1];                                   // the state only moves on success
    return ZNG_ROCM_OK;
}

int zng_rocm_crc32_fold_copy_try(zng_rocm_crc32_fold_t *crc, uint8_t *dst, const uint8_t *src, size_t len) {
    if (!crc || (len && (!dst || !src))) return ZNG_ROCM_EINVAL;
    uint32_t r[2];
    if (int rc = run_host(false, true, 0, crc->value, src, dst, len, r)) return rc;
    crc->value = r[1];
    return ZNG_ROCM_OK;
}

int zng_rocm_adler32_combine_dev(const uint32_t *d_checks, const uint64_t *d_lens, size_t count, uint32_t *d_out,
                                 void *stream) {
    return launch_combine<false>(d_checks, d_lens, count, d_out, (hipStream_t)stream);
}

int zng_rocm_crc32_combine_dev(const uint32_t *d_checks, const uint64_t *d_lens, size_t count, uint32_t *d_out,
                               void *stream) {
    return launch_combine<true>(d_checks, d_lens, count, d_out, (hipStream_t)stream);
}

int zng_rocm_combine_rows_dev(const zng_rocm_check_row *d_rows, size_t count, uint32_t *d_out2, void *stream) {
    if (!d_out2 || (count && !d_rows)) return ZNG_ROCM_EINVAL;
    static_assert(sizeof(zng_rocm_check_row) == 16, "row layout is part of the ABI");
    const uint32_t *first = reinterpret_cast<const uint32_t *>(d_rows);
    const uint64_t *lens = reinterpret_cast<const uint64_t *>(d_rows) + 1;
    if (int rc = launch_combine<false>(first, lens, count, d_out2, (hipStream_t)stream, 4, 2)) return rc;
    return launch_combine<true>(first + 1, lens, count, d_out2 + 1, (hipStream_t)stream, 4, 2);
}

uint32_t zng_rocm_adler32_combine(uint32_t adler1, uint32_t adler2, int64_t len2) {
    if (len2 < 0) return 0xffffffffu;                    // adler32.c:37-39
    // adler32.c:42-53, including its bounded subtractions (not a plain modulo)
    const uint32_t rem = (uint32_t)(len2 % kAdlerBase);
    const uint32_t lo1 = adler1 & 0xffffu;
    uint32_t sum2 = (rem * lo1) % kAdlerBase;
    uint32_t sum1 = lo1 + (adler2 & 0xffffu) + kAdlerBase - 1;
    sum2 += ((adler1 >> 16) & 0xffffu) + ((adler2 >> 16) & 0xffffu) + kAdlerBase - rem;
    if (sum1 >= kAdlerBase) sum1 -= kAdlerBase;
    if (sum1 >= kAdlerBase) sum1 -= kAdlerBase;
    if (sum2 >= (kAdlerBase << 1)) sum2 -= (kAdlerBase << 1);
    if (sum2 >= kAdlerBase) sum2 -= kAdlerBase;
    return sum1 | (sum2 << 16);
}

uint32_t zng_rocm_crc32_combine(uint32_t crc1, uint32_t crc2, int64_t len2) {
    return mulmod(host_x2nmodp(len2, 3), crc1) ^ crc2;   // crc32_braid_comb.c:16-18
}

uint32_t zng_rocm_crc32_combine_gen(int64_t len2) { return host_x2nmodp(len2, 3); }

uint32_t zng_rocm_crc32_combine_op(uint32_t crc1, uint32_t crc2, uint32_t op) { return mulmod(op, crc1) ^ crc2; }

}  // extern "C"
