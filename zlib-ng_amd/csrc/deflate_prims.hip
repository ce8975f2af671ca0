// deflate_prims.hip -- batched device forms of the deflate-side functable slots and of the
// inflate match-copy slot, on HBM-resident stream state (zng_rocm_deflate_view).
//
//   slide_hash          arch/generic/slide_hash_c.c:15-52
//   compare256          arch/generic/compare256_c.c:12-47
//   update_hash / quick_insert_string / insert_string
//                       insert_string.c:11-19, insert_string_tpl.h:48-104
//   longest_match       match_tpl.h:26-280 (non-SLOW)
//   chunkmemset_safe    chunkset_tpl.h:229-261 (+ CHUNKMEMSET :112-227)
//   chunksize           chunkset_tpl.h:9-11
#include "context.h"
#include "deflate_dev.h"

namespace zr {

// ---- slide_hash ---------------------------------------------------------------
// HBM-bound read-modify-write: 2 * (65536 + w_size) * 2 bytes per stream (384 KiB at w_size 32768).
// One lane handles 8 Pos entries (one dwordx4 load + store).  blockIdx.y = stream.
__device__ __forceinline__ uint32_t slide_pair(uint32_t two, uint32_t w) {
    uint32_t lo = two & 0xffffu, hi = two >> 16;
    lo = lo >= w ? lo - w : 0u;
    hi = hi >= w ? hi - w : 0u;
    return lo | (hi << 16);
}

__global__ __launch_bounds__(256)
void slide_hash_kernel(const zng_rocm_deflate_view *__restrict__ views) {
    const zng_rocm_deflate_view v = views[blockIdx.y];
    const uint32_t w = v.w_size & 0xffffu;                       // (uint16_t)s->w_size, slide_hash_c.c:48
    const uint32_t head_vec = kHashSize / 8, prev_vec = v.w_size / 8;
    const uint32_t total = head_vec + prev_vec;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        uint4 *p = i < head_vec ? reinterpret_cast<uint4 *>(v.head) + i
                                : reinterpret_cast<uint4 *>(v.prev) + (i - head_vec);
        uint4 x = *p;
        x.x = slide_pair(x.x, w);
        x.y = slide_pair(x.y, w);
        x.z = slide_pair(x.z, w);
        x.w = slide_pair(x.w, w);
        *p = x;
    }
    // w_size not a multiple of 8 (w_bits 8 gives 256, so this never triggers for valid windows)
    if (blockIdx.x == 0 && threadIdx.x < (v.w_size & 7u)) {
        uint16_t *q = v.prev + (v.w_size & ~7u) + threadIdx.x;
        uint32_t m = *q;
        *q = (uint16_t)(m >= w ? m - w : 0u);
    }
}

// ---- compare256 ------------------------------------------------------------------
__global__ __launch_bounds__(256)
void compare256_kernel(const uint8_t *__restrict__ base, const uint64_t *__restrict__ off0,
                       const uint64_t *__restrict__ off1, size_t npairs, uint32_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= npairs) return;
    const uint32_t len = compare256_wave(base + off0[wave], base + off1[wave], lane);
    if (lane == 0) out[wave] = len;
}

__global__ __launch_bounds__(256)
void update_hash_kernel(const uint32_t *__restrict__ val, size_t n, uint32_t *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = hash_calc(val[i]);
}

__global__ __launch_bounds__(256)
void update_hash_roll_kernel(const uint32_t *__restrict__ h, const uint32_t *__restrict__ val, size_t n,
                             uint32_t *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = hash_roll(h[i], val[i]);
}

// ---- insert_string family -----------------------------------------------------------
// QUICK: one position per stream, one LANE per stream (insert_string_tpl.h:58-75).
__global__ __launch_bounds__(256)
void quick_insert_kernel(const zng_rocm_deflate_view *__restrict__ views, size_t nstreams,
                         const uint32_t *__restrict__ str, uint16_t *__restrict__ head_out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nstreams) return;
    const zng_rocm_deflate_view v = views[i];
    const uint32_t pos = str[i];
    const uint32_t h = hash_calc(load_u32(v.window + pos));
    const uint16_t head = v.head[h];
    if ((uint32_t)head != pos) {
        v.prev[pos & v.w_mask] = head;
        v.head[h] = (uint16_t)pos;
    }
    head_out[i] = head;
}

// rolling variant (insert_string_roll.c): the key is carried in ins_h[i] (s->ins_h) across calls
__global__ __launch_bounds__(256)
void quick_insert_roll_kernel(const zng_rocm_deflate_view *__restrict__ views, size_t nstreams,
                              const uint32_t *__restrict__ str, uint32_t *__restrict__ ins_h,
                              uint16_t *__restrict__ head_out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nstreams) return;
    const zng_rocm_deflate_view v = views[i];
    const uint32_t pos = str[i];
    const uint32_t h = hash_roll(ins_h[i], load_u8(v.window + pos + kStdMinMatch - 1));
    ins_h[i] = h;
    const uint16_t head = v.head[h];
    if ((uint32_t)head != pos) {
        v.prev[pos & v.w_mask] = head;
        v.head[h] = (uint16_t)pos;
    }
    head_out[i] = head;
}

__global__ __launch_bounds__(256)
void insert_string_roll_kernel(const zng_rocm_deflate_view *__restrict__ views, size_t nstreams,
                               const uint32_t *__restrict__ str, const uint32_t *__restrict__ count,
                               uint32_t *__restrict__ ins_h) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= nstreams) return;
    const zng_rocm_deflate_view v = views[wave];
    const uint32_t key = insert_string_wave<true>(v.window, v.head, v.prev, v.w_mask, str[wave], count[wave], lane,
                                                  ins_h[wave]);
    if (lane == 0) ins_h[wave] = key;
}

// one wave per stream
__global__ __launch_bounds__(256)
void insert_string_kernel(const zng_rocm_deflate_view *__restrict__ views, size_t nstreams,
                          const uint32_t *__restrict__ str, const uint32_t *__restrict__ count) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= nstreams) return;
    const zng_rocm_deflate_view v = views[wave];
    insert_string_wave(v.window, v.head, v.prev, v.w_mask, str[wave], count[wave], lane);
}

// ---- longest_match -----------------------------------------------------------------
__global__ __launch_bounds__(256)
void longest_match_kernel(const zng_rocm_deflate_view *__restrict__ views, size_t nstreams,
                          const uint16_t *__restrict__ cur_match, uint32_t *__restrict__ len_out,
                          uint32_t *__restrict__ start_out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= nstreams) return;
    const zng_rocm_deflate_view v = views[wave];
    MatchParams mp;
    mp.window = v.window;
    mp.prev = v.prev;
    mp.w_size = v.w_size;
    mp.w_mask = v.w_mask;
    mp.strstart = v.strstart;
    mp.lookahead = v.lookahead;
    mp.prev_length = v.prev_length;
    mp.max_chain_length = v.max_chain_length;
    mp.good_match = v.good_match;
    mp.nice_match = (uint32_t)v.nice_match;
    mp.level = v.level;
    uint32_t ms = v.match_start;
    const uint32_t len = longest_match_wave(mp, cur_match[wave], &ms, lane);
    if (lane == 0) {
        len_out[wave] = len;
        start_out[wave] = ms;
    }
}

__global__ __launch_bounds__(256)
void longest_match_slow_kernel(const zng_rocm_deflate_view *__restrict__ views, size_t nstreams,
                               const uint16_t *__restrict__ cur_match, uint32_t *__restrict__ len_out,
                               uint32_t *__restrict__ start_out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= nstreams) return;
    const zng_rocm_deflate_view v = views[wave];
    MatchParams mp;
    mp.window = v.window;
    mp.prev = v.prev;
    mp.w_size = v.w_size;
    mp.w_mask = v.w_mask;
    mp.strstart = v.strstart;
    mp.lookahead = v.lookahead;
    mp.prev_length = v.prev_length;
    mp.max_chain_length = v.max_chain_length;
    mp.good_match = v.good_match;
    mp.nice_match = (uint32_t)v.nice_match;
    mp.level = v.level;
    uint32_t ms = v.match_start;
    // lm_init binds the rolling update_hash when max_chain_length > 1024, i.e. level 9 (deflate.c:1223-1234)
    const uint32_t len = v.max_chain_length > 1024u
                             ? longest_match_slow_wave<true>(mp, v.head, cur_match[wave], &ms, lane)
                             : longest_match_slow_wave<false>(mp, v.head, cur_match[wave], &ms, lane);
    if (lane == 0) {
        len_out[wave] = len;
        start_out[wave] = ms;
    }
}

// ---- chunkmemset_safe ----------------------------------------------------------------
// Sixteen lanes per copy (four copies per wave), sixteen bytes per lane: a round moves 256 bytes of a copy
// with one unaligned dwordx4 load and one dwordx4 store per lane, so the <= 258-byte copies of the inflate
// caller are a single round plus a two-byte tail.  Any len is accepted; rounds go in ascending order and every
// load of a round lands before any store of that round is issued, which reproduces the forward byte-serial
// semantics of the reference:
//   from < out, dist < len : byte i comes from from[i % dist]  (only original bytes are ever read)
//   otherwise              : byte i comes from from[i]; for `from` ahead of `out`, ascending rounds with
//                            load-before-store are exactly memmove.
// The ragged end of a copy (len % 16 bytes) goes one byte per lane in the copy's last round, so a 258-byte copy
// is one dwordx4 and one byte access each way per lane; a lane whose sixteen bytes are not one contiguous run of
// the source (the pattern wraps inside them) gathers them bytewise.  Nothing outside [out, out+len) is written
// and nothing outside [from, from+len) is read.
// The vector stores are non-temporal (the outputs are written once and not read back: 0.50 -> 0.63 of peak at len
// 256, 0.65 -> 0.77 at len 4096); non-temporal LOADS of the sources lose (0.63 -> 0.55, 0.77 -> 0.51).
// Measured and rejected: 2 or 4 copies in flight per 16-lane group (slower: 0.59 -> 0.53 / 0.45 of peak at len
// 256), and a byte head that aligns the dwordx4 stores (no gain; the extra byte access costs more).
typedef uint32_t u32x4_plain __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256)
void chunkmemset_kernel(uint8_t *__restrict__ base, const uint64_t *__restrict__ out_off,
                        const uint64_t *__restrict__ from_off, const uint32_t *__restrict__ len_in,
                        const uint32_t *__restrict__ left_in, size_t ncopies) {
    const uint32_t l = threadIdx.x & 15u;
    const size_t copy = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    uint8_t *out = base;
    const uint8_t *from = base;
    uint32_t len = 0;
    if (copy < ncopies) {
        out = base + out_off[copy];
        from = base + from_off[copy];
        len = len_in[copy];
        const uint32_t left = left_in[copy];
        if (len > left) len = left;                                 // chunkset_tpl.h:236
    }
    const bool behind = from < out;
    const uint64_t dist = behind ? (uint64_t)(out - from) : (uint64_t)(from - out);
    const bool pattern = behind && dist < len;
    const uint32_t period = pattern ? (uint32_t)dist : 0xffffffffu;   // source bytes repeat with this period
    uint32_t round = 0;
    bool live = len != 0u;
    while (__ballot(live) != 0ull) {
        // this round: `full` sixteen-byte lanes, then `ntail` < 16 single bytes (only in a copy's last round)
        const uint32_t remaining = live ? len - round : 0u;
        const uint32_t span = remaining < 272u ? remaining : 256u;
        const uint32_t full = span >> 4, ntail = span & 15u;
        const uint32_t i = round + 16u * l;
        u32x4_plain v = {0u, 0u, 0u, 0u};
        if (l < full) {
            uint32_t r = pattern ? i % period : i;
            if (period - r >= 16u) {
                v = load_u128(from + r);
            } else {                                                 // the pattern wraps inside these 16 bytes
#pragma unroll
                for (uint32_t j = 0; j < 16u; ++j) {
                    v[j >> 2] |= (uint32_t)load_u8(from + r) << (8u * (j & 3u));
                    r = r + 1u == period ? 0u : r + 1u;
                }
            }
        }
        const uint32_t p = round + 16u * full + l;
        uint8_t tb = 0;
        if (l < ntail) tb = load_u8(from + (pattern ? p % period : p));
        __builtin_amdgcn_s_waitcnt(0);                               // every load of the round has landed
        if (l < full) __builtin_nontemporal_store(v, (ZR_GLOBAL u32x4_unaligned *)(out + i));   // written once, streamed
        if (l < ntail) *(ZR_GLOBAL uint8_t *)(out + p) = tb;
        round += span;
        live = live && round < len;
    }
}

static int need_ctx() {
    if (!ctx()) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    return ZNG_ROCM_OK;
}

static inline unsigned waves_to_blocks(size_t waves) { return (unsigned)((waves + 3) / 4); }

}  // namespace zr

using namespace zr;

extern "C" {

int zng_rocm_slide_hash_dev(const zng_rocm_deflate_view *d_views, size_t nstreams, void *stream) {
    if (int rc = need_ctx()) return rc;
    DeviceGuard dev;
    if (!nstreams) return ZNG_ROCM_OK;
    if (!d_views || nstreams > 65535) return ZNG_ROCM_EINVAL;
    // (65536 + 32768) / 8 = 12288 vectors per stream = 48 blocks of 256 lanes
    hipLaunchKernelGGL(slide_hash_kernel, dim3(48, (unsigned)nstreams), dim3(256), 0, (hipStream_t)stream, d_views);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

int zng_rocm_compare256_dev(const uint8_t *d_base, const uint64_t *d_off0, const uint64_t *d_off1, size_t npairs,
                            uint32_t *d_len, void *stream) {
    if (int rc = need_ctx()) return rc;
    DeviceGuard dev;
    if (!npairs) return ZNG_ROCM_OK;
    if (!d_base || !d_off0 || !d_off1 || !d_len) return ZNG_ROCM_EINVAL;
    hipLaunchKernelGGL(compare256_kernel, dim3(waves_to_blocks(npairs)), dim3(256), 0, (hipStream_t)stream, d_base,
                       d_off0, d_off1, npairs, d_len);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

int zng_rocm_update_hash_dev(const uint32_t *d_val, size_t n, uint32_t *d_hash, void *stream) {
    if (int rc = need_ctx()) return rc;
    DeviceGuard dev;
    if (!n) return ZNG_ROCM_OK;
    if (!d_val || !d_hash) return ZNG_ROCM_EINVAL;
    hipLaunchKernelGGL(update_hash_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_val,
                       n, d_hash);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

int zng_rocm_update_hash_roll_dev(const uint32_t *d_h, const uint32_t *d_val, size_t n, uint32_t *d_hash, void *stream) {
    if (int rc = need_ctx()) return rc;
    DeviceGuard dev;
    if (!n) return ZNG_ROCM_OK;
    if (!d_h || !d_val || !d_hash) return ZNG_ROCM_EINVAL;
    hipLaunchKernelGGL(update_hash_roll_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       d_h, d_val, n, d_hash);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

int zng_rocm_quick_insert_string_roll_dev(const zng_rocm_deflate_view *d_views, size_t nstreams, const uint32_t *d_str,
                                          uint32_t *d_ins_h, uint16_t *d_head_out, void *stream) {
    if (int rc = need_ctx()) return rc;
    DeviceGuard dev;
    if (!nstreams) return ZNG_ROCM_OK;
    if (!d_views || !d_str || !d_ins_h || !d_head_out) return ZNG_ROCM_EINVAL;
    hipLaunchKernelGGL(quick_insert_roll_kernel, dim3((unsigned)((nstreams + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, d_views, nstreams, d_str, d_ins_h, d_head_out);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

int zng_rocm_insert_string_roll_dev(const zng_rocm_deflate_view *d_views, size_t nstreams, const uint32_t *d_str,
                                    const uint32_t *d_count, uint32_t *d_ins_h, void *stream) {
    if (int rc = need_ctx()) return rc;
    DeviceGuard dev;
    if (!nstreams) return ZNG_ROCM_OK;
    if (!d_views || !d_str || !d_count || !d_ins_h) return ZNG_ROCM_EINVAL;
    hipLaunchKernelGGL(insert_string_roll_kernel, dim3(waves_to_blocks(nstreams)), dim3(256), 0, (hipStream_t)stream,
                       d_views, nstreams, d_str, d_count, d_ins_h);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

int zng_rocm_quick_insert_string_dev(const zng_rocm_deflate_view *d_views, size_t nstreams, const uint32_t *d_str,
                                     uint16_t *d_head_out, void *stream) {
    if (int rc = need_ctx()) return rc;
    DeviceGuard dev;
    if (!nstreams) return ZNG_ROCM_OK;
    if (!d_views || !d_str || !d_head_out) return ZNG_ROCM_EINVAL;
    hipLaunchKernelGGL(quick_insert_kernel, dim3((unsigned)((nstreams + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, d_views, nstreams, d_str, d_head_out);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

int zng_rocm_insert_string_dev(const zng_rocm_deflate_view *d_views, size_t nstreams, const uint32_t *d_str,
                               const uint32_t *d_count, void *stream) {
    if (int rc = need_ctx()) return rc;
    DeviceGuard dev;
    if (!nstreams) return ZNG_ROCM_OK;
    if (!d_views || !d_str || !d_count) return ZNG_ROCM_EINVAL;
    hipLaunchKernelGGL(insert_string_kernel, dim3(waves_to_blocks(nstreams)), dim3(256), 0, (hipStream_t)stream,
                       d_views, nstreams, d_str, d_count);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

int zng_rocm_longest_match_dev(const zng_rocm_deflate_view *d_views, size_t nstreams, const uint16_t *d_cur_match,
                               uint32_t *d_len_out, uint32_t *d_match_start_out, void *stream) {
    if (int rc = need_ctx()) return rc;
    DeviceGuard dev;
    if (!nstreams) return ZNG_ROCM_OK;
    if (!d_views || !d_cur_match || !d_len_out || !d_match_start_out) return ZNG_ROCM_EINVAL;
    hipLaunchKernelGGL(longest_match_kernel, dim3(waves_to_blocks(nstreams)), dim3(256), 0, (hipStream_t)stream,
                       d_views, nstreams, d_cur_match, d_len_out, d_match_start_out);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

int zng_rocm_longest_match_slow_dev(const zng_rocm_deflate_view *d_views, size_t nstreams,
                                    const uint16_t *d_cur_match, uint32_t *d_len_out, uint32_t *d_match_start_out,
                                    void *stream) {
    if (int rc = need_ctx()) return rc;
    DeviceGuard dev;
    if (!nstreams) return ZNG_ROCM_OK;
    if (!d_views || !d_cur_match || !d_len_out || !d_match_start_out) return ZNG_ROCM_EINVAL;
    hipLaunchKernelGGL(longest_match_slow_kernel, dim3(waves_to_blocks(nstreams)), dim3(256), 0, (hipStream_t)stream,
                       d_views, nstreams, d_cur_match, d_len_out, d_match_start_out);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

int zng_rocm_chunkmemset_safe_dev(uint8_t *d_base, const uint64_t *d_out_off, const uint64_t *d_from_off,
                                  const uint32_t *d_len, const uint32_t *d_left, size_t ncopies, void *stream) {
    if (int rc = need_ctx()) return rc;
    DeviceGuard dev;
    if (!ncopies) return ZNG_ROCM_OK;
    if (!d_base || !d_out_off || !d_from_off || !d_len || !d_left) return ZNG_ROCM_EINVAL;
    hipLaunchKernelGGL(chunkmemset_kernel, dim3((unsigned)((ncopies + 15) / 16)), dim3(256), 0, (hipStream_t)stream, d_base,
                       d_out_off, d_from_off, d_len, d_left, ncopies);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

uint32_t zng_rocm_chunksize(void) { return 16; }

}  // extern "C"
