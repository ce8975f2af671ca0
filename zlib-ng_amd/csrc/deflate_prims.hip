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

#include <mutex>

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
// chunkmemset_safe as a batch, two kernels on the caller's stream.
//
// chunkmemset_packed_kernel: one wavefront takes 48 consecutive copies and PACKS their 16-byte pieces over its lanes --
// piece j of copy k sits in slot start_k + j, slot s is lane s % 64 of round s / 64 -- so every lane moves 16 bytes in
// every round whatever the mix of lengths.  (Round 1 gave each copy a fixed group of 16 lanes: on the inflate caller's
// mix, lengths 3..258, half of them idled and a wave had 0.5 KiB in flight: 0.38 of peak.)  The descriptors are read
// once, one copy per lane (coalesced); the slot -> copy map is a binary search over the running piece counts in LDS.
// A wave loads eight rounds (8 KiB; 48 copies of <= 258 bytes are 418 slots on average) before it issues the first
// store: the kernel is bound by memory latency, and bytes in flight per wave is what buys throughput.
//   from < out, dist < len : byte i comes from from[i % dist]  (only original bytes are ever read)
//   otherwise              : byte i comes from from[i]
// A copy's ragged end is one more 16-byte piece laid over its last 16 bytes (it rewrites up to 15 bytes with the
// same values); copies shorter than 16 bytes go as 8 + 4 + 2 + 1 byte accesses, pieces inside which the pattern wraps
// bytewise.  Two kinds of copy stay out of the packing and go on a work list in device memory:
//   * copies longer than kLongCopy (never the inflate caller: its matches end at 258 bytes) -- packed, one wave would
//     move 48 of them by itself;
//   * copies whose source lies AHEAD of and overlaps the destination: they need memmove order.
// chunkmemset_list_kernel (a grid the size of the machine, 16 lanes per copy, rounds of 256 bytes in ascending order
// with every load of a round landed before its stores) takes them off that list; it finds the list empty and leaves
// at once for the inflate caller's batches.
// Nothing outside [out, out+len) is written and nothing outside [from, from+len) is read.  The vector stores are
// non-temporal (the outputs are written once and not read back).
typedef uint32_t u32x4_plain __attribute__((ext_vector_type(4)));

struct CopyDesc {                       // LDS, one per copy of the wave
    uint64_t out, from;                 // byte offsets from `base`
    uint32_t len, start;                // bytes; first slot
    uint32_t period;                    // pattern period (dist) or 0xffffffff
    uint32_t pad;
};

constexpr int kCopyRounds = 8;          // rounds a wave loads before its first store
constexpr int kCopiesPerWave = 48;
constexpr uint32_t kLongCopy = 2048;

// up to 15 bytes as 8 + 4 + 2 + 1: the `n` low bytes of the piece are valid.  No register is indexed dynamically.
typedef uint16_t u16_unaligned __attribute__((aligned(1)));
__device__ __forceinline__ void load_short(const uint8_t *p, uint32_t n, u32x4_plain &v) {
    const uint8_t *q = p + (n & 8u);                                  // the part behind the whole 8 bytes
    uint32_t x0 = 0, x1 = 0, at = 0;                                  // up to 7 bytes: x0 = bytes 0..3, x1 = bytes 4..6
    if (n & 4u) { x0 = load_u32(q); at = 4; }
    uint32_t w = 0, sh = 0;
    if (n & 2u) { w = *(const ZR_GLOBAL u16_unaligned *)(q + at); sh = 16; at += 2; }
    if (n & 1u) w |= (uint32_t)load_u8(q + at) << sh;
    if (n & 4u) x1 = w; else x0 = w;
    if (n & 8u) { v[0] = load_u32(p); v[1] = load_u32(p + 4); v[2] = x0; v[3] = x1; }
    else { v[0] = x0; v[1] = x1; }
}
__device__ __forceinline__ void store_short(uint8_t *p, uint32_t n, const u32x4_plain &v) {
    uint8_t *q = p + (n & 8u);
    if (n & 8u) { *(ZR_GLOBAL u32_unaligned *)(p) = v[0]; *(ZR_GLOBAL u32_unaligned *)(p + 4) = v[1]; }
    const uint32_t x0 = (n & 8u) ? v[2] : v[0], x1 = (n & 8u) ? v[3] : v[1];
    uint32_t at = 0;
    if (n & 4u) { *(ZR_GLOBAL u32_unaligned *)(q) = x0; at = 4; }
    uint32_t w = (n & 4u) ? x1 : x0;
    if (n & 2u) { *(ZR_GLOBAL u16_unaligned *)(q + at) = (uint16_t)w; w >>= 16; at += 2; }
    if (n & 1u) *(ZR_GLOBAL uint8_t *)(q + at) = (uint8_t)w;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6)))
void chunkmemset_packed_kernel(uint8_t *__restrict__ base, const uint64_t *__restrict__ out_off,
                               const uint64_t *__restrict__ from_off, const uint32_t *__restrict__ len_in,
                               const uint32_t *__restrict__ left_in, size_t ncopies, uint32_t *__restrict__ list) {
    __shared__ CopyDesc desc_all[4][64];
    __shared__ uint32_t ends_all[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    CopyDesc *desc = desc_all[w];
    uint32_t *ends = ends_all[w];
    const size_t copy = ((size_t)blockIdx.x * 4 + (size_t)w) * kCopiesPerWave + (size_t)lane;

    CopyDesc d;
    d.out = d.from = 0;
    d.len = 0;
    d.pad = 0;
    if (lane < kCopiesPerWave && copy < ncopies) {
        d.out = out_off[copy];
        d.from = from_off[copy];
        d.len = len_in[copy];
        const uint32_t left = left_in[copy];
        if (d.len > left) d.len = left;                              // chunkset_tpl.h:236
    }
    const bool behind = d.from < d.out;
    const uint64_t dist = behind ? d.out - d.from : d.from - d.out;
    d.period = (behind && dist < d.len) ? (uint32_t)dist : 0xffffffffu;
    const bool listed = d.len > kLongCopy || (!behind && dist < d.len);   // long, or memmove-ahead overlap
    if (listed) list[2u + atomicAdd(&list[0], 1u)] = (uint32_t)copy;
    const uint32_t npieces = listed ? 0u : (d.len + 15u) >> 4;
    uint32_t end = npieces;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)end, k, 64);
        if (lane >= k) end += up;
    }
    d.start = end - npieces;
    desc[lane] = d;
    ends[lane] = end;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)end, 63);
    const bool any_pattern = __ballot(d.period != 0xffffffffu) != 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");           // the LDS image is this wave's own
    __builtin_amdgcn_s_waitcnt(0xc07f);                              // lgkmcnt(0)

    for (uint32_t s0 = 0; s0 < total; s0 += 64u * kCopyRounds) {
        u32x4_plain v[kCopyRounds];
        uint8_t *dst[kCopyRounds];
        uint32_t nb[kCopyRounds];           // bytes of the piece: 16 = one vector store, 1..15 = short, 0x100 | n = bytewise
        // ---- loads of the group
#pragma unroll
        for (int r = 0; r < kCopyRounds; ++r) {
            const uint32_t s = s0 + 64u * (uint32_t)r + (uint32_t)lane;
            nb[r] = 0;
            dst[r] = base;
            v[r] = u32x4_plain{0u, 0u, 0u, 0u};
            if (s < total) {
                uint32_t lo = 0, hi = 63;                             // first k with ends[k] > s
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (ends[mid] > s) hi = mid; else lo = mid + 1;
                }
                const CopyDesc c = desc[lo];
                const uint32_t j = s - c.start;
                uint32_t i = 16u * j, n = c.len - i;                  // piece [i, i + n)
                if (n >= 16u) n = 16u;
                else if (c.len >= 16u) { i = c.len - 16u; n = 16u; }  // ragged end: lay it over the last 16 bytes
                const uint8_t *from = base + c.from;
                dst[r] = base + c.out + i;
                uint32_t rr = i;
                bool wraps = false;
                if (any_pattern && c.period != 0xffffffffu) {         // wave-uniform gate: the modulo costs ~30 instructions
                    rr = i % c.period;
                    wraps = c.period - rr < n;
                }
                if (!wraps) {
                    if (n == 16u) v[r] = load_u128(from + rr);
                    else load_short(from + rr, n, v[r]);
                    nb[r] = n;
                } else {                                              // the pattern wraps inside the piece
                    for (uint32_t q = 0; q < n; ++q) {
                        v[r][q >> 2] |= (uint32_t)load_u8(from + rr) << (8u * (q & 3u));
                        rr = rr + 1u == c.period ? 0u : rr + 1u;
                    }
                    nb[r] = n | 0x100u;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);                               // every load of the group has landed
        // ---- stores of the group
#pragma unroll
        for (int r = 0; r < kCopyRounds; ++r) {
            if (nb[r] == 16u) {
                __builtin_nontemporal_store(v[r], (ZR_GLOBAL u32x4_unaligned *)dst[r]);   // written once, streamed
            } else if (nb[r] & 0x100u) {
                const uint32_t n = nb[r] & 0xffu;
                for (uint32_t q = 0; q < n; ++q) *(ZR_GLOBAL uint8_t *)(dst[r] + q) = (uint8_t)(v[r][q >> 2] >> (8u * (q & 3u)));
            } else if (nb[r]) {
                store_short(dst[r], nb[r], v[r]);
            }
        }
    }
}

// The copies the packed kernel put on the list: sixteen lanes per copy, sixteen bytes per lane, a round moves 256 bytes
// with one unaligned dwordx4 load and one dwordx4 store per lane; rounds go in ascending order and every load of a
// round lands before any store of that round is issued, which for `from` ahead of `out` is exactly memmove.
// list[0] = entries, list[2..] = copy indices.
__global__ __launch_bounds__(256)
void chunkmemset_list_kernel(uint8_t *__restrict__ base, const uint64_t *__restrict__ out_off,
                             const uint64_t *__restrict__ from_off, const uint32_t *__restrict__ len_in,
                             const uint32_t *__restrict__ left_in, uint32_t *__restrict__ list) {
    const uint32_t l = threadIdx.x & 15u;
    const uint32_t count = list[0];
    const uint32_t ngroups = gridDim.x * (blockDim.x >> 4);
    // entries are dealt round robin over the 16-lane groups of the grid (a shared "next" counter would be one
    // contended atomic per group even when the list is empty, which is the inflate caller's case)
    for (uint32_t k = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4); k < count; k += ngroups) {
        const uint32_t copy = list[2u + k];
        uint8_t *out = base + out_off[copy];
        const uint8_t *from = base + from_off[copy];
        uint32_t len = len_in[copy];
        const uint32_t left = left_in[copy];
        if (len > left) len = left;
        const bool behind = from < out;
        const uint64_t dist = behind ? (uint64_t)(out - from) : (uint64_t)(from - out);
        const bool pattern = behind && dist < len;
        const uint32_t period = pattern ? (uint32_t)dist : 0xffffffffu;   // source bytes repeat with this period
        for (uint32_t round = 0; round < len;) {
            // this round: `full` sixteen-byte lanes, then `ntail` < 16 single bytes (only in a copy's last round)
            const uint32_t remaining = len - round;
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
            if (l < full) __builtin_nontemporal_store(v, (ZR_GLOBAL u32x4_unaligned *)(out + i));
            if (l < ntail) *(ZR_GLOBAL uint8_t *)(out + p) = tb;
            round += span;
        }
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
    if (ncopies >> 32) {
        set_error("more than 2^32 copies in one batch");
        return ZNG_ROCM_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    Workspace *ws = workspace_for(st);
    if (!ws) return ZNG_ROCM_ENOMEM;
    std::lock_guard<std::mutex> use(ws->mu);
    uint32_t *list = nullptr;                    // {entries, next, copy indices ...}: the copies the packed kernel leaves out
    if (int rc = scratch_reserve(ws, kScrChunkList, (ncopies + 2) * sizeof(uint32_t), false, (void **)&list)) return rc;
    ZR_HIP(hipMemsetAsync(list, 0, 2 * sizeof(uint32_t), st));
    const size_t per_block = 4 * kCopiesPerWave;
    ZR_LAUNCH_TRACED(chunkmemset_packed_kernel, dim3((unsigned)((ncopies + per_block - 1) / per_block)), dim3(256), st, d_base,
                     d_out_off, d_from_off, d_len, d_left, ncopies, list);
    ZR_HIP(hipGetLastError());
    hipLaunchKernelGGL(chunkmemset_list_kernel, dim3((unsigned)(ctx()->cus * 8)), dim3(256), 0, st, d_base, d_out_off,
                       d_from_off, d_len, d_left, list);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

uint32_t zng_rocm_chunksize(void) { return 16; }

}  // extern "C"
