// deflate_lz.h -- the wave-parallel LZ77 front end shared by the device deflate kernels.
//
// One 256-lane workgroup per stream, 256 consecutive positions per step ("batch"):
//   1. wavefront-wide insert_string: every lane hashes its 4 bytes (insert_string.c:11-13) and swaps its
//      position into the LDS head table with ONE ds_wrxchg_rtn_b32.  The LDS unit serialises lanes that
//      hit the same slot in lane order (checked by tools/micro/lds_xchg_order.hip on gfx950), so the value a
//      lane gets back is exactly what quick_insert_string (insert_string_tpl.h:58-75) would have returned
//      had the 64 positions been inserted one after the other; the four waves of a batch take turns.
//      (Correctness never depends on that order: every candidate is verified byte by byte.)
//   2. per-lane probe: first kProbe bytes against the candidate (the zng_memcmp_2 + compare256 of
//      deflate_quick.c:96-97, bounded).
//   3. greedy parse in position order.  Each wave builds ballot(len >= 4) and hops from match to match
//      (literal runs are skipped with one ctz), long matches are extended with the wavefront-wide compare256.
// The head table holds absolute positions + 1 (u32, 0 = empty), so there is no window slide; a stale entry
// simply fails the distance check (dist <= MAX_DIST, deflate.h:410-415).
#pragma once
#include "deflate_dev.h"

namespace zr {

constexpr int      kLzHashBits = 14;                          // 16384 x u32 = 64 KiB of LDS: two streams per CU
constexpr uint32_t kLzMaxDist = 32768u - kMinLookahead;       // MAX_DIST(s)
constexpr uint32_t kLzProbe = 32;                             // bytes compared per lane before the parse
constexpr uint32_t kLzMinMatch = 4;                           // WANT_MIN_MATCH (deflate.h)

__device__ __forceinline__ uint32_t lz_hash(uint32_t val) { return (val * 2654435761u) >> (32 - kLzHashBits); }

// number of equal leading bytes of a[0..rem) and b[0..rem), rem <= 256, all 64 lanes take part
__device__ __forceinline__ uint32_t lz_extend_wave(const uint8_t *a, const uint8_t *b, uint32_t rem, int lane) {
    const uint32_t off = 4u * (uint32_t)lane;
    uint32_t x = 0;
    if (off + 4 <= rem) {
        x = load_u32(a + off) ^ load_u32(b + off);
    } else if (off < rem) {
        for (uint32_t j = 0; off + j < rem; ++j) x |= (uint32_t)(a[off + j] ^ b[off + j]) << (8 * j);
    }
    const unsigned long long diff = __ballot(x != 0);
    if (diff == 0) return rem;
    const int first = __ffsll((long long)diff) - 1;
    const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)x, first);
    const uint32_t n = (uint32_t)first * 4u + ((uint32_t)(__ffs((int)d) - 1) >> 3);
    return n < rem ? n : rem;
}

__device__ __forceinline__ unsigned long long lz_bits_below(uint32_t b) {      // bits [0, b), b <= 64
    return b >= 64 ? ~0ull : ((1ull << b) - 1ull);
}

// Result of one batch for this lane's position.
struct LzPick {
    bool     visited;     // the parse stops at this position (emits a literal or starts a match here)
    uint32_t len;         // >= kLzMinMatch: match length, else literal
    uint32_t dist;
};

// One batch.  `head` = LDS table (1 << kLzHashBits entries), `sh_next` = LDS word holding the next
// position the parse will visit.  `val` = the 4 bytes at this lane's position (0 if fewer remain).
__device__ __forceinline__ LzPick lz_batch(const uint8_t *__restrict__ in, uint32_t n, uint32_t P, uint32_t val,
                                           uint32_t *head, uint32_t *sh_next, int t, uint32_t ablate = 0) {
    const int lane = t & 63, wave = t >> 6;
    const uint32_t p = P + (uint32_t)t;
    const bool can = p + kLzMinMatch <= n;              // lookahead >= WANT_MIN_MATCH, deflate_quick.c:88
    const uint32_t h = lz_hash(val);

    // 1. insert, waves in position order
    uint32_t old = 0;
    if (ablate & 8u) {                       // timing experiment only: unordered insert
        if (can) old = atomicExch(&head[h], p + 1u);
        __syncthreads();
    } else {
        for (int w = 0; w < 4; ++w) {
            if (wave == w && can) old = atomicExch(&head[h], p + 1u);
            __syncthreads();
        }
    }

    // 2. probe
    uint32_t len = 0, dist = 0;
    if (old) {
        const uint32_t c = old - 1u;
        if (c < p && p - c <= kLzMaxDist && load_u32(in + c) == val) {
            const uint32_t maxlen = (n - p) < kStdMaxMatch ? (n - p) : kStdMaxMatch;
            len = 4;
            while (len < kLzProbe && !(ablate & 1u)) {
                if (len + 4 <= maxlen) {
                    const uint32_t x = load_u32(in + p + len) ^ load_u32(in + c + len);
                    if (x) {
                        len += (uint32_t)(__ffs((int)x) - 1) >> 3;
                        break;
                    }
                    len += 4;
                } else {
                    while (len < maxlen && in[p + len] == in[c + len]) ++len;
                    break;
                }
            }
            dist = p - c;
        }
    }

    // 3. parse
    unsigned long long visited = 0;
    for (int w = 0; w < 4 && !(ablate & 4u); ++w) {
        if (wave == w) {
            const uint32_t w0 = P + 64u * (uint32_t)w;
            const unsigned long long M = __ballot(len >= kLzMinMatch);
            const uint32_t nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)*sh_next);
            uint32_t pos = nxt > w0 ? nxt - w0 : 0u;
            const uint32_t lim = w0 >= n ? 0u : ((n - w0) < 64u ? (n - w0) : 64u);   // positions of this wave that exist
            while (pos < lim) {
                const unsigned long long rest = M >> pos;
                if (rest == 0) {                                       // literals to the end of the wave
                    visited |= lz_bits_below(lim) & ~lz_bits_below(pos);
                    pos = lim;
                    break;
                }
                const uint32_t m = pos + (uint32_t)(__ffsll((long long)rest) - 1);
                visited |= lz_bits_below(m + 1) & ~lz_bits_below(pos);  // literals [pos, m) and the match start m
                uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)m);
                const uint32_t pabs = w0 + m;
                const uint32_t maxlen = (n - pabs) < kStdMaxMatch ? (n - pabs) : kStdMaxMatch;
                if (L >= kLzProbe && L < maxlen && !(ablate & 2u)) {
                    const uint32_t D = (uint32_t)__builtin_amdgcn_readlane((int)dist, (int)m);
                    uint32_t rem = maxlen - L;
                    if (rem > 256u) rem = 256u;
                    L += lz_extend_wave(in + pabs + L, in + pabs - D + L, rem, lane);
                    if ((uint32_t)lane == m) len = L;
                }
                pos = m + L;
            }
            if (lane == 0) *sh_next = (w0 + pos) > nxt ? (w0 + pos) : nxt;
        }
        __syncthreads();
    }
    LzPick r;
    r.visited = (visited >> lane) & 1ull;
    r.len = len;
    r.dist = dist;
    return r;
}

}  // namespace zr
