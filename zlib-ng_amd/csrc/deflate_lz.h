// deflate_lz.h -- the wave-parallel LZ77 front end shared by the device deflate kernels.
//
// One 256-lane workgroup per stream, 256 consecutive positions per step ("batch"):
//   1. wavefront-wide insert_string: every lane hashes its 4 bytes (insert_string.c:11-13) and swaps its
//      position into the LDS head table with ONE ds_wrxchg_rtn_b32.  The LDS unit serialises lanes that
//      hit the same slot in lane order (checked by tools/micro/lds_xchg_order.hip on gfx950), so the value a
//      lane gets back is exactly what quick_insert_string (insert_string_tpl.h:58-75) would have returned
//      had the 64 positions been inserted one after the other; the four waves of a batch take turns.
//      (Correctness never depends on that order: every candidate is verified byte by byte.)
//   2. per-lane probe: first kLzProbe bytes against the candidate (the zng_memcmp_2 + compare256 of
//      deflate_quick.c:96-97, bounded).  The lane's own 16 bytes arrive with the batch (one contiguous prefetch a
//      batch ahead); the candidate's 16 bytes are ONE unaligned dwordx4 gather, so a batch has one dependent memory
//      round trip on its critical path -- a second one only for the lanes whose first 16 bytes all agree.
//   3. parse.  A greedy parse is a serial chain through the whole stream; here every 64-position REGION
//      (one wave) parses speculatively from its own first byte, all regions at once: ballot(len >= 4),
//      hop from match to match with ctz, extend long matches with the wavefront-wide compare256.
//      Afterwards the regions are stitched: a match that runs past its region covers the head of the next
//      one(s); tokens that start inside covered bytes are dropped, and the tail of a dropped token that
//      sticks out of the cover is emitted as literals.  The stitch is a 4-step max-recurrence on
//      {last match start, region exit}, so the only serial work per batch is O(1).
//      Every byte is produced exactly once -> the token stream is a valid LZ77 parse; compared with the
//      strictly serial greedy parse it differs only around region-crossing matches.
// The head table holds absolute positions + 1 (u32, 0 = empty), so there is no window slide; a stale entry
// simply fails the distance check (dist <= MAX_DIST, deflate.h:410-415).
#pragma once
#include "deflate_dev.h"

namespace zr {

constexpr uint32_t kLzMaxDist = 32768u - kMinLookahead;       // MAX_DIST(s)
constexpr uint32_t kLzProbe = 32;                             // bytes compared per lane before the parse
constexpr uint32_t kLzMinMatch = 4;                           // WANT_MIN_MATCH (deflate.h)
constexpr uint32_t kLzNone = 0xffffffffu;



// HBITS: log2 of the head table.  The table is the LDS budget of a stream, and the number of streams a CU can
// keep in flight is what hides the latency of this (serial-per-stream) code: measured on MI355X, 4096 x 1 MiB,
// level-1 class: 14 bits (2 streams/CU) 33.5 GB/s ratio 1.950, 13 bits (4/CU) 47.7 GB/s ratio 1.934.
template <int HBITS>
struct LzShared {                    // LDS state of one stream
    uint32_t head[1 << HBITS];
    uint32_t last_start[16];         // per region: start of its last token if that token is a match, else kLzNone
    uint32_t exit_pos[16];           // per region: first position after its last token
    uint32_t cover;                  // bytes below this position are already produced (carried across batches)
};

template <int HBITS>
__device__ __forceinline__ uint32_t lz_hash(uint32_t val) { return (val * 2654435761u) >> (32 - HBITS); }

// number of equal leading bytes of a[0..rem) and b[0..rem), rem <= 256, all 64 lanes take part
__device__ __forceinline__ uint32_t lz_extend_wave(const uint8_t *a, const uint8_t *b, uint32_t rem, int lane) {
    const uint32_t off = 4u * (uint32_t)lane;
    uint32_t x = 0;
    if (off + 4 <= rem) {
        x = load_u32(a + off) ^ load_u32(b + off);
    } else if (off < rem) {
        for (uint32_t j = 0; off + j < rem; ++j) x |= (uint32_t)(load_u8(a + off + j) ^ load_u8(b + off + j)) << (8 * j);
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

struct LzPick {
    uint32_t kind;        // 0 = nothing to emit here, 1 = literal, 2 = match
    uint32_t len, dist;
};

// bytes [p, p + 16) of the stream; bytes at or beyond n read as 0 (only the last batches of a stream get here)
__device__ __forceinline__ u32x4_unaligned load_16_guarded(const uint8_t *in, uint32_t p, uint32_t n) {
    if (p + 16u <= n && p + 16u >= p) return load_u128(in + p);
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    for (uint32_t j = 0; j < 16u && p + j < n && p + j >= p; ++j) w[j >> 2] |= (uint32_t)load_u8(in + p + j) << (8u * (j & 3u));
    u32x4_unaligned r = {w[0], w[1], w[2], w[3]};
    return r;
}

// number of equal leading bytes of two 16-byte pieces (16 = all equal)
__device__ __forceinline__ uint32_t lz_prefix16(const u32x4_unaligned &a, const u32x4_unaligned &b) {
    const uint32_t x[4] = {a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w};
    uint32_t l = 16;
#pragma unroll
    for (int j = 3; j >= 0; --j) l = x[j] ? 4u * (uint32_t)j + ((uint32_t)(__ffs((int)x[j]) - 1) >> 3) : l;
    return l;
}

// A batch that only enters its positions into the hash (dictionary priming: deflateSetDictionary's
// insert_string loop, deflate.c:499-515): the same wave-ordered exchange as lz_batch, nothing else.
template <int HBITS, int NW = 4>
__device__ __forceinline__ void lz_insert_batch(uint32_t n, uint32_t P, uint32_t first4, LzShared<HBITS> *sh, int t) {
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t p = P + (uint32_t)t;
    const bool can = p + kLzMinMatch <= n;
    const uint32_t h = lz_hash<HBITS>(first4);
    for (int w = 0; w < NW; ++w) {
        if (wave == w && can) (void)atomicExch(&sh->head[h], p + 1u);
        __syncthreads();
    }
}

// One batch of the level-1 class: a single chain-head probe per position (deflate_quick.c:89-97).
// `own` = the 16 bytes at this lane's position (zero beyond the end of the stream).
// FULL: the caller guarantees that every position of the batch has its whole 258-byte lookahead inside the stream
// (all batches but the last two of a stream) -- lim, maxlen and the end-of-input guards fold to constants, which
// matters because the scalar instructions they cost are the kernel's bound.
template <int HBITS, int NW = 4, bool FULL = false>   // NW = wavefronts per workgroup (batch = 64 * NW positions)
__device__ __forceinline__ LzPick lz_batch(const uint8_t *__restrict__ in, uint32_t n, uint32_t P,
                                           const u32x4_unaligned &own, LzShared<HBITS> *sh, int t) {
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);        // wave-uniform: keeps the parse scalar
    const uint32_t p = P + (uint32_t)t;
    const bool can = FULL || p + kLzMinMatch <= n;      // lookahead >= WANT_MIN_MATCH, deflate_quick.c:88
    const uint32_t h = lz_hash<HBITS>(own.x);

    // 1. insert, waves in position order
    uint32_t old = 0;
    for (int w = 0; w < NW; ++w) {
        if (wave == w && can) old = atomicExch(&sh->head[h], p + 1u);
        __syncthreads();
    }

    // 2. probe: one gather of the candidate's first 16 bytes; a second round only behind 16 equal bytes
    uint32_t len = 0, dist = 0;
    const uint32_t maxlen = FULL ? kStdMaxMatch : (p < n ? ((n - p) < kStdMaxMatch ? (n - p) : kStdMaxMatch) : 0u);
    {
        const uint32_t c = old - 1u;                     // old == 0 -> 0xffffffff: fails c < p
        if (c < p && p - c <= kLzMaxDist) {
            uint32_t l = lz_prefix16(FULL ? load_u128(in + c) : load_16_guarded(in, c, n), own);
            if (l == 16u && maxlen > 16u)
                l += lz_prefix16(FULL ? load_u128(in + c + 16u) : load_16_guarded(in, c + 16u, n),
                                 FULL ? load_u128(in + p + 16u) : load_16_guarded(in, p + 16u, n));
            l = l < maxlen ? l : maxlen;
            len = l >= kLzMinMatch ? l : 0u;             // kLzProbe = both rounds matched: the parse extends it
            dist = p - c;
        }
    }

    // 3a. speculative parse of this wave's region [w0, w0 + lim)
    const uint32_t w0 = P + 64u * (uint32_t)wave;
    const uint32_t lim = FULL ? 64u : (w0 >= n ? 0u : ((n - w0) < 64u ? (n - w0) : 64u));
    // This loop is scalar code, and the scalar unit is what bounds the kernel (rocprofv3 counters: 364 SALU against
    // 127 VALU instructions per 64 positions before this form) -- so it is written to stay short: no 64-bit selects,
    // every shift count below 64 by construction, the covered-bytes mask instead of per-run literal masks.
    const unsigned long long limmask = FULL ? ~0ull : lz_bits_below(lim);
    unsigned long long avail = __ballot(len >= kLzMinMatch) & limmask;   // matches not yet hopped over
    unsigned long long covered = 0;                      // bytes inside a chosen match, behind its first byte
    uint32_t last_m = 64u, end = 0;                      // the last chosen match and the first byte after it
    while (avail) {
        const uint32_t m = (uint32_t)__builtin_ctzll(avail);
        uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)m);
        if (L >= kLzProbe) {                             // the probe saturated: measure the rest wave-wide
            const uint32_t pabs = w0 + m;
            const uint32_t mlen = FULL ? kStdMaxMatch : ((n - pabs) < kStdMaxMatch ? (n - pabs) : kStdMaxMatch);
            if (L < mlen) {
                const uint32_t D = (uint32_t)__builtin_amdgcn_readlane((int)dist, (int)m);
                uint32_t rem = mlen - L;
                if (rem > 256u) rem = 256u;
                L += lz_extend_wave(in + pabs + L, in + pabs - D + L, rem, lane);
                if ((uint32_t)lane == m) len = L;
            }
        }
        last_m = m;
        end = m + L;
        if (end >= lim) break;                           // lim <= 64: every shift below is by less than 64
        covered |= (~0ull << (m + 1u)) & ~(~0ull << end);
        avail &= ~0ull << end;
    }
    uint32_t pos = lim, last_start = kLzNone;            // literals to the end of the region ...
    unsigned long long mstarts = 0;                      // chosen matches: the byte before a covered run ...
    if (end >= lim && last_m < 64u) {                    // ... unless its last token is a match
        pos = end;
        last_start = w0 + last_m;
        if (last_m < 63u) covered |= ~0ull << (last_m + 1u);
        else mstarts = 1ull << 63;                       // ... and a match that starts on the region's last byte
    }
    mstarts |= ~covered & (covered >> 1);
    const unsigned long long starts = ~covered & limmask;   // token starts: literals and matches
    if (lane == 0) {
        sh->last_start[wave] = last_start;
        sh->exit_pos[wave] = w0 + pos;
    }
    __syncthreads();

    // 3b. stitch: cover = first position not yet produced when this region starts
    uint32_t cover = sh->cover;
    uint32_t carry = cover;
    for (int v = 0; v < NW; ++v) {
        if (v == wave) cover = carry;
        const uint32_t ls = sh->last_start[v], ex = sh->exit_pos[v];
        const uint32_t region_end = P + 64u * (uint32_t)(v + 1);
        const uint32_t lim_end = FULL ? region_end : (region_end < n ? region_end : n);
        const bool kept = ls == kLzNone || ls >= carry;            // the region's last token survives the cover
        const uint32_t out = kept ? ex : lim_end;
        carry = carry > out ? carry : out;
    }
    __syncthreads();                                     // everyone has read sh->cover / last_start / exit_pos
    if (t == 0) sh->cover = carry;

    LzPick r;
    r.kind = 0;
    r.len = len;
    r.dist = dist;
    if ((FULL || p < n) && p >= cover) {
        if ((starts >> lane) & 1ull) {
            r.kind = ((mstarts >> lane) & 1ull) ? 2u : 1u;
        } else {
            // inside a speculative match that starts at q: orphaned if that match was dropped
            const unsigned long long below = starts & lz_bits_below((uint32_t)lane);
            const uint32_t q = w0 + (uint32_t)(63 - __clzll((long long)below));
            if (q < cover) r.kind = 1u;
        }
    }
    return r;
}

}  // namespace zr
