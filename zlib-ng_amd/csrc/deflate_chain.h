// deflate_chain.h -- the chain-walking LZ77 front end of the level-6 class (deflate_medium.c:145-277 with
// longest_match, match_tpl.h:26-280), for ONE workgroup of 16 wavefronts working on one segment.
//
// Everything a chain walk touches lives in LDS (144.6 KiB of the CU's 160):
//   head   4096 x u32   most recent position + 1 per hash bucket (insert_string_tpl.h:58-75)
//   prev   32768 x u16  distance from a position to the previous one in its bucket (the `prev` links of
//                       insert_string_tpl.h:98-102, as deltas; 0 = none)
//   ring   64 KiB + 512 the plaintext itself, position q at ring[q & 0xffff]; the first 512 bytes are mirrored
//                       behind the end so that a compare never has to wrap.  This is the reference's sliding
//                       window (deflate.h:176-190, fill_window deflate.c:1241-1330) held on chip.
// The walk of longest_match is a gather: every lane follows its own chain and compares its own candidates.  With
// the window in HBM each probe is a 64-cache-line gather and the texture addresser is the bound (measured: 82 us
// per batch of 1024 positions, 82 ms for 256 MiB); from LDS the same probes are ds_read_b32.
//
// The batch structure (wave-ordered insert, speculative per-wave parse, stitch) is that of deflate_lz.h.
#pragma once
#include "deflate_lz.h"

namespace zr {

constexpr int      kChainHashBits = 12;
constexpr int      kChainWaves = 16;               // 1024 lanes walk chains per segment, one segment per CU
constexpr int      kChainBatch = 64 * kChainWaves;
constexpr uint32_t kRingBytes = 65536u, kRingMirror = 512u;
constexpr uint32_t kRingAhead = 1024u;             // plaintext kept beyond the end of the running batch

struct ChainShared {
    uint32_t head[1 << kChainHashBits];
    uint16_t prev[32768];
    uint8_t  ring[kRingBytes + kRingMirror];
    uint32_t last_start[kChainWaves];   // per region: start of its last token if that token is a match, else kLzNone
    uint32_t exit_pos[kChainWaves];     // per region: first position after its last token
    uint32_t cover;                     // bytes below this position are already produced (carried across batches)
};

typedef uint32_t u32_lds_unaligned __attribute__((aligned(1)));

__device__ __forceinline__ uint32_t ring_u32(const uint8_t *ring, uint32_t idx) {      // idx < kRingBytes + mirror - 3
    return *reinterpret_cast<const u32_lds_unaligned *>(ring + idx);
}

// common prefix of the strings at positions p and c, counted from `from` (bytes below are known equal), capped at
// `cap` and `maxlen`; 16 bytes per step
__device__ __forceinline__ uint32_t ring_common_prefix(const uint8_t *ring, uint32_t p, uint32_t c, uint32_t from,
                                                       uint32_t cap, uint32_t maxlen) {
    const uint32_t pi = p & (kRingBytes - 1u), ci = c & (kRingBytes - 1u);     // from + 16 <= cap + 16 < mirror
    uint32_t l = from;
    while (l < cap) {
        if (l + 16 <= maxlen) {
            uint32_t x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = ring_u32(ring, pi + l + 4u * j) ^ ring_u32(ring, ci + l + 4u * j);
            uint32_t add = 16;
#pragma unroll
            for (int j = 3; j >= 0; --j)
                if (x[j]) add = 4u * (uint32_t)j + ((uint32_t)(__ffs((int)x[j]) - 1) >> 3);
            l += add;
            if (add < 16u) return l;
        } else {
            while (l < maxlen && ring[pi + l] == ring[ci + l]) ++l;
            return l < cap ? l : (l < maxlen ? l : maxlen);
        }
    }
    return l;
}

// number of equal leading bytes of the strings at positions a and b over [0, rem), rem <= 256, all 64 lanes
__device__ __forceinline__ uint32_t ring_extend_wave(const uint8_t *ring, uint32_t a, uint32_t b, uint32_t rem, int lane) {
    const uint32_t ai = a & (kRingBytes - 1u), bi = b & (kRingBytes - 1u);
    const uint32_t off = 4u * (uint32_t)lane;
    uint32_t x = 0;
    if (off + 4 <= rem) {
        x = ring_u32(ring, ai + off) ^ ring_u32(ring, bi + off);
    } else if (off < rem) {
        for (uint32_t j = 0; off + j < rem; ++j) x |= (uint32_t)(ring[ai + off + j] ^ ring[bi + off + j]) << (8 * j);
    }
    const unsigned long long diff = __ballot(x != 0);
    if (diff == 0) return rem;
    const int first = __ffsll((long long)diff) - 1;
    const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)x, first);
    const uint32_t n = (uint32_t)first * 4u + ((uint32_t)(__ffs((int)d) - 1) >> 3);
    return n < rem ? n : rem;
}

// One batch of kChainBatch positions starting at P; the ring holds every byte of [P - 32768, P + batch + kRingAhead)
// that lies below n.  `insert_only` batches just enter their positions (dictionary priming).
//   walk: up to `max_chain` links keeping the longest match, the loop of longest_match (match_tpl.h:129-268) with its
//   end-of-best-match quick reject (:167-173) and the good_match easing (:86-89); then a one-step lazy evaluation in
//   the parse (a longer match at the next byte wins, cf. max_lazy 16 of deflate.c:163).
__device__ __forceinline__ LzPick chain_batch(uint32_t n, uint32_t P, ChainShared *sh, int t, uint32_t max_chain,
                                              bool insert_only, uint32_t good_match) {
    const uint8_t *ring = sh->ring;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);        // wave-uniform: keeps the parse scalar
    const uint32_t p = P + (uint32_t)t;
    const bool can = p + kLzMinMatch <= n;              // lookahead >= WANT_MIN_MATCH, deflate_quick.c:88
    const uint32_t val = ring_u32(ring, p & (kRingBytes - 1u));
    const uint32_t h = lz_hash<kChainHashBits>(val);

    // 1. insert, waves in position order
    uint32_t old = 0;
    for (int w = 0; w < kChainWaves; ++w) {
        if (wave == w && can) old = atomicExch(&sh->head[h], p + 1u);
        __syncthreads();
    }
    if (can) {
        const uint32_t d = old ? p - (old - 1u) : 0u;
        sh->prev[p & 32767u] = (uint16_t)(d <= 65535u ? d : 0u);
    }
    LzPick r;
    r.kind = 0;
    r.len = r.dist = 0;
    if (insert_only) return r;
    __syncthreads();                                     // links of this batch are visible to every walker

    // 2a. the chain head alone (what the level-1 class does): a first estimate for every position
    uint32_t len = 0, dist = 0;
    const uint32_t maxlen = p < n ? ((n - p) < kStdMaxMatch ? (n - p) : kStdMaxMatch) : 0u;
    uint32_t best = 3;                                   // a match must reach WANT_MIN_MATCH to count
    uint32_t c = kLzNone;
    if (old && maxlen >= kLzMinMatch) {
        c = old - 1u;
        if (c < p && p - c <= kLzMaxDist) {
            if (ring_u32(ring, c & (kRingBytes - 1u)) == val) {
                const uint32_t l = ring_common_prefix(ring, p, c, 4, kLzChainProbe, maxlen);
                best = l;
                dist = p - c;
            }
            const uint32_t d = sh->prev[c & 32767u];
            c = (d == 0 || d > c) ? kLzNone : c - d;
        } else {
            c = kLzNone;
        }
    }
    // 2b. which positions can the parse stop at?  Hop through this region with the estimates: token starts
    //     (literals and matches) and the byte after a match start (lazy evaluation looks there).  Only those
    //     lanes walk their chain -- the serial coder never searches inside a match it has already taken
    //     (deflate_medium.c:187-239).
    bool hot;
    {
        const uint32_t w0q = P + 64u * (uint32_t)wave;
        const uint32_t limq = w0q >= n ? 0u : ((n - w0q) < 64u ? (n - w0q) : 64u);
        const unsigned long long M0 = __ballot(best >= kLzMinMatch);
        unsigned long long hm = 0;
        uint32_t pos = 0;
        while (pos < limq) {
            const unsigned long long rest = M0 >> pos;
            if (rest == 0) {
                hm |= lz_bits_below(limq) & ~lz_bits_below(pos);
                break;
            }
            const uint32_t m = pos + (uint32_t)(__ffsll((long long)rest) - 1);
            const uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)best, (int)m);
            // [pos, m] and m + 1 always; the inside of a SHORT estimated match too (a deeper walk there often
            // finds something better); only the inside of a long match is left alone
            const uint32_t upto = L < 12u ? m + L : m + 2u;
            hm |= lz_bits_below(upto < 64u ? upto : 64u) & ~lz_bits_below(pos);
            pos = m + L;
        }
        hot = (hm >> lane) & 1ull;
    }
    // 2c. the walk
    if (hot && c != kLzNone && best < kLzChainProbe && best < maxlen) {
        uint32_t chain = max_chain > 1 ? max_chain - 1 : 0;
        bool done = chain == 0, eased = false;
        if (best >= good_match) {
            chain >>= 2;
            eased = true;
            done = chain == 0;
        }
        const uint32_t pi = p & (kRingBytes - 1u);
        // One candidate per trip: its link and its quick-reject word (match_tpl.h:141-173: the 4 bytes ending at
        // `best` must agree; the first-bytes check of the template is left to the full compare, one random LDS read
        // fewer per candidate) are two independent LDS reads, so a trip costs one LDS latency.
        uint32_t tail_off = best - 3;
        uint32_t want_tail = ring_u32(ring, pi + tail_off);          // best < maxlen here
        uint32_t mine[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) mine[j] = ring_u32(ring, pi + 4u * j);
        if (!(c < p && p - c <= kLzMaxDist)) done = true;
        while (!done) {
            const uint32_t ci = c & (kRingBytes - 1u);
            const uint32_t d = sh->prev[c & 32767u];
            const uint32_t tail = ring_u32(ring, ci + tail_off);
            if (tail == want_tail) {
                uint32_t l;
                if (maxlen >= 16u) {                    // our first 16 bytes are in registers: read only the candidate's
                    uint32_t add = 16;
#pragma unroll
                    for (int j = 3; j >= 0; --j) {
                        const uint32_t x = ring_u32(ring, ci + 4u * j) ^ mine[j];
                        if (x) add = 4u * (uint32_t)j + ((uint32_t)(__ffs((int)x) - 1) >> 3);
                    }
                    l = add < 16u ? add : ring_common_prefix(ring, p, c, 16, kLzChainProbe, maxlen);
                } else {
                    l = ring_common_prefix(ring, p, c, 0, kLzChainProbe, maxlen);
                }
                if (l > best) {
                    best = l;
                    dist = p - c;
                    if (l >= kLzChainProbe || l >= maxlen) done = true;     // nice_match reached
                    if (!eased && best >= good_match) {     // "do not waste too much time if we already have a
                        chain >>= 2;                        //  good match" (match_tpl.h:86-89), applied as soon
                        eased = true;                       //  as the walk itself has found one
                    }
                    if (!done) {
                        tail_off = best - 3;
                        want_tail = ring_u32(ring, pi + tail_off);
                    }
                }
            }
            if (chain <= 1u || d == 0u || d > c) {
                done = true;
            } else {
                --chain;
                c -= d;
                if (p - c > kLzMaxDist) done = true;
            }
        }
    }
    if (best >= kLzMinMatch) len = best;

    // 3a. speculative parse of this wave's region [w0, w0 + lim)
    const uint32_t w0 = P + 64u * (uint32_t)wave;
    const uint32_t lim = w0 >= n ? 0u : ((n - w0) < 64u ? (n - w0) : 64u);
    // Scalar code, kept short (see deflate_lz.h: the scalar unit is the first thing such a loop saturates): no 64-bit
    // selects, every shift count below 64 by construction, a covered-bytes mask instead of per-run literal masks.
    const unsigned long long limmask = lz_bits_below(lim);
    unsigned long long avail = __ballot(len >= kLzMinMatch) & limmask;   // matches not yet hopped over
    unsigned long long covered = 0;                      // bytes inside a chosen match, behind its first byte
    uint32_t last_m = 64u, end = 0;                      // the last chosen match and the first byte after it
    while (avail) {
        uint32_t m = (uint32_t)__builtin_ctzll(avail);
        uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)m);
        // lazy evaluation: a strictly longer match one byte later turns this byte into a literal
        while (L < 16u && m + 1u < lim) {
            const uint32_t L2 = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)(m + 1u));
            if (L2 <= L) break;
            ++m;
            L = L2;
        }
        if (L >= kLzChainProbe) {                        // the probe saturated: measure the rest wave-wide
            const uint32_t pabs = w0 + m;
            const uint32_t mlen = (n - pabs) < kStdMaxMatch ? (n - pabs) : kStdMaxMatch;
            if (L < mlen) {
                const uint32_t D = (uint32_t)__builtin_amdgcn_readlane((int)dist, (int)m);
                uint32_t rem = mlen - L;
                if (rem > 256u) rem = 256u;
                L += ring_extend_wave(ring, pabs + L, pabs - D + L, rem, lane);
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
    for (int v = 0; v < kChainWaves; ++v) {
        if (v == wave) cover = carry;
        const uint32_t ls = sh->last_start[v], ex = sh->exit_pos[v];
        const uint32_t region_end = P + 64u * (uint32_t)(v + 1);
        const uint32_t lim_end = region_end < n ? region_end : n;
        const bool kept = ls == kLzNone || ls >= carry;            // the region's last token survives the cover
        const uint32_t out = kept ? ex : lim_end;
        carry = carry > out ? carry : out;
    }
    __syncthreads();                                     // everyone has read sh->cover / last_start / exit_pos
    if (t == 0) sh->cover = carry;

    r.len = len;
    r.dist = dist;
    if (p < n && p >= cover) {
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
