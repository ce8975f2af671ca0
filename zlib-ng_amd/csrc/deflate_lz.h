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
//      deflate_quick.c:96-97, bounded).
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
constexpr uint32_t kLzProbe = 36;                             // bytes compared per lane before the parse
constexpr uint32_t kLzMinMatch = 4;                           // WANT_MIN_MATCH (deflate.h)
constexpr uint32_t kLzNone = 0xffffffffu;

constexpr uint32_t kLzChainProbe = 64;                        // chain mode: per-lane compare cap (= "nice" length)

// HBITS: log2 of the head table.  The table is the LDS budget of a stream, and the number of streams a CU can
// keep in flight is what hides the latency of this (serial-per-stream) code: measured on MI355X, 4096 x 1 MiB,
// level-1 class: 14 bits (2 streams/CU) 33.5 GB/s ratio 1.950, 13 bits (4/CU) 47.7 GB/s ratio 1.934.
template <bool CHAIN, int HBITS>
struct LzShared {                    // LDS state of one stream
    uint32_t head[1 << HBITS];
    uint16_t prev[CHAIN ? 32768 : 2];  // CHAIN: distance from a position to the previous one with the same hash (0 = none);
                                     //        the `prev` links of insert_string_tpl.h:98-102, stored as deltas
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

// common prefix of in[p..] and in[c..], counted from `from` (bytes below are known equal), capped at `cap`
// and `maxlen`.  Eight dwords of each side are requested together, so one memory round trip covers 32 bytes.
__device__ __forceinline__ uint32_t lz_common_prefix(const uint8_t *__restrict__ in, uint32_t p, uint32_t c,
                                                     uint32_t from, uint32_t cap, uint32_t maxlen) {
    uint32_t l = from;
    while (l < cap) {
        if (l + 32 <= maxlen) {
            // two unaligned dwordx4 loads per side: 32 bytes for two TA passes instead of eight
            const u32x4_unaligned a0 = load_u128(in + p + l);
            const u32x4_unaligned a1 = load_u128(in + p + l + 16);
            const u32x4_unaligned b0 = load_u128(in + c + l);
            const u32x4_unaligned b1 = load_u128(in + c + l + 16);
            const uint32_t a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const uint32_t b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            int k = 0;
#pragma unroll
            for (int j = 7; j >= 0; --j)
                if (a[j] != b[j]) k = j - 8;            // lowest differing dword, encoded as j - 8 (< 0)
            if (k < 0) {
                const int j = k + 8;
                uint32_t x = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q == j) x = a[q] ^ b[q];
                return l + 4u * (uint32_t)j + ((uint32_t)(__ffs((int)x) - 1) >> 3);
            }
            l += 32;
        } else {
            while (l < maxlen && load_u8(in + p + l) == load_u8(in + c + l)) ++l;
            return l < cap ? l : (l < maxlen ? l : maxlen);
        }
    }
    return l;
}

struct LzPick {
    uint32_t kind;        // 0 = nothing to emit here, 1 = literal, 2 = match
    uint32_t len, dist;
};

// One batch.  `val` = the 4 bytes at this lane's position (0 if fewer remain).
// CHAIN = false: level-1 class, a single chain-head probe (deflate_quick.c:89-97).
// CHAIN = true : level-6 class: walk up to `max_chain` links keeping the longest match, the loop of
//                longest_match (match_tpl.h:129-268) with its end-of-best-match quick reject (:167-173), then a
//                one-step lazy evaluation in the parse (a longer match at the next byte wins, cf. max_lazy 16 of
//                deflate.c:163).  `insert_only` batches just enter their positions (dictionary priming).
template <bool CHAIN, int HBITS, int NW = 4>   // NW = wavefronts per workgroup (batch = 64 * NW positions)
__device__ __forceinline__ LzPick lz_batch(const uint8_t *__restrict__ in, uint32_t n, uint32_t P, uint32_t val,
                                           LzShared<CHAIN, HBITS> *sh, int t, uint32_t max_chain = 0,
                                           bool insert_only = false, uint32_t good_match = 0xffffu) {
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);        // wave-uniform: keeps the parse scalar
    const uint32_t p = P + (uint32_t)t;
    const bool can = p + kLzMinMatch <= n;              // lookahead >= WANT_MIN_MATCH, deflate_quick.c:88
    const uint32_t h = lz_hash<HBITS>(val);

    // 1. insert, waves in position order
    uint32_t old = 0;
    for (int w = 0; w < NW; ++w) {
        if (wave == w && can) old = atomicExch(&sh->head[h], p + 1u);
        __syncthreads();
    }
    if constexpr (CHAIN) {
        if (can) {
            const uint32_t d = old ? p - (old - 1u) : 0u;
            sh->prev[p & 32767u] = (uint16_t)(d <= 65535u ? d : 0u);
        }
        if (insert_only) {
            LzPick none;
            none.kind = 0;
            none.len = none.dist = 0;
            return none;
        }
        __syncthreads();                                 // links of this batch are visible to every walker
    }

    // 2. probe
    uint32_t len = 0, dist = 0;
    const uint32_t maxlen = p < n ? ((n - p) < kStdMaxMatch ? (n - p) : kStdMaxMatch) : 0u;
    if constexpr (!CHAIN) {
        if (old) {
            const uint32_t c = old - 1u;
            // Staged probe: the candidate side is a per-lane gather (every lane its own cache line), which is what
            // this kernel is bound by -- so 4 bytes first (most candidates die there), then 16, then 16 more.
            if (c < p && p - c <= kLzMaxDist && load_u32(in + c) == val) {
                uint32_t l = 4;
#pragma unroll
                for (int stage = 0; stage < 2 && l == 4u + 16u * (uint32_t)stage; ++stage) {
                    if (l + 16 <= maxlen) {
                        const u32x4_unaligned a = load_u128(in + p + l), b = load_u128(in + c + l);
                        const uint32_t x[4] = {a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w};
                        uint32_t add = 16;
#pragma unroll
                        for (int j = 3; j >= 0; --j)
                            if (x[j]) add = 4u * (uint32_t)j + ((uint32_t)(__ffs((int)x[j]) - 1) >> 3);
                        l += add;
                    } else {
                        while (l < maxlen && load_u8(in + p + l) == load_u8(in + c + l)) ++l;
                        break;
                    }
                }
                len = l;                                  // 36 = both stages matched: the parse extends it
                dist = p - c;
            }
        }
    } else {
        // 2a. the chain head alone (what the level-1 class does): a first estimate for every position
        uint32_t best = 3;                               // a match must reach WANT_MIN_MATCH to count
        uint32_t c = kLzNone;
        if (old && maxlen >= kLzMinMatch) {
            c = old - 1u;
            if (c < p && p - c <= kLzMaxDist) {
                const uint32_t l = lz_common_prefix(in, p, c, 0, kLzChainProbe, maxlen);
                if (l >= kLzMinMatch) {
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
        //     (deflate_medium.c:187-239), and the probes of a walk are what this kernel's time is made of.
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
            while (!done && c != kLzNone) {
                // walk up to kGather links first (LDS only), then fetch every candidate's probe words at once:
                // one memory round trip per kGather candidates instead of one per candidate
                constexpr int kGather = 8;
                uint32_t cand[kGather];
                int nc = 0;
#pragma unroll
                for (int k = 0; k < kGather; ++k) {
                    if (c == kLzNone || !(c < p && p - c <= kLzMaxDist) || chain == 0) {
                        c = kLzNone;
                        break;
                    }
                    cand[k] = c;
                    nc = k + 1;
                    --chain;
                    const uint32_t d = sh->prev[c & 32767u];
                    c = (d == 0 || d > c) ? kLzNone : c - d;
                }
                // quick reject (match_tpl.h:141-173): the 4 bytes ending at `best` and the first 4 must agree
                const uint32_t tail_off = best - 3;
                const bool can_tail = best + 1 <= maxlen;
                const uint32_t want_tail = can_tail ? load_u32(in + p + tail_off) : 0u;
                uint32_t ct[kGather], cf[kGather];
#pragma unroll
                for (int k = 0; k < kGather; ++k) {
                    ct[k] = cf[k] = 0;
                    if (k < nc) {
                        cf[k] = load_u32(in + cand[k]);
                        // no match yet (best == 3): the tail word IS the first word -- one gather, not two
                        ct[k] = !can_tail ? 0u : (tail_off == 0 ? cf[k] : load_u32(in + cand[k] + tail_off));
                    }
                }
                // Survivors of the quick reject, as a per-lane bit mask.  Each lane then measures ITS OWN next
                // survivor per pass: the passes a wave spends are the largest survivor count of any lane
                // (typically 1-2), not kGather -- the full compares are where the walk's time went.
                uint32_t pm = 0;
#pragma unroll
                for (int k = 0; k < kGather; ++k)
                    if (k < nc && can_tail && cf[k] == val && ct[k] == want_tail) pm |= 1u << k;
                while (pm && !done) {
                    const int k = __ffs((int)pm) - 1;
                    pm &= pm - 1u;
                    uint32_t ck = cand[0];
#pragma unroll
                    for (int q = 1; q < kGather; ++q) ck = (q == k) ? cand[q] : ck;
                    const uint32_t l = lz_common_prefix(in, p, ck, 4, kLzChainProbe, maxlen);
                    if (l > best) {
                        best = l;
                        dist = p - ck;
                        if (l >= kLzChainProbe || l >= maxlen) done = true;     // nice_match reached
                        if (!eased && best >= good_match) {     // "do not waste too much time if we already have a
                            chain >>= 2;                        //  good match" (match_tpl.h:86-89), applied as soon
                            eased = true;                       //  as the walk itself has found one
                            if (chain == 0) done = true;
                        }
                    }
                }
            }
        }
        if (best >= kLzMinMatch) len = best;
    }

    // 3a. speculative parse of this wave's region [w0, w0 + lim)
    const uint32_t w0 = P + 64u * (uint32_t)wave;
    const uint32_t lim = w0 >= n ? 0u : ((n - w0) < 64u ? (n - w0) : 64u);
    const unsigned long long M = __ballot(len >= kLzMinMatch);
    unsigned long long starts = 0;                       // token starts (literals and matches)
    unsigned long long mstarts = 0;                      // the starts that are matches (lazy evaluation can demote a lane
                                                         // with len >= 4 to a literal)
    uint32_t pos = 0, last_start = kLzNone;
    while (pos < lim) {
        const unsigned long long rest = M >> pos;
        if (rest == 0) {                                 // literals to the end of the region
            starts |= lz_bits_below(lim) & ~lz_bits_below(pos);
            pos = lim;
            last_start = kLzNone;
            break;
        }
        uint32_t m = pos + (uint32_t)(__ffsll((long long)rest) - 1);
        uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)m);
        if constexpr (CHAIN) {
            // lazy evaluation: a strictly longer match one byte later turns this byte into a literal
            while (L < 16u && m + 1u < lim) {
                const uint32_t L2 = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)(m + 1u));
                if (L2 <= L) break;
                ++m;
                L = L2;
            }
        }
        starts |= lz_bits_below(m + 1) & ~lz_bits_below(pos);          // literals [pos, m) and the match start m
        mstarts |= 1ull << m;
        const uint32_t pabs = w0 + m;
        const uint32_t maxlen = (n - pabs) < kStdMaxMatch ? (n - pabs) : kStdMaxMatch;
        if (L >= (CHAIN ? kLzChainProbe : kLzProbe) && L < maxlen) {
            const uint32_t D = (uint32_t)__builtin_amdgcn_readlane((int)dist, (int)m);
            uint32_t rem = maxlen - L;
            if (rem > 256u) rem = 256u;
            L += lz_extend_wave(in + pabs + L, in + pabs - D + L, rem, lane);
            if ((uint32_t)lane == m) len = L;
        }
        last_start = pabs;
        pos = m + L;
    }
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
        const uint32_t lim_end = region_end < n ? region_end : n;
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
