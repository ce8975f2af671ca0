// deflate_rows.h -- the LZ77 front end of the level-6 class (the work of deflate_medium.c:145-277 with longest_match,
// match_tpl.h:26-280, and insert_string, insert_string_tpl.h:48-104), for ONE workgroup of 16 wavefronts on one segment.
//
// Round 3 form.  The reference finds matches by walking a hash chain (head[65536] -> prev[] links, match_tpl.h:129-268):
// a pointer chase, one dependent access per link.  On a CU that walk was the whole kernel (round 2: 37 links per walking
// position, 4.4 LDS wave-instructions per input byte, LDS index unit 75 % busy, profiles/r03_pmc_lz_chain_before_*).
// Here the search structure is an ASSOCIATIVE ROW instead of a chain:
//   rows  kRows x kRowEnt entries {position & 0xffff (u16), tag (u8)}, a round-robin byte counter per row.
//         A 4-byte string selects its row with the reference's multiplicative hash (insert_string.c:11-13, scaled to
//         kRows by a multiply-high) and carries an 8-bit tag from a second multiplier.  insert = one LDS add
//         (the counter) and two LDS stores; a row always holds the kRowEnt most recent positions with its hash.
//         search = ONE 8-byte read (the tags) + ONE 16-byte read (the positions): the candidates are the entries whose
//         tag equals the searcher's -- with 12 + 8 bits of key nearly all of them are true 4-byte matches -- so a
//         position costs ~3 candidate compares instead of ~37 links, and none of them depends on another.
//   ring  64 KiB + 512: the plaintext, position q at ring[q & 0xffff] (the reference's sliding window, deflate.h:176-190,
//         fill_window deflate.c:1241-1330, held on chip), the first 512 bytes mirrored behind the end.
// A batch is 1024 consecutive positions, one per lane.  Every lane reads its row BEFORE the batch is inserted (the
// candidates in front of the batch, complete), the waves insert in position order, every lane reads its row again (the
// candidates inside the batch), so what a lane sees never depends on how the waves were scheduled: the output is a
// function of the input alone.
//
// The parse is not deflate_medium's greedy/lazy heuristic (deflate_medium.c:187-277) but a shortest-path parse per
// 64-position region (one wavefront): with the longest match known at EVERY position, cost[i] = min(literal + cost[i+1],
// min over l in 4..len[i] of match(l, dist[i]) + cost[i+l]) is evaluated backwards, 64 candidate lengths per step
// across the lanes (one add, one masked min-reduction by DPP).  Symbol costs come from the histogram of the segment's
// own tokens so far (refreshed every few batches), so the parse adapts to the data the way a second zlib pass would.
// A backward parse gives the best continuation from EVERY start position, which makes the stitching of the regions
// exact: region r + 1 starts where the path of region r really ends (an exit map per region, chased through LDS), no
// token is dropped and no byte falls back to a literal because two speculative parses disagreed (round 2's stitch).
#pragma once
#include "deflate_lz.h"

namespace zr {

constexpr int      kRowWaves = 16;                 // 1024 lanes per segment, one segment per CU
constexpr int      kRowBatch = 64 * kRowWaves;
constexpr int      kRows = 3584;                   // 7 x 512: what fits beside ring and tables in 160 KiB
constexpr int      kRowEnt = 8;
constexpr uint32_t kRingBytes = 65536u, kRingMirror = 512u;
constexpr uint32_t kRingAhead = 2048u;             // the ring is filled this far beyond the START of the running batch before the
                                                   // batch begins: a batch reads up to 1024 + 258 + 16 bytes beyond its start, and
                                                   // a chunk stored during a batch must not be something that batch reads
constexpr uint32_t kNiceLen = 64u;                 // a candidate this long ends the search (nice_match, deflate.c:163: 128)
constexpr uint32_t kCostBit = 16u;                 // cost units per bit
constexpr uint32_t kCostBias = 1u << 17;           // keeps the (negative) credit for bytes beyond a region positive
constexpr uint32_t kCostBeta = 40u;                // credit per byte a token reaches beyond its region: 2.5 bits
constexpr uint32_t kCostMax = 15u * kCostBit;
constexpr uint32_t kCostInf = 0xffffffffu;
constexpr int      kRefreshBatches = 4;            // symbol costs are recomputed from the histogram this often

#ifdef ZR_ROWS_STAMPS
#define ZR_STAMP(k) do { if (lane == 0) sh->stamps[wave][k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ZR_STAMP(k) do { } while (0)
#endif

struct RowShared {
    uint8_t  ring[kRingBytes + kRingMirror];
    uint16_t pos[kRows * kRowEnt];
    uint8_t  tag[kRows * kRowEnt];
    uint32_t cnt[kRows / 4];            // one round-robin byte counter per row
    uint16_t exitmap[kRowBatch];        // batch-relative: where the path through position i leaves i's region
    uint32_t hist_l[288], hist_d[32];   // tokens of this segment so far (what K2 builds its codes from)
    uint16_t cost_l[288], cost_d[32];   // current estimate, units of 1/16 bit, without extra bits
    uint32_t tot_l, tot_d;
    uint32_t cover;                     // absolute: first position not yet produced (carried across batches)
#ifdef ZR_ROWS_STAMPS
    unsigned long long stamps[kRowWaves][10];
#endif
};
static_assert(sizeof(RowShared) <= 160 * 1024, "RowShared must fit the CU's LDS");

typedef uint32_t u32_lds_unaligned __attribute__((aligned(1)));

// Workgroup barrier that orders LDS traffic only: nothing the workgroup exchanges goes through global memory, and
// __syncthreads() would also wait for the plaintext prefetch and the token stores in flight (vmcnt).
__device__ __forceinline__ void rows_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ uint32_t ring_u32(const uint8_t *ring, uint32_t idx) {      // idx < kRingBytes + mirror - 3
    return *reinterpret_cast<const u32_lds_unaligned *>(ring + idx);
}

__device__ __forceinline__ void row_key(uint32_t val, uint32_t &row, uint32_t &tag) {
    row = __umulhi(val * 2654435761u, (uint32_t)kRows);        // insert_string.c:11-13's multiplier, top bits
    tag = (val * 0x85EBCA6Bu) >> 24;
}

// common prefix of the strings at ring indices pi and ci, counted from `from` (bytes below are known equal), up to maxlen
__device__ __forceinline__ uint32_t ring_common_prefix(const uint8_t *ring, uint32_t pi, uint32_t ci, uint32_t from,
                                                       uint32_t maxlen) {
    uint32_t l = from;
    while (l + 16u <= maxlen) {
        uint32_t x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = ring_u32(ring, pi + l + 4u * j) ^ ring_u32(ring, ci + l + 4u * j);
        uint32_t add = 16;
#pragma unroll
        for (int j = 3; j >= 0; --j)
            if (x[j]) add = 4u * (uint32_t)j + ((uint32_t)(__ffs((int)x[j]) - 1) >> 3);
        l += add;
        if (add < 16u) return l;
    }
    while (l < maxlen && ring[pi + l] == ring[ci + l]) ++l;
    return l;
}

// RFC 1951 3.2.5: length / distance -> symbol and number of extra bits
__device__ __forceinline__ void rows_len_symbol(uint32_t len, uint32_t &sym, uint32_t &eb) {
    const uint32_t l = len - 3u;
    eb = 0;
    if (l < 8u) sym = 257u + l;
    else if (l == 255u) sym = 285u;
    else {
        const uint32_t lg = 31u - (uint32_t)__clz((int)l);
        eb = lg - 2u;
        sym = 257u + 4u * eb + 4u + ((l >> eb) & 3u);
    }
}
__device__ __forceinline__ void rows_dist_symbol(uint32_t dist, uint32_t &sym, uint32_t &eb) {
    const uint32_t x = dist - 1u;
    eb = 0;
    if (x < 4u) sym = x;
    else {
        const uint32_t lg = 31u - (uint32_t)__clz((int)x);
        eb = lg - 1u;
        sym = 2u * lg + ((x >> eb) & 1u);
    }
}

// minimum over the wavefront, wave-uniform result (DPP: four steps inside the 16-lane rows, two across them)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)kCostInf, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)kCostInf, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)kCostInf, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)kCostInf, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)kCostInf, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)kCostInf, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// lane j <- lane j - 1, lane 0 <- `first` (wave-uniform)
__device__ __forceinline__ uint32_t wave_shift_up1(uint32_t v, uint32_t first) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xf, 0xf, false);          // wave_shr:1
}

// The waves enter their positions in position order (one turn per wave): the slot a position gets, and with it which older
// entry it replaces, is then a function of the input alone.
__device__ __forceinline__ void rows_insert(RowShared *sh, bool can, uint32_t row, uint32_t tag, uint32_t p, int wave) {
    for (int w = 0; w < kRowWaves; ++w) {
        if (wave == w && can) {
            const uint32_t sft = 8u * (row & 3u);
            const uint32_t old = atomicAdd(&sh->cnt[row >> 2], 1u << sft);     // a carry into the neighbour's counter only
            const uint32_t slot = (old >> sft) & (uint32_t)(kRowEnt - 1);      // makes that row skip a slot
            sh->pos[row * kRowEnt + slot] = (uint16_t)p;
            sh->tag[row * kRowEnt + slot] = (uint8_t)tag;
        }
        __syncthreads();
    }
}

// symbol costs from the histogram of the segment's tokens so far: 16 * log2(total / count), clamped to 1..15 bits
__device__ __forceinline__ void rows_refresh_costs(RowShared *sh, int t) {
    if (t < 288) {
        const float c = 16.0f * (__log2f((float)sh->tot_l + 16.0f) - __log2f((float)sh->hist_l[t] + 0.5f));
        const uint32_t u = (uint32_t)(c < (float)kCostBit ? (float)kCostBit : c);
        sh->cost_l[t] = (uint16_t)(u > kCostMax ? kCostMax : u);
    } else if (t < 320) {
        const float c = 16.0f * (__log2f((float)sh->tot_d + 4.0f) - __log2f((float)sh->hist_d[t - 288] + 0.5f));
        const uint32_t u = (uint32_t)(c < (float)kCostBit ? (float)kCostBit : c);
        sh->cost_d[t - 288] = (uint16_t)(u > kCostMax ? kCostMax : u);
    }
}

struct RowsToken {
    uint32_t kind;        // 0 = no token starts here, 1 = literal, 2 = match
    uint32_t len, dist;
};

// entries of one row whose tag is `tag` and whose distance from p lies in [1, dhi]: bit e of the result; d[k] gets the
// two 16-bit distances of entries 2k and 2k + 1
__device__ __forceinline__ uint32_t rows_candidates(const uint2 T, const uint4 Pz, uint32_t tag, uint32_t p, uint32_t dhi,
                                                    uint32_t (&d)[4]) {
    const uint32_t pp = (p & 0xffffu) * 0x10001u;
    const uint32_t pw[4] = {Pz.x, Pz.y, Pz.z, Pz.w};
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t lo = (pp - pw[k]) & 0xffffu, hi = ((pp >> 16) - (pw[k] >> 16)) & 0xffffu;
        d[k] = lo | (hi << 16);
        const uint32_t tw = k < 2 ? T.x : T.y;
        const uint32_t t0 = (tw >> (16 * (k & 1))) & 0xffu, t1 = (tw >> (16 * (k & 1) + 8)) & 0xffu;
        if (t0 == tag && lo - 1u < dhi) m |= 1u << (2 * k);
        if (t1 == tag && hi - 1u < dhi) m |= 1u << (2 * k + 1);
    }
    return m;
}

// One step of the backward recurrence at region position I (compile-time, so that every lane select is an immediate):
//   cost[I] = min(literal + cost[I + 1], min over l in 4..min(len, 64) of match(l) + cost[I + l], the full length if > 64)
// W: lane j holds cost[I + 1 + j] << 6; LC: lane j holds the cost of length j + 1, shifted, with j in its low six bits, so
// the minimum carries the chosen length; the literal rides along as candidate lane 0.
template <int I>
__device__ __forceinline__ void rows_dp_step(uint32_t L, uint32_t LIT6, uint32_t DC6, uint32_t FULL, uint32_t LC, int lane,
                                             uint32_t lim, uint32_t &W, uint32_t &CH, uint32_t &cnext6) {
    const uint32_t sL = (uint32_t)__builtin_amdgcn_readlane((int)L, I);
    const uint32_t sLit6 = (uint32_t)__builtin_amdgcn_readlane((int)LIT6, I);
    const uint32_t sDc6 = (uint32_t)__builtin_amdgcn_readlane((int)DC6, I);
    const uint32_t top = sL < 64u ? sL : 64u;
    uint32_t v = (uint32_t)lane < top ? W + LC + sDc6 : kCostInf;       // lanes 0..2 of LC are out of reach
    v = lane == 0 ? sLit6 + cnext6 : v;
    uint32_t m = wave_min_u32(v);
    uint32_t bl = (m & 63u) + 1u;
    m &= ~63u;
    if (sL > 64u) {                                       // the full length reaches beyond the region: a credit per byte
        const uint32_t cf6 = (((uint32_t)__builtin_amdgcn_readlane((int)FULL, I) + kCostBias -
                               kCostBeta * ((uint32_t)I + sL - lim)) << 6) + sDc6;
        if (cf6 < m) {
            m = cf6;
            bl = sL;
        }
    }
    cnext6 = m;
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(CH) : "s"(bl), "n"(I));
    W = wave_shift_up1(W, m);
}

// ---- one batch, in three pieces so that the kernel can software-pipeline them ----------------------------------------------
// The candidate compares of a batch are bound by LDS throughput (unaligned 16-byte reads at random addresses), its parse by
// vector-instruction issue, and both are per-wave work without a barrier between them.  Run one after the other in every
// wave (round 3's first form) each unit idles through the other's phase; so the kernel keeps TWO batches in flight --
// compares of batch k beside the parse of batch k - 1 -- and half of the waves take them in the opposite order.
//   rows_front    row image in front of the batch, barrier, ordered insert (16 turns), row image with the batch in it
//   rows_compare  longest match at this lane's position                                  (LDS-bound)
//   rows_parse    shortest-path parse of this wave's region of the PREVIOUS batch         (VALU-bound)
//   rows_finish   exit map, barrier, stitch, path: the previous batch's tokens
struct RowsFront {                      // what rows_front hands to rows_compare
    uint2    TA, TC;
    uint4    PA, PC;
    uint32_t mine[4];
    uint32_t before, val, tag;
    bool     can;
};
struct RowsMatch {                      // per lane: what a batch carries from its compares to its parse
    uint32_t L, dist, val;
};

// The ring holds every byte of [max(P0, P - 32768), P + kRingAhead) that lies below n, P0 = the first position that
// was ever loaded.  `refresh`: recompute the symbol costs from the histogram behind the first barrier.  *cover_in = what the
// previous call's rows_finish left in sh->cover (read at a point no writer can reach).
__device__ __forceinline__ void rows_front(uint32_t n, uint32_t P, RowShared *sh, int t, bool refresh, RowsFront &f,
                                           uint32_t *cover_in) {
    const uint8_t *ring = sh->ring;
    [[maybe_unused]] const int lane = t & 63;            // (the stamps of the diagnostic build use it)
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t p = P + (uint32_t)t;
    const uint32_t pi = p & (kRingBytes - 1u);
    f.can = p + kLzMinMatch <= n;                        // lookahead >= WANT_MIN_MATCH, deflate_quick.c:88
    f.val = ring_u32(ring, pi);
    uint32_t row;
    row_key(f.val, row, f.tag);
    ZR_STAMP(0);
    // A. the row as it is in front of this batch, this lane's own 16 bytes, the bytes in front of them
    f.TA = *reinterpret_cast<const uint2 *>(&sh->tag[row * kRowEnt]);
    f.PA = *reinterpret_cast<const uint4 *>(&sh->pos[row * kRowEnt]);
#pragma unroll
    for (int j = 0; j < 4; ++j) f.mine[j] = ring_u32(ring, pi + 4u * j);
    f.before = ring_u32(ring, (pi - 1u) & (kRingBytes - 1u));
    rows_barrier();
    *cover_in = sh->cover;                               // written by the previous rows_finish in front of this barrier
    if (refresh) rows_refresh_costs(sh, t);              // its readers come behind the insert turns' barriers
    ZR_STAMP(1);
    // B. insert, waves in position order
    rows_insert(sh, f.can, row, f.tag, p, wave);
    ZR_STAMP(2);
    // C. the row with the batch in it: only entries inside the batch and below p are news
    f.TC = *reinterpret_cast<const uint2 *>(&sh->tag[row * kRowEnt]);
    f.PC = *reinterpret_cast<const uint4 *>(&sh->pos[row * kRowEnt]);
}

__device__ __forceinline__ RowsMatch rows_compare(uint32_t n, uint32_t P, uint32_t P0, const RowShared *sh, int t,
                                                  uint32_t max_cand, const RowsFront &f) {
    const uint8_t *ring = sh->ring;
    const uint32_t p = P + (uint32_t)t;
    const uint32_t pi = p & (kRingBytes - 1u);
    const uint32_t maxlen = p < n ? ((n - p) < kStdMaxMatch ? (n - p) : kStdMaxMatch) : 0u;
    uint32_t best = 3, dist = 0;                         // a match must reach WANT_MIN_MATCH to count
    if (f.can) {
        const uint32_t back = p - P0;                    // bytes of history the ring really holds
        const uint32_t dmax = back < kLzMaxDist ? back : kLzMaxDist;
        uint32_t dC[4], dA[4];
        const uint32_t tC = (uint32_t)t < dmax ? (uint32_t)t : dmax;
        uint32_t mC = rows_candidates(f.TC, f.PC, f.tag, p, tC, dC);
        uint32_t mA = rows_candidates(f.TA, f.PA, f.tag, p, dmax, dA);
        uint32_t tail_off = 0, want_tail = f.mine[0];    // the 4 bytes that end at `best` must agree (match_tpl.h:141-173)
        bool done = false;
        uint32_t budget = max_cand;

        auto consider = [&](uint32_t d) __attribute__((always_inline)) {
            const uint32_t ci = (pi - d) & (kRingBytes - 1u);
            const uint32_t tail = ring_u32(ring, ci + tail_off);
            uint32_t cw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) cw[j] = ring_u32(ring, ci + 4u * j);
            if (tail != want_tail) return;
            uint32_t l = 16;
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                const uint32_t x = cw[j] ^ f.mine[j];
                if (x) l = 4u * (uint32_t)j + ((uint32_t)(__ffs((int)x) - 1) >> 3);
            }
            if (l >= 16u && maxlen > 16u) {
                const uint32_t cap = maxlen < kNiceLen ? maxlen : kNiceLen;
                l = ring_common_prefix(ring, pi, ci, 16, cap);
            }
            l = l < maxlen ? l : maxlen;
            if (l > best || (l == best && l >= kLzMinMatch && d < dist)) {
                best = l;
                dist = d;
                if (l >= kNiceLen || l >= maxlen) done = true;
                else {
                    tail_off = best - 3u;
                    want_tail = ring_u32(ring, pi + tail_off);
                }
            }
        };

        // a run (distance 1) first: every position of a run lands in ONE row, which keeps only the last few of them
        if (p > P0 && f.before == f.val) consider(1u);
        while (mC && !done && budget) {
            const uint32_t e = (uint32_t)__ffs((int)mC) - 1u;
            mC &= mC - 1u;
            --budget;
            const uint32_t w01 = (e & 2u) ? dC[1] : dC[0], w23 = (e & 2u) ? dC[3] : dC[2];
            const uint32_t w = (e & 4u) ? w23 : w01;
            consider((e & 1u) ? (w >> 16) : (w & 0xffffu));
        }
        while (mA && !done && budget) {
            const uint32_t e = (uint32_t)__ffs((int)mA) - 1u;
            mA &= mA - 1u;
            --budget;
            const uint32_t w01 = (e & 2u) ? dA[1] : dA[0], w23 = (e & 2u) ? dA[3] : dA[2];
            const uint32_t w = (e & 4u) ? w23 : w01;
            consider((e & 1u) ? (w >> 16) : (w & 0xffffu));
        }
        if (best >= kNiceLen && best < maxlen)           // the probe saturated: measure the rest
            best = ring_common_prefix(ring, pi, (pi - dist) & (kRingBytes - 1u), best, maxlen);
    }
    RowsMatch m;
    m.L = best >= kLzMinMatch ? best : 0u;
    m.dist = dist;
    m.val = f.val;
    return m;
}

// shortest-path parse of this wave's region [w0, w0 + lim) of the batch at P: returns CH (lane i: length of the token the
// best path takes at region position i)
__device__ __forceinline__ uint32_t rows_parse(uint32_t n, uint32_t P, const RowShared *sh, int t, const RowsMatch &m) {
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t w0 = P + 64u * (uint32_t)wave;
    const uint32_t lim = w0 >= n ? 0u : ((n - w0) < 64u ? (n - w0) : 64u);        // wave-uniform
    const uint32_t L = m.L;
    uint32_t LIT = sh->cost_l[m.val & 0xffu], DC = 0, FULL = 0;
    if (L) {
        uint32_t sy, eb;
        rows_len_symbol(L, sy, eb);
        FULL = sh->cost_l[sy] + kCostBit * eb;
        rows_dist_symbol(m.dist, sy, eb);
        DC = sh->cost_d[sy] + kCostBit * eb;
    }
    uint32_t LC;                                         // lane j: cost of a match of length j + 1, lane id in the low bits
    {
        uint32_t sy, eb;
        rows_len_symbol((uint32_t)lane + 1u < 3u ? 3u : (uint32_t)lane + 1u, sy, eb);
        LC = (uint32_t)lane < 3u ? 0x7fffffffu : (((sh->cost_l[sy] + kCostBit * eb) << 6) | (uint32_t)lane);
    }
    uint32_t CH = 1u;
    if (lim) {
        // positions at or beyond lim (the segment's end inside this region) cost what the credit gives: the recurrence then
        // needs no special case for a short region and runs its 64 steps unrolled, every lane index an immediate
        const uint32_t LIT6 = ((uint32_t)lane < lim ? LIT : kCostBeta) << 6, DC6 = DC << 6;
        const uint32_t Lr = (uint32_t)lane < lim ? L : 0u;
        uint32_t W = (kCostBias - kCostBeta * (64u + (uint32_t)lane - lim)) << 6;      // lane j: cost[64 + j] << 6
        uint32_t cnext6 = (kCostBias - kCostBeta * (64u - lim)) << 6;
#define ZR_DP1(I) rows_dp_step<I>(Lr, LIT6, DC6, FULL, LC, lane, lim, W, CH, cnext6);
#define ZR_DP8(I) ZR_DP1(I + 7) ZR_DP1(I + 6) ZR_DP1(I + 5) ZR_DP1(I + 4) ZR_DP1(I + 3) ZR_DP1(I + 2) ZR_DP1(I + 1) ZR_DP1(I)
        ZR_DP8(56) ZR_DP8(48) ZR_DP8(40) ZR_DP8(32) ZR_DP8(24) ZR_DP8(16) ZR_DP8(8) ZR_DP8(0)
#undef ZR_DP8
#undef ZR_DP1
    }
    return CH;
}

// the tokens of the batch at P from its parse: exit maps, one barrier, stitch from `cover_in`, this wave's path.  Leaves the
// next batch's cover in sh->cover.  Returns this lane's token; the region's token-start mask goes to *starts.
__device__ __forceinline__ RowsToken rows_finish(uint32_t n, uint32_t P, RowShared *sh, int t, uint32_t CH, uint32_t dist,
                                                 uint32_t cover_in, unsigned long long *starts) {
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t w0 = P + 64u * (uint32_t)wave;
    const uint32_t lim = w0 >= n ? 0u : ((n - w0) < 64u ? (n - w0) : 64u);
    ZR_STAMP(4);
    // exit map: where does the path through position i leave the region?  pointer doubling over next[i] = i + CH[i]
    {
        uint32_t J = (uint32_t)lane + CH;
        if ((uint32_t)lane >= lim) J = 64u + 258u;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const uint32_t Jn = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((J & 63u) << 2), (int)J);
            J = J < lim ? Jn : J;
        }
        sh->exitmap[t] = (uint16_t)(64u * (uint32_t)wave + J);
    }
    rows_barrier();
    ZR_STAMP(5);
    // stitch: follow the exits of the regions in front of this one from where the batch before ended
    const uint32_t nrel = n - P;                          // n > P for every batch that runs
    uint32_t s = cover_in - P;
    for (int r = 0; r < wave; ++r) {
        if (s < 64u * (uint32_t)(r + 1) && s < nrel)
            s = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->exitmap[s]);
    }
    ZR_STAMP(6);
    unsigned long long mask = 0;
    {
        uint32_t q = s - 64u * (uint32_t)wave;            // s >= 64 * wave by construction
        while (q < lim) {
            mask |= 1ull << q;
            q += (uint32_t)__builtin_amdgcn_readlane((int)CH, (int)q);
        }
        // where the batch's path ends = the next batch's cover (read there behind a rows_front barrier)
        if (wave == kRowWaves - 1 && lane == 0)
            sh->cover = P + (s < 64u * (uint32_t)(wave + 1) && s < nrel ? 64u * (uint32_t)wave + q : s);
    }
    ZR_STAMP(7);
    *starts = mask;
    RowsToken r;
    r.kind = 0;
    r.len = CH;
    r.dist = dist;
    if ((mask >> lane) & 1ull) r.kind = CH > 1u ? 2u : 1u;
    return r;
}

}  // namespace zr
