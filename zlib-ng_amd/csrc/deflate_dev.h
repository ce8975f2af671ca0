// deflate_dev.h -- wave64 device building blocks for zlib-ng's deflate-side functable
// primitives.  One wavefront works on one stream / one query; scalars (positions, lengths,
// chain state) are wave-uniform, the 64 lanes are used for the data-parallel parts
// (256-byte compares = 64 lanes x 4 bytes, 64 consecutive hash insertions).
//
// Reference semantics restated here (zlib-ng 2.2.2):
//   compare256        arch/generic/compare256_c.c:12-47
//   update_hash etc.  insert_string.c:11-19, insert_string_tpl.h:48-104
//   longest_match     match_tpl.h:26-280 (non-SLOW instantiation)
//   slide_hash        arch/generic/slide_hash_c.c:15-52
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/zng_rocm.h"

namespace zr {

constexpr uint32_t kHashSize = 65536u;        // deflate.h:81-85
constexpr uint32_t kStdMinMatch = 3, kStdMaxMatch = 258;
constexpr uint32_t kMinLookahead = kStdMaxMatch + kStdMinMatch + 1;   // deflate.h:405

typedef uint32_t u32_unaligned __attribute__((aligned(1)));
typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
#define ZR_GLOBAL __attribute__((address_space(1)))

// Window / input bytes always live in HBM: say so (pointers that come out of a descriptor struct are generic
// to the compiler, which would otherwise emit flat_load_* and tie up lgkmcnt as well as vmcnt).
__device__ __forceinline__ uint32_t load_u32(const uint8_t *p) {
    return *(const ZR_GLOBAL u32_unaligned *)(p);
}
__device__ __forceinline__ u32x4_unaligned load_u128(const uint8_t *p) {
    return *(const ZR_GLOBAL u32x4_unaligned *)(p);
}
__device__ __forceinline__ uint8_t load_u8(const uint8_t *p) {
    return *(const ZR_GLOBAL uint8_t *)(p);
}

// insert_string.c:11-13 + insert_string_tpl.h:48-51 (HASH_SLIDE 16, HASH_MASK 0xffff)
__device__ __forceinline__ uint32_t hash_calc(uint32_t val) {
    return ((val * 2654435761u) >> 16) & (kHashSize - 1u);
}

// insert_string_roll.c:10-24 (level 9): the key is ((h << 5) ^ byte) & 32767 over consecutive bytes, so once three
// bytes went through it only those three are left in it
constexpr uint32_t kRollMask = 32767u;
__device__ __forceinline__ uint32_t hash_roll(uint32_t h, uint32_t byte) {
    return ((h << 5) ^ (byte & 0xffu)) & kRollMask;
}
__device__ __forceinline__ uint32_t hash_roll3(uint32_t b0, uint32_t b1, uint32_t b2) {
    return ((b0 << 10) ^ (b1 << 5) ^ b2) & kRollMask;
}

// compare256: first differing byte index of two 256-byte strings (256 if equal).
// All 64 lanes participate; result is wave-uniform.
__device__ __forceinline__ uint32_t compare256_wave(const uint8_t *a, const uint8_t *b, int lane) {
    const uint32_t x = load_u32(a + 4 * lane);
    const uint32_t y = load_u32(b + 4 * lane);
    const unsigned long long diff = __ballot(x != y);
    if (diff == 0) return 256u;
    const int first = __ffsll((long long)diff) - 1;
    const uint32_t d = (uint32_t)__shfl((int)(x ^ y), first, 64);
    return (uint32_t)first * 4u + ((uint32_t)(__ffs((int)d) - 1) >> 3);
}

// insert_string_tpl.h:85-104 for `count` consecutive positions, 64 per pass.
// Sequential semantics are kept exactly: a position whose hash equals that of an earlier
// position of the same pass sees that position as the chain head.
// ROLL = the rolling variant (insert_string_roll.c): position str+i is keyed by the bytes at str+i .. str+i+2, except
// that the first two positions still carry bits of the incoming s->ins_h; returns the key of the last position
// (the new s->ins_h), or ins_h itself when count is 0.
template <bool ROLL = false>
__device__ __forceinline__ uint32_t insert_string_wave(const uint8_t *window, uint16_t *head, uint16_t *prev,
                                                       uint32_t w_mask, uint32_t str, uint32_t count, int lane,
                                                       uint32_t ins_h = 0) {
    uint32_t last_key = ins_h;
    for (uint32_t base = 0; base < count; base += 64) {
        const uint32_t i = base + (uint32_t)lane;
        const bool live = i < count;
        const uint32_t pos = str + i;
        const uint16_t idx = (uint16_t)pos;                   // `Pos idx` wraps at 16 bits
        uint32_t h = 0xffffffffu;
        if (live) {
            const uint32_t four = load_u32(window + pos);
            if (ROLL) {
                const uint32_t b0 = four & 0xffu, b1 = (four >> 8) & 0xffu, b2 = (four >> 16) & 0xffu;
                if (i == 0) h = hash_roll(ins_h, b2);
                else if (i == 1) h = hash_roll(hash_roll(ins_h, b1), b2);
                else h = hash_roll3(b0, b1, b2);
            } else {
                h = hash_calc(four);
            }
        }
        if (ROLL) {
            const uint32_t tail = count - 1u - base;          // lane of the last position, if in this pass
            if (tail < 64u) last_key = (uint32_t)__shfl((int)h, (int)tail, 64);
        }
        // nearest earlier lane with the same hash, and whether a later one exists
        int before = -1;
        bool later = false;
        for (int j = 0; j < 64; ++j) {
            const uint32_t hj = (uint32_t)__shfl((int)h, j, 64);
            const bool same = live && hj == h;
            if (same && j < lane) before = j;
            if (same && j > lane) later = true;
        }
        const uint16_t idx_before = (uint16_t)__shfl((int)idx, before < 0 ? 0 : before, 64);
        if (live) {
            const uint16_t old = before >= 0 ? idx_before : head[h];
            if (old != idx) {
                prev[idx & w_mask] = old;
                if (!later) head[h] = idx;
            }
        }
        // a later pass must see this pass's stores
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0);
    }
    return last_key;
}

struct MatchParams {           // the deflate_state fields match_tpl.h reads (deflate.h:164-244)
    const uint8_t  *window;
    const uint16_t *prev;
    uint32_t w_size, w_mask, strstart, lookahead, prev_length, max_chain_length, good_match, nice_match;
    int32_t  level;
};

// match_tpl.h:26-280 (non-SLOW).  Returns the match length; *match_start is written when a
// longer match is found (match_tpl.h:178-179).  Wave-uniform control flow; the byte-pair probes
// are the OPTIMAL_CMP < 32 form (:167-173) -- for level >= 5 every form accepts the same
// candidates (the probes only skip candidates that cannot beat best_len); for level < 5 the
// early exit (:261-266) is evaluated with these probes, i.e. it follows longest_match_c.
__device__ __forceinline__ uint32_t longest_match_wave(const MatchParams &s, uint32_t cur_match,
                                                       uint32_t *match_start, int lane) {
    const uint8_t *window = s.window;
    const uint8_t *scan = window + s.strstart;
    uint32_t best_len = s.prev_length ? s.prev_length : kStdMinMatch - 1;
    uint32_t offset = best_len - 1;
    uint32_t chain_length = s.max_chain_length;
    if (best_len >= s.good_match) chain_length >>= 2;
    const uint32_t max_dist = s.w_size - kMinLookahead;
    const uint32_t limit = s.strstart > max_dist ? (uint16_t)(s.strstart - max_dist) : 0;
    const bool early_exit = s.level < 5;
    const uint8_t sc0 = scan[0], sc1 = scan[1];
    uint8_t end0 = scan[offset], end1 = scan[offset + 1];

    for (;;) {
        if (cur_match >= s.strstart) break;
        // skip candidates that cannot improve on best_len
        for (;;) {
            const uint8_t *cand = window + cur_match;
            if (cand[offset] == end0 && cand[offset + 1] == end1 && cand[0] == sc0 && cand[1] == sc1) break;
            if (--chain_length && (cur_match = s.prev[cur_match & s.w_mask]) > limit) continue;
            return best_len;
        }
        const uint32_t len = compare256_wave(scan + 2, window + cur_match + 2, lane) + 2;
        if (len > best_len) {
            *match_start = cur_match;
            if (len > s.lookahead) return s.lookahead;
            best_len = len;
            if (best_len >= s.nice_match) return best_len;
            offset = best_len - 1;
            end0 = scan[offset];
            end1 = scan[offset + 1];
        } else if (early_exit) {
            break;
        }
        if (--chain_length && (cur_match = s.prev[cur_match & s.w_mask]) > limit) continue;
        return best_len;
    }
    return best_len;
}

// match_tpl.h:26-280 with LONGEST_MATCH_SLOW (the functable slot `longest_match_slow`, levels 7-9): besides the
// chain walk it re-anchors the search on the most distant chain among the bytes of the current best match
// (:94-125, :208-256), using head[] lookups of s->update_hash fed byte by byte exactly as the template does:
// with the multiplicative hash (levels 7-8, insert_string.c:11-13) only the last byte counts, with the rolling
// hash (ROLL, level 9, insert_string_roll.c) the last three.  Wave-uniform control flow, wave-wide compare256.
template <bool ROLL>
__device__ __forceinline__ uint32_t longest_match_slow_wave(const MatchParams &s, const uint16_t *head,
                                                            uint32_t cur_match, uint32_t *match_start_out, int lane) {
    const uint8_t *window = s.window;
    const uint8_t *scan = window + s.strstart;
    uint32_t match_offset = 0;
    uint32_t best_len = s.prev_length ? s.prev_length : kStdMinMatch - 1;
    uint32_t offset = best_len - 1;
    uint32_t chain_length = s.max_chain_length;
    if (best_len >= s.good_match) chain_length >>= 2;
    const uint32_t max_dist = s.w_size - kMinLookahead;
    const uint32_t limit_base = s.strstart > max_dist ? (uint16_t)(s.strstart - max_dist) : 0;
    uint32_t limit = limit_base;
    const uint8_t sc0 = scan[0], sc1 = scan[1];
    uint8_t end0 = scan[offset], end1 = scan[offset + 1];
    auto give_up = [&]() { return best_len < s.lookahead ? best_len : s.lookahead; };      // break_matching :272-278

    if (best_len >= kStdMinMatch) {
        for (uint32_t i = 3; i <= best_len; ++i) {
            const uint32_t pos = head[ROLL ? hash_roll3(scan[i - 2], scan[i - 1], scan[i]) : hash_calc(scan[i])];
            if (pos < cur_match) {
                match_offset = i - 2;
                cur_match = pos;
            }
        }
        limit = (uint16_t)(limit_base + match_offset);
        if (cur_match <= limit) return give_up();
    }
    for (;;) {
        if (cur_match >= s.strstart) break;
        for (;;) {
            const uint8_t *cand = window + cur_match - match_offset;
            if (cand[offset] == end0 && cand[offset + 1] == end1 && cand[0] == sc0 && cand[1] == sc1) break;
            if (--chain_length && (cur_match = s.prev[cur_match & s.w_mask]) > limit) continue;
            return best_len;
        }
        const uint32_t len = compare256_wave(scan + 2, window + cur_match - match_offset + 2, lane) + 2;
        if (len > best_len) {
            const uint32_t match_start = cur_match - match_offset;
            *match_start_out = match_start;
            if (len > s.lookahead) return s.lookahead;
            best_len = len;
            if (best_len >= s.nice_match) return best_len;
            offset = best_len - 1;
            end0 = scan[offset];
            end1 = scan[offset + 1];
            if (len > kStdMinMatch && match_start + len < s.strstart) {
                cur_match = (uint16_t)(cur_match - match_offset);
                match_offset = 0;
                uint32_t next_pos = cur_match;
                for (uint32_t i = 0; i <= len - kStdMinMatch; ++i) {
                    const uint32_t pos = s.prev[(cur_match + i) & s.w_mask];
                    if (pos < next_pos) {
                        if (pos <= limit_base + i) return give_up();
                        next_pos = pos;
                        match_offset = i;
                    }
                }
                cur_match = next_pos;
                const uint8_t *endstr = scan + len - (kStdMinMatch + 1);
                const uint32_t pos = head[ROLL ? hash_roll3(endstr[0], endstr[1], endstr[2]) : hash_calc(endstr[2])];
                if (pos < cur_match) {
                    match_offset = len - (kStdMinMatch + 1);
                    if (pos <= limit_base + match_offset) return give_up();
                    cur_match = pos;
                }
                limit = (uint16_t)(limit_base + match_offset);
                continue;
            }
        }
        if (--chain_length && (cur_match = s.prev[cur_match & s.w_mask]) > limit) continue;
        return best_len;
    }
    return best_len;
}

}  // namespace zr
