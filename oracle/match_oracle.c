/* match_oracle.c -- CPU restatement of zlib-ng's deflate-side functable
 * primitives: slide_hash, compare256, the insert_string hash family and
 * longest_match.  TEST INFRASTRUCTURE ONLY (see zng_oracle.h).
 *
 * Follows /root/reference (zlib-ng 2.2.2):
 *   arch/generic/slide_hash_c.c:15-52    slide_hash_c          -> oracle_slide_hash
 *   arch/generic/compare256_c.c:12-47    compare256_c          -> oracle_compare256
 *   insert_string.c:11-19 + insert_string_tpl.h:48-104
 *                                        update_hash / quick_insert_string / insert_string
 *   match_tpl.h:26-280 (non-SLOW, OPTIMAL_CMP < 32 instantiation = longest_match_c,
 *                       arch/generic/compare256_c.c:49-52)    -> oracle_longest_match
 */
#include <string.h>
#include "zng_oracle.h"

/* slide_hash_c.c:15-44: every Pos moves down by wsize, saturating at 0. */
static void slide_table(oracle_pos *tab, uint32_t entries, uint16_t wsize) {
    for (uint32_t i = 0; i < entries; i++) {
        oracle_pos m = tab[i];
        tab[i] = (oracle_pos)(m >= wsize ? m - wsize : 0);
    }
}

void oracle_slide_hash(oracle_deflate_state *s) {
    uint16_t wsize = (uint16_t)s->w_size;                   /* slide_hash_c.c:48 */
    slide_table(s->head, ORACLE_HASH_SIZE, wsize);          /* :50 */
    slide_table(s->prev, wsize, wsize);                     /* :51 */
}

/* compare256_c.c:12-43: index of the first differing byte, capped at 256. */
uint32_t oracle_compare256(const uint8_t *src0, const uint8_t *src1) {
    uint32_t n = 0;
    while (n < 256 && src0[n] == src1[n])
        n++;
    return n;
}

/* insert_string.c:11-13 HASH_CALC with HASH_SLIDE 16; insert_string_tpl.h:48-51
 * masks with HASH_MASK.  `h` is unused by the multiplicative hash. */
uint32_t oracle_update_hash(uint32_t h, uint32_t val) {
    (void)h;
    return ((val * 2654435761u) >> 16) & (ORACLE_HASH_SIZE - 1u);
}

static inline uint32_t load_le32(const uint8_t *p) {        /* insert_string_tpl.h:30-38 */
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* insert_string_tpl.h:58-75 */
oracle_pos oracle_quick_insert_string(oracle_deflate_state *s, uint32_t str) {
    uint32_t slot = oracle_update_hash(0, load_le32(s->window + str));
    oracle_pos head = s->head[slot];
    if (head != str) {
        s->prev[str & s->w_mask] = head;
        s->head[slot] = (oracle_pos)str;
    }
    return head;
}

/* insert_string_tpl.h:85-104: positions str .. str+count-1, strictly in order
 * (a later position must see the head an earlier one just wrote).  The index
 * is a Pos, i.e. it wraps at 16 bits exactly like the reference's `Pos idx`. */
void oracle_insert_string(oracle_deflate_state *s, uint32_t str, uint32_t count) {
    const uint8_t *p = s->window + str;
    oracle_pos idx = (oracle_pos)str;
    for (uint32_t i = 0; i < count; i++, idx++, p++) {
        uint32_t slot = oracle_update_hash(0, load_le32(p));
        oracle_pos head = s->head[slot];
        if (head != idx) {
            s->prev[idx & s->w_mask] = head;
            s->head[slot] = idx;
        }
    }
}

/* insert_string_roll.c:10-24 instantiating insert_string_tpl.h: HASH_SLIDE 5, one byte read at
 * str + STD_MIN_MATCH-1, mask 32767, and the key lives in s->ins_h between calls. */
uint32_t oracle_update_hash_roll(uint32_t h, uint32_t val) {
    return ((h << 5) ^ (uint8_t)val) & 32767u;
}

oracle_pos oracle_quick_insert_string_roll(oracle_deflate_state *s, uint32_t str) {
    s->ins_h = oracle_update_hash_roll(s->ins_h, s->window[str + ORACLE_STD_MIN_MATCH - 1]);
    const uint32_t slot = s->ins_h;
    oracle_pos head = s->head[slot];
    if (head != str) {
        s->prev[str & s->w_mask] = head;
        s->head[slot] = (oracle_pos)str;
    }
    return head;
}

void oracle_insert_string_roll(oracle_deflate_state *s, uint32_t str, uint32_t count) {
    const uint8_t *p = s->window + str + ORACLE_STD_MIN_MATCH - 1;
    oracle_pos idx = (oracle_pos)str;
    for (uint32_t i = 0; i < count; i++, idx++, p++) {
        s->ins_h = oracle_update_hash_roll(s->ins_h, *p);
        const uint32_t slot = s->ins_h;
        oracle_pos head = s->head[slot];
        if (head != idx) {
            s->prev[idx & s->w_mask] = head;
            s->head[slot] = idx;
        }
    }
}

/* match_tpl.h:26-280, non-SLOW, byte-pair probes (OPTIMAL_CMP < 32). */
uint32_t oracle_longest_match(oracle_deflate_state *s, oracle_pos cur_match) {
    const uint32_t strstart = s->strstart;
    const uint32_t wmask = s->w_mask;
    const uint8_t *window = s->window;
    const uint8_t *scan = window + strstart;
    const oracle_pos *prev = s->prev;
    const uint32_t lookahead = s->lookahead;

    /* :59 */
    uint32_t best_len = s->prev_length ? s->prev_length : ORACLE_STD_MIN_MATCH - 1;
    /* :64 (the 4/8-byte offset tweaks at :65-73 only exist for OPTIMAL_CMP >= 32) */
    uint32_t offset = best_len - 1;
    uint8_t end0 = scan[offset], end1 = scan[offset + 1];   /* :81-84 */

    /* :88-91 */
    uint32_t chain_length = s->max_chain_length;
    if (best_len >= s->good_match)
        chain_length >>= 2;
    const uint32_t nice_match = (uint32_t)s->nice_match;

    /* :96  MAX_DIST = w_size - MIN_LOOKAHEAD (deflate.h:410-415) */
    const uint32_t max_dist = s->w_size - ORACLE_MIN_LOOKAHEAD;
    const oracle_pos limit = strstart > max_dist ? (oracle_pos)(strstart - max_dist) : 0;
    const int early_exit = s->level < 5;                     /* :127, trigger level :14 */

    for (;;) {
        if (cur_match >= strstart)                           /* :131-132 */
            break;

        /* :167-173: skip candidates whose bytes at [offset, offset+1] or
         * [0,1] differ -- they cannot beat best_len. */
        int found = 0;
        for (;;) {
            const uint8_t *cand = window + cur_match;
            if (cand[offset] == end0 && cand[offset + 1] == end1 &&
                cand[0] == scan[0] && cand[1] == scan[1]) {
                found = 1;
                break;
            }
            /* GOTO_NEXT_CHAIN :49-52 */
            if (--chain_length && (cur_match = prev[cur_match & wmask]) > limit)
                continue;
            return best_len;
        }
        (void)found;

        uint32_t len = oracle_compare256(scan + 2, window + cur_match + 2) + 2;   /* :174 */

        if (len > best_len) {                                /* :177 */
            s->match_start = cur_match;                      /* :178-179 (match_offset is 0) */
            if (len > lookahead)                             /* :182-183 */
                return lookahead;
            best_len = len;
            if (best_len >= nice_match)                      /* :185-186 */
                return best_len;
            offset = best_len - 1;                           /* :188 */
            end0 = scan[offset];                             /* :203-204 */
            end1 = scan[offset + 1];
        } else if (early_exit) {                             /* :261-266 */
            break;
        }
        /* GOTO_NEXT_CHAIN :268 */
        if (--chain_length && (cur_match = prev[cur_match & wmask]) > limit)
            continue;
        return best_len;
    }
    return best_len;                                         /* :270 */
}

/* match_tpl.h:26-280 with LONGEST_MATCH_SLOW (arch/generic/compare256_c.c:54-58 -> longest_match_slow_c),
 * byte-pair probes.  s->update_hash is the multiplicative update_hash for levels 7-8 and the rolling one for
 * level 9; lm_init binds it by max_chain_length > 1024 (deflate.c:1223-1234), and so does this. */
uint32_t oracle_longest_match_slow(oracle_deflate_state *s, oracle_pos cur_match) {
    uint32_t (*const update_hash)(uint32_t, uint32_t) =
        s->max_chain_length > 1024 ? oracle_update_hash_roll : oracle_update_hash;
    const uint32_t strstart = s->strstart;
    const uint32_t wmask = s->w_mask;
    const uint8_t *window = s->window;
    const uint8_t *scan = window + strstart;
    const uint8_t *mbase_start = window;
    const oracle_pos *prev = s->prev;
    const uint32_t lookahead = s->lookahead;
    oracle_pos match_offset = 0;

    uint32_t best_len = s->prev_length ? s->prev_length : ORACLE_STD_MIN_MATCH - 1;     /* :59 */
    uint32_t offset = best_len - 1;                                                     /* :64 */
    uint8_t end0 = scan[offset], end1 = scan[offset + 1];                               /* :81-84 */
    uint32_t chain_length = s->max_chain_length;                                        /* :88-91 */
    if (best_len >= s->good_match)
        chain_length >>= 2;
    const uint32_t nice_match = (uint32_t)s->nice_match;
    const uint32_t max_dist = s->w_size - ORACLE_MIN_LOOKAHEAD;
    oracle_pos limit = strstart > max_dist ? (oracle_pos)(strstart - max_dist) : 0;     /* :96 */
    const oracle_pos limit_base = limit;                                                /* :98 */

    if (best_len >= ORACLE_STD_MIN_MATCH) {                                             /* :99-125 */
        uint32_t hash = update_hash(0, scan[1]);
        hash = update_hash(hash, scan[2]);
        for (uint32_t i = 3; i <= best_len; i++) {
            hash = update_hash(hash, scan[i]);
            oracle_pos pos = s->head[hash];
            if (pos < cur_match) {
                match_offset = (oracle_pos)(i - 2);
                cur_match = pos;
            }
        }
        limit = (oracle_pos)(limit_base + match_offset);
        if (cur_match <= limit)
            goto break_matching;
        mbase_start -= match_offset;
    }

    for (;;) {
        if (cur_match >= strstart)                                                      /* :131-132 */
            break;
        for (;;) {                                                                      /* :167-173 */
            const uint8_t *cand = mbase_start + cur_match;
            if (cand[offset] == end0 && cand[offset + 1] == end1 && cand[0] == scan[0] && cand[1] == scan[1])
                break;
            if (--chain_length && (cur_match = prev[cur_match & wmask]) > limit)        /* GOTO_NEXT_CHAIN */
                continue;
            return best_len;
        }
        uint32_t len = oracle_compare256(scan + 2, mbase_start + cur_match + 2) + 2;    /* :174 */
        if (len > best_len) {
            uint32_t match_start = (uint32_t)cur_match - match_offset;                  /* :178-179 */
            s->match_start = match_start;
            if (len > lookahead)
                return lookahead;
            best_len = len;
            if (best_len >= nice_match)
                return best_len;
            offset = best_len - 1;
            end0 = scan[offset];
            end1 = scan[offset + 1];
            if (len > ORACLE_STD_MIN_MATCH && match_start + len < strstart) {           /* :208-256 */
                oracle_pos pos, next_pos;
                cur_match = (oracle_pos)(cur_match - match_offset);
                match_offset = 0;
                next_pos = cur_match;
                for (uint32_t i = 0; i <= len - ORACLE_STD_MIN_MATCH; i++) {
                    pos = prev[(cur_match + i) & wmask];
                    if (pos < next_pos) {
                        if (pos <= limit_base + i)
                            goto break_matching;
                        next_pos = pos;
                        match_offset = (oracle_pos)i;
                    }
                }
                cur_match = next_pos;
                const uint8_t *endstr = scan + len - (ORACLE_STD_MIN_MATCH + 1);
                uint32_t hash = update_hash(0, endstr[0]);
                hash = update_hash(hash, endstr[1]);
                hash = update_hash(hash, endstr[2]);
                pos = s->head[hash];
                if (pos < cur_match) {
                    match_offset = (oracle_pos)(len - (ORACLE_STD_MIN_MATCH + 1));
                    if (pos <= limit_base + match_offset)
                        goto break_matching;
                    cur_match = pos;
                }
                limit = (oracle_pos)(limit_base + match_offset);
                mbase_start = window - match_offset;
                continue;
            }
        }
        if (--chain_length && (cur_match = prev[cur_match & wmask]) > limit)            /* :268 */
            continue;
        return best_len;
    }
    return best_len;

break_matching:                                                                         /* :272-278 */
    return best_len < s->lookahead ? best_len : s->lookahead;
}
