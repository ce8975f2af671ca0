/* zng_oracle.h -- CPU restatement of zlib-ng's functable hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product (zlib-ng_amd/, include/) never links or calls it.
 *
 * Every function restates the algorithm of one reference function (zlib-ng
 * 2.2.2); the reference file:line it follows is cited at each definition.
 * Parity pinning (see oracle/README.md): the reference's own known-answer
 * vectors (tests/golden/, extracted from test/test_adler32.cc,
 * test/test_crc32.cc, test/infcover.c ...) plus oracle/_ref (the reference's
 * arch/generic/crc32_braid_c.c compiled from its own sources).
 */
#ifndef ZNG_ORACLE_H
#define ZNG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Adler-32 (adler32.c, adler32_p.h, arch/generic/adler32_c.c) ------- */
uint32_t oracle_adler32(uint32_t adler, const uint8_t *buf, size_t len);
uint32_t oracle_adler32_fold_copy(uint32_t adler, uint8_t *dst, const uint8_t *src, size_t len);
uint32_t oracle_adler32_combine(uint32_t adler1, uint32_t adler2, int64_t len2);

/* ---- CRC-32 (crc32.c, arch/generic/crc32_braid_c.c, crc32_braid_comb*.c) */
typedef struct oracle_crc32_fold_s {
    uint8_t  fold[64];      /* crc32.h:8-14: only SIMD variants use fold[] */
    uint32_t value;
} oracle_crc32_fold_t;

uint32_t oracle_crc32_braid(uint32_t crc, const uint8_t *buf, size_t len);
uint32_t oracle_crc32_bytewise(uint32_t crc, const uint8_t *buf, size_t len);
uint32_t oracle_crc32(uint32_t crc, const uint8_t *buf, size_t len);   /* export layer: NULL -> 0 */
uint32_t oracle_crc32_fold_reset(oracle_crc32_fold_t *crc);
void     oracle_crc32_fold(oracle_crc32_fold_t *crc, const uint8_t *src, size_t len, uint32_t init_crc);
void     oracle_crc32_fold_copy(oracle_crc32_fold_t *crc, uint8_t *dst, const uint8_t *src, size_t len);
uint32_t oracle_crc32_fold_final(oracle_crc32_fold_t *crc);
uint32_t oracle_multmodp(uint32_t a, uint32_t b);
uint32_t oracle_x2nmodp(int64_t n, unsigned k);
uint32_t oracle_crc32_combine(uint32_t crc1, uint32_t crc2, int64_t len2);
uint32_t oracle_crc32_combine_gen(int64_t len2);
uint32_t oracle_crc32_combine_op(uint32_t crc1, uint32_t crc2, uint32_t op);
const uint32_t *oracle_get_crc_table(void);

/* ---- deflate-side primitives (deflate.h state subset) ------------------ */
#define ORACLE_HASH_SIZE 65536u
#define ORACLE_STD_MIN_MATCH 3
#define ORACLE_STD_MAX_MATCH 258
#define ORACLE_MIN_LOOKAHEAD (ORACLE_STD_MAX_MATCH + ORACLE_STD_MIN_MATCH + 1)

typedef uint16_t oracle_pos;

/* The subset of deflate_state (deflate.h:144-334) the hot-path kernels touch
 * (SURVEY.md section 8 a13). */
typedef struct oracle_deflate_state {
    uint32_t    w_size, w_bits, w_mask;
    uint32_t    lookahead;
    uint32_t    window_size;
    uint8_t    *window;
    oracle_pos *prev;
    oracle_pos *head;
    uint32_t    strstart;
    uint32_t    match_start;
    uint32_t    prev_length;
    uint32_t    max_chain_length;
    uint32_t    good_match;
    int32_t     nice_match;
    int32_t     level;
    uint32_t    ins_h;          /* running key of the rolling hash (deflate.h:196; level 9 only) */
} oracle_deflate_state;

void       oracle_slide_hash(oracle_deflate_state *s);
uint32_t   oracle_compare256(const uint8_t *src0, const uint8_t *src1);
uint32_t   oracle_update_hash(uint32_t h, uint32_t val);
oracle_pos oracle_quick_insert_string(oracle_deflate_state *s, uint32_t str);
void       oracle_insert_string(oracle_deflate_state *s, uint32_t str, uint32_t count);
/* insert_string_roll.c (level 9: bound when max_chain_length > 1024, deflate.c:1223-1234) */
uint32_t   oracle_update_hash_roll(uint32_t h, uint32_t val);
oracle_pos oracle_quick_insert_string_roll(oracle_deflate_state *s, uint32_t str);
void       oracle_insert_string_roll(oracle_deflate_state *s, uint32_t str, uint32_t count);
uint32_t   oracle_longest_match(oracle_deflate_state *s, oracle_pos cur_match);
uint32_t   oracle_longest_match_slow(oracle_deflate_state *s, oracle_pos cur_match);

/* ---- inflate-side primitives (chunkset_tpl.h) -------------------------- */
uint32_t   oracle_chunksize(void);
uint8_t   *oracle_chunkmemset_safe(uint8_t *out, uint8_t *from, unsigned len, unsigned left);

/* ---- raw inflate (inflate.c / inffast_tpl.h / inftrees.c behaviour) ---- */
#define ORACLE_Z_OK           0
#define ORACLE_Z_STREAM_END   1
#define ORACLE_Z_DATA_ERROR (-3)
#define ORACLE_Z_BUF_ERROR  (-5)     /* input ended early, or dst full */

typedef struct oracle_inflate_result {
    int         status;
    const char *msg;        /* the reference's strm->msg text for data errors */
    size_t      out_len;    /* bytes produced (also on error: everything before the bad symbol) */
    size_t      in_used;
} oracle_inflate_result;

int oracle_inflate_raw(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                       oracle_inflate_result *res);
/* test aid: record the bit position (of BFINAL) and BTYPE of every block the next oracle_inflate_raw* call walks
 * (at most cap; pass NULL to switch off); oracle_inflate_traced_blocks() = how many there were */
void oracle_inflate_trace_blocks(uint64_t *bits, uint8_t *types, size_t cap);
size_t oracle_inflate_traced_blocks(void);
/* the same after inflateSetDictionary (inflate.c:1214-1261) */
int oracle_inflate_raw_dict(const uint8_t *src, size_t src_len, const uint8_t *dict, size_t dict_len,
                            uint8_t *dst, size_t dst_cap, oracle_inflate_result *res);

#ifdef __cplusplus
}
#endif
#endif /* ZNG_ORACLE_H */
