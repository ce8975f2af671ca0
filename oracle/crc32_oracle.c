/* crc32_oracle.c -- CPU restatement of zlib-ng's CRC-32 path.
 * TEST INFRASTRUCTURE ONLY (see zng_oracle.h).
 *
 * Follows /root/reference (zlib-ng 2.2.2):
 *   arch/generic/crc32_braid_c.c:62-216  crc32_braid (N=5 braids, W=8 bytes) -> oracle_crc32_braid
 *   arch/generic/crc32_braid_c.c:43-57   crc_word
 *   crc32_braid_p.h:58-62                DO1 / POLY
 *   crc32.c:16-41                        export layer (NULL -> 0)       -> oracle_crc32
 *   arch/generic/crc32_fold_c.c:10-31    fold quartet                   -> oracle_crc32_fold*
 *   crc32_braid_comb_p.h:8-40            multmodp, x2nmodp
 *   crc32_braid_comb.c:16-24             combine / combine_gen / combine_op
 *   tools/makecrct.c:66-79,99-112        how crc_table, x2n_table and the braid
 *                                        tables are defined (the tables are
 *                                        regenerated here, not copied)
 */
#include <string.h>
#include "zng_oracle.h"

#define CRC_POLY 0xedb88320u    /* crc32_braid_p.h:62, reflected */
#define BRAIDS   5              /* crc32_braid_p.h:9  N */
#define WORDSZ   8              /* crc32_braid_p.h:29 W on 64-bit little-endian hosts */

static uint32_t byte_tab[256];              /* crc_table        (crc32_braid_tbl.h:8)    */
static uint32_t pow2_tab[32];               /* x2n_table        (crc32_braid_tbl.h:9437) */
static uint32_t braid_tab[WORDSZ][256];     /* crc_braid_table, N=5 W=8 (crc32_braid_tbl.h:6366) */
static int      tabs_ready;

/* crc32_braid_comb_p.h:8-24.  Polynomials are bit-reversed: bit 31 is x^0. */
uint32_t oracle_multmodp(uint32_t a, uint32_t b) {
    uint32_t acc = 0;
    for (uint32_t bit = 0x80000000u; bit; bit >>= 1) {
        if (a & bit)
            acc ^= b;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return acc;
}

static void build_tables(void) {
    /* makecrct.c:66-73: crc_table[i] = (i as a degree-7 polynomial) * x^32 mod p */
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t r = i;
        for (int k = 0; k < 8; k++)
            r = (r & 1u) ? (r >> 1) ^ CRC_POLY : r >> 1;
        byte_tab[i] = r;
    }
    /* makecrct.c:75-79: x2n_table[n] = x^(2^n) mod p */
    uint32_t p = 0x40000000u;   /* x^1 */
    pow2_tab[0] = p;
    for (int n = 1; n < 32; n++)
        pow2_tab[n] = p = oracle_multmodp(p, p);
    tabs_ready = 1;             /* x2nmodp below needs pow2_tab */
    /* makecrct.c:99-112: braid table k maps byte k of a word to that byte
     * advanced over one whole braid stride: x^(8*(N*W + 3 - k)). */
    for (int k = 0; k < WORDSZ; k++) {
        uint32_t adv = oracle_x2nmodp(((int64_t)BRAIDS * WORDSZ + 3 - k) << 3, 0);
        braid_tab[k][0] = 0;
        for (uint32_t i = 1; i < 256; i++)
            braid_tab[k][i] = oracle_multmodp(i << 24, adv);
    }
}

static inline void need_tables(void) {
    if (!tabs_ready)
        build_tables();
}

const uint32_t *oracle_get_crc_table(void) {
    need_tables();
    return byte_tab;
}

/* crc32_braid_comb_p.h:29-40: x^(n * 2^k) mod p by square-and-multiply. */
uint32_t oracle_x2nmodp(int64_t n, unsigned k) {
    if (!tabs_ready)
        build_tables();
    uint32_t p = 0x80000000u;   /* x^0 */
    while (n) {
        if (n & 1)
            p = oracle_multmodp(pow2_tab[k & 31], p);
        n >>= 1;
        k++;
    }
    return p;
}

/* crc32_braid_c.c:43-49: push the W bytes of a word through the byte table. */
static inline uint32_t word_through_bytes(uint64_t v) {
    for (int k = 0; k < WORDSZ; k++)
        v = (v >> 8) ^ byte_tab[v & 0xffu];
    return (uint32_t)v;
}

static inline uint32_t step_byte(uint32_t c, uint8_t b) {      /* DO1, crc32_braid_p.h:58 */
    return byte_tab[(c ^ b) & 0xffu] ^ (c >> 8);
}

uint32_t oracle_crc32_bytewise(uint32_t crc, const uint8_t *buf, size_t len) {
    need_tables();
    uint32_t c = ~crc;
    while (len--)
        c = step_byte(c, *buf++);
    return ~c;
}

uint32_t oracle_crc32_braid(uint32_t crc, const uint8_t *buf, size_t len) {
    need_tables();
    uint32_t c = ~crc;                                          /* :66 pre-condition */

    if (len >= BRAIDS * WORDSZ + WORDSZ - 1) {                  /* :70 */
        /* :75-78 bytewise up to a word boundary */
        while (len && ((uintptr_t)buf & (WORDSZ - 1)) != 0) {
            c = step_byte(c, *buf++);
            len--;
        }
        size_t blocks = len / (BRAIDS * WORDSZ);                /* :81-82 */
        len -= blocks * BRAIDS * WORDSZ;

        uint64_t lane[BRAIDS] = { c, 0, 0, 0, 0 };              /* :102-118 */
        uint64_t w[BRAIDS];
        /* :121-176 all blocks but the last: every braid advances by N*W bytes */
        while (--blocks) {
            for (int b = 0; b < BRAIDS; b++) {
                memcpy(&w[b], buf + b * WORDSZ, WORDSZ);
                w[b] ^= lane[b];
            }
            buf += BRAIDS * WORDSZ;
            for (int b = 0; b < BRAIDS; b++) {
                uint64_t v = w[b];
                uint32_t r = braid_tab[0][v & 0xffu];
                for (int k = 1; k < WORDSZ; k++)
                    r ^= braid_tab[k][(v >> (k << 3)) & 0xffu];
                lane[b] = r;
            }
        }
        /* :179-198 last block: fold the braids together word by word */
        uint64_t comb = 0;
        for (int b = 0; b < BRAIDS; b++) {
            uint64_t v;
            memcpy(&v, buf + b * WORDSZ, WORDSZ);
            comb = word_through_bytes(lane[b] ^ v ^ comb);
        }
        buf += BRAIDS * WORDSZ;
        c = (uint32_t)comb;
    }
    /* :206-213 tail */
    while (len--)
        c = step_byte(c, *buf++);
    return ~c;                                                  /* :215 */
}

uint32_t oracle_crc32(uint32_t crc, const uint8_t *buf, size_t len) {
    if (buf == NULL)                                            /* crc32.c:22,28 */
        return 0;
    return oracle_crc32_braid(crc, buf, len);
}

/* arch/generic/crc32_fold_c.c:10-31: the generic fold state is just `value`. */
uint32_t oracle_crc32_fold_reset(oracle_crc32_fold_t *crc) {
    crc->value = 0;                                             /* CRC32_INITIAL_VALUE */
    return crc->value;
}

void oracle_crc32_fold(oracle_crc32_fold_t *crc, const uint8_t *src, size_t len, uint32_t init_crc) {
    (void)init_crc;                                             /* :24 unused in the generic variant */
    crc->value = oracle_crc32_braid(crc->value, src, len);
}

void oracle_crc32_fold_copy(oracle_crc32_fold_t *crc, uint8_t *dst, const uint8_t *src, size_t len) {
    crc->value = oracle_crc32_braid(crc->value, src, len);
    memcpy(dst, src, len);
}

uint32_t oracle_crc32_fold_final(oracle_crc32_fold_t *crc) {
    return crc->value;
}

/* crc32_braid_comb.c:16-24 */
uint32_t oracle_crc32_combine(uint32_t crc1, uint32_t crc2, int64_t len2) {
    return oracle_multmodp(oracle_x2nmodp(len2, 3), crc1) ^ crc2;
}

uint32_t oracle_crc32_combine_gen(int64_t len2) {
    return oracle_x2nmodp(len2, 3);
}

uint32_t oracle_crc32_combine_op(uint32_t crc1, uint32_t crc2, uint32_t op) {
    return oracle_multmodp(op, crc1) ^ crc2;
}
