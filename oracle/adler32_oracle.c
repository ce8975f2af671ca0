/* adler32_oracle.c -- CPU restatement of zlib-ng's Adler-32 path.
 * TEST INFRASTRUCTURE ONLY (see zng_oracle.h).
 *
 * Follows /root/reference (zlib-ng 2.2.2):
 *   arch/generic/adler32_c.c:11-54   adler32_c            -> oracle_adler32
 *   adler32_p.h:11-68                BASE, NMAX, len_1/16/64 helpers
 *   arch/generic/adler32_fold_c.c:11-15 adler32_fold_copy_c -> oracle_adler32_fold_copy
 *   adler32.c:32-54                  adler32_combine_      -> oracle_adler32_combine
 */
#include <string.h>
#include "zng_oracle.h"

#define ADLER_BASE 65521u   /* adler32_p.h:11 largest prime below 2^16 */
#define ADLER_NMAX 5552u    /* adler32_p.h:12 bytes a u32 pair absorbs without overflow */

/* Accumulate n bytes without reducing (the DO1..DO16 ladder, adler32_p.h:15-19). */
static inline const uint8_t *absorb(const uint8_t *p, size_t n, uint32_t *s1, uint32_t *s2) {
    uint32_t a = *s1, b = *s2;
    for (size_t i = 0; i < n; i++) {
        a += p[i];
        b += a;
    }
    *s1 = a;
    *s2 = b;
    return p + n;
}

uint32_t oracle_adler32(uint32_t adler, const uint8_t *buf, size_t len) {
    /* adler32_c.c:16-17: halves are masked, NOT reduced, so a non-canonical
     * seed (half >= BASE) only gets folded by the first modulo below. */
    uint32_t s2 = (adler >> 16) & 0xffffu;
    uint32_t s1 = adler & 0xffffu;

    /* adler32_c.c:20-21 + adler32_p.h:21-27: the one-byte fast path is taken
     * BEFORE the NULL test. */
    if (len == 1) {
        s1 = (s1 + buf[0]) % ADLER_BASE;
        s2 = (s2 + s1) % ADLER_BASE;
        return s1 | (s2 << 16);
    }
    /* adler32_c.c:24-25 */
    if (buf == NULL)
        return 1u;

    /* adler32_c.c:28-29 (short) and :32-50 (NMAX blocks) and :53 (remainder)
     * are all "sum exactly, reduce once per <= NMAX bytes"; the results are
     * identical because the sums never wrap (NMAX bound, adler32_p.h:13). */
    while (len >= ADLER_NMAX) {
        buf = absorb(buf, ADLER_NMAX, &s1, &s2);
        len -= ADLER_NMAX;
        s1 %= ADLER_BASE;
        s2 %= ADLER_BASE;
    }
    absorb(buf, len, &s1, &s2);
    s1 %= ADLER_BASE;
    s2 %= ADLER_BASE;
    return s1 | (s2 << 16);
}

uint32_t oracle_adler32_fold_copy(uint32_t adler, uint8_t *dst, const uint8_t *src, size_t len) {
    /* adler32_fold_c.c:12-14: checksum of src, then a plain copy. */
    adler = oracle_adler32(adler, src, len);
    memcpy(dst, src, len);
    return adler;
}

uint32_t oracle_adler32_combine(uint32_t adler1, uint32_t adler2, int64_t len2) {
    /* adler32.c:37-39 */
    if (len2 < 0)
        return 0xffffffffu;

    /* adler32.c:42-53.  The branch-reduced form is kept (not replaced by a
     * plain %): with non-canonical inputs the two would differ. */
    uint32_t rem = (uint32_t)(len2 % ADLER_BASE);
    uint32_t lo1 = adler1 & 0xffffu;
    uint32_t sum2 = (rem * lo1) % ADLER_BASE;
    uint32_t sum1 = lo1 + (adler2 & 0xffffu) + ADLER_BASE - 1;
    sum2 += ((adler1 >> 16) & 0xffffu) + ((adler2 >> 16) & 0xffffu) + ADLER_BASE - rem;
    if (sum1 >= ADLER_BASE) sum1 -= ADLER_BASE;
    if (sum1 >= ADLER_BASE) sum1 -= ADLER_BASE;
    if (sum2 >= (ADLER_BASE << 1)) sum2 -= (ADLER_BASE << 1);
    if (sum2 >= ADLER_BASE) sum2 -= ADLER_BASE;
    return sum1 | (sum2 << 16);
}
