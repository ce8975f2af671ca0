/* tests/c/abi_c11.c -- TEST: include/zng_rocm.h as a C11 consumer sees it, and the reference-side adapters of
 * integration/arch/rocm linked against libzng_rocm.so.
 *   - the header compiles as strict C11 (-std=c11 -pedantic -Wall -Wextra -Werror)
 *   - zng_rocm_crc32_fold_t has the layout of struct crc32_fold_s (crc32.h:11-14): 64 + 4 bytes
 *   - without a GPU the adapters fall back to the CPU tier they remembered (SURVEY.md 8b error convention) instead
 *     of aborting; with one, they return the device's value, and both agree with the reference's own KATs
 * prints "ok <device|fallback>" and exits 0. */
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "zlibng_min.h"
#include "zng_rocm.h"
#include "rocm_functions.h"

_Static_assert(sizeof(zng_rocm_crc32_fold_t) == 68, "struct crc32_fold_s: uint8_t fold[64]; uint32_t value");
_Static_assert(offsetof(zng_rocm_crc32_fold_t, fold) == 0, "fold first");
_Static_assert(offsetof(zng_rocm_crc32_fold_t, value) == 64, "value after the 64-byte fold buffer");
_Static_assert(sizeof(zng_rocm_check_row) == 16, "packed {adler, crc, len} row");
_Static_assert(sizeof(zng_rocm_stream_job) == 32, "two pointers + four u32");

/* the "CPU tier chosen so far": plain bytewise forms (adler32_c.c:11-54 / crc32_braid_p.h DO1 semantics) */
static uint32_t cpu_adler(uint32_t adler, const uint8_t *buf, size_t len) {
    uint32_t s1 = adler & 0xffff, s2 = (adler >> 16) & 0xffff;
    if (buf == NULL) return 1;
    for (size_t i = 0; i < len; ++i) {
        s1 = (s1 + buf[i]) % 65521u;
        s2 = (s2 + s1) % 65521u;
    }
    return s1 | (s2 << 16);
}
static uint32_t cpu_crc(uint32_t crc, const uint8_t *buf, size_t len) {
    crc = ~crc;
    for (size_t i = 0; i < len; ++i) {
        crc ^= buf[i];
        for (int k = 0; k < 8; ++k) crc = (crc >> 1) ^ (0xedb88320u & (0u - (crc & 1u)));
    }
    return ~crc;
}

int main(void) {
    struct rocm_cpu_features f;
    rocm_check_features(&f);                            /* must not abort without a device */
    rocm_remember_cpu_tier(cpu_adler, cpu_crc);

    /* values SURVEY.md 9.1 records from the real reference: adler32_c("abacus") / zng_crc32_braid("abacus") */
    const uint8_t *abacus = (const uint8_t *)"abacus";
    if (adler32_rocm(1, abacus, 6) != 0x08400270u || crc32_rocm(0, abacus, 6) != 0xc3d7115bu) {
        fprintf(stderr, "KAT mismatch\n");
        return 1;
    }
    /* a buffer above the (test-lowered) threshold: device path when there is one, CPU tier otherwise */
    size_t n = 3u << 20;
    uint8_t *buf = malloc(n), *dst = malloc(n);
    for (size_t i = 0; i < n; ++i) buf[i] = (uint8_t)(i * 2654435761u >> 13);
    uint32_t a = adler32_rocm(0xdeadc0deu, buf, n), c = crc32_rocm(0x12345678u, buf, n);
    if (a != cpu_adler(0xdeadc0deu, buf, n) || c != cpu_crc(0x12345678u, buf, n)) {
        fprintf(stderr, "adapter value differs from the CPU tier: %08x %08x\n", a, c);
        return 1;
    }
    struct crc32_fold_s st;
    crc32_fold_reset_rocm(&st);
    crc32_fold_rocm(&st, buf, n / 2, 0);
    crc32_fold_copy_rocm(&st, dst, buf + n / 2, n - n / 2);
    if (crc32_fold_final_rocm(&st) != cpu_crc(0, buf, n) || memcmp(dst, buf + n / 2, n - n / 2) != 0) {
        fprintf(stderr, "fold quartet mismatch\n");
        return 1;
    }
    memset(dst, 0, n);
    if (adler32_fold_copy_rocm(1, dst, buf, n) != cpu_adler(1, buf, n) || memcmp(dst, buf, n) != 0) {
        fprintf(stderr, "adler32_fold_copy mismatch\n");
        return 1;
    }
    /* the _try forms never abort: without a device they report ZNG_ROCM_ENODEV */
    uint32_t out = 0;
    int rc = zng_rocm_adler32_try(1, buf, n, &out);
    if (f.has_gfx950 ? rc != ZNG_ROCM_OK : rc != ZNG_ROCM_ENODEV) {
        fprintf(stderr, "unexpected status %d\n", rc);
        return 1;
    }
    printf("ok %s\n", f.has_gfx950 ? "device" : "fallback");
    free(buf);
    free(dst);
    return 0;
}
