/* arch/rocm/rocm_slots.c -- functable slot adapters of the arch/rocm backend.
 *
 * Two rules live here, on the zlib-ng side of the boundary:
 *   1. size threshold: a functable call inside deflate()/inflate() is <= 64 KiB (deflate.c:1190-1212); a GPU
 *      launch + PCIe staging only pays for the big one-shot zng_adler32_z / zng_crc32_z calls.  Precedent for a
 *      threshold inside a dispatched kernel: crc32_pclmulqdq (arch/x86/crc32_pclmulqdq_tpl.h:354-375),
 *      adler32_avx2 (arch/x86/adler32_avx2.c:29-42).
 *   2. error convention: the slots have no error channel, so a HIP failure must degrade to the CPU tier and never
 *      surface.  The adapters therefore bind the status-returning `_try` forms of libzng_rocm and fall back to the
 *      CPU functions init_functable() had chosen before it installed them.
 * libzng_rocm itself never computes a checksum on the host. */
#ifdef ZNG_ROCM_STANDALONE_CHECK
#  include "zlibng_min.h"          /* tests/c: the few reference declarations this file needs, for a compile check */
#else
#  include "zbuild.h"
#  include "crc32.h"               /* struct crc32_fold_s, crc32.h:8-14 */
#endif
#include <string.h>
#include "zng_rocm.h"
#include "rocm_functions.h"

#ifndef ROCM_MIN_BYTES
#  define ROCM_MIN_BYTES (4u << 20)    /* below this PCIe staging costs more than the host loop */
#endif

/* struct crc32_fold_s and zng_rocm_crc32_fold_t are the same layout: the adapters cast */
_Static_assert(sizeof(struct crc32_fold_s) == sizeof(zng_rocm_crc32_fold_t), "crc32_fold_s layout");
_Static_assert(offsetof(struct crc32_fold_s, value) == offsetof(zng_rocm_crc32_fold_t, value), "crc32_fold_s layout");

static uint32_t (*cpu_adler32)(uint32_t, const uint8_t *, size_t);
static uint32_t (*cpu_crc32)(uint32_t, const uint8_t *, size_t);

void Z_INTERNAL rocm_remember_cpu_tier(uint32_t (*adler32)(uint32_t, const uint8_t *, size_t),
                                       uint32_t (*crc32)(uint32_t, const uint8_t *, size_t)) {
    cpu_adler32 = adler32;
    cpu_crc32 = crc32;
}

Z_INTERNAL uint32_t rocm_cpu_adler32(uint32_t adler, const uint8_t *buf, size_t len) { return cpu_adler32(adler, buf, len); }
Z_INTERNAL uint32_t rocm_cpu_crc32(uint32_t crc, const uint8_t *buf, size_t len) { return cpu_crc32(crc, buf, len); }

Z_INTERNAL uint32_t adler32_rocm(uint32_t adler, const uint8_t *buf, size_t len) {
    uint32_t out;
    if (len >= ROCM_MIN_BYTES && zng_rocm_adler32_try(adler, buf, len, &out) == ZNG_ROCM_OK)
        return out;                                     /* replaces adler32_c, arch/generic/adler32_c.c:11-54 */
    return cpu_adler32(adler, buf, len);
}

Z_INTERNAL uint32_t crc32_rocm(uint32_t crc, const uint8_t *buf, size_t len) {
    uint32_t out;
    if (len >= ROCM_MIN_BYTES && zng_rocm_crc32_try(crc, buf, len, &out) == ZNG_ROCM_OK)
        return out;                                     /* replaces crc32_braid, arch/generic/crc32_braid_c.c:62-216 */
    return cpu_crc32(crc, buf, len);
}

Z_INTERNAL uint32_t adler32_fold_copy_rocm(uint32_t adler, uint8_t *dst, const uint8_t *src, size_t len) {
    uint32_t out;
    if (len >= ROCM_MIN_BYTES && zng_rocm_adler32_fold_copy_try(adler, dst, src, len, &out) == ZNG_ROCM_OK)
        return out;
    adler = cpu_adler32(adler, src, len);               /* adler32_fold_copy_c, arch/generic/adler32_fold_c.c:11-15 */
    memcpy(dst, src, len);
    return adler;
}

/* generic fold semantics (arch/generic/crc32_fold_c.c:10-31): only `value` is used */
Z_INTERNAL uint32_t crc32_fold_reset_rocm(struct crc32_fold_s *crc) {
    crc->value = 0;
    return crc->value;
}

Z_INTERNAL void crc32_fold_rocm(struct crc32_fold_s *crc, const uint8_t *src, size_t len, uint32_t init_crc) {
    if (len >= ROCM_MIN_BYTES && zng_rocm_crc32_fold_try((zng_rocm_crc32_fold_t *)crc, src, len, init_crc) == ZNG_ROCM_OK)
        return;
    crc->value = cpu_crc32(crc->value, src, len);
}

Z_INTERNAL void crc32_fold_copy_rocm(struct crc32_fold_s *crc, uint8_t *dst, const uint8_t *src, size_t len) {
    if (len >= ROCM_MIN_BYTES && zng_rocm_crc32_fold_copy_try((zng_rocm_crc32_fold_t *)crc, dst, src, len) == ZNG_ROCM_OK)
        return;
    crc->value = cpu_crc32(crc->value, src, len);
    memcpy(dst, src, len);
}

Z_INTERNAL uint32_t crc32_fold_final_rocm(struct crc32_fold_s *crc) {
    return crc->value;
}
