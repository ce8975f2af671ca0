/* tests/c/zlibng_min.h -- TEST INFRASTRUCTURE.  The handful of declarations the integration/arch/rocm sources take from the
 * reference tree (zbuild.h:108-114 Z_INTERNAL; crc32.h:8-14 struct crc32_fold_s), restated so that those adapter
 * files can be compile- and link-checked against libzng_rocm.so without the reference.  Nothing here is shipped. */
#ifndef ZLIBNG_MIN_H_
#define ZLIBNG_MIN_H_
#include <stddef.h>
#include <stdint.h>

#define Z_INTERNAL
#define ROCM_GFX950 1

#define CRC32_FOLD_BUFFER_SIZE (16 * 4)                 /* crc32.h:8 */
struct crc32_fold_s {                                   /* crc32.h:11-14 */
    uint8_t  fold[CRC32_FOLD_BUFFER_SIZE];
    uint32_t value;
};
#endif
