/* arch/rocm/rocm_functions.h -- what a zlib-ng maintainer adds to the REFERENCE tree (not part of this repository's
 * product; INTEGRATION.md sections 2-4).  Pattern: arch/x86/x86_functions.h, arch/s390/s390_functions.h. */
#ifndef ROCM_FUNCTIONS_H_
#define ROCM_FUNCTIONS_H_

#ifdef ROCM_GFX950
struct rocm_cpu_features { int has_gfx950; };
void rocm_check_features(struct rocm_cpu_features *features);

uint32_t adler32_rocm(uint32_t adler, const uint8_t *buf, size_t len);
uint32_t adler32_fold_copy_rocm(uint32_t adler, uint8_t *dst, const uint8_t *src, size_t len);
uint32_t crc32_rocm(uint32_t crc, const uint8_t *buf, size_t len);
uint32_t crc32_fold_reset_rocm(struct crc32_fold_s *crc);
void     crc32_fold_rocm(struct crc32_fold_s *crc, const uint8_t *src, size_t len, uint32_t init_crc);
void     crc32_fold_copy_rocm(struct crc32_fold_s *crc, uint8_t *dst, const uint8_t *src, size_t len);
uint32_t crc32_fold_final_rocm(struct crc32_fold_s *crc);
/* init_functable() hands over the CPU tier it has chosen so far, before it installs the slots above */
void rocm_remember_cpu_tier(uint32_t (*adler32)(uint32_t, const uint8_t *, size_t),
                            uint32_t (*crc32)(uint32_t, const uint8_t *, size_t));
/* ... and the rest of arch/rocm uses it where the device has to be left out (rocm_deflate.c, rocm_inflate.c) */
uint32_t rocm_cpu_adler32(uint32_t adler, const uint8_t *buf, size_t len);
uint32_t rocm_cpu_crc32(uint32_t crc, const uint8_t *buf, size_t len);
#endif

#endif
