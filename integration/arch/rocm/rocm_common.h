/* arch/rocm/rocm_common.h -- extension blocks of deflate_state / inflate_state for the arch/rocm backend
 * (deflate.h:319-321 `arch_deflate_state arch`, inflate.h:160-162; precedent arch/s390/dfltcc_common.h:66-84).
 * The device side of both (history in HBM, staging, HIP stream) is one opaque zng_rocm_hook of libzng_rocm. */
#ifndef ROCM_COMMON_H_
#define ROCM_COMMON_H_
#include <stddef.h>
#include <stdint.h>
#include "zng_rocm.h"

#ifndef ROCM_DEFLATE_BLOCK_BYTES
#  define ROCM_DEFLATE_BLOCK_BYTES (8u << 20)   /* input gathered per device block when the caller does not flush */
#endif

typedef struct {
    zng_rocm_hook *hook;        /* created at the first deflate() the device takes; NULL = not tried yet */
    uint8_t *in_buf;            /* input gathered until it is worth a launch (deflate() may be called with a few bytes) */
    size_t   in_len, in_cap;
    uint8_t *out_buf;           /* compressed bytes next_out has not had room for yet */
    size_t   out_pos, out_len, out_cap;
    int      used;              /* the device has produced part of this stream */
    int      finished;          /* the BFINAL block has been produced */
    int      disabled;          /* no device, or it failed: software from here on */
} arch_deflate_state;

typedef struct {
    zng_rocm_hook *hook;
    uint8_t *in_buf;            /* compressed bytes gathered until the stream's end is among them */
    size_t   in_len, in_cap;
    const uint8_t *out;         /* plaintext (memory of the hook) next_out has not had room for yet */
    size_t   out_pos, out_len;
    uint32_t check;             /* its check value, handed to state->check once everything is delivered */
    int      used, done, disabled;
} arch_inflate_state;
#endif
