/* tests/c/zlibng_coarse_min.h -- TEST INFRASTRUCTURE.  The declarations integration/arch/rocm/rocm_deflate.c and
 * rocm_inflate.c take from the reference tree, restated (names, meaning and field order of the parts that matter as in
 * zlib-ng.h.in:99-119 zng_stream, deflate.h:164-344 deflate_state / block_state, inflate.h:27-162 inflate_state /
 * inflate_mode, zlib-ng.h.in:165-188 the flush and return codes) so that the adapters can be compiled as strict C11,
 * linked against libzng_rocm.so and driven by tests/c/coarse_driver.c without the reference.  zlib-ng.h itself cannot
 * be used: it exists only after the reference's build system has generated it.  Nothing here is shipped. */
#ifndef ZLIBNG_COARSE_MIN_H_
#define ZLIBNG_COARSE_MIN_H_
#include "zlibng_min.h"

#define PREFIX(x) zng_##x
#define PREFIX3(x) zng_##x
#define Z_NO_FLUSH 0
#define Z_PARTIAL_FLUSH 1
#define Z_SYNC_FLUSH 2
#define Z_FULL_FLUSH 3
#define Z_FINISH 4
#define Z_BLOCK 5
#define Z_TREES 6
#define Z_OK 0
#define Z_STREAM_END 1
#define Z_NEED_DICT 2
#define Z_STREAM_ERROR (-2)
#define Z_DATA_ERROR (-3)
#define Z_MEM_ERROR (-4)
#define Z_BUF_ERROR (-5)
#define Z_DEFAULT_STRATEGY 0
#define Z_FILTERED 1
#define HAVE_ARCH_DEFLATE_STATE 1
#define HAVE_ARCH_INFLATE_STATE 1

struct internal_state;
typedef struct zng_stream_s {                   /* zlib-ng.h.in:99-119 */
    const uint8_t         *next_in;
    uint32_t               avail_in;
    size_t                 total_in;
    uint8_t               *next_out;
    uint32_t               avail_out;
    size_t                 total_out;
    const char            *msg;
    struct internal_state *state;
    void *(*zalloc)(void *opaque, unsigned items, unsigned size);
    void  (*zfree)(void *opaque, void *address);
    void                  *opaque;
    int                    data_type;
    uint32_t               adler;
    unsigned long          reserved;
} zng_stream;
typedef zng_stream *zng_streamp;

typedef enum { need_more, block_done, finish_started, finish_done } block_state;      /* deflate.h:336-341 */

#include "rocm_common.h"                        /* arch_deflate_state / arch_inflate_state */

typedef struct internal_state {                 /* the deflate_state fields the adapter reads (deflate.h:164-321) */
    zng_stream *strm;
    int         wrap;                           /* 0 raw, 1 zlib, 2 gzip */
    int         level, strategy;
    unsigned    w_bits;
    struct crc32_fold_s crc_fold;               /* deflate.h:250 */
    arch_deflate_state arch;                    /* deflate.h:319-321 */
} deflate_state;

typedef enum { TYPEDO = 16191, CHECK = 16206, DONE = 16208, BAD = 16209 } inflate_mode;     /* inflate.h:27-58 */
struct inflate_state {                          /* the fields the adapter touches (inflate.h:108-162) */
    zng_stream  *strm;
    inflate_mode mode;
    int          last, wrap, flags;
    unsigned     wbits, bits;
    uint32_t     check;
    arch_inflate_state arch;                    /* inflate.h:160-162 */
};
#endif
