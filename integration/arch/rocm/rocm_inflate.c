/* arch/rocm/rocm_inflate.c -- INFLATE_TYPEDO_HOOK backend of arch/rocm (inflate.c:728; precedent
 * arch/s390/dfltcc_inflate.c:52-114).  inflate() keeps the wrapper (zlib / gzip header and trailer, inflate.c:509-700,
 * :1105-1147) and the zng_stream bookkeeping; the deflate data between them is decoded by libzng_rocm (token decode on
 * the host, every match copy on the device) and its check value comes back with it (INFLATE_NEED_CHECKSUM = 0).
 * The device takes a stream whole: compressed bytes are gathered until the end of the deflate data is among them -- a
 * decode is attempted on every call that brings input, so whatever lies behind the end (the trailer, a next member) is
 * still inside the caller's current buffer and next_in is set back to it.  A caller that hands over the stream in one
 * piece pays one decode; one that trickles it pays one per call (the software path is the better choice there:
 * ROCM_INFLATE_MIN_BYTES).  Anything the device path cannot do leaves the stream to software BEFORE a byte is consumed. */
#ifdef ZNG_ROCM_STANDALONE_CHECK
#  include "zlibng_coarse_min.h"
#else
#  include "zbuild.h"
#  include "inflate.h"
#endif
#include <stdlib.h>
#include <string.h>
#include "zng_rocm.h"
#include "rocm_functions.h"
#include "rocm_inflate.h"

#ifndef ROCM_INFLATE_MIN_BYTES
#  define ROCM_INFLATE_MIN_BYTES (1u << 20)     /* a first call with less input than this stays in software */
#endif

static void *arch_alloc(PREFIX3(streamp) strm, size_t n) {
    return strm->zalloc ? strm->zalloc(strm->opaque, 1, (unsigned)n) : malloc(n);
}
static void arch_free(PREFIX3(streamp) strm, void *p) {
    if (!p) return;
    if (strm->zfree) strm->zfree(strm->opaque, p);
    else free(p);
}

void Z_INTERNAL PREFIX(archrocm_reset_inflate_state)(PREFIX3(streamp) strm) {      /* INFLATE_RESET_KEEP_HOOK, inflate.c:87 */
    arch_inflate_state *a = &((struct inflate_state *)strm->state)->arch;
    a->in_len = a->out_pos = a->out_len = 0;
    a->out = NULL;
    a->used = a->done = 0;
    if (a->hook && zng_rocm_hook_reset(a->hook) != ZNG_ROCM_OK) a->disabled = 1;
}

void Z_INTERNAL PREFIX(archrocm_inflate_end)(PREFIX3(streamp) strm) {              /* INFLATE_END_HOOK, in inflateEnd() */
    arch_inflate_state *a = &((struct inflate_state *)strm->state)->arch;
    zng_rocm_hook_destroy(a->hook);
    arch_free(strm, a->in_buf);
    memset(a, 0, sizeof *a);
}

int Z_INTERNAL PREFIX(archrocm_can_inflate)(PREFIX3(streamp) strm) {
    struct inflate_state *state = (struct inflate_state *)strm->state;
    arch_inflate_state *a = &state->arch;
    if (a->disabled || state->wbits != 15) return 0;
    if (a->used) return 1;
    /* a stream begins here: whole bytes only (a block that starts inside a byte follows blocks software has decoded), and
     * enough input to be worth the launches */
    if (state->bits != 0 || strm->avail_in < ROCM_INFLATE_MIN_BYTES) return 0;
    if (!a->hook) {
        if (zng_rocm_device_count() <= 0 || zng_rocm_init(-1) != ZNG_ROCM_OK ||
            zng_rocm_hook_create(&a->hook, 1u << 20) != ZNG_ROCM_OK) {
            a->hook = NULL;
            a->disabled = 1;
            return 0;
        }
    }
    return 1;
}

int Z_INTERNAL PREFIX(archrocm_was_inflate_used)(PREFIX3(streamp) strm) {
    return ((struct inflate_state *)strm->state)->arch.used;
}

int Z_INTERNAL PREFIX(archrocm_inflate_disable)(PREFIX3(streamp) strm) {           /* inflatePrime: bit granular, software only */
    arch_inflate_state *a = &((struct inflate_state *)strm->state)->arch;
    if (a->used) return 1;                              /* too late: the caller gets Z_STREAM_ERROR */
    a->disabled = 1;
    return 0;
}

static void drain(PREFIX3(streamp) strm, arch_inflate_state *a) {
    size_t n = a->out_len - a->out_pos;
    if (n > strm->avail_out) n = strm->avail_out;
    if (n) {
        memcpy(strm->next_out, a->out + a->out_pos, n);
        strm->next_out += n;
        strm->avail_out -= (uint32_t)n;
        a->out_pos += n;                                /* total_out: inflate() adds what avail_out lost, inflate.c:1186-1188 */
    }
}

rocm_inflate_action Z_INTERNAL PREFIX(archrocm_inflate)(PREFIX3(streamp) strm, int flush, int *ret) {
    struct inflate_state *state = (struct inflate_state *)strm->state;
    arch_inflate_state *a = &state->arch;

    if (flush == Z_BLOCK || flush == Z_TREES) {         /* stopping at block boundaries: software only */
        if (a->used) {
            *ret = Z_STREAM_ERROR;
            return ROCM_INFLATE_BREAK;
        }
        a->disabled = 1;
        return ROCM_INFLATE_SOFTWARE;
    }
    if (a->done) {
        drain(strm, a);
        if (a->out_pos < a->out_len) {                  /* next_out is full */
            *ret = Z_OK;
            return ROCM_INFLATE_BREAK;
        }
        if (state->wrap & 4) strm->adler = state->check = a->check;
        state->last = 1;
        state->mode = CHECK;                            /* the trailer is inflate()'s, inflate.c:1105-1147 */
        return ROCM_INFLATE_CONTINUE;
    }
    if (strm->avail_in == 0) {
        *ret = Z_OK;                                    /* inflate() turns "no progress" into Z_BUF_ERROR itself */
        return ROCM_INFLATE_BREAK;
    }
    /* gather; nothing is marked consumed for good until the decode says where the stream ends */
    const size_t taken = strm->avail_in;
    if (a->in_cap < a->in_len + taken) {
        const size_t want = (a->in_len + taken) * 2;
        uint8_t *n = (uint8_t *)arch_alloc(strm, want);
        if (!n) {
            if (a->used) { *ret = Z_MEM_ERROR; return ROCM_INFLATE_BREAK; }
            a->disabled = 1;
            return ROCM_INFLATE_SOFTWARE;
        }
        if (a->in_len) memcpy(n, a->in_buf, a->in_len);
        arch_free(strm, a->in_buf);
        a->in_buf = n;
        a->in_cap = want;
    }
    memcpy(a->in_buf + a->in_len, strm->next_in, taken);

    const int kind = !(state->wrap & 4) ? 0 : state->flags ? 2 : 1;             /* inflate_p.h:47-49: gzip -> crc32, zlib -> adler32 */
    uint32_t cv = state->check;
    const uint8_t *out = NULL;
    size_t out_len = 0, used = 0;
    const char *msg = NULL;
    const int rc = zng_rocm_hook_inflate(a->hook, a->in_buf, a->in_len + taken, kind, &cv, &out, &out_len, &used, &msg);
    if (rc == 1) {                                      /* Z_STREAM_END: `used` bytes were the deflate data */
        const size_t mine = used - a->in_len;           /* > 0: the end was not among the bytes of earlier calls */
        strm->next_in += mine;
        strm->avail_in -= (uint32_t)mine;
        a->in_len = 0;
        a->out = out;
        a->out_pos = 0;
        a->out_len = out_len;
        a->check = cv;
        a->used = a->done = 1;
        return PREFIX(archrocm_inflate)(strm, flush, ret);  /* deliver */
    }
    if (rc == -5) {                                     /* the stream does not end here: keep the bytes, ask for more */
        a->in_len += taken;
        strm->next_in += taken;
        strm->avail_in = 0;
        a->used = 1;
        *ret = Z_OK;
        return ROCM_INFLATE_BREAK;
    }
    if (rc == -3) {                                     /* Z_DATA_ERROR with the reference's text (inflate_p.h:130-134) */
        strm->msg = msg;
        state->mode = BAD;
        return ROCM_INFLATE_CONTINUE;
    }
    /* device trouble: nothing of this call has been consumed; a stream that never needed a second call goes to software */
    if (!a->used) {
        a->disabled = 1;
        return ROCM_INFLATE_SOFTWARE;
    }
    *ret = Z_MEM_ERROR;
    return ROCM_INFLATE_BREAK;
}

int Z_INTERNAL PREFIX(archrocm_inflate_set_dictionary)(PREFIX3(streamp) strm, const unsigned char *dictionary, unsigned dict_length) {
    struct inflate_state *state = (struct inflate_state *)strm->state;
    if (zng_rocm_hook_set_history(state->arch.hook, dictionary, dict_length) != ZNG_ROCM_OK) return Z_STREAM_ERROR;
    return Z_OK;
}

int Z_INTERNAL PREFIX(archrocm_inflate_get_dictionary)(PREFIX3(streamp) strm, unsigned char *dictionary, unsigned *dict_length) {
    struct inflate_state *state = (struct inflate_state *)strm->state;
    uint32_t len = 0;
    if (zng_rocm_hook_get_history(state->arch.hook, dictionary, &len) != ZNG_ROCM_OK) return Z_STREAM_ERROR;
    if (dict_length) *dict_length = len;
    return Z_OK;
}
