/* arch/rocm/rocm_deflate.c -- DEFLATE_HOOK backend of arch/rocm: deflate() keeps its API and its zng_stream
 * bookkeeping, the blocks are produced on the MI355X.  Structure and return protocol follow
 * arch/s390/dfltcc_deflate.c:109-300; what differs is granularity: a GPU launch pays from about a megabyte, so input is
 * gathered (as fill_window gathers it into the window, deflate.c:1241-1330) until ROCM_DEFLATE_BLOCK_BYTES are there or
 * the caller flushes, and a block's compressed bytes wait in out_buf until next_out has taken them (as pending_buf does,
 * deflate.c:786-812).  Every device block ends on a byte boundary, so s->bi_buf stays empty and what zlib-ng appends
 * itself (the sync marker of deflate.c:1064-1076, the trailer) lands correctly behind it.
 * Any failure of the device degrades to software and never surfaces: bytes already gathered leave as
 * stored blocks (RFC 1951 3.2.4), later input goes through deflate_* as if the hook did not exist. */
#ifdef ZNG_ROCM_STANDALONE_CHECK
#  include "zlibng_coarse_min.h"
#else
#  include "zbuild.h"
#  include "deflate.h"
#endif
#include <stdlib.h>
#include <string.h>
#include "zng_rocm.h"
#include "rocm_functions.h"
#include "rocm_deflate.h"

static void *arch_alloc(PREFIX3(streamp) strm, size_t n) {
    return strm->zalloc ? strm->zalloc(strm->opaque, 1, (unsigned)n) : malloc(n);
}
static void arch_free(PREFIX3(streamp) strm, void *p) {
    if (!p) return;
    if (strm->zfree) strm->zfree(strm->opaque, p);
    else free(p);
}
static int grow(PREFIX3(streamp) strm, uint8_t **buf, size_t *cap, size_t keep, size_t want) {
    if (*cap >= want) return 1;
    uint8_t *n = (uint8_t *)arch_alloc(strm, want);
    if (!n) return 0;
    if (keep) memcpy(n, *buf, keep);
    arch_free(strm, *buf);
    *buf = n;
    *cap = want;
    return 1;
}

void Z_INTERNAL PREFIX(archrocm_reset_deflate_state)(PREFIX3(streamp) strm) {      /* DEFLATE_RESET_KEEP_HOOK, deflate.c:567 */
    arch_deflate_state *a = &((deflate_state *)strm->state)->arch;
    a->in_len = a->out_pos = a->out_len = 0;
    a->used = a->finished = 0;
    if (a->hook && zng_rocm_hook_reset(a->hook) != ZNG_ROCM_OK) a->disabled = 1;
}

void Z_INTERNAL PREFIX(archrocm_deflate_end)(PREFIX3(streamp) strm) {              /* DEFLATE_END_HOOK, in deflateEnd() */
    arch_deflate_state *a = &((deflate_state *)strm->state)->arch;
    zng_rocm_hook_destroy(a->hook);
    arch_free(strm, a->in_buf);
    arch_free(strm, a->out_buf);
    memset(a, 0, sizeof *a);
}

/* what the device takes: every level, the default and filtered strategies (the others prescribe a block format:
 * deflate_huff / deflate_rle / Z_FIXED), the 32 KiB window its history is kept for */
static int params_ok(int level, unsigned w_bits, int strategy) {
    return level >= 0 && level <= 9 && w_bits == 15 && (strategy == Z_DEFAULT_STRATEGY || strategy == Z_FILTERED);
}

int Z_INTERNAL PREFIX(archrocm_can_deflate)(PREFIX3(streamp) strm) {
    deflate_state *s = (deflate_state *)strm->state;
    arch_deflate_state *a = &s->arch;
    if (a->disabled || !params_ok(s->level, s->w_bits, s->strategy)) return 0;
    if (!a->hook) {                                     /* first use: is there a device at all? */
        if (zng_rocm_device_count() <= 0 || zng_rocm_init(-1) != ZNG_ROCM_OK ||
            zng_rocm_hook_create(&a->hook, ROCM_DEFLATE_BLOCK_BYTES) != ZNG_ROCM_OK) {
            a->hook = NULL;
            a->disabled = 1;
            return 0;
        }
    }
    return 1;
}

static void drain(PREFIX3(streamp) strm, arch_deflate_state *a) {
    size_t n = a->out_len - a->out_pos;
    if (n > strm->avail_out) n = strm->avail_out;
    if (n) {
        memcpy(strm->next_out, a->out_buf + a->out_pos, n);
        strm->next_out += n;
        strm->avail_out -= (uint32_t)n;
        strm->total_out += n;
        a->out_pos += n;
    }
    if (a->out_pos == a->out_len) a->out_pos = a->out_len = 0;
}

/* software form of one block: stored blocks (RFC 1951 3.2.4; deflate_stored.c:27-186 writes the same bytes) */
static size_t stored_block(const uint8_t *in, size_t n, int final, uint8_t *out) {
    size_t o = 0;
    do {
        const size_t ln = n > 65535u ? 65535u : n;
        out[o++] = (uint8_t)((final && ln == n) ? 1 : 0);
        out[o++] = (uint8_t)ln;
        out[o++] = (uint8_t)(ln >> 8);
        out[o++] = (uint8_t)~ln;
        out[o++] = (uint8_t)(~ln >> 8);
        memcpy(out + o, in, ln);
        o += ln;
        in += ln;
        n -= ln;
    } while (n);
    return o;
}

/* the gathered input -> one block in out_buf.  Returns 0 only when memory for the output cannot be had. */
static int produce(PREFIX3(streamp) strm, deflate_state *s, arch_deflate_state *a, int final) {
    const size_t cap = zng_rocm_hook_deflate_bound(a->in_len) + 5 * (a->in_len / 65535u + 1);
    if (!grow(strm, &a->out_buf, &a->out_cap, 0, cap)) return 0;
    uint32_t cv = s->wrap == 2 ? s->crc_fold.value : strm->adler;
    size_t clen = 0;
    int rc = ZNG_ROCM_ENODEV;
    if (!a->disabled)
        rc = zng_rocm_hook_deflate_block(a->hook, s->level, a->in_buf, a->in_len, final ? 0u : ZNG_ROCM_BLOCK_NOT_FINAL,
                                         s->wrap, &cv, a->out_buf, a->out_cap, &clen);
    if (rc != ZNG_ROCM_OK) {                            /* degrade, never surface */
        a->disabled = 1;
        clen = stored_block(a->in_buf, a->in_len, final, a->out_buf);
        cv = s->wrap == 2 ? rocm_cpu_crc32(cv, a->in_buf, a->in_len) : s->wrap == 1 ? rocm_cpu_adler32(cv, a->in_buf, a->in_len) : cv;
    }
    if (s->wrap == 2) s->crc_fold.value = cv;           /* what DEFLATE_NEED_CHECKSUM = 0 leaves to us, deflate.c:1197-1212 */
    else if (s->wrap == 1) strm->adler = cv;
    a->out_pos = 0;
    a->out_len = clen;
    a->in_len = 0;
    a->used = 1;
    a->finished = final;
    return 1;
}

int Z_INTERNAL PREFIX(archrocm_deflate)(PREFIX3(streamp) strm, int flush, block_state *result) {
    deflate_state *s = (deflate_state *)strm->state;
    arch_deflate_state *a = &s->arch;
    /* a stream the device has begun is finished through this function even if the device has failed since
     * (`disabled` then makes produce() write stored blocks): deflate_* know nothing of the bytes gathered here */
    if (!a->used && a->in_len == 0 && !PREFIX(archrocm_can_deflate)(strm)) return 0;
    if (a->disabled && a->in_len == 0 && a->out_len == 0 && !a->finished) return 0;    /* nothing of ours in flight: software */

    drain(strm, a);                                     /* bytes of an earlier block first */
    if (a->out_len) {                                   /* next_out is full.  Never finish_started: deflate() would not come */
        *result = need_more;                            /* back here (deflate.c:1036 tests s->status), and what is left */
        return 1;                                       /* waits in OUR buffer, not in s->pending_buf */
    }
    if (a->finished) {
        *result = finish_done;
        return 1;
    }
    for (;;) {
        size_t take = ROCM_DEFLATE_BLOCK_BYTES - a->in_len;
        if (take > strm->avail_in) take = strm->avail_in;
        if (take) {
            if (!grow(strm, &a->in_buf, &a->in_cap, a->in_len, ROCM_DEFLATE_BLOCK_BYTES)) return a->used ? (*result = need_more, 1) : 0;
            memcpy(a->in_buf + a->in_len, strm->next_in, take);
            a->in_len += take;
            strm->next_in += take;
            strm->avail_in -= (uint32_t)take;
            strm->total_in += take;
        }
        const int closing = flush != Z_NO_FLUSH && strm->avail_in == 0;
        if (a->in_len < ROCM_DEFLATE_BLOCK_BYTES && !closing) {
            *result = need_more;                        /* all input taken, nothing to write yet */
            return 1;
        }
        const int final = flush == Z_FINISH && strm->avail_in == 0;
        if (a->in_len || final) {
            if (!produce(strm, s, a, final)) {
                *result = need_more;
                return 1;
            }
            drain(strm, a);
            if (a->out_len) {
                *result = need_more;
                return 1;
            }
        }
        if (strm->avail_in == 0) break;                 /* deflate() must use all input or all output */
    }
    if (flush == Z_FULL_FLUSH && !a->disabled && zng_rocm_hook_reset(a->hook) != ZNG_ROCM_OK) a->disabled = 1;   /* deflate.c:1073-1080 */
    *result = flush == Z_NO_FLUSH ? need_more : flush == Z_FINISH ? finish_done : block_done;
    return 1;
}

/* DEFLATE_PARAMS_HOOK (deflate.c:649): the level is read per block, so a change between settings the device takes needs
 * nothing; leaving them mid-stream would need the history back in the software window -- not supported */
int Z_INTERNAL PREFIX(archrocm_deflate_params)(PREFIX3(streamp) strm, int level, int strategy, int *flush) {
    deflate_state *s = (deflate_state *)strm->state;
    (void)flush;
    if (s->arch.used && !params_ok(level, s->w_bits, strategy)) return Z_STREAM_ERROR;
    return Z_OK;
}

int Z_INTERNAL PREFIX(archrocm_deflate_done)(PREFIX3(streamp) strm, int flush) {   /* DEFLATE_DONE, deflate.c:659 */
    const arch_deflate_state *a = &((deflate_state *)strm->state)->arch;
    (void)flush;
    return a->in_len == 0 && a->out_len == 0;
}

int Z_INTERNAL PREFIX(archrocm_deflate_set_dictionary)(PREFIX3(streamp) strm, const unsigned char *dictionary, unsigned dict_length) {
    deflate_state *s = (deflate_state *)strm->state;
    if (zng_rocm_hook_set_history(s->arch.hook, dictionary, dict_length) != ZNG_ROCM_OK) {
        s->arch.disabled = 1;
        return Z_STREAM_ERROR;
    }
    if (s->wrap == 1) strm->adler = rocm_cpu_adler32(strm->adler, dictionary, dict_length);     /* deflate.c:476-478 */
    return Z_OK;
}

int Z_INTERNAL PREFIX(archrocm_deflate_get_dictionary)(PREFIX3(streamp) strm, unsigned char *dictionary, unsigned *dict_length) {
    deflate_state *s = (deflate_state *)strm->state;
    uint32_t len = 0;
    if (zng_rocm_hook_get_history(s->arch.hook, dictionary, &len) != ZNG_ROCM_OK) return Z_STREAM_ERROR;
    if (dict_length) *dict_length = len;
    return Z_OK;
}

size_t Z_INTERNAL PREFIX(archrocm_deflate_bound)(size_t source_len) {              /* DEFLATE_BOUND_ADJUST_COMPLEN, deflate.c:473 */
    const size_t blocks = source_len / ROCM_DEFLATE_BLOCK_BYTES + 1;           /* + what a flush per call could add is the caller's */
    return source_len + blocks * (zng_rocm_hook_deflate_bound(ROCM_DEFLATE_BLOCK_BYTES) - ROCM_DEFLATE_BLOCK_BYTES) + 16;
}
