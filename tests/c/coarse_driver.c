/* tests/c/coarse_driver.c -- TEST: drives integration/arch/rocm/rocm_deflate.c and rocm_inflate.c the way deflate() and
 * inflate() drive an arch backend.  The two functions below restate ONLY the control flow around the hook macros
 * (deflate.c:826-846 pending bytes, :868-892 zlib header, :1036-1083 DEFLATE_HOOK and what follows each block_state,
 * :1091-1100 trailer; inflate.c:509-555 zlib header, :728 INFLATE_TYPEDO_HOOK inside the mode switch, :1105-1147 CHECK)
 * -- there is no software deflate / inflate here: where the reference would continue in software the driver reports
 * "fallback" and stops, which is what the no-GPU test expects.
 *   coarse_driver d <level> <wrap> <in_chunk> <out_chunk> <sync_every> <infile> <outfile>
 *   coarse_driver i <wrap> <in_chunk> <out_chunk> <infile> <outfile> <plaintext bytes expected>
 * prints "device <bytes in> <bytes out>" or "fallback". */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zng_rocm.h"
#include "zlibng_coarse_min.h"
#include "rocm_functions.h"
#include "rocm_deflate.h"
#include "rocm_inflate.h"

static uint32_t cpu_adler(uint32_t adler, const uint8_t *buf, size_t len) {
    uint32_t s1 = adler & 0xffff, s2 = (adler >> 16) & 0xffff;
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

/* ---- deflate() around DEFLATE_HOOK ------------------------------------------------------------------------ */
static uint8_t pend[32];
static unsigned npend, header_done, finish_state;

static void flush_pending(zng_stream *strm) {                   /* deflate.c:786-812 */
    unsigned n = npend < strm->avail_out ? npend : strm->avail_out;
    memcpy(strm->next_out, pend, n);
    memmove(pend, pend + n, npend - n);
    npend -= n;
    strm->next_out += n;
    strm->avail_out -= n;
    strm->total_out += n;
}

static int driver_deflate(zng_stream *strm, int flush) {
    deflate_state *s = strm->state;
    if (!header_done && s->wrap == 1) {                         /* deflate.c:868-892 */
        pend[npend++] = 0x78;
        pend[npend++] = 0x9c;
        strm->adler = 1;
    }
    header_done = 1;
    if (npend) {                                                /* deflate.c:826-846 */
        flush_pending(strm);
        if (strm->avail_out == 0) return Z_OK;
    }
    if (strm->avail_in != 0 || (flush != Z_NO_FLUSH && !finish_state)) {      /* deflate.c:1036 */
        block_state bstate;
        if (!DEFLATE_HOOK(strm, flush, &bstate)) return -100;   /* the reference would call deflate_* here */
        if (bstate == finish_started || bstate == finish_done) finish_state = 1;
        if (bstate == need_more || bstate == finish_started) return Z_OK;
        if (bstate == block_done) {
            if (flush != Z_PARTIAL_FLUSH && flush != Z_BLOCK) {     /* zng_tr_stored_block(s, NULL, 0, 0) with bi_valid == 0 */
                static const uint8_t marker[5] = {0x00, 0x00, 0x00, 0xff, 0xff};
                memcpy(pend + npend, marker, 5);
                npend += 5;
            }
            flush_pending(strm);
            if (strm->avail_out == 0) return Z_OK;
        }
    }
    if (flush != Z_FINISH) return Z_OK;
    if (s->wrap == 1) {                                         /* deflate.c:1091-1100 */
        pend[npend++] = (uint8_t)(strm->adler >> 24);
        pend[npend++] = (uint8_t)(strm->adler >> 16);
        pend[npend++] = (uint8_t)(strm->adler >> 8);
        pend[npend++] = (uint8_t)strm->adler;
        s->wrap = -1;
    }
    flush_pending(strm);
    return npend ? Z_OK : Z_STREAM_END;
}

/* ---- inflate() around INFLATE_TYPEDO_HOOK ----------------------------------------------------------------- */
#define RESTORE() do { } while (0)
#define LOAD() do { } while (0)
enum { D_HEAD = 1 };

static int driver_inflate(zng_stream *strm, int flush) {
    struct inflate_state *state = (struct inflate_state *)strm->state;
    int ret = Z_OK;
    const uint32_t in0 = strm->avail_in, out0 = strm->avail_out;
    for (;;) {
        switch ((int)state->mode) {
        case D_HEAD:                                            /* inflate.c:509-555, zlib wrapper only */
            if (state->wrap == 0) {
                state->mode = TYPEDO;
                break;
            }
            if (strm->avail_in < 2) goto inf_leave;
            if (((strm->next_in[0] << 8) + strm->next_in[1]) % 31 || (strm->next_in[0] & 0xf) != 8) {
                strm->msg = "incorrect header check";
                state->mode = BAD;
                break;
            }
            strm->next_in += 2;
            strm->avail_in -= 2;
            strm->adler = state->check = 1;
            state->mode = TYPEDO;
            break;
        case TYPEDO:
            INFLATE_TYPEDO_HOOK(strm, flush);
            return -100;                                        /* the reference would decode the block in software here */
        case CHECK:                                             /* inflate.c:1105-1147 */
            if (state->wrap) {
                if (strm->avail_in < 4) goto inf_leave;
                const uint32_t want = ((uint32_t)strm->next_in[0] << 24) | ((uint32_t)strm->next_in[1] << 16) |
                                      ((uint32_t)strm->next_in[2] << 8) | strm->next_in[3];
                strm->next_in += 4;
                strm->avail_in -= 4;
                if (want != state->check) {
                    strm->msg = "incorrect data check";
                    state->mode = BAD;
                    break;
                }
            }
            state->mode = DONE;
            break;
        case DONE:
            ret = Z_STREAM_END;
            goto inf_leave;
        case BAD:
            ret = Z_DATA_ERROR;
            goto inf_leave;
        default:
            return Z_STREAM_ERROR;
        }
    }
inf_leave:
    strm->total_in += in0 - strm->avail_in;                     /* inflate.c:1185-1188 */
    strm->total_out += out0 - strm->avail_out;
    if (((in0 == strm->avail_in && out0 == strm->avail_out) || flush == Z_FINISH) && ret == Z_OK) ret = Z_BUF_ERROR;
    return ret;
}

static uint8_t *read_file(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    *n = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *b = malloc(*n + 1);
    if (fread(b, 1, *n, f) != *n) return NULL;
    fclose(f);
    return b;
}

int main(int argc, char **argv) {
    if (argc < 8) return 2;
    rocm_remember_cpu_tier(cpu_adler, cpu_crc);
    zng_stream strm;
    memset(&strm, 0, sizeof strm);
    size_t n = 0, cap, produced = 0;
    uint8_t *out;
    FILE *fo;
    if (argv[1][0] == 'd') {
        if (argc < 9) return 2;
        deflate_state st;
        memset(&st, 0, sizeof st);
        st.strm = &strm;
        st.level = atoi(argv[2]);
        st.wrap = atoi(argv[3]);
        st.strategy = Z_DEFAULT_STRATEGY;
        st.w_bits = 15;
        strm.state = &st;
        const size_t in_chunk = (size_t)atol(argv[4]), out_chunk = (size_t)atol(argv[5]);
        const int sync_every = atoi(argv[6]);
        uint8_t *in = read_file(argv[7], &n);
        if (!in) return 2;
        cap = n + n / 4 + (n / in_chunk + 4) * 4096 + 65536 + out_chunk;
        out = malloc(cap);
        DEFLATE_RESET_KEEP_HOOK(&strm);
        size_t fed = 0;
        int calls = 0, rc = Z_OK;
        strm.next_in = in;
        while (rc != Z_STREAM_END) {
            int flush = Z_NO_FLUSH;
            if (strm.avail_in == 0) {
                if (fed < n) {
                    const size_t c = n - fed < in_chunk ? n - fed : in_chunk;
                    strm.next_in = in + fed;
                    strm.avail_in = (uint32_t)c;
                    fed += c;
                    ++calls;
                }
            }
            if (fed == n) flush = Z_FINISH;
            else if (sync_every && calls % sync_every == 0) flush = Z_SYNC_FLUSH;
            do {                                                /* the caller's loop: same flush until the input is used up */
                if (produced + out_chunk > cap) return 3;
                strm.next_out = out + produced;
                strm.avail_out = (uint32_t)out_chunk;
                rc = driver_deflate(&strm, flush);
                if (rc == -100) {
                    printf("fallback\n");
                    return 0;
                }
                if (rc < 0) return 4;
                produced += out_chunk - strm.avail_out;
            } while (rc != Z_STREAM_END && (strm.avail_out == 0 || (flush == Z_FINISH)));
            if (strm.avail_in != 0) return 5;                   /* deflate() must use all input or all output */
        }
        if (strm.total_in != n || strm.total_out != produced) return 6;
        if (!DEFLATE_DONE(&strm, Z_FINISH)) return 7;
        DEFLATE_END_HOOK(&strm);
    } else {
        struct inflate_state st;
        memset(&st, 0, sizeof st);
        st.strm = &strm;
        st.wrap = atoi(argv[2]) ? 5 : 0;                        /* inflate.h: bit 0 zlib, bit 2 validate the check value */
        st.wbits = 15;
        st.mode = (inflate_mode)D_HEAD;
        strm.state = (struct internal_state *)&st;
        const size_t in_chunk = (size_t)atol(argv[3]), out_chunk = (size_t)atol(argv[4]);
        uint8_t *in = read_file(argv[5], &n);
        if (!in) return 2;
        cap = (size_t)atol(argv[7]) + out_chunk;                /* expected plaintext size */
        out = malloc(cap);
        INFLATE_RESET_KEEP_HOOK(&strm);
        size_t fed = 0;
        int rc = Z_OK;
        while (rc != Z_STREAM_END) {
            if (strm.avail_in == 0 && fed < n) {
                const size_t c = n - fed < in_chunk ? n - fed : in_chunk;
                strm.next_in = in + fed;
                strm.avail_in = (uint32_t)c;
                fed += c;
            }
            if (produced + out_chunk > cap) return 3;
            strm.next_out = out + produced;
            strm.avail_out = (uint32_t)out_chunk;
            rc = driver_inflate(&strm, Z_NO_FLUSH);
            if (rc == -100) {
                printf("fallback\n");
                return 0;
            }
            if (rc == Z_DATA_ERROR) {
                printf("data error: %s\n", strm.msg ? strm.msg : "?");
                return 0;
            }
            if (rc < 0 && rc != Z_BUF_ERROR) return 4;
            if (rc == Z_BUF_ERROR && fed == n && strm.avail_in == 0 && strm.avail_out == out_chunk) return 8;   /* stuck */
            produced += out_chunk - strm.avail_out;
        }
        n = strm.total_in;
        if (strm.total_out != produced) return 6;
        INFLATE_END_HOOK(&strm);
    }
    fo = fopen(argv[1][0] == 'd' ? argv[8] : argv[6], "wb");
    if (!fo || fwrite(out, 1, produced, fo) != produced) return 9;
    fclose(fo);
    printf("device %zu %zu\n", n, produced);
    if (getenv("COARSE_DRIVER_PARTS"))              /* how many parts the device decoded the last member in (0: host decoder) */
        printf("parts %d\n", zng_rocm_inflate_large_last_parts());
    return 0;
}
