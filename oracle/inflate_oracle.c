/* inflate_oracle.c -- CPU restatement of raw inflate (RFC 1951) with zlib-ng's error taxonomy.
 * TEST INFRASTRUCTURE ONLY (see zng_oracle.h).
 *
 * Deliberately the SIMPLEST correct decoder (bit-serial canonical Huffman decoding, byte-serial
 * LZ77 copies), so that it is an independent check of the table-driven product decoder.  What it
 * mirrors from /root/reference (zlib-ng 2.2.2) is behaviour, in this order of checks:
 *   inflate.c:735-757   block header, "invalid block type"
 *   inflate.c:759-799   stored blocks, "invalid stored block lengths"
 *   inflate.c:801-816   "too many length or distance symbols"
 *   inflate.c:818-836   code-length code, "invalid code lengths set"
 *   inflate.c:838-895   run-length decoding of the lengths, "invalid bit length repeat",
 *                       "invalid code -- missing end-of-block"
 *   inflate.c:897-917   "invalid literal/lengths set", "invalid distances set"
 *   inftrees.c:108-131  what makes a length set invalid (over-subscribed; incomplete unless it is a
 *                       single 1-bit code of a literal/length or distance alphabet); an EMPTY set is
 *                       accepted and only fails when a code is actually needed
 *   inffast_tpl.h:151-298 / inflate.c:938-1099  symbol loop: "invalid literal/length code",
 *                       "invalid distance code", "invalid distance too far back"
 *   inftrees.c:52-65    length / distance base and extra-bit tables (RFC 1951 section 3.2.5)
 * One-shot semantics: all input present, one output buffer, so "too far back" means a distance
 * larger than the number of bytes produced so far (inffast_tpl.h:198-226 with whave == 0).
 */
#include <stdlib.h>
#include <string.h>
#include "zng_oracle.h"

enum { MAXBITS = 15, MAXL = 288, MAXD = 32 };

typedef struct {
    const uint8_t *in;
    size_t in_len, in_pos;
    uint32_t bitbuf;
    int bitcnt;
    int starved;                /* ran out of input */
} bitreader;

static int getbits(bitreader *br, int need) {
    uint32_t val = br->bitbuf;
    while (br->bitcnt < need) {
        if (br->in_pos == br->in_len) {
            br->starved = 1;
            return 0;
        }
        val |= (uint32_t)br->in[br->in_pos++] << br->bitcnt;
        br->bitcnt += 8;
    }
    br->bitbuf = val >> need;
    br->bitcnt -= need;
    return (int)(val & ((1u << need) - 1));
}

typedef struct {
    uint16_t count[MAXBITS + 1];
    uint16_t symbol[MAXL];
    int empty;
    int incomplete;             /* one code of length 1, the other 1-bit pattern unused */
} huff;

/* returns 0 ok, -1 invalid per inftrees.c:108-131 (`is_codes` = the CODES type) */
static int build(huff *h, const uint16_t *lens, int n, int is_codes) {
    int offs[MAXBITS + 1];
    memset(h->count, 0, sizeof(h->count));
    for (int s = 0; s < n; s++)
        h->count[lens[s]]++;
    int max = MAXBITS;
    while (max >= 1 && h->count[max] == 0)
        max--;
    h->empty = (max == 0);
    h->incomplete = 0;
    if (h->empty)
        return 0;                                   /* inftrees.c:114-122 */
    int left = 1;
    for (int len = 1; len <= MAXBITS; len++) {
        left <<= 1;
        left -= h->count[len];
        if (left < 0)
            return -1;                              /* over-subscribed */
    }
    if (left > 0 && (is_codes || max != 1))
        return -1;                                  /* incomplete */
    h->incomplete = left > 0 && !h->empty;
    offs[1] = 0;
    for (int len = 1; len < MAXBITS; len++)
        offs[len + 1] = offs[len] + h->count[len];
    for (int s = 0; s < n; s++)
        if (lens[s])
            h->symbol[offs[lens[s]]++] = (uint16_t)s;
    return 0;
}

/* canonical decode, one bit at a time; -1 = no such code, -2 = out of input.
 * An empty set, and the one incomplete set inflate_table lets through (a single code of length 1), are one-bit tables in
 * the reference whose unused entries read {op 64, bits 1} (inftrees.c:114-122, :286-293): the invalid code is seen after
 * ONE bit -- with more input missing that is still a data error, not a request for input (inflate.c:990-1003). */
static int decode(bitreader *br, const huff *h) {
    int code = 0, first = 0, index = 0;
    if (h->empty || h->incomplete) {
        int bit = getbits(br, 1);
        if (br->starved)
            return -2;
        return (h->empty || bit) ? -1 : h->symbol[0];
    }
    for (int len = 1; len <= MAXBITS; len++) {
        code |= getbits(br, 1);
        if (br->starved)
            return -2;
        int count = h->count[len];
        if (code - count < first)
            return h->symbol[index + (code - first)];
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
                                      35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint16_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                       3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193,
                                       257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                       8193, 12289, 16385, 24577};
static const uint16_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6,
                                        7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

/* test aid: where the blocks of the next oracle_inflate_raw* call start (bit position of BFINAL) and their BTYPE */
static uint64_t *trace_bits;
static uint8_t *trace_types;
static size_t trace_cap, trace_n;
void oracle_inflate_trace_blocks(uint64_t *bits, uint8_t *types, size_t cap) {
    trace_bits = bits;
    trace_types = types;
    trace_cap = cap;
    trace_n = 0;
}
size_t oracle_inflate_traced_blocks(void) { return trace_n; }

#define FAIL(m) do { res->status = ORACLE_Z_DATA_ERROR; res->msg = (m); goto done; } while (0)
#define STARVED() do { res->status = ORACLE_Z_BUF_ERROR; res->msg = "input ended before the final block"; goto done; } while (0)

/* `dst` holds `start` bytes of history (a preset dictionary: inflateSetDictionary, inflate.c:1214-1261, makes them
 * the window, whave = start) in front of where the output goes; dst_cap counts from dst[0]. */
static int inflate_core(const uint8_t *src, size_t src_len, uint8_t *dst, size_t start, size_t dst_cap,
                        oracle_inflate_result *res) {
    bitreader br = { src, src_len, 0, 0, 0, 0 };
    size_t out = start;
    huff lencode, distcode, clcode;
    uint16_t lens[MAXL + MAXD];
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int last;

    res->status = ORACLE_Z_OK;
    res->msg = "";
    do {
        const uint64_t block_bit = 8ull * br.in_pos - (uint64_t)br.bitcnt;
        last = getbits(&br, 1);
        int type = getbits(&br, 2);
        if (br.starved) STARVED();
        if (trace_bits) {
            if (trace_n < trace_cap) {
                trace_bits[trace_n] = block_bit;
                trace_types[trace_n] = (uint8_t)type;
            }
            trace_n++;
        }
        if (type == 3) FAIL("invalid block type");
        if (type == 0) {
            br.bitbuf = 0;                              /* BYTEBITS, inflate.c:761 */
            br.bitcnt = 0;
            if (br.in_len - br.in_pos < 4) { br.in_pos = br.in_len; STARVED(); }
            unsigned len = src[br.in_pos] | (src[br.in_pos + 1] << 8);
            unsigned nlen = src[br.in_pos + 2] | (src[br.in_pos + 3] << 8);
            br.in_pos += 4;
            if (len != (nlen ^ 0xffffu)) FAIL("invalid stored block lengths");
            size_t avail = br.in_len - br.in_pos;
            size_t n = len < avail ? len : avail;
            if (out + n > dst_cap) { res->status = ORACLE_Z_BUF_ERROR; res->msg = "output buffer full"; goto done; }
            memcpy(dst + out, src + br.in_pos, n);
            out += n;
            br.in_pos += n;
            if (n < len) STARVED();
            continue;
        }
        if (type == 1) {
            /* fixed code, RFC 1951 3.2.6 (inffixed_tbl.h is the table form of this) */
            int s = 0;
            for (; s < 144; s++) lens[s] = 8;
            for (; s < 256; s++) lens[s] = 9;
            for (; s < 280; s++) lens[s] = 7;
            for (; s < 288; s++) lens[s] = 8;
            build(&lencode, lens, 288, 0);
            for (s = 0; s < 32; s++) lens[s] = 5;
            build(&distcode, lens, 32, 0);
        } else {
            int nlen = getbits(&br, 5) + 257;
            int ndist = getbits(&br, 5) + 1;
            int ncode = getbits(&br, 4) + 4;
            if (br.starved) STARVED();
            if (nlen > 286 || ndist > 30) FAIL("too many length or distance symbols");
            uint16_t cl[19];
            memset(cl, 0, sizeof(cl));
            for (int i = 0; i < ncode; i++) {
                cl[order[i]] = (uint16_t)getbits(&br, 3);
                if (br.starved) STARVED();
            }
            if (build(&clcode, cl, 19, 1)) FAIL("invalid code lengths set");
            int have = 0;
            while (have < nlen + ndist) {
                int sym;
                if (clcode.empty) {
                    /* inftrees.c:114-122 + inflate.c:841-849: an empty code-length code decodes every
                     * entry as length 0, one bit each; the block then fails the end-of-block check */
                    (void)getbits(&br, 1);
                    sym = br.starved ? -2 : 0;
                } else {
                    sym = decode(&br, &clcode);
                }
                if (sym == -2) STARVED();
                /* the CODES table is always complete (is_codes), so sym >= 0 here unless it is empty */
                if (sym < 0) FAIL("invalid code lengths set");
                if (sym < 16) {
                    lens[have++] = (uint16_t)sym;
                } else {
                    int rep, val = 0;
                    if (sym == 16) {
                        rep = 3 + getbits(&br, 2);
                        if (br.starved) STARVED();
                        if (have == 0) FAIL("invalid bit length repeat");
                        val = lens[have - 1];
                    } else if (sym == 17) {
                        rep = 3 + getbits(&br, 3);
                    } else {
                        rep = 11 + getbits(&br, 7);
                    }
                    if (br.starved) STARVED();
                    if (have + rep > nlen + ndist) FAIL("invalid bit length repeat");
                    while (rep--) lens[have++] = (uint16_t)val;
                }
            }
            if (lens[256] == 0) FAIL("invalid code -- missing end-of-block");
            if (build(&lencode, lens, nlen, 0)) FAIL("invalid literal/lengths set");
            if (build(&distcode, lens + nlen, ndist, 0)) FAIL("invalid distances set");
        }
        for (;;) {
            int sym = decode(&br, &lencode);
            if (sym == -2) STARVED();
            if (sym < 0 || sym > 285) FAIL("invalid literal/length code");
            if (sym < 256) {
                if (out == dst_cap) { res->status = ORACLE_Z_BUF_ERROR; res->msg = "output buffer full"; goto done; }
                dst[out++] = (uint8_t)sym;
                continue;
            }
            if (sym == 256)
                break;
            sym -= 257;
            unsigned len = len_base[sym] + (unsigned)getbits(&br, len_extra[sym]);
            if (br.starved) STARVED();
            int ds = decode(&br, &distcode);
            if (ds == -2) STARVED();
            if (ds < 0 || ds > 29) FAIL("invalid distance code");
            unsigned dist = dist_base[ds] + (unsigned)getbits(&br, dist_extra[ds]);
            if (br.starved) STARVED();
            if (dist > out) FAIL("invalid distance too far back");
            if (out + len > dst_cap) { res->status = ORACLE_Z_BUF_ERROR; res->msg = "output buffer full"; goto done; }
            for (unsigned i = 0; i < len; i++, out++)
                dst[out] = dst[out - dist];
        }
    } while (!last);
    res->status = ORACLE_Z_STREAM_END;
done:
    res->out_len = out - start;
    res->in_used = br.in_pos - (size_t)(br.bitcnt >> 3);   /* whole unread bytes are handed back */
    return res->status;
}

int oracle_inflate_raw(const uint8_t *src, size_t src_len, uint8_t *dst, size_t dst_cap,
                       oracle_inflate_result *res) {
    return inflate_core(src, src_len, dst, 0, dst_cap, res);
}

/* raw inflate after inflateSetDictionary(dict, dict_len) (inflate.c:1214-1261; only the last 32 KiB count,
 * inflate.c:325-378 updatewindow): distances may reach dict_len bytes in front of the output */
int oracle_inflate_raw_dict(const uint8_t *src, size_t src_len, const uint8_t *dict, size_t dict_len,
                            uint8_t *dst, size_t dst_cap, oracle_inflate_result *res) {
    if (dict_len > 32768) {
        dict += dict_len - 32768;
        dict_len = 32768;
    }
    uint8_t *work = (uint8_t *)malloc(dict_len + dst_cap + 1);
    if (!work) {
        res->status = -4;
        res->msg = "out of memory";
        res->out_len = res->in_used = 0;
        return res->status;
    }
    memcpy(work, dict, dict_len);
    int rc = inflate_core(src, src_len, work, dict_len, dict_len + dst_cap, res);
    memcpy(dst, work + dict_len, res->out_len);
    free(work);
    return rc;
}
