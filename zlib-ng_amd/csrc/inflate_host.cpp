// inflate_host.cpp -- host stage of the arch/rocm inflate path: the sequential DEFLATE bitstream
// (RFC 1951) is decoded on the host into a TOKEN stream; the device resolves every copy
// (inflate_resolve.hip).  This is the split of zlib-ng's inflate_fast (inffast_tpl.h:53-318) at the
// point where a decoded symbol becomes a store: the table-driven decode loop (:151-226) stays here,
// the literal stores and CHUNKCOPY/CHUNKMEMSET match copies (:155-171, :228-279) become device work.
//
// Behaviour mirrored from the reference (error texts are the reference's strm->msg strings):
//   block header / stored / dynamic header checks   inflate.c:735-917
//   code validity rules                             inftrees.c:108-131
//   length/distance base+extra tables               inftrees.c:52-65 (RFC 1951 3.2.5)
//   64-bit refill, root tables of 10 (lit/len) and 9 (dist) bits   inffast_tpl.h:44-46,142-147; inflate.c:899,909
// One-shot contract: the whole raw stream is in `src`; "invalid distance too far back" = distance
// larger than the bytes produced so far plus the `window_len` bytes of history the caller says precede the
// stream (inffast_tpl.h:198-226; whave after inflateSetDictionary, inflate.c:1214-1261).
#ifndef _GNU_SOURCE
#define _GNU_SOURCE                  // memmem
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/zng_rocm.h"
#include "inflate_threads.h"

namespace {

enum { Z_OK_ = 0, Z_STREAM_END_ = 1, Z_DATA_ERROR_ = -3, Z_MEM_ERROR_ = -4, Z_BUF_ERROR_ = -5 };

// decode-table entry (4 bytes, like the reference's `code`, inftrees.h:27-31, but own encoding)
struct Ent {
    uint8_t  kind;      // K_* below
    uint8_t  bits;      // code bits to drop (root entries: <= root; sub entries: total - root)
    uint16_t val;       // literal byte | base value | sub-table offset
};
enum : uint8_t {
    K_LIT = 0,          // val = byte
    K_BASE = 16,        // | extra-bit count (0..13): val = base length / distance
    K_EOB = 32,
    K_BAD = 64,
    K_LINK = 128,       // | sub-table index bits: val = offset of the sub-table
};

const uint16_t kLenBase[31] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
                               35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258, 0, 0};
const uint8_t kLenExtra[31] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                               3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0};
const uint16_t kDistBase[32] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193,
                                257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                8193, 12289, 16385, 24577, 0, 0};
const uint8_t kDistExtra[32] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6,
                                7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 0, 0};

enum TableKind { T_CODES, T_LENS, T_DISTS };

inline unsigned bitrev(unsigned code, int len) {
    unsigned r = 0;
    for (int i = 0; i < len; ++i) {
        r = (r << 1) | (code & 1u);
        code >>= 1;
    }
    return r;
}

// Builds a two-level table.  Returns 0, or -1 for an invalid length set (inftrees.c:108-131).
// `table` needs room for (1<<root) + sub-tables: 1334 entries suffice for root 10 / 286 symbols
// and 402 for root 9 / 30 symbols worst case sized generously below.
int build_table(TableKind kind, const uint16_t *lens, int n, int root, Ent *table, int *root_out) {
    int count[16] = {0};
    for (int s = 0; s < n; ++s) count[lens[s]]++;
    int max = 15;
    while (max >= 1 && count[max] == 0) --max;
    if (max == 0) {                                   // no codes: every lookup is an error
        table[0] = table[1] = Ent{K_BAD, 1, 0};
        *root_out = 1;
        return 0;
    }
    int left = 1;
    for (int len = 1; len <= 15; ++len) {
        left = (left << 1) - count[len];
        if (left < 0) return -1;                      // over-subscribed
    }
    if (left > 0 && (kind == T_CODES || max != 1)) return -1;   // incomplete
    if (root > max) root = max;
    int min = 1;
    while (min < max && count[min] == 0) ++min;
    if (root < min) root = min;
    *root_out = root;

    const int root_size = 1 << root;
    for (int i = 0; i < root_size; ++i) table[i] = Ent{K_BAD, (uint8_t)root, 0};

    // canonical first code of each length
    unsigned next_code[16];
    unsigned code = 0;
    count[0] = 0;
    for (int len = 1; len <= 15; ++len) {
        code = (code + (unsigned)count[len - 1]) << 1;
        next_code[len] = code;
    }
    // pass 1: how deep is each root prefix (longest code sharing it)
    static thread_local uint8_t depth[1 << 10];
    memset(depth, 0, (size_t)root_size);
    unsigned nc[16];
    memcpy(nc, next_code, sizeof(nc));
    for (int s = 0; s < n; ++s) {
        int len = lens[s];
        if (len > root) {
            unsigned rev = bitrev(nc[len], len);
            unsigned pre = rev & (unsigned)(root_size - 1);
            if (depth[pre] < len - root) depth[pre] = (uint8_t)(len - root);
        }
        if (len) nc[len]++;
    }
    int used = root_size;
    for (int pre = 0; pre < root_size; ++pre) {
        if (depth[pre]) {
            table[pre] = Ent{(uint8_t)(K_LINK | depth[pre]), (uint8_t)root, (uint16_t)used};
            int sz = 1 << depth[pre];
            for (int i = 0; i < sz; ++i) table[used + i] = Ent{K_BAD, depth[pre], 0};
            used += sz;
        }
    }
    // pass 2: fill
    memcpy(nc, next_code, sizeof(nc));
    for (int s = 0; s < n; ++s) {
        int len = lens[s];
        if (!len) continue;
        unsigned rev = bitrev(nc[len]++, len);
        Ent e;
        if (kind == T_CODES) {
            e = Ent{K_LIT, 0, (uint16_t)s};
        } else if (kind == T_LENS) {
            if (s < 256) e = Ent{K_LIT, 0, (uint16_t)s};
            else if (s == 256) e = Ent{K_EOB, 0, 0};
            else if (s - 257 < 29) e = Ent{(uint8_t)(K_BASE | kLenExtra[s - 257]), 0, kLenBase[s - 257]};
            else e = Ent{K_BAD, 0, 0};                            // 286, 287: inftrees.c lext 203/77 = invalid
        } else {
            if (s < 30) e = Ent{(uint8_t)(K_BASE | kDistExtra[s]), 0, kDistBase[s]};
            else e = Ent{K_BAD, 0, 0};
        }
        if (len <= root) {
            e.bits = (uint8_t)len;
            for (unsigned i = rev; i < (unsigned)root_size; i += 1u << len) table[i] = e;
        } else {
            unsigned pre = rev & (unsigned)(root_size - 1);
            int sub_bits = table[pre].kind & 15;
            Ent *sub = table + table[pre].val;
            e.bits = (uint8_t)(len - root);
            for (unsigned i = rev >> root; i < (1u << sub_bits); i += 1u << (len - root)) sub[i] = e;
        }
    }
    return 0;
}

// (re)allocation of the three token-stream arrays: plain realloc, or the pinned-memory form the batched inflate
// passes in (its workers keep their arrays across streams and hand them to the DMA engine without a staging copy)
typedef void *(*ReallocFn)(void *old, size_t old_bytes, size_t new_bytes);
static void *plain_realloc(void *old, size_t, size_t new_bytes) { return realloc(old, new_bytes); }

struct Out {
    zng_rocm_inflate_tokens *t;
    ReallocFn re;
    size_t tok_cap, lit_cap, seg_cap;
    uint64_t out_pos;           // bytes the stream has produced
    uint64_t seg_out;           // out_pos at which the current segment started
    uint32_t run;               // literals accumulated since the last token
    bool oom;
};

constexpr uint64_t kSegTarget = 32u << 10;     // a segment closes once it holds >= 32 KiB of output
constexpr uint32_t kMaxRun = 0x7fffffffu;

inline void grow(ReallocFn re, void **p, size_t *cap, size_t need, size_t elt, bool *oom) {
    if (need <= *cap) return;
    size_t ncap = *cap ? *cap : 4096;
    while (ncap < need) ncap += ncap >> 1;
    void *q = re(*p, *cap * elt, ncap * elt);
    if (!q) { *oom = true; return; }
    *p = q;
    *cap = ncap;
}

inline void push_token(Out &o, uint32_t tok) {
    zng_rocm_inflate_tokens *t = o.t;
    if (t->ntokens + 1 > o.tok_cap) grow(o.re, (void **)&t->tokens, &o.tok_cap, t->ntokens + 1, 4, &o.oom);
    if (!o.oom) t->tokens[t->ntokens++] = tok;
}

inline void flush_run(Out &o) {
    if (o.run) {
        push_token(o, o.run);
        o.run = 0;
    }
}

inline void maybe_new_segment(Out &o) {
    // only called at a token boundary (run == 0)
    if (o.out_pos - o.seg_out >= kSegTarget) {
        zng_rocm_inflate_tokens *t = o.t;
        if ((t->nsegs + 2) * 3 > o.seg_cap) grow(o.re, (void **)&t->segs, &o.seg_cap, (t->nsegs + 2) * 3, 8, &o.oom);
        if (o.oom) return;
        uint64_t *s = t->segs + 3 * t->nsegs++;
        s[0] = t->ntokens;
        s[1] = o.out_pos;
        s[2] = t->nliterals;
        o.seg_out = o.out_pos;
    }
}

inline void reserve_literals(Out &o, size_t more) {
    zng_rocm_inflate_tokens *t = o.t;
    if (t->nliterals + more > o.lit_cap) grow(o.re, (void **)&t->literals, &o.lit_cap, t->nliterals + more, 1, &o.oom);
}

struct Bits {
    const uint8_t *next, *end;
    uint64_t hold;
    unsigned cnt;
    // make at least `n` (<= 32) bits available; false if the input is exhausted
    inline bool need(unsigned n) {
        while (cnt < n) {
            if (next == end) return false;
            hold |= (uint64_t)*next++ << cnt;
            cnt += 8;
        }
        return true;
    }
    inline void refill_fast() {          // caller guarantees >= 8 readable bytes (inffast_tpl.h:142-147)
        uint64_t w;
        memcpy(&w, next, 8);
        hold |= w << cnt;
        unsigned adv = (63 - cnt) >> 3;
        next += adv;
        cnt += adv << 3;
    }
    inline unsigned peek(unsigned n) const { return (unsigned)(hold & ((1ull << n) - 1)); }
    inline void drop(unsigned n) { hold >>= n; cnt -= n; }
};

#define FAIL(m) do { t->status = Z_DATA_ERROR_; t->msg = (m); goto done; } while (0)
#define STARVE() do { t->status = Z_BUF_ERROR_; t->msg = "input ended before the final block"; goto done; } while (0)

}  // namespace

// Control block of a PARTIAL decode (the parts of the multi-threaded single-stream inflate, inflate_threads.cpp):
// start in the middle of the stream, at the bit where a block header begins (start_bit), and stop when a block ends
// exactly on one of the given ascending bit positions (stops: where other threads' decodes start).  On return
// end_bit = first bit behind the last block decoded, max_reach = max over all matches of (distance - bytes produced
// before the match), hit_stop = 1 when it stopped on a stop position (status Z_OK: the stream goes on there).
namespace {

// caps: capacities (in elements: tokens, literal bytes, seg words) of the arrays `t` already owns; updated on return
int decode_stream(const uint8_t *src, size_t src_len, uint64_t window_len, zng_rocm_inflate_tokens *t,
                  ReallocFn re = plain_realloc, size_t *caps = nullptr, ZrDecodeCtl *ctl = nullptr) {
    Out o;
    o.t = t;
    o.re = re;
    o.tok_cap = caps ? caps[0] : 0;
    o.lit_cap = caps ? caps[1] : 0;
    o.seg_cap = caps ? caps[2] : 0;
    struct SaveCaps {                    // every exit path hands the capacities back
        Out &o; size_t *caps;
        ~SaveCaps() { if (caps) { caps[0] = o.tok_cap; caps[1] = o.lit_cap; caps[2] = o.seg_cap; } }
    } save{o, caps};
    o.out_pos = o.seg_out = 0;
    o.run = 0;
    o.oom = false;
    Bits b{src, src + src_len, 0, 0};
    size_t stop_i = 0;
    uint64_t max_reach = 0;
    const size_t expect = ctl && ctl->size_hint ? ctl->size_hint : src_len;
    if (ctl) {
        ctl->hit_stop = 0;
        b.next = src + (ctl->start_bit >> 3);
        if (b.next > b.end) b.next = b.end;
        while (stop_i < ctl->nstops && ctl->stops[stop_i] <= ctl->start_bit) ++stop_i;
    }
    static thread_local Ent lentab[(1 << 10) + 1024], disttab[(1 << 9) + 512], cltab[(1 << 7) + 128];
    int lenroot = 0, distroot = 0, clroot = 0;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint16_t lens[320];
    bool last = false;

    // segment 0
    grow(o.re, (void **)&t->segs, &o.seg_cap, 6, 8, &o.oom);
    if (o.oom) goto oom;
    t->segs[0] = t->segs[1] = t->segs[2] = 0;
    t->nsegs = 1;
    grow(o.re, (void **)&t->tokens, &o.tok_cap, expect / 2 + 1024, 4, &o.oom);
    grow(o.re, (void **)&t->literals, &o.lit_cap, expect + 4096, 1, &o.oom);
    if (o.oom) goto oom;
    t->status = Z_OK_;
    t->msg = "";
    if (ctl && (ctl->start_bit & 7)) {
        if (!b.need((unsigned)(ctl->start_bit & 7))) STARVE();
        b.drop((unsigned)(ctl->start_bit & 7));
    }

    while (!last) {
        if (ctl && ctl->nstops) {
            // a block has just ended (or none has begun): is this where another thread's decode starts?
            const uint64_t pos = 8ull * (uint64_t)(b.next - src) - b.cnt;
            while (stop_i < ctl->nstops && ctl->stops[stop_i] < pos) ++stop_i;
            if (pos != ctl->start_bit && stop_i < ctl->nstops && ctl->stops[stop_i] == pos) {
                ctl->hit_stop = 1;
                goto done;                   // status stays Z_OK: the stream goes on
            }
        }
        if (!b.need(3)) STARVE();
        last = b.peek(1);
        b.drop(1);
        unsigned type = b.peek(2);
        b.drop(2);
        if (type == 3) FAIL("invalid block type");
        if (type == 0) {
            b.drop(b.cnt & 7);                                   // BYTEBITS (inflate.c:761)
            if (!b.need(32)) STARVE();
            unsigned len = b.peek(16);
            unsigned nlen = (unsigned)((b.hold >> 16) & 0xffff);
            if (len != (nlen ^ 0xffffu)) FAIL("invalid stored block lengths");
            b.drop(32);
            // whole bytes still sitting in the bit buffer go back to the byte stream
            b.next -= b.cnt >> 3;
            b.hold = 0;
            b.cnt = 0;
            size_t avail = (size_t)(b.end - b.next);
            size_t n = len < avail ? len : avail;
            // stored bytes are literals of the token stream
            size_t done_n = 0;
            while (done_n < n && !o.oom) {
                size_t room = (size_t)(kSegTarget - (o.out_pos - o.seg_out));
                if (room == 0) {
                    flush_run(o);
                    maybe_new_segment(o);
                    continue;
                }
                size_t piece = n - done_n < room ? n - done_n : room;
                if (piece > kMaxRun - o.run) piece = kMaxRun - o.run;
                reserve_literals(o, piece);
                if (o.oom) break;
                memcpy(t->literals + t->nliterals, b.next + done_n, piece);
                t->nliterals += piece;
                o.run += (uint32_t)piece;
                o.out_pos += piece;
                done_n += piece;
            }
            if (o.oom) goto oom;
            b.next += n;
            if (n < len) STARVE();
            continue;
        }
        if (type == 1) {
            int s = 0;
            for (; s < 144; ++s) lens[s] = 8;                     // RFC 1951 3.2.6
            for (; s < 256; ++s) lens[s] = 9;
            for (; s < 280; ++s) lens[s] = 7;
            for (; s < 288; ++s) lens[s] = 8;
            build_table(T_LENS, lens, 288, 9, lentab, &lenroot);
            for (s = 0; s < 32; ++s) lens[s] = 5;
            build_table(T_DISTS, lens, 32, 5, disttab, &distroot);
        } else {
            if (!b.need(14)) STARVE();
            unsigned nlen = b.peek(5) + 257; b.drop(5);
            unsigned ndist = b.peek(5) + 1;  b.drop(5);
            unsigned ncode = b.peek(4) + 4;  b.drop(4);
            if (nlen > 286 || ndist > 30) FAIL("too many length or distance symbols");
            uint16_t cl[19] = {0};
            for (unsigned i = 0; i < ncode; ++i) {
                if (!b.need(3)) STARVE();
                cl[order[i]] = (uint16_t)b.peek(3);
                b.drop(3);
            }
            if (build_table(T_CODES, cl, 19, 7, cltab, &clroot)) FAIL("invalid code lengths set");
            unsigned have = 0;
            while (have < nlen + ndist) {
                Ent e;
                for (;;) {                                         // inflate.c:841-845
                    e = cltab[b.peek((unsigned)clroot)];
                    if (e.bits <= b.cnt) break;
                    if (b.next == b.end) STARVE();
                    b.hold |= (uint64_t)*b.next++ << b.cnt;
                    b.cnt += 8;
                }
                // an empty code-length code yields K_BAD entries of 1 bit with val 0: every entry reads
                // as length 0 (inftrees.c:114-122 + inflate.c:846-849)
                unsigned sym = e.val;
                if (sym < 16) {
                    b.drop(e.bits);
                    lens[have++] = (uint16_t)sym;
                    continue;
                }
                unsigned rep, val = 0;
                if (sym == 16) {
                    if (!b.need(e.bits + 2u)) STARVE();
                    b.drop(e.bits);
                    if (have == 0) FAIL("invalid bit length repeat");
                    val = lens[have - 1];
                    rep = 3 + b.peek(2);
                    b.drop(2);
                } else if (sym == 17) {
                    if (!b.need(e.bits + 3u)) STARVE();
                    b.drop(e.bits);
                    rep = 3 + b.peek(3);
                    b.drop(3);
                } else {
                    if (!b.need(e.bits + 7u)) STARVE();
                    b.drop(e.bits);
                    rep = 11 + b.peek(7);
                    b.drop(7);
                }
                if (have + rep > nlen + ndist) FAIL("invalid bit length repeat");
                while (rep--) lens[have++] = (uint16_t)val;
            }
            if (lens[256] == 0) FAIL("invalid code -- missing end-of-block");
            if (build_table(T_LENS, lens, (int)nlen, 10, lentab, &lenroot)) FAIL("invalid literal/lengths set");
            if (build_table(T_DISTS, lens + nlen, (int)ndist, 9, disttab, &distroot)) FAIL("invalid distances set");
        }

        // ---- symbol loop (the decode half of inflate_fast) ----------------------------------
        for (;;) {
            if (o.oom) goto oom;
            reserve_literals(o, 1);
            if (o.oom) goto oom;
            // a symbol needs at most 15+5 + 15+13 = 48 bits
            if ((size_t)(b.end - b.next) >= 8) {
                b.refill_fast();
            } else {
                (void)b.need(48);                   // best effort near the end; checked below
            }
            Ent e = lentab[b.peek((unsigned)lenroot)];
            if (e.kind & K_LINK) {
                unsigned sub = e.kind & 15;
                unsigned idx = (unsigned)((b.hold >> e.bits) & ((1u << sub) - 1));
                Ent e2 = lentab[e.val + idx];
                if (b.cnt < (unsigned)e.bits + e2.bits) STARVE();
                b.drop(e.bits);
                e = e2;
            }
            if (b.cnt < e.bits) STARVE();
            if (e.kind == K_LIT) {
                b.drop(e.bits);
                t->literals[t->nliterals++] = (uint8_t)e.val;
                o.out_pos++;
                if (++o.run == kMaxRun || o.out_pos - o.seg_out >= kSegTarget) {
                    flush_run(o);
                    maybe_new_segment(o);
                }
                continue;
            }
            if (e.kind & K_BAD) FAIL("invalid literal/length code");
            if (e.kind & K_EOB) {
                b.drop(e.bits);
                break;
            }
            // length
            unsigned xb = e.kind & 15;
            if (b.cnt < (unsigned)e.bits + xb) STARVE();
            b.drop(e.bits);
            unsigned len = e.val + b.peek(xb);
            b.drop(xb);
            // distance
            Ent d = disttab[b.peek((unsigned)distroot)];
            if (d.kind & K_LINK) {
                unsigned sub = d.kind & 15;
                unsigned idx = (unsigned)((b.hold >> d.bits) & ((1u << sub) - 1));
                Ent d2 = disttab[d.val + idx];
                if (b.cnt < (unsigned)d.bits + d2.bits) STARVE();
                b.drop(d.bits);
                d = d2;
            }
            if (b.cnt < d.bits) STARVE();
            if (d.kind & K_BAD) FAIL("invalid distance code");
            xb = d.kind & 15;
            if (b.cnt < (unsigned)d.bits + xb) STARVE();
            b.drop(d.bits);
            unsigned dist = d.val + b.peek(xb);
            b.drop(xb);
            if (dist > o.out_pos) {
                if (dist > o.out_pos + window_len) FAIL("invalid distance too far back");
                if (dist - o.out_pos > max_reach) max_reach = dist - o.out_pos;
            }
            flush_run(o);
            maybe_new_segment(o);
            push_token(o, 0x80000000u | ((len - 3) << 16) | (dist - 1));
            o.out_pos += len;
        }
    }
    t->status = Z_STREAM_END_;
done:
    if (o.oom) goto oom;
    flush_run(o);
    if (o.oom) goto oom;
    {
        // terminal triple
        if ((t->nsegs + 1) * 3 > o.seg_cap) grow(o.re, (void **)&t->segs, &o.seg_cap, (t->nsegs + 1) * 3, 8, &o.oom);
        if (o.oom) goto oom;
        uint64_t *s = t->segs + 3 * t->nsegs;
        s[0] = t->ntokens;
        s[1] = o.out_pos;
        s[2] = t->nliterals;
    }
    t->out_len = o.out_pos;
    t->in_used = (size_t)(b.next - src) - (b.cnt >> 3);
    if (ctl) {
        ctl->end_bit = 8ull * (uint64_t)(b.next - src) - b.cnt;
        ctl->max_reach = max_reach;
    }
    return t->status;
oom:
    t->status = Z_MEM_ERROR_;
    t->msg = "out of memory";
    return t->status;
}

}  // namespace

// ---- where does a deflate block start?  (multi-threaded single-stream inflate) ------------------------------------
// A thread that is to decode the middle of a stream first has to find a block boundary.  Block headers carry no
// marker, so every bit position of the thread's range is tried: it is a candidate when a DYNAMIC block header that
// the decoder would accept starts there (RFC 1951 3.2.7 and the validity rules of inftrees.c:108-131 leave about one
// random position in 10^6 standing), or when the empty stored block of a Z_SYNC_FLUSH / Z_FULL_FLUSH does (three zero
// bits, zero padding, 00 00 ff ff: deflate.c:1064-1076 -- what pigz and this library's own encoders put between
// their blocks).  Fixed-Huffman and non-empty stored blocks cannot be told from noise and are never candidates; a
// stream made of nothing else simply decodes on one thread.  A candidate is only a guess: it counts once the decode
// of the part before it ENDS exactly there.
namespace {

inline uint64_t bits_at(const uint8_t *src, size_t src_len, uint64_t bit) {        // 57+ bits starting at `bit`
    const size_t byte = (size_t)(bit >> 3);
    uint64_t w = 0;
    if (byte + 8 <= src_len) memcpy(&w, src + byte, 8);
    else if (byte < src_len) memcpy(&w, src + byte, src_len - byte);
    return w >> (bit & 7);
}

bool plausible_dynamic_header(const uint8_t *src, size_t src_len, uint64_t bit) {
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint64_t w = bits_at(src, src_len, bit + 3);
    const unsigned nlen = (unsigned)(w & 31) + 257, ndist = (unsigned)((w >> 5) & 31) + 1, ncode = (unsigned)((w >> 10) & 15) + 4;
    if (nlen > 286 || ndist > 30) return false;
    uint16_t cl[19] = {0};
    uint64_t pos = bit + 17;
    unsigned kraft = 0;                                   // a CODES set must be complete (inftrees.c:126-131)
    for (unsigned i = 0; i < ncode; ++i, pos += 3) {
        const unsigned l = (unsigned)(bits_at(src, src_len, pos) & 7);
        cl[order[i]] = (uint16_t)l;
        if (l) kraft += 128u >> l;
    }
    if (kraft != 128u) return false;
    if ((pos >> 3) + 8 > src_len) return false;           // too close to the end to be worth a thread
    Ent cltab[(1 << 7) + 128];
    int clroot = 0;
    if (build_table(T_CODES, cl, 19, 7, cltab, &clroot)) return false;
    uint16_t lens[320];
    unsigned have = 0;
    while (have < nlen + ndist) {
        const uint64_t v = bits_at(src, src_len, pos);
        const Ent e = cltab[v & ((1u << clroot) - 1)];
        if (e.kind & K_BAD) return false;
        pos += e.bits;
        const unsigned sym = e.val;
        if (sym < 16) {
            lens[have++] = (uint16_t)sym;
            continue;
        }
        unsigned rep, val = 0;
        const uint64_t x = v >> e.bits;
        if (sym == 16) {
            if (have == 0) return false;
            val = lens[have - 1];
            rep = 3 + (unsigned)(x & 3);
            pos += 2;
        } else if (sym == 17) {
            rep = 3 + (unsigned)(x & 7);
            pos += 3;
        } else {
            rep = 11 + (unsigned)(x & 127);
            pos += 7;
        }
        if (have + rep > nlen + ndist) return false;
        while (rep--) lens[have++] = (uint16_t)val;
        if ((pos >> 3) + 8 > src_len) return false;
    }
    if (lens[256] == 0) return false;
    static thread_local Ent lentab[(1 << 10) + 1024], disttab[(1 << 9) + 512];
    int root = 0;
    if (build_table(T_LENS, lens, (int)nlen, 10, lentab, &root)) return false;
    if (build_table(T_DISTS, lens + nlen, (int)ndist, 9, disttab, &root)) return false;
    return true;
}

}  // namespace

namespace {

// the first 13 bits of a block header: BFINAL, BTYPE, HLIT, HDIST -- set when they can belong to a dynamic block
// (BTYPE 10, at most 286 literal/length and 30 distance codes: inflate.c:808-813)
struct FinderTables {
    uint8_t  hdr_ok[8192];
    uint16_t kraft9[512];       // three 3-bit code lengths -> their share of the Kraft sum, in 1/128
    FinderTables() {
        for (unsigned v = 0; v < 8192; ++v)
            hdr_ok[v] = ((v >> 1) & 3) == 2 && ((v >> 3) & 31) <= 29 && ((v >> 8) & 31) <= 29;
        for (unsigned v = 0; v < 512; ++v) {
            unsigned sum = 0;
            for (int f = 0; f < 3; ++f) {
                const unsigned l = (v >> (3 * f)) & 7;
                if (l) sum += 128u >> l;
            }
            kraft9[v] = (uint16_t)sum;
        }
    }
};
const FinderTables &finder_tables() {
    static const FinderTables t;
    return t;
}

// is the code-length code of the dynamic header at `bit` complete?  (the cheap half of plausible_dynamic_header)
inline bool cl_code_complete(const FinderTables &ft, const uint8_t *src, size_t src_len, uint64_t bit, uint64_t v) {
    const unsigned ncode = (unsigned)((v >> 13) & 15) + 4;
    uint64_t c = bits_at(src, src_len, bit + 17);                       // 57 bits: all 19 fields
    if (ncode < 19) c &= (1ull << (3 * ncode)) - 1;
    unsigned sum = 0;
    for (int g = 0; g < 7; ++g) sum += ft.kraft9[(c >> (9 * g)) & 511];
    return sum == 128u;
}

// a stored block whose header byte sits at byte `b` (BTYPE 00 in bits 1-2, LEN and NLEN behind it)?
inline bool stored_at(const uint8_t *src, size_t src_len, uint64_t b, uint32_t *len) {
    if (b + 5 > src_len || (src[b] & 6) != 0) return false;
    const uint32_t ln = src[b + 1] | ((uint32_t)src[b + 2] << 8), nl = src[b + 3] | ((uint32_t)src[b + 4] << 8);
    if ((ln ^ nl) != 0xffffu) return false;
    *len = ln;
    return true;
}

}  // namespace

// first candidate block start in [from_bit, to_bit), or ~0.  Candidates:
//   * a dynamic block header the decoder would accept, at any bit offset;
//   * the block BEHIND a sync-flush marker (empty stored block: LEN = 0, NLEN = 0xffff -- the bytes 00 00 ff ff, on a
//     byte boundary whatever came before): found by a plain byte search over the whole range first, at memory speed --
//     what pigz, Z_SYNC_FLUSH / Z_FULL_FLUSH users and this library's own encoders put between their blocks;
//   * a stored block whose header byte is on a byte boundary (it follows another stored block: incompressible
//     stretches are nothing else) and behind whose payload another recognisable header follows.
// The search gives up after kScanBytes: a part without a candidate is simply decoded by the thread in front of it.
constexpr uint64_t kScanBytes = 192u << 10;

uint64_t zr_inflate_find_block(const uint8_t *src, size_t src_len, uint64_t from_bit, uint64_t to_bit) {
    const FinderTables &ft = finder_tables();
    const uint64_t limit = 8ull * src_len;
    if (to_bit > limit) to_bit = limit;
    {
        // markers first: the block that follows one starts on the byte boundary behind it
        static const uint8_t marker[4] = {0x00, 0x00, 0xff, 0xff};
        const size_t lo = (size_t)((from_bit + 7) >> 3), hi = (size_t)(to_bit >> 3);
        if (hi > lo + 4) {
            const void *hit = memmem(src + lo, hi - lo, marker, 4);
            if (hit) {
                const size_t q = (size_t)((const uint8_t *)hit - src) + 4;
                if (q + 2 <= src_len) return 8ull * q;
            }
        }
    }
    if (to_bit > from_bit + 8 * kScanBytes) to_bit = from_bit + 8 * kScanBytes;
    for (uint64_t bit = from_bit; bit < to_bit; ++bit) {
        const uint64_t v = bits_at(src, src_len, bit);
        if (ft.hdr_ok[v & 8191]) {
            if (cl_code_complete(ft, src, src_len, bit, v) && plausible_dynamic_header(src, src_len, bit)) return bit;
            continue;
        }
        if ((v & 6) != 0) continue;                                  // BTYPE 00 from here on
        if ((v & 1) == 0) {
            // the sync marker: header 000, zero padding up to the byte boundary, LEN = 0, NLEN = 0xffff
            const unsigned pad = (unsigned)((8 - ((bit + 3) & 7)) & 7);
            if (((v >> 3) & ((1u << pad) - 1)) == 0) {
                const uint64_t q = (bit + 3 + pad) >> 3;
                if (q + 4 <= src_len && src[q] == 0 && src[q + 1] == 0 && src[q + 2] == 0xff && src[q + 3] == 0xff) return bit;
            }
        }
        if ((bit & 7) == 0) {
            uint32_t len = 0, len2 = 0;
            const uint64_t b = bit >> 3;
            if (stored_at(src, src_len, b, &len) && len) {
                const uint64_t nb = b + 5 + len;                     // where the next block header must be
                if (nb + 8 <= src_len) {
                    const uint64_t nv = bits_at(src, src_len, 8 * nb);
                    if (stored_at(src, src_len, nb, &len2) ||
                        (ft.hdr_ok[nv & 8191] && cl_code_complete(ft, src, src_len, 8 * nb, nv)))
                        return bit;
                }
            }
        }
    }
    return ~0ull;
}

// decode from ctl->start_bit until a block ends on one of ctl->stops or the stream ends; arrays are carried over
int zr_inflate_decode_part(const uint8_t *src, size_t src_len, zng_rocm_inflate_tokens *t, size_t caps[3],
                           void *(*re)(void *, size_t, size_t), ZrDecodeCtl *ctl) {
    if (!re) re = plain_realloc;
    t->ntokens = t->nliterals = t->nsegs = 0;
    t->out_len = 0;
    t->in_used = 0;
    t->status = 0;
    t->msg = "";
    return decode_stream(src, src_len, 32768, t, re, caps, ctl);       // history is checked when the parts are joined
}

// internal (inflate_many.hip): decode into arrays the caller keeps across streams -- `t`'s pointers and `caps` are
// carried over, its counters are reset; arrays grow through `re`
int zr_inflate_decode_reuse(const uint8_t *src, size_t src_len, uint32_t window_len, zng_rocm_inflate_tokens *t,
                            size_t caps[3], void *(*re)(void *, size_t, size_t)) {
    t->ntokens = t->nliterals = t->nsegs = 0;
    t->out_len = 0;
    t->in_used = 0;
    t->status = 0;
    t->msg = "";
    return decode_stream(src, src_len, window_len, t, re, caps);
}

extern "C" {

int zng_rocm_inflate_tokens_decode(const uint8_t *src, size_t src_len, zng_rocm_inflate_tokens *out) {
    if (!out || (!src && src_len)) return ZNG_ROCM_EINVAL;
    memset(out, 0, sizeof(*out));
    return decode_stream(src, src_len, 0, out);
}

int zng_rocm_inflate_tokens_decode_window(const uint8_t *src, size_t src_len, uint32_t window_len,
                                          zng_rocm_inflate_tokens *out) {
    if (!out || (!src && src_len) || window_len > 32768u) return ZNG_ROCM_EINVAL;
    memset(out, 0, sizeof(*out));
    return decode_stream(src, src_len, window_len, out);
}

void zng_rocm_inflate_tokens_free(zng_rocm_inflate_tokens *t) {
    if (!t) return;
    free(t->tokens);
    free(t->literals);
    free(t->segs);
    memset(t, 0, sizeof(*t));
}

}  // extern "C"
