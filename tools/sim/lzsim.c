// lzsim.c -- CPU model of the level-6 class match finders, to choose the LDS search structure before writing the
// kernel.  Not product code, not the oracle: it estimates the compressed size a finder would
// give on a plaintext file (serial greedy + one-step-lazy parse, per-segment Shannon cost of the dynamic block).
//   ./lzsim file chain HBITS MAXCHAIN            hash chains (head 2^HBITS, prev deltas): round-2 kernel's structure
//   ./lzsim file row ROWS ENTRIES TAGBITS MAXCAND   associative rows: ROWS x ENTRIES (pos16, tag) FIFO per row
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXDIST (32768 - 262)
#define MAXLEN 258
#define PROBE 64          /* the kernel's per-lane compare cap; longer matches are extended afterwards */
static const uint8_t *buf;
static size_t n;
static unsigned long long n_verify, n_cand_reads, n_pos;

static inline uint32_t rd32(size_t p) { uint32_t v; memcpy(&v, buf + p, 4); return v; }
static inline uint32_t match_len(size_t p, size_t c, uint32_t maxlen) {
    uint32_t l = 0;
    while (l < maxlen && buf[p + l] == buf[c + l]) ++l;
    return l;
}

// ---- chains ----
static uint32_t *head; static uint16_t *prevd; static int hbits, maxchain;
static void chain_insert(size_t p) {
    uint32_t h = (rd32(p) * 2654435761u) >> (32 - hbits);
    uint32_t old = head[h];
    head[h] = (uint32_t)p + 1;
    uint32_t d = old ? (uint32_t)p - (old - 1) : 0;
    prevd[p & 32767] = d <= 65535 ? d : 0;
}
static uint32_t chain_find(size_t p, uint32_t maxlen, uint32_t *dist) {   // p already inserted
    uint32_t best = 3, bd = 0, chain = maxchain; int eased = 0;
    size_t c = p;
    while (chain--) {
        uint32_t d = prevd[c & 32767];
        if (!d || d > c) break;
        c -= d;
        if (p - c > MAXDIST) break;
        ++n_cand_reads;
        if (rd32(c) == rd32(p) || 1) {
            if (buf[c + best] == buf[p + best] || best >= maxlen) { ++n_verify;
            uint32_t l = match_len(p, c, maxlen);
            if (l > best) { best = l; bd = p - c; if (l >= PROBE || l >= maxlen) break;
                if (!eased && best >= 8) { chain >>= 2; eased = 1; } } }
        }
    }
    *dist = bd;
    return best >= 4 ? best : 0;
}

// ---- rows ----
static int rows, entries, tagbits, maxcand;
static uint16_t *rpos; static uint16_t *rtag; static uint32_t *rcnt;
static inline void row_key(size_t p, uint32_t *row, uint32_t *tag) {
    uint32_t h = rd32(p) * 2654435761u;
    *row = (uint32_t)(((uint64_t)(h >> 8) * (uint64_t)rows) >> 24);
    *tag = (h & ((1u << tagbits) - 1)) ;
}
static void row_insert(size_t p) {
    uint32_t r, t; row_key(p, &r, &t);
    uint32_t s = rcnt[r]++ % entries;
    rpos[(size_t)r * entries + s] = (uint16_t)p;
    rtag[(size_t)r * entries + s] = (uint16_t)(t | 0x8000);
}
static uint32_t row_find(size_t p, uint32_t maxlen, uint32_t *dist) {     // p already inserted (slot rcnt-1)
    uint32_t r, t; row_key(p, &r, &t);
    uint32_t best = 3, bd = 0; int cand = maxcand, eased = 0;
    uint32_t s = rcnt[r] - 1;
    for (int k = 1; k < entries && cand > 0; ++k) {                 // newest first
        uint32_t e = (s - k) % entries;
        if (s < (uint32_t)k) break;
        if (rtag[(size_t)r * entries + e] != (uint16_t)(t | 0x8000)) continue;
        uint32_t d = (uint16_t)((uint16_t)p - rpos[(size_t)r * entries + e]);
        if (d == 0 || d > MAXDIST || d > p) continue;
        size_t c = p - d;
        --cand; ++n_cand_reads;
        if (best < maxlen && buf[c + best] != buf[p + best]) continue;
        ++n_verify;
        uint32_t l = match_len(p, c, maxlen);
        if (l > best) { best = l; bd = d; if (l >= PROBE || l >= maxlen) break;
            if (!eased && best >= 8) { cand = (cand + 3) >> 2; eased = 1; } }
    }
    *dist = bd;
    return best >= 4 ? best : 0;
}

static int mode; static int rep_probe;
static uint32_t find(size_t p, uint32_t *dist) {
    uint32_t maxlen = n - p < MAXLEN ? (uint32_t)(n - p) : MAXLEN;
    if (maxlen < 4) return 0;
    uint32_t l = mode == 0 ? chain_find(p, maxlen, dist) : row_find(p, maxlen, dist);
    if (rep_probe && p >= 1) {                                      // distance-1 probe (runs)
        uint32_t l1 = match_len(p, p - 1, maxlen);
        if (l1 >= 4 && l1 > l) { l = l1; *dist = 1; }
    }
    if (l >= PROBE) l = match_len(p, p - *dist, maxlen);
    return l;
}

static double cost_block(const uint32_t *lf, const uint32_t *df, double extra) {
    double tl = 0, td = 0, bits = extra;
    for (int i = 0; i < 286; ++i) tl += lf[i];
    for (int i = 0; i < 30; ++i) td += df[i];
    for (int i = 0; i < 286; ++i) if (lf[i]) bits += lf[i] * -log2(lf[i] / tl);
    for (int i = 0; i < 30; ++i) if (df[i]) bits += df[i] * -log2(df[i] / td);
    return bits * 1.004 + 8 * 90;       // Huffman vs entropy slack + header
}
static int len_sym(uint32_t len, uint32_t *eb) { uint32_t l = len - 3; *eb = 0; if (l < 8) return 257 + l; if (l == 255) return 285;
    uint32_t lg = 31 - __builtin_clz(l); *eb = lg - 2; return 257 + 4 * *eb + 4 + ((l >> *eb) & 3); }
static int dist_sym(uint32_t dist, uint32_t *eb) { uint32_t x = dist - 1; *eb = 0; if (x < 4) return x;
    uint32_t lg = 31 - __builtin_clz(x); *eb = lg - 1; return 2 * lg + ((x >> *eb) & 1); }

int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *b = malloc(n + 512); if (fread(b, 1, n, f) != n) return 1; memset(b + n, 0, 512); buf = b;
    int a = 3;
    if (!strcmp(argv[2], "chain")) { mode = 0; hbits = atoi(argv[a++]); maxchain = atoi(argv[a++]);
        head = calloc(1u << hbits, 4); prevd = calloc(32768, 2); }
    else { mode = 1; rows = atoi(argv[a++]); entries = atoi(argv[a++]); tagbits = atoi(argv[a++]); maxcand = atoi(argv[a++]);
        rpos = calloc((size_t)rows * entries, 2); rtag = calloc((size_t)rows * entries, 2); rcnt = calloc(rows, 4); }
    rep_probe = argc > a ? atoi(argv[a++]) : 0;
    int lazy2 = argc > a ? atoi(argv[a++]) : 0;
    const size_t seg = 512 << 10;
    double total_bits = 0; uint32_t lf[286], df[30]; double extra = 0;
    memset(lf, 0, sizeof lf); memset(df, 0, sizeof df);
    size_t p = 0, next_seg = seg, ins = 0; unsigned long long nmatch = 0, nlit = 0, mbytes = 0;
    // the kernel searches EVERY position (matches are known everywhere); the serial parse below uses them lazily
    uint32_t curl = 0, curd = 0; int have = 0;
    while (p < n) {
        if (p >= next_seg) { lf[256]++; total_bits += cost_block(lf, df, extra); memset(lf, 0, sizeof lf); memset(df, 0, sizeof df); extra = 0; next_seg += seg; }
        while (ins <= p + 1 && ins + 4 <= n) { if (mode == 0) chain_insert(ins); else row_insert(ins); ++ins; }
        uint32_t d0, l0;
        if (have) { l0 = curl; d0 = curd; have = 0; } else { l0 = find(p, &d0); ++n_pos; }
        uint32_t l1 = 0, d1 = 0;
        if (l0 >= 4 && l0 < 16 && p + 1 < n) { l1 = find(p + 1, &d1); ++n_pos; }
        if (l0 >= 4 && !(l1 > l0)) {
            uint32_t eb; lf[len_sym(l0, &eb)]++; extra += eb; df[dist_sym(d0, &eb)]++; extra += eb;
            ++nmatch; mbytes += l0;
            // positions inside the match are inserted (deflate_medium inserts them all)
            size_t e = p + l0;
            while (ins < e && ins + 4 <= n) { if (mode == 0) chain_insert(ins); else row_insert(ins); ++ins; }
            p = e;
        } else {
            lf[buf[p]]++; ++nlit; ++p;
            if (l1) { curl = l1; curd = d1; have = 1; }
        }
        (void)lazy2;
    }
    lf[256]++; total_bits += cost_block(lf, df, extra);
    printf("%s: in %zu out %.0f ratio %.4f | matches %llu (avg len %.1f) literals %llu | per searched position: cand reads %.2f verifies %.2f\n",
           argv[2], n, total_bits / 8, n / (total_bits / 8), nmatch, (double)mbytes / (nmatch ? nmatch : 1), nlit,
           (double)n_cand_reads / n_pos, (double)n_verify / n_pos);
    return 0;
}
