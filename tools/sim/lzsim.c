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

// ---- rows, batch form: the kernel inserts a whole batch of BATCH positions, then searches; a lane sees (A) the row as it
// was before the batch (read before the inserts) and (C) the row after all inserts, of which only positions inside the
// batch and below its own are new information.  Up to two tables, keyed by the first KB bytes (4 and e.g. 7).
static int batch = 1024; static size_t batch_base = (size_t)-1;
typedef struct { int rows, entries, kb; uint16_t *pos, *tag, *spos, *stag; uint32_t *cnt; } Tab;
static Tab tabs[2]; static int ntabs;
static inline uint64_t rd64(size_t p) { uint64_t v; memcpy(&v, buf + p, 8); return v; }
static inline void tab_key(const Tab *T, size_t p, uint32_t *row, uint32_t *tag) {
    uint64_t v = rd64(p); if (T->kb < 8) v &= (1ull << (8 * T->kb)) - 1;
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    uint32_t h = (lo * 2654435761u) ^ (hi * 0x7FEB352Du) ^ ((hi * 0x846CA68Bu) >> 15);
    uint32_t g = (lo * 0x85EBCA6Bu) ^ (hi * 0xC2B2AE35u);
    *row = (uint32_t)(((uint64_t)h * (uint64_t)T->rows) >> 32);
    *tag = (g >> 24) & ((1u << tagbits) - 1);
}
static void tab_init(Tab *T, int rows_, int entries_, int kb) {
    T->rows = rows_; T->entries = entries_; T->kb = kb; size_t m = (size_t)rows_ * entries_;
    T->pos = calloc(m, 2); T->tag = calloc(m, 2); T->spos = calloc(m, 2); T->stag = calloc(m, 2); T->cnt = calloc(rows_, 4);
}
static void rowb_begin(size_t P) {          // snapshot, then insert [P, P + batch)
    for (int k = 0; k < ntabs; ++k) { Tab *T = &tabs[k]; size_t m = (size_t)T->rows * T->entries;
        memcpy(T->spos, T->pos, m * 2); memcpy(T->stag, T->tag, m * 2);
        for (size_t q = P; q < P + batch && q + 8 <= n + 8; ++q) { uint32_t r, t; tab_key(T, q, &r, &t);
            uint32_t s = T->cnt[r]++ % T->entries; T->pos[(size_t)r * T->entries + s] = (uint16_t)q; T->tag[(size_t)r * T->entries + s] = (uint16_t)(t | 0x8000); } }
    batch_base = P;
}
static int cmp_u32(const void *a, const void *b) { uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : x > y; }
static uint32_t rowb_find(size_t p, uint32_t maxlen, uint32_t *dist) {
    size_t P = p - p % batch;
    if (P != batch_base) rowb_begin(P);
    uint32_t best = 3, bd = 0;
    for (int k = ntabs - 1; k >= 0; --k) {                                  // the long key first
        Tab *T = &tabs[k]; uint32_t r, t; tab_key(T, p, &r, &t);
        uint32_t ds[64]; int nd = 0;
        for (int e = 0; e < T->entries; ++e) {                              // C: after the inserts, inside the batch, below p
            if (T->tag[(size_t)r * T->entries + e] != (uint16_t)(t | 0x8000)) continue;
            uint32_t d = (uint16_t)((uint16_t)p - T->pos[(size_t)r * T->entries + e]);
            if (d >= 1 && d <= p - P && d <= MAXDIST) ds[nd++] = d;
        }
        for (int e = 0; e < T->entries; ++e) {                              // A: before the batch
            if (T->stag[(size_t)r * T->entries + e] != (uint16_t)(t | 0x8000)) continue;
            uint32_t d = (uint16_t)((uint16_t)p - T->spos[(size_t)r * T->entries + e]);
            if (d > p - P && d <= MAXDIST && d <= p) ds[nd++] = d;
        }
        qsort(ds, nd, 4, cmp_u32);
        int cand = maxcand;
        for (int i = 0; i < nd && cand > 0; ++i) {
            size_t c = p - ds[i];
            --cand; ++n_cand_reads;
            if (best < maxlen && buf[c + best] != buf[p + best]) continue;
            ++n_verify;
            uint32_t l = match_len(p, c, maxlen);
            if (l > best) { best = l; bd = ds[i]; if (l >= PROBE || l >= maxlen) goto out; }
        }
    }
out:
    *dist = bd;
    return best >= 4 ? best : 0;
}
static int mode; static int rep_probe;
static uint32_t find(size_t p, uint32_t *dist) {
    uint32_t maxlen = n - p < MAXLEN ? (uint32_t)(n - p) : MAXLEN;
    if (maxlen < 4) return 0;
    uint32_t l = mode == 0 ? chain_find(p, maxlen, dist) : mode == 1 ? row_find(p, maxlen, dist) : rowb_find(p, maxlen, dist);
    for (int d = 1; d <= rep_probe && (size_t)d <= p; ++d) {        // short-distance probes (runs, small periods)
        if (rd32(p) != rd32(p - d)) continue;
        uint32_t l1 = match_len(p, p - d, maxlen);
        if (l1 >= 4 && l1 > l) { l = l1; *dist = d; }
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
    else if (!strcmp(argv[2], "rowb")) { mode = 2; rows = atoi(argv[a++]); entries = atoi(argv[a++]); tagbits = atoi(argv[a++]); maxcand = atoi(argv[a++]);
        tab_init(&tabs[0], rows, entries, 4); ntabs = 1; }
    else { mode = 1; rows = atoi(argv[a++]); entries = atoi(argv[a++]); tagbits = atoi(argv[a++]); maxcand = atoi(argv[a++]);
        rpos = calloc((size_t)rows * entries, 2); rtag = calloc((size_t)rows * entries, 2); rcnt = calloc(rows, 4); }
    rep_probe = argc > a ? atoi(argv[a++]) : 0;
    int lazy2 = argc > a ? atoi(argv[a++]) : 0; if (argc > a) batch = atoi(argv[a++]);
    if (argc > a + 2) { int r2 = atoi(argv[a++]), e2 = atoi(argv[a++]), kb = atoi(argv[a++]); tab_init(&tabs[1], r2, e2, kb); ntabs = 2; }
    const size_t seg = 512 << 10;
    double total_bits = 0; uint32_t lf[286], df[30]; double extra = 0;
    memset(lf, 0, sizeof lf); memset(df, 0, sizeof df);
    size_t p = 0, next_seg = seg, ins = 0; unsigned long long nmatch = 0, nlit = 0, mbytes = 0;
    // the kernel searches EVERY position (matches are known everywhere); the serial parse below uses them lazily
    uint32_t curl = 0, curd = 0; int have = 0;
    if (getenv("PARSE")) {
        // region DP: every REGION positions are parsed optimally for the current cost estimate given ONE longest match per
        // position (optionally truncated); the parse of a region is followed from wherever the previous region's last token
        // ended (exact stitching); costs come from the symbols chosen so far in this segment
        const int REGION = getenv("REGION") ? atoi(getenv("REGION")) : 64;
        const int trunc = strstr(getenv("PARSE"), "trunc") != NULL;
        const int adapt = strstr(getenv("PARSE"), "static") == NULL;
        static uint32_t L[4096], D[4096]; static double C[4096 + 300]; static uint16_t ch[4096];
        double lc[286], dc[30];
        uint32_t hl[286], hd[30];
        #define RESET_COSTS() do { for (int i = 0; i < 256; ++i) lc[i] = 8; for (int i = 256; i < 286; ++i) lc[i] = 7; for (int i = 0; i < 30; ++i) dc[i] = 5; \
            memset(hl, 0, sizeof hl); memset(hd, 0, sizeof hd); } while (0)
        RESET_COSTS();
        size_t cover = 0, since = 0; double beta = getenv("BETA") ? atof(getenv("BETA")) : 4.0; const size_t refresh = getenv("REFRESH") ? atoi(getenv("REFRESH")) : 4096;
        for (size_t w0 = 0; w0 < n; w0 += REGION) {
            if (w0 >= next_seg) { lf[256]++; total_bits += cost_block(lf, df, extra); memset(lf, 0, sizeof lf); memset(df, 0, sizeof df); extra = 0; next_seg += seg; RESET_COSTS(); since = 0; }
            size_t end = w0 + REGION < n ? w0 + REGION : n; int m = (int)(end - w0);
            for (int i = 0; i < m; ++i) { uint32_t d = 0; L[i] = find(w0 + i, &d); D[i] = d; ++n_pos;
                if (w0 + i + L[i] > next_seg) L[i] = (uint32_t)(next_seg - (w0 + i)) >= 4 ? (uint32_t)(next_seg - (w0 + i)) : 0; }
            for (int x = m; x < m + 300; ++x) C[x] = -beta * (x - m);
            for (int i = m - 1; i >= 0; --i) {
                double best = lc[buf[w0 + i]] + C[i + 1]; int bl = 1;
                if (L[i] >= 4) {
                    uint32_t eb; int dsym = dist_sym(D[i], &eb); double dcost = dc[dsym] + eb;
                    for (uint32_t l = trunc ? 4 : L[i]; l <= L[i]; ++l) {
                        uint32_t el; if (l > 64 && l != L[i]) continue; int ls = len_sym(l, &el); double c = lc[ls] + el + dcost + C[i + l];
                        if (c < best) { best = c; bl = (int)l; }
                    }
                }
                C[i] = best; ch[i] = (uint16_t)bl;
            }
            while (cover < end) {
                int i = (int)(cover - w0); int l = ch[i];
                if (l == 1) { lf[buf[cover]]++; hl[buf[cover]]++; ++nlit; }
                else { uint32_t eb; int ls = len_sym(l, &eb); lf[ls]++; hl[ls]++; extra += eb; int dsym = dist_sym(D[i], &eb); df[dsym]++; hd[dsym]++; extra += eb; ++nmatch; mbytes += l; }
                cover += l;
            }
            since += m;
            if (adapt && since >= refresh) {            // refresh the cost estimate from the symbols of this segment so far
                double tl = 1, td = 1; for (int i = 0; i < 286; ++i) tl += hl[i] + 0.5; for (int i = 0; i < 30; ++i) td += hd[i] + 0.5;
                for (int i = 0; i < 286; ++i) { lc[i] = -log2((hl[i] + 0.5) / tl); if (lc[i] > 15) lc[i] = 15; }
                for (int i = 0; i < 30; ++i) { dc[i] = -log2((hd[i] + 0.5) / td); if (dc[i] > 15) dc[i] = 15; }
                since = 0;
            }
        }
        p = n;
    }
    while (p < n) {
        if (p >= next_seg) { lf[256]++; total_bits += cost_block(lf, df, extra); memset(lf, 0, sizeof lf); memset(df, 0, sizeof df); extra = 0; next_seg += seg; }
        while (mode != 2 && ins <= p + 1 && ins + 4 <= n) { if (mode == 0) chain_insert(ins); else row_insert(ins); ++ins; }
        uint32_t d0, l0;
        if (have) { l0 = curl; d0 = curd; have = 0; } else { l0 = find(p, &d0); ++n_pos; }
        uint32_t l1 = 0, d1 = 0;
        if (l0 >= 4 && l0 < 16 && p + 1 < n) { l1 = find(p + 1, &d1); ++n_pos; }
        if (l0 >= 4 && !(l1 > l0)) {
            uint32_t eb; lf[len_sym(l0, &eb)]++; extra += eb; df[dist_sym(d0, &eb)]++; extra += eb;
            ++nmatch; mbytes += l0;
            // positions inside the match are inserted (deflate_medium inserts them all)
            size_t e = p + l0;
            while (mode != 2 && ins < e && ins + 4 <= n) { if (mode == 0) chain_insert(ins); else row_insert(ins); ++ins; }
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
