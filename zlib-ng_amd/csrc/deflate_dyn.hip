// deflate_dyn.hip -- level-6 class deflate of ONE large stream on device (BASELINE.json configs[3]).
//
// The caller this replaces is deflate_medium (deflate_medium.c:145-277: hash insert, longest_match, look-ahead for a
// better match) followed by zng_tr_flush_block's dynamic-tree block (trees.c:625-741), behind DEFLATE_HOOK
// (deflate.c:1039).  A single zlib stream is strictly serial on a CPU; here the plaintext is already complete in HBM,
// so it is cut into SEGMENTS that run in parallel:
//
//   K1 lz_rows_kernel    one workgroup per segment.  The search rows are primed with the 32 KiB that precede the
//        segment (what deflateSetDictionary / the sliding window would have left there), then the segment is parsed with
//        the front end of deflate_rows.h (associative rows AND the 64 KiB sliding window in LDS; shortest-path parse).
//        Matches may reach back into the previous segment: the output is ONE continuous deflate stream.
//        Output per segment: a bitmap of token starts (1 bit per position; a token's length is the distance to the next
//        start), the distance of every match (u16 per 4 positions: matches are >= 4 long, so two never share a slot) --
//        0.625 bytes of scratch per input byte (round 2: a 32-bit selector per byte) -- and the symbol histogram.
//   K2 emit_dynamic_kernel one workgroup per segment = one dynamic-Huffman block: Huffman code lengths from K1's
//        histogram (rank sort + two-queue merge, depth limit by frequency halving), canonical codes, the RLE-coded
//        length header (trees.c send_all_trees equivalent), then the body bits assembled in an LDS tile.  Each segment
//        ends with an empty stored block so that it is byte aligned (what Z_SYNC_FLUSH emits, deflate.c:1064-1076);
//        the last one appends the final empty static block.
//   K3 segments_scan_kernel: the segments' places in the output (a one-block scan, per stream), the compressed sizes,
//        the out_cap check -- on the device, so the call has no synchronisation before its last kernel;
//   K4 gather_segments_kernel packs the segments back to back.
//
// Per segment the block type is chosen as zng_tr_flush_block does (trees.c:660-719): stored if that is not larger,
// else static if not larger than dynamic.  Not bit-identical to the reference's stream (different parse, one block
// per segment); validity is defined by round trip, as in the reference's own tests.
#include "context.h"
#include "deflate_dev.h"
#include "deflate_rows.h"

#include <mutex>

namespace zr {

struct SegJob {
    const uint8_t *in;        // stream base (position 0)
    uint8_t       *out;       // this segment's output slot (4-byte aligned)
    uint8_t       *dst;       // the stream's output buffer (what K4 packs into)
    uint64_t       dst_cap;
    uint64_t       bm_off;    // this segment's token-start bitmap: u64 words from the scratch base
    uint64_t       d16_off;   // this segment's match distances: u16 entries from the scratch base
    uint32_t       seg_start, seg_end;
    uint32_t       out_cap;
    uint32_t       is_last;     // the block that carries BFINAL
    uint32_t       first_seg;   // index of the first segment of this segment's stream
    uint32_t       stream;      // row of the per-stream result table; 0x80000000 set on the stream's last segment
};

// one BLOCK of the output (K2 .. K4): the tokens of its segment that start in [blo, bhi)
struct BlkJob {
    const uint8_t *in;        // stream base (position 0)
    uint8_t       *out;       // this block's output slot (4-byte aligned)
    uint8_t       *dst;       // the stream's output buffer (what K4 packs into)
    uint64_t       dst_cap;
    uint64_t       bm_off, d16_off;       // its segment's token scratch
    uint32_t       seg_start, seg_end;    // its segment
    uint32_t       blo, bhi;
    uint32_t       hist_idx, hist_prev;   // histogram snapshot of this block's end; 1 if the snapshot before it is the block's begin
    uint32_t       out_cap;
    uint32_t       is_last;               // the block that carries BFINAL
    uint32_t       first_seg;             // index of the first BLOCK of this block's stream
    uint32_t       stream;                // row of the per-stream result table; 0x80000000 set on the stream's last block
};

constexpr uint32_t kSegBytes = 512u << 10;     // plaintext per segment / per dynamic block: the largest, and ...
constexpr uint32_t kSegBytesMin = 128u << 10;  // ... the smallest.  A segment is one workgroup and one CU holds one
                                               // workgroup, so a stream is cut into at least ~2 segments per CU when
                                               // it is long enough (each segment re-inserts the 32 KiB before it:
                                               // 6 % extra work at 512 KiB, 25 % at 128 KiB)
static uint32_t segment_bytes(size_t in_len, int cus) {
    uint32_t seg = kSegBytes;
    while (seg > kSegBytesMin && in_len / seg < 2u * (size_t)cus) seg >>= 1;
    return seg;
}
constexpr uint32_t kPrime = 32768u;            // dictionary primed from the previous segment
constexpr int      kHistWords = 320;           // 288 literal/length + 32 distance counts
// A segment is written as several BLOCKS, one per kSubBytes of positions (counted from the segment's first batch): zlib's
// blocks hold about as much (lit_bufsize 16384 symbols), the codes follow the data more closely, a block may be stored
// (<= 65535 bytes), and -- what it was done for -- every block start is a place where a parallel inflater can cut the
// stream (inflate_large.hip, inflate_threads.cpp): one block per 512 KiB segment gave a 256 MiB stream 434 parts.
constexpr uint32_t kSubBatches = 60;
constexpr uint32_t kSubBytes = kSubBatches * kRowBatch;          // 61440
constexpr uint32_t kMaxSub = (kSegBytes + kRowBatch + kSubBytes - 1) / kSubBytes + 1;      // histogram snapshots per segment

// scratch a segment of `bytes` plaintext bytes needs: bitmap words / distance entries (its first batch may start up to
// one batch in front of the segment)
static inline size_t seg_bm_words(uint32_t bytes) { return ((size_t)bytes + 2u * kRowBatch) / 64 + 2; }
static inline size_t seg_d16_entries(uint32_t bytes) { return (((size_t)bytes + 2u * kRowBatch) / 4 + 8) & ~(size_t)3; }

// grid.x = segment.  The plaintext streams through the LDS ring one batch ahead of the parse: 256 lanes fetch the
// next 1 KiB chunk into a register at the top of a batch and store it into the ring at the top of the next one, so
// the HBM latency of the fetch is hidden behind a whole batch of searching.
__global__ __launch_bounds__(kRowBatch)
void lz_rows_kernel(const SegJob *__restrict__ jobs, unsigned long long *__restrict__ bm_base, uint16_t *__restrict__ d16_base,
                    uint32_t *__restrict__ hist_out, uint32_t max_cand, unsigned long long *__restrict__ stamp_out) {
    __shared__ RowShared sh;

    const SegJob job = jobs[blockIdx.x];
    const uint8_t *in = job.in;
    const uint32_t n = job.seg_end;                 // matches never run past the segment
    const int t = threadIdx.x, lane = t & 63;

    for (int i = t; i < kRows * kRowEnt / 2; i += kRowBatch) reinterpret_cast<uint32_t *>(sh.pos)[i] = 0;
    for (int i = t; i < kRows * kRowEnt / 4; i += kRowBatch) reinterpret_cast<uint32_t *>(sh.tag)[i] = 0;
    for (int i = t; i < kRows / 4; i += kRowBatch) sh.cnt[i] = 0;
    if (t < 288) {
        sh.hist_l[t] = 0;
        // before the segment has tokens of its own: 8 bits per literal, 7 per length symbol, 5 per distance symbol
        sh.cost_l[t] = (uint16_t)((t < 256 ? 8u : 7u) * kCostBit);
    } else if (t < 320) {
        sh.hist_d[t - 288] = 0;
        sh.cost_d[t - 288] = (uint16_t)(5u * kCostBit);
    }
    if (t == 0) {
        sh.cover = job.seg_start;
        sh.tot_l = sh.tot_d = 0;
    }

    uint32_t P0 = job.seg_start > kPrime ? job.seg_start - kPrime : 0u;
    P0 -= P0 % kRowBatch;
    const uint32_t first = job.seg_start - job.seg_start % kRowBatch;   // batch holding the segment's first byte
    unsigned long long *bm = bm_base + job.bm_off;
    uint16_t *d16 = d16_base + job.d16_off;

    // chunk [F, F + 1024) of the plaintext, one dword per lane of the first four waves; bytes at or beyond n read 0
    auto fetch = [&](uint32_t F) -> uint32_t {
        const uint32_t q = F + 4u * (uint32_t)t;
        if (t >= 256 || q >= n) return 0u;
        if (q + 4u <= n) return load_u32(in + q);
        uint32_t v = 0;
        for (uint32_t j = 0; q + j < n; ++j) v |= (uint32_t)load_u8(in + q + j) << (8u * j);
        return v;
    };
    auto put = [&](uint32_t F, uint32_t v) {
        if (t < 256) {
            const uint32_t idx = (F + 4u * (uint32_t)t) & (kRingBytes - 1u);
            *reinterpret_cast<uint32_t *>(sh.ring + idx) = v;
            if (idx < kRingMirror) *reinterpret_cast<uint32_t *>(sh.ring + kRingBytes + idx) = v;
        }
    };
    put(P0, fetch(P0));
    put(P0 + 1024u, fetch(P0 + 1024u));
    uint32_t chunk = fetch(P0 + 2048u);              // stored at the top of the FIRST batch: [P0 + 2048, P0 + 3072)
    __syncthreads();

    // the tokens of a finished batch: bitmap word, distances, histogram
    auto emit_tokens = [&](uint32_t Pb, const RowsToken &r, unsigned long long starts) __attribute__((always_inline)) {
        const uint32_t p = Pb + (uint32_t)t;
        const unsigned long long matches = __ballot(r.kind == 2u);
        if (lane == 0) {
            bm[(Pb - first) / 64u + (uint32_t)(t >> 6)] = starts;
            const uint32_t nt = (uint32_t)__popcll(starts), nm = (uint32_t)__popcll(matches);
            if (nt) atomicAdd(&sh.tot_l, nt);
            if (nm) atomicAdd(&sh.tot_d, nm);
        }
        if (r.kind == 2u) {
            uint32_t sy, eb;
            rows_len_symbol(r.len, sy, eb);
            atomicAdd(&sh.hist_l[sy], 1u);
            rows_dist_symbol(r.dist, sy, eb);
            atomicAdd(&sh.hist_d[sy], 1u);
            d16[(p - first) >> 2] = (uint16_t)(r.dist - 1u);
        } else if (r.kind == 1u) {
            atomicAdd(&sh.hist_l[sh.ring[p & (kRingBytes - 1u)]], 1u);
        }
    };

    // Two batches in flight: the compares of batch P (LDS-bound) run beside the parse of the batch before it (VALU-bound);
    // waves 4-7 and 12-15 take the two in the opposite order, so every SIMD has both kinds of work at any time.
    const int wave_id = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool parse_first = ((wave_id >> 2) & 1) != 0;
    int since_refresh = 0, snap_due = -1;            // a histogram snapshot (sub-block index) to write once every wave's counts are in
    uint32_t *hist_seg = hist_out + (size_t)blockIdx.x * kMaxSub * kHistWords;
    bool have_prev = false;
    uint32_t P_prev = 0;
    RowsMatch prev;
    prev.L = prev.dist = prev.val = 0;
    for (uint32_t P = P0;; P += kRowBatch) {
        const bool live = P < n;                    // one more round after the last batch: its parse
        if (!live && !have_prev) break;
        if (live) {
            put(P + kRingAhead, chunk);             // [P + 2048, P + 3072): what the NEXT batch reads beyond its own positions
            chunk = fetch(P + kRingAhead + 1024u);
        }
        const uint32_t p = P + (uint32_t)t;
        if (live && P < first) {                    // priming: enter the positions, nothing else
            const bool can = p + kLzMinMatch <= n;
            uint32_t row, tag;
            row_key(ring_u32(sh.ring, p & (kRingBytes - 1u)), row, tag);
            rows_insert(&sh, can, row, tag, p, wave_id);
            continue;
        }
        RowsFront f;
        uint32_t cover_in;
        if (live) {
            const bool refresh = since_refresh >= kRefreshBatches;
            since_refresh = refresh ? 1 : since_refresh + 1;
            rows_front(n, P, &sh, t, refresh, f, &cover_in);
        } else {
            rows_barrier();
            cover_in = sh.cover;
        }
        if (snap_due >= 0) {                        // the barriers above are behind the last batch's histogram updates
            if (t < kHistWords) hist_seg[(size_t)snap_due * kHistWords + t] = t < 288 ? sh.hist_l[t] : sh.hist_d[t - 288];
            snap_due = -1;
        }
        RowsMatch cur;
        cur.L = cur.dist = cur.val = 0;
        uint32_t CH = 1u;
#pragma nounroll
        for (int ph = 0; ph < 2; ++ph) {            // ONE copy of each piece in the code, the order a run-time matter
            if ((ph == 0) == parse_first) {
                if (have_prev) CH = rows_parse(n, P_prev, &sh, t, prev);
            } else if (live) {
                cur = rows_compare(n, P, P0, &sh, t, max_cand, f);
            }
        }
        if (have_prev) {
            unsigned long long starts;
            const RowsToken r = rows_finish(n, P_prev, &sh, t, CH, prev.dist, cover_in, &starts);
            emit_tokens(P_prev, r, starts);
            const uint32_t done_batches = (P_prev - first) / kRowBatch + 1u;
            if (done_batches % kSubBatches == 0) snap_due = (int)(done_batches / kSubBatches) - 1;
        }
#ifdef ZR_ROWS_STAMPS
        if (lane == 0) sh.stamps[t >> 6][8] = __builtin_amdgcn_s_memtime();
        if (blockIdx.x == 3 && stamp_out && lane < 9 && live && P >= first + 64u * kRowBatch && P < first + 96u * kRowBatch)
            stamp_out[(((P - first) / kRowBatch - 64u) * kRowWaves + (uint32_t)(t >> 6)) * 9u + (uint32_t)lane] = sh.stamps[t >> 6][lane];
#endif
        if (!live) break;
        prev = cur;
        P_prev = P;
        have_prev = true;
    }
    __syncthreads();
    {                                                // the totals: the last sub-block's snapshot
        const uint32_t span = n - first;
        const uint32_t nsub = span ? (span + kSubBytes - 1) / kSubBytes : 1u;
        if (t < kHistWords) hist_seg[(size_t)(nsub - 1u) * kHistWords + t] = t < 288 ? sh.hist_l[t] : sh.hist_d[t - 288];
    }
}

// ---- dynamic Huffman ---------------------------------------------------------------------------------------
struct DynTables {                 // LDS
    uint32_t lfreq[288];
    uint32_t dfreq[32];
    uint8_t  llen[288];
    uint8_t  dlen[32];
    uint16_t lcode[288];           // bit-reversed canonical codes
    uint16_t dcode[32];
    // scratch of the tree builder
    uint32_t w[2 * 288];           // weights: sorted leaves [0, m), internal nodes [288, 288 + m - 1)
    uint16_t sym[288];             // leaf rank -> symbol
    uint16_t kid[2][288];          // children of internal node k (index < 288: leaf rank, >= 288: internal)
    uint8_t  depth[2 * 288];
    // code-length header
    uint8_t  clsym[320];
    uint8_t  clext[320];
    uint32_t clfreq[19];
    uint8_t  cllen[19];
    uint16_t clcode[19];
    uint32_t ncl, hlit, hdist, hclen;
    // block-type choice (zng_tr_flush_block, trees.c:660-719)
    uint32_t lfreq0[288], dfreq0[32];     // the histograms as counted (the tree builder may rescale its inputs)
    uint32_t cost_dyn, cost_static;       // body bits under the dynamic / the static codes
    uint32_t mode;                        // 0 = dynamic, 1 = static, 2 = stored
};

// Code lengths (<= maxbits) for freq[0..n) into len[0..n).  All 256 lanes call it; serial parts run on lane 0.
// Equivalent of build_tree + gen_bitlen (trees.c) -- lengths are optimal Huffman lengths unless the depth
// limit is hit, in which case frequencies are halved until the tree fits (still a complete prefix code).
__device__ void huff_lengths(DynTables *T, uint32_t *freq, int n, int maxbits, uint8_t *len, int t) {
    // at least two codes, as zlib forces (trees.c build_tree: "force at least two codes of non zero frequency")
    __syncthreads();
    if (t == 0) {
        int used = 0;
        for (int i = 0; i < n; ++i) used += freq[i] != 0;
        for (int i = 0; used < 2 && i < n; ++i)
            if (freq[i] == 0) {
                freq[i] = 1;
                ++used;
            }
    }
    __syncthreads();
    for (;;) {
        // rank sort of the used symbols by (freq, symbol)
        for (int s = t; s < n; s += 256) {
            const uint32_t f = freq[s];
            if (f) {
                int rank = 0;
                for (int j = 0; j < n; ++j) {
                    const uint32_t g = freq[j];
                    rank += (g != 0) && (g < f || (g == f && j < s));
                }
                T->w[rank] = f;
                T->sym[rank] = (uint16_t)s;
            }
        }
        __syncthreads();
        __shared__ int too_deep;
        if (t == 0) {
            int m = 0;
            for (int i = 0; i < n; ++i) m += freq[i] != 0;
            // two-queue merge: leaves ascending in w[0..m), internal nodes appended at w[288..]
            int li = 0, ii = 0, ni = 0;                       // next leaf, next internal, internal count
            for (int k = 0; k + 1 < m; ++k) {
                uint32_t sum = 0;
                for (int c = 0; c < 2; ++c) {
                    const bool take_leaf = li < m && (ii >= ni || T->w[li] <= T->w[288 + ii]);
                    if (take_leaf) { sum += T->w[li]; T->kid[c][ni] = (uint16_t)li; ++li; }
                    else           { sum += T->w[288 + ii]; T->kid[c][ni] = (uint16_t)(288 + ii); ++ii; }
                }
                T->w[288 + ni] = sum;
                ++ni;
            }
            // depths, root = last internal node
            int deepest = 0;
            T->depth[288 + ni - 1] = 0;
            for (int k = ni - 1; k >= 0; --k) {
                const uint8_t d = (uint8_t)(T->depth[288 + k] + 1);
                for (int c = 0; c < 2; ++c) T->depth[T->kid[c][k]] = d;
                if (d > deepest) deepest = d;
            }
            too_deep = deepest > maxbits;
        }
        __syncthreads();
        if (!too_deep) break;
        for (int s = t; s < n; s += 256)
            if (freq[s]) freq[s] = (freq[s] + 1) >> 1;
        __syncthreads();
    }
    for (int s = t; s < n; s += 256) len[s] = 0;
    __syncthreads();
    __shared__ int m_copy;
    if (t == 0) {
        int m = 0;
        for (int i = 0; i < n; ++i) m += freq[i] != 0;
        m_copy = m;
    }
    __syncthreads();
    for (int r = t; r < m_copy; r += 256) len[T->sym[r]] = T->depth[r];
    __syncthreads();
}

// canonical codes (RFC 1951 3.2.2, gen_codes of trees.c), bit-reversed for LSB-first packing
__device__ void huff_codes(const uint8_t *len, int n, uint16_t *code, int t) {
    for (int s = t; s < n; s += 256) {
        const int l = len[s];
        uint32_t c = 0;
        if (l) {
            // first code of length l = sum over shorter lengths, then rank among equal lengths
            uint32_t cnt[16];
            for (int i = 0; i < 16; ++i) cnt[i] = 0;
            uint32_t before = 0;
            for (int j = 0; j < n; ++j) {
                const int lj = len[j];
                cnt[lj]++;
                before += (lj == l && j < s);
            }
            uint32_t first = 0;
            cnt[0] = 0;
            for (int b = 1; b <= l; ++b) first = (first + cnt[b - 1]) << 1;
            c = __brev(first + before) >> (32 - l);
        }
        code[s] = (uint16_t)c;
    }
    __syncthreads();
}

__device__ __forceinline__ void len_symbol(uint32_t len, uint32_t &sym, uint32_t &eb, uint32_t &ev) {
    const uint32_t l = len - 3;
    eb = 0;
    ev = 0;
    if (l < 8) sym = 257 + l;
    else if (l == 255) sym = 285;
    else {
        const uint32_t lg = 31u - (uint32_t)__clz((int)l);
        eb = lg - 2;
        sym = 257 + 4 * eb + 4 + ((l >> eb) & 3u);
        ev = l & ((1u << eb) - 1u);
    }
}

__device__ __forceinline__ void dist_symbol(uint32_t dist, uint32_t &sym, uint32_t &eb, uint32_t &ev) {
    const uint32_t x = dist - 1;
    eb = 0;
    ev = 0;
    if (x < 4) sym = x;
    else {
        const uint32_t lg = 31u - (uint32_t)__clz((int)x);
        eb = lg - 1;
        sym = 2 * lg + ((x >> eb) & 1u);
        ev = x & ((1u << eb) - 1u);
    }
}

constexpr int kDynPer = 16;
constexpr int kDynTile = 256 * kDynPer;
constexpr int kDynWords = 2 * kDynTile + 16;       // worst case 48 bits per position

__device__ __forceinline__ void lds_put(uint32_t *obuf, uint32_t &cur, uint32_t bits, uint32_t nb) {
    if (!nb) return;
    const uint32_t word = cur >> 5, sh = cur & 31u;
    atomicOr(&obuf[word], bits << sh);
    if (sh + nb > 32u) atomicOr(&obuf[word + 1], bits >> (32u - sh));
    cur += nb;
}

// first token start behind bit `rel` of a segment's bitmap, or hi_rel (the segment's end) if there is none below it
__device__ __forceinline__ uint32_t next_start_rel(const unsigned long long *bm, uint32_t rel, uint32_t hi_rel) {
    uint32_t q = rel + 1u;
    if (q >= hi_rel) return hi_rel;
    unsigned long long w = bm[q >> 6] >> (q & 63u);
    if (w) {
        q += (uint32_t)__builtin_ctzll(w);
        return q < hi_rel ? q : hi_rel;
    }
    q = (q | 63u) + 1u;
    while (q < hi_rel) {
        w = bm[q >> 6];
        if (w) {
            q += (uint32_t)__builtin_ctzll(w);
            return q < hi_rel ? q : hi_rel;
        }
        q += 64u;
    }
    return hi_rel;
}

__global__ __launch_bounds__(256)
void emit_dynamic_kernel(const BlkJob *__restrict__ jobs, const unsigned long long *__restrict__ bm_base,
                         const uint16_t *__restrict__ d16_base, const uint32_t *__restrict__ hist_in,
                         uint32_t *__restrict__ seg_len) {
    __shared__ DynTables T;
    __shared__ uint32_t obuf[kDynWords];
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t sh_cw, sh_cbits, sh_wbase;

    const BlkJob job = jobs[blockIdx.x];
    const uint8_t *in = job.in;
    const uint32_t first = job.seg_start - job.seg_start % kRowBatch;      // position of bit 0 of the segment's bitmap
    const unsigned long long *bm = bm_base + job.bm_off;
    const uint16_t *d16 = d16_base + job.d16_off;
    const uint32_t hi_rel = job.seg_end - first;
    // The tokens that START in [blo, bhi) are this block's; the bytes they cover are [lo, hi): the last token may reach
    // beyond bhi, and the first bytes of the range may still belong to the block before.
    uint32_t lo = job.blo, hi = job.bhi;
    if (lo < hi) {
        const uint32_t r0 = lo - first;
        lo = ((bm[r0 >> 6] >> (r0 & 63u)) & 1ull) ? lo : first + next_start_rel(bm, r0, hi_rel);
        hi = first + next_start_rel(bm, job.bhi - 1u - first, hi_rel);
        if (lo > hi) lo = hi;
    }
    const uint32_t tok_hi = job.bhi;                         // token starts at or beyond this are the next block's
    uint32_t *outw = reinterpret_cast<uint32_t *>(job.out);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    // 1. histogram: counted by K1 while it chose the tokens
    {
        const uint32_t *now = hist_in + (size_t)job.hist_idx * kHistWords, *was = job.hist_prev ? now - kHistWords : nullptr;
        for (int i = t; i < 288; i += 256) T.lfreq[i] = now[i] - (was ? was[i] : 0u) + (i == 256 ? 1u : 0u);     // + end-of-block
        if (t < 32) T.dfreq[t] = now[288 + t] - (was ? was[288 + t] : 0u);
    }
    __syncthreads();
    for (int i = t; i < 288; i += 256) T.lfreq0[i] = T.lfreq[i];
    if (t < 32) T.dfreq0[t] = T.dfreq[t];
    if (t == 0) T.cost_dyn = T.cost_static = 0;

    // 2. code lengths and codes
    huff_lengths(&T, T.lfreq, 286, 15, T.llen, t);
    huff_lengths(&T, T.dfreq, 30, 15, T.dlen, t);
    huff_codes(T.llen, 286, T.lcode, t);
    huff_codes(T.dlen, 30, T.dcode, t);

    // 3. run-length code the lengths (scan_tree / send_tree of trees.c; RFC 1951 3.2.7)
    if (t == 0) {
        int hlit = 286, hdist = 30;
        while (hlit > 257 && T.llen[hlit - 1] == 0) --hlit;
        while (hdist > 1 && T.dlen[hdist - 1] == 0) --hdist;
        T.hlit = hlit;
        T.hdist = hdist;
        for (int i = 0; i < 19; ++i) T.clfreq[i] = 0;
        const int total = hlit + hdist;
        int ncl = 0, i = 0;
        while (i < total) {
            const int v = i < hlit ? T.llen[i] : T.dlen[i - hlit];
            int run = 1;
            while (i + run < total && (i + run < hlit ? T.llen[i + run] : T.dlen[i + run - hlit]) == v) ++run;
            i += run;
            if (v == 0) {
                while (run >= 11) {
                    const int r = run > 138 ? 138 : run;
                    T.clsym[ncl] = 18; T.clext[ncl++] = (uint8_t)(r - 11); T.clfreq[18]++;
                    run -= r;
                }
                if (run >= 3) {
                    T.clsym[ncl] = 17; T.clext[ncl++] = (uint8_t)(run - 3); T.clfreq[17]++;
                    run = 0;
                }
                while (run-- > 0) { T.clsym[ncl] = 0; T.clext[ncl++] = 0; T.clfreq[0]++; }
            } else {
                T.clsym[ncl] = (uint8_t)v; T.clext[ncl++] = 0; T.clfreq[v]++;
                --run;
                while (run >= 3) {
                    const int r = run > 6 ? 6 : run;
                    T.clsym[ncl] = 16; T.clext[ncl++] = (uint8_t)(r - 3); T.clfreq[16]++;
                    run -= r;
                }
                while (run-- > 0) { T.clsym[ncl] = (uint8_t)v; T.clext[ncl++] = 0; T.clfreq[v]++; }
            }
        }
        T.ncl = ncl;
    }
    __syncthreads();
    huff_lengths(&T, T.clfreq, 19, 7, T.cllen, t);
    huff_codes(T.cllen, 19, T.clcode, t);

    // 3b. which block type?  Body cost under both code sets from the histograms (extra bits included), the
    //     dynamic header from its run-length form, the stored size exactly -- the comparison zng_tr_flush_block
    //     makes (trees.c:660-719: stored if not larger, else static if not larger than dynamic, else dynamic).
    {
        uint32_t cd = 0, cs = 0;
        for (int sy = t; sy < 286; sy += 256) {
            const uint32_t f = T.lfreq0[sy];
            if (f) {
                const uint32_t eb = (sy < 265 || sy == 285) ? 0u : (uint32_t)(sy - 261) >> 2;
                const uint32_t sl = sy < 144 ? 8u : (sy < 256 ? 9u : (sy < 280 ? 7u : 8u));   // RFC 1951 3.2.6
                cd += f * (T.llen[sy] + eb);
                cs += f * (sl + eb);
            }
        }
        if (t < 30) {
            const uint32_t f = T.dfreq0[t];
            const uint32_t eb = t < 4 ? 0u : ((uint32_t)t >> 1) - 1u;
            cd += f * (T.dlen[t] + eb);
            cs += f * (5u + eb);
        }
        if (cd) atomicAdd(&T.cost_dyn, cd);
        if (cs) atomicAdd(&T.cost_static, cs);
    }
    for (int i = t; i < kDynWords; i += 256) obuf[i] = 0;
    __syncthreads();
    const uint32_t nbytes = hi - lo;
    const uint32_t nstored = nbytes ? (nbytes + 65534u) / 65535u : 1u;     // stored blocks hold <= 65535 bytes
    if (t == 0) {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        int hclen = 19;
        while (hclen > 4 && T.cllen[order[hclen - 1]] == 0) --hclen;
        T.hclen = (uint32_t)hclen;
        uint32_t hdr = 3 + 5 + 5 + 4 + 3 * (uint32_t)hclen;
        for (uint32_t i = 0; i < T.ncl; ++i) {
            const uint32_t sy = T.clsym[i];
            hdr += T.cllen[sy] + (sy == 16 ? 2u : (sy == 17 ? 3u : (sy == 18 ? 7u : 0u)));
        }
        const uint32_t dyn_lenb = (hdr + T.cost_dyn + 7u) >> 3;
        const uint32_t static_lenb = (3u + T.cost_static + 7u) >> 3;
        const uint32_t opt_lenb = static_lenb <= dyn_lenb ? static_lenb : dyn_lenb;
        const uint32_t stored_lenb = nbytes + 5u * nstored;
        T.mode = stored_lenb <= opt_lenb ? 2u : (static_lenb <= dyn_lenb ? 1u : 0u);
    }
    __syncthreads();
    const uint32_t mode = T.mode;

    if (mode == 2u) {
        // stored blocks (RFC 1951 3.2.4): the segment starts on a byte boundary, so this is a byte copy with a
        // 5-byte header every 65535 bytes
        uint8_t *outb = job.out;
        for (uint32_t k = (uint32_t)t; k < nstored; k += 256) {
            const uint32_t left = nbytes - k * 65535u;
            const uint32_t ln = left < 65535u ? left : 65535u;
            uint8_t *h = outb + (size_t)k * 65540u;
            h[0] = 0;                                                  // BFINAL = 0, BTYPE = 00, padding
            h[1] = (uint8_t)ln;
            h[2] = (uint8_t)(ln >> 8);
            h[3] = (uint8_t)~ln;
            h[4] = (uint8_t)(~ln >> 8);
        }
        for (uint32_t i = (uint32_t)t; i < nbytes; i += 256) outb[i + 5u * (i / 65535u + 1u)] = in[lo + i];
        if (t == 0) {
            uint32_t bytes = nbytes + 5u * nstored;
            if (job.is_last) {                                         // the final empty static block
                outb[bytes++] = 0x03;
                outb[bytes++] = 0x00;
            }
            seg_len[blockIdx.x] = bytes;
        }
        return;
    }
    if (mode == 1u) {                                                  // the static code set, RFC 1951 3.2.6
        for (int sy = t; sy < 288; sy += 256) T.llen[sy] = sy < 144 ? 8 : (sy < 256 ? 9 : (sy < 280 ? 7 : 8));
        if (t < 32) T.dlen[t] = 5;
        __syncthreads();
        huff_codes(T.llen, 288, T.lcode, t);
        huff_codes(T.dlen, 32, T.dcode, t);
    }

    // 4. header bits, serial (a few hundred bits)
    if (t == 0) {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        const int hclen = (int)T.hclen;
        uint32_t cur = 0;
        if (mode == 1u) {
            lds_put(obuf, cur, 0u | (1u << 1), 3);                     // BFINAL = 0, BTYPE = 01 (static)
        } else {
            lds_put(obuf, cur, 0u | (2u << 1), 3);                     // BFINAL = 0, BTYPE = 10 (dynamic)
            lds_put(obuf, cur, T.hlit - 257, 5);
            lds_put(obuf, cur, T.hdist - 1, 5);
            lds_put(obuf, cur, (uint32_t)hclen - 4, 4);
            for (int i = 0; i < hclen; ++i) lds_put(obuf, cur, T.cllen[order[i]], 3);
            for (uint32_t i = 0; i < T.ncl; ++i) {
                const uint32_t sy = T.clsym[i];
                lds_put(obuf, cur, T.clcode[sy], T.cllen[sy]);
                if (sy == 16) lds_put(obuf, cur, T.clext[i], 2);
                else if (sy == 17) lds_put(obuf, cur, T.clext[i], 3);
                else if (sy == 18) lds_put(obuf, cur, T.clext[i], 7);
            }
        }
        // flush whole words of the header, keep the partial one as the carry
        const uint32_t full = cur >> 5;
        for (uint32_t i = 0; i < full; ++i) outw[i] = obuf[i];
        sh_wbase = full;
        sh_cw = obuf[full];
        sh_cbits = cur & 31u;
    }
    __syncthreads();
    uint32_t wbase = sh_wbase, cw = sh_cw, cbits = sh_cbits;

    // 5. body
    for (uint32_t base = lo; base < hi; base += kDynTile) {
        const uint32_t p0 = base + (uint32_t)t * kDynPer;
        uint32_t c1[kDynPer], n1[kDynPer], c2[kDynPer], n2[kDynPer];
        uint32_t mine = 0;
        // this lane's 16 input bytes in one (unaligned) dwordx4 instead of up to 16 byte loads; bytes beyond the end of
        // the stream are never looked at (p < hi below), the slack after the last segment is the caller's 16 bytes
        uint32_t rw[4] = {0u, 0u, 0u, 0u};
        if (p0 + 16u <= hi) {
            const u32x4_unaligned raw = load_u128(in + p0);
            rw[0] = raw.x; rw[1] = raw.y; rw[2] = raw.z; rw[3] = raw.w;
        } else {
            for (uint32_t q = p0; q < hi; ++q) rw[(q - p0) >> 2] |= (uint32_t)in[q] << (8u * ((q - p0) & 3u));
        }
        // the token starts among this lane's 16 positions, and 48 bits of look-ahead for their lengths
        unsigned long long win = 0;
        const uint32_t rel0 = p0 - first;
        if (p0 < tok_hi) {
            const uint32_t sft = rel0 & 63u;
            win = bm[rel0 >> 6] >> sft;
            if (sft) win |= bm[(rel0 >> 6) + 1] << (64u - sft);
        }
#pragma unroll
        for (int j = 0; j < kDynPer; ++j) {
            const uint32_t p = p0 + (uint32_t)j;
            c1[j] = n1[j] = c2[j] = n2[j] = 0;
            if (p < tok_hi && ((win >> j) & 1ull)) {
                const unsigned long long ahead = win >> (j + 1);
                uint32_t nx = ahead ? rel0 + (uint32_t)j + 1u + (uint32_t)__builtin_ctzll(ahead)
                                    : next_start_rel(bm, rel0 + 63u, hi_rel);      // a match that outruns the window
                if (nx > hi_rel) nx = hi_rel;
                const uint32_t len = nx - (rel0 + (uint32_t)j);
                if (len > 1u) {
                    uint32_t sym, eb, ev;
                    len_symbol(len, sym, eb, ev);
                    c1[j] = (uint32_t)T.lcode[sym] | (ev << T.llen[sym]);
                    n1[j] = T.llen[sym] + eb;
                    dist_symbol((uint32_t)d16[(rel0 + (uint32_t)j) >> 2] + 1u, sym, eb, ev);
                    c2[j] = (uint32_t)T.dcode[sym] | (ev << T.dlen[sym]);
                    n2[j] = T.dlen[sym] + eb;
                } else {
                    const uint32_t b = (rw[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    c1[j] = T.lcode[b];
                    n1[j] = T.llen[b];
                }
            }
            mine += n1[j] + n2[j];
        }
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        for (int i = t; i < kDynWords; i += 256) obuf[i] = 0;
        __syncthreads();
        uint32_t wave_off = 0, tile_bits = 0;
        for (int w = 0; w < 4; ++w) {
            if (w < wave) wave_off += wave_tot[w];
            tile_bits += wave_tot[w];
        }
        if (t == 0) obuf[0] = cw;
        __syncthreads();
        uint32_t cur = cbits + wave_off + incl - mine;
#pragma unroll
        for (int j = 0; j < kDynPer; ++j) {
            lds_put(obuf, cur, c1[j], n1[j]);
            lds_put(obuf, cur, c2[j], n2[j]);
        }
        __syncthreads();
        const uint32_t total = cbits + tile_bits;
        const uint32_t full = total >> 5;
        for (uint32_t i = (uint32_t)t; i < full; i += 256) __builtin_nontemporal_store(obuf[i], outw + wbase + i);
        const uint32_t next_cw = obuf[full];
        __syncthreads();
        wbase += full;
        cw = next_cw;
        cbits = total & 31u;
    }

    // 6. end of block, then an empty stored block to reach a byte boundary (and the final block at the very end)
    if (t == 0) {
        uint8_t *outb = job.out;
        unsigned long long acc = cw;                      // bit accumulator
        uint32_t nb = cbits;
        uint32_t bytes = wbase * 4u;
        auto put = [&](uint32_t bits, uint32_t n) {
            acc |= (unsigned long long)bits << nb;
            nb += n;
            while (nb >= 8) {
                outb[bytes++] = (uint8_t)acc;
                acc >>= 8;
                nb -= 8;
            }
        };
        put(T.lcode[256], T.llen[256]);
        put(0, 3);                                        // stored block header, BFINAL = 0
        if (nb) put(0, 8 - nb);                           // pad to a byte
        put(0x0000, 16);
        put(0xffff, 16);                                  // LEN = 0, NLEN = ~0
        if (job.is_last) {
            put(3, 3);                                    // BFINAL = 1, BTYPE = 01
            put(0, 7);                                    // its end-of-block code
            if (nb) put(0, 8 - nb);
        }
        seg_len[blockIdx.x] = bytes;
    }
}

// K3: every segment's place in its stream's output -- an exclusive scan of the segment lengths, restarted per stream --
// plus the per-stream totals and the out_cap check.  One workgroup; `excl` = nseg + 1 u64 of scratch.
//   results[2 * s] = compressed size of stream s, results[2 * s + 1] = 1 if it does not fit its buffer
__global__ __launch_bounds__(1024)
void segments_scan_kernel(const BlkJob *__restrict__ jobs, const uint32_t *__restrict__ seg_len, uint32_t nseg,
                          unsigned long long *__restrict__ excl, unsigned long long *__restrict__ dst_off,
                          unsigned long long *__restrict__ results) {
    __shared__ unsigned long long wave_sum[16];
    __shared__ unsigned long long carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nseg; base += 1024u) {
        const uint32_t k = base + (uint32_t)t;
        const unsigned long long v = k < nseg ? seg_len[k] : 0ull;
        unsigned long long incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long up = (unsigned long long)__shfl_up((long long)incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        unsigned long long before = carry;
        for (int w = 0; w < wave; ++w) before += wave_sum[w];
        if (k < nseg) excl[k] = before + incl - v;
        __syncthreads();
        if (t == 1023) carry = before + incl;
        __syncthreads();
    }
    if (t == 0) excl[nseg] = carry;
    __syncthreads();                                     // one workgroup: its own global stores are visible to it
    for (uint32_t k = (uint32_t)t; k < nseg; k += 1024u) {
        const BlkJob &j = jobs[k];
        const unsigned long long off = excl[k] - excl[j.first_seg];
        dst_off[k] = off;
        if (j.stream & 0x80000000u) {
            const uint32_t srow = j.stream & 0x7fffffffu;
            const unsigned long long total = off + seg_len[k];
            results[2 * srow] = total;
            results[2 * srow + 1] = total > j.dst_cap ? 1ull : 0ull;
        }
    }
}

// K4: pack the segments back to back (byte granular); a stream that does not fit its buffer is cut at the buffer's end
// (K3 has flagged it)
__global__ __launch_bounds__(256)
void gather_segments_kernel(const BlkJob *__restrict__ jobs, const uint32_t *__restrict__ seg_len,
                            const unsigned long long *__restrict__ dst_off) {
    const BlkJob job = jobs[blockIdx.y];
    const unsigned long long off = dst_off[blockIdx.y];
    unsigned long long n = seg_len[blockIdx.y];
    if (off >= job.dst_cap) return;
    if (off + n > job.dst_cap) n = job.dst_cap - off;
    uint8_t *d = job.dst + off;
    for (unsigned long long i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        d[i] = job.out[i];
}

// Level 0: deflate_stored (deflate_stored.c:27-186) for a complete, device-resident plaintext and an output buffer
// that holds everything -- the case in which the reference copies straight from next_in to next_out in stored blocks
// of MAX_STORED = 65535 bytes (deflate_stored.c:46-95), the last one carrying BFINAL.  A byte copy with a 5-byte
// header (RFC 1951 3.2.4) in front of every 65535 bytes: 2N of HBM traffic, 16 bytes per lane.
constexpr uint32_t kMaxStored = 65535u;

__global__ __launch_bounds__(256)
void stored_kernel(const uint8_t *__restrict__ in, size_t n, uint8_t *__restrict__ out, int final_block) {
    const size_t nblk = n ? (n + kMaxStored - 1) / kMaxStored : 1;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (size_t)gridDim.x * blockDim.x;
    for (size_t k = gid; k < nblk; k += gsz) {
        const size_t left = n - k * kMaxStored;
        const uint32_t ln = left < kMaxStored ? (uint32_t)left : kMaxStored;
        uint8_t *h = out + k * (kMaxStored + 5u);
        h[0] = (k + 1 == nblk && final_block) ? 1 : 0;              // BFINAL, BTYPE = 00, padding bits
        h[1] = (uint8_t)ln;
        h[2] = (uint8_t)(ln >> 8);
        h[3] = (uint8_t)~ln;
        h[4] = (uint8_t)(~ln >> 8);
    }
    // payload: lane i moves input bytes [16 i, 16 i + 16); a piece that straddles a block border is split by bytes
    const size_t npiece = (n + 15) / 16;
    for (size_t i = gid; i < npiece; i += gsz) {
        const size_t p = i * 16;
        const size_t blk = p / kMaxStored;
        uint8_t *dst = out + p + 5u * (blk + 1);
        if (p + 16 <= n && (p + 15) / kMaxStored == blk) {
            const u32x4_unaligned v = load_u128(in + p);
            *reinterpret_cast<u32x4_unaligned *>(dst) = v;
        } else {
            for (size_t q = p; q < p + 16 && q < n; ++q) out[q + 5u * (q / kMaxStored + 1)] = in[q];
        }
    }
}

static inline size_t seg_slot_bytes(uint32_t n) {
    // <= 9 bits per literal for an almost flat alphabet, + header (< 400 bytes) + trailer; 4-byte aligned
    return (((size_t)n * 9 + 7) / 8 + 1024 + 3) & ~(size_t)3;
}

unsigned long long *g_rows_stamps = nullptr;      // ZR_ROWS_STAMPS builds: device buffer of phase time stamps

// candidates looked at per position, per level: the row form of max_chain_length (deflate.c:142-168: 4, 6, 24, 32, 128
// links at levels 2..6); level 1 = deflate_quick's single probe (deflate_quick.c:89-97)
static const uint32_t kLevelCand[10] = {0, 1, 2, 3, 6, 8, 16, 16, 16, 16};

// The launches of one round: `njobs` streams, all their segments in one job list.  Asynchronous on `st`; *results =
// pinned host words {compressed size, does-not-fit flag} per stream, valid once the stream has been synchronised.
// cap_override: out_cap of job 0 when the caller's job struct cannot hold it (a single stream of >= 4 GiB of room).
static int deflate_rows_enqueue(int level, const zng_rocm_stream_job *sjobs, size_t njobs, const size_t *cap_override,
                                uint32_t seg_bytes, Workspace *ws, hipStream_t st, unsigned long long **results,
                                unsigned long long *d_results_copy = nullptr) {
    size_t nseg = 0;
    for (size_t s = 0; s < njobs; ++s) nseg += sjobs[s].in_len ? ((size_t)sjobs[s].in_len + seg_bytes - 1) / seg_bytes : 1;
    const size_t max_blk = nseg * kMaxSub;
    SegJob *d_jobs = nullptr, *jobs = nullptr;
    BlkJob *d_blk = nullptr, *blk = nullptr;
    uint32_t *d_seg_len = nullptr, *d_hist = nullptr;
    unsigned long long *d_scan = nullptr, *d_bm = nullptr, *d_res = nullptr, *h_res = nullptr;
    uint16_t *d_d16 = nullptr;
    uint8_t *d_slots = nullptr;
    if (int rc = host_tables_acquire(ws)) return rc;
    // pinned: SegJob[nseg] | BlkJob[max_blk]
    const size_t seg_tab = (nseg * sizeof(SegJob) + 255) & ~(size_t)255;
    if (int rc = scratch_reserve(ws, kScrDynJobsHost, seg_tab + max_blk * sizeof(BlkJob), true, (void **)&jobs)) return rc;
    blk = reinterpret_cast<BlkJob *>(reinterpret_cast<uint8_t *>(jobs) + seg_tab);
    if (int rc = scratch_reserve(ws, kScrDynSegLenHost, njobs * 2 * sizeof(unsigned long long), true, (void **)&h_res)) return rc;
    if (int rc = scratch_reserve(ws, kScrDynJobs, seg_tab + max_blk * sizeof(BlkJob), false, (void **)&d_jobs)) return rc;
    d_blk = reinterpret_cast<BlkJob *>(reinterpret_cast<uint8_t *>(d_jobs) + seg_tab);
    if (int rc = scratch_reserve(ws, kScrDynSegLen, max_blk * sizeof(uint32_t) + 64 + nseg * (size_t)kMaxSub * kHistWords * sizeof(uint32_t), false, (void **)&d_seg_len))
        return rc;
    d_hist = d_seg_len + ((max_blk + 3) & ~(size_t)3);
    if (int rc = scratch_reserve(ws, kScrDynDstOff, (2 * max_blk + 2 + 2 * njobs) * sizeof(unsigned long long), false, (void **)&d_scan))
        return rc;
    size_t slot_total = 0, bm_total = 0, d16_total = 0, k = 0, nb = 0;
    for (size_t s = 0; s < njobs; ++s) {
        const zng_rocm_stream_job &j = sjobs[s];
        const uint32_t dict = j.dict_len;
        const bool final_block = (j.flags & ZNG_ROCM_BLOCK_NOT_FINAL) == 0;
        const size_t n = j.in_len ? ((size_t)j.in_len + seg_bytes - 1) / seg_bytes : 1;
        const size_t b0 = nb;
        for (size_t i = 0; i < n; ++i, ++k) {
            // positions count from the first dictionary byte: the segments' own 32 KiB priming reaches into it
            const uint32_t a = dict + (uint32_t)(i * seg_bytes);
            const uint32_t b = dict + (uint32_t)((i + 1) * (size_t)seg_bytes < j.in_len ? (i + 1) * (size_t)seg_bytes : j.in_len);
            jobs[k].in = (const uint8_t *)j.in - dict;
            jobs[k].out = nullptr;
            jobs[k].dst = (uint8_t *)j.out;
            jobs[k].dst_cap = 0;
            jobs[k].seg_start = a;
            jobs[k].seg_end = b;
            jobs[k].out_cap = 0;
            jobs[k].is_last = 0;
            jobs[k].first_seg = 0;
            jobs[k].stream = (uint32_t)s;
            jobs[k].bm_off = bm_total;
            jobs[k].d16_off = d16_total;
            // the segment's blocks: one per kSubBytes of positions counted from its first batch (the matcher snapshots
            // its histogram at the same places); a block without positions of its own is left out
            const uint32_t first = a - a % kRowBatch;
            const uint32_t nsub = b > first ? (b - first + kSubBytes - 1) / kSubBytes : 1u;
            bool prev_written = false;
            for (uint32_t sub = 0; sub < nsub; ++sub) {
                const uint32_t p0 = first + sub * kSubBytes, p1 = p0 + kSubBytes;
                const uint32_t blo = p0 > a ? p0 : a, bhi = p1 < b ? p1 : b;
                if (blo >= bhi && !(b == a && sub + 1 == nsub)) {          // (an empty segment still gets its one block)
                    prev_written = true;                                  // its snapshot exists all the same
                    continue;
                }
                BlkJob &q = blk[nb++];
                q.in = jobs[k].in;
                q.dst = (uint8_t *)j.out;
                q.dst_cap = cap_override && s == 0 ? *cap_override : j.out_cap;
                q.bm_off = bm_total;
                q.d16_off = d16_total;
                q.seg_start = a;
                q.seg_end = b;
                q.blo = blo;
                q.bhi = bhi;
                q.hist_idx = (uint32_t)(k * kMaxSub + sub);
                q.hist_prev = sub > 0 && prev_written ? 1u : 0u;
                q.out_cap = (uint32_t)seg_slot_bytes(bhi - blo + 258u);
                q.is_last = 0;
                q.first_seg = (uint32_t)b0;
                q.stream = (uint32_t)s;
                q.out = (uint8_t *)slot_total;            // offset for now
                slot_total += q.out_cap;
                prev_written = true;
            }
            bm_total += seg_bm_words(b - a);
            d16_total += seg_d16_entries(b - a);
        }
        blk[nb - 1].stream |= 0x80000000u;               // the stream's last block ...
        blk[nb - 1].is_last = final_block ? 1u : 0u;      // ... carries BFINAL
    }
    unsigned long long *d_excl = d_scan, *d_dst_off = d_scan + nb + 1;
    d_res = d_dst_off + nb + 1;
    if (int rc = scratch_reserve(ws, kScrDynSlots, slot_total, false, (void **)&d_slots)) return rc;
    // token scratch of the matcher: 1 bit + half a u16 per position (K1 -> K2)
    if (int rc = scratch_reserve(ws, kScrDynSel, bm_total * sizeof(unsigned long long) + d16_total * sizeof(uint16_t), false, (void **)&d_bm))
        return rc;
    d_d16 = reinterpret_cast<uint16_t *>(d_bm + bm_total);
    for (size_t i = 0; i < nb; ++i) blk[i].out = d_slots + (size_t)blk[i].out;
    ZR_HIP(hipMemcpyAsync(d_jobs, jobs, seg_tab + nb * sizeof(BlkJob), hipMemcpyHostToDevice, st));

    ZR_LAUNCH_TRACED(lz_rows_kernel, dim3((unsigned)nseg), dim3(kRowBatch), st, d_jobs, d_bm, d_d16, d_hist, kLevelCand[level], g_rows_stamps);
    ZR_HIP(hipGetLastError());
    hipLaunchKernelGGL(emit_dynamic_kernel, dim3((unsigned)nb), dim3(256), 0, st, d_blk, d_bm, d_d16, d_hist, d_seg_len);
    ZR_HIP(hipGetLastError());
    hipLaunchKernelGGL(segments_scan_kernel, dim3(1), dim3(1024), 0, st, d_blk, d_seg_len, (uint32_t)nb, d_excl, d_dst_off, d_res);
    ZR_HIP(hipGetLastError());
    hipLaunchKernelGGL(gather_segments_kernel, dim3(njobs > 64 ? 2 : 4, (unsigned)nb), dim3(256), 0, st, d_blk, d_seg_len, d_dst_off);
    ZR_HIP(hipGetLastError());
    if (d_results_copy)
        ZR_HIP(hipMemcpyAsync(d_results_copy, d_res, njobs * 2 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));
    else
        ZR_HIP(hipMemcpyAsync(h_res, d_res, njobs * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    if (int rc = host_tables_release(ws, st)) return rc;
    *results = h_res;
    return ZNG_ROCM_OK;
}

}  // namespace zr

using namespace zr;

extern "C" {

#ifdef ZR_ROWS_STAMPS
// diagnostic builds only (tools/micro/rows_stamps.sh): where lz_rows_kernel writes its phase time stamps
// (32 batches x 16 waves x 9 stamps of workgroup 3)
void zng_rocm_debug_rows_stamps(void *d_buf) { g_rows_stamps = (unsigned long long *)d_buf; }
#endif

size_t zng_rocm_deflate_bound(size_t source_len) {
    const size_t nseg = source_len ? (source_len + kSegBytesMin - 1) / kSegBytesMin : 1;     // the most there can be
    // covers level 0 too: 5 bytes per 65535 (deflate_stored) is far below the n/8 term
    return source_len + source_len / 8 + nseg * 1032 + 16;
}

int zng_rocm_deflate_dev(int level, const uint8_t *d_in, size_t in_len, uint8_t *d_out, size_t out_cap,
                         size_t *out_len, void *stream) {
    return zng_rocm_deflate_block_dev(level, d_in, in_len, 0, 0, d_out, out_cap, out_len, stream);
}

int zng_rocm_deflate_block_dev(int level, const uint8_t *d_in, size_t in_len, uint32_t dict_len, uint32_t flags,
                               uint8_t *d_out, size_t out_cap, size_t *out_len, void *stream) {
    if (!ctx()) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    if (!out_len || !d_out || (!d_in && in_len)) return ZNG_ROCM_EINVAL;
    if (level < 0 || level > 9) {
        set_error("level %d is outside 0..9", level);
        return ZNG_ROCM_EINVAL;
    }
    if (dict_len > kPrime || (flags & ~(uint32_t)(ZNG_ROCM_BLOCK_NOT_FINAL | ZNG_ROCM_BLOCK_SYNC_FLUSH))) {
        set_error("dict_len above 32768 or unknown flags");
        return ZNG_ROCM_EINVAL;
    }
    const bool final_block = (flags & ZNG_ROCM_BLOCK_NOT_FINAL) == 0;
    if (in_len + dict_len >= (1ull << 32) - kSegBytes) {
        set_error("streams of 4 GiB and more are not supported by the 32-bit position format");
        return ZNG_ROCM_EINVAL;
    }
    if (out_cap < zng_rocm_deflate_bound(in_len)) {
        set_error("out_cap below zng_rocm_deflate_bound()");
        return -5;
    }
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard dev;
    // scratch is keyed by the caller's HIP stream (context.h): concurrent callers on different streams share nothing
    Workspace *ws = workspace_for(st);
    if (!ws) return ZNG_ROCM_ENOMEM;
    std::lock_guard<std::mutex> use(ws->mu);
    // max_chain_length per level, deflate.c:142-168
    if (level == 0) {                                               // deflate_stored: no scratch, no host round trip
        const size_t nblk = in_len ? (in_len + kMaxStored - 1) / kMaxStored : 1;
        const bool marker = !final_block && (flags & ZNG_ROCM_BLOCK_SYNC_FLUSH);   // deflate.c:1064-1076
        const size_t total0 = in_len + 5 * nblk + (marker ? 5 : 0);
        if (total0 > out_cap) {
            set_error("stored size %llu exceeds out_cap", (unsigned long long)total0);
            return -5;
        }
        const size_t pieces = (in_len + 15) / 16 + nblk;
        unsigned grid = (unsigned)((pieces + 255) / 256);
        if (grid > 65536u) grid = 65536u;
        if (grid == 0) grid = 1;
        ZR_LAUNCH_TRACED(stored_kernel, dim3(grid), dim3(256), st, d_in, in_len, d_out, final_block ? 1 : 0);
        ZR_HIP(hipGetLastError());
        if (marker) {
            static const uint8_t empty_stored[5] = {0x00, 0x00, 0x00, 0xff, 0xff};
            ZR_HIP(hipMemcpyAsync(d_out + in_len + 5 * nblk, empty_stored, 5, hipMemcpyHostToDevice, st));
        }
        ZR_HIP(hipStreamSynchronize(st));
        *out_len = total0;
        return ZNG_ROCM_OK;
    }
    const size_t seg_bytes = segment_bytes(in_len, ctx()->cus);
    zng_rocm_stream_job one;
    one.in = d_in;
    one.out = d_out;
    one.in_len = (uint32_t)in_len;
    one.out_cap = 0;
    one.dict_len = dict_len;
    one.flags = flags;
    unsigned long long *res = nullptr;
    if (int rc = deflate_rows_enqueue(level, &one, 1, &out_cap, (uint32_t)seg_bytes, ws, st, &res)) return rc;
    ZR_HIP(hipStreamSynchronize(st));
    if (res[1]) {
        set_error("compressed size %llu exceeds out_cap", res[0]);
        return -5;
    }
    *out_len = (size_t)res[0];
    return ZNG_ROCM_OK;
}

// The same with NO synchronisation: everything -- matcher, block emitter, the scan that places the segments, the packing --
// is enqueued on `stream` and the call returns; d_result (device, 2 x u64) receives {compressed size, 1 if it did not fit
// out_cap} when the stream gets there.  Levels 1..9.
int zng_rocm_deflate_async_dev(int level, const uint8_t *d_in, size_t in_len, uint32_t dict_len, uint32_t flags, uint8_t *d_out,
                               size_t out_cap, uint64_t *d_result, void *stream) {
    if (!ctx()) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    if (!d_result || !d_out || (!d_in && in_len) || level < 1 || level > 9 || dict_len > kPrime ||
        (flags & ~(uint32_t)(ZNG_ROCM_BLOCK_NOT_FINAL | ZNG_ROCM_BLOCK_SYNC_FLUSH)) || in_len + dict_len >= (1ull << 32) - kSegBytes)
        return ZNG_ROCM_EINVAL;
    if (out_cap < zng_rocm_deflate_bound(in_len)) {
        set_error("out_cap below zng_rocm_deflate_bound()");
        return -5;
    }
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard dev;
    Workspace *ws = workspace_for(st);
    if (!ws) return ZNG_ROCM_ENOMEM;
    std::lock_guard<std::mutex> use(ws->mu);
    zng_rocm_stream_job one;
    one.in = d_in;
    one.out = d_out;
    one.in_len = (uint32_t)in_len;
    one.out_cap = 0;
    one.dict_len = dict_len;
    one.flags = flags;
    unsigned long long *res = nullptr;
    return deflate_rows_enqueue(level, &one, 1, &out_cap, segment_bytes(in_len, ctx()->cus), ws, st, &res,
                                (unsigned long long *)d_result);
}

// Many independent streams at one of the chain levels (the reference's many-stream model, test/pigz/CMakeLists.txt:123-200,
// at pigz's default level): every stream is cut into segments as in zng_rocm_deflate_block_dev and the segments of ALL
// streams go through ONE set of launches per round (a round holds up to ~4 GiB of plaintext).  Synchronous, like
// zng_rocm_deflate_block_dev; out_lens is a host array.
int zng_rocm_deflate_streams_dev(int level, const zng_rocm_stream_job *sjobs, size_t njobs, size_t *out_lens, void *stream) {
    if (!ctx()) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    if (!njobs) return ZNG_ROCM_OK;
    if (!sjobs || !out_lens) return ZNG_ROCM_EINVAL;
    if (level < 1 || level > 9) {
        set_error("level %d is outside 1..9 (level 0: zng_rocm_deflate_block_dev per stream)", level);
        return ZNG_ROCM_EINVAL;
    }
    size_t total_in = 0;
    for (size_t i = 0; i < njobs; ++i) {
        const zng_rocm_stream_job &j = sjobs[i];
        if (!j.out || (j.in_len && !j.in) || (j.dict_len && !j.in) || j.dict_len > kPrime ||
            (j.flags & ~(uint32_t)(ZNG_ROCM_BLOCK_NOT_FINAL | ZNG_ROCM_BLOCK_SYNC_FLUSH)) ||
            (uint64_t)j.in_len + j.dict_len >= (1ull << 32) - kSegBytes) {
            set_error("job %zu: null buffer, dict_len above 32768, unknown flags, or a stream of 4 GiB and more", i);
            return ZNG_ROCM_EINVAL;
        }
        if (j.out_cap < zng_rocm_deflate_bound(j.in_len)) {
            set_error("job %zu: out_cap below zng_rocm_deflate_bound()", i);
            return -5;
        }
        total_in += j.in_len;
    }
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard dev;
    Workspace *ws = workspace_for(st);
    if (!ws) return ZNG_ROCM_ENOMEM;
    std::lock_guard<std::mutex> use(ws->mu);
    const uint32_t seg_bytes = segment_bytes(total_in, ctx()->cus);
    constexpr size_t kRoundBytes = 4ull << 30;

    size_t first = 0;
    while (first < njobs) {
        size_t last = first, bytes = 0;
        while (last < njobs && (last == first || bytes + sjobs[last].in_len + sjobs[last].dict_len <= kRoundBytes)) {
            bytes += sjobs[last].in_len + sjobs[last].dict_len;
            ++last;
        }
        unsigned long long *res = nullptr;
        if (int rc = deflate_rows_enqueue(level, sjobs + first, last - first, nullptr, seg_bytes, ws, st, &res)) return rc;
        ZR_HIP(hipStreamSynchronize(st));
        for (size_t s = first; s < last; ++s) {
            if (res[2 * (s - first) + 1]) {
                set_error("job %zu: compressed size %llu exceeds out_cap", s, res[2 * (s - first)]);
                return -5;
            }
            out_lens[s] = (size_t)res[2 * (s - first)];
        }
        first = last;
    }
    return ZNG_ROCM_OK;
}

}  // extern "C"
