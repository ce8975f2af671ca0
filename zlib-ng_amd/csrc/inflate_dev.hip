// inflate_dev.hip -- slot `inflate_fast` (inffast_tpl.h:53-318) together with the block decoding around it
// (inflate.c:735-917: stored / fixed / dynamic block headers, inftrees.c:32-297 table construction) as ONE device
// kernel for MANY independent raw deflate streams that are already in device memory -- the shape of the reference's
// many-stream model (test/pigz/CMakeLists.txt:123-200) on the decode side, and the inverse of
// zng_rocm_deflate_quick_dev: compress on the device, decompress on the device, nothing crosses PCIe.
//
// One wavefront per stream.  A deflate stream is a serial bit parse, so the parse itself is wave-UNIFORM work: the
// bit buffer, the table entry, the position all live in scalar registers (every lane would compute the same), and the
// 64 lanes are used where the stream offers width:
//   * the compressed words are fetched 64 at a time (one dword per lane, the next 64 prefetched) and handed to the
//     bit buffer with v_readlane -- the parse never waits on a memory load;
//   * Huffman tables are built lane-parallel (counting sort by ballots, one table entry per lane per pass);
//   * a match copy moves up to 64 bytes per pass, a stored block 1 KiB per pass;
//   * the last kRing output bytes live in an LDS ring (the device form of inflate's sliding window, inflate.c:325-378):
//     literals and near matches never touch HBM, the ring leaves in aligned 16-byte stores, and only a match that
//     reaches further back than the ring reads its source from HBM (the stream's own earlier output, or the
//     dictionary / previous window that precedes `out`).
// This kernel scales with the NUMBER of streams; ONE large stream is cut into parts at block starts found on the device
// and every part is a job of this kernel in part mode (inflate_large.hip).  The decode loop itself is hand-written
// (ZR_INFLATE_FAST_LOOP below).
//
// Status and messages are the reference's (inflate.c strm->msg texts, via zng_rocm_inflate_message).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <type_traits>

#include "context.h"
#include "deflate_dev.h"
#include "inflate_dev.h"

namespace zr {

// Diagnostic builds (-DZR_INFLATE_BOUNDS, tools/micro/inflate_bounds.sh): every index the table builder computes from
// stream data is checked against the array it goes into; a violation is counted (zng_rocm_debug_inflate_bounds()) and the
// index clamped.  Round 2's aperture violation was traced to addressing, not to an index; this build is the
// evidence that no index leaves its array on the mutated-stream corpus either.
// Diagnostic builds (-DZR_INFLATE_STATS, tools/micro/inflate_stats.sh): how often the hand-written loop hands a symbol to the
// general code, and why (zng_rocm_debug_inflate_stats()).
#ifdef ZR_INFLATE_STATS
__device__ unsigned long long g_inflate_stats[16];
__device__ unsigned long long g_inflate_span[2 * 16384];         // per job (the first 16384): s_memtime at its start and end
#ifdef ZR_INFLATE_STATS_EXITS
#define ZR_STAT(i) do { if (lane == 0) atomicAdd(&g_inflate_stats[i], 1ull); } while (0)
#else
#define ZR_STAT(i) do { } while (0)
#endif
#else
#define ZR_STAT(i) do { } while (0)
#endif
#ifdef ZR_INFLATE_BOUNDS
__device__ unsigned int g_inflate_bounds_violations;
__device__ __forceinline__ uint32_t zr_idx(uint32_t i, uint32_t n) {
    if (i < n) return i;
    atomicAdd(&g_inflate_bounds_violations, 1u);
    return 0;
}
#define ZR_IDX(i, n) zr_idx((uint32_t)(i), (uint32_t)(n))
#else
#define ZR_IDX(i, n) (i)
#endif

// Table entry, 16 bits: code length in bits 0-3, SYMBOL in bits 4-15 (literal/length 0..287, distance 0..31, code-length
// symbol 0..18).  Base values and extra-bit counts are arithmetic in the symbol (length_of / distance_of), so nothing
// wider is needed: the three tables of a stream take 3.3 KiB of LDS, and LDS per wave is what bounds how many streams
// a CU decodes at once.  kLongMark: the code is longer than the table's root; kBadMark: no code has this prefix.
constexpr uint32_t kLongMark = (0xfffu << 4), kBadMark = (0xffeu << 4) | 1u;
__device__ __forceinline__ uint32_t make_entry(uint32_t nbits, uint32_t sym) { return nbits | (sym << 4); }

constexpr int kLitRootStream = 10, kLitRootPart = 10, kClRoot = 7;   // (a 9-bit literal root in part mode -- a 13th wave per CU -- was slower: 6.1 -> 6.8 ms, too many codes take the long path)
// root bits of the distance table: 9 for whole streams (16 per CU fit either way), 8 in part mode, where the 1 KiB it saves
// is the twelfth wave of a CU (part kernel 6.6 -> 6.1 ms; whole streams lose 2.5 % with 8: more codes take the long path)
constexpr int kDistRootStream = 9, kDistRootPart = 8;
// order in which the code-length code's lengths are sent (RFC 1951 3.2.7; inflate.c:832-833 holds the same permutation)
__device__ const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
constexpr int kCodeLit = 0, kCodeDist = 1, kCodeCl = 2;

// Whole streams, and the parts of a call with few of them: every table has its own place (9.9 KiB with a ring of bytes,
// exactly 16 streams per CU; 12.9 KiB with a ring of 16-bit symbols, 12 parts per CU).
template <int RING, typename T, int DR, int LR>
struct InflateLdsStream {
    uint16_t lit[1 << LR];
    uint32_t dist[1 << DR];             // wide entries: code length | extra bits << 4 | base distance << 8 (wide_distance)
    uint16_t cl_[1 << kClRoot];
    uint32_t cnt_[3][16], first_[3][16], offs_[3][16], run[16];
    uint16_t sorted_lit[288], sorted_dist[32], sorted_cl_[32];
    uint8_t  lens[320 + 8];
    uint8_t  cl_lens_[24];
    T        ring[RING] __attribute__((aligned(16)));
    __device__ __forceinline__ uint32_t *cnt(int which) { return cnt_[which]; }
    __device__ __forceinline__ uint32_t *first(int which) { return first_[which]; }
    __device__ __forceinline__ uint32_t *offs(int which) { return offs_[which]; }
    __device__ __forceinline__ uint16_t *cl() { return cl_; }
    __device__ __forceinline__ uint16_t *sorted_cl() { return sorted_cl_; }
    __device__ __forceinline__ uint8_t *cl_lens() { return cl_lens_; }
};
// Parts, when there are many (16-bit symbols: the ring alone is 8 KiB): LDS is what limits how many decode at once, so
// the layout is packed.
// The code-length code -- root table, sorted symbols, counts, the 19 lengths -- only lives while a dynamic header is read,
// and the distance table is only built when that is over: one place for both; the builder's first-code and offset words
// are 16 bits.  12.2 KiB: a 13th part per CU (part kernel of the cfg3 stream 6.0 -> 5.25 ms).  (The same layout for
// whole streams cost their kernel 5 % -- 42.1 -> 39.8 GB/s on the level-1 class's streams, with the LDS size padded
// back to what it was -- so they keep theirs.)
template <int RING, typename T, int DR, int LR>
struct InflateLdsPart {
    uint16_t lit[1 << LR];
    union {
        uint32_t dist[1 << DR];
        struct {
            uint16_t cl[1 << kClRoot];
            uint16_t sorted_cl[32];
            uint32_t cnt[16];
            uint16_t first[16], offs[16];
            uint8_t  cl_lens[24];
        } h;
    };
    uint32_t cnt_[2][16];               // [code][length]; slot 0 of a code: its longest length
    uint16_t first_[2][16], offs_[2][16], run[16];
    uint16_t sorted_lit[288], sorted_dist[32];
    uint8_t  lens[320 + 8];
    T        ring[RING] __attribute__((aligned(16)));
    __device__ __forceinline__ uint32_t *cnt(int which) { return which == kCodeCl ? h.cnt : cnt_[which]; }
    __device__ __forceinline__ uint16_t *first(int which) { return which == kCodeCl ? h.first : first_[which]; }
    __device__ __forceinline__ uint16_t *offs(int which) { return which == kCodeCl ? h.offs : offs_[which]; }
    __device__ __forceinline__ uint16_t *cl() { return h.cl; }
    __device__ __forceinline__ uint16_t *sorted_cl() { return h.sorted_cl; }
    __device__ __forceinline__ uint8_t *cl_lens() { return h.cl_lens; }
};
static_assert(sizeof(((InflateLdsPart<4096, uint16_t, 8, 10> *)nullptr)->h) <= sizeof(uint32_t) << 8, "the header's tables must fit the distance table's place");


// base values / extra bits of the length and distance symbols (RFC 1951 3.2.5; inftrees.c:38-49 hold the same numbers)
__device__ __forceinline__ void length_of(uint32_t k, uint32_t *base, uint32_t *extra) {     // k = symbol - 257, 0..28
    if (k < 8) { *base = 3 + k; *extra = 0; }
    else if (k == 28) { *base = 258; *extra = 0; }
    else { const uint32_t e = (k - 4) >> 2; *extra = e; *base = 3 + ((4 + (k & 3)) << e); }
}
__device__ __forceinline__ void distance_of(uint32_t k, uint32_t *base, uint32_t *extra) {   // k = symbol, 0..29
    if (k < 4) { *base = 1 + k; *extra = 0; }
    else { const uint32_t e = (k - 2) >> 1; *extra = e; *base = 1 + ((2 + (k & 1)) << e); }
}

// the distance table's decode-loop form: base and extra-bit count ride in the entry (the loop is scalar-issue bound and
// the arithmetic of distance_of is 11 scalar instructions per match; 1 KiB more LDS per stream)
// kBadWide | n: an invalid code of n bits -- symbols 30, 31, or the unused entry of an incomplete set (1 bit).  The length
// matters at the end of a truncated stream: the reference reports an invalid code only when all of its bits are input
// (inflate.c's NEEDBITS / PULLBYTE loops ask for more first), and bits behind the input read as zeros here.
constexpr uint32_t kLongWide = 0xfffffff0u, kBadWide = 0xffffffe0u;
__device__ __forceinline__ uint32_t wide_distance(uint32_t e) {           // e: a 16-bit entry
    if (e == kLongMark) return kLongWide;
    const uint32_t sym = e >> 4;
    if (sym > 29u) return kBadWide | (e & 15u);                               // 30, 31 (inftrees.c:48-49), no code
    uint32_t b = 0, x = 0;
    distance_of(sym, &b, &x);
    return (e & 15u) | (x << 4) | (b << 8);
}

__device__ __forceinline__ void wave_sync() {            // LDS written by some lanes is read by others of the same wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
typedef uint32_t u32x4_v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// Canonical code from `n` code lengths (inftrees.c:32-297 re-thought for a wavefront): per-length counts, the
// over-subscribed / incomplete checks of inftrees.c:104-137, symbols sorted by (length, symbol), and the root-bit
// primary table, one entry per lane per pass, each found by the canonical comparison "code - first[L] < count[L]".
// Codes longer than the root get a kLong entry; the decode loop resolves those with the same comparison (they are the
// rare symbols by construction).  Returns 0, or 1 for an invalid set.
template <typename LDS>
__device__ __forceinline__ int build_code(LDS &L, int which, const uint8_t *lens, int n, int root, uint16_t *table,
                          uint16_t *sorted, int lane) {
    auto *cnt = L.cnt(which);
    auto *first = L.first(which);
    auto *offs = L.offs(which);
    if (lane < 16) cnt[lane] = 0;
    wave_sync();
    for (int s = lane; s < n; s += 64) atomicAdd(&cnt[ZR_IDX(lens[s], 16)], 1u);
    wave_sync();
    const uint32_t mine = lane < 16 ? cnt[lane] : 0u;
    int left = 1, max = 0;
    uint32_t code = 0, off = 0;
    bool over = false;
    for (int len = 1; len <= 15; ++len) {
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)mine, len);
        left = (left << 1) - (int)c;
        if (left < 0) over = true;
        if (lane == 0) {
            first[len] = (std::remove_reference_t<decltype(first[0])>)code;      // (< 2^len for every set that is not rejected)
            offs[len] = (std::remove_reference_t<decltype(offs[0])>)off;
            L.run[len] = (std::remove_reference_t<decltype(L.run[0])>)off;
        }
        code = (code + c) << 1;
        off += c;
        if (c) max = len;
    }
    if (over) return 1;                                              // over-subscribed (inftrees.c:128-132)
    if (max == 0) {                                                  // no codes at all (inftrees.c:114-122): every entry invalid
        for (int e = lane; e < (1 << root); e += 64) table[e] = (uint16_t)(which == kCodeCl ? make_entry(1, 0) : kBadMark);
        if (lane == 0) cnt[0] = 0;
        wave_sync();
        return 0;
    }
    if (left > 0 && (which == kCodeCl || max != 1)) return 1;        // incomplete set (inftrees.c:133-134)
    wave_sync();
    // stable counting sort: rank of a symbol among the symbols of its length = lower lanes of the same ballot
    for (int base = 0; base < n; base += 64) {
        const int s = base + lane;
        const uint32_t l = s < n ? lens[s] : 0u;
        bool active = l != 0;
        unsigned long long m = __ballot(active);
        while (m) {
            const int f = __builtin_ctzll(m);
            const uint32_t lu = (uint32_t)__builtin_amdgcn_readlane((int)l, f);
            const bool hit = active && l == lu;
            const unsigned long long same = __ballot(hit);
            const uint32_t at = uni(L.run[ZR_IDX(lu, 16)]);
            if (hit) sorted[ZR_IDX(at + (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull)), which == kCodeLit ? 288 : 32)] = (uint16_t)s;
            if (lane == f) L.run[ZR_IDX(lu, 16)] = (std::remove_reference_t<decltype(L.run[0])>)(at + (uint32_t)__builtin_popcountll(same));
            active = active && !hit;
            m &= ~same;
            wave_sync();
        }
    }
    wave_sync();
    const int top = max < root ? max : root;
    for (int e = lane; e < (1 << root); e += 64) {
        const uint32_t rev = __builtin_bitreverse32((uint32_t)e) >> (32 - root);      // the root bits as a code prefix
        uint32_t ent = max > root ? kLongMark : kBadMark;
        for (int len = 1; len <= top; ++len) {
            const uint32_t d = (rev >> (root - len)) - first[len];
            if (d < cnt[len]) {
                ent = make_entry((uint32_t)len, sorted[ZR_IDX(offs[len] + d, which == kCodeLit ? 288 : 32)]);
                break;
            }
        }
        table[e] = (uint16_t)ent;
    }
    if (lane == 0) cnt[0] = (uint32_t)max;                           // slot 0 is unused by the comparison: keep max there
    wave_sync();
    return 0;
}

// A code longer than the root: the canonical comparison over the remaining lengths, wave-uniform.
template <typename LDS>
__device__ __forceinline__ uint32_t long_code(LDS &L, int which, int root, const uint16_t *sorted,
                                              unsigned long long hold) {
    const uint32_t rev15 = __builtin_bitreverse32((uint32_t)hold & 0x7fffu) >> 17;
    const int max = (int)uni(L.cnt(which)[0]);
    for (int len = root + 1; len <= max; ++len) {
        const uint32_t d = (rev15 >> (15 - len)) - uni(L.first(which)[len]);
        if (d < uni(L.cnt(which)[len])) return make_entry((uint32_t)len, uni(sorted[ZR_IDX(uni(L.offs(which)[len]) + d, which == kCodeLit ? 288 : 32)]));
    }
    return kBadMark;
}

// The decode loop's fast paths, hand-written (inffast_tpl.h:151-226 is the loop this stands for).  The parse is
// wave-uniform and serial: it runs on the CU's one scalar unit, a wave gets an instruction issued every four or five
// cycles, and a CU holds at most 16 streams (LDS) -- so what counts is the NUMBER of instructions and taken branches per
// symbol.  The compiler's version of this loop spent 27 scalar instructions on a literal and about 80 on a match, most of
// them carrying loop-exit conditions around as lane masks; here a root-table literal costs 7 and a short copy about 40.
// The loop handles, per symbol:
//   * a refill of the bit buffer from the 64 fetched words (it leaves when the 64 are used up);
//   * a literal whose code fits the root table: into the waiting run (`lit`, literal j in lane j);
//   * a length symbol whose code fits the root table + a distance whose code fits its root table, when the copy is one of
//     the common shapes and nothing else is due (op + len <= oplim: room in `out`, no flush; 64 per pass): source in the
//     ring without overlap (dist <= kNear, dist >= len), source already flushed (dist > kNear: read back from HBM), or
//     -- parts only -- source in front of the part (ZR_INFLATE_BEFORE_PART).
// Anything else leaves the loop with `stage` = what has been consumed of the next symbol: 0 nothing, 1 the length (`len`
// valid), 2 length and distance (`len`, `dist` valid); the C++ code below finishes that symbol in full generality and
// comes back.  Positions: `opb` is the position of the first waiting literal, op = opb + npend.
// (A variant that looked both tables up at 64 bit offsets at once -- lane j with the buffer shifted by j, the next
// symbol's entry then a v_readlane away -- removed two of three table accesses and was no faster for one wave and 12 %
// slower for 16 per CU: the accesses are not what a symbol waits for.)
// Hazards (the assembler inserts no wait states into inline asm): on gfx950 a VALU write of an SGPR/VCC needs two
// instructions before a VALU reads it -- the v_cmp / masked-operation pairs below are spaced accordingly; LDS operations
// of one wave execute in order, so a ds_read after a ds_write of the same bytes needs no wait.
#define ZR_INFLATE_FAST_LOOP(RD, WR, GL, BEFORE)                                                                       \
    "s_mov_b32 s43, 0\n\t"                                                                                             \
    /* Inside the loop the two counters the literal path tests live BIASED, so that the instruction that updates them    \
       sets SCC and no compare is needed (the scalar unit is what sixteen streams per CU share): cnt as cnt - 32 (a       \
       borrow = fewer than 32 bits left), npend as npend - 64 (a carry = the run is full), with opb + 64 and lane - 64   \
       beside it.  The C++ around the statement converts. */                                                          \
    "L_top_%=:\n\t"                                                                                                    \
    "s_cmp_lt_i32 %[cnt], 0\n\t"                                                                                       \
    "s_cbranch_scc0 L_look_%=\n\t"                                                                                     \
    "L_refill_%=:\n\t"                                                                                                 \
    "s_cmp_eq_u32 %[widx], 64\n\t"                                                                                     \
    "s_cbranch_scc1 L_exit0_%=\n\t"                                                                                    \
    "v_readlane_b32 s42, %[cur], %[widx]\n\t"                                                                          \
    "s_add_u32 %[widx], %[widx], 1\n\t"                                                                                \
    "s_add_u32 %[t0], %[cnt], 32\n\t"                                                                                  \
    "s_lshl_b64 s[44:45], s[42:43], %[t0]\n\t"                                                                         \
    "s_or_b64 s[40:41], s[40:41], s[44:45]\n\t"                                                                        \
    "s_add_u32 %[cnt], %[cnt], 32\n\t"                                                                                 \
    "L_look_%=:\n\t"                                                                                                   \
    "v_bfe_u32 %[va], s40, 0, %[litroot]\n\t"                                                                          \
    "v_lshl_add_u32 %[va], %[va], 1, %[litb]\n\t"                                                                      \
    "ds_read_u16 %[vb], %[va]\n\t"                                                                                     \
    "v_cmp_eq_u32 vcc, %[npend], %[laneb]\n\t"                                                                         \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
    "v_readfirstlane_b32 %[e], %[vb]\n\t"                                                                              \
    "v_lshrrev_b32 %[vb], 4, %[vb]\n\t"                                                                                \
    "s_cmpk_gt_u32 %[e], 0xfff\n\t"                                                                                    \
    "s_cbranch_scc1 L_notlit_%=\n\t"                                                                                   \
    "s_and_b32 %[t0], %[e], 15\n\t"                                                                                    \
    "v_cndmask_b32 %[lit], %[lit], %[vb], vcc\n\t"                                                                     \
    "s_lshr_b64 s[40:41], s[40:41], %[t0]\n\t"                                                                         \
    "s_add_u32 %[npend], %[npend], 1\n\t"                                                                              \
    "s_cbranch_scc1 L_full_%=\n\t"                                                                                     \
    "s_sub_u32 %[cnt], %[cnt], %[t0]\n\t"                                                                              \
    "s_cbranch_scc0 L_look_%=\n\t"                                                                                     \
    "s_branch L_refill_%=\n\t"                                                                                         \
    "L_full_%=:\n\t"                                                                                                   \
    "s_sub_u32 %[cnt], %[cnt], %[t0]\n\t"                                                                              \
    "s_branch L_exit0_%=\n\t"                                                                                          \
    "L_notlit_%=:\n\t"                                                                                                 \
    /* e - 0x1010 = code length | (symbol - 257) << 4; symbols 257..285 only (256, 286.., the long and bad marks leave) */ \
    "s_sub_u32 %[t1], %[e], 0x1010\n\t"                                                                                \
    "s_cmpk_gt_u32 %[t1], 0x1cf\n\t"                                                                                   \
    "s_cbranch_scc1 L_exit0_%=\n\t"                                                                                    \
    "s_and_b32 %[t0], %[e], 15\n\t"                                                                                    \
    "s_lshr_b32 %[t1], %[t1], 4\n\t"                                                                                   \
    "s_lshr_b64 s[40:41], s[40:41], %[t0]\n\t"                                                                         \
    "s_sub_u32 %[cnt], %[cnt], %[t0]\n\t"                                                                              \
    "s_add_u32 %[len], %[t1], 3\n\t"                                                                                   \
    "s_cmp_lt_u32 %[t1], 8\n\t"                                                                                        \
    "s_cbranch_scc1 L_havelen_%=\n\t"                                                                                  \
    "s_movk_i32 %[len], 258\n\t"                                                                                       \
    "s_cmp_eq_u32 %[t1], 28\n\t"                                                                                       \
    "s_cbranch_scc1 L_havelen_%=\n\t"                                                                                  \
    "s_sub_u32 %[t0], %[t1], 4\n\t"                                                                                    \
    "s_lshr_b32 %[t0], %[t0], 2\n\t"                                                                                   \
    "s_and_b32 %[t1], %[t1], 3\n\t"                                                                                    \
    "s_or_b32 %[t1], %[t1], 4\n\t"                                                                                     \
    "s_lshl_b32 %[t1], %[t1], %[t0]\n\t"                                                                               \
    "s_bfm_b32 %[len], %[t0], 0\n\t"                                                                                   \
    "s_and_b32 %[len], %[len], s40\n\t"                                                                                \
    "s_add_u32 %[len], %[len], %[t1]\n\t"                                                                              \
    "s_add_u32 %[len], %[len], 3\n\t"                                                                                  \
    "s_lshr_b64 s[40:41], s[40:41], %[t0]\n\t"                                                                         \
    "s_sub_u32 %[cnt], %[cnt], %[t0]\n\t"                                                                              \
    "L_havelen_%=:\n\t"                                                                                                \
    "s_cmp_lt_i32 %[cnt], 0\n\t"                                                                                       \
    "s_cbranch_scc0 L_dist_%=\n\t"                                                                                     \
    "s_cmp_eq_u32 %[widx], 64\n\t"                                                                                     \
    "s_cbranch_scc1 L_exit1_%=\n\t"                                                                                    \
    "v_readlane_b32 s42, %[cur], %[widx]\n\t"                                                                          \
    "s_add_u32 %[widx], %[widx], 1\n\t"                                                                                \
    "s_add_u32 %[t0], %[cnt], 32\n\t"                                                                                  \
    "s_lshl_b64 s[44:45], s[42:43], %[t0]\n\t"                                                                         \
    "s_or_b64 s[40:41], s[40:41], s[44:45]\n\t"                                                                        \
    "s_add_u32 %[cnt], %[cnt], 32\n\t"                                                                                 \
    "L_dist_%=:\n\t"                                                                                                   \
    "v_bfe_u32 %[va], s40, 0, %[distroot]\n\t"                                                                   \
    "v_lshl_add_u32 %[va], %[va], 2, %[distb]\n\t"                                                                     \
    "ds_read_b32 %[vb], %[va]\n\t"                                                                                     \
    "s_add_u32 %[op], %[opb], %[npend]\n\t"                                                                            \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
    "v_readfirstlane_b32 %[e], %[vb]\n\t"                                                                              \
    "s_cmp_gt_u32 %[e], 0xffffffe0\n\t"                                                                                \
    "s_cbranch_scc1 L_exit1_%=\n\t"                                                                                    \
    "s_and_b32 %[t0], %[e], 15\n\t"                                                                                    \
    "s_bfe_u32 %[t1], %[e], 0x40004\n\t"                                                                               \
    "s_lshr_b64 s[40:41], s[40:41], %[t0]\n\t"                                                                         \
    "s_sub_u32 %[cnt], %[cnt], %[t0]\n\t"                                                                              \
    "s_lshr_b32 %[dist], %[e], 8\n\t"                                                                                  \
    "s_bfm_b32 %[t0], %[t1], 0\n\t"                                                                                    \
    "s_and_b32 %[t0], %[t0], s40\n\t"                                                                                  \
    "s_add_u32 %[dist], %[dist], %[t0]\n\t"                                                                            \
    "s_lshr_b64 s[40:41], s[40:41], %[t1]\n\t"                                                                         \
    "s_sub_u32 %[cnt], %[cnt], %[t1]\n\t"                                                                              \
    /* nothing else due (room in `out`, no flush)? */                                                                 \
    "s_add_u32 %[t1], %[op], %[len]\n\t"                                                                               \
    "s_cmp_gt_u32 %[t1], %[oplim]\n\t"                                                                                 \
    "s_cbranch_scc1 L_exit2_%=\n\t"                                                                                    \
    /* the waiting literals go to the ring: lane j < npend -> position opb + j */                                     \
    "s_cmp_eq_u32 %[npend], 0xffffffc0\n\t"                                                                             \
    "s_cbranch_scc1 L_copy_%=\n\t"                                                                                     \
    "v_cmp_gt_u32 vcc, %[npend], %[laneb]\n\t"                                                                          \
    "s_add_u32 %[e], %[opb], %[a0m]\n\t"                                                                               \
    "v_add_u32 %[va], %[e], %[lane]\n\t"                                                                               \
    "v_and_b32 %[va], %[mask], %[va]\n\t"                                                                              \
    "v_lshl_add_u32 %[va], %[va], %[sh], %[ringb]\n\t"                                                                 \
    "s_and_saveexec_b64 s[44:45], vcc\n\t"                                                                             \
    WR " %[va], %[lit]\n\t"                                                                                            \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                                     \
    "s_add_u32 %[opb], %[op], 64\n\t"                                                                                   \
    "s_mov_b32 %[npend], 0xffffffc0\n\t"                                                                               \
    "L_copy_%=:\n\t"                                                                                                   \
    /* destination: ring slot of op + lane, lanes below len */                                                        \
    "v_cmp_gt_u32 vcc, %[len], %[lane]\n\t"                                                                            \
    "s_add_u32 %[e], %[op], %[a0]\n\t"                                                                                 \
    "v_add_u32 %[va], %[e], %[lane]\n\t"                                                                               \
    "v_subrev_u32 %[vb], %[dist], %[va]\n\t"                                                                           \
    "v_and_b32 %[va], %[mask], %[va]\n\t"                                                                              \
    "v_lshl_add_u32 %[va], %[va], %[sh], %[ringb]\n\t"                                                                 \
    "s_cmp_gt_u32 %[dist], %[op]\n\t"                                                                                  \
    "s_cbranch_scc1 L_before_%=\n\t"                                                                                   \
    "s_cmp_gt_u32 %[dist], %[near]\n\t"                                                                                \
    "s_cbranch_scc1 L_far_%=\n\t"                                                                                      \
    "s_cmp_lt_u32 %[dist], %[len]\n\t"                                                                                 \
    "s_cbranch_scc1 L_overlap_%=\n\t"                                                                                  \
    /* source in the ring; no overlap, or a distance of 64 and more (every pass of 64 then reads what is already there: \
       LDS operations of one wave execute in order) */                                                                \
    "L_plain_%=:\n\t"                                                                                                  \
    "v_and_b32 %[vb], %[mask], %[vb]\n\t"                                                                              \
    "v_lshl_add_u32 %[vb], %[vb], %[sh], %[ringb]\n\t"                                                                 \
    "s_and_saveexec_b64 s[44:45], vcc\n\t"                                                                             \
    RD " %[vb], %[vb]\n\t"                                                                                             \
    "s_add_u32 %[opb], %[opb], %[len]\n\t"                                                                             \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
    WR " %[va], %[vb]\n\t"                                                                                             \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                                     \
    "s_cmp_gt_u32 %[len], 64\n\t"                                                                                      \
    "s_cbranch_scc1 L_nearmore_%=\n\t"                                                                                 \
    "s_branch L_top_%=\n\t"                                                                                            \
    /* 65 .. 258: the other chunks of 64 (no overlap: the chunks are independent of each other) */                    \
    "L_nearmore_%=:\n\t"                                                                                               \
    "s_mov_b32 %[t1], 64\n\t"                                                                                          \
    "L_nearloop_%=:\n\t"                                                                                               \
    "s_sub_u32 %[t0], %[len], %[t1]\n\t"                                                                               \
    "v_cmp_gt_u32 vcc, %[t0], %[lane]\n\t"                                                                             \
    "s_add_u32 %[e], %[op], %[a0]\n\t"                                                                                 \
    "s_add_u32 %[e], %[e], %[t1]\n\t"                                                                                  \
    "v_add_u32 %[va], %[e], %[lane]\n\t"                                                                               \
    "v_subrev_u32 %[vb], %[dist], %[va]\n\t"                                                                           \
    "v_and_b32 %[va], %[mask], %[va]\n\t"                                                                              \
    "v_and_b32 %[vb], %[mask], %[vb]\n\t"                                                                              \
    "v_lshl_add_u32 %[va], %[va], %[sh], %[ringb]\n\t"                                                                 \
    "v_lshl_add_u32 %[vb], %[vb], %[sh], %[ringb]\n\t"                                                                 \
    "s_and_saveexec_b64 s[44:45], vcc\n\t"                                                                             \
    RD " %[vb], %[vb]\n\t"                                                                                             \
    "s_add_u32 %[t1], %[t1], 64\n\t"                                                                                   \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
    WR " %[va], %[vb]\n\t"                                                                                             \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                                     \
    "s_cmp_lt_u32 %[t1], %[len]\n\t"                                                                                   \
    "s_cbranch_scc1 L_nearloop_%=\n\t"                                                                                 \
    "s_branch L_top_%=\n\t"                                                                                            \
    /* dist < len and dist < 64: the copy repeats the dist bytes in front of it (runs, short periods -- a block of zeros is \
       nothing else).  Lane j of a pass at offset o reads pattern byte (o + j) mod dist from [op - dist, op), which no pass \
       writes: r = j mod dist by six compare-free steps (r = min(r, r - dist * 2^k), unsigned), then r += 64 mod dist per \
       pass. */                                                                                                       \
    "L_overlap_%=:\n\t"                                                                                                \
    "s_cmp_gt_u32 %[dist], 63\n\t"                                                                                     \
    "s_cbranch_scc1 L_plain_%=\n\t"                                                                                    \
    "s_lshl_b32 %[t0], %[dist], 5\n\t"                                                                                 \
    "v_subrev_u32 %[vb], %[t0], %[lane]\n\t"                                                                           \
    "s_lshl_b32 %[t0], %[dist], 4\n\t"                                                                                 \
    "v_min_u32 %[vr], %[vb], %[lane]\n\t"                                                                              \
    "v_subrev_u32 %[vb], %[t0], %[vr]\n\t"                                                                             \
    "s_lshl_b32 %[t0], %[dist], 3\n\t"                                                                                 \
    "v_min_u32 %[vr], %[vb], %[vr]\n\t"                                                                                \
    "v_subrev_u32 %[vb], %[t0], %[vr]\n\t"                                                                             \
    "s_lshl_b32 %[t0], %[dist], 2\n\t"                                                                                 \
    "v_min_u32 %[vr], %[vb], %[vr]\n\t"                                                                                \
    "v_subrev_u32 %[vb], %[t0], %[vr]\n\t"                                                                             \
    "s_lshl_b32 %[t0], %[dist], 1\n\t"                                                                                 \
    "v_min_u32 %[vr], %[vb], %[vr]\n\t"                                                                                \
    "v_subrev_u32 %[vb], %[t0], %[vr]\n\t"                                                                             \
    "v_min_u32 %[vr], %[vb], %[vr]\n\t"                                                                                \
    "v_subrev_u32 %[vb], %[dist], %[vr]\n\t"                                                                           \
    "v_min_u32 %[vr], %[vb], %[vr]\n\t"                                                                                \
    "s_add_u32 %[opb], %[opb], %[len]\n\t"                                                                             \
    "s_nop 0\n\t"                                                                                                      \
    "v_readlane_b32 %[t2], %[vr], 63\n\t"                                                                              \
    "s_mov_b32 %[t1], 0\n\t"                                                                                           \
    "s_add_u32 %[t2], %[t2], 1\n\t"                                                                                    \
    "s_cmp_eq_u32 %[t2], %[dist]\n\t"                                                                                  \
    "s_cselect_b32 %[t2], 0, %[t2]\n\t"                                                                                \
    "L_patloop_%=:\n\t"                                                                                                \
    "s_sub_u32 %[t0], %[len], %[t1]\n\t"                                                                               \
    "v_cmp_gt_u32 vcc, %[t0], %[lane]\n\t"                                                                             \
    "s_add_u32 %[e], %[op], %[a0]\n\t"                                                                                 \
    "s_sub_u32 %[t0], %[e], %[dist]\n\t"                                                                               \
    "s_add_u32 %[e], %[e], %[t1]\n\t"                                                                                  \
    "v_add_u32 %[vb], %[t0], %[vr]\n\t"                                                                                \
    "v_add_u32 %[va], %[e], %[lane]\n\t"                                                                               \
    "v_and_b32 %[vb], %[mask], %[vb]\n\t"                                                                              \
    "v_and_b32 %[va], %[mask], %[va]\n\t"                                                                              \
    "v_lshl_add_u32 %[vb], %[vb], %[sh], %[ringb]\n\t"                                                                 \
    "v_lshl_add_u32 %[va], %[va], %[sh], %[ringb]\n\t"                                                                 \
    "s_and_saveexec_b64 s[44:45], vcc\n\t"                                                                             \
    RD " %[vb], %[vb]\n\t"                                                                                             \
    "s_add_u32 %[t1], %[t1], 64\n\t"                                                                                   \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                         \
    WR " %[va], %[vb]\n\t"                                                                                             \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                                     \
    "v_add_u32 %[vr], %[t2], %[vr]\n\t"                                                                                \
    "v_subrev_u32 %[vb], %[dist], %[vr]\n\t"                                                                           \
    "s_cmp_lt_u32 %[t1], %[len]\n\t"                                                                                   \
    "v_min_u32 %[vr], %[vb], %[vr]\n\t"                                                                                \
    "s_cbranch_scc1 L_patloop_%=\n\t"                                                                                  \
    "s_branch L_top_%=\n\t"                                                                                            \
    /* source behind the ring's reach: it has been flushed (op - flushed < kFlushAt < kNear - 64), so it is in HBM,   \
       and the flush had its stores acknowledged */                                                                   \
    "L_far_%=:\n\t"                                                                                                    \
    /* up to five passes of 64: ALL their loads first (one address register, instruction offsets 64, 128, ... elements), \
       one wait, then the writes -- a pass at a time this was a memory round trip per 64 symbols, and a block of long     \
       copies at a distance just beyond the ring was the slowest part of a foreign stream */                           \
    "s_sub_u32 %[t0], %[op], %[dist]\n\t"                                                                              \
    "v_add_u32 %[vb], %[t0], %[lane]\n\t"                                                                              \
    "v_lshlrev_b32 %[vb], %[sh], %[vb]\n\t"                                                                            \
    "s_add_u32 %[opb], %[opb], %[len]\n\t"                                                                             \
    "s_and_saveexec_b64 s[44:45], vcc\n\t"                                                                             \
    GL " %[vd0], %[vb], %[outp]\n\t"                                                                                   \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                                     \
    "s_cmp_gt_u32 %[len], 64\n\t"                                                                                      \
    "s_cbranch_scc0 L_farw_%=\n\t"                                                                                     \
    "s_sub_u32 %[t0], %[len], 64\n\t"                                                                                  \
    "v_cmp_gt_u32 vcc, %[t0], %[lane]\n\t"                                                                             \
    "s_and_saveexec_b64 s[44:45], vcc\n\t"                                                                             \
    GL " %[vd1], %[vb], %[outp] offset:%[o1]\n\t"                                                                      \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                                     \
    "s_cmp_gt_u32 %[len], 128\n\t"                                                                                     \
    "s_cbranch_scc0 L_farw_%=\n\t"                                                                                     \
    "s_sub_u32 %[t0], %[len], 128\n\t"                                                                                 \
    "v_cmp_gt_u32 vcc, %[t0], %[lane]\n\t"                                                                             \
    "s_and_saveexec_b64 s[44:45], vcc\n\t"                                                                             \
    GL " %[vd2], %[vb], %[outp] offset:%[o2]\n\t"                                                                      \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                                     \
    "s_cmp_gt_u32 %[len], 192\n\t"                                                                                     \
    "s_cbranch_scc0 L_farw_%=\n\t"                                                                                     \
    "s_sub_u32 %[t0], %[len], 192\n\t"                                                                                 \
    "v_cmp_gt_u32 vcc, %[t0], %[lane]\n\t"                                                                             \
    "s_and_saveexec_b64 s[44:45], vcc\n\t"                                                                             \
    GL " %[vd3], %[vb], %[outp] offset:%[o3]\n\t"                                                                      \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                                     \
    "s_cmp_gt_u32 %[len], 256\n\t"                                                                                     \
    "s_cbranch_scc0 L_farw_%=\n\t"                                                                                     \
    "s_sub_u32 %[t0], %[len], 256\n\t"                                                                                 \
    "v_cmp_gt_u32 vcc, %[t0], %[lane]\n\t"                                                                             \
    "s_and_saveexec_b64 s[44:45], vcc\n\t"                                                                             \
    GL " %[vd4], %[vb], %[outp] offset:%[o4]\n\t"                                                                      \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                                     \
    "L_farw_%=:\n\t"                                                                                                   \
    "v_cmp_gt_u32 vcc, %[len], %[lane]\n\t"                                                                            \
    "s_waitcnt vmcnt(0)\n\t"                                                                                           \
    "s_nop 0\n\t"                                                                                                      \
    "s_and_saveexec_b64 s[44:45], vcc\n\t"                                                                             \
    WR " %[va], %[vd0]\n\t"                                                                                            \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                                     \
    "s_cmp_gt_u32 %[len], 64\n\t"                                                                                      \
    "s_cbranch_scc0 L_top_%=\n\t"                                                                                      \
    "s_sub_u32 %[t0], %[len], 64\n\t"                                                                                  \
    "s_add_u32 %[e], %[e], 64\n\t"                                                                                     \
    "v_cmp_gt_u32 vcc, %[t0], %[lane]\n\t"                                                                             \
    "v_add_u32 %[va], %[e], %[lane]\n\t"                                                                               \
    "v_and_b32 %[va], %[mask], %[va]\n\t"                                                                              \
    "v_lshl_add_u32 %[va], %[va], %[sh], %[ringb]\n\t"                                                                 \
    "s_and_saveexec_b64 s[44:45], vcc\n\t"                                                                             \
    WR " %[va], %[vd1]\n\t"                                                                                            \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                                     \
    "s_cmp_gt_u32 %[len], 128\n\t"                                                                                     \
    "s_cbranch_scc0 L_top_%=\n\t"                                                                                      \
    "s_sub_u32 %[t0], %[len], 128\n\t"                                                                                 \
    "s_add_u32 %[e], %[e], 64\n\t"                                                                                     \
    "v_cmp_gt_u32 vcc, %[t0], %[lane]\n\t"                                                                             \
    "v_add_u32 %[va], %[e], %[lane]\n\t"                                                                               \
    "v_and_b32 %[va], %[mask], %[va]\n\t"                                                                              \
    "v_lshl_add_u32 %[va], %[va], %[sh], %[ringb]\n\t"                                                                 \
    "s_and_saveexec_b64 s[44:45], vcc\n\t"                                                                             \
    WR " %[va], %[vd2]\n\t"                                                                                            \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                                     \
    "s_cmp_gt_u32 %[len], 192\n\t"                                                                                     \
    "s_cbranch_scc0 L_top_%=\n\t"                                                                                      \
    "s_sub_u32 %[t0], %[len], 192\n\t"                                                                                 \
    "s_add_u32 %[e], %[e], 64\n\t"                                                                                     \
    "v_cmp_gt_u32 vcc, %[t0], %[lane]\n\t"                                                                             \
    "v_add_u32 %[va], %[e], %[lane]\n\t"                                                                               \
    "v_and_b32 %[va], %[mask], %[va]\n\t"                                                                              \
    "v_lshl_add_u32 %[va], %[va], %[sh], %[ringb]\n\t"                                                                 \
    "s_and_saveexec_b64 s[44:45], vcc\n\t"                                                                             \
    WR " %[va], %[vd3]\n\t"                                                                                            \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                                     \
    "s_cmp_gt_u32 %[len], 256\n\t"                                                                                     \
    "s_cbranch_scc0 L_top_%=\n\t"                                                                                      \
    "s_sub_u32 %[t0], %[len], 256\n\t"                                                                                 \
    "s_add_u32 %[e], %[e], 64\n\t"                                                                                     \
    "v_cmp_gt_u32 vcc, %[t0], %[lane]\n\t"                                                                             \
    "v_add_u32 %[va], %[e], %[lane]\n\t"                                                                               \
    "v_and_b32 %[va], %[mask], %[va]\n\t"                                                                              \
    "v_lshl_add_u32 %[va], %[va], %[sh], %[ringb]\n\t"                                                                 \
    "s_and_saveexec_b64 s[44:45], vcc\n\t"                                                                             \
    WR " %[va], %[vd4]\n\t"                                                                                            \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                                     \
    "s_branch L_top_%=\n\t"                                                                                            \
    "L_before_%=:\n\t"                                                                                                 \
    BEFORE                                                                                                             \
    "L_exit2_%=:\n\t"                                                                                                  \
    "s_mov_b32 %[stage], 2\n\t"                                                                                        \
    "s_branch L_end_%=\n\t"                                                                                            \
    "L_exit1_%=:\n\t"                                                                                                  \
    "s_mov_b32 %[stage], 1\n\t"                                                                                        \
    "s_branch L_end_%=\n\t"                                                                                            \
    "L_exit0_%=:\n\t"                                                                                                  \
    "s_mov_b32 %[stage], 0\n\t"                                                                                        \
    "L_end_%=:\n\t"

// a source in front of `out`.  Whole streams: the dictionary -- the general code's business.  Parts: the bytes in front of
// a part are not known yet, the copy writes their NAMES (symbol 256 + 32768 + position, position < 0); done here when the
// whole copy lies in front (a copy that crosses into the part is left to the general code), after the reference's
// "too far back" test and with the furthest reach noted.
#define ZR_INFLATE_BEFORE_STREAM "s_branch L_exit2_%=\n\t"
#define ZR_INFLATE_BEFORE_PART                                                                                         \
    "s_add_u32 %[t1], %[op], %[dictlen]\n\t"                                                                           \
    "s_cmp_gt_u32 %[dist], %[t1]\n\t"                                                                                  \
    "s_cbranch_scc1 L_exit2_%=\n\t"                                                                                    \
    "s_sub_u32 %[t0], %[dist], %[op]\n\t"                                                                              \
    "s_max_u32 %[reach], %[reach], %[t0]\n\t"                                                                          \
    "s_cmp_gt_u32 %[len], %[t0]\n\t"                                                                                   \
    "s_cbranch_scc1 L_exit2_%=\n\t"                                                                                    \
    "s_sub_u32 %[t1], 33024, %[t0]\n\t"                                                                                \
    "v_add_u32 %[vb], %[t1], %[lane]\n\t"                                                                              \
    "s_and_saveexec_b64 s[44:45], vcc\n\t"                                                                             \
    "ds_write_b16 %[va], %[vb]\n\t"                                                                                    \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                                     \
    "s_add_u32 %[opb], %[opb], %[len]\n\t"                                                                             \
    "s_cmp_gt_u32 %[len], 64\n\t"                                                                                      \
    "s_cbranch_scc1 L_beforemore_%=\n\t"                                                                               \
    "s_branch L_top_%=\n\t"                                                                                            \
    "L_beforemore_%=:\n\t"                                                                                             \
    "s_mov_b32 %[t0], 64\n\t"                                                                                          \
    "L_beforeloop_%=:\n\t"                                                                                             \
    "s_sub_u32 %[e], %[len], %[t0]\n\t"                                                                                \
    "v_cmp_gt_u32 vcc, %[e], %[lane]\n\t"                                                                              \
    "s_add_u32 %[e], %[op], %[a0]\n\t"                                                                                 \
    "s_add_u32 %[e], %[e], %[t0]\n\t"                                                                                  \
    "v_add_u32 %[va], %[e], %[lane]\n\t"                                                                               \
    "s_add_u32 %[e], %[t1], %[t0]\n\t"                                                                                 \
    "v_and_b32 %[va], %[mask], %[va]\n\t"                                                                              \
    "v_add_u32 %[vb], %[e], %[lane]\n\t"                                                                               \
    "v_lshl_add_u32 %[va], %[va], %[sh], %[ringb]\n\t"                                                                 \
    "s_and_saveexec_b64 s[44:45], vcc\n\t"                                                                             \
    "ds_write_b16 %[va], %[vb]\n\t"                                                                                    \
    "s_mov_b64 exec, s[44:45]\n\t"                                                                                     \
    "s_add_u32 %[t0], %[t0], 64\n\t"                                                                                   \
    "s_cmp_lt_u32 %[t0], %[len]\n\t"                                                                                   \
    "s_cbranch_scc1 L_beforeloop_%=\n\t"                                                                               \
    "s_branch L_top_%=\n\t"

// PART = false: a job is a whole stream, the output is bytes (the many-stream entry points).
// PART = true (inflate_large.hip): a job is a PART of one large stream -- the decode starts at bit starts[job] of the
// stream, in the middle of it, and runs until a block ends exactly on a later entry of `starts` (or the stream ends).
// What lies in front of a part is not known while it is decoded, so the output is 16-bit SYMBOLS in the format of
// inflate_resolve.hip: a byte, or 256 + k = "byte k of the 32 KiB in front of this part"; copies move symbols, so
// unresolved references propagate by themselves, and the context chain of inflate_resolve.hip turns them into bytes.
// Results per part: 8 words {symbols produced, end bit (lo, hi), status, message, furthest reach in front of the part,
// index of the start it ended on, BFINAL seen}.
template <int RING, bool PART, bool COMPACT = false>
__global__ __launch_bounds__(64)
void inflate_streams_kernel(const InflateJobDev *__restrict__ jobs, uint32_t njobs, uint32_t *__restrict__ results,
                            const unsigned long long *__restrict__ starts) {
    typedef typename std::conditional<PART, uint16_t, uint8_t>::type T;
    constexpr uint32_t E = 16u / (uint32_t)sizeof(T);   // elements per 16-byte store
    constexpr uint32_t M = RING - 1;
    constexpr uint32_t kFlushAt = RING >= 4096 ? RING / 2 : RING / 4;   // unflushed bytes that trigger a flush
    constexpr uint32_t kPrioStep = 128u << 10;                          // part mode: symbols produced per step of wave priority
    constexpr uint32_t kStoredPiece = RING >= 4096 ? 1024u : RING / 4;  // a stored block enters the ring in pieces of this
    constexpr uint32_t kNear = RING - 258;               // a source this close is still in the ring while the match is written
    // bytes not yet flushed never exceed kFlushAt + 16 + max(258, kStoredPiece); a match of 258 more must not overwrite them
    static_assert(kFlushAt + 16 + kStoredPiece + 258 <= RING - 258, "ring too small for the flush / stored-chunk sizes");
    static_assert(kFlushAt + 258 < kNear, "a source beyond kNear must have left the ring (the fast loop reads it from HBM)");
    constexpr int kDistRoot = PART ? kDistRootPart : kDistRootStream, kLitRoot = PART ? kLitRootPart : kLitRootStream;
    __shared__ typename std::conditional<COMPACT, InflateLdsPart<RING, T, kDistRoot, kLitRoot>, InflateLdsStream<RING, T, kDistRoot, kLitRoot>>::type L;
    const int lane = threadIdx.x;
    const uint32_t job = blockIdx.x;
    if (job >= njobs) return;
    const InflateJobDev J = jobs[job];
    const ZR_GLOBAL uint8_t *const in = (const ZR_GLOBAL uint8_t *)J.in;
    ZR_GLOBAL T *const out = (ZR_GLOBAL T *)J.out;
    const uint32_t in_len = (uint32_t)J.in_len, out_cap = (uint32_t)J.out_cap, dict_len = J.dict_len;
    if (PART && out_cap == 0) return;                    // a part whose earlier result stands (inflate_large.hip reruns only some)
    const uint32_t a0 = (uint32_t)((uintptr_t)J.out & 15u) / (uint32_t)sizeof(T);   // ring slot of position p is (p + a0) & M

    // ---- compressed words: 64 per fetch, the next 64 prefetched -------------------------------------------------
    const uint32_t lead = (uint32_t)((uintptr_t)J.in & 3u);
    const ZR_GLOBAL uint32_t *const words = (const ZR_GLOBAL uint32_t *)(in - lead);
    const uint32_t total_words = (lead + in_len + 3u) >> 2;
    auto fetch = [&](uint32_t base) __attribute__((always_inline)) -> uint32_t {
        const uint32_t k = base + (uint32_t)lane;
        return k < total_words ? words[k] : 0u;
    };
    uint32_t cbase = 0, cur = 0, nxt = 0;
    uint32_t wnext = 0;                                  // index of the next word to enter the bit buffer
    unsigned long long hold = 0;
    uint32_t cnt = 0;
    auto append = [&]() __attribute__((always_inline)) {                                // cnt <= 32 on entry
        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(wnext - cbase));
        hold |= (unsigned long long)w << cnt;
        cnt += 32;
        ++wnext;
        if (wnext - cbase == 64) {
            cbase += 64;
            cur = nxt;
            nxt = fetch(cbase + 64);
        }
    };
    auto seek = [&](uint32_t byte_off) __attribute__((always_inline)) {                 // restart the bit buffer at a byte of the stream
        const uint32_t a = lead + byte_off;
        wnext = a >> 2;
        cbase = wnext;
        cur = fetch(cbase);
        nxt = fetch(cbase + 64);
        hold = 0;
        cnt = 0;
        append();
        hold >>= 8 * (a & 3u);
        cnt -= 8 * (a & 3u);
    };
    auto bit_pos = [&]() __attribute__((always_inline)) -> unsigned long long {         // stream bits consumed so far
        return 32ull * wnext - 8ull * lead - cnt;
    };
    uint32_t reach = 0, hit = 0xffffffffu;               // PART: furthest source in front of the part; the start it ended on
    if (PART) {
        const unsigned long long sb = starts[job];
        seek((uint32_t)(sb >> 3));
        hold >>= (uint32_t)(sb & 7ull);
        cnt -= (uint32_t)(sb & 7ull);
    } else {
        seek(0);
    }

    uint32_t op = 0, flushed = 0;
    uint32_t msg = kMsgNone;
#ifdef ZR_INFLATE_STATS
    unsigned long long zr_tasm = 0;
    const unsigned long long zr_tstart = __builtin_readcyclecounter();
#endif
    // A run of literals waits in ONE vector register, literal j in lane j (a compare and a select per literal,
    // no LDS access, no EXEC juggling); the run goes to the ring in one ds_write when a match, a flush or the 64th
    // literal comes.  `op` already counts the waiting literals.
    uint32_t litbuf = 0, npend = 0;
    auto dump = [&]() __attribute__((always_inline)) {
        if (npend) {
            if ((uint32_t)lane < npend) L.ring[(a0 + op - npend + (uint32_t)lane) & M] = (T)litbuf;
            npend = 0;
        }
    };
    // ring -> HBM: everything below `limit` (all of it when `final`), in aligned 16-byte stores
    auto flush = [&](uint32_t limit, bool final) __attribute__((always_inline)) {
        if (((a0 + flushed) & (E - 1u)) && flushed < limit) {
            uint32_t h = E - ((a0 + flushed) & (E - 1u));
            if (h > limit - flushed) h = limit - flushed;
            if ((uint32_t)lane < h) out[flushed + lane] = L.ring[(a0 + flushed + lane) & M];
            flushed += h;
        }
        const uint32_t chunks = (limit - flushed) / E;
        for (uint32_t c = (uint32_t)lane; c < chunks; c += 64) {
            const uint32_t p = flushed + E * c;
            *(ZR_GLOBAL u32x4_v *)(out + p) = *reinterpret_cast<const u32x4_v *>(&L.ring[(a0 + p) & M]);
        }
        flushed += E * chunks;
        if (final && flushed < limit) {
            if ((uint32_t)lane < limit - flushed) out[flushed + lane] = L.ring[(a0 + flushed + lane) & M];
            flushed = limit;
        }
        // a later far match may read these bytes back from HBM: have the stores acknowledged first
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0);
        if constexpr (PART) {
            // A call lasts as long as its longest part, and a part that has produced several times what the others do (a
            // block of long copies: 1.4 M symbols from 28 KB) is that part: it gets its instructions issued in front of the
            // waves it shares a SIMD with (tools/micro/copy_cost.py: a copy costs a lone wave 0.49 us, one among twelve 0.66).
            if (flushed >= kPrioStep * 3u) __builtin_amdgcn_s_setprio(3);
            else if (flushed >= kPrioStep * 2u) __builtin_amdgcn_s_setprio(2);
            else if (flushed >= kPrioStep) __builtin_amdgcn_s_setprio(1);
        }
    };
    // Everything that is not decoding happens when the literal run is written out, i.e. once per match or per 64
    // literals: the run goes to the ring (clipped at out_cap: a run may have decoded past it), a flush when one is due,
    // and the test that ends the decode of a truncated stream (the zero bits behind the input decode to something for ever).
    auto service = [&]() __attribute__((always_inline)) {
        if (op > out_cap) {
            const uint32_t fit = npend - (op - out_cap);
            if ((uint32_t)lane < fit) L.ring[(a0 + op - npend + (uint32_t)lane) & M] = (T)litbuf;
            op = out_cap;
            npend = 0;
            msg = kMsgOutFull;
            return;
        }
        dump();
        if (wnext > total_words + 2u) msg = kMsgStarved;     // whole words past the end of the input are in the bit buffer
        if (op - flushed >= kFlushAt) {
            wave_sync();
            flush(op, false);
        }
    };

    // the 16-bit distance entries the builder left in the first half of L.dist -> wide entries, through registers
    auto widen_distances = [&]() __attribute__((always_inline)) {
        uint32_t w[(1 << kDistRoot) / 64];
#pragma unroll
        for (int j = 0; j < (1 << kDistRoot) / 64; ++j) w[j] = wide_distance(reinterpret_cast<const uint16_t *>(L.dist)[lane + 64 * j]);
        wave_sync();
#pragma unroll
        for (int j = 0; j < (1 << kDistRoot) / 64; ++j) L.dist[lane + 64 * j] = w[j];
        wave_sync();
    };

    // (Every lambda above is always_inline: one that is called out of line gets its captures through the stack -- the
    // whole bit-parse state would live in scratch memory.)
    // The control flow below is kept to single-exit loops with an error word (no jumps out of nested loops): every branch
    // here is wave-uniform, and anything else makes the compiler carry loop-exit conditions as lane masks through the
    // hot loop (the first version of this kernel executed 73 scalar instructions per symbol, most of them that).
    // PART: does the block that just ended end exactly on a later start?  (binary search, wave-uniform)
    auto block_end_stop = [&]() __attribute__((always_inline)) -> bool {
        const unsigned long long b = bit_pos();
        uint32_t lo = job + 1u, hi = njobs;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (starts[mid] < b) lo = mid + 1u;
            else hi = mid;
        }
        if (lo < njobs && starts[lo] == b) {
            hit = lo;
            return true;
        }
        return false;
    };
    bool last = false;
    while (!last && msg == kMsgNone) {
        if (cnt < 32) append();
        last = hold & 1u;
        const uint32_t type = (uint32_t)(hold >> 1) & 3u;
        hold >>= 3;
        cnt -= 3;
        if (type == 3) { msg = kMsgBlockType; break; }
        if (type == 0) {
            // stored block (inflate.c:759-800): LEN / NLEN at the next byte boundary, then LEN raw bytes
            service();
            if (msg != kMsgNone) break;
            wave_sync();
            hold >>= cnt & 7u;
            cnt -= cnt & 7u;
            if (cnt < 32) append();
            const uint32_t len = (uint32_t)hold & 0xffffu, nlen = (uint32_t)(hold >> 16) & 0xffffu;
            hold >>= 32;
            cnt -= 32;
            if (bit_pos() > 8ull * in_len) { msg = kMsgStarved; break; }
            if (len != (nlen ^ 0xffffu)) { msg = kMsgStoredLen; break; }
            const uint32_t from = (uint32_t)(bit_pos() >> 3);          // byte aligned here
            const uint32_t avail = in_len - from;
            uint32_t n = len < avail ? len : avail;
            if (n > out_cap - op) n = out_cap - op;
            for (uint32_t done_n = 0; done_n < n;) {
                if (op - flushed >= kFlushAt) flush(op, false);
                const uint32_t piece = n - done_n < kStoredPiece ? n - done_n : kStoredPiece;
                const uint32_t lo = 16u * (uint32_t)lane;
                if (lo < piece) {
                    const ZR_GLOBAL uint8_t *src = in + from + done_n + lo;
                    if (lo + 16u <= piece) {
                        const u32x4_unaligned v = *(const ZR_GLOBAL u32x4_unaligned *)src;
                        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int k = 0; k < 16; ++k) L.ring[(a0 + op + lo + k) & M] = (T)(uint8_t)(w[k >> 2] >> (8 * (k & 3)));
                    } else {
                        for (uint32_t k = 0; lo + k < piece; ++k) L.ring[(a0 + op + lo + k) & M] = src[k];
                    }
                }
                op += piece;
                done_n += piece;
                wave_sync();
            }
            if (len > avail) { msg = kMsgStarved; break; }
            if (n < len) { msg = kMsgOutFull; break; }
            seek(from + len);
            if (PART && !last && block_end_stop()) break;
            continue;
        }
        if (type == 1) {
            // fixed codes (RFC 1951 3.2.6, inflate.c:801-813): the same builder, from the fixed lengths
            for (int s = lane; s < 288; s += 64) L.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < 32) L.lens[288 + lane] = 5;
            wave_sync();
            build_code(L, kCodeLit, L.lens, 288, kLitRoot, L.lit, L.sorted_lit, lane);
            build_code(L, kCodeDist, L.lens + 288, 32, kDistRoot, reinterpret_cast<uint16_t *>(L.dist), L.sorted_dist, lane);
            widen_distances();
        } else {
            // dynamic block header (inflate.c:814-917)
            if (cnt < 32) append();
            const uint32_t nlen = ((uint32_t)hold & 31u) + 257u, ndist = ((uint32_t)(hold >> 5) & 31u) + 1u,
                           ncode = ((uint32_t)(hold >> 10) & 15u) + 4u;
            hold >>= 14;
            cnt -= 14;
            if (nlen > 286 || ndist > 30) { msg = kMsgTooMany; break; }
            if (lane < 19) L.cl_lens()[lane] = 0;
            wave_sync();
            for (uint32_t i = 0; i < ncode; ++i) {
                if (cnt < 32) append();
                if (lane == 0) L.cl_lens()[kClOrder[i]] = (uint8_t)(hold & 7u);
                hold >>= 3;
                cnt -= 3;
            }
            wave_sync();
            if (uni((uint32_t)build_code(L, kCodeCl, L.cl_lens(), 19, kClRoot, L.cl(), L.sorted_cl(), lane))) { msg = kMsgCodeLengthsSet; break; }
            uint32_t have = 0;
            while (have < nlen + ndist) {
                if (cnt < 32) append();
                const uint32_t e = uni(L.cl()[(uint32_t)hold & ((1u << kClRoot) - 1u)]);
                // an empty code-length code yields one-bit entries of value 0: each reads as length 0
                // (inftrees.c:114-122 + inflate.c:846-849)
                const uint32_t nb = e & 15u, sym = e >> 4;
                hold >>= nb;
                cnt -= nb;
                if (sym < 16) {
                    if (lane == 0) L.lens[ZR_IDX(have, 320)] = (uint8_t)sym;
                    ++have;
                    continue;
                }
                uint32_t rep, val = 0;
                if (sym == 16) {
                    rep = 3u + ((uint32_t)hold & 3u);                 // NEEDBITS(here.bits + 2) comes first (inflate.c:856-864):
                    hold >>= 2;                                       // at the end of a truncated stream the answer is
                    cnt -= 2;                                         // "input ended", not this error
                    if (have == 0) { msg = kMsgBitRepeat; break; }
                    wave_sync();
                    val = uni(L.lens[have - 1]);
                } else if (sym == 17) {
                    rep = 3u + ((uint32_t)hold & 7u);
                    hold >>= 3;
                    cnt -= 3;
                } else {
                    rep = 11u + ((uint32_t)hold & 127u);
                    hold >>= 7;
                    cnt -= 7;
                }
                if (have + rep > nlen + ndist) { msg = kMsgBitRepeat; break; }
                for (uint32_t k = (uint32_t)lane; k < rep; k += 64) L.lens[ZR_IDX(have + k, 320)] = (uint8_t)val;
                have += rep;
            }
            if (msg != kMsgNone) break;
            wave_sync();
            if (bit_pos() > 8ull * in_len) { msg = kMsgStarved; break; }
            if (uni(L.lens[256]) == 0) { msg = kMsgNoEob; break; }
            if (uni((uint32_t)build_code(L, kCodeLit, L.lens, (int)nlen, kLitRoot, L.lit, L.sorted_lit, lane))) { msg = kMsgLitLenSet; break; }
            if (uni((uint32_t)build_code(L, kCodeDist, L.lens + nlen, (int)ndist, kDistRoot, reinterpret_cast<uint16_t *>(L.dist),
                                         L.sorted_dist, lane))) {
                msg = kMsgDistSet;
                break;
            }
            widen_distances();
        }

        // ---- symbol loop: the decode AND store halves of inflate_fast (inffast_tpl.h:140-300) -----------------------
        // Each round: the hand-written fast loop (ZR_INFLATE_FAST_LOOP) runs until a symbol needs more than it does, then
        // that ONE symbol is finished here, from the stage the fast loop left it in.
        const uint32_t lds_lit = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)L.lit,
                       lds_dist = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)L.dist,
                       lds_ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)L.ring;
        for (;;) {
            if (npend == 64) {
                service();
                if (msg != kMsgNone) break;
            }
            uint32_t stage, len = 0, dist = 0;
            {
                uint32_t widx = wnext - cbase, opb = op - npend + 64u, t0, t1, t2, ee, opx, va, vb, vr, vd0, vd1, vd2, vd3, vd4;
                // no flush is due and the match fits `out` while op + len <= oplim (what service() would test)
                const uint32_t fl = flushed + kFlushAt - 1u, oplim = out_cap < fl ? out_cap : fl;
                const unsigned long long out_addr = (unsigned long long)(uintptr_t)J.out;
                const uint32_t laneb = (uint32_t)lane - 64u;
                cnt -= 32u;                                       // the loop's biased forms (ZR_INFLATE_FAST_LOOP)
                npend -= 64u;
#ifdef ZR_INFLATE_STATS
                const unsigned long long zr_t0 = __builtin_readcyclecounter();
#endif
                if constexpr (sizeof(T) == 1) {
                    asm volatile(ZR_INFLATE_FAST_LOOP("ds_read_u8", "ds_write_b8", "global_load_ubyte", ZR_INFLATE_BEFORE_STREAM)
                                 : "+{s[40:41]}"(hold), [cnt] "+s"(cnt), [widx] "+s"(widx), [npend] "+s"(npend), [opb] "+s"(opb),
                                   [lit] "+v"(litbuf), [reach] "+s"(reach), [stage] "=&s"(stage), [len] "=&s"(len), [dist] "=&s"(dist), [e] "=&s"(ee),
                                   [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [op] "=&s"(opx), [va] "=&v"(va), [vb] "=&v"(vb), [vr] "=&v"(vr), [vd0] "=&v"(vd0), [vd1] "=&v"(vd1),
                                   [vd2] "=&v"(vd2), [vd3] "=&v"(vd3), [vd4] "=&v"(vd4)
                                 : [lane] "v"(lane), [laneb] "v"(laneb), [cur] "v"(cur), [a0] "s"(a0), [a0m] "s"(a0 - 64u), [oplim] "s"(oplim), [litb] "s"(lds_lit),
                                   [distb] "s"(lds_dist), [ringb] "s"(lds_ring), [outp] "s"(out_addr), [dictlen] "s"(dict_len), [near] "n"(kNear), [mask] "n"(M), [sh] "n"(0),
                                   [litroot] "n"(kLitRoot), [distroot] "n"(kDistRoot), [o1] "n"(64 * sizeof(T)), [o2] "n"(128 * sizeof(T)),
                                   [o3] "n"(192 * sizeof(T)), [o4] "n"(256 * sizeof(T))
                                 : "scc", "vcc", "memory", "s42", "s43", "s44", "s45");
                } else {
                    asm volatile(ZR_INFLATE_FAST_LOOP("ds_read_u16", "ds_write_b16", "global_load_ushort", ZR_INFLATE_BEFORE_PART)
                                 : "+{s[40:41]}"(hold), [cnt] "+s"(cnt), [widx] "+s"(widx), [npend] "+s"(npend), [opb] "+s"(opb),
                                   [lit] "+v"(litbuf), [reach] "+s"(reach), [stage] "=&s"(stage), [len] "=&s"(len), [dist] "=&s"(dist), [e] "=&s"(ee),
                                   [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [op] "=&s"(opx), [va] "=&v"(va), [vb] "=&v"(vb), [vr] "=&v"(vr), [vd0] "=&v"(vd0), [vd1] "=&v"(vd1),
                                   [vd2] "=&v"(vd2), [vd3] "=&v"(vd3), [vd4] "=&v"(vd4)
                                 : [lane] "v"(lane), [laneb] "v"(laneb), [cur] "v"(cur), [a0] "s"(a0), [a0m] "s"(a0 - 64u), [oplim] "s"(oplim), [litb] "s"(lds_lit),
                                   [distb] "s"(lds_dist), [ringb] "s"(lds_ring), [outp] "s"(out_addr), [dictlen] "s"(dict_len), [near] "n"(kNear), [mask] "n"(M), [sh] "n"(1),
                                   [litroot] "n"(kLitRoot), [distroot] "n"(kDistRoot), [o1] "n"(64 * sizeof(T)), [o2] "n"(128 * sizeof(T)),
                                   [o3] "n"(192 * sizeof(T)), [o4] "n"(256 * sizeof(T))
                                 : "scc", "vcc", "memory", "s42", "s43", "s44", "s45");
                }
#ifdef ZR_INFLATE_STATS
                zr_tasm += __builtin_readcyclecounter() - zr_t0;
#endif
                cnt += 32u;
                npend += 64u;
                opb -= 64u;
                wnext = cbase + widx;
                if (widx == 64) {                                                     // the 64 fetched words are used up
                    cbase += 64;
                    cur = nxt;
                    nxt = fetch(cbase + 64);
                }
                op = opb + npend;
                ZR_STAT(0);
                if (stage == 0) ZR_STAT(npend == 64 ? 1 : widx == 64 ? 2 : 3);          // 64 literals wait / the fetched words are used up / EOB, long or bad code
                if (stage == 1) ZR_STAT(widx == 64 ? 4 : 5);                          // words used up / long or bad distance code
                if (stage == 2) ZR_STAT(len > 64 ? 6 : op + len > oplim ? 7 : dist > op ? 8 : dist < len ? 9 : 10);
                if (stage == 0 && npend == 64) continue;
            }
            if (stage == 0) {
                if (cnt < 32) append();
                uint32_t e = uni(L.lit[(uint32_t)hold & ((1u << kLitRoot) - 1u)]);
                if (e == kLongMark) e = long_code(L, kCodeLit, kLitRoot, L.sorted_lit, hold);
                if (e < (256u << 4)) {                                                // a literal (here: one with a long code)
                    const uint32_t nb = e & 15u;
                    hold >>= nb;
                    cnt -= nb;
                    litbuf = (uint32_t)lane == npend ? (e >> 4) : litbuf;
                    ++npend;
                    ++op;
                    continue;
                }
                const uint32_t sym = e >> 4, nb = e & 15u;
                if (sym > 285u) {                                                     // 286, 287 (inftrees.c:44-45), no code
                    msg = bit_pos() + nb > 8ull * in_len ? kMsgStarved : kMsgLitLenCode;   // (its bits must all be input)
                    break;
                }
                hold >>= nb;
                cnt -= nb;
                if (sym == 256u) break;                                               // end of block
                // length: base and extra bits from the symbol (RFC 1951 3.2.5; inftrees.c:38-45 tabulate the same)
                const uint32_t k = sym - 257u;
                if (k < 8u) {
                    len = 3u + k;
                } else if (k == 28u) {
                    len = 258u;
                } else {
                    const uint32_t xb = (k - 4u) >> 2;
                    len = 3u + ((4u + (k & 3u)) << xb) + ((uint32_t)hold & ((1u << xb) - 1u));
                    hold >>= xb;
                    cnt -= xb;
                }
            }
            if (stage <= 1) {
                if (cnt < 32) append();
                uint32_t d = uni(L.dist[(uint32_t)hold & ((1u << kDistRoot) - 1u)]);
                if (d >= kBadWide) {
                    if (d == kLongWide) d = wide_distance(long_code(L, kCodeDist, kDistRoot, L.sorted_dist, hold));
                    if (d >= kBadWide) {
                        msg = bit_pos() + (d & 15u) > 8ull * in_len ? kMsgStarved : kMsgDistCode;
                        break;
                    }
                }
                const uint32_t dnb = d & 15u, dxb = (d >> 4) & 15u;
                hold >>= dnb;
                dist = (d >> 8) + ((uint32_t)hold & ((1u << dxb) - 1u));
                hold >>= dxb;
                cnt -= dnb + dxb;
            }
            if (dist > op + dict_len) { msg = kMsgTooFar; break; }                    // inffast_tpl.h:203-210
            if (PART && dist > op && dist - op > reach) reach = dist - op;
            service();
            if (msg != kMsgNone) break;
            if (len > out_cap - op) { msg = kMsgOutFull; break; }
            wave_sync();                                                              // earlier literals are in the ring
            const int src0 = (int)op - (int)dist;
            if (dist <= kNear && dist <= op && dist >= len && len <= 64u) {
                // the common shape, one pass, no loop: a short match that does not overlap itself, source in the ring
                if ((uint32_t)lane < len)
                    L.ring[(a0 + op + (uint32_t)lane) & M] = L.ring[(a0 + (uint32_t)src0 + (uint32_t)lane) & M];
            } else if (dist <= kNear && dist <= op) {
                // ring to ring.  Every source byte was produced before this match began (with dist < len the
                // sources are the `dist` bytes before op, repeated): the passes of the copy are independent.
                if (dist >= len) {
                    for (uint32_t i = (uint32_t)lane; i < len; i += 64)
                        L.ring[(a0 + op + i) & M] = L.ring[(a0 + (uint32_t)src0 + i) & M];
                } else if (dist == 1) {
                    const T v = L.ring[(a0 + (uint32_t)src0) & M];
                    for (uint32_t i = (uint32_t)lane; i < len; i += 64) L.ring[(a0 + op + i) & M] = v;
                } else {
                    const float inv = 1.0f / (float)dist;
                    for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                        uint32_t q = (uint32_t)((float)i * inv);
                        int r = (int)i - (int)(q * dist);
                        if (r < 0) r += (int)dist;
                        if (r >= (int)dist) r -= (int)dist;
                        L.ring[(a0 + op + i) & M] = L.ring[(a0 + (uint32_t)src0 + (uint32_t)r) & M];
                    }
                }
            } else {
                // a source beyond the ring's reach, or in the history in front of `out`: HBM, except for the bytes
                // that have not left the ring yet (a match that starts in the dictionary and runs into this stream)
                const float inv = 1.0f / (float)dist;
                for (uint32_t i = (uint32_t)lane; i < len; i += 64) {
                    uint32_t j = i;
                    if (dist < len) {
                        uint32_t q = (uint32_t)((float)i * inv);
                        int r = (int)i - (int)(q * dist);
                        if (r < 0) r += (int)dist;
                        if (r >= (int)dist) r -= (int)dist;
                        j = (uint32_t)r;
                    }
                    const int sp = src0 + (int)j;
                    T v;
                    if (sp >= (int)flushed) v = L.ring[(a0 + (uint32_t)sp) & M];
                    else if (PART && sp < 0) v = (T)(256 + 32768 + sp);      // a byte of the 32 KiB in front of this part
                    else v = out[sp];
                    L.ring[(a0 + op + i) & M] = v;
                }
            }
            op += len;
        }
        if (PART && !last && msg == kMsgNone && block_end_stop()) break;
    }
#ifdef ZR_INFLATE_STATS
    if (lane == 0) {
        const unsigned long long zr_tend = __builtin_readcyclecounter();
        atomicAdd(&g_inflate_stats[11], zr_tasm);
        atomicAdd(&g_inflate_stats[12], zr_tend - zr_tstart);
        if (job < 16384u) {
            g_inflate_span[2 * job] = zr_tstart;
            g_inflate_span[2 * job + 1] = zr_tend;
        }
    }
#endif
    // bits that do not exist were consumed: whatever happened after that point, the stream ended early
    if (msg != kMsgOutFull) service();                   // (an out-of-room exit has clipped the run already)
    // (a part keeps its out-of-room exit: inflate_large.hip gives it a larger slot and runs it again)
    if (!(PART && msg == kMsgOutFull) && bit_pos() > 8ull * in_len) msg = kMsgStarved;
    wave_sync();
    flush(op, true);
    if (PART) {
        if (lane == 0) {
            const unsigned long long b = bit_pos();
            results[8 * job + 0] = op;
            results[8 * job + 1] = (uint32_t)b;
            results[8 * job + 2] = (uint32_t)(b >> 32);
            results[8 * job + 3] = msg == kMsgNone ? (last ? 1u : 0u) : (msg == kMsgStarved || msg == kMsgOutFull) ? (uint32_t)-5 : (uint32_t)-3;
            results[8 * job + 4] = msg;
            results[8 * job + 5] = reach;
            results[8 * job + 6] = hit;
            results[8 * job + 7] = last ? 1u : 0u;
        }
        return;
    }
    if (lane == 0) {
        const unsigned long long used = (bit_pos() + 7ull) >> 3;
        results[4 * job + 0] = op;
        results[4 * job + 1] = used > in_len ? in_len : (uint32_t)used;
        results[4 * job + 2] = msg == kMsgNone ? 1u : (msg == kMsgStarved || msg == kMsgOutFull) ? (uint32_t)-5 : (uint32_t)-3;
        results[4 * job + 3] = msg;
    }
}

int launch_inflate_streams_device(const InflateJobDev *d_jobs, size_t njobs, uint32_t *d_results, hipStream_t st) {
    if (!njobs) return ZNG_ROCM_OK;
    // ring size: 4 KiB -> 16 streams per CU; 8 KiB (11 per CU) measured 25 % slower on both corpora.  The other
    // instantiations exist in measurement builds only (-DZR_MEASURE_FORMS, ZNG_ROCM_INFLATE_RING); the product reads no
    // environment variable.
#ifdef ZR_MEASURE_FORMS
    static const int ring = [] {
        const char *r = getenv("ZNG_ROCM_INFLATE_RING");
        return r ? atoi(r) : 4096;
    }();
    if (ring == 8192) ZR_LAUNCH_TRACED((inflate_streams_kernel<8192, false>), dim3((unsigned)njobs), dim3(64), st, d_jobs, (uint32_t)njobs, d_results, (const unsigned long long *)nullptr);
    else if (ring == 16384) ZR_LAUNCH_TRACED((inflate_streams_kernel<16384, false>), dim3((unsigned)njobs), dim3(64), st, d_jobs, (uint32_t)njobs, d_results, (const unsigned long long *)nullptr);
    else if (ring == 32768) ZR_LAUNCH_TRACED((inflate_streams_kernel<32768, false>), dim3((unsigned)njobs), dim3(64), st, d_jobs, (uint32_t)njobs, d_results, (const unsigned long long *)nullptr);
    else
#endif
    ZR_LAUNCH_TRACED((inflate_streams_kernel<4096, false>), dim3((unsigned)njobs), dim3(64), st, d_jobs, (uint32_t)njobs, d_results, (const unsigned long long *)nullptr);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

// the parts of ONE large stream (inflate_large.hip): d_starts = sorted start bits, one per job; results: 8 words per part
// `many`: more parts of real work than a chip holds at 12 per CU -- then the packed LDS layout (InflateLdsPart: 13 per CU)
// decodes more symbols per second (the cfg3 stream of this library's level-6 class: 8.2 -> 7.4 ms).  With fewer parts the
// call lasts as long as its longest part, and a part is faster in the plain layout among 12 (a CPython stream: 12.4
// against 13.2 ms): two instantiations, chosen per call.
int launch_inflate_parts_device(const InflateJobDev *d_jobs, size_t njobs, uint32_t *d_results, const unsigned long long *d_starts,
                                bool many, hipStream_t st) {
    if (!njobs) return ZNG_ROCM_OK;
#ifdef ZR_MEASURE_FORMS
    static const int ring = [] {
        const char *r = getenv("ZNG_ROCM_PART_RING");
        return r ? atoi(r) : 4096;
    }();
    if (ring == 2048 && many) {
        ZR_LAUNCH_TRACED((inflate_streams_kernel<2048, true, true>), dim3((unsigned)njobs), dim3(64), st, d_jobs, (uint32_t)njobs, d_results, d_starts);
        ZR_HIP(hipGetLastError());
        return ZNG_ROCM_OK;
    }
#endif
    if (many)
        ZR_LAUNCH_TRACED((inflate_streams_kernel<4096, true, true>), dim3((unsigned)njobs), dim3(64), st, d_jobs, (uint32_t)njobs, d_results, d_starts);
    else
        ZR_LAUNCH_TRACED((inflate_streams_kernel<4096, true, false>), dim3((unsigned)njobs), dim3(64), st, d_jobs, (uint32_t)njobs, d_results, d_starts);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

}  // namespace zr

using namespace zr;

extern "C" {

#ifdef ZR_INFLATE_STATS
void zng_rocm_debug_inflate_spans(unsigned long long *out, unsigned n) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(zr::g_inflate_span), (size_t)(n < 16384u ? n : 16384u) * 16);
}
void zng_rocm_debug_inflate_stats(unsigned long long *out16, int reset) {
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(zr::g_inflate_stats), 16 * sizeof(unsigned long long));
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(zr::g_inflate_stats), z, sizeof z);
    }
}
#endif
#ifdef ZR_INFLATE_BOUNDS
unsigned int zng_rocm_debug_inflate_bounds(void) {
    unsigned int v = 0;
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(zr::g_inflate_bounds_violations), sizeof v);
    return v;
}
#endif

const char *zng_rocm_inflate_message(uint32_t id) {
    static const char *const text[kMsgCount] = {
        "", "invalid block type", "invalid stored block lengths", "too many length or distance symbols",
        "invalid code lengths set", "invalid bit length repeat", "invalid code -- missing end-of-block",
        "invalid literal/lengths set", "invalid distances set", "invalid literal/length code", "invalid distance code",
        "invalid distance too far back", "input ended before the final block", "output buffer too small",
        "incorrect header check", "unknown compression method", "invalid window size", "header crc mismatch",
        "need dictionary", "incorrect data check", "incorrect length check"};
    return id < kMsgCount ? text[id] : "";
}

int zng_rocm_inflate_streams_dev(const zng_rocm_inflate_dev_job *jobs, size_t njobs, uint32_t *d_results, void *stream) {
    if (!ctx()) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    if (!njobs) return ZNG_ROCM_OK;
    if (!jobs || !d_results || njobs > 0x7fffffffull) return ZNG_ROCM_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard dev;
    Workspace *ws = workspace_for(st);
    if (!ws) return ZNG_ROCM_ENOMEM;
    std::lock_guard<std::mutex> use(ws->mu);
    InflateJobDev *d_jobs = nullptr, *h_jobs = nullptr;
    if (int rc = scratch_reserve(ws, kScrInflateDevJobs, njobs * sizeof(InflateJobDev), false, (void **)&d_jobs)) return rc;
    if (int rc = host_tables_acquire(ws)) return rc;
    if (int rc = scratch_reserve(ws, kScrInflateDevJobsHost, njobs * sizeof(InflateJobDev), true, (void **)&h_jobs)) return rc;
    for (size_t i = 0; i < njobs; ++i) {
        const zng_rocm_inflate_dev_job &j = jobs[i];
        if ((j.in_len && !j.in) || (j.out_cap && !j.out) || j.in_len > 0x7fffffffull || j.out_cap > 0x7fffffffull ||
            j.dict_len > 32768u || (j.dict_len && !j.out) || j.flags) {
            set_error("job %zu: null buffer, a stream or output of 2 GiB and more, dict_len above 32768, or unknown flags", i);
            return ZNG_ROCM_EINVAL;
        }
        h_jobs[i] = InflateJobDev{(const uint8_t *)j.in, (uint8_t *)j.out, j.in_len, j.out_cap, j.dict_len, j.flags};
    }
    ZR_HIP(hipMemcpyAsync(d_jobs, h_jobs, njobs * sizeof(InflateJobDev), hipMemcpyHostToDevice, st));
    if (int rc = host_tables_release(ws, st)) return rc;
    if (int rc = launch_inflate_streams_device(d_jobs, njobs, d_results, st)) return rc;
    return ZNG_ROCM_OK;
}

}  // extern "C"
