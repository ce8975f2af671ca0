// inflate_large.hip -- ONE large raw deflate stream inflated on the device (BASELINE.json configs[2]).
//
// inflate_fast (inffast_tpl.h:53-318) is a serial bit parse: where a code starts depends on every code before it, and a
// block header carries no marker.  inflate_dev.hip therefore needs many streams to fill the chip (one wavefront each).
// This file cuts ONE stream into parts the same kernel can take -- the scheme of inflate_threads.cpp (pugz / rapidgzip),
// with the device in place of the host threads:
//   F1 find_headers_kernel      every BIT position of the stream is tested for the start of a dynamic block the decoder
//        would accept, cheap tests first: BTYPE / HLIT / HDIST in range (inflate.c:808-813), the code-length code complete
//        (inftrees.c:126-131; a Kraft sum over the 3-bit fields, three fields per table lookup).  About 1 position in
//        10^3 is left.  Also: the byte pattern 00 00 ff ff of a sync-flush marker (deflate.c:1064-1076) -- the block
//        behind it starts on the next byte, whatever its type -- and a stored block's header on a byte boundary
//        (header byte, LEN, NLEN = ~LEN; inflate.c:759-775): incompressible data is a chain of those.
//   F2 validate_headers_kernel  one lane per survivor decodes the whole header (inflate.c:814-917) and applies the
//        validity rules of inflate_table to the two code sets (inftrees.c:104-137): about 1 random position in 10^6
//        survives both kernels, real block starts all do.
//   host                        sorts the few thousand candidates (at least 2 KiB of compressed bytes apart); bit 0 is
//        always a start.
//   P  inflate_streams_kernel<PART>  one wavefront per part, from its start until a block ends EXACTLY on a later start
//        (or BFINAL), 16-bit symbols into the part's own slot (a symbol >= 256 names a byte of the 32 KiB in front of
//        the part: nobody knows yet what is there).
//   host                        chains the parts from bit 0: part 0 is genuine, the part that starts where it ended is
//        therefore genuine too, ...; a candidate that was noise is simply never reached (and the part in front of it has
//        run across it).  Then every part's place in the output is known.
//   C  compact_parts_kernel     slots -> one symbol array, consecutive parts grouped into segments of >= 40 KiB (the
//        context chain of inflate_resolve.hip looks one segment back) with their references re-based to the segment;
//        K2 / K3 of inflate_resolve.hip resolve and translate.
// Anything irregular on the chain -- a data error, a truncated stream, a distance too far back, no candidates (a stream of fixed-Huffman blocks only) -- sends the call to the sequential host decoder, which
// then reports exactly what the reference would (status, message, bytes produced, bytes consumed).
#include <algorithm>
#include <mutex>
#include <vector>

#include "context.h"
#include "deflate_dev.h"
#include "inflate_dev.h"

namespace zr {

constexpr uint32_t kSpacingBytes = 2u << 10;      // least compressed bytes between two starts -- applied only when there are
constexpr uint32_t kPartsUnthinned = 12288;       // more candidates than this.  (Thinning a short list loses real starts:
                                                  // about one random bit position in 10^6 passes F1 + F2, and a false start
                                                  // within the spacing in front of a real one would take its place; a false
                                                  // start that is kept costs one wasted part and nothing else.)
constexpr uint32_t kSegmentBytes = 40u << 10;     // parts are grouped into segments of at least this much OUTPUT for the
                                                  // context chain (which looks one segment back: >= 32 KiB each)
constexpr uint32_t kSlotRatio = 64;               // symbols of slot per compressed byte of the part ...
constexpr uint32_t kSlotSlack = 640u << 10;       // ... plus this (a 512 KiB run of one byte is ~600 bytes of deflate data)

// 64 bits of the stream starting at `bit` (bits beyond the end read 0)
__device__ __forceinline__ unsigned long long bits_at_dev(const uint8_t *src, unsigned long long src_len, unsigned long long bit) {
    const unsigned long long byte = bit >> 3;
    unsigned long long lo = 0, hi = 0;
    if (byte + 16 <= src_len) {
        const u32x4_unaligned v = load_u128(src + byte);
        lo = (unsigned long long)v.x | ((unsigned long long)v.y << 32);
        hi = (unsigned long long)v.z | ((unsigned long long)v.w << 32);
    } else {
        for (unsigned k = 0; k < 16 && byte + k < src_len; ++k) {
            const unsigned long long b = load_u8(src + byte + k);
            if (k < 8) lo |= b << (8 * k);
            else hi |= b << (8 * (k - 8));
        }
    }
    const unsigned s = (unsigned)(bit & 7ull);
    return s ? (lo >> s) | (hi << (64 - s)) : lo;
}

// F1.  One lane per byte, its eight bit positions in turn; a workgroup scans a contiguous piece of the stream and collects its
// survivors in LDS, so the global list takes one atomic per workgroup and flush (a single contended word manages ~90
// atomics per microsecond: one per survivor would cost more than the scan).  cand: bit position, bit 63 set for "block
// behind a marker".
// PATTERNS = true: the two byte patterns only (no bit position is tested): the first pass -- a stream whose encoder left a
// sync marker or a stored block every few tens of KiB (this library's level-6 class, pigz) offers enough starts that way.
constexpr uint32_t kFindLocal = 2048;
template <bool PATTERNS>
__global__ __launch_bounds__(256)
void find_headers_kernel(const uint8_t *__restrict__ src, unsigned long long src_len, unsigned long long *__restrict__ cand,
                         uint32_t *__restrict__ ncand, uint32_t cap) {
    __shared__ uint8_t kraft9[512];             // three 3-bit code lengths -> their share of the Kraft sum, in 1/128
    __shared__ unsigned long long local[kFindLocal];
    __shared__ uint32_t nlocal, gbase;
    for (int v = threadIdx.x; v < 512 && !PATTERNS; v += 256) {
        unsigned sum = 0;
        for (int f = 0; f < 3; ++f) {
            const unsigned l = (v >> (3 * f)) & 7;
            if (l) sum += 128u >> l;
        }
        kraft9[v] = (uint8_t)sum;
    }
    if (threadIdx.x == 0) nlocal = 0;
    __syncthreads();
    auto flush = [&]() {                         // all threads
        __syncthreads();
        const uint32_t n = nlocal < kFindLocal ? nlocal : kFindLocal;
        if (threadIdx.x == 0) gbase = n ? atomicAdd(ncand, n) : 0u;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += 256)
            if (gbase + i < cap) cand[gbase + i] = local[i];
        __syncthreads();
        if (threadIdx.x == 0) nlocal = 0;
        __syncthreads();
    };
    const unsigned long long per = (src_len + gridDim.x - 1) / gridDim.x;
    const unsigned long long lo_b = per * blockIdx.x, hi_b = lo_b + per < src_len ? lo_b + per : src_len;
    if constexpr (PATTERNS) {
        // the two byte patterns alone: a lane takes 16 byte positions from one 24-byte window (position by position, with a
        // 16-byte load each, the pass read every byte 16 times and ran at 0.44 TB/s: 0.22 ms for 96 MB)
        auto test = [&](unsigned long long w, unsigned long long b) {      // w: the 8 bytes at position b
            if (b >= hi_b || b + 8 >= src_len) return;
            if ((uint32_t)w == 0xffff0000u && b + 6 <= src_len) {
                const uint32_t i = atomicAdd(&nlocal, 1u);
                if (i < kFindLocal) local[i] = (8ull * (b + 4)) | (1ull << 63);
            }
            const uint32_t len16 = (uint32_t)(w >> 8) & 0xffffu, nlen16 = (uint32_t)(w >> 24) & 0xffffu;
            if (((uint32_t)w & 0xfeu) == 0u && len16 != 0u && (len16 ^ nlen16) == 0xffffu && b + 5 + len16 <= src_len) {
                const uint32_t i = atomicAdd(&nlocal, 1u);
                if (i < kFindLocal) local[i] = (8ull * b) | (3ull << 62);
            }
        };
        for (unsigned long long base = lo_b; base < hi_b; base += 4096) {
            const unsigned long long b0 = base + 16ull * threadIdx.x;
            if (b0 < hi_b) {
                unsigned long long lo = 0, hi = 0, nx = 0;
                if (b0 + 24 <= src_len) {
                    const u32x4_unaligned v = load_u128(src + b0);
                    lo = (unsigned long long)v.x | ((unsigned long long)v.y << 32);
                    hi = (unsigned long long)v.z | ((unsigned long long)v.w << 32);
                    const u32x4_unaligned n = load_u128(src + b0 + 8);                   // bytes 8 .. 23: the upper half is what is new
                    nx = (unsigned long long)n.z | ((unsigned long long)n.w << 32);
                } else {
                    for (unsigned k = 0; k < 24 && b0 + k < src_len; ++k) {
                        const unsigned long long x = load_u8(src + b0 + k);
                        if (k < 8) lo |= x << (8 * k);
                        else if (k < 16) hi |= x << (8 * (k - 8));
                        else nx |= x << (8 * (k - 16));
                    }
                }
                // a byte 0 or 1 in front, or 00 00 ff ff: positions whose first byte is neither 0x00 nor 0x01 cannot match
#pragma unroll
                for (unsigned k = 0; k < 16; ++k) {
                    const unsigned long long a = k < 8 ? lo : hi, c = k < 8 ? hi : nx;
                    const unsigned sh = 8 * (k & 7);
                    const unsigned long long w = sh ? (a >> sh) | (c << (64 - sh)) : a;
                    if (((uint32_t)w & 0xfeu) == 0u) test(w, b0 + k);
                }
            }
            __syncthreads();
            const uint32_t nl = nlocal;
            __syncthreads();
            if (nl > kFindLocal / 4) flush();
        }
        flush();
        return;
    }
    for (unsigned long long base = lo_b; base < hi_b; base += 256) {
        const unsigned long long b = base + threadIdx.x;
        if (b < hi_b && b + 8 < src_len) {
            unsigned long long lo = 0, hi = 0;
            if (b + 16 <= src_len) {
                const u32x4_unaligned v = load_u128(src + b);
                lo = (unsigned long long)v.x | ((unsigned long long)v.y << 32);
                hi = (unsigned long long)v.z | ((unsigned long long)v.w << 32);
            } else {
                for (unsigned k = 0; k < 16 && b + k < src_len; ++k) {
                    const unsigned long long x = load_u8(src + b + k);
                    if (k < 8) lo |= x << (8 * k);
                    else hi |= x << (8 * (k - 8));
                }
            }
            if ((uint32_t)lo == 0xffff0000u && b + 6 <= src_len) {       // 00 00 ff ff: the block behind the marker
                const uint32_t i = atomicAdd(&nlocal, 1u);
                if (i < kFindLocal) local[i] = (8ull * (b + 4)) | (1ull << 63);
            }
            // a stored block that starts on a byte boundary (as every stored block behind another one does: a run of
            // incompressible data is a chain of them, and without these starts it would be ONE part -- 8 MiB of the cfg3 mix
            // were, and that part was the whole kernel's tail): header byte 0 or 1, LEN, NLEN = ~LEN.  Noise passes this
            // once in 2^23 bytes; a false start costs one wasted part (nothing ever ends on it).
            {
                const uint32_t len16 = (uint32_t)(lo >> 8) & 0xffffu, nlen16 = (uint32_t)(lo >> 24) & 0xffffu;
                if (((uint32_t)lo & 0xfeu) == 0u && len16 != 0u && (len16 ^ nlen16) == 0xffffu && b + 5 + len16 <= src_len) {
                    const uint32_t i = atomicAdd(&nlocal, 1u);
                    if (i < kFindLocal) local[i] = (8ull * b) | (3ull << 62);          // bit 62: a stored block (light work)
                }
            }
            if constexpr (!PATTERNS) {
                // the cheap header tests for the eight bit positions of this byte at once, as bit masks over the window (bit k
                // = position k): BTYPE = 2 (bit k+1 clear, bit k+2 set), HLIT and HDIST not 30 or 31 (their upper four bits not
                // all set); the Kraft sum only for the positions left, one by one (a wavefront executes as many turns as
                // its busiest lane has positions left -- ~5 -- where the unrolled form executed the sum eight times)
                const uint32_t w = (uint32_t)lo;
                const uint32_t hl = (w >> 4) & (w >> 5) & (w >> 6) & (w >> 7), hd = (w >> 9) & (w >> 10) & (w >> 11) & (w >> 12);
                uint32_t left = (~w >> 1) & (w >> 2) & ~hl & ~hd & 0xffu;
                while (left) {
                    const unsigned k = (unsigned)__builtin_ctz(left);
                    left &= left - 1u;
                    const unsigned long long v = k ? (lo >> k) | (hi << (64 - k)) : lo;
                    const unsigned ncode = ((uint32_t)(v >> 13) & 15u) + 4u;
                    unsigned long long c = (lo >> (k + 17)) | (hi << (64 - (k + 17)));       // 57 bits: all 19 fields
                    c &= (1ull << (3 * ncode)) - 1ull;                                          // (19 fields are 57 bits: the table lookups below read 63)
                    unsigned sum = 0;
#pragma unroll
                    for (int g = 0; g < 7; ++g) sum += kraft9[(uint32_t)(c >> (9 * g)) & 511u];
                    if (sum != 128u) continue;
                    const uint32_t i = atomicAdd(&nlocal, 1u);
                    if (i < kFindLocal) local[i] = 8ull * b + k;
                }
            }
        }
        // 2048 positions per round leave ~2 survivors on real data; a round of adversarial data could fill the list
        __syncthreads();
        const uint32_t nl = nlocal;                     // the same value for every thread: nobody adds between the barriers
        __syncthreads();
        if (nl > kFindLocal / 4) flush();
    }
    flush();
}

// F2.  One lane per survivor of F1: the whole dynamic header (inflate.c:814-917) and inflate_table's validity rules for the
// literal/length and the distance code (inftrees.c:104-137).  Those rules are statements about Kraft sums, so the code
// lengths are summed as they are decoded and never stored: a set is over-subscribed when its sum passes 1 (the lane
// leaves at once -- noise does so within a few symbols, which is what keeps this kernel short), complete when it ends at
// exactly 1, and an incomplete set is allowed only as ONE code of length 1 (inftrees.c:133-134: left > 0 needs max == 1).
// Per lane in LDS: the code-length code as a 128-entry direct table (length << 5 | symbol).  Survivors are appended to
// good[] (bit positions; the marker bit removed).
constexpr int kValLanes = 64;
__device__ void validate_one(const uint8_t *__restrict__ src, unsigned long long src_len, const unsigned long long c0, uint8_t *T,
                             unsigned long long *__restrict__ good, uint32_t *__restrict__ ngood, uint32_t cap);
__global__ __launch_bounds__(kValLanes)
void validate_headers_kernel(const uint8_t *__restrict__ src, unsigned long long src_len,
                             const unsigned long long *__restrict__ cand, const uint32_t *__restrict__ ncand, uint32_t cand_cap,
                             unsigned long long *__restrict__ good, uint32_t *__restrict__ ngood, uint32_t cap) {
    __shared__ uint8_t tab[kValLanes][128];
    // the number of candidates is read here, not on the host: F1 -> F2 needs no round trip (the grid is fixed, each
    // workgroup takes every gridDim.x-th group of 64 candidates; a lane's candidates are independent of each other)
    const uint32_t n = *ncand < cand_cap ? *ncand : cand_cap;
    for (uint32_t i = blockIdx.x * kValLanes + threadIdx.x; i < n; i += gridDim.x * kValLanes)
        validate_one(src, src_len, cand[i], tab[threadIdx.x], good, ngood, cap);
}

__device__ void validate_one(const uint8_t *__restrict__ src, unsigned long long src_len, const unsigned long long c0, uint8_t *T,
                             unsigned long long *__restrict__ good, uint32_t *__restrict__ ngood, uint32_t cap) {
    auto accept = [&]() {
        const uint32_t at = atomicAdd(ngood, 1u);
        if (at < cap) good[at] = c0 & ~(1ull << 63);      // (bit 62, "a stored block", stays)
    };
    if (c0 >> 63) {                                       // behind a marker: taken as it is
        accept();
        return;
    }
    const unsigned long long bit = c0;
    unsigned long long w = bits_at_dev(src, src_len, bit + 3);
    const unsigned nlen = (unsigned)(w & 31) + 257, ndist = (unsigned)((w >> 5) & 31) + 1, ncode = (unsigned)((w >> 10) & 15) + 4;
    if (nlen > 286 || ndist > 30) return;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19];
#pragma unroll
    for (int k = 0; k < 19; ++k) cl[k] = 0;
    unsigned long long pos = bit + 17;
    {
        const unsigned long long f = bits_at_dev(src, src_len, pos);
        for (unsigned k = 0; k < ncode; ++k) {
            const unsigned l = (unsigned)(f >> (3 * k)) & 7u;
            // cl[order[k]] = l without a dynamically indexed register array
#pragma unroll
            for (int s = 0; s < 19; ++s)
                if (order[k] == s) cl[s] = (uint8_t)l;
        }
        pos += 3ull * ncode;
    }
    if ((pos >> 3) + 8 > src_len) return;
    // canonical code of the code-length code -> direct table (complete: F1 checked the Kraft sum)
    unsigned count[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 19; ++s) {
#pragma unroll
        for (int l = 1; l < 8; ++l)
            if (cl[s] == l) ++count[l];
    }
    unsigned next[8];
    {
        unsigned code = 0;
#pragma unroll
        for (int l = 1; l < 8; ++l) {
            code = (code + count[l - 1]) << 1;
            next[l] = code;
        }
    }
#pragma unroll
    for (int s = 0; s < 19; ++s) {
        const unsigned l = cl[s];
        if (!l) continue;
        unsigned code = 0;
#pragma unroll
        for (int q = 1; q < 8; ++q)
            if (l == (unsigned)q) code = next[q]++;
        const unsigned rev = __builtin_bitreverse32(code) >> (32 - l);
        for (unsigned e = rev; e < 128; e += 1u << l) T[e] = (uint8_t)((l << 5) | (unsigned)s);
    }
    // Kraft sums in 1/32768; codes used and the longest code per set; the length the end-of-block symbol got
    unsigned have = 0, prev = 0, eob = 0;
    unsigned kraft[2] = {0, 0}, used[2] = {0, 0}, longest[2] = {0, 0};
    const unsigned total = nlen + ndist;
    // `rep` symbols from `have` on get code length l.  No loop over the run: the lanes of a wave execute the longest run
    // any of them has (138 zeros is common), and that loop was most of this kernel's time.
    auto put = [&](unsigned l, unsigned rep) -> bool {
        if (have <= 256u && 256u < have + rep) eob = l;
        if (l) {
            const unsigned n0 = have < nlen ? (rep < nlen - have ? rep : nlen - have) : 0u, n1 = rep - n0;
            kraft[0] += n0 * (32768u >> l);
            kraft[1] += n1 * (32768u >> l);
            used[0] += n0;
            used[1] += n1;
            if (n0 && l > longest[0]) longest[0] = l;
            if (n1 && l > longest[1]) longest[1] = l;
        }
        have += rep;
        return kraft[0] <= 32768u && kraft[1] <= 32768u;
    };
    unsigned long long hold = bits_at_dev(src, src_len, pos);
    unsigned spent = 0;                                   // bits of `hold` used up
    while (have < total) {
        if (spent > 48u) {                                // a symbol takes at most 7 + 7 bits
            pos += spent;
            if ((pos >> 3) + 8 > src_len) return;
            hold = bits_at_dev(src, src_len, pos);
            spent = 0;
        }
        const unsigned long long v = hold >> spent;
        const unsigned e = T[(unsigned)v & 127u];
        const unsigned nb = e >> 5, sym = e & 31u;
        spent += nb;
        if (sym < 16) {
            prev = sym;
            if (!put(sym, 1)) return;
            continue;
        }
        unsigned rep, val = 0;
        const unsigned long long x = v >> nb;
        if (sym == 16) {
            if (have == 0) return;
            val = prev;
            rep = 3 + (unsigned)(x & 3);
            spent += 2;
        } else if (sym == 17) {
            rep = 3 + (unsigned)(x & 7);
            spent += 3;
        } else {
            rep = 11 + (unsigned)(x & 127);
            spent += 7;
        }
        if (have + rep > total) return;
        prev = val;
        if (!put(val, rep)) return;
    }
    if ((((pos + spent) >> 3) + 8) > src_len) return;
    if (eob == 0) return;                                 // no end-of-block code (inflate.c:897-901)
    // literal/length set: complete, or the single code of length 1; distance set: the same, or no code at all
    if (kraft[0] != 32768u && !(used[0] == 1u && longest[0] == 1u)) return;
    if (kraft[1] != 32768u && !(used[1] == 0u || (used[1] == 1u && longest[1] == 1u))) return;
    accept();
}

struct PartCopy {
    const uint16_t *src;      // the part's slot
    uint64_t        dst;      // its first symbol's index in the stream's symbol array
    uint64_t        gstart;   // first symbol of the SEGMENT (group of consecutive parts) it belongs to
    uint32_t        n;
    uint32_t        first;    // chain index of the segment's first part
};

// C.  grid.y = part on the chain: its symbols go to their place in the stream's symbol array.  A part's references count
// back from the PART's first byte; the context chain wants them counted from the SEGMENT's: a reference that lands inside
// an earlier part of the same segment is replaced by the symbol that part has there (itself possibly a reference: the
// loop follows it), one that lands in front of the segment is re-based.
__global__ __launch_bounds__(256)
void compact_parts_kernel(const PartCopy *__restrict__ parts, uint16_t *__restrict__ sym) {
    const uint32_t j0 = blockIdx.y;
    const PartCopy p = parts[j0];
    uint16_t *d = sym + p.dst;
    const bool rebase = p.dst != p.gstart;               // a segment's first part needs nothing
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    auto fix = [&](uint32_t v) -> uint32_t {             // a reference of this part, counted from the segment's first byte
        uint32_t j = j0;
        uint64_t base = p.dst;                           // first symbol of the part v counts back from
        for (;;) {
            const uint64_t k = 33024u - v;               // bytes in front of `base`, 1 .. 32768
            if (base - p.gstart < k) return 33024u - (uint32_t)(k - (base - p.gstart));      // in front of the segment
            const uint64_t g = base - k;                 // a symbol of an earlier part of this segment
            do --j; while (parts[j].dst > g);
            v = parts[j].src[g - parts[j].dst];
            if (v < 256u) return v;
            base = parts[j].dst;
        }
    };
    // 8 symbols (16 bytes) per lane where the destination allows it (slots are 16-byte aligned)
    const uint32_t lead = (uint32_t)((8u - (((uintptr_t)d >> 1) & 7u)) & 7u);
    for (uint32_t q = gid; q < lead && q < p.n; q += gsz) {
        uint32_t v = p.src[q];
        if (rebase && v >= 256u) v = fix(v);
        d[q] = (uint16_t)v;
    }
    if (p.n > lead) {
        const uint32_t chunks = (p.n - lead) >> 3;
        for (uint32_t c = gid; c < chunks; c += gsz) {
            const uint32_t at = lead + 8u * c;
            uint16_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p.src[at + k];
            if (rebase) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (v[k] >= 256u) v[k] = (uint16_t)fix(v[k]);
            }
            *reinterpret_cast<uint4 *>(d + at) = *reinterpret_cast<const uint4 *>(v);
        }
        for (uint32_t q = lead + 8u * chunks + gid; q < p.n; q += gsz) {
            uint32_t v = p.src[q];
            if (rebase && v >= 256u) v = fix(v);
            d[q] = (uint16_t)v;
        }
    }
}

// inflate_resolve.hip: the sequential decoder + device resolve
int inflate_raw_window_sequential(const uint8_t *src, size_t src_len, const uint8_t *d_window, uint32_t window_len, uint8_t *d_dst,
                                  size_t dst_cap, uint64_t *out_len, size_t *in_used, hipStream_t st);
// inflate_resolve.hip: symbols -> bytes (window in front, context chain, translate)
int inflate_resolve_symbols(const uint64_t *d_segs, size_t nsegs, uint16_t *sym, uint8_t *d_out, const uint8_t *d_window,
                            uint32_t window_len, hipStream_t st);

static thread_local int t_large_parts = 0;
#ifdef ZR_INFLATE_STATS
static std::vector<unsigned long long> g_dbg_starts;     // diagnostic builds: the last call's part starts and result words
static std::vector<uint32_t> g_dbg_res;
#endif

// why the device path handed the stream to the sequential decoder (readable through zng_rocm_last_error())
static int why(const char *reason) {
    set_error("inflate_large: sequential decoder (%s)", reason);
    return 0;
}

// returns 1 with *out_len / *in_used set, or 0 = "irregular: use the sequential decoder", or a negative error
static int inflate_large_try(Workspace *ws, const uint8_t *d_src, size_t src_len, const uint8_t *d_window, uint32_t window_len,
                             uint8_t *d_dst, size_t dst_cap, uint64_t *out_len, size_t *in_used, hipStream_t st) {
    if (src_len < (128u << 10) || src_len >= (1ull << 31)) return why("stream below 128 KiB (or 2 GiB and more)");
    // ---- candidates ---------------------------------------------------------------------------------------------
    const uint32_t cap1 = (uint32_t)std::min<size_t>(src_len / 64 + 4096, 64u << 20);
    const uint32_t cap2 = (uint32_t)std::min<size_t>(src_len / 512 + 4096, 8u << 20);      // starts are >= 2 KiB apart in the end
    uint8_t *fp = nullptr;
    if (int rc = scratch_reserve(ws, kScrLargeCand, ((size_t)cap1 + cap2) * 8 + 64, false, (void **)&fp)) return rc;
    unsigned long long *d_cand = (unsigned long long *)fp, *d_good = d_cand + cap1;
    uint32_t *d_n = (uint32_t *)(d_good + cap2);          // [0] survivors of F1, [1] of F2
    // first pass: the byte patterns alone (sync markers, byte-aligned stored blocks); when they leave no 128 KiB of compressed
    // bytes without a start, that is all the cutting the stream needs and the search of every bit position (F1's
    // main loop) and F2 are skipped -- 1.3 of 9.4 ms for the 256 MiB cfg3 stream of this library's level-6 class
    std::vector<unsigned long long> good;
    const uint32_t first = std::min<uint32_t>(cap2, 16384u);
    // what comes back from and goes up to the device between the launches passes through pinned memory (pageable copies of
    // these few hundred KiB were a good part of the 0.55 ms the host spends between the kernels of a 256 MiB stream)
    uint8_t *hp = nullptr;
    if (int rc = scratch_reserve(ws, kScrLargeCandHost, 64 + (size_t)first * 8, true, (void **)&hp)) return rc;
    uint32_t *n12 = (uint32_t *)hp;
    unsigned long long *h_good = (unsigned long long *)(hp + 64);
    n12[0] = n12[1] = 0;
    ZR_HIP(hipMemsetAsync(d_n, 0, 8, st));
    hipLaunchKernelGGL(find_headers_kernel<true>, dim3(4096), dim3(256), 0, st, d_src, (unsigned long long)src_len, d_good, d_n + 1, cap2);
    ZR_HIP(hipGetLastError());
    ZR_HIP(hipMemcpyAsync(n12, d_n, 8, hipMemcpyDeviceToHost, st));
    ZR_HIP(hipMemcpyAsync(h_good, d_good, (size_t)first * 8, hipMemcpyDeviceToHost, st));
    ZR_HIP(hipStreamSynchronize(st));
    // ... "enough" = no stretch of more than 128 KiB without one (a count would not do: the stored blocks of one incompressible
    // region are thousands of starts and say nothing about the Huffman blocks elsewhere)
    bool patterns_do = n12[1] >= 64u && n12[1] <= first;
    if (patterns_do) {
        good.assign(h_good, h_good + n12[1]);
        for (unsigned long long &b : good) b &= ~(1ull << 63);
        std::sort(good.begin(), good.end(), [](unsigned long long x, unsigned long long y) { return (x & ~(1ull << 62)) < (y & ~(1ull << 62)); });
        unsigned long long prev = 0;
        for (unsigned long long b62 : good) {
            const unsigned long long b = b62 & ~(1ull << 62);
            if (b - prev > 8ull * (128u << 10)) patterns_do = false;
            prev = b;
        }
        if (8ull * src_len - prev > 8ull * (128u << 10)) patterns_do = false;
    }
    if (!patterns_do) {
        ZR_HIP(hipMemsetAsync(d_n, 0, 8, st));
        hipLaunchKernelGGL(find_headers_kernel<false>, dim3(4096), dim3(256), 0, st, d_src, (unsigned long long)src_len, d_cand, d_n, cap1);
        ZR_HIP(hipGetLastError());
        hipLaunchKernelGGL(validate_headers_kernel, dim3(8192), dim3(kValLanes), 0, st, d_src, (unsigned long long)src_len, d_cand,
                           d_n, cap1, d_good, d_n + 1, cap2);
        ZR_HIP(hipGetLastError());
        // the counts and (what is almost always all of) the list in one round trip
        ZR_HIP(hipMemcpyAsync(n12, d_n, 8, hipMemcpyDeviceToHost, st));
        ZR_HIP(hipMemcpyAsync(h_good, d_good, (size_t)first * 8, hipMemcpyDeviceToHost, st));
        ZR_HIP(hipStreamSynchronize(st));
        if (n12[0] == 0 || n12[0] > cap1) return why("no candidate block starts, or far more than a deflate stream has");
        if (n12[1] <= cap2) good.assign(h_good, h_good + std::min(n12[1], first));
    }
    const uint32_t n2 = n12[1];
    if (n2 > cap2) return why("far more valid block headers than a deflate stream has");
    good.resize(n2);
    if (n2 > first) {
        ZR_HIP(hipMemcpyAsync(good.data() + first, d_good + first, (size_t)(n2 - first) * 8, hipMemcpyDeviceToHost, st));
        ZR_HIP(hipStreamSynchronize(st));
    }
    if (!patterns_do) {                                              // (the patterns' list has been sorted above)
        for (unsigned long long &b : good) b &= ~(1ull << 63);
        std::sort(good.begin(), good.end(), [](unsigned long long x, unsigned long long y) { return (x & ~(1ull << 62)) < (y & ~(1ull << 62)); });
    }
    std::vector<unsigned long long> starts;
    starts.push_back(0);
    const unsigned long long spacing = good.size() > kPartsUnthinned ? 8ull * kSpacingBytes : 1ull;
    size_t heavy = 1;                                     // parts that are not a stored block: the ones that take time
    for (unsigned long long b62 : good) {
        const unsigned long long b = b62 & ~(1ull << 62);
        if (b >= starts.back() + spacing && (b >> 3) + 16 < src_len) {
            starts.push_back(b);
            heavy += !(b62 >> 62);
        }
    }
    const size_t np = starts.size();
    const bool many = heavy > 12u * (size_t)ctx()->cus;
    if (np < 4) return why("fewer than four block starts found");

    // ---- parts ----------------------------------------------------------------------------------------------------
    std::vector<uint64_t> slot_off(np + 1, 0);
    const uint64_t slack = std::max<uint64_t>(64u << 10, std::min<uint64_t>(kSlotSlack, (2ull << 30) / np));
    for (size_t i = 0; i < np; ++i) {
        const uint64_t bytes = ((i + 1 < np ? starts[i + 1] : 8ull * src_len) - starts[i] + 7) >> 3;
        uint64_t capi = bytes * kSlotRatio + slack;
        capi = (capi + 7) & ~7ull;
        slot_off[i + 1] = slot_off[i] + capi;
    }
    uint8_t *sp = nullptr;
    const size_t jobs_b = (np * sizeof(InflateJobDev) + 255) & ~(size_t)255, starts_b = (np * 8 + 255) & ~(size_t)255,
                 res_b = (np * 32 + 255) & ~(size_t)255, slots_b = slot_off[np] * 2;
    if (scratch_reserve(ws, kScrLargeParts, jobs_b + starts_b + res_b + slots_b, false, (void **)&sp) != ZNG_ROCM_OK)
        return why("no room for the part slots");
    InflateJobDev *d_jobs = (InflateJobDev *)sp;
    unsigned long long *d_starts = (unsigned long long *)(sp + jobs_b);
    uint32_t *d_res = (uint32_t *)(sp + jobs_b + starts_b);
    uint16_t *d_slots = (uint16_t *)(sp + jobs_b + starts_b + res_b);
    uint8_t *hq = nullptr;
    if (scratch_reserve(ws, kScrLargePartsHost, jobs_b + starts_b + res_b, true, (void **)&hq) != ZNG_ROCM_OK)
        return why("no pinned memory for the part tables");
    InflateJobDev *jobs = (InflateJobDev *)hq;
    unsigned long long *h_starts = (unsigned long long *)(hq + jobs_b);
    uint32_t *res = (uint32_t *)(hq + jobs_b + starts_b);
    std::copy(starts.begin(), starts.end(), h_starts);
    for (size_t i = 0; i < np; ++i)
        jobs[i] = InflateJobDev{d_src, (uint8_t *)(d_slots + slot_off[i]), src_len, slot_off[i + 1] - slot_off[i],
                                i == 0 ? window_len : 32768u, 0u};
    ZR_HIP(hipMemcpyAsync(d_jobs, jobs, np * sizeof(InflateJobDev), hipMemcpyHostToDevice, st));
    ZR_HIP(hipMemcpyAsync(d_starts, h_starts, np * 8, hipMemcpyHostToDevice, st));
    if (int rc = launch_inflate_parts_device(d_jobs, np, d_res, d_starts, many, st)) return rc;
    ZR_HIP(hipMemcpyAsync(res, d_res, np * 32, hipMemcpyDeviceToHost, st));
    ZR_HIP(hipStreamSynchronize(st));
    // parts whose slot was too small (more than kSlotRatio : 1): once more, with room for deflate's worst case (1032 : 1)
    uint8_t *bigp = nullptr;
    std::vector<uint16_t *> slot_ptr(np);
    for (size_t i = 0; i < np; ++i) slot_ptr[i] = d_slots + slot_off[i];
    {
        std::vector<size_t> again;
        uint64_t need = 0;
        for (size_t i = 0; i < np; ++i)
            if (res[8 * i + 4] == kMsgOutFull) {
                again.push_back(i);
                const uint64_t bytes = ((i + 1 < np ? starts[i + 1] : 8ull * src_len) - starts[i] + 7) >> 3;
                need += (bytes * 1032u + kSlotSlack + 7) & ~7ull;
            }
        if (!again.empty()) {
            if (need * 2 > (24ull << 30) || scratch_reserve(ws, kScrLargeRetry, need * 2, false, (void **)&bigp) != ZNG_ROCM_OK)
                return why("parts with a ratio above 64 need more scratch than is reasonable");
            for (size_t i = 0; i < np; ++i) jobs[i].out_cap = 0;
            uint64_t at = 0;
            for (size_t i : again) {
                const uint64_t bytes = ((i + 1 < np ? starts[i + 1] : 8ull * src_len) - starts[i] + 7) >> 3;
                const uint64_t capi = (bytes * 1032u + kSlotSlack + 7) & ~7ull;
                slot_ptr[i] = (uint16_t *)bigp + at;
                jobs[i].out = (uint8_t *)slot_ptr[i];
                jobs[i].out_cap = capi;
                at += capi;
            }
            ZR_HIP(hipMemcpyAsync(d_jobs, jobs, np * sizeof(InflateJobDev), hipMemcpyHostToDevice, st));
            if (int rc = launch_inflate_parts_device(d_jobs, np, d_res, d_starts, many, st)) return rc;
            ZR_HIP(hipMemcpyAsync(res, d_res, np * 32, hipMemcpyDeviceToHost, st));
            ZR_HIP(hipStreamSynchronize(st));
        }
    }

#ifdef ZR_INFLATE_STATS
    g_dbg_starts = starts;
    g_dbg_res.assign(res, res + np * 8);
#endif
    // ---- the chain from bit 0 -------------------------------------------------------------------------------------
    std::vector<PartCopy> copies;
    uint64_t produced = 0;
    size_t cur = 0;
    unsigned long long end_bit = 0;
    for (;;) {
        const uint32_t *r = &res[8 * cur];
        const bool ended = r[3] == 1u;
        if (r[4] != kMsgNone || !(ended || r[6] != 0xffffffffu)) {                // error / truncation on the chain
            set_error("inflate_large: sequential decoder (part %zu of %zu at bit %llu: message %u \"%s\", %u symbols of %llu, "
                      "ended on %d)", cur, np, starts[cur], r[4], zng_rocm_inflate_message(r[4]), r[0],
                      (unsigned long long)jobs[cur].out_cap, (int)r[6]);
            return 0;
        }
        if ((uint64_t)r[5] > produced + window_len) return why("a distance reaches in front of the stream");
        copies.push_back(PartCopy{slot_ptr[cur], produced, 0, r[0], 0u});
        produced += r[0];
        end_bit = (unsigned long long)r[1] | ((unsigned long long)r[2] << 32);
        if (ended) break;
        cur = r[6];
        if (cur >= np) return why("bad chain link");
    }
    // segments for the context chain: consecutive parts until kSegmentBytes of output are together; what is left at the end
    // joins the last segment unless it is a segment's worth itself (only the stream's LAST segment may be short)
    std::vector<uint64_t> segs;                           // triples as inflate_resolve.hip wants them; only [3 s + 1] is used
    {
        size_t first = 0;
        std::vector<size_t> seg_first;
        for (size_t c = 0; c < copies.size(); ++c) {
            if (copies[c].dst + copies[c].n - copies[first].dst >= kSegmentBytes || c + 1 == copies.size()) {
                seg_first.push_back(first);
                first = c + 1;
            }
        }
        if (seg_first.size() > 1 && produced - copies[seg_first.back()].dst < 32768u) seg_first.pop_back();
        seg_first.push_back(copies.size());
        for (size_t g = 0; g + 1 < seg_first.size(); ++g) {
            for (size_t c = seg_first[g]; c < seg_first[g + 1]; ++c) {
                copies[c].gstart = copies[seg_first[g]].dst;
                copies[c].first = (uint32_t)seg_first[g];
            }
            segs.push_back(0);
            segs.push_back(copies[seg_first[g]].dst);
            segs.push_back(0);
        }
    }
    segs.push_back(0);
    segs.push_back(produced);
    segs.push_back(0);
    if (out_len) *out_len = produced;
    if (in_used) *in_used = (size_t)((end_bit + 7) >> 3);
    if (produced > dst_cap) {
        set_error("inflate output (%llu bytes) exceeds dst_cap", (unsigned long long)produced);
        return -5;
    }
    const size_t nparts = copies.size();
    const size_t nsegs = segs.size() / 3 - 1;
    // ---- symbols in place, resolve, translate ---------------------------------------------------------------------
    uint8_t *yp = nullptr;
    const size_t sym_b = ((size_t)produced + 32768 + 64) * 2, segs_b = (segs.size() * 8 + 255) & ~(size_t)255,
                 cp_b = (nparts * sizeof(PartCopy) + 255) & ~(size_t)255;
    if (scratch_reserve(ws, kScrLargeSym, segs_b + cp_b + sym_b, false, (void **)&yp) != ZNG_ROCM_OK) return why("no room for the symbols");
    uint64_t *d_segs = (uint64_t *)yp;
    PartCopy *d_cp = (PartCopy *)(yp + segs_b);
    uint16_t *sym = (uint16_t *)(yp + segs_b + cp_b) + 32768;
    ZR_HIP(hipMemcpyAsync(d_segs, segs.data(), segs.size() * 8, hipMemcpyHostToDevice, st));
    ZR_HIP(hipMemcpyAsync(d_cp, copies.data(), nparts * sizeof(PartCopy), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(compact_parts_kernel, dim3(16, (unsigned)nparts), dim3(256), 0, st, d_cp, sym);
    ZR_HIP(hipGetLastError());
    if (int rc = inflate_resolve_symbols(d_segs, nsegs, sym, d_dst, d_window, window_len, st)) return rc;
    ZR_HIP(hipStreamSynchronize(st));
    t_large_parts = (int)nparts;
    return 1;
}

void inflate_large_forget_parts() { t_large_parts = 0; }

// A large stream that is in HOST memory (zng_rocm_inflate_raw*, and with them uncompress2 and the gzip / zlib one-shots):
// up over PCIe once and through the device path; 0 = not done here (the caller's sequential decoder takes it).
int inflate_large_from_host(const uint8_t *src, size_t src_len, const uint8_t *d_window, uint32_t window_len, uint8_t *d_dst,
                            size_t dst_cap, uint64_t *out_len, size_t *in_used, hipStream_t st) {
    Workspace *ws = workspace_for(st);
    if (!ws) return 0;
    std::lock_guard<std::mutex> use(ws->mu);
    uint8_t *d_src = nullptr;
    if (scratch_reserve(ws, kScrLargeSrc, src_len + 64, false, (void **)&d_src) != ZNG_ROCM_OK) return 0;
    if (hipMemcpyAsync(d_src, src, src_len, hipMemcpyHostToDevice, st) != hipSuccess) return 0;
    uint64_t n = 0;
    size_t used = 0;
    const int rc = inflate_large_try(ws, d_src, src_len, d_window, window_len, d_dst, dst_cap, &n, &used, st);
    if (rc == 1 || rc == -5) {
        if (out_len) *out_len = n;
        if (in_used) *in_used = used;
        return rc;
    }
    return 0;
}

// The device path alone, for callers inside the library that have their own sequential decoder to fall back on
// (hook.hip): 1 = done, 0 = irregular (nothing usable was written), -5 with *out_len set = dst_cap too small, other
// negatives = errors.
int inflate_large_device_only(const uint8_t *d_src, size_t src_len, const uint8_t *d_window, uint32_t window_len, uint8_t *d_dst,
                              size_t dst_cap, uint64_t *out_len, size_t *in_used, hipStream_t st) {
    t_large_parts = 0;
    Workspace *ws = workspace_for(st);
    if (!ws) return ZNG_ROCM_ENOMEM;
    std::lock_guard<std::mutex> use(ws->mu);
    return inflate_large_try(ws, d_src, src_len, d_window, window_len, d_dst, dst_cap, out_len, in_used, st);
}

}  // namespace zr

using namespace zr;

extern "C" {

int zng_rocm_inflate_large_last_parts(void) { return t_large_parts; }
#ifdef ZR_INFLATE_STATS
unsigned zng_rocm_debug_large_parts(unsigned long long *starts, uint32_t *res8, unsigned cap) {
    const unsigned n = (unsigned)std::min<size_t>(cap, g_dbg_starts.size());
    for (unsigned i = 0; i < n; ++i) {
        starts[i] = g_dbg_starts[i];
        for (int k = 0; k < 8; ++k) res8[8 * i + k] = g_dbg_res[8 * (size_t)i + k];
    }
    return (unsigned)g_dbg_starts.size();
}
#endif

int zng_rocm_inflate_large_dev(const uint8_t *d_src, size_t src_len, const uint8_t *d_window, uint32_t window_len,
                               uint8_t *d_dst, size_t dst_cap, uint64_t *out_len, size_t *in_used, void *stream) {
    if (!ctx()) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    if ((!d_src && src_len) || window_len > 32768u || (window_len && !d_window)) return ZNG_ROCM_EINVAL;
    DeviceGuard dev;
    hipStream_t st = (hipStream_t)stream;
    t_large_parts = 0;
    {
        Workspace *ws = workspace_for(st);               // scratch is keyed by the caller's HIP stream (context.h)
        if (!ws) return ZNG_ROCM_ENOMEM;
        std::lock_guard<std::mutex> use(ws->mu);
        const int rc = inflate_large_try(ws, d_src, src_len, d_window, window_len, d_dst, dst_cap, out_len, in_used, st);
        if (rc != 0) return rc;
    }
    // irregular: the sequential decoder on a host copy of the stream says exactly what the reference would
    std::vector<uint8_t> host(src_len ? src_len : 1);
    if (src_len) {
        ZR_HIP(hipMemcpyAsync(host.data(), d_src, src_len, hipMemcpyDeviceToHost, st));
        ZR_HIP(hipStreamSynchronize(st));
    }
    return inflate_raw_window_sequential(host.data(), src_len, d_window, window_len, d_dst, dst_cap, out_len, in_used, st);
}

}  // extern "C"
