// deflate_stream.hip -- whole-stream deflate on device for MANY independent streams (pigz-style),
// level-1 class: the caller is zlib-ng's deflate_quick (deflate_quick.c:47-130), reached through
// DEFLATE_HOOK (deflate.c:1039).  Same building blocks -- quick_insert_string's hash
// (insert_string.c:11-13), a single chain-head probe, compare256, static Huffman trees
// (zng_tr_emit_lit / zng_tr_emit_dist, trees_emit.h:102-164; RFC 1951 3.2.6) -- re-laid-out for wave64:
//
//   deflate_quick_kernel  one 256-lane workgroup per stream, 256 consecutive positions per step (deflate_lz.h):
//        wavefront-wide insert_string through one LDS exchange per lane, head table (2^12 x u32 absolute
//        positions) in LDS, per-lane probe compare, every 64-position region parsed speculatively in parallel and
//        stitched, long matches extended with the wavefront-wide compare256.  The token a lane ends up with is coded
//        on the spot -- static-Huffman code + extra bits, bit offset from a wave scan -- and ORed into a 3 KiB bit
//        ring in LDS that is streamed out 128 bytes at a time.  The emit rides on the barriers the parse has anyway:
//        the tokens of batch k are placed after the first barriers of batch k+1, so the fused kernel has exactly the
//        barriers of the parse alone.  (Round 1 wrote one 32-bit selector per input byte to HBM and ran a second
//        kernel over them: 8 N bytes of traffic and 4 N of scratch -- 16 GiB for BASELINE.json configs[4].)
//        Also Adler-32 of the input (the {clen, check, ulen} row of SURVEY.md section 8e).
//
// The output is ONE final static block per stream (what deflate_quick emits for a Z_FINISH call),
// valid RFC 1951; it is not bit-identical to the reference's (every position is inserted into the
// hash, 12-bit table, region-parallel parse), which the reference's own test strategy never requires.
#include "context.h"
#include "deflate_dev.h"
#include "deflate_lz.h"

#include <stdlib.h>
#include <mutex>
#include <type_traits>

namespace zr {

constexpr int kQuickHashBits = 12;      // 16 KiB head table: up to 8 streams in flight per CU (see deflate_lz.h)

struct StreamJobDev {
    const uint8_t *in;          // first byte of the dictionary (= the plaintext when there is none)
    uint8_t       *out;
    uint32_t       n;           // dictionary + plaintext bytes
    uint32_t       out_cap;
    uint32_t       start;       // dictionary bytes: positions below `start` are entered into the hash, never emitted
    uint32_t       flags;       // ZNG_ROCM_BLOCK_*
};

// ---- static Huffman (RFC 1951 3.2.6; the reference's static_ltree / static_dtree, trees_tbl.h) -----------
__device__ __forceinline__ uint32_t rev_bits(uint32_t code, uint32_t n) { return __brev(code) >> (32 - n); }

// Both coders are written WITHOUT branches (selects only): this kernel is bound by the CU's scalar unit, and every
// divergent branch is four to six scalar instructions per wave whatever the lanes do (measured: the branchy forms of
// round 1's emitter cost 9 ms per 4 GiB inside the parse kernel against 2.5 ms as a kernel of their own).
__device__ __forceinline__ void static_literal(uint32_t b, uint32_t &bits, uint32_t &nb) {
    const uint32_t big = b >= 144u ? 1u : 0u;                       // 9-bit codes 110010000.. for 144..255, else 8-bit
    const uint32_t c = b + (big ? 0x190u - 144u : 0x30u);
    bits = __brev(c) >> (24u - big);
    nb = 8u + big;
}

__device__ __forceinline__ void static_match(uint32_t len, uint32_t dist, uint32_t &bits, uint32_t &nb) {
    // length symbol 257..285 with extra bits (RFC 1951 3.2.5; base_length/extra_lbits of trees_tbl.h):
    //   l = len - 3 < 8: 257 + l; l = 255: 285; else 257 + 4 eb + 4 + ((l >> eb) & 3) with eb = floor(log2 l) - 2.
    // With l | 4 under the logarithm the general form also yields 261 + (l & 3) for l < 8: right for 4..7, four too many below 4.
    const uint32_t l = len - 3u;
    const uint32_t top = l == 255u ? 1u : 0u;
    uint32_t eb = (31u - (uint32_t)__clz((int)(l | 4u))) - 2u;
    uint32_t sym = 261u + 4u * eb + ((l >> eb) & 3u) - (l < 4u ? 4u : 0u);
    uint32_t ev = l & ((1u << eb) - 1u);
    sym = top ? 285u : sym;
    eb = top ? 0u : eb;
    ev = top ? 0u : ev;
    const uint32_t wide = sym >= 280u ? 1u : 0u;                    // 7-bit codes for 256..279, 8-bit 11000000.. for 280..287
    const uint32_t c = sym - (wide ? 280u - 0xC0u : 256u);
    const uint32_t cn = 7u + wide;
    bits = (__brev(c) >> (25u - wide)) | (ev << cn);
    nb = cn + eb;
    // distance code 0..29 (5 bits, static) + extra bits: x = dist - 1 < 4: x; else 2 lg + ((x >> (lg-1)) & 1), lg - 1 extra bits.
    // With x | 2 under the logarithm the general form gives 2 + (x & 1) below 4: right for 2 and 3, two too many for 0 and 1.
    const uint32_t x = dist - 1u;
    const uint32_t lg = 31u - (uint32_t)__clz((int)(x | 2u));
    const uint32_t deb = lg - 1u;
    const uint32_t dc = 2u * lg + ((x >> deb) & 1u) - (x < 2u ? 2u : 0u);
    const uint32_t dev = x & ((1u << deb) - 1u);
    bits |= (__brev(dc) >> 27) << nb;
    nb += 5u;
    bits |= dev << nb;
    nb += deb;
}

// The bit ring: output bit b of the stream lives in ring[(b / 32) & (kRingWords - 1u)] while it is being assembled.  A batch
// produces at most 256 nine-bit literals = 72 words; words are streamed out once 128 are complete, 32 (one 128-byte
// line) at a time, so fewer than 128 + 72 + 32 < 512 are ever pending.
constexpr uint32_t kRingWords = 512;       // 2 KiB: with the 16 KiB head table still eight workgroups per CU
constexpr uint32_t kFlushWords = 128;

struct QuickShared {
    LzShared<kQuickHashBits> lz;
    uint32_t ring[kRingWords];
    uint4    wave_bits[2];                 // bits produced by each of the four waves in the last two batches (by batch parity)
    unsigned long long red_a[4], red_b[4];
};

__global__ __launch_bounds__(256)
void deflate_quick_kernel(const StreamJobDev *__restrict__ jobs, uint32_t *__restrict__ results) {
    __shared__ QuickShared sh;

    const StreamJobDev job = jobs[blockIdx.x];
    const uint8_t *in = job.in;
    const uint32_t n = job.n, start = job.start;
    const bool final_block = (job.flags & ZNG_ROCM_BLOCK_NOT_FINAL) == 0;
    uint32_t *outw = reinterpret_cast<uint32_t *>(job.out);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    for (int i = t; i < (1 << kQuickHashBits); i += 256) sh.lz.head[i] = 0;
    // block header: BFINAL, BTYPE = 01 (deflate_quick.c:30-34 emits it through zng_tr_emit_tree(s, STATIC_TREES, last))
    for (int i = t; i < (int)kRingWords; i += 256) sh.ring[i] = i == 0 ? (final_block ? 3u : 2u) : 0u;
    if (t == 0) sh.lz.cover = start;     // nothing below the first plaintext byte is ever produced
    __syncthreads();

    // dictionary priming: the whole batches below `start` only enter their positions into the hash
    uint32_t P0 = 0;
    for (; P0 + 256u <= start; P0 += 256u) {
        const uint32_t p = P0 + (uint32_t)t;
        lz_insert_batch<kQuickHashBits, 4>(n, P0, p + kLzMinMatch <= n ? load_u32(in + p) : 0u, &sh.lz, t);
    }

    uint32_t cursor = 3;                 // bits placed so far (uniform over the workgroup)
    uint32_t flushed = 0;                // whole words already written to `out`
    uint32_t pend_code = 0, pend_nb = 0, pend_pre = 0;     // this lane's token of the previous batch, not yet placed
    bool have_pending = false;
    unsigned long long accA = 0, accB = 0;                 // Adler-32, linear form: B += (n - pos) * byte

    // place the pending batch: every lane ORs its token at cursor + (bits of the waves before it) + (its wave prefix)
    auto place = [&](int parity) {
        const uint4 wb = sh.wave_bits[parity];
        const uint32_t before = (wave > 0 ? wb.x : 0u) + (wave > 1 ? wb.y : 0u) + (wave > 2 ? wb.z : 0u);
        if (pend_nb) {
            const uint32_t at = cursor + before + pend_pre;
            const uint32_t word = at >> 5;
            const unsigned long long wide = (unsigned long long)pend_code << (at & 31u);     // <= 31 + 31 bits
            atomicOr(&sh.ring[word & (kRingWords - 1u)], (uint32_t)wide);
            atomicOr(&sh.ring[(word + 1) & (kRingWords - 1u)], (uint32_t)(wide >> 32));
        }
        cursor += wb.x + wb.y + wb.z + wb.w;
    };
    // stream out complete words; everything below `cursor` was ORed before the last barrier
    auto flush = [&](uint32_t keep_below) {
        const uint32_t full = cursor >> 5;
        if (full - flushed >= keep_below) {
            const uint32_t cnt = keep_below > 1 ? ((full - flushed) & ~31u) : full - flushed;
            for (uint32_t i = (uint32_t)t; i < cnt; i += 256) {
                const uint32_t slot = (flushed + i) & (kRingWords - 1u);
                __builtin_nontemporal_store(sh.ring[slot], outw + flushed + i);
                sh.ring[slot] = 0;
            }
            flushed += cnt;
        }
    };

    u32x4_unaligned own = load_16_guarded(in, P0 + (uint32_t)t, n);  // this lane's 16 bytes of the current batch
    int parity = 0;
    for (uint32_t P = P0; P < n; P += 256) {
        // every position of the NEXT batch has its 16 bytes inside the stream / of this batch its whole lookahead
        const uint32_t pn = P + 256u + (uint32_t)t;
        const bool full = n >= 256u + kStdMaxMatch + 4u && P <= n - (256u + kStdMaxMatch + 4u);
        const bool next_inside = n >= 512u + 16u && P <= n - (512u + 16u);
        u32x4_unaligned own_next = {0u, 0u, 0u, 0u};                           // prefetch of the next batch
        if (next_inside) own_next = load_u128(in + pn);
        else if (pn >= P) own_next = load_16_guarded(in, pn, n);
        const LzPick r = full ? lz_batch<kQuickHashBits, 4, true>(in, n, P, own, &sh.lz, t)
                              : lz_batch<kQuickHashBits, 4, false>(in, n, P, own, &sh.lz, t);
        // (lz_batch ended behind barriers: the wave totals and the ORs of the previous iteration are visible)
        flush(kFlushWords);
        if (have_pending) place(parity ^ 1);
        const uint32_t p = P + (uint32_t)t;
        const uint32_t byte = p >= start ? own.x & 0xffu : 0u;   // zero beyond the end of the stream; the dictionary
        accA += byte;                                            // is not part of the checksum
        accB += (unsigned long long)(n - p) * byte;
        // both codes are computed and one is selected: no branch (lanes without a token carry r.len = r.dist = 0,
        // which the match coder turns into harmless garbage that the select drops)
        uint32_t mcode, mnb, lcode, lnb;
        static_match(r.kind == 2u ? r.len : 3u, r.kind == 2u ? r.dist : 1u, mcode, mnb);
        static_literal(byte, lcode, lnb);
        const uint32_t code = r.kind == 2u ? mcode : (r.kind == 1u ? lcode : 0u);
        const uint32_t nb = r.kind == 2u ? mnb : (r.kind == 1u ? lnb : 0u);
        uint32_t incl = nb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) reinterpret_cast<uint32_t *>(&sh.wave_bits[parity])[wave] = incl;
        pend_code = code;
        pend_nb = nb;
        pend_pre = incl - nb;
        have_pending = true;
        parity ^= 1;
        own = own_next;
    }
    __syncthreads();
    if (have_pending) place(parity ^ 1);
    __syncthreads();
    flush(1);                                                // every complete word

    // end-of-block code 256 = seven 0 bits (zng_emit_end_block, trees_emit.h:169-180), then pad to a byte
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        accA += __shfl_xor(accA, m, 64);
        accB += __shfl_xor(accB, m, 64);
    }
    if (lane == 0) {
        sh.red_a[wave] = accA % kAdlerBase;
        sh.red_b[wave] = accB % kAdlerBase;
    }
    __syncthreads();
    if (t == 0) {
        uint8_t *outb = job.out;
        uint32_t cbits = (cursor & 31u) + 7u;                // the partial word + EOB
        uint32_t cw = sh.ring[(cursor >> 5) & (kRingWords - 1u)];
        uint32_t wbase = flushed;                            // == cursor >> 5
        if (cbits >= 32) {
            outw[wbase++] = cw;
            cw = 0;
            cbits -= 32;
        }
        uint32_t bytes = wbase * 4u;
        const bool sync = !final_block && (job.flags & ZNG_ROCM_BLOCK_SYNC_FLUSH) != 0;
        if (sync) cbits += 3;                                // header of an empty stored block: BFINAL = 0, BTYPE = 00
        for (uint32_t k = 0; k < (cbits + 7u) / 8u; ++k) outb[bytes++] = (uint8_t)((unsigned long long)cw >> (8 * k));   // up to 34 bits
        if (sync) {                                          // ... byte aligned, LEN = 0, NLEN = 0xffff: what Z_SYNC_FLUSH
            outb[bytes++] = 0x00;                            // appends (deflate.c:1064-1076, zng_tr_stored_block)
            outb[bytes++] = 0x00;
            outb[bytes++] = 0xff;
            outb[bytes++] = 0xff;
        }
        const unsigned long long A = (sh.red_a[0] + sh.red_a[1] + sh.red_a[2] + sh.red_a[3]) % kAdlerBase;
        const unsigned long long B = (sh.red_b[0] + sh.red_b[1] + sh.red_b[2] + sh.red_b[3]) % kAdlerBase;
        results[2 * blockIdx.x] = bytes;
        results[2 * blockIdx.x + 1] = (uint32_t)(((1 + A) % kAdlerBase) | ((((unsigned long long)(n - start) + B) % kAdlerBase) << 16));
    }
}

// ---- the same coder with ONE wavefront per stream ------------------------------------------------------------------
// For batches with enough streams to fill the machine by themselves (>= 8 per CU).  A stream is walked 64 positions at a
// time by a single wave: no workgroup barrier exists in this kernel (the 256-lane form above has six per 256
// positions), the greedy parse simply carries its "first byte not yet produced" from step to step in a scalar
// register -- no speculation, no stitching -- and the one long-latency operation of a step, the gather of the
// candidates' bytes, is issued a whole step ahead: while step k is parsed and coded, the hash insertions and the
// candidate loads of step k+1 are already under way.  LDS per stream is the same (16 KiB head + 2 KiB bit ring), so a
// CU holds eight such waves.
struct WaveShared {
    uint32_t head[1 << kQuickHashBits];
    uint32_t ring[kRingWords];
};

__device__ __forceinline__ u32x4_unaligned shfl16(const u32x4_unaligned &v, int src) {
    u32x4_unaligned r = {(uint32_t)__shfl((int)v.x, src, 64), (uint32_t)__shfl((int)v.y, src, 64),
                         (uint32_t)__shfl((int)v.z, src, 64), (uint32_t)__shfl((int)v.w, src, 64)};
    return r;
}

__global__ __launch_bounds__(64)
void deflate_quick_wave_kernel(const StreamJobDev *__restrict__ jobs, uint32_t *__restrict__ results) {
    __shared__ WaveShared sh;

    const StreamJobDev job = jobs[blockIdx.x];
    const uint8_t *in = job.in;
    const uint32_t n = job.n, start = job.start;
    const bool final_block = (job.flags & ZNG_ROCM_BLOCK_NOT_FINAL) == 0;
    uint32_t *outw = reinterpret_cast<uint32_t *>(job.out);
    const int lane = threadIdx.x;

    for (int i = lane; i < (1 << kQuickHashBits) / 4; i += 64) reinterpret_cast<uint4 *>(sh.head)[i] = make_uint4(0, 0, 0, 0);
    for (int i = lane; i < (int)kRingWords; i += 64) sh.ring[i] = i == 0 ? (final_block ? 3u : 2u) : 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    // one step's hash insertion: lanes of one LDS exchange are served in lane order (tools/micro/lds_xchg_order.hip), so
    // every lane gets what quick_insert_string (insert_string_tpl.h:58-75) would have returned
    auto insert = [&](uint32_t W, uint32_t first4) -> uint32_t {
        const uint32_t p = W + (uint32_t)lane;
        uint32_t old = 0;
        if (p + kLzMinMatch <= n && p + kLzMinMatch > p) old = atomicExch(&sh.head[lz_hash<kQuickHashBits>(first4)], p + 1u);
        return old;
    };
    auto load16 = [&](uint32_t p, bool inside) -> u32x4_unaligned {
        return inside ? load_u128(in + p) : load_16_guarded(in, p, n);
    };
    const u32x4_unaligned zero16 = {0u, 0u, 0u, 0u};

    // dictionary priming: whole steps below `start` only enter their positions into the hash
    uint32_t W0 = 0;
    for (; W0 + 64u <= start; W0 += 64u) {
        const uint32_t p = W0 + (uint32_t)lane;
        (void)insert(W0, p + kLzMinMatch <= n ? load_u32(in + p) : 0u);
    }

    uint32_t cover = start;              // first byte not yet produced (uniform)
    uint32_t cursor = 3, flushed = 0;    // bits placed / whole words written (uniform)
    unsigned long long accA = 0, accB = 0;

    // ---- software pipeline: own bytes two steps ahead, insertion + candidate gather one step ahead.
    // Two things keep the loads of step k+1 in flight while step k is parsed and coded (measured: without them the
    // kernel took 95 ms for cfg5 against the 256-lane form's 48):
    //   * a value still on its way from memory is never COPIED (the copy would wait for it and for every load issued
    //     before it): the candidates' bytes live in two register sets used alternately, the loop is unrolled by two,
    //     and a step's own-bytes prefetch is issued BEFORE its gathers, so that its consumer waits for it alone;
    //   * every path through a step of the main loop issues the SAME number of loads (a lane without a candidate
    //     gathers its own position instead), or the compiler could not count them and would drain them all.
    // The last few steps of a stream (FAST = false) take the guarded forms of everything and need no such care.
    auto inside16 = [&](uint32_t W) { return n >= 64u + 16u && W <= n - (64u + 16u); };        // every lane has 16 bytes
    auto inside48 = [&](uint32_t W) { return n >= 64u + 48u && W <= n - (64u + 48u); };        // ... and its candidate 32
    auto gather_guarded = [&](uint32_t W, uint32_t old, u32x4_unaligned &c0, u32x4_unaligned &c1) {
        const uint32_t p = W + (uint32_t)lane, c = old - 1u;             // old == 0 -> 0xffffffff: fails c < p
        c0 = c1 = zero16;
        if (c < p && p - c <= kLzMaxDist) {
            const bool in32 = inside48(W);                               // c + 32 <= p + 31 < n
            c0 = in32 ? load_u128(in + c) : load_16_guarded(in, c, n);
            c1 = in32 ? load_u128(in + c + 16u) : load_16_guarded(in, c + 16u, n);
        }
    };
    u32x4_unaligned own = load16(W0 + (uint32_t)lane, inside16(W0));
    uint32_t old = W0 < n ? insert(W0, own.x) : 0u;
    u32x4_unaligned own1 = W0 + 64u < n && W0 + 64u > W0 ? load16(W0 + 64u + (uint32_t)lane, inside16(W0 + 64u)) : zero16;
    u32x4_unaligned candA0, candA1, candB0, candB1;
    gather_guarded(W0, old, candA0, candA1);

    auto step = [&](auto fast_tag, uint32_t W, const u32x4_unaligned &cand0, const u32x4_unaligned &cand1,
                    u32x4_unaligned &cn0, u32x4_unaligned &cn1) {
        constexpr bool FAST = decltype(fast_tag)::value;     // W + 336 <= n: this step, the next and the prefetch are inside
        const uint32_t Wn = W + 64u, Wnn = W + 128u;
        // stage the next step: its insertions and candidate loads fly while this step is parsed and coded
        uint32_t old1 = 0;
        u32x4_unaligned own2 = zero16;
        if constexpr (FAST) {
            own2 = load_u128(in + Wnn + (uint32_t)lane);
            const uint32_t p1 = Wn + (uint32_t)lane;
            old1 = atomicExch(&sh.head[lz_hash<kQuickHashBits>(own1.x)], p1 + 1u);
            const uint32_t c1v = old1 - 1u;
            const uint32_t at = (c1v < p1 && p1 - c1v <= kLzMaxDist) ? c1v : p1;     // no candidate: its own bytes (l = 32, dropped below)
            cn0 = load_u128(in + at);
            cn1 = load_u128(in + at + 16u);
        } else {
            cn0 = cn1 = zero16;
            if (Wn < n && Wn > W) {
                if (Wnn < n && Wnn > Wn) own2 = load16(Wnn + (uint32_t)lane, inside16(Wnn));
                old1 = insert(Wn, own1.x);
                gather_guarded(Wn, old1, cn0, cn1);
            }
        }

        // ---- probe of this step: up to 32 bytes against the candidate
        const uint32_t p = W + (uint32_t)lane;
        const uint32_t maxlen = FAST ? kStdMaxMatch : (p < n ? ((n - p) < kStdMaxMatch ? (n - p) : kStdMaxMatch) : 0u);
        uint32_t len = 0, dist = 0;
        {
            // the lane's bytes 16..31 are the first 16 of the lane sixteen positions on: this step's or the next one's.
            // (Cross-lane reads: executed by ALL lanes, outside any divergent branch.)
            const u32x4_unaligned a = shfl16(own, (lane + 16) & 63), b = shfl16(own1, (lane + 16) & 63);
            const u32x4_unaligned mine16 = lane < 48 ? a : b;
            const uint32_t c = old - 1u;
            uint32_t l = lz_prefix16(cand0, own);
            if (l == 16u) l += lz_prefix16(cand1, mine16);
            l = l < maxlen ? l : maxlen;
            if (c < p && p - c <= kLzMaxDist) {
                len = l >= kLzMinMatch ? l : 0u;
                dist = p - c;
            }
        }

        // ---- greedy parse of [W, W + lim), starting at the first byte the previous steps left unproduced
        const uint32_t lim = FAST ? 64u : ((n - W) < 64u ? (n - W) : 64u);
        const uint32_t skip = cover > W ? cover - W : 0u;
        unsigned long long chosen = 0, covered = lz_bits_below(skip < 64u ? skip : 64u);
        if (skip < lim) {
            const unsigned long long limmask = FAST ? ~0ull : lz_bits_below(lim);
            unsigned long long avail = __ballot(len >= kLzMinMatch) & limmask & ~covered;
            uint32_t end = lim;                                          // first byte after this step's last token
            while (avail) {
                const uint32_t m = (uint32_t)__builtin_ctzll(avail);
                uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)m);
                if (L >= kLzProbe) {                                     // the probe saturated: measure the rest wave-wide
                    const uint32_t pabs = W + m;
                    const uint32_t mlen = FAST ? kStdMaxMatch : ((n - pabs) < kStdMaxMatch ? (n - pabs) : kStdMaxMatch);
                    if (L < mlen) {
                        const uint32_t D = (uint32_t)__builtin_amdgcn_readlane((int)dist, (int)m);
                        uint32_t rem = mlen - L;
                        if (rem > 256u) rem = 256u;
                        L += lz_extend_wave(in + pabs + L, in + pabs - D + L, rem, lane);
                        if ((uint32_t)lane == m) len = L;
                    }
                }
                chosen |= 1ull << m;
                const uint32_t e = m + L;
                if (e >= 64u) {
                    if (m < 63u) covered |= ~0ull << (m + 1u);
                    end = e;
                    break;
                }
                covered |= (~0ull << (m + 1u)) & ~(~0ull << e);
                avail &= ~0ull << e;
            }
            cover = W + end;
            const unsigned long long lits = ~covered & ~chosen & limmask;
            // ---- code this lane's token, place it
            const uint32_t byte = own.x & 0xffu;
            const bool is_match = (chosen >> lane) & 1ull, is_lit = (lits >> lane) & 1ull;
            uint32_t mcode, mnb, lcode, lnb;
            static_match(is_match ? len : 3u, is_match ? dist : 1u, mcode, mnb);
            static_literal(byte, lcode, lnb);
            const uint32_t code = is_match ? mcode : (is_lit ? lcode : 0u);
            const uint32_t nb = is_match ? mnb : (is_lit ? lnb : 0u);
            uint32_t incl = nb;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
                if (lane >= d) incl += up;
            }
            if (nb) {
                const uint32_t at = cursor + incl - nb;
                const uint32_t word = at >> 5;
                const unsigned long long wide = (unsigned long long)code << (at & 31u);
                atomicOr(&sh.ring[word & (kRingWords - 1u)], (uint32_t)wide);
                atomicOr(&sh.ring[(word + 1) & (kRingWords - 1u)], (uint32_t)(wide >> 32));
            }
            cursor += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t full = cursor >> 5;
            if (full - flushed >= kFlushWords) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                const uint32_t cnt = (full - flushed) & ~31u;
                for (uint32_t i = (uint32_t)lane; i < cnt; i += 64u) {
                    const uint32_t slot = (flushed + i) & (kRingWords - 1u);
                    __builtin_nontemporal_store(sh.ring[slot], outw + flushed + i);
                    sh.ring[slot] = 0;
                }
                flushed += cnt;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
        }
        {
            const uint32_t byte = (p >= start && (FAST || p < n)) ? own.x & 0xffu : 0u;   // the dictionary is not part of the checksum
            accA += byte;
            accB += (unsigned long long)(n - p) * byte;
        }
        own = own1;
        own1 = own2;
        old = old1;
    };
    uint32_t W = W0;
    while (n >= 336u && W <= n - 336u) {
        step(std::true_type{}, W, candA0, candA1, candB0, candB1);
        W += 64u;
        if (!(W <= n - 336u)) {                              // odd number of fast steps: the sets have swapped roles
            candA0 = candB0;
            candA1 = candB1;
            break;
        }
        step(std::true_type{}, W, candB0, candB1, candA0, candA1);
        W += 64u;
    }
    for (; W < n && W >= W0; W += 64u) {                     // the last steps of the stream: guarded, no pipeline to keep
        step(std::false_type{}, W, candA0, candA1, candB0, candB1);
        candA0 = candB0;
        candA1 = candB1;
        if (W + 64u < W) break;
    }

    // ---- everything that is left in the ring, the end-of-block code, the optional sync marker
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    {
        const uint32_t full = cursor >> 5;
        for (uint32_t i = flushed + (uint32_t)lane; i < full; i += 64u) {
            __builtin_nontemporal_store(sh.ring[i & (kRingWords - 1u)], outw + i);
        }
        flushed = full;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        accA += __shfl_xor(accA, m, 64);
        accB += __shfl_xor(accB, m, 64);
    }
    if (lane == 0) {
        uint8_t *outb = job.out;
        uint32_t cbits = (cursor & 31u) + 7u;                // the partial word + EOB (seven 0 bits, trees_emit.h:169-180)
        uint32_t cw = sh.ring[(cursor >> 5) & (kRingWords - 1u)];
        uint32_t wbase = flushed;
        if (cbits >= 32) {
            outw[wbase++] = cw;
            cw = 0;
            cbits -= 32;
        }
        uint32_t bytes = wbase * 4u;
        const bool sync = !final_block && (job.flags & ZNG_ROCM_BLOCK_SYNC_FLUSH) != 0;
        if (sync) cbits += 3;                                // header of an empty stored block: BFINAL = 0, BTYPE = 00
        for (uint32_t k = 0; k < (cbits + 7u) / 8u; ++k) outb[bytes++] = (uint8_t)((unsigned long long)cw >> (8 * k));
        if (sync) {                                          // deflate.c:1064-1076
            outb[bytes++] = 0x00;
            outb[bytes++] = 0x00;
            outb[bytes++] = 0xff;
            outb[bytes++] = 0xff;
        }
        const unsigned long long A = accA % kAdlerBase, B = accB % kAdlerBase;
        results[2 * blockIdx.x] = bytes;
        results[2 * blockIdx.x + 1] = (uint32_t)(((1 + A) % kAdlerBase) | ((((unsigned long long)(n - start) + B) % kAdlerBase) << 16));
    }
}

}  // namespace zr

using namespace zr;

extern "C" {

size_t zng_rocm_deflate_quick_bound(size_t source_len) {
    // 3 header bits + at most 9 bits per input byte (a match of >= 4 bytes costs <= 31 bits) + 7 EOB bits,
    // cf. DEFLATE_QUICK_OVERHEAD / DEFLATE_BLOCK_OVERHEAD (zutil.h:71-79, deflate.c:773-777); rounded so that
    // consecutive streams stay 16-byte aligned.
    size_t b = source_len + ((source_len + 7) >> 3) + 8 + 5;          // + the sync-flush terminator of a non-final block
    return (b + 15) & ~(size_t)15;
}

int zng_rocm_deflate_quick_dev(const zng_rocm_stream_job *jobs, size_t njobs, uint32_t *d_results, void *stream) {
    if (!ctx()) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    if (!njobs) return ZNG_ROCM_OK;
    if (!jobs || !d_results) return ZNG_ROCM_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard dev;
    // All scratch of this entry point belongs to the caller's HIP stream: two callers on two streams (two pigz-style
    // host threads) never share a buffer, and calls on one stream are ordered by the stream itself.
    Workspace *ws = workspace_for(st);
    if (!ws) return ZNG_ROCM_ENOMEM;
    std::lock_guard<std::mutex> use(ws->mu);
    StreamJobDev *d_jobs = nullptr, *h_jobs = nullptr;
    if (int rc = scratch_reserve(ws, kScrQuickJobs, njobs * sizeof(StreamJobDev), false, (void **)&d_jobs)) return rc;
    // the previous call's table may still be on its way to the device: wait for that copy, not for its kernels
    if (int rc = host_tables_acquire(ws)) return rc;
    if (int rc = scratch_reserve(ws, kScrQuickJobsHost, njobs * sizeof(StreamJobDev), true, (void **)&h_jobs)) return rc;
    for (size_t i = 0; i < njobs; ++i) {
        const zng_rocm_stream_job &j = jobs[i];
        if (!j.out || (j.in_len && !j.in) || ((uintptr_t)j.out & 3) || j.out_cap < zng_rocm_deflate_quick_bound(j.in_len)) {
            set_error("job %zu: out must be 4-byte aligned and out_cap >= zng_rocm_deflate_quick_bound(in_len)", i);
            return ZNG_ROCM_EINVAL;
        }
        if (j.dict_len > 32768u || (uint64_t)j.in_len + j.dict_len > 0xffffffffull || (j.dict_len && !j.in) ||
            (j.flags & ~(uint32_t)(ZNG_ROCM_BLOCK_NOT_FINAL | ZNG_ROCM_BLOCK_SYNC_FLUSH))) {
            set_error("job %zu: dict_len above 32768, unknown flags, or dictionary + plaintext of 4 GiB and more", i);
            return ZNG_ROCM_EINVAL;
        }
        h_jobs[i] = StreamJobDev{j.in - j.dict_len, j.out, j.in_len + j.dict_len, j.out_cap, j.dict_len, j.flags};
    }
    ZR_HIP(hipMemcpyAsync(d_jobs, h_jobs, njobs * sizeof(StreamJobDev), hipMemcpyHostToDevice, st));
    if (int rc = host_tables_release(ws, st)) return rc;
    // Four waves per stream is the form that ships.  The one-wave-per-stream form (no barriers, 18 KiB of LDS per wave)
    // measured slower on every stream count tried (both forms are instruction-issue bound, and
    // two waves per SIMD cover less latency than eight).  Measurement builds (-DZR_MEASURE_FORMS) select it with
    // ZNG_ROCM_QUICK_FORM=wave; the product reads no environment variable.
#ifdef ZR_MEASURE_FORMS
    static const bool wave_form = [] {                          // read once
        const char *f = getenv("ZNG_ROCM_QUICK_FORM");
        return f && f[0] == 'w';
    }();
    if (wave_form) ZR_LAUNCH_TRACED(deflate_quick_wave_kernel, dim3((unsigned)njobs), dim3(64), st, d_jobs, d_results);
    else
#endif
    ZR_LAUNCH_TRACED(deflate_quick_kernel, dim3((unsigned)njobs), dim3(256), st, d_jobs, d_results);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

}  // extern "C"
