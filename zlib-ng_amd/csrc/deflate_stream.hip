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
    ZR_LAUNCH_TRACED(deflate_quick_kernel, dim3((unsigned)njobs), dim3(256), st, d_jobs, d_results);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

}  // extern "C"
