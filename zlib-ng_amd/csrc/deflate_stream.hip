// deflate_stream.hip -- whole-stream deflate on device for MANY independent streams (pigz-style),
// level-1 class: the caller is zlib-ng's deflate_quick (deflate_quick.c:47-130), reached through
// DEFLATE_HOOK (deflate.c:1039).  Same building blocks -- quick_insert_string's hash
// (insert_string.c:11-13), a single chain-head probe, compare256, static Huffman trees
// (zng_tr_emit_lit / zng_tr_emit_dist, trees_emit.h:102-164; RFC 1951 3.2.6) -- re-laid-out for wave64:
//
//   K1 lz_parse_kernel   one 256-lane workgroup per stream, 256 consecutive positions per step
//        (deflate_lz.h): wavefront-wide insert_string through one LDS exchange per lane, head table
//        (2^14 x u32 absolute positions) in LDS, per-lane probe compare, every 64-position region parsed
//        speculatively in parallel and stitched, long matches extended with the wavefront-wide compare256.
//        Output: one 32-bit selector per input position (skip / literal / match{len,dist}).
//   K2 emit_static_kernel one workgroup per stream: static-Huffman code per selector, bit offsets by
//        a block scan, bits assembled in an LDS tile with ds_or and streamed out with plain stores;
//        also Adler-32 of the input (the {clen, check, ulen} row of SURVEY.md section 8e).
//
// The output is ONE final static block per stream (what deflate_quick emits for a Z_FINISH call),
// valid RFC 1951; it is not bit-identical to the reference's (every position is inserted into the
// hash, 14-bit table, region-parallel parse), which the reference's own test strategy never requires.
#include "context.h"
#include "deflate_dev.h"
#include "deflate_lz.h"

#include <stdlib.h>
#include <mutex>

namespace zr {

constexpr int kQuickHashBits = 12;      // 16 KiB head table: up to 8 streams in flight per CU (see deflate_lz.h)

struct StreamJobDev {
    const uint8_t *in;
    uint8_t       *out;
    uint32_t       in_len;
    uint32_t       out_cap;
    uint64_t       sel_off;      // first selector of this stream in the workspace
};

// K1: LZ77 front end (deflate_lz.h) -- one selector per input position:
//   0 = produced by an earlier match, 0x40000000 = literal, 0x80000000 | (len-3) << 16 | (dist-1) = match.
// The next batch's input is prefetched while the current one is parsed.
__global__ __launch_bounds__(256)
void lz_parse_kernel(const StreamJobDev *__restrict__ jobs, uint32_t *__restrict__ sel_base) {
    __shared__ LzShared<kQuickHashBits> sh;

    const StreamJobDev job = jobs[blockIdx.x];
    const uint8_t *in = job.in;
    const uint32_t n = job.in_len;
    uint32_t *sel = sel_base + job.sel_off;
    const int t = threadIdx.x;

    for (int i = t; i < (1 << kQuickHashBits); i += 256) sh.head[i] = 0;
    if (t == 0) sh.cover = 0;
    __syncthreads();

    uint32_t val = (uint32_t)t + kLzMinMatch <= n ? load_u32(in + t) : 0u;
    for (uint32_t P = 0; P < n; P += 256) {
        const uint32_t pn = P + 256u + (uint32_t)t;
        const uint32_t val_next = (pn + kLzMinMatch <= n && pn >= P) ? load_u32(in + pn) : 0u;   // prefetch
        // every position of the batch has its whole lookahead inside the stream: the guard-free instantiation
        const bool full = n >= 256u + kStdMaxMatch + 4u && P <= n - (256u + kStdMaxMatch + 4u);
        const LzPick r = full ? lz_batch<kQuickHashBits, 4, true>(in, n, P, val, &sh, t)
                              : lz_batch<kQuickHashBits, 4, false>(in, n, P, val, &sh, t);
        const uint32_t p = P + (uint32_t)t;
        if (p < n) {
            uint32_t s = 0;
            if (r.kind == 2u) s = 0x80000000u | ((r.len - 3u) << 16) | (r.dist - 1u);
            else if (r.kind == 1u) s = 0x40000000u;
            __builtin_nontemporal_store(s, sel + p);        // read once, by the emitter, long after
        }
        val = val_next;
    }
}

// ---- static Huffman (RFC 1951 3.2.6; the reference's static_ltree / static_dtree, trees_tbl.h) -----------
__device__ __forceinline__ uint32_t rev_bits(uint32_t code, uint32_t n) { return __brev(code) >> (32 - n); }

__device__ __forceinline__ void static_literal(uint32_t b, uint32_t &bits, uint32_t &nb) {
    if (b < 144) { bits = rev_bits(0x30 + b, 8); nb = 8; }
    else         { bits = rev_bits(0x190 + (b - 144), 9); nb = 9; }
}

__device__ __forceinline__ void static_match(uint32_t len, uint32_t dist, uint32_t &bits, uint32_t &nb) {
    // length symbol 257..285 with extra bits (RFC 1951 3.2.5; base_length/extra_lbits of trees_tbl.h)
    const uint32_t l = len - 3;
    uint32_t sym, eb = 0, ev = 0;
    if (l < 8) sym = 257 + l;
    else if (l == 255) sym = 285;
    else {
        const uint32_t lg = 31u - (uint32_t)__clz((int)l);          // 3..7
        eb = lg - 2;
        sym = 257 + 4 * eb + 4 + ((l >> eb) & 3u);
        ev = l & ((1u << eb) - 1u);
    }
    uint32_t code, cn;
    if (sym < 280) { code = rev_bits(sym - 256, 7); cn = 7; }
    else           { code = rev_bits(0xC0 + (sym - 280), 8); cn = 8; }
    bits = code | (ev << cn);
    nb = cn + eb;
    // distance code 0..29, 5-bit static code, extra bits
    const uint32_t x = dist - 1;
    uint32_t dc, deb = 0, dev = 0;
    if (x < 4) dc = x;
    else {
        const uint32_t lg = 31u - (uint32_t)__clz((int)x);          // 2..14
        deb = lg - 1;
        dc = 2 * lg + ((x >> deb) & 1u);
        dev = x & ((1u << deb) - 1u);
    }
    bits |= rev_bits(dc, 5) << nb;
    nb += 5;
    bits |= dev << nb;
    nb += deb;
}

constexpr int kEmitPer = 16;                 // positions per lane per tile
constexpr int kEmitTile = 256 * kEmitPer;    // 4096 positions
constexpr int kEmitWords = kEmitTile + 8;    // worst case 31 bits per position < 1 word each

__global__ __launch_bounds__(256)
void emit_static_kernel(const StreamJobDev *__restrict__ jobs, const uint32_t *__restrict__ sel_base,
                        uint32_t *__restrict__ results) {
    __shared__ uint32_t obuf[kEmitWords];
    __shared__ uint32_t wave_tot[4];
    __shared__ unsigned long long red_a[4], red_b[4];

    const StreamJobDev job = jobs[blockIdx.x];
    const uint8_t *in = job.in;
    const uint32_t n = job.in_len;
    const uint32_t *sel = sel_base + job.sel_off;
    uint32_t *outw = reinterpret_cast<uint32_t *>(job.out);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    uint32_t wbase = 0;          // whole words already written
    uint32_t cw = 3u, cbits = 3; // carry word: block header BFINAL=1, BTYPE=01 (deflate_quick.c:30-34)
    unsigned long long accA = 0, accB = 0;

    for (uint32_t base = 0; base < n; base += kEmitTile) {
        const uint32_t p0 = base + (uint32_t)t * kEmitPer;
        uint32_t code[kEmitPer], nbv[kEmitPer];
        uint32_t mine = 0;
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (p0 < n) {
            raw = *reinterpret_cast<const uint4 *>(in + p0);      // streams are 16-byte aligned and padded
            const uint32_t valid = n - p0;
            if (valid < 16) {
                uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
                for (int i = 0; i < 4; ++i) {
                    const int keep = (int)valid - 4 * i;
                    if (keep <= 0) w[i] = 0;
                    else if (keep < 4) w[i] &= (1u << (8 * keep)) - 1u;
                }
                raw = make_uint4(w[0], w[1], w[2], w[3]);
            }
            // Adler-32 of the input, linear form: B += (n - pos) * byte
            uint32_t a = __builtin_amdgcn_sad_u8(raw.x, 0u, 0u);
            a = __builtin_amdgcn_sad_u8(raw.y, 0u, a);
            a = __builtin_amdgcn_sad_u8(raw.z, 0u, a);
            a = __builtin_amdgcn_sad_u8(raw.w, 0u, a);
            uint32_t wsum = __builtin_amdgcn_udot4(raw.x, 0x0D0E0F10u, 0u, false);
            wsum = __builtin_amdgcn_udot4(raw.y, 0x090A0B0Cu, wsum, false);
            wsum = __builtin_amdgcn_udot4(raw.z, 0x05060708u, wsum, false);
            wsum = __builtin_amdgcn_udot4(raw.w, 0x01020304u, wsum, false);
            const long long lead = (long long)n - (long long)p0 - 16;     // may be negative on the last piece
            const long long term = lead * (long long)a + (long long)wsum; // >= 0 as a whole
            accB = (accB + (unsigned long long)term) % kAdlerBase;
            accA += a;
        }
        const uint32_t rw[4] = {raw.x, raw.y, raw.z, raw.w};
        // this lane's 16 selectors as four dwordx4 loads (the stream's selector region is 16-byte aligned and the
        // workspace is padded): 16 single loads would each make the wave touch 64 different cache lines
        uint32_t sv[kEmitPer];
#pragma unroll
        for (int q = 0; q < kEmitPer / 4; ++q) {
            uint4 v4 = make_uint4(0, 0, 0, 0);
            if (p0 < n) v4 = reinterpret_cast<const uint4 *>(sel + p0)[q];
            sv[4 * q] = v4.x; sv[4 * q + 1] = v4.y; sv[4 * q + 2] = v4.z; sv[4 * q + 3] = v4.w;
        }
#pragma unroll
        for (int j = 0; j < kEmitPer; ++j) {
            const uint32_t p = p0 + (uint32_t)j;
            uint32_t s = p < n ? sv[j] : 0u;
            code[j] = 0;
            nbv[j] = 0;
            if (s & 0x80000000u) static_match(((s >> 16) & 0xffu) + 3u, (s & 0xffffu) + 1u, code[j], nbv[j]);
            else if (s & 0x40000000u) static_literal((rw[j >> 2] >> (8 * (j & 3))) & 0xffu, code[j], nbv[j]);
            mine += nbv[j];
        }
        // block exclusive scan of `mine`
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        for (int i = t; i < kEmitWords; i += 256) obuf[i] = 0;
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
        for (int j = 0; j < kEmitPer; ++j) {
            if (nbv[j]) {
                const uint32_t word = cur >> 5, sh = cur & 31u;
                atomicOr(&obuf[word], code[j] << sh);
                if (sh + nbv[j] > 32u) atomicOr(&obuf[word + 1], code[j] >> (32u - sh));
                cur += nbv[j];
            }
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

    // end-of-block code 256 = seven 0 bits (zng_emit_end_block, trees_emit.h:169-180), then pad to a byte
    cbits += 7;
    // reductions for Adler
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        accA += __shfl_xor(accA, m, 64);
        accB += __shfl_xor(accB, m, 64);
    }
    if (lane == 0) {
        red_a[wave] = accA;
        red_b[wave] = accB;
    }
    __syncthreads();
    if (t == 0) {
        uint8_t *outb = job.out;
        if (cbits >= 32) {
            outw[wbase++] = cw;
            cw = 0;
            cbits -= 32;
        }
        uint32_t bytes = wbase * 4u;
        for (uint32_t k = 0; k < (cbits + 7u) / 8u; ++k) outb[bytes++] = (uint8_t)(cw >> (8 * k));
        const unsigned long long A = (red_a[0] + red_a[1] + red_a[2] + red_a[3]) % kAdlerBase;
        const unsigned long long B = (red_b[0] + red_b[1] + red_b[2] + red_b[3]) % kAdlerBase;
        results[2 * blockIdx.x] = bytes;
        results[2 * blockIdx.x + 1] = (uint32_t)(((1 + A) % kAdlerBase) | ((((unsigned long long)n + B) % kAdlerBase) << 16));
    }
}

}  // namespace zr

using namespace zr;

extern "C" {

size_t zng_rocm_deflate_quick_bound(size_t source_len) {
    // 3 header bits + at most 9 bits per input byte (a match of >= 4 bytes costs <= 31 bits) + 7 EOB bits,
    // cf. DEFLATE_QUICK_OVERHEAD / DEFLATE_BLOCK_OVERHEAD (zutil.h:71-79, deflate.c:773-777); rounded so that
    // consecutive streams stay 16-byte aligned.
    size_t b = source_len + ((source_len + 7) >> 3) + 8;
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
    uint64_t total = 0;
    for (size_t i = 0; i < njobs; ++i) {
        const zng_rocm_stream_job &j = jobs[i];
        if ((j.in_len && (!j.in || !j.out)) || ((uintptr_t)j.in & 15) || ((uintptr_t)j.out & 3) ||
            j.out_cap < zng_rocm_deflate_quick_bound(j.in_len)) {
            set_error("job %zu: streams must be 16-byte aligned (in), 4-byte aligned (out) and out_cap >= "
                      "zng_rocm_deflate_quick_bound(in_len)", i);
            return ZNG_ROCM_EINVAL;
        }
        h_jobs[i] = StreamJobDev{j.in, j.out, j.in_len, j.out_cap, total};
        total += ((uint64_t)j.in_len + 3u) & ~3ull;           // selector regions stay 16-byte aligned
    }
    uint32_t *d_sel = nullptr;
    if (int rc = scratch_reserve(ws, kScrQuickSel, (total + 1024) * sizeof(uint32_t), false, (void **)&d_sel)) return rc;
    ZR_HIP(hipMemcpyAsync(d_jobs, h_jobs, njobs * sizeof(StreamJobDev), hipMemcpyHostToDevice, st));
    if (int rc = host_tables_release(ws, st)) return rc;
    ZR_LAUNCH_TRACED(lz_parse_kernel, dim3((unsigned)njobs), dim3(256), st, d_jobs, d_sel);
    ZR_HIP(hipGetLastError());
    hipLaunchKernelGGL(emit_static_kernel, dim3((unsigned)njobs), dim3(256), 0, st, d_jobs, d_sel, d_results);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

}  // extern "C"
