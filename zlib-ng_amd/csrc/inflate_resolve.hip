// inflate_resolve.hip -- device stage of the inflate path: turn the host decoder's token stream into
// plaintext in HBM.  This is the store half of zlib-ng's inflate_fast (inffast_tpl.h:155-171 literal
// stores, :228-279 match copies via CHUNKCOPY / CHUNKMEMSET, chunkset_tpl.h:24-227) re-designed for a
// machine that wants thousands of independent copy streams:
//
//   K1 segments  one wavefront per >=32 KiB output segment  walks its tokens in order and writes 16-bit
//                SYMBOLS: a literal byte, or 256 + k meaning "byte k of the 32 KiB that precede this
//                segment".  Copies inside the segment move symbols, so unresolved references
//                propagate by themselves; no segment waits for another.  The wave's last 4096 symbols
//                live in an LDS ring (the device form of inflate's sliding window, inflate.c:325-378).
//   K2 context   only the last 32 KiB of a segment can be named by the next one.  Those tails are resolved
//                against each other in two levels (groups of 32 in parallel, then one walk over the groups):
//                32 + nsegs / 32 dependent steps of 32 KiB instead of nsegs.
//   K3 translate all symbols -> bytes, fully parallel: 2N read + N written, HBM-bound.
//
// Algorithmic bytes: C + U per stream; the symbol detour adds 4U of HBM traffic.
#include "context.h"
#include "deflate_dev.h"

namespace zr {

constexpr long long kCtx = 32768;        // MAX_WBITS 15: a distance never exceeds this
constexpr size_t kLargeHostStream = 4u << 20;   // compressed bytes from which a host-resident stream is decoded on the device
int inflate_large_from_host(const uint8_t *src, size_t src_len, const uint8_t *d_window, uint32_t window_len, uint8_t *d_dst,
                            size_t dst_cap, uint64_t *out_len, size_t *in_used, hipStream_t st);      // inflate_large.hip
void inflate_large_forget_parts();       // the part counter of the calling thread back to 0 ("the sequential decoder did it")
int inflate_raw_window_sequential(const uint8_t *src, size_t src_len, const uint8_t *d_window, uint32_t window_len, uint8_t *d_dst,
                                  size_t dst_cap, uint64_t *out_len, size_t *in_used, hipStream_t st);

// K1.  One wavefront per segment, four per workgroup.  Per wave in LDS: a ring with the last 4096 symbols it
// produced (recent back-references never touch HBM, and a match no longer waits for the wave's own stores to
// be acknowledged), a 1 KiB window of upcoming literal bytes, and the tokens 64 at a time in registers.
// Symbols leave the ring in coalesced 2048-symbol flushes; a source below `flushed` is read back from HBM.
constexpr int kRing = 4096;               // symbols per wave (8 KiB)
constexpr int kFlush = 2048;
constexpr int kLitWin = 1024;

__global__ __launch_bounds__(256)
void inflate_segments_kernel(const uint32_t *__restrict__ tokens, const uint8_t *__restrict__ literals,
                             size_t nliterals, const uint64_t *__restrict__ segs, size_t nsegs,
                             uint16_t *__restrict__ sym) {
    __shared__ uint16_t ring_all[4][kRing];
    __shared__ uint8_t lit_all[4][kLitWin];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const size_t seg = (size_t)blockIdx.x * 4 + (size_t)w;
    if (seg >= nsegs) return;
    uint16_t *ring = ring_all[w];
    uint8_t *lit = lit_all[w];
    const uint64_t t0 = segs[3 * seg], t1 = segs[3 * seg + 3];
    const long long o0 = (long long)segs[3 * seg + 1];
    uint16_t *out = sym + o0;                // positions below are RELATIVE to the segment's first symbol (32 bit):
    uint32_t op = 0, flushed = 0;            // a segment holds < 33 KiB, a source reaches at most 32 KiB before it
    unsigned long long lp = segs[3 * seg + 2];
    unsigned long long lit_base = lp;

    auto stage_literals = [&]() {            // lit[] <- literals[lit_base .. lit_base + kLitWin)
        const unsigned long long at = lit_base + 16ull * (unsigned long long)lane;
        if (at + 16 <= nliterals) {
            const u32x4_unaligned v = load_u128(literals + at);           // one unaligned dwordx4 per lane
            *reinterpret_cast<uint4 *>(lit + 16 * lane) = make_uint4(v.x, v.y, v.z, v.w);
        } else {
            for (int k = 0; k < 16; ++k) {
                const unsigned long long q = at + (unsigned long long)k;
                lit[16 * lane + k] = q < nliterals ? literals[q] : 0;
            }
        }
    };
    auto flush = [&](uint32_t upto) {        // symbols [flushed, upto) ring -> HBM, coalesced
        for (uint32_t i = flushed + (uint32_t)lane; i < upto; i += 64) out[i] = ring[i & (kRing - 1)];
        flushed = upto;
        // a later match may read these symbols back from HBM: have the stores acknowledged first
        // (once per 2048 symbols, not once per token)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0);
    };
    // symbol at relative position rp (may be negative: the 32 KiB before the segment)
    auto fetch = [&](int rp) -> uint16_t {
        if (rp >= (int)flushed) return ring[(uint32_t)rp & (kRing - 1)];
        if (rp >= 0) return out[rp];
        return (uint16_t)(256 + (int)kCtx + rp);
    };

    stage_literals();
    for (uint64_t base = t0; base < t1; base += 64) {
        const uint32_t mytok = base + (uint64_t)lane < t1 ? tokens[base + lane] : 0u;
        const int cnt = (int)((t1 - base) < 64 ? (t1 - base) : 64);
        for (int j = 0; j < cnt; ++j) {
            const uint32_t tok = (uint32_t)__builtin_amdgcn_readlane((int)mytok, j);
            if (op - flushed >= kFlush) flush(flushed + kFlush);
            if (!(tok >> 31)) {
                uint32_t n = tok;
                while (n) {
                    if (op - flushed >= kFlush) flush(flushed + kFlush);
                    uint32_t avail = (uint32_t)(lit_base + kLitWin - lp);
                    if (avail == 0) {
                        lit_base = lp;
                        stage_literals();
                        avail = kLitWin;
                    }
                    uint32_t chunk = n < avail ? n : avail;
                    if (chunk > kFlush) chunk = kFlush;           // the ring keeps kRing - kFlush unflushed symbols
                    const uint32_t lo = (uint32_t)(lp - lit_base);
                    for (uint32_t i = (uint32_t)lane; i < chunk; i += 64)
                        ring[(op + i) & (kRing - 1)] = lit[lo + i];
                    op += chunk;
                    lp += chunk;
                    n -= chunk;
                }
            } else {
                // len, dist and op are wave-uniform: the shape of the copy is decided by scalar branches.  Every
                // source symbol of a match was produced before the token started (with dist < len the sources are
                // the `dist` symbols before op, replicated), so the rounds of a copy are independent of each other.
                const uint32_t len = ((tok >> 16) & 0xffu) + 3u;
                const uint32_t dist = (tok & 0xffffu) + 1u;
                const int src = (int)op - (int)dist;
                if (dist >= len) {
                    if (src >= (int)flushed) {                    // the common case: ring to ring
                        for (uint32_t i = (uint32_t)lane; i < len; i += 64)
                            ring[(op + i) & (kRing - 1)] = ring[((uint32_t)src + i) & (kRing - 1)];
                    } else {
                        for (uint32_t i = (uint32_t)lane; i < len; i += 64)
                            ring[(op + i) & (kRing - 1)] = fetch(src + (int)i);
                    }
                } else if (dist == 1u) {                          // a run of one symbol
                    const uint16_t v = fetch(src);
                    for (uint32_t i = (uint32_t)lane; i < len; i += 64) ring[(op + i) & (kRing - 1)] = v;
                } else {
                    for (uint32_t i = (uint32_t)lane; i < len; i += 64)
                        ring[(op + i) & (kRing - 1)] = fetch(src + (int)(i % dist));
                }
                op += len;
            }
        }
    }
    flush(op);
}

// K2.  Only the last 32 KiB of a segment (its TAIL) can be named by the next segment, and the tail of segment
// s names the tail of segment s-1 and nothing else (every segment but the last holds >= 32 KiB).  A straight
// walk over the tails is nsegs dependent steps; two levels make it kGroup + nsegs / kGroup:
//   pass A  one workgroup per group of kGroup segments rewrites, in order, each tail of its group against the
//           tail before it.  Afterwards a tail's remaining references (>= 256) all name the tail that
//           precedes the GROUP (its base), whichever segment of the group they sit in.
//   pass B  one workgroup walks the groups in order and resolves the last tail of each against its base;
//           those are then plain bytes.
// K3 finishes the rest: a tail symbol needs one lookup in its group's base, any other symbol one lookup in the
// previous segment's tail plus, when that is still a reference, one in that tail's base.
// Each lane owns 32 symbols of the 32 KiB.
constexpr size_t kGroup = 64;

// A run of tail steps in which every step's `prev` is the step before's `cur` (both passes are such runs): the tail just
// rewritten stays in LDS (two buffers of 64 KiB, one barrier per step) and the next tail to rewrite is already on its way
// from HBM while this one is resolved -- a step is an LDS gather instead of three dependent trips to memory
// (the 6500 segments of a 256 MiB stream: pass A 0.69 -> 0.33 ms, pass B 0.21 -> 0.18 ms).  step(i) = symbol index where the i-th tail ENDS.
template <typename EndOf>
__device__ __forceinline__ void tail_run(uint16_t *__restrict__ sym, long long prev_end, size_t nsteps, EndOf end_of) {
    constexpr int PER = (int)(kCtx / 1024);            // 32 symbols per lane: symbol threadIdx.x + 1024 * k of the tail
    __shared__ uint16_t tails[2][kCtx];
    if (!nsteps) return;
    uint16_t v[PER], nx[PER];
    {
        const uint16_t *prev = sym + (prev_end - kCtx);
#pragma unroll
        for (int k = 0; k < PER; ++k) nx[k] = prev[threadIdx.x + 1024 * k];
        const uint16_t *cur = sym + (end_of(0) - kCtx);
#pragma unroll
        for (int k = 0; k < PER; ++k) v[k] = cur[threadIdx.x + 1024 * k];
#pragma unroll
        for (int k = 0; k < PER; ++k) tails[0][threadIdx.x + 1024 * k] = nx[k];
    }
    __syncthreads();
    int b = 0;
    for (size_t i = 0; i < nsteps; ++i) {
        uint16_t *cur = sym + (end_of(i) - kCtx);
        if (i + 1 < nsteps) {
            const uint16_t *next = sym + (end_of(i + 1) - kCtx);
#pragma unroll
            for (int k = 0; k < PER; ++k) nx[k] = next[threadIdx.x + 1024 * k];
        }
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (v[k] >= 256) v[k] = tails[b][v[k] - 256];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            cur[threadIdx.x + 1024 * k] = v[k];
            tails[b ^ 1][threadIdx.x + 1024 * k] = v[k];
        }
        __syncthreads();
        b ^= 1;
#pragma unroll
        for (int k = 0; k < PER; ++k) v[k] = nx[k];
    }
}

// grid.x = group.  Tails exist for segments 0 .. nsegs-2.  With a prior window (the 32768 symbols in front of
// sym[0] hold it) the first tail is resolved against that window first, so that group 0 ends in plain bytes.
__global__ __launch_bounds__(1024)
void inflate_context_group_kernel(const uint64_t *__restrict__ segs, size_t nsegs, uint16_t *__restrict__ sym,
                                  int has_window) {
    const size_t first = (size_t)blockIdx.x * kGroup;
    size_t last = first + kGroup;
    if (last > nsegs - 1) last = nsegs - 1;
    // tails first+1 .. last-1, each against the one before it; group 0 with a window starts one earlier: tail 0 against the window
    const size_t s0 = (has_window && blockIdx.x == 0) ? first : first + 1;
    if (s0 >= last) return;
    tail_run(sym, (long long)segs[3 * s0 + 1], last - s0, [&](size_t i) { return (long long)segs[3 * (s0 + i) + 4]; });
}

// one workgroup; group g >= 1: its last tail against the last tail of group g-1
__global__ __launch_bounds__(1024)
void inflate_context_chain_kernel(const uint64_t *__restrict__ segs, size_t nsegs, uint16_t *__restrict__ sym) {
    if (nsegs - 1 <= kGroup) return;
    const size_t ngroups = (nsegs - 1 + kGroup - 1) / kGroup;
    tail_run(sym, (long long)segs[3 * kGroup + 1], ngroups - 1, [&](size_t i) {
        size_t last = (i + 2) * kGroup;
        if (last > nsegs - 1) last = nsegs - 1;
        return (long long)segs[3 * (last - 1) + 4];
    });
}

// the symbols in front of sym[0]: the caller's window, right-aligned to the stream's first byte (zeros before it)
__global__ __launch_bounds__(256)
void inflate_window_kernel(const uint8_t *__restrict__ window, uint32_t window_len, uint16_t *__restrict__ sym) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;            // 0 .. 32767
    const uint32_t gap = (uint32_t)kCtx - window_len;
    sym[(long long)k - kCtx] = k >= gap ? (uint16_t)window[k - gap] : (uint16_t)0;
}

// the same for a BATCH of streams laid out one behind the other in symbol space, each behind its own 32768-symbol
// window gap (see resolve_batch below): blockIdx.y = stream
struct BatchStream {
    uint64_t       v_start;     // symbol index of the stream's first byte
    const uint8_t *d_window;    // its window_len bytes of history (device) or null
    uint64_t       window_len;
};
__global__ __launch_bounds__(256)
void inflate_windows_kernel(const BatchStream *__restrict__ streams, uint16_t *__restrict__ sym) {
    const BatchStream b = streams[blockIdx.y];
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;            // 0 .. 32767
    const uint32_t gap = (uint32_t)kCtx - (uint32_t)b.window_len;
    sym[(long long)b.v_start - kCtx + k] = k >= gap ? (uint16_t)b.d_window[k - gap] : (uint16_t)0;
}

// K3.  grid.x = segment.  seg_dst / seg_end (batches only): where the segment's bytes go (the streams of a batch have
// separate destinations) and where its real bytes end (a stream's last segment runs on through the next window gap).
__global__ __launch_bounds__(1024)
void inflate_translate_kernel(const uint64_t *__restrict__ segs, size_t nsegs, const uint16_t *__restrict__ sym,
                              uint8_t *__restrict__ out, const uint64_t *__restrict__ seg_dst,
                              const uint64_t *__restrict__ seg_end) {
    const size_t s = blockIdx.x;
    const long long o0 = (long long)segs[3 * s + 1], o1 = (long long)segs[3 * s + 4];
    const long long stop = seg_end ? (long long)seg_end[s] : o1;
    if (seg_dst) out = reinterpret_cast<uint8_t *>(seg_dst[s]) - o0;
    const long long tail = s + 1 < nsegs ? o1 - kCtx : o1;          // first symbol pass A/B rewrote
    const uint16_t *prev = sym + (o0 - kCtx);                       // tail of segment s-1
    const size_t gs = (s / kGroup) * kGroup;                        // base of this segment's group ...
    const uint16_t *base_own = sym + ((long long)segs[3 * gs + 1] - kCtx);
    const size_t gp = s ? ((s - 1) / kGroup) * kGroup : 0;          // ... and of the previous segment's
    const uint16_t *base_prev = sym + ((long long)segs[3 * gp + 1] - kCtx);
    const long long prev_off = prev - sym, own_off = base_own - sym, bprev_off = base_prev - sym;   // the same as offsets into sym
    auto resolve = [&](uint32_t v, long long i) -> uint32_t {
        if (v >= 256u) {
            if (i >= tail) {
                v = base_own[v - 256u];
            } else {
                v = prev[v - 256u];
                if (v >= 256u) v = base_prev[v - 256u];
            }
        }
        return v;
    };
    // eight symbols per lane and step where the symbol array allows a 16-byte load: one 8-byte store instead of eight
    // byte stores (round 2's form moved 64 bytes per wave store; with the large-stream path's segments of 100+ KiB the
    // kernel took 5 ms per 256 MiB)
    const long long head = (8 - ((o0 + ((long long)((uintptr_t)sym >> 1) & 7)) & 7)) & 7;      // symbols up to 16-byte alignment of sym + i
    const long long a0 = o0 + head < stop ? o0 + head : stop;
    for (long long i = o0 + threadIdx.x; i < a0; i += blockDim.x)
        __builtin_nontemporal_store((uint8_t)resolve(sym[i], i), out + i);
    const long long chunks = (stop - a0) >> 3;
    for (long long c = threadIdx.x; c < chunks; c += blockDim.x) {
        const long long i = a0 + 8 * c;
        const uint4 q = *reinterpret_cast<const uint4 *>(sym + i);
        uint32_t lo, hi;
        // References come in stretches (behind a part's start) and most wavefronts see none: those pack the low bytes of
        // their eight symbols with two byte permutes (unpacking, testing and packing symbol by symbol was ~150 vector
        // instructions per 8 symbols and the kernel ran at 1.7 TB/s whatever the data).
        if (!__any(((q.x | q.y | q.z | q.w) & 0xff00ff00u) != 0u)) {
            lo = __builtin_amdgcn_perm(q.y, q.x, 0x06040200u);
            hi = __builtin_amdgcn_perm(q.w, q.z, 0x06040200u);
        } else {
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (w[k >> 1] >> (16 * (k & 1))) & 0xffffu;
            // the lookups of all eight symbols are issued TOGETHER, unconditionally: inside their `if`s they were eight
            // dependent trips to L2 one after the other
            uint16_t g[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const long long tab = i + k >= tail ? own_off : prev_off;      // (a table may not exist where nothing names it:
                g[k] = sym[v[k] >= 256u ? tab + (long long)(v[k] - 256u) : i + k];  //  a symbol that needs no lookup reads itself again)
            }
            bool again = false;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (v[k] >= 256u) {
                    v[k] = g[k];
                    again |= i + k < tail && v[k] >= 256u;
                }
            }
            if (__any(again)) {
#pragma unroll
                for (int k = 0; k < 8; ++k) g[k] = sym[(i + k < tail && v[k] >= 256u) ? bprev_off + (long long)(v[k] - 256u) : i + k];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (i + k < tail && v[k] >= 256u) v[k] = g[k];
            }
            lo = hi = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k < 4) lo |= (v[k] & 0xffu) << (8 * k);
                else hi |= (v[k] & 0xffu) << (8 * (k - 4));
            }
        }
        uint8_t *d = out + i;
        if ((((uintptr_t)d) & 7u) == 0) {
            __builtin_nontemporal_store(((unsigned long long)hi << 32) | lo, reinterpret_cast<unsigned long long *>(d));
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] = (uint8_t)((k < 4 ? lo >> (8 * k) : hi >> (8 * (k - 4))));
        }
    }
    for (long long i = a0 + 8 * chunks + threadIdx.x; i < stop; i += blockDim.x)
        __builtin_nontemporal_store((uint8_t)resolve(sym[i], i), out + i);
}

// Device stage for a batch of independent streams in ONE set of launches (inflate_many.hip).  The streams sit one
// behind the other in symbol space, stream k at v_start_k = v_start_(k-1) + len_(k-1) + 32768: the 32768 symbols in
// front of each are its window (the caller's history, or zeros), and the LAST segment of every stream is extended
// through the gap that follows it, so that every segment but the batch's last still holds >= 32 KiB and the context
// chain runs over the whole batch as over one stream -- a gap holds plain bytes only, so nothing crosses it.
//   d_segs: (nsegs + 1) triples with batch-wide token / symbol / literal positions
//   d_seg_dst / d_seg_end: per segment, destination address of its first byte and symbol index where its bytes end
int inflate_resolve_batch(const uint32_t *d_tokens, const uint8_t *d_literals, size_t nliterals, const uint64_t *d_segs,
                          size_t nsegs, uint16_t *sym, const uint64_t *d_seg_dst, const uint64_t *d_seg_end,
                          const void *d_streams, size_t nstreams, hipStream_t st) {
    if (!nsegs || !nstreams) return ZNG_ROCM_OK;
    hipLaunchKernelGGL(inflate_windows_kernel, dim3((unsigned)(kCtx / 256), (unsigned)nstreams), dim3(256), 0, st,
                       (const BatchStream *)d_streams, sym);
    ZR_HIP(hipGetLastError());
    ZR_LAUNCH_TRACED(inflate_segments_kernel, dim3((unsigned)((nsegs + 3) / 4)), dim3(256), st, d_tokens, d_literals,
                     nliterals, d_segs, nsegs, sym);
    ZR_HIP(hipGetLastError());
    if (nsegs > 1) {
        const unsigned ngroups = (unsigned)((nsegs - 1 + kGroup - 1) / kGroup);
        hipLaunchKernelGGL(inflate_context_group_kernel, dim3(ngroups), dim3(1024), 0, st, d_segs, nsegs, sym, 1);
        ZR_HIP(hipGetLastError());
        if (ngroups > 1) {
            hipLaunchKernelGGL(inflate_context_chain_kernel, dim3(1), dim3(1024), 0, st, d_segs, nsegs, sym);
            ZR_HIP(hipGetLastError());
        }
    }
    hipLaunchKernelGGL(inflate_translate_kernel, dim3((unsigned)nsegs), dim3(1024), 0, st, d_segs, nsegs, sym,
                       (uint8_t *)nullptr, d_seg_dst, d_seg_end);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

// symbols that are already in place (inflate_large.hip) -> bytes: the window in front, the context chain, the translation
int inflate_resolve_symbols(const uint64_t *d_segs, size_t nsegs, uint16_t *sym, uint8_t *d_out, const uint8_t *d_window,
                            uint32_t window_len, hipStream_t st) {
    if (!nsegs) return ZNG_ROCM_OK;
    hipLaunchKernelGGL(inflate_window_kernel, dim3((unsigned)(kCtx / 256)), dim3(256), 0, st, d_window, window_len, sym);
    ZR_HIP(hipGetLastError());
    if (nsegs > 1) {
        const unsigned ngroups = (unsigned)((nsegs - 1 + kGroup - 1) / kGroup);
        hipLaunchKernelGGL(inflate_context_group_kernel, dim3(ngroups), dim3(1024), 0, st, d_segs, nsegs, sym, 1);
        ZR_HIP(hipGetLastError());
        if (ngroups > 1) {
            hipLaunchKernelGGL(inflate_context_chain_kernel, dim3(1), dim3(1024), 0, st, d_segs, nsegs, sym);
            ZR_HIP(hipGetLastError());
        }
    }
    ZR_LAUNCH_TRACED(inflate_translate_kernel, dim3((unsigned)nsegs), dim3(1024), st, d_segs, nsegs, sym, d_out,
                     (const uint64_t *)nullptr, (const uint64_t *)nullptr);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

}  // namespace zr

using namespace zr;

extern "C" {

static int resolve_impl(const uint32_t *d_tokens, const uint8_t *d_literals, size_t nliterals, const uint64_t *d_segs,
                        size_t nsegs, uint16_t *sym, uint8_t *d_out, uint64_t out_len, const uint8_t *d_window,
                        uint32_t window_len, bool has_window, hipStream_t st) {
    if (out_len == 0 || nsegs == 0) return ZNG_ROCM_OK;
    if (!d_tokens || !d_segs || !sym || !d_out) return ZNG_ROCM_EINVAL;
    if (has_window) {
        if (window_len > (uint32_t)kCtx || (window_len && !d_window)) return ZNG_ROCM_EINVAL;
        hipLaunchKernelGGL(inflate_window_kernel, dim3((unsigned)(kCtx / 256)), dim3(256), 0, st, d_window, window_len, sym);
        ZR_HIP(hipGetLastError());
    }
    ZR_LAUNCH_TRACED(inflate_segments_kernel, dim3((unsigned)((nsegs + 3) / 4)), dim3(256), st, d_tokens, d_literals,
                     nliterals, d_segs, nsegs, sym);
    ZR_HIP(hipGetLastError());
    if (nsegs > 1) {
        const unsigned ngroups = (unsigned)((nsegs - 1 + kGroup - 1) / kGroup);
        hipLaunchKernelGGL(inflate_context_group_kernel, dim3(ngroups), dim3(1024), 0, st, d_segs, nsegs, sym,
                           has_window ? 1 : 0);
        ZR_HIP(hipGetLastError());
        if (ngroups > 1) {
            hipLaunchKernelGGL(inflate_context_chain_kernel, dim3(1), dim3(1024), 0, st, d_segs, nsegs, sym);
            ZR_HIP(hipGetLastError());
        }
    }
    hipLaunchKernelGGL(inflate_translate_kernel, dim3((unsigned)nsegs), dim3(1024), 0, st, d_segs, nsegs, sym, d_out,
                       (const uint64_t *)nullptr, (const uint64_t *)nullptr);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

int zng_rocm_inflate_resolve_dev(const uint32_t *d_tokens, size_t ntokens, const uint8_t *d_literals,
                                 size_t nliterals, const uint64_t *d_segs, size_t nsegs, uint16_t *d_symbols,
                                 uint8_t *d_out, uint64_t out_len, void *stream) {
    if (!ctx()) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    DeviceGuard dev;
    (void)ntokens;
    return resolve_impl(d_tokens, d_literals, nliterals, d_segs, nsegs, d_symbols, d_out, out_len, nullptr, 0, false,
                        (hipStream_t)stream);
}

int zng_rocm_inflate_resolve_window_dev(const uint32_t *d_tokens, size_t ntokens, const uint8_t *d_literals,
                                        size_t nliterals, const uint64_t *d_segs, size_t nsegs, uint16_t *d_symbols,
                                        uint8_t *d_out, uint64_t out_len, const uint8_t *d_window,
                                        uint32_t window_len, void *stream) {
    if (!ctx()) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    DeviceGuard dev;
    (void)ntokens;
    if (!d_symbols) return ZNG_ROCM_EINVAL;
    return resolve_impl(d_tokens, d_literals, nliterals, d_segs, nsegs, d_symbols + kCtx, d_out, out_len, d_window,
                        window_len, true, (hipStream_t)stream);
}

}  // extern "C"

namespace zr {
// host token stream -> plaintext at d_dst (tk->out_len bytes): uploads tokens | segs | literals, resolves against the
// optional device-resident window, synchronises the stream (the staging buffers are released behind it)
int inflate_tokens_to_device(const zng_rocm_inflate_tokens *tkp, const uint8_t *d_window, uint32_t window_len,
                             uint8_t *d_dst, hipStream_t st) {
    const zng_rocm_inflate_tokens &tk = *tkp;
    int rc = ZNG_ROCM_OK;
    if (tk.out_len) {
        // one device allocation: tokens | segs | literals | symbols
        const size_t tok_b = (tk.ntokens * 4 + 255) & ~(size_t)255;
        const size_t seg_b = ((tk.nsegs + 1) * 24 + 255) & ~(size_t)255;
        const size_t lit_b = (tk.nliterals + 255) & ~(size_t)255;
        const size_t sym_b = ((size_t)tk.out_len + (size_t)kCtx) * 2;          // + the window's symbols in front
        uint8_t *d = nullptr;
        if (hipMalloc(&d, tok_b + seg_b + lit_b + sym_b) != hipSuccess) {
            set_error("device allocation for the token stream failed");
            return ZNG_ROCM_ENOMEM;
        }
        hipError_t e = hipMemcpyAsync(d, tk.tokens, tk.ntokens * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(d + tok_b, tk.segs, (tk.nsegs + 1) * 24, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && tk.nliterals)
            e = hipMemcpyAsync(d + tok_b + seg_b, tk.literals, tk.nliterals, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) {
            set_error("H2D of the token stream failed: %s", hipGetErrorString(e));
            rc = ZNG_ROCM_EHIP;
        } else {
            rc = zng_rocm_inflate_resolve_window_dev((const uint32_t *)d, tk.ntokens, d + tok_b + seg_b, tk.nliterals,
                                                     (const uint64_t *)(d + tok_b), tk.nsegs,
                                                     (uint16_t *)(d + tok_b + seg_b + lit_b), d_dst, tk.out_len,
                                                     d_window, window_len, st);
        }
        // the host buffers and the device workspace are released after the stream drained
        if (hipStreamSynchronize(st) != hipSuccess && rc == ZNG_ROCM_OK) {
            set_error("stream synchronize failed");
            rc = ZNG_ROCM_EHIP;
        }
        (void)hipFree(d);
    }
    return rc;
}
}  // namespace zr

extern "C" {

int zng_rocm_inflate_raw_window(const uint8_t *src, size_t src_len, const uint8_t *d_window, uint32_t window_len,
                                uint8_t *d_dst, size_t dst_cap, uint64_t *out_len, size_t *in_used, void *stream);

int zng_rocm_inflate_raw_ex(const uint8_t *src, size_t src_len, uint8_t *d_dst, size_t dst_cap, uint64_t *out_len,
                            size_t *in_used, void *stream);

int zng_rocm_inflate_raw(const uint8_t *src, size_t src_len, uint8_t *d_dst, size_t dst_cap, uint64_t *out_len,
                         void *stream) {
    return zng_rocm_inflate_raw_ex(src, src_len, d_dst, dst_cap, out_len, nullptr, stream);
}

// same, also reporting how many input bytes the stream occupied (the framing front ends find the trailer there)
int zng_rocm_inflate_raw_ex(const uint8_t *src, size_t src_len, uint8_t *d_dst, size_t dst_cap, uint64_t *out_len,
                            size_t *in_used, void *stream) {
    return zng_rocm_inflate_raw_window(src, src_len, nullptr, 0, d_dst, dst_cap, out_len, in_used, stream);
}

// ... and with `window_len` bytes of history (device resident) in front of the stream: a preset dictionary
// (inflateSetDictionary on a raw stream, inflate.c:1214-1261) or the tail of what an earlier call produced
int zng_rocm_inflate_raw_window(const uint8_t *src, size_t src_len, const uint8_t *d_window, uint32_t window_len,
                                uint8_t *d_dst, size_t dst_cap, uint64_t *out_len, size_t *in_used, void *stream) {
    if (!ctx()) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    DeviceGuard dev;
    if (window_len > (uint32_t)kCtx || (window_len && !d_window)) return ZNG_ROCM_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    inflate_large_forget_parts();
    // a large stream goes up whole and is decoded on the device (inflate_large.hip: ~20 GB/s of output against 0.8 on
    // this thread); anything but a clean end of stream or a too-small destination is decoded again by the sequential
    // decoder, so status, message and byte counts are its
    if (src_len >= kLargeHostStream) {
        const int rc = inflate_large_from_host(src, src_len, d_window, window_len, d_dst, dst_cap, out_len, in_used, st);
        if (rc != 0) return rc;
    }
    return inflate_raw_window_sequential(src, src_len, d_window, window_len, d_dst, dst_cap, out_len, in_used, st);
}

}  // extern "C"

namespace zr {

// the sequential decoder on this thread + the device resolve: what every irregular stream ends up in
int inflate_raw_window_sequential(const uint8_t *src, size_t src_len, const uint8_t *d_window, uint32_t window_len, uint8_t *d_dst,
                                  size_t dst_cap, uint64_t *out_len, size_t *in_used, hipStream_t st) {
    zng_rocm_inflate_tokens tk;
    int status = zng_rocm_inflate_tokens_decode_window(src, src_len, window_len, &tk);
    if (out_len) *out_len = tk.out_len;
    if (in_used) *in_used = tk.in_used;
    if (status == -4) {
        zng_rocm_inflate_tokens_free(&tk);
        return ZNG_ROCM_ENOMEM;
    }
    if (tk.out_len > dst_cap) {
        set_error("inflate output (%llu bytes) exceeds dst_cap", (unsigned long long)tk.out_len);
        zng_rocm_inflate_tokens_free(&tk);
        return -5;
    }
    if (status < 0) set_error("%s", tk.msg);
    const int rc = inflate_tokens_to_device(&tk, d_window, window_len, d_dst, st);
    zng_rocm_inflate_tokens_free(&tk);
    return rc != ZNG_ROCM_OK ? rc : status;
}

}  // namespace zr
