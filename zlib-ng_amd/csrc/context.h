// context.h -- process-global device context of the arch/rocm backend:
// constant tables in HBM, per-stream workspaces and scratch, bounded staging, error state.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stddef.h>

#include <mutex>

#include "../../include/zng_rocm.h"
#include "gf2.h"

namespace zr {

// ---- launch geometry of the streaming checksum kernels -------------------
constexpr int kWgThreads  = 1024;                 // 16 wave64 per workgroup, one workgroup per CU
constexpr int kPieceBytes = 16;                   // one dwordx4 load per lane
constexpr int kUnitBytes  = kWgThreads * kPieceBytes;   // 16 KiB: what one workgroup consumes per step
constexpr int kMaxGroups  = 1024;                 // upper bound on workgroups (partials array size)

// Constant tables, built once on the host at init and kept in HBM.
struct DeviceTables {
    uint32_t byte_tab[256];              // crc_table: byte b -> b(x) * x^32   (crc32_braid_tbl.h:8 equivalent)
    uint32_t stride_tab[4][256];         // byte k of a word -> advanced by one unit stride: x^(8*(kUnitBytes+3-k))
                                         //   (same construction as tools/makecrct.c:99-112 with n*w = kUnitBytes)
    uint32_t lane_weight[kWgThreads][4]; // word c of lane t -> x^(8*(kUnitBytes - 16t - 4c)): distance to unit end
    uint32_t pow_tab[kPowDigits * 128];  // x^(8 * digit * 128^i), see gf2.h xpow_bytes
    uint32_t unit_pow[2][1024];          // x^(8 * kUnitBytes * d * 1024^i): weight of "d units later", two 10-bit digits
    uint32_t lane_pow[32][kWgThreads];   // lane_weight[t][3] * x^k, k = 0..31 (k-major: lane-consecutive loads): the 32
                                         //   partial products of the epilogue's per-lane GF(2) multiply, precomputed
    uint32_t x32_tab[4][256];            // byte k of a word -> times x^32 (the fold of one braid into the next, one
                                         //   LDS latency instead of four dependent byte-table steps)
    uint32_t tree_pp[1024][4];           // two-level form of the lane weights, as the 16-byte rows the epilogue stashes in LDS:
                                         //   rows 0..511   [kq * 64 + l]: (weight of lane l of the LAST wave) * x^(4kq+j), j = 0..3
                                         //   rows 512..639 [kq * 16 + w]: x^(8 * 1024 * (15 - w)) * x^(4kq+j): wave w -> last wave
                                         //   rows 640..1023: zero (every thread loads one row, no branch around the load)
};

struct Partial {            // one per workgroup, written by the streaming kernel
    uint32_t crc;           // register contribution relative to the end of the group's span
    uint32_t a;             // adler A contribution (mod 65521)
    uint32_t b;             // adler B contribution, already weighted to the end of the message (mod 65521)
    uint32_t pad;
};

// Growable per-stream scratch buffers of the stream-level entry points.  They are keyed by HIP stream: work on one
// stream is ordered, so a buffer is never under a running kernel of ANOTHER caller (SURVEY.md 8b "Threading":
// distinct zlib-ng streams are used concurrently from different host threads).
enum ScratchSlot {
    kScrQuickJobs = 0,      // device: StreamJobDev[] of zng_rocm_deflate_quick_dev
    kScrQuickJobsHost,      // pinned host mirror of the same (async H2D source)
    kScrDynJobs,            // device: SegJob[] of zng_rocm_deflate_dev
    kScrDynJobsHost,        // pinned
    kScrDynSel,             // device: one 32-bit selector per input byte (level-6 class)
    kScrDynSlots,           // device: per-segment output slots
    kScrDynSegLen,          // device: u32 per segment
    kScrDynSegLenHost,      // pinned
    kScrDynDstOff,          // device: u64 per segment
    kScrDynDstOffHost,      // pinned
    kScrChunkList,          // device: work list of zng_rocm_chunkmemset_safe_dev (long / memmove-order copies)
    kScrInflate,            // device: tokens | segs | literals | symbols of the one-shot inflate
    kScrInflateHost,        // pinned token staging of the batched inflate
    kScrInflateDevJobs,     // device: InflateJobDev[] of zng_rocm_inflate_streams_dev
    kScrInflateDevJobsHost, // pinned
    kScrCheckMessages,      // device: StreamArgs[] | FinalArgs[] of zng_rocm_checksums_dev
    kScrCheckMessagesHost,  // pinned
    kScrCheckPartials,      // device: Partial[] (messages x workgroups per message)
    kScrFrameWords,         // device: per-stream words of the many-stream framing entry points (framing_dev.hip)
    kScrFrameJobs,          // device: FrameJob[]
    kScrFrameJobsHost,      // pinned
    kScrLargeCand,          // device: candidate block starts of zng_rocm_inflate_large_dev
    kScrLargeParts,         // device: jobs | starts | results | symbol slots of the parts
    kScrLargeRetry,         // device: worst-case slots of the parts that overflowed theirs
    kScrLargeSym,           // device: segs | copies | the stream's symbol array
    kScrLargeSrc,           // device: the compressed bytes of a large HOST-resident stream (zng_rocm_inflate_raw*)
    kScrLargeCandHost,      // pinned: the finder's counts and the first candidates
    kScrLargePartsHost,     // pinned: jobs | starts | results of the parts
    kScrCount
};

struct Scratch {
    void  *p;
    size_t cap;
    bool   host;            // pinned host memory (hipHostMalloc) instead of device memory
};

struct Workspace {          // one per HIP stream
    Partial  *partials;     // kMaxGroups entries
    uint32_t *acc;          // spare device words (zero between calls)
    uint32_t *result;       // 2 x u32 scratch result (device) + 2 x u32 chained seed
    uint32_t *pinned;       // 2 x u32 host-pinned mirror
    uint8_t  *stage;        // device staging chunk(s) for host-pointer slots, bounded (kStageChunk)
    size_t    stage_bytes;
    Scratch   scratch[kScrCount];
    hipEvent_t host_done;   // recorded behind the last async copy that READS a pinned scratch buffer
    bool       host_busy;   // such a copy may still be in flight: synchronise host_done before rewriting
    std::mutex mu;          // host-side use of this workspace (growth, pinned tables): one enqueue at a time per stream
};

struct Context {
    uint64_t      generation;   // distinguishes this context from one created after a zng_rocm_shutdown()
    int           device;
    int           cus;
    int           lds_bytes;
    int           xcds;
    DeviceTables *tables;   // device pointer
    DeviceTables  host_tables;
};

// error plumbing -----------------------------------------------------------
void set_error(const char *fmt, ...);
#define ZR_HIP(call)                                                            \
    do {                                                                        \
        hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess) {                                                 \
            zr::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return ZNG_ROCM_EHIP;                                               \
        }                                                                       \
    } while (0)

// measurement hooks (zng_rocm_trace_begin/_end): when this launch of a dominant kernel is to be timed, hands out the
// event pair to ATTACH to its dispatch (hipExtLaunchKernelGGL: the kernel's own start / stop timestamps, what
// rocprofv3 --kernel-trace reports, with no extra packets in the stream); otherwise leaves both null
void trace_pick(hipEvent_t *start, hipEvent_t *stop);
#define ZR_LAUNCH_TRACED(kernel, grid, block, stream, ...)                                          \
    do {                                                                                            \
        hipEvent_t ev0_ = nullptr, ev1_ = nullptr;                                                  \
        zr::trace_pick(&ev0_, &ev1_);                                                               \
        hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, ev0_, ev1_, 0, __VA_ARGS__);          \
    } while (0)

Context   *ctx();                          // nullptr until zng_rocm_init succeeded
Workspace *workspace_for(hipStream_t s);   // lazily created on the context's device, nullptr on failure
int        ensure_stage(Workspace *ws, size_t bytes);
// scratch slot `slot` of the workspace with at least `bytes` (contents are NOT preserved when it grows)
int        scratch_reserve(Workspace *ws, int slot, size_t bytes, bool pinned_host, void **out);
// pinned tables are rewritten by the host: wait until the last async copy that read them has finished
int        host_tables_acquire(Workspace *ws);
int        host_tables_release(Workspace *ws, hipStream_t s);      // record host_done behind the copies just issued
[[noreturn]] void die(const char *what);   // loud failure for slots without an error channel

// HIP's current device is per host thread (default 0): every entry point that allocates or launches runs under one
// of these, so that a call from a thread that never selected the context's device still lands on it.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    DeviceGuard();
    ~DeviceGuard();
};

void checksum_reset_reserved_cus();        // checksum.hip
void inflate_pool_shutdown();              // inflate_many.hip: pooled worker resources
// inflate_resolve.hip: the device stage for a batch of streams in one set of launches (layout: see there);
// d_streams = array of {u64 v_start, const u8 *d_window, u64 window_len}
int inflate_resolve_batch(const uint32_t *d_tokens, const uint8_t *d_literals, size_t nliterals, const uint64_t *d_segs,
                          size_t nsegs, uint16_t *sym, const uint64_t *d_seg_dst, const uint64_t *d_seg_end,
                          const void *d_streams, size_t nstreams, hipStream_t st);

// inflate_resolve.hip: host token stream -> plaintext at d_dst; synchronises `st`
int inflate_tokens_to_device(const zng_rocm_inflate_tokens *tk, const uint8_t *d_window, uint32_t window_len, uint8_t *d_dst,
                             hipStream_t st);

// checksum.hip: one streaming pass (+ finalize) over device-resident bytes.  d_dst != nullptr = fold_copy.  When a
// d_seed_* pointer is given, that checksum's seed is read ON THE DEVICE from it at finalize time (the word an earlier
// launch on the same stream wrote) instead of from the scalar argument.
int launch_checksum(bool do_adler, bool do_crc, uint32_t adler, uint32_t crc, const void *d_buf, void *d_dst,
                    size_t len, uint32_t *d_out_adler, uint32_t *d_out_crc, hipStream_t stream,
                    const uint32_t *d_seed_adler = nullptr, const uint32_t *d_seed_crc = nullptr);

}  // namespace zr
