// context.h -- process-global device context of the arch/rocm backend:
// constant tables in HBM, per-stream workspaces, pinned staging, error state.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

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
};

struct Partial {            // one per workgroup, written by the streaming kernel
    uint32_t crc;           // register contribution relative to the end of the group's span
    uint32_t a;             // adler A contribution (mod 65521)
    uint32_t b;             // adler B contribution, already weighted to the end of the message (mod 65521)
    uint32_t pad;
};

struct Workspace {          // one per HIP stream
    Partial  *partials;     // kMaxGroups entries
    uint32_t *acc;          // {crc xor, adler A sum, adler B sum, ticket}: device-scope accumulators, zero between calls
    uint32_t *result;       // 2 x u32 scratch result (device)
    uint32_t *pinned;       // 2 x u32 host-pinned mirror
    uint8_t  *stage;        // device staging for host-pointer slots
    size_t    stage_bytes;
    uint8_t  *pinned_stage; // pinned bounce buffer for H2D
    size_t    pinned_bytes;
};

struct Context {
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

// measurement hooks (zng_rocm_trace_begin/_end): event pair around the dominant kernel
void trace_mark(hipStream_t s, bool begin);

Context   *ctx();                          // nullptr until zng_rocm_init succeeded
Workspace *workspace_for(hipStream_t s);   // lazily created, nullptr on failure
int        ensure_stage(Workspace *ws, size_t bytes);
[[noreturn]] void die(const char *what);   // loud failure for slots without an error channel

}  // namespace zr
