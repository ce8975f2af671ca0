// checksum_kernel.h -- the streaming Adler-32 / CRC-32 kernels (device code only; launched by checksum.hip and
// timed variant by variant by tools/micro/crc_phases.hip).
//
// Replaces, behind the functable boundary (functable.h:26-42):
//   adler32            arch/generic/adler32_c.c:11-54
//   crc32              arch/generic/crc32_braid_c.c:62-216
//   adler32_fold_copy  arch/generic/adler32_fold_c.c:11-15
//   crc32_fold_copy    arch/generic/crc32_fold_c.c:15-18
//
// Data layout.  The message [buf, buf+n) is cut on 16-byte address granules:
//   a0        = buf rounded down to 16        (first granule; its bytes below buf are masked to 0)
//   tail_base = (buf+n) rounded down to 16    (the <16 trailing bytes are folded in by the finalize kernel)
//   body      = [a0, tail_base), consumed in UNITS of 16 KiB = 1024 lanes x one dwordx4 each,
//               units right-aligned to tail_base (a partial unit can only be the FIRST one, and
//               missing leading pieces behave as leading zero bytes, which change neither checksum).
// Workgroup g owns a contiguous run of units; per step its 1024 lanes read one fully
// coalesced 16 KiB row.  Each lane therefore sees a strided sub-stream (stride 16 KiB):
//   CRC:   four braids per lane (the four dwords of its piece).  One step is the braid step of
//          crc32_braid_c.c:121-176 with the stride n*w = 16 KiB:  s <- s * x^(8*16384) ^ word,
//          done with four byte-indexed tables held in LDS, each replicated over the 32 banks
//          (entry e of lane l at bank l%32) so that every ds_read_b32 is conflict-free.
//   Adler: lane-local byte sum A (v_sad_u8) and in-piece weighted sum (v_dot4_u32_u8); positions
//          are applied in closed form, B = sum (n - pos) * byte, so blocks combine by plain addition
//          (the linear form of adler32_combine_).
// Each workgroup weights its CRC partial to the end of the body (x^(8 * 16 KiB * units_after), two table
// multiplies) and leaves one Partial; the small finalize kernel XORs / sums them, folds the tail bytes and
// the seed and writes the checksum(s) to device memory.
#pragma once
#include "context.h"
#include "checksum_args.h"

namespace zr {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// streaming store of one piece: written once, never read back by this kernel
__device__ __forceinline__ void st_stream(uint8_t *p, uint4 v) {
    u32x4_t x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<u32x4_t *>(p));
}

// streaming (read-once) 16-byte load: non-temporal hint, the data is never re-read
__device__ __forceinline__ uint4 ld_stream(const uint8_t *p) {
#ifdef ZR_EXPERIMENT_PLAIN_LOADS                     // tools/micro only: the same pass without the non-temporal hint
    const u32x4_t v = *reinterpret_cast<const u32x4_t *>(p);
#else
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
#endif
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() drains vmcnt as well, i.e. it waits for every global
// load the wave has in flight -- here the first 128 KiB of the pass, requested on purpose before the tables are built:
// measured, each __syncthreads() of the prologue then sat ~3 us on the HBM latency it was meant to overlap.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// first `k` bytes (k in 0..15) of a piece -> 0
__device__ __forceinline__ uint4 mask_low_bytes(uint4 v, int k) {
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int kk = k - 4 * i;                      // bytes of dword i to clear
        if (kk >= 4) w[i] = 0;
        else if (kk > 0) w[i] &= 0xffffffffu << (8 * kk);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// One braid step for one dword: s * x^(8*stride) ^ w, via the bank-replicated LDS tables.
// lut layout: dword index ((k*256 + e) << 5) + (lane & 31).
__device__ __forceinline__ uint32_t braid_step(const uint32_t *lut, uint32_t rep, uint32_t s, uint32_t w) {
    uint32_t r0 = lut[(((s)       & 0xffu) << 5) + rep];
    uint32_t r1 = lut[(1u << 13) + (((s >> 8)  & 0xffu) << 5) + rep];
    uint32_t r2 = lut[(2u << 13) + (((s >> 16) & 0xffu) << 5) + rep];
    uint32_t r3 = lut[(3u << 13) + (((s >> 24)       ) << 5) + rep];
    return r0 ^ r1 ^ r2 ^ r3 ^ w;
}

// a ^ b ^ c in one instruction (v_bitop3_b32, truth table 0x96); the compiler does not form it by itself
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}
// (a & b) ^ c
__device__ __forceinline__ uint32_t and_xor(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x6a);
}

// Byte-addressed form of the same step (kVarByteAddr).  The main loop is bound by vector-instruction issue, not by LDS
// or HBM: the form above spends bfe + shift-or on every lookup address (48 vector instructions per 16-byte piece, 16 of
// them XORs).  Here the table image is laid out so that the index is exactly BYTE 1 of the LDS byte address:
//     address = (k >> 1) * 65536 + e * 256 + (k & 1) * 128 + (lane & 31) * 4       (table k, entry e)
// (still 128 KiB, still bank = lane % 32), so one SDWA move that writes byte k of the state into byte 1 of a
// lane-constant address register (dst_unused:UNUSED_PRESERVE keeps the other three bytes) IS the address computation:
// 16 moves + 8 three-input XORs = 24 vector instructions per piece.  The four address registers live across the loop;
// each is rewritten only after the ds_read that used it has been issued (DS reads its address at issue).
typedef const __attribute__((address_space(3))) uint32_t *lds_u32_ptr;
template <int K>
__device__ __forceinline__ void put_index_byte(uint32_t &addr, uint32_t s) {
    if constexpr (K == 0) asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(addr) : "v"(s));
    if constexpr (K == 1) asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1" : "+v"(addr) : "v"(s));
    if constexpr (K == 2) asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(addr) : "v"(s));
    if constexpr (K == 3) asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(addr) : "v"(s));
}
struct BraidAddr {
    uint32_t a0, a1, a2, a3;      // LDS byte addresses of entry 0 of tables 0..3 for this lane; byte 1 is the index slot
};
__device__ __forceinline__ uint32_t braid_step_bytes(BraidAddr &A, uint32_t s, uint32_t w) {
    put_index_byte<0>(A.a0, s);
    const uint32_t r0 = *(lds_u32_ptr)(uintptr_t)A.a0;
    put_index_byte<1>(A.a1, s);
    const uint32_t r1 = *(lds_u32_ptr)(uintptr_t)A.a1;
    put_index_byte<2>(A.a2, s);
    const uint32_t r2 = *(lds_u32_ptr)(uintptr_t)A.a2;
    put_index_byte<3>(A.a3, s);
    const uint32_t r3 = *(lds_u32_ptr)(uintptr_t)A.a3;
    return xor3(xor3(r0, r1, r2), r3, w);
}

// XOR / sum of a value over the 64 lanes of the wave, returned wave-uniform.  Four DPP steps (lane permutes inside the
// vector ALU, no LDS round trip as with ds_bpermute) leave every 16-lane row holding its row total; four readlanes
// combine the rows on the scalar unit.
template <bool XOR>
__device__ __forceinline__ uint32_t wave_reduce(uint32_t v) {
    auto op = [](uint32_t a, uint32_t b) { return XOR ? (a ^ b) : (a + b); };
    v = op(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v = op(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v = op(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true));   // row_half_mirror
    v = op(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true));   // row_mirror
    return op(op((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
              op((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}

// r * w from w's 32 partial products held four to a 16-byte LDS row (rows kq * stride + idx): bit 31 of r is the x^0
// coefficient (mulmod, gf2.h).
__device__ __forceinline__ uint32_t mul_by_rows(const uint4 *rows, int stride, int idx, uint32_t r) {
    uint32_t acc = 0;
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) {
        const uint4 a = rows[kq * stride + idx];
        acc = and_xor((uint32_t)__builtin_amdgcn_sbfe((int)r, 31 - 4 * kq, 1), a.x, acc);
        acc = and_xor((uint32_t)__builtin_amdgcn_sbfe((int)r, 30 - 4 * kq, 1), a.y, acc);
        acc = and_xor((uint32_t)__builtin_amdgcn_sbfe((int)r, 29 - 4 * kq, 1), a.z, acc);
        acc = and_xor((uint32_t)__builtin_amdgcn_sbfe((int)r, 28 - 4 * kq, 1), a.w, acc);
    }
    return acc;
}

// Tuning switches of the CRC prologue / epilogue (bit mask V).  The PRODUCT build can instantiate kCrcVariant only (a
// static_assert in stream_body); tools/micro/crc_phases.hip defines ZR_CRC_VARIANTS and times every other setting -- the
// kernel's history -- on the same build.  The two timing-only settings that skip work (and return a wrong CRC) do not
// exist without that macro.
constexpr int kVarBothBuffers = 1;    // request BOTH register buffers before the table image is built
constexpr int kVarFoldX32     = 2;    // braid fold through x^32 tables: 3 LDS latencies instead of 12
constexpr int kVarEarlyPow    = 4;    // fetch the lane's 32 partial products before the main loop, not after it
#ifdef ZR_CRC_VARIANTS
constexpr int kVarNoReplicate = 16;   // TIMING ONLY (wrong CRC): skip the 128 KiB replication
constexpr int kVarNoMultiply  = 32;   // TIMING ONLY (wrong CRC): skip the per-lane weight multiply
#else
constexpr int kVarNoReplicate = 0, kVarNoMultiply = 0;       // not in the product: the tests below fold to "do the work"
#endif
constexpr int kVarBuildTables = 64;   // build the table entries in registers from CrcBits: no table fetch in the prologue
constexpr int kVarLdsBarrier  = 128;  // prologue barriers wait for LDS only, not for the data loads already in flight
constexpr int kVarByteAddr    = 1024; // table image with the index in byte 1 of the LDS address: one SDWA move per lookup
                                      //   address, three-input XORs (braid_step_bytes) -- 24 instead of 48 VALU per piece
constexpr int kVarDirectImage = 2048; // every lane computes the eight image entries it stores straight from CrcBits: no staging
                                      //   copy of the stride tables, ONE prologue barrier (needs kVarBuildTables|kVarByteAddr|kVarFoldX32)
constexpr int kVarTreeEpilogue = 4096; // lane weights in two levels (lane-in-wave from LDS, then wave) instead of 32 loads per lane;
                                      //   DPP wave reductions; the group-weight multiplies spread over the lanes of wave 0
constexpr int kVarWavePrio    = 8192; // later-launched waves of the workgroup get the higher issue priority (s_setprio wave / 4)
constexpr int kVarEarlyRows   = 768;  // two bits: how many rows of group 0 are requested BEFORE the table build when
                                      //   kVarBuildTables is on: 0 -> all (4), 256 -> none, 512 -> one, 768 -> two
// Measured on MI355X (tools/micro/crc_phases.hip, profiles/r02_crc_phases.md), crc32 over 64 MiB, dispatch time:
// round-1 form 17.9 us; fold + built tables + LDS barriers 17.1 us; byte-addressed image 15.9 us; direct image 15.5 us;
// two-level weights and DPP reductions 14.0 us (0.60 of the 8 TB/s peak; Adler-32 alone 11.3 us).  Requesting the second
// buffer, more rows, or the lane products before the table work made it SLOWER: a CU accepts only so many loads in
// flight, the wave stalls at issue.
constexpr int kCrcVariant = kVarFoldX32 | kVarBuildTables | kVarLdsBarrier | kVarByteAddr | kVarDirectImage | kVarTreeEpilogue;

// DO_ADLER / DO_CRC select the checksums, COPY additionally stores every piece (fold_copy).
// The kernel body: workgroup `g` of the `G` that share the message `args`.  Two entry points below: one message per
// launch (stream_kernel, the message in the kernel arguments) and many messages per launch (stream_kernel_batch, one
// descriptor per grid row -- the many-stream form of the same pass).
template <bool DO_ADLER, bool DO_CRC, bool COPY, int V, bool PROFILE, int UNROLL>
__device__ __forceinline__ void stream_body(const StreamArgs &args, const DeviceTables *__restrict__ tabs,
                                            Partial *__restrict__ partials, const long long G, const long long g) {
#ifndef ZR_CRC_VARIANTS
    static_assert(V == kCrcVariant, "the product instantiates the one tuned variant; the others are for tools/micro/crc_phases.hip");
#endif
    // 64 KiB alignment: with kVarByteAddr byte 1 of a lookup address must be free for the index (braid_step_bytes)
    __shared__ __attribute__((aligned(65536))) uint32_t lut[DO_CRC ? 4 * 256 * 32 : 32];
    // linear copies: the four stride tables, the byte table, the four x^32 tables
    __shared__ uint32_t stage[DO_CRC ? 4 * 256 + 256 + 4 * 256 : 4];
    __shared__ uint32_t red[3][kWgThreads / 64];
    constexpr bool TREE = (V & kVarTreeEpilogue) != 0;
    __shared__ uint4 tree_rows[(DO_CRC && TREE) ? 640 : 1];

    const int t = threadIdx.x;
    const uint32_t rep = t & 31;
    auto stamp = [&](int k) {
        if constexpr (PROFILE) {
            if (t == 0) args.phase_stamps[8 * g + k] = wall_clock64();
        }
    };
    stamp(0);
    if constexpr ((V & kVarWavePrio) != 0) {
        // the 16 waves of the workgroup start over ~1.1 us and the issue arbiter prefers the oldest: the last wave
        // finishes the stream well behind the first, and the workgroup is as slow as its last wave
        const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
        if (wv >= 12) __builtin_amdgcn_s_setprio(3);
        else if (wv >= 8) __builtin_amdgcn_s_setprio(2);
        else if (wv >= 4) __builtin_amdgcn_s_setprio(1);
    }

    // contiguous run of units for this workgroup
    const long long q = args.nunits / G, r = args.nunits % G;
    const long long u_lo = g * q + (g < r ? g : r);
    const long long u_hi = u_lo + q + (g < r ? 1 : 0);

    // byte offset (relative to a0) of this lane's piece in unit u:  body - (nunits-u)*U + 16t
    long long off = args.body - (args.nunits - u_lo) * (long long)kUnitBytes + (long long)t * kPieceBytes;

    // Main-loop geometry first, so that the first group(s) of HBM loads can be issued BEFORE the tables are built:
    // their latency then runs under the table build instead of after it.
    const bool head_unit = u_lo < u_hi && u_lo == 0;     // the message's first unit is handled apart (head mask)
    const long long groups = (u_hi - u_lo - (head_unit ? 1 : 0)) / UNROLL;
    const long long gstride = (long long)UNROLL * kUnitBytes;
    uint4 bufA[UNROLL], bufB[UNROLL];
    auto request = [&](uint4 (&buf)[UNROLL], long long at) {
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) buf[j] = ld_stream(args.a0 + at + (long long)j * kUnitBytes);
    };
    const long long first_at = off + (head_unit ? (long long)kUnitBytes : 0ll);
    bool b_requested = false;            // bufB already holds group 1
    uint32_t group_w0 = 0, group_w1 = 0;
    uint32_t part[32];                   // the lane's 32 partial products (epilogue multiply)
    uint4 tree_row = make_uint4(0, 0, 0, 0);   // kVarTreeEpilogue: this thread's row of tabs->tree_pp, stashed in LDS later
    if constexpr (DO_CRC) {
        // the two digits of this group's end-of-body weight (epilogue, thread 0): fetched now, not on the tail
        const unsigned long long k_after = (unsigned long long)(args.nunits - u_hi);
        if constexpr ((V & kVarBuildTables) != 0) {
            // A CU takes only so many loads in flight; a wave that asks for more stalls at ISSUE and with it everything
            // behind the load in its instruction stream.  So only the first rows are requested before the table work.
            constexpr int early = (V & kVarEarlyRows) == 0 ? UNROLL : ((V & kVarEarlyRows) >> 8) - 1;
            if (groups > 0) {
#pragma unroll
                for (int j = 0; j < early; ++j) bufA[j] = ld_stream(args.a0 + first_at + (long long)j * kUnitBytes);
            }
            if constexpr ((V & kVarBothBuffers) != 0 && early == UNROLL) {
                if (groups > 1) {
                    request(bufB, first_at + gstride);
                    b_requested = true;
                }
            }
            group_w0 = tabs->unit_pow[0][k_after & 1023u];
            group_w1 = tabs->unit_pow[1][(k_after >> 10) & 1023u];
            if constexpr (TREE) tree_row = *reinterpret_cast<const uint4 *>(tabs->tree_pp[t]);
            if constexpr ((V & kVarEarlyPow) != 0) {
#pragma unroll
                for (int k = 0; k < 32; ++k) part[k] = tabs->lane_pow[k][t];
            }
            // lane t builds entry e = t % 256 of stride table and of x^32 table k = t / 256 (k is wave-uniform), and the
            // first 256 lanes one byte-table entry each (the shift register of tools/makecrct.c:66-73)
            const int k = __builtin_amdgcn_readfirstlane(t >> 8);
            const uint32_t e = (uint32_t)t & 255u;
            if constexpr ((V & kVarDirectImage) != 0) {
                static_assert((V & kVarDirectImage) == 0 || ((V & kVarByteAddr) != 0 && (V & kVarFoldX32) != 0),
                              "direct image: byte-addressed layout and x^32 fold only");
                // The image straight from the bit words: lane t stores quarter t % 8 (four replicas, one b128) of the
                // 128-byte slots of the eight entries {table j, index h * 128 + t / 8}.  The low seven index bits are the
                // lane's own, so a table costs seven and-xor steps for both halves (bit 7 = one more XOR): 39 vector
                // instructions, no staging copy, and each wave can start the moment it is launched -- the workgroup's
                // 16 waves arrive over ~1.1 us, and a barrier in front of the image made every wave wait for the last.
                uint32_t m[7];
#pragma unroll
                for (int b = 0; b < 7; ++b) m[b] = (uint32_t)__builtin_amdgcn_sbfe((int)((uint32_t)t >> 3), b, 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t v = 0;
#pragma unroll
                    for (int b = 0; b < 7; ++b) v = and_xor(m[b], args.bits.stride[j][b], v);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t vv = h ? v ^ args.bits.stride[j][7] : v;
                        const int slot = (j >> 1) * 512 + ((h * 128 + (t >> 3)) << 1) + (j & 1);
                        reinterpret_cast<uint4 *>(lut)[slot * 8 + (t & 7)] = make_uint4(vv, vv, vv, vv);
                    }
                }
                uint32_t xv = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) xv = and_xor((uint32_t)__builtin_amdgcn_sbfe((int)e, i, 1), args.bits.x32[k][i], xv);
                stage[1280 + t] = xv;
            } else {
                uint32_t sv = 0, xv = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t m = 0u - ((e >> i) & 1u);
                    sv ^= args.bits.stride[k][i] & m;
                    xv ^= args.bits.x32[k][i] & m;
                }
                stage[t] = sv;
                stage[1280 + t] = xv;
                if (t < 256) {
                    uint32_t r = e;
#pragma unroll
                    for (int i = 0; i < 8; ++i) r = (r >> 1) ^ (kCrcPoly & (0u - (r & 1u)));
                    stage[1024 + t] = r;
                }
            }
        } else {
            // tables: HBM/L2 -> LDS once (9 KiB, one dwordx4 per lane of the first 9 waves)
            uint4 tab = make_uint4(0, 0, 0, 0);
            if (t < 256) {
                tab = reinterpret_cast<const uint4 *>(&tabs->stride_tab[0][0])[t];
            } else if (t < 320) {
                tab = reinterpret_cast<const uint4 *>(tabs->byte_tab)[t - 256];
            } else if (t < 576) {
                tab = reinterpret_cast<const uint4 *>(&tabs->x32_tab[0][0])[t - 320];
            }
            group_w0 = tabs->unit_pow[0][k_after & 1023u];
            group_w1 = tabs->unit_pow[1][(k_after >> 10) & 1023u];
            if (groups > 0) request(bufA, first_at);
            if constexpr ((V & kVarBothBuffers) != 0) {
                if (groups > 1) {
                    request(bufB, first_at + gstride);
                    b_requested = true;
                }
            }
            if constexpr ((V & kVarEarlyPow) != 0) {
#pragma unroll
                for (int k = 0; k < 32; ++k) part[k] = tabs->lane_pow[k][t];
            }
            if (t < 576) reinterpret_cast<uint4 *>(stage)[t] = tab;
        }
        if constexpr ((V & kVarDirectImage) == 0) {
            if constexpr ((V & kVarLdsBarrier) != 0) lds_barrier();
            else __syncthreads();
        }
        stamp(1);
        // replicated LDS -> LDS: 32768 dwords, lane-consecutive writes; the 32 replicas of one entry are 32 adjacent
        // dwords: eight lanes write one entry, four replicas (one b128) each
        if constexpr ((V & kVarNoReplicate) == 0 && (V & kVarDirectImage) == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // entry E = i * 128 + t / 8 = table i / 2, index (i & 1) * 128 + t / 8
                const uint32_t v = stage[i * 128 + (t >> 3)];
                if constexpr ((V & kVarByteAddr) != 0) {
                    // 128-byte slot number (i / 4) * 512 + index * 2 + (table & 1); eight b128 per slot
                    const int slot = (i >> 2) * 512 + (((i & 1) * 128 + (t >> 3)) << 1) + ((i >> 1) & 1);
                    reinterpret_cast<uint4 *>(lut)[slot * 8 + (t & 7)] = make_uint4(v, v, v, v);
                } else {
                    reinterpret_cast<uint4 *>(lut)[i * kWgThreads + t] = make_uint4(v, v, v, v);
                }
            }
        }
        if constexpr ((V & kVarLdsBarrier) != 0) lds_barrier();
        else __syncthreads();
        if constexpr ((V & kVarBuildTables) != 0 && (V & kVarEarlyRows) != 0) {
            constexpr int early = ((V & kVarEarlyRows) >> 8) - 1;
            if (groups > 0) {
#pragma unroll
                for (int j = early; j < UNROLL; ++j) bufA[j] = ld_stream(args.a0 + first_at + (long long)j * kUnitBytes);
            }
            if constexpr ((V & kVarBothBuffers) != 0) {
                if (groups > 1) {
                    request(bufB, first_at + gstride);
                    b_requested = true;
                }
            }
        }
    } else {
        if (groups > 0) request(bufA, first_at);
    }
    stamp(2);

    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;             // CRC braids
    BraidAddr braid_addr;
    if constexpr (DO_CRC && (V & kVarByteAddr) != 0) {
        const uint32_t base = (uint32_t)(uintptr_t)(lds_u32_ptr)lut + rep * 4u;
        if ((base & 0xff00u) != 0) __builtin_trap();     // the image is not where the address arithmetic needs it
        braid_addr.a0 = base;
        braid_addr.a1 = base + 128u;
        braid_addr.a2 = base + 65536u;
        braid_addr.a3 = base + 65536u + 128u;
    }
    uint32_t S1 = 0, SR = 0, SW = 0;                     // Adler: byte sum, prefix-of-sums, in-piece weights
    uint32_t accA = 0, accB = 0;                         // Adler, reduced mod BASE between batches
    int batch = 0;

    auto fold_adler = [&](long long off_last) {
        // B over the batch = sum_k (n - o_k - 16) * A_k + W_k, o_k = piece offset relative to buf.
        // With pieces one unit apart:  (n - o_last - 16) * S1 + U * SR + SW  (all terms >= 0).
        // Everything mod BASE in 32-bit arithmetic: the lead term is wave-uniform up to the lane's 16 t, so its 64-bit
        // reduction runs once on the scalar unit; the 64-bit form cost three software divisions per lane on the tail.
        const unsigned long long off0 = (unsigned long long)(off_last - (long long)t * kPieceBytes);
        const unsigned long long off0_u = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(off0 >> 32)) << 32) |
                                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)off0);
        const uint32_t lead0 = (uint32_t)((unsigned long long)(args.n - ((long long)off0_u - args.head) - kPieceBytes) % kAdlerBase);
        const uint32_t back = (uint32_t)t * (uint32_t)kPieceBytes;          // < 16384 < BASE
        const uint32_t lead = lead0 >= back ? lead0 - back : lead0 + kAdlerBase - back;
        const uint32_t s1m = S1 % kAdlerBase;
        uint32_t v = (lead * s1m) % kAdlerBase;                             // < 65521^2 < 2^32
        v += ((uint32_t)kUnitBytes * (SR % kAdlerBase)) % kAdlerBase;       // 16384 * 65520 < 2^30
        v += SW % kAdlerBase;
        accB = (accB + v) % kAdlerBase;
        accA = (accA + s1m) % kAdlerBase;
        S1 = SR = SW = 0;
        batch = 0;
    };

    auto consume = [&](uint4 v) {
        if constexpr (DO_CRC) {
            if constexpr ((V & kVarByteAddr) != 0) {
                s0 = braid_step_bytes(braid_addr, s0, v.x);
                s1 = braid_step_bytes(braid_addr, s1, v.y);
                s2 = braid_step_bytes(braid_addr, s2, v.z);
                s3 = braid_step_bytes(braid_addr, s3, v.w);
            } else {
                s0 = braid_step(lut, rep, s0, v.x);
                s1 = braid_step(lut, rep, s1, v.y);
                s2 = braid_step(lut, rep, s2, v.z);
                s3 = braid_step(lut, rep, s3, v.w);
            }
        }
        if constexpr (DO_ADLER) {
            // the running sums ride in the accumulator operand: nine vector instructions per piece
            SR += S1;
            S1 = __builtin_amdgcn_sad_u8(v.x, 0u, S1);
            S1 = __builtin_amdgcn_sad_u8(v.y, 0u, S1);
            S1 = __builtin_amdgcn_sad_u8(v.z, 0u, S1);
            S1 = __builtin_amdgcn_sad_u8(v.w, 0u, S1);
            SW = __builtin_amdgcn_udot4(v.x, 0x0D0E0F10u, SW, false);          // weights 16,15,14,13
            SW = __builtin_amdgcn_udot4(v.y, 0x090A0B0Cu, SW, false);          // 12..9
            SW = __builtin_amdgcn_udot4(v.z, 0x05060708u, SW, false);          // 8..5
            SW = __builtin_amdgcn_udot4(v.w, 0x01020304u, SW, false);          // 4..1
        }
    };

    long long u = u_lo;
    // first unit of the whole message may be partial (pieces below a0 do not exist) and holds the head mask
    if (u < u_hi && u == 0) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (off >= 0) {
            v = *reinterpret_cast<const uint4 *>(args.a0 + off);
            if (off == 0 && args.head) v = mask_low_bytes(v, args.head);
            if constexpr (COPY) {
                if (off == 0 && args.head) {
                    for (int j = args.head; j < 16; ++j) args.dst0[j] = args.a0[j];
                } else {
                    *reinterpret_cast<uint4 *>(args.dst0 + off) = v;
                }
            }
        }
        consume(v);
        if constexpr (DO_ADLER) { batch = 1; }
        off += kUnitBytes;
        ++u;
    }

    // Main loop, software pipelined with two register buffers (ping-pong, no copies): while group k is
    // consumed the UNROLL rows of group k+1 are already in flight, so every lane keeps UNROLL..2*UNROLL
    // dwordx4 loads outstanding (16 waves x 4..8 KiB per CU) and each wait is a counted vmcnt(UNROLL).
    constexpr int BATCH_MAX = 240;     // u32 bounds: SR <= 4080 * 240*239/2 < 2^27
    if (groups > 0) {
        auto retire = [&](uint4 (&buf)[UNROLL]) {
            if constexpr (COPY) {
#pragma unroll
                for (int j = 0; j < UNROLL; ++j)
                    st_stream(args.dst0 + off + (long long)j * kUnitBytes, buf[j]);
            }
#pragma unroll
            for (int j = 0; j < UNROLL; ++j) consume(buf[j]);
            off += (long long)UNROLL * kUnitBytes;
            if constexpr (DO_ADLER) {
                batch += UNROLL;
                if (batch >= BATCH_MAX) fold_adler(off - kUnitBytes);
            }
        };
        long long k = 0;                                   // group 0 is already in flight (issued above)
        if (b_requested) {                                 // ... and so is group 1: retire group 0 against it
            __builtin_amdgcn_sched_barrier(0);
            retire(bufA);
            k = 1;
            // from here on the roles of the buffers are swapped: the group in flight sits in bufB
            while (k + 2 < groups) {
                request(bufA, off + gstride);
                __builtin_amdgcn_sched_barrier(0);
                retire(bufB);
                request(bufB, off + gstride);
                __builtin_amdgcn_sched_barrier(0);
                retire(bufA);
                k += 2;
            }
            if (groups - k == 2) {
                request(bufA, off + gstride);
                __builtin_amdgcn_sched_barrier(0);
                retire(bufB);
                retire(bufA);
            } else {
                retire(bufB);
            }
        } else {
            while (k + 2 < groups) {                       // steady state: no conditional loads, counted waits
                request(bufB, off + gstride);              // group k+1
                __builtin_amdgcn_sched_barrier(0);
                retire(bufA);                              // group k (off advances by one group)
                request(bufA, off + gstride);              // group k+2
                __builtin_amdgcn_sched_barrier(0);
                retire(bufB);
                k += 2;
            }
            if (groups - k == 2) {
                request(bufB, off + gstride);
                __builtin_amdgcn_sched_barrier(0);
                retire(bufA);
                retire(bufB);
            } else {
                retire(bufA);
            }
        }
        u += groups * UNROLL;
    }
    for (; u < u_hi; ++u) {
        uint4 v = *reinterpret_cast<const uint4 *>(args.a0 + off);
        if constexpr (COPY) *reinterpret_cast<uint4 *>(args.dst0 + off) = v;
        consume(v);
        off += kUnitBytes;
        if constexpr (DO_ADLER) ++batch;
    }
    if constexpr (DO_ADLER) {
        if (batch) fold_adler(off - kUnitBytes);
    }
    stamp(3);

    // ---- workgroup reduction ------------------------------------------------
    uint32_t pc = 0;
    if constexpr (DO_CRC) {
        if (u_hi > u_lo) {
            // fold the four braids into one word located at the lane's last dword:
            //   r = ((s0 * x^32 ^ s1) * x^32 ^ s2) * x^32 ^ s3, then ONE GF(2) multiply by that dword's weight.
            uint32_t r = s0;
            if constexpr ((V & kVarFoldX32) != 0) {
                // each x^32 = four independent lookups in the x^32 tables (one LDS latency)
                const uint32_t *xt = stage + 1280;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    r = xt[r & 0xffu] ^ xt[256 + ((r >> 8) & 0xffu)] ^ xt[512 + ((r >> 16) & 0xffu)] ^ xt[768 + (r >> 24)];
                    r ^= c == 0 ? s1 : (c == 1 ? s2 : s3);
                }
            } else {
                // each x^32 = four dependent byte-table steps (DO1 of crc32_braid_p.h:58)
                const uint32_t *bt = stage + 1024;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) r = bt[r & 0xffu] ^ (r >> 8);
                    r ^= c == 0 ? s1 : (c == 1 ? s2 : s3);
                }
            }
            // r * (this lane's weight)
            if constexpr ((V & kVarNoMultiply) != 0) {
                pc = r;
            } else if constexpr (TREE) {
                pc = r;                 // multiplied below, behind the barrier that publishes tree_rows
            } else {
                // the 32 partial products weight * x^k come from a table (32 lane-consecutive loads in flight together)
                // instead of a 32-step shift-and-reduce loop.  Bit 31 of r is the x^0 coefficient.
                if constexpr ((V & kVarEarlyPow) == 0) {
#pragma unroll
                    for (int k = 0; k < 32; ++k) part[k] = tabs->lane_pow[k][t];
                }
#pragma unroll
                for (int k = 0; k < 32; ++k) pc ^= part[k] & (0u - ((r >> (31 - k)) & 1u));
            }
        }
    }
    const int wave = t >> 6;
    const int lane = t & 63;
    if constexpr (TREE) {
        // Two-level lane weights.  Lane t = 64 w + l weighs (lane l of the last wave) * x^(8 * 1024 * (15 - w)): the first
        // factor's partial products are the same 8 KiB for every wave and come from LDS (8 x ds_read_b128 per lane; the
        // 32 loads per lane of the one-level form were 128 KiB per workgroup through the texture path, ~1.2 us), the second
        // is applied to the 16 wave results by 16 lanes of wave 0.
        uint32_t gpp = 0;
        if constexpr (DO_CRC) {
            if (t < 640) tree_rows[t] = tree_row;
            // wave 0 is the first to get here and waits for the slowest wave anyway: it uses the time to spread the two
            // group-weight factors into their partial products, lane j: w0 * x^j, lane 32 + j: w1 * x^j
            if (wave == 0) {
                gpp = lane < 32 ? group_w0 : group_w1;
                const int steps = lane & 31;
                for (int i = 0; i < 31; ++i) {
                    const uint32_t nx = (gpp >> 1) ^ (kCrcPoly & (0u - (gpp & 1u)));
                    gpp = i < steps ? nx : gpp;
                }
            }
            lds_barrier();
            if constexpr ((V & kVarNoMultiply) == 0) pc = mul_by_rows(tree_rows, 64, lane, pc);
        }
        stamp(4);
        uint32_t wc = 0, wa = 0, wb = 0;
        if constexpr (DO_CRC) wc = wave_reduce<true>(pc);
        if constexpr (DO_ADLER) {
            wa = wave_reduce<false>(accA);      // <= 64 * 65520
            wb = wave_reduce<false>(accB);
        }
        if (lane == 0) {
            red[0][wave] = wc;
            red[1][wave] = wa;
            red[2][wave] = wb;
        }
        lds_barrier();
        if (wave == 0) {
            uint32_t c = 0, a = 0, b = 0;
            if constexpr (DO_CRC) {
                uint32_t v = lane < 16 ? red[0][lane & 15] : 0u;
                v = mul_by_rows(tree_rows + 512, 16, lane & 15, v);
                c = wave_reduce<true>(v);
                // Weight the group's CRC to the end of the body: c * x^(8 * U * units_after) = c * w0 * w1 (two table
                // digits; digit 0 is x^0, so no case split).  The common factor x^(8 * tail bytes) is applied once, by the
                // finalize kernel.  Each multiply = one masked term per lane + a wave reduction (a 32-step dependent
                // shift-and-reduce loop on one lane took ~0.5 us each).
                const uint32_t sel = 31u - (uint32_t)(lane & 31);
                c = wave_reduce<true>(lane < 32 ? gpp & (0u - ((c >> sel) & 1u)) : 0u);
                c = wave_reduce<true>(lane >= 32 ? gpp & (0u - ((c >> sel) & 1u)) : 0u);
            }
            if constexpr (DO_ADLER) {
                a = wave_reduce<false>(lane < 16 ? red[1][lane & 15] : 0u);     // <= 1024 * 65520 < 2^27
                b = wave_reduce<false>(lane < 16 ? red[2][lane & 15] : 0u);
            }
            if (lane == 0) {
                Partial pt;
                pt.crc = c;                               // already weighted to the end of the body
                pt.a = a % kAdlerBase;
                pt.b = b % kAdlerBase;
                pt.pad = 0;
                partials[g] = pt;
            }
        }
    } else {
        stamp(4);
        uint32_t pa = accA, pb = accB;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            if constexpr (DO_CRC) pc ^= __shfl_xor(pc, m, 64);
            if constexpr (DO_ADLER) {
                pa += __shfl_xor(pa, m, 64);
                pb += __shfl_xor(pb, m, 64);
            }
        }
        if ((t & 63) == 0) {
            red[0][wave] = pc;
            red[1][wave] = pa;     // <= 64 * 65520
            red[2][wave] = pb;
        }
        __syncthreads();
        if (t == 0) {
            uint32_t c = 0;
            unsigned long long a = 0, b = 0;
            for (int w = 0; w < kWgThreads / 64; ++w) {
                c ^= red[0][w];
                a += red[1][w];
                b += red[2][w];
            }
            // Weight this group's CRC to the end of the body: x^(8 * U * units_after), two table digits.  The
            // common factor x^(8 * tail bytes) is applied once, by the last group.
            if constexpr (DO_CRC) {
                const unsigned long long k = (unsigned long long)(args.nunits - u_hi);
                if (c && k) {
                    c = mulmod(c, group_w0);
                    if (k >> 10) c = mulmod(c, group_w1);
                }
            }
            Partial pt;
            pt.crc = c;                                   // already weighted to the end of the body
            pt.a = (uint32_t)(a % kAdlerBase);
            pt.b = (uint32_t)(b % kAdlerBase);
            pt.pad = 0;
            partials[g] = pt;
        }
    }
    stamp(5);
}

// UNROLL = rows per register buffer.  For the read-only passes one row per buffer (two rows in flight per lane, 32 per
// CU) measured best at every size (tools/micro/crc_phases, profiles/r02_crc_phases.md run 8: fused 1 GiB 0.798 -> 0.830 of
// peak, 256 MiB 0.69 -> 0.79, 64 MiB 0.53 -> 0.59 against four rows): a CU accepts only so many loads in flight, and
// deeper buffers only queue.  The copying passes (fold_copy: a store stream beside the load stream) keep four: with
// one they fall from 0.72 to 0.63 of peak at 1 GiB.
constexpr int kRowsPerBuffer = 1, kRowsPerBufferCopy = 4;
template <bool DO_ADLER, bool DO_CRC, bool COPY, int V = kCrcVariant, bool PROFILE = false,
          int UNROLL = COPY ? kRowsPerBufferCopy : kRowsPerBuffer>
__global__ __launch_bounds__(kWgThreads)
void stream_kernel(StreamArgs args, const DeviceTables *__restrict__ tabs, Partial *__restrict__ partials) {
    stream_body<DO_ADLER, DO_CRC, COPY, V, PROFILE, UNROLL>(args, tabs, partials, gridDim.x, blockIdx.x);
}

// grid = (workgroups per message, messages): message blockIdx.y, its partials at partials[blockIdx.y * gridDim.x ...]
template <bool DO_ADLER, bool DO_CRC, int V = kCrcVariant>
__global__ __launch_bounds__(kWgThreads)
void stream_kernel_batch(const StreamArgs *__restrict__ messages, const DeviceTables *__restrict__ tabs,
                         Partial *__restrict__ partials) {
    stream_body<DO_ADLER, DO_CRC, false, V, false, kRowsPerBuffer>(messages[blockIdx.y], tabs, partials + (size_t)blockIdx.y * gridDim.x,
                                                      gridDim.x, blockIdx.x);
}


__device__ __forceinline__
void finalize_body(const FinalArgs &fa, const DeviceTables *__restrict__ tabs, const Partial *__restrict__ partials,
                   uint32_t *__restrict__ out_adler, uint32_t *__restrict__ out_crc) {
    __shared__ uint32_t red[3][4];
    const int t = threadIdx.x;
    const long long G = fa.groups;
    const int ntail = fa.tail_hi - fa.tail_lo;

    uint32_t c = 0;
    unsigned long long a = 0, b = 0;
    for (long long g = t; g < G; g += blockDim.x) {
        Partial p = partials[g];
        c ^= p.crc;                                  // weighted to the end of the body by its workgroup
        a += p.a;
        b += p.b;
    }
    // trailing (< 16) bytes: one lane each; kept apart from the body sum, which still has to be advanced
    // over them (one multiply, by lane 0 at the end)
    uint32_t tc = 0;
    if (t < ntail) {
        uint32_t byte = fa.tail_base[fa.tail_lo + t];
        if (fa.tail_dst) fa.tail_dst[fa.tail_lo + t] = (uint8_t)byte;
        int after = ntail - 1 - t;                  // message bytes behind this one
        if (fa.do_crc) tc = mulmod(tabs->byte_tab[byte], tabs->pow_tab[after]);
        a += byte;
        b += (unsigned long long)byte * (unsigned)(after + 1);
    }
    __shared__ uint32_t red_t[4];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        c ^= __shfl_xor(c, m, 64);
        tc ^= __shfl_xor(tc, m, 64);
        a += __shfl_xor(a, m, 64);
        b += __shfl_xor(b, m, 64);
    }
    if ((t & 63) == 0) {
        red_t[t >> 6] = tc;
        red[0][t >> 6] = c;
        red[1][t >> 6] = (uint32_t)(a % kAdlerBase);
        red[2][t >> 6] = (uint32_t)(b % kAdlerBase);
    }
    __syncthreads();
    if (t == 0) {
        uint32_t cc = red[0][0] ^ red[0][1] ^ red[0][2] ^ red[0][3];
        unsigned long long A = (unsigned long long)red[1][0] + red[1][1] + red[1][2] + red[1][3];
        unsigned long long B = (unsigned long long)red[2][0] + red[2][1] + red[2][2] + red[2][3];
        if (fa.do_adler) {
            // seed halves are masked, not reduced (adler32_c.c:16-17); s1' = s1 + A, s2' = s2 + n*s1 + B
            const uint32_t seed = fa.adler_seed_ptr ? *fa.adler_seed_ptr : fa.adler_seed;
            unsigned long long s1 = seed & 0xffffu, s2 = (seed >> 16) & 0xffffu;
            unsigned long long n_mod = (unsigned long long)fa.n % kAdlerBase;
            unsigned long long r1 = (s1 + A) % kAdlerBase;
            unsigned long long r2 = (s2 + n_mod * s1 + B) % kAdlerBase;
            *out_adler = (uint32_t)(r1 | (r2 << 16));
        }
        if (fa.do_crc) {
            // body advanced over the tail bytes, tail terms, and the seed's image ~seed * x^(8n)
            const uint32_t seed = fa.crc_seed_ptr ? *fa.crc_seed_ptr : fa.crc_seed;
            const uint32_t seed_term = mulmod(~seed, fa.crc_len_pow);
            const uint32_t body = (ntail && cc) ? mulmod(cc, tabs->pow_tab[ntail]) : cc;
            *out_crc = ~(body ^ red_t[0] ^ red_t[1] ^ red_t[2] ^ red_t[3] ^ seed_term);
        }
    }
}

__global__ __launch_bounds__(256)
void finalize_kernel(FinalArgs fa, const DeviceTables *__restrict__ tabs, const Partial *__restrict__ partials,
                     uint32_t *__restrict__ out_adler, uint32_t *__restrict__ out_crc) {
    finalize_body(fa, tabs, partials, out_adler, out_crc);
}

// one workgroup per message: message blockIdx.x, `groups` partials each, results at out2[2 * message] (adler, crc)
__global__ __launch_bounds__(256)
void finalize_kernel_batch(const FinalArgs *__restrict__ messages, const DeviceTables *__restrict__ tabs,
                           const Partial *__restrict__ partials, uint32_t *__restrict__ out2) {
    const FinalArgs fa = messages[blockIdx.x];
    finalize_body(fa, tabs, partials + (size_t)blockIdx.x * fa.groups, out2 + 2 * (size_t)blockIdx.x,
                  out2 + 2 * (size_t)blockIdx.x + 1);
}


}  // namespace zr
