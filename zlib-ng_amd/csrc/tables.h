// tables.h -- host-side construction of the constant tables (DeviceTables, context.h) that zng_rocm_init uploads.
// Everything is derived from the CRC-32 polynomial; nothing is copied from crc32_braid_tbl.h.
#pragma once
#include "context.h"

namespace zr {

inline void build_tables(DeviceTables &t) {
    // byte table: the shift-register construction of tools/makecrct.c:66-73
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t r = i;
        for (int k = 0; k < 8; ++k) r = (r >> 1) ^ (kCrcPoly & (0u - (r & 1u)));
        t.byte_tab[i] = r;
    }
    // stride tables: tools/makecrct.c:99-112 with the braid stride n*w replaced
    // by this kernel's stride (one 16 KiB workgroup row)
    for (int k = 0; k < 4; ++k) {
        uint32_t adv = xpow_bits(8ull * (uint64_t)(kUnitBytes + 3 - k));
        uint32_t adv32 = xpow_bits(8ull * (uint64_t)(3 - k) + 32ull);
        for (uint32_t b = 0; b < 256; ++b) {
            t.stride_tab[k][b] = mulmod(b << 24, adv);
            t.x32_tab[k][b] = mulmod(b << 24, adv32);
        }
    }
    // lane weights: x^(8*(U - 16t - 4c)), built incrementally from the far end
    //   w(t,c) with distance d = U - 16t - 4c; d decreases by 4 per (c+1)
    {
        uint32_t x32 = xpow_bits(32);
        uint32_t cur = x32;  // d = 4 : t = kWgThreads-1, c = 3
        for (int lane = kWgThreads - 1; lane >= 0; --lane)
            for (int c = 3; c >= 0; --c) {
                t.lane_weight[lane][c] = cur;
                cur = mulmod(cur, x32);
            }
    }
    for (int lane = 0; lane < kWgThreads; ++lane) {
        uint32_t b = t.lane_weight[lane][3];
        for (int k = 0; k < 32; ++k) {
            t.lane_pow[k][lane] = b;
            b = (b >> 1) ^ (kCrcPoly & (0u - (b & 1u)));                  // times x, reflected representation
        }
    }
    // two-level lane weights: lane t = 64 w + l weighs x^(8*(16*(63-l)+4)) [= the last wave's lane weight] times
    // x^(8*1024*(15-w)); each with its 32 partial products, four per 16-byte row
    for (int i = 0; i < 1024; ++i)
        for (int j = 0; j < 4; ++j) t.tree_pp[i][j] = 0;
    for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 32; ++k) t.tree_pp[(k >> 2) * 64 + l][k & 3] = t.lane_pow[k][kWgThreads - 64 + l];
    for (int w = 0; w < 16; ++w) {
        uint32_t b = xpow_bits(8ull * 1024ull * (uint64_t)(15 - w));
        for (int k = 0; k < 32; ++k) {
            t.tree_pp[512 + (k >> 2) * 16 + w][k & 3] = b;
            b = (b >> 1) ^ (kCrcPoly & (0u - (b & 1u)));
        }
    }
    for (int i = 0; i < 2; ++i) {
        uint32_t step = xpow_bits(8ull * (uint64_t)kUnitBytes << (10 * i));   // x^(8 * U * 1024^i)
        uint32_t cur = 0x80000000u;
        for (int d = 0; d < 1024; ++d) {
            t.unit_pow[i][d] = cur;
            cur = mulmod(cur, step);
        }
    }
    for (int i = 0; i < kPowDigits; ++i) {
        uint32_t step = xpow_bits(8ull << (7 * i));      // x^(8 * 128^i)
        uint32_t cur = 0x80000000u;                      // digit 0 -> x^0
        for (int d = 0; d < 128; ++d) {
            t.pow_tab[i * 128 + d] = cur;
            cur = mulmod(cur, step);
        }
    }
}

}  // namespace zr
