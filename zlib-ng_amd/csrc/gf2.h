// gf2.h -- GF(2)[x] / p(x) arithmetic for CRC-32, shared by host table setup and device kernels.
//
// Representation is zlib's reflected one (crc32_braid_p.h:62, tools/makecrct.c:38-62):
// bit 31 of a word is x^0, bit 0 is x^31, p(x) = 0xedb88320 with x^32 implied.
// A "register value" r advanced over k more message bytes is r * x^(8k) mod p.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZR_HD __host__ __device__ __forceinline__
#else
#define ZR_HD inline
#endif

namespace zr {

constexpr uint32_t kCrcPoly = 0xedb88320u;
constexpr uint32_t kAdlerBase = 65521u;

// a(x) * b(x) mod p(x); branch-free, defined for every a (the reference's
// multmodp, crc32_braid_comb_p.h:8-24, requires a != 0; same value otherwise).
ZR_HD uint32_t mulmod(uint32_t a, uint32_t b) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 31; i >= 0; --i) {
        acc ^= b & (0u - ((a >> i) & 1u));
        b = (b >> 1) ^ (kCrcPoly & (0u - (b & 1u)));
    }
    return acc;
}

// x^e mod p for a bit exponent e, square-and-multiply from x^1 (host-side table setup only).
inline uint32_t xpow_bits(uint64_t e) {
    uint32_t result = 0x80000000u;      // x^0
    uint32_t sq = 0x40000000u;          // x^1
    while (e) {
        if (e & 1u) result = mulmod(sq, result);
        sq = mulmod(sq, sq);
        e >>= 1;
    }
    return result;
}

// x^(8*d) as a product of 7-bit-digit table entries: digit i of d selects
// tab[i*128 + digit] = x^(8 * digit * 128^i).  Five digits cover d < 2^35 bytes.
constexpr int kPowDigits = 5;
ZR_HD uint32_t xpow_bytes(const uint32_t *tab, uint64_t d) {
    uint32_t r = tab[d & 127u];
#pragma unroll
    for (int i = 1; i < kPowDigits; ++i) {
        uint32_t dig = (uint32_t)(d >> (7 * i)) & 127u;
        if (dig) r = mulmod(r, tab[i * 128 + dig]);
    }
    return r;
}

}  // namespace zr
