// checksum_args.h -- the message descriptors of the streaming checksum kernels (checksum_kernel.h), shared with the
// translation units that fill them on the DEVICE (framing_dev.hip: the checksums of many inflated streams whose
// lengths only the device knows).
#pragma once
#include "context.h"

namespace zr {

// What the CRC prologue needs to BUILD its tables in registers instead of fetching them: entry e of a table that is
// linear over GF(2) is the XOR of the entries of e's set bits, so eight words per table suffice.  They travel as
// kernel arguments (scalar loads from the kernarg segment): the prologue touches no table in HBM at all -- measured,
// the fetch of the 9 KiB of tables cost 2.1-2.6 us per launch (L2 is flushed by the pass between two launches).
struct CrcBits {
    uint32_t stride[4][8];    // stride_tab[k][1 << i]
    uint32_t x32[4][8];       // x32_tab[k][1 << i]
};

struct StreamArgs {
    const uint8_t *a0;        // 16-byte aligned
    uint8_t       *dst0;      // COPY: destination of byte a0[0] (same 16-byte phase as a0), else unused
    long long      n;         // message bytes
    long long      body;      // tail_base - a0 (multiple of 16, >= 0)
    long long      nunits;    // ceil(body / kUnitBytes)
    int            head;      // buf - a0, 0..15
    int            tail;      // bytes of the message living in the granule at tail_base
    unsigned long long *phase_stamps;   // tools/micro only (PROFILE instantiations): 8 stamps per workgroup
    CrcBits        bits;
};

struct FinalArgs {
    const uint8_t *tail_base;   // granule holding the trailing bytes
    uint8_t       *tail_dst;    // COPY: where those bytes go (else nullptr)
    long long      n;
    long long      nunits;
    int            tail_lo;     // valid bytes of that granule: [tail_lo, tail_hi)
    int            tail_hi;
    int            groups;
    uint32_t       adler_seed;
    uint32_t       crc_seed;
    uint32_t       crc_len_pow;     // x^(8n), evaluated on the host (a handful of table multiplies)
    // seeds chained on the device (the bounded host-pointer staging of slots.hip): when non-null they replace the
    // two scalar seeds above with the words an earlier launch on the same stream wrote
    const uint32_t *adler_seed_ptr;
    const uint32_t *crc_seed_ptr;
    int            do_adler, do_crc;
};

// many-message pass over descriptors that already sit in device memory (checksum.hip): rows messages, one workgroup
// each, `d_part` rows Partial of scratch, two result words per message at d_out2
int launch_checksum_batch_device(bool do_adler, bool do_crc, const StreamArgs *d_messages, const FinalArgs *d_finals,
                                 Partial *d_part, size_t rows, uint32_t *d_out2, hipStream_t stream);

}  // namespace zr
