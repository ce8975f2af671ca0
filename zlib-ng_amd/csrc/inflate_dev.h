// inflate_dev.h -- what inflate_dev.hip shares with framing_dev.hip: the device job descriptor, the message ids of
// d_results (texts: zng_rocm_inflate_message) and the launcher for a job table that already sits in device memory.
#pragma once
#include "context.h"

namespace zr {

struct InflateJobDev {
    const uint8_t *in;
    uint8_t       *out;
    uint64_t       in_len;
    uint64_t       out_cap;
    uint32_t       dict_len;
    uint32_t       flags;
};

enum InflateMsg : uint32_t {
    kMsgNone = 0, kMsgBlockType, kMsgStoredLen, kMsgTooMany, kMsgCodeLengthsSet, kMsgBitRepeat, kMsgNoEob,
    kMsgLitLenSet, kMsgDistSet, kMsgLitLenCode, kMsgDistCode, kMsgTooFar, kMsgStarved, kMsgOutFull,
    // wrappers (framing_dev.hip; inflate.c:509-555 header checks, :686-692 FHCRC, :1105-1147 trailer checks)
    kMsgHeaderCheck, kMsgMethod, kMsgWindow, kMsgHeaderCrc, kMsgNeedDict, kMsgDataCheck, kMsgLengthCheck, kMsgCount
};


// one wavefront per job; d_jobs and d_results are device memory (results: 4 words per job)
int launch_inflate_streams_device(const InflateJobDev *d_jobs, size_t njobs, uint32_t *d_results, hipStream_t stream);

// the parts of ONE large stream, 16-bit symbols out (inflate_large.hip); results: 8 words per part
int launch_inflate_parts_device(const InflateJobDev *d_jobs, size_t njobs, uint32_t *d_results, const unsigned long long *d_starts,
                                bool many, hipStream_t stream);

}  // namespace zr
