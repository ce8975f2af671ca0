// inflate_threads.h -- internal interface between the host decoder (inflate_host.cpp), the multi-threaded
// single-stream decode (inflate_threads.cpp) and the device side that consumes its parts (inflate_many.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "../../include/zng_rocm.h"

struct ZrDecodeCtl {                 // control block of a partial decode: see inflate_host.cpp
    uint64_t        start_bit;
    const uint64_t *stops;
    size_t          nstops;
    size_t          size_hint;
    uint64_t        end_bit;
    uint64_t        max_reach;
    int             hit_stop;
};

struct ZrPart {                      // one part of the stream: its token arrays are carried over between calls
    zng_rocm_inflate_tokens tk;
    size_t      caps[3];
    ZrDecodeCtl ctl;
    int         status;
};

struct ZrThreadsResult {
    int         status;
    const char *msg;
    uint64_t    out_len;
    size_t      in_used;
};

enum { ZR_THREADS_OK = 0, ZR_THREADS_SEQUENTIAL = 100 };     // SEQUENTIAL: use the one-thread decoder for this stream

uint64_t zr_inflate_find_block(const uint8_t *src, size_t src_len, uint64_t from_bit, uint64_t to_bit);
int zr_inflate_decode_part(const uint8_t *src, size_t src_len, zng_rocm_inflate_tokens *t, size_t caps[3],
                           void *(*re)(void *, size_t, size_t), ZrDecodeCtl *ctl);
// parts[0 .. max_parts): scratch the caller owns (arrays grow through `re`, nullptr = realloc); on ZR_THREADS_OK
// `chain` lists the parts that make up the stream, in order
int zr_inflate_decode_threads(const uint8_t *src, size_t src_len, uint32_t window_len, unsigned nthreads,
                              void *(*re)(void *, size_t, size_t), ZrPart *parts, size_t max_parts,
                              std::vector<size_t> *chain, ZrThreadsResult *res);
void zr_inflate_note_parts(int n);        // for zng_rocm_inflate_threads_last_parts()
unsigned zr_default_threads();            // hardware threads, capped by the control group's CPU quota
