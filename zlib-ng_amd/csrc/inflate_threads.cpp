// inflate_threads.cpp -- ONE raw deflate stream decoded on several host threads (host code only).
//
// The Huffman decode of a stream is sequential: where a code starts depends on every code before it
// (inffast_tpl.h:151-298), and a block header carries no marker.  What makes threads possible here is the token
// architecture of this backend's inflate (inflate_host.cpp): a decoder that starts in the middle of the stream does
// not need the bytes in front of it -- it emits "length, distance" tokens, and the device resolves every distance
// afterwards (inflate_resolve.hip).  So:
//   1. the compressed bytes are cut into parts; every thread searches its part for the first bit position at which a
//      block the decoder would accept starts (zr_inflate_find_block: a valid dynamic header or a sync-flush marker);
//   2. every part is decoded from its candidate until a block ends EXACTLY on a later candidate (or the stream ends);
//   3. the parts are chained from bit 0: part 0 is genuine, the part that starts where it ended is therefore genuine
//      too, and so on; parts that are not on the chain (their candidate was noise, or an earlier part ran across
//      them) are dropped.  Only then is each part's reach into history checked against what really precedes it.
// Anything irregular on the chain -- a data error, a truncated stream, a distance too far back -- sends the call to
// the plain sequential decoder, which then reports exactly what the reference would (status, message, bytes produced).
// The scheme is the one pugz / rapidgzip use for gzip files; streams whose blocks are all fixed-Huffman or stored
// offer no candidates and decode on one thread.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/zng_rocm.h"
#include "inflate_threads.h"

namespace {

template <typename F>
void parallel_for(size_t n, unsigned nthreads, F f) {
    if (nthreads > n) nthreads = (unsigned)n;
    if (nthreads <= 1) {
        for (size_t i = 0; i < n; ++i) f(i);
        return;
    }
    std::atomic<size_t> next{0};
    auto body = [&] {
        for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n) return;
            f(i);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(nthreads - 1);
    for (unsigned k = 1; k < nthreads; ++k) pool.emplace_back(body);
    body();
    for (auto &t : pool) t.join();
}

}  // namespace

int zr_inflate_decode_threads(const uint8_t *src, size_t src_len, uint32_t window_len, unsigned nthreads,
                              void *(*re)(void *, size_t, size_t), ZrPart *parts, size_t max_parts,
                              std::vector<size_t> *chain, ZrThreadsResult *res) {
    chain->clear();
    // parts of >= 256 KiB of compressed bytes, several per thread (a part whose candidate is noise costs its thread
    // little, and the thread that runs across it has less to make up)
    size_t k = src_len / (256u << 10);
    if (k > 8u * (size_t)nthreads) k = 8u * (size_t)nthreads;
    if (k > max_parts) k = max_parts;
    if (k < 2 || nthreads < 2) return ZR_THREADS_SEQUENTIAL;

    // 1. candidates
    std::vector<uint64_t> start(k, ~0ull);
    start[0] = 0;
    parallel_for(k - 1, nthreads, [&](size_t i) {
        const size_t part = i + 1;
        const uint64_t lo = 8ull * (uint64_t)(src_len * part / k), hi = 8ull * (uint64_t)(src_len * (part + 1) / k);
        start[part] = zr_inflate_find_block(src, src_len, lo, hi);
    });
    std::vector<uint64_t> stops;
    std::vector<size_t> live;                 // parts that have a candidate
    for (size_t i = 0; i < k; ++i)
        if (start[i] != ~0ull) {
            live.push_back(i);
            if (i) stops.push_back(start[i]);
        }
    if (live.size() < 2) return ZR_THREADS_SEQUENTIAL;

    // 2. every part from its candidate to the next candidate a block ends on
    parallel_for(live.size(), nthreads, [&](size_t li) {
        ZrPart &p = parts[live[li]];
        memset(&p.ctl, 0, sizeof(p.ctl));
        p.ctl.start_bit = start[live[li]];
        p.ctl.stops = stops.data();
        p.ctl.nstops = stops.size();
        p.ctl.size_hint = src_len / k + (src_len / k >> 2);
        p.status = zr_inflate_decode_part(src, src_len, &p.tk, p.caps, re, &p.ctl);
    });

    // 3. the chain from bit 0
    uint64_t produced = 0;
    size_t cur = 0;
    for (;;) {
        ZrPart &p = parts[cur];
        if (p.status == -4) return -4;
        const bool ended = p.status == 1;
        if (!(ended || (p.status == 0 && p.ctl.hit_stop))) return ZR_THREADS_SEQUENTIAL;      // error / truncation on the chain
        if (p.ctl.max_reach > produced + window_len) return ZR_THREADS_SEQUENTIAL;            // "invalid distance too far back"
        chain->push_back(cur);
        produced += p.tk.out_len;
        if (ended) {
            res->status = 1;
            res->msg = "";
            res->out_len = produced;
            res->in_used = (size_t)((p.ctl.end_bit + 7) >> 3);
            return ZR_THREADS_OK;
        }
        // the part that starts where this one stopped
        const auto it = std::lower_bound(live.begin(), live.end(), p.ctl.end_bit,
                                         [&](size_t part, uint64_t bit) { return start[part] < bit; });
        if (it == live.end() || start[*it] != p.ctl.end_bit) return ZR_THREADS_SEQUENTIAL;    // cannot happen: it stopped on a stop
        cur = *it;
    }
}

// threads for "as many as the host gives us": the hardware threads, capped by the CPU quota of the control group the
// process runs in (a container with 16 CPUs' worth of quota on a 256-thread host gains nothing from 256 threads)
unsigned zr_default_threads() {
    unsigned t = std::thread::hardware_concurrency();
    if (t == 0) t = 1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = "";
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && period > 0 && strcmp(quota, "max") != 0) {
            const long q = atol(quota);
            if (q > 0) {
                const unsigned cap = (unsigned)((q + period - 1) / period);
                if (cap && cap < t) t = cap;
            }
        }
        fclose(f);
    }
    return t;
}

static thread_local int t_last_parts = 0;     // parts on the chain of this thread's last multi-threaded decode (0: one thread did it)
void zr_inflate_note_parts(int n) { t_last_parts = n; }

extern "C" {

int zng_rocm_inflate_threads_last_parts(void) { return t_last_parts; }

int zng_rocm_inflate_tokens_decode_threads(const uint8_t *src, size_t src_len, uint32_t window_len, int nthreads,
                                           zng_rocm_inflate_tokens *out) {
    if (!out || (!src && src_len) || window_len > 32768u) return ZNG_ROCM_EINVAL;
    const unsigned t = nthreads > 0 ? (unsigned)nthreads : zr_default_threads();
    const size_t max_parts = 8u * (size_t)t;
    std::vector<ZrPart> parts(max_parts);
    for (ZrPart &p : parts) memset(&p, 0, sizeof(p));
    std::vector<size_t> chain;
    ZrThreadsResult res;
    const int rc = zr_inflate_decode_threads(src, src_len, window_len, t, nullptr, parts.data(), max_parts, &chain, &res);
    auto free_parts = [&] {
        for (ZrPart &p : parts) zng_rocm_inflate_tokens_free(&p.tk);
    };
    t_last_parts = rc == ZR_THREADS_OK ? (int)chain.size() : 0;
    if (rc != ZR_THREADS_OK) {
        free_parts();
        if (rc == -4) {
            memset(out, 0, sizeof(*out));
            out->status = -4;
            out->msg = "out of memory";
            return -4;
        }
        return zng_rocm_inflate_tokens_decode_window(src, src_len, window_len, out);     // the sequential decoder
    }
    // join the parts on the chain into one token stream.  A part's first segment is merged into the last segment of
    // the part before it (that one may hold less than 32 KiB; every segment but the stream's last must hold more).
    memset(out, 0, sizeof(*out));
    size_t ntok = 0, nlit = 0, nseg = 0;
    for (size_t c = 0; c < chain.size(); ++c) {
        const zng_rocm_inflate_tokens &tk = parts[chain[c]].tk;
        ntok += tk.ntokens;
        nlit += tk.nliterals;
        nseg += tk.nsegs - (c ? 1 : 0);
    }
    out->tokens = (uint32_t *)malloc((ntok + 1) * 4);
    out->literals = (uint8_t *)malloc(nlit + 1);
    out->segs = (uint64_t *)malloc((nseg + 1) * 24);
    if (!out->tokens || !out->literals || !out->segs) {
        free_parts();
        zng_rocm_inflate_tokens_free(out);
        out->status = -4;
        out->msg = "out of memory";
        return -4;
    }
    uint64_t tok0 = 0, lit0 = 0, o0 = 0;
    size_t s0 = 0;
    for (size_t c = 0; c < chain.size(); ++c) {
        const zng_rocm_inflate_tokens &tk = parts[chain[c]].tk;
        memcpy(out->tokens + tok0, tk.tokens, tk.ntokens * 4);
        memcpy(out->literals + lit0, tk.literals, tk.nliterals);
        for (size_t s = c ? 1 : 0; s < tk.nsegs; ++s, ++s0) {
            out->segs[3 * s0] = tk.segs[3 * s] + tok0;
            out->segs[3 * s0 + 1] = tk.segs[3 * s + 1] + o0;
            out->segs[3 * s0 + 2] = tk.segs[3 * s + 2] + lit0;
        }
        tok0 += tk.ntokens;
        lit0 += tk.nliterals;
        o0 += tk.out_len;
    }
    out->segs[3 * s0] = tok0;
    out->segs[3 * s0 + 1] = o0;
    out->segs[3 * s0 + 2] = lit0;
    out->ntokens = (size_t)tok0;
    out->nliterals = (size_t)lit0;
    out->nsegs = s0;
    out->out_len = res.out_len;
    out->in_used = res.in_used;
    out->status = res.status;
    out->msg = res.msg;
    free_parts();
    return out->status;
}

}  // extern "C"
