// checksum.hip -- host launcher of the streaming Adler-32 / CRC-32 kernels (checksum_kernel.h) and their
// *_dev entry points; the combine identities of adler32.c:32-54 / crc32_braid_comb.c:16-18 are evaluated on the
// device by the finalize kernel (same header) and by slots.hip's combine kernels.
// (A single-launch variant -- device-scope accumulators + arrival ticket -- was measured and rejected: the
// dependent device-scope atomics cost ~6 us per call, more than the 3-4 us of the second launch.)
#include "checksum_kernel.h"

#include <atomic>

namespace zr {

// ---- host launcher ----------------------------------------------------------
// CUs the persistent checksum grid leaves alone (zng_rocm_reserve_cus): a workgroup of this kernel fills its CU, so
// a kernel of another stream -- an RCCL collective, a combine -- that lands on one of them stalls that CU's whole
// share of the pass, and with it the pass.  With a few CUs left free the other stream has somewhere to go.
static std::atomic<int> g_reserved_cus{0};

static int pick_groups(const Context *c, long long nunits) {
    long long g = c->cus - g_reserved_cus.load(std::memory_order_relaxed);   // one 1024-thread workgroup per CU
    if (g < 1) g = 1;
    if (g > kMaxGroups) g = kMaxGroups;
    if (nunits < g) g = nunits > 0 ? nunits : 1;
    return (int)g;
}

void checksum_reset_reserved_cus() { g_reserved_cus.store(0, std::memory_order_relaxed); }

int launch_checksum(bool do_adler, bool do_crc, uint32_t adler, uint32_t crc, const void *d_buf, void *d_dst,
                    size_t len, uint32_t *d_out_adler, uint32_t *d_out_crc, hipStream_t stream,
                    const uint32_t *d_seed_adler, const uint32_t *d_seed_crc) {
    Context *c = ctx();
    if (!c) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    DeviceGuard dev;
    if ((!d_buf && len) || (do_adler && !d_out_adler) || (do_crc && !d_out_crc)) {
        set_error("null device pointer");
        return ZNG_ROCM_EINVAL;
    }
    if (len >> 34) {
        set_error("length above 16 GiB per call is not supported");
        return ZNG_ROCM_EINVAL;
    }
    Workspace *ws = workspace_for(stream);
    if (!ws) return ZNG_ROCM_ENOMEM;

    const uintptr_t p = (uintptr_t)d_buf;
    const uintptr_t a0 = p & ~(uintptr_t)15;
    const uintptr_t tail_base = (p + len) & ~(uintptr_t)15;
    const bool copy = d_dst != nullptr;
    if (copy && (((uintptr_t)d_dst ^ p) & 15)) {
        set_error("fold_copy needs src and dst with the same address mod 16");
        return ZNG_ROCM_EINVAL;
    }

    StreamArgs sa;
    sa.a0 = (const uint8_t *)a0;
    sa.dst0 = copy ? (uint8_t *)d_dst - (p - a0) : nullptr;
    sa.n = (long long)len;
    sa.body = (long long)(tail_base - a0);
    sa.nunits = (sa.body + kUnitBytes - 1) / kUnitBytes;
    sa.head = (int)(p - a0);
    sa.tail = (int)((p + len) - tail_base);
    sa.phase_stamps = nullptr;
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 8; ++i) {
            sa.bits.stride[k][i] = c->host_tables.stride_tab[k][1u << i];
            sa.bits.x32[k][i] = c->host_tables.x32_tab[k][1u << i];
        }

    FinalArgs fa;
    fa.tail_base = (const uint8_t *)tail_base;
    fa.tail_dst = copy ? (uint8_t *)d_dst + (tail_base - p) : nullptr;   // may point before d_dst when body == 0
    fa.n = sa.n;
    fa.nunits = sa.nunits;
    fa.tail_lo = sa.body == 0 ? sa.head : 0;
    fa.tail_hi = sa.tail;
    if (len == 0) fa.tail_lo = fa.tail_hi = 0;
    fa.adler_seed = adler;
    fa.crc_seed = crc;
    fa.crc_len_pow = do_crc ? xpow_bytes(c->host_tables.pow_tab, (uint64_t)len) : 0u;
    fa.adler_seed_ptr = d_seed_adler;
    fa.crc_seed_ptr = d_seed_crc;
    fa.do_adler = do_adler;
    fa.do_crc = do_crc;

    const int groups = pick_groups(c, sa.nunits);
    fa.groups = groups;
    if (sa.nunits > 0) {
        dim3 grid(groups), block(kWgThreads);
#define ZR_LAUNCH(A, C, K) \
        ZR_LAUNCH_TRACED((stream_kernel<A, C, K>), grid, block, stream, sa, c->tables, ws->partials)
        if (copy) {
            if (do_adler && do_crc) ZR_LAUNCH(true, true, true);
            else if (do_adler) ZR_LAUNCH(true, false, true);
            else ZR_LAUNCH(false, true, true);
        } else {
            if (do_adler && do_crc) ZR_LAUNCH(true, true, false);
            else if (do_adler) ZR_LAUNCH(true, false, false);
            else ZR_LAUNCH(false, true, false);
        }
#undef ZR_LAUNCH
        ZR_HIP(hipGetLastError());
    } else {
        ZR_HIP(hipMemsetAsync(ws->partials, 0, sizeof(Partial) * groups, stream));
    }
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, stream, fa, c->tables, ws->partials,
                       d_out_adler, d_out_crc);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

int launch_checksum_batch_device(bool do_adler, bool do_crc, const StreamArgs *d_messages, const FinalArgs *d_finals,
                                 Partial *d_part, size_t rows, uint32_t *d_out2, hipStream_t stream) {
    Context *c = ctx();
    if (!c) return ZNG_ROCM_ENODEV;
    if (!rows) return ZNG_ROCM_OK;
    for (size_t first = 0; first < rows; first += 32768) {
        const size_t n = rows - first < 32768 ? rows - first : 32768;
        dim3 grid(1, (unsigned)n), block(kWgThreads);
        if (do_adler && do_crc)
            ZR_LAUNCH_TRACED((stream_kernel_batch<true, true>), grid, block, stream, d_messages + first, c->tables, d_part + first);
        else if (do_adler)
            ZR_LAUNCH_TRACED((stream_kernel_batch<true, false>), grid, block, stream, d_messages + first, c->tables, d_part + first);
        else
            ZR_LAUNCH_TRACED((stream_kernel_batch<false, true>), grid, block, stream, d_messages + first, c->tables, d_part + first);
        ZR_HIP(hipGetLastError());
        hipLaunchKernelGGL(finalize_kernel_batch, dim3((unsigned)n), dim3(256), 0, stream, d_finals + first, c->tables,
                           d_part + first, d_out2 + 2 * first);
        ZR_HIP(hipGetLastError());
    }
    return ZNG_ROCM_OK;
}

}  // namespace zr

using namespace zr;

extern "C" {

int zng_rocm_adler32_dev(uint32_t adler, const void *d_buf, size_t len, uint32_t *d_out, void *stream) {
    return launch_checksum(true, false, adler, 0, d_buf, nullptr, len, d_out, nullptr, (hipStream_t)stream);
}

int zng_rocm_crc32_dev(uint32_t crc, const void *d_buf, size_t len, uint32_t *d_out, void *stream) {
    return launch_checksum(false, true, 0, crc, d_buf, nullptr, len, nullptr, d_out, (hipStream_t)stream);
}

int zng_rocm_adler32_crc32_dev(uint32_t adler, uint32_t crc, const void *d_buf, size_t len, uint32_t *d_out2,
                               void *stream) {
    if (!d_out2) return ZNG_ROCM_EINVAL;
    return launch_checksum(true, true, adler, crc, d_buf, nullptr, len, d_out2, d_out2 + 1, (hipStream_t)stream);
}

int zng_rocm_fold_copy_dev(int which, uint32_t adler, uint32_t crc, void *d_dst, const void *d_src, size_t len,
                           uint32_t *d_out2, void *stream) {
    if (which < 1 || which > 3 || !d_out2 || (!d_dst && len)) return ZNG_ROCM_EINVAL;
    if (len == 0) d_dst = nullptr;
    return launch_checksum((which & 1) != 0, (which & 2) != 0, adler, crc, d_src, d_dst, len, d_out2, d_out2 + 1,
                           (hipStream_t)stream);
}

int zng_rocm_reserve_cus(int n) {
    Context *c = ctx();
    if (!c) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    if (n < 0 || n >= c->cus) return ZNG_ROCM_EINVAL;
    g_reserved_cus.store(n, std::memory_order_relaxed);
    return ZNG_ROCM_OK;
}

// Many messages in one pass: the many-stream form of the checksum slots (a pigz-style job checks every block / stream,
// pigz.c; per-stream launches would be launch-bound: two launches per message).  Grid row m of the streaming kernel is
// message m (its descriptor read from an array instead of the kernel arguments), one workgroup per message up to
// 16 MiB, more above; one finalize workgroup per message.  which: 1 = Adler-32, 2 = CRC-32, 3 = both.
// d_out2: two words per message (adler, crc; the one not asked for is left untouched).
int zng_rocm_checksums_dev(int which, const zng_rocm_check_job *jobs, size_t njobs, uint32_t *d_out2, void *stream) {
    Context *c = ctx();
    if (!c) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    if (!njobs) return ZNG_ROCM_OK;
    if (which < 1 || which > 3 || !jobs || !d_out2) return ZNG_ROCM_EINVAL;
    const bool do_adler = (which & 1) != 0, do_crc = (which & 2) != 0;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard dev;
    Workspace *ws = workspace_for(st);
    if (!ws) return ZNG_ROCM_ENOMEM;
    std::lock_guard<std::mutex> use(ws->mu);
    size_t max_units = 1;
    for (size_t i = 0; i < njobs; ++i) {
        if ((!jobs[i].buf && jobs[i].len) || (jobs[i].len >> 34)) {
            set_error("job %zu: null buffer or more than 16 GiB", i);
            return ZNG_ROCM_EINVAL;
        }
        const size_t u = (size_t)((jobs[i].len + 15 + kUnitBytes) / kUnitBytes);
        if (u > max_units) max_units = u;
    }
    // workgroups per message: one up to 1024 units (16 MiB), more above, at most 16
    size_t G = (max_units + 1023) / 1024;
    if (G > 16) G = 16;             // every message of the call gets G workgroups: one very long message among many
                                    //   short ones must not multiply the grid (it belongs to zng_rocm_*_dev anyway)
    constexpr size_t kRows = 32768;                     // grid.y per launch
    const size_t per = sizeof(StreamArgs) + sizeof(FinalArgs);
    const size_t rows_max = njobs < kRows ? njobs : kRows;
    uint8_t *d_msg = nullptr, *h_msg = nullptr;
    Partial *d_part = nullptr;
    if (int rc = scratch_reserve(ws, kScrCheckMessages, rows_max * per, false, (void **)&d_msg)) return rc;
    if (int rc = scratch_reserve(ws, kScrCheckPartials, rows_max * G * sizeof(Partial), false, (void **)&d_part)) return rc;
    for (size_t first = 0; first < njobs; first += kRows) {
        const size_t rows = njobs - first < kRows ? njobs - first : kRows;
        if (int rc = host_tables_acquire(ws)) return rc;
        if (int rc = scratch_reserve(ws, kScrCheckMessagesHost, rows_max * per, true, (void **)&h_msg)) return rc;
        StreamArgs *sa = reinterpret_cast<StreamArgs *>(h_msg);
        FinalArgs *fa = reinterpret_cast<FinalArgs *>(h_msg + rows * sizeof(StreamArgs));
        for (size_t r = 0; r < rows; ++r) {
            const zng_rocm_check_job &j = jobs[first + r];
            const uintptr_t p = (uintptr_t)j.buf;
            const uintptr_t a0 = p & ~(uintptr_t)15;
            const uintptr_t tail_base = (p + j.len) & ~(uintptr_t)15;
            StreamArgs &s = sa[r];
            s.a0 = (const uint8_t *)a0;
            s.dst0 = nullptr;
            s.n = (long long)j.len;
            s.body = (long long)(tail_base - a0);
            s.nunits = (s.body + kUnitBytes - 1) / kUnitBytes;
            s.head = (int)(p - a0);
            s.tail = (int)((p + j.len) - tail_base);
            s.phase_stamps = nullptr;
            for (int k = 0; k < 4; ++k)
                for (int i = 0; i < 8; ++i) {
                    s.bits.stride[k][i] = c->host_tables.stride_tab[k][1u << i];
                    s.bits.x32[k][i] = c->host_tables.x32_tab[k][1u << i];
                }
            FinalArgs &f = fa[r];
            f.tail_base = (const uint8_t *)tail_base;
            f.tail_dst = nullptr;
            f.n = s.n;
            f.nunits = s.nunits;
            f.tail_lo = s.body == 0 ? s.head : 0;
            f.tail_hi = s.tail;
            if (j.len == 0) f.tail_lo = f.tail_hi = 0;
            f.groups = (int)G;
            f.adler_seed = j.adler;
            f.crc_seed = j.crc;
            f.crc_len_pow = do_crc ? xpow_bytes(c->host_tables.pow_tab, (uint64_t)j.len) : 0u;
            f.adler_seed_ptr = nullptr;
            f.crc_seed_ptr = nullptr;
            f.do_adler = do_adler;
            f.do_crc = do_crc;
        }
        ZR_HIP(hipMemcpyAsync(d_msg, h_msg, rows * per, hipMemcpyHostToDevice, st));
        if (int rc = host_tables_release(ws, st)) return rc;
        const StreamArgs *d_sa = reinterpret_cast<const StreamArgs *>(d_msg);
        const FinalArgs *d_fa = reinterpret_cast<const FinalArgs *>(d_msg + rows * sizeof(StreamArgs));
        dim3 grid((unsigned)G, (unsigned)rows), block(kWgThreads);
        if (do_adler && do_crc) ZR_LAUNCH_TRACED((stream_kernel_batch<true, true>), grid, block, st, d_sa, c->tables, d_part);
        else if (do_adler) ZR_LAUNCH_TRACED((stream_kernel_batch<true, false>), grid, block, st, d_sa, c->tables, d_part);
        else ZR_LAUNCH_TRACED((stream_kernel_batch<false, true>), grid, block, st, d_sa, c->tables, d_part);
        ZR_HIP(hipGetLastError());
        hipLaunchKernelGGL(finalize_kernel_batch, dim3((unsigned)rows), dim3(256), 0, st, d_fa, c->tables, d_part,
                           d_out2 + 2 * first);
        ZR_HIP(hipGetLastError());
    }
    return ZNG_ROCM_OK;
}

}  // extern "C"
