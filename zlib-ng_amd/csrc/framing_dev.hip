// framing_dev.hip -- zlib (RFC 1950) and gzip (RFC 1952) framing for MANY device-resident streams: the compress2 /
// uncompress2 class front ends (compress.c:31-69, uncompr.c:25-76) in the shape of the reference's many-stream model
// (test/pigz/CMakeLists.txt:123-200), with nothing on the host between the kernels:
//   compress    zng_rocm_deflate_quick_dev (level-1 class) -> [CRC-32 of every plaintext in one pass, gzip only]
//               -> one small kernel writes every header and trailer (deflate.c:868-892 / :902-1031, :1091-1103)
//   uncompress  one small kernel parses every header (inflate.c:509-555 zlib, :556-700 gzip incl. FHCRC) and patches
//               the job table -> inflate_streams_kernel -> the checksum descriptors of the outputs are filled ON THE
//               DEVICE (only it knows the lengths) -> many-message checksum pass -> one small kernel compares the
//               trailers (inflate.c:1105-1147: "incorrect data check", "incorrect length check")
// `format`: 0 = raw, 1 = zlib, 2 = gzip (as oneshot.hip).
//
// The level-1 class writes its block at a 4-byte aligned address, so both headers are 12 bytes long: gzip declares an
// empty FEXTRA field (XLEN = 0), zlib puts two empty stored blocks (the 5-byte Z_SYNC_FLUSH marker, deflate.c:1064-1076,
// twice) between its 2-byte header and the block -- valid RFC 1951 that every inflater skips.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "checksum_args.h"
#include "context.h"
#include "gf2.h"
#include "inflate_dev.h"

extern "C" size_t zng_rocm_deflate_quick_bound(size_t source_len);
extern "C" int zng_rocm_deflate_quick_dev(const zng_rocm_stream_job *jobs, size_t njobs, uint32_t *d_results, void *stream);
extern "C" int zng_rocm_checksums_dev(int which, const zng_rocm_check_job *jobs, size_t njobs, uint32_t *d_out2, void *stream);

namespace zr {

constexpr uint32_t kWrapHead = 12;                       // both wrapped formats (see above)
__host__ __device__ inline uint32_t wrap_tail(int format) { return format == 1 ? 4u : format == 2 ? 8u : 0u; }

struct FrameJob {
    uint8_t *out;
    uint64_t in_len;
};

__global__ __launch_bounds__(256)
void frame_compress_kernel(const FrameJob *__restrict__ jobs, uint32_t njobs, int format, const uint32_t *__restrict__ quick,
                           const uint32_t *__restrict__ checks, uint32_t *__restrict__ results) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= njobs) return;
    uint8_t *out = jobs[i].out;
    const uint32_t clen = quick[2 * i], adler = quick[2 * i + 1];
    if (format == 0) {
        results[2 * i] = clen;
        results[2 * i + 1] = adler;
        return;
    }
    static const uint8_t zhead[12] = {0x78, 0x01, 0x00, 0x00, 0x00, 0xff, 0xff, 0x00, 0x00, 0x00, 0xff, 0xff};
    // ID1 ID2 CM=8 FLG=FEXTRA MTIME=0 XFL=4 (fastest, deflate.c:913-914) OS=3 (Unix) XLEN=0
    static const uint8_t ghead[12] = {0x1f, 0x8b, 0x08, 0x04, 0x00, 0x00, 0x00, 0x00, 0x04, 0x03, 0x00, 0x00};
    for (int k = 0; k < 12; ++k) out[k] = format == 1 ? zhead[k] : ghead[k];
    uint8_t *t = out + kWrapHead + clen;
    if (format == 1) {                                   // Adler-32, most significant byte first (deflate.c:1098-1101)
        t[0] = (uint8_t)(adler >> 24); t[1] = (uint8_t)(adler >> 16); t[2] = (uint8_t)(adler >> 8); t[3] = (uint8_t)adler;
        results[2 * i + 1] = adler;
    } else {                                             // CRC-32 and ISIZE, least significant byte first (deflate.c:1091-1096)
        const uint32_t crc = checks[2 * i + 1], isize = (uint32_t)jobs[i].in_len;
        for (int k = 0; k < 4; ++k) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)(isize >> (8 * k)); }
        results[2 * i + 1] = crc;
    }
    results[2 * i] = kWrapHead + clen + wrap_tail(format);
}

// ---- uncompress side ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void parse_header_kernel(const InflateJobDev *__restrict__ given, uint32_t njobs, int format,
                         const DeviceTables *__restrict__ tabs, InflateJobDev *__restrict__ patched,
                         uint32_t *__restrict__ head /* 2 per job: header bytes, message id */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= njobs) return;
    const InflateJobDev j = given[i];
    const uint8_t *in = j.in;
    const uint64_t n = j.in_len;
    uint64_t pos = 0;
    uint32_t msg = kMsgNone;
    if (format == 1) {                                   // inflate.c:509-555 with windowBits 15
        if (n < 2) {
            msg = kMsgStarved;
        } else {
            const uint32_t cmf = in[0], flg = in[1];
            if (((cmf << 8) | flg) % 31u) msg = kMsgHeaderCheck;
            else if ((cmf & 15u) != 8u) msg = kMsgMethod;
            else if ((cmf >> 4) + 8u > 15u) msg = kMsgWindow;
            else if (flg & 0x20u) msg = kMsgNeedDict;    // a preset dictionary: Z_DATA_ERROR for the one-shot caller (uncompr.c:70-75)
            pos = 2;
        }
    } else if (format == 2) {                            // inflate.c:556-700
        if (n < 10) {
            msg = kMsgStarved;
        } else if (in[0] != 0x1f || in[1] != 0x8b) {
            msg = kMsgHeaderCheck;
        } else if (in[2] != 8) {
            msg = kMsgMethod;
        } else if (in[3] & 0xe0u) {
            msg = kMsgHeaderCheck;                       // "unknown header flags set": reported as a header check failure
        } else {
            const uint32_t flags = in[3];
            pos = 10;
            if (flags & 4u) {                            // FEXTRA
                if (pos + 2 > n) msg = kMsgStarved;
                else pos += 2u + (in[pos] | ((uint32_t)in[pos + 1] << 8));
            }
            for (uint32_t bit = 8; bit <= 16 && msg == kMsgNone; bit <<= 1) {      // FNAME, FCOMMENT: zero-terminated
                if (!(flags & bit)) continue;
                while (pos < n && in[pos]) ++pos;
                if (pos >= n) msg = kMsgStarved;
                else ++pos;
            }
            if (msg == kMsgNone && (flags & 2u)) {       // FHCRC: the low 16 bits of the CRC-32 of the header so far
                if (pos + 2 > n) {
                    msg = kMsgStarved;
                } else {
                    uint32_t c = 0xffffffffu;
                    for (uint64_t k = 0; k < pos; ++k) c = tabs->byte_tab[(c ^ in[k]) & 0xffu] ^ (c >> 8);
                    c = ~c;
                    if ((c & 0xffffu) != (in[pos] | ((uint32_t)in[pos + 1] << 8))) msg = kMsgHeaderCrc;
                    pos += 2;
                }
            }
            if (msg == kMsgNone && pos > n) msg = kMsgStarved;
        }
    }
    InflateJobDev p = j;
    p.dict_len = 0;
    if (msg != kMsgNone) {                               // nothing to decode: an empty job costs the inflater nothing
        p.in_len = 0;
        p.out_cap = 0;
    } else {
        p.in = in + pos;
        p.in_len = n - pos;
    }
    patched[i] = p;
    head[2 * i] = (uint32_t)pos;
    head[2 * i + 1] = msg;
}

// the checksum descriptors of the inflated streams, as checksum.hip's host code builds them -- but from lengths that
// only exist on the device
__global__ __launch_bounds__(256)
void fill_check_args_kernel(const InflateJobDev *__restrict__ jobs, const uint32_t *__restrict__ inflated, uint32_t njobs,
                            const DeviceTables *__restrict__ tabs, int do_adler, int do_crc, StreamArgs *__restrict__ sa,
                            FinalArgs *__restrict__ fa) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= njobs) return;
    const uint64_t len = (int32_t)inflated[4 * i + 2] == 1 ? inflated[4 * i] : 0u;
    const uintptr_t p = (uintptr_t)jobs[i].out;
    const uintptr_t a0 = p & ~(uintptr_t)15, tail_base = (p + len) & ~(uintptr_t)15;
    StreamArgs s;
    s.a0 = (const uint8_t *)a0;
    s.dst0 = nullptr;
    s.n = (long long)len;
    s.body = (long long)(tail_base - a0);
    s.nunits = (s.body + kUnitBytes - 1) / kUnitBytes;
    s.head = (int)(p - a0);
    s.tail = (int)((p + len) - tail_base);
    s.phase_stamps = nullptr;
    for (int k = 0; k < 4; ++k)
        for (int b = 0; b < 8; ++b) {
            s.bits.stride[k][b] = tabs->stride_tab[k][1u << b];
            s.bits.x32[k][b] = tabs->x32_tab[k][1u << b];
        }
    sa[i] = s;
    FinalArgs f;
    f.tail_base = (const uint8_t *)tail_base;
    f.tail_dst = nullptr;
    f.n = s.n;
    f.nunits = s.nunits;
    f.tail_lo = s.body == 0 ? s.head : 0;
    f.tail_hi = s.tail;
    if (len == 0) f.tail_lo = f.tail_hi = 0;
    f.groups = 1;
    f.adler_seed = 1;
    f.crc_seed = 0;
    f.crc_len_pow = do_crc ? xpow_bytes(tabs->pow_tab, len) : 0u;
    f.adler_seed_ptr = nullptr;
    f.crc_seed_ptr = nullptr;
    f.do_adler = do_adler;
    f.do_crc = do_crc;
    fa[i] = f;
}

__global__ __launch_bounds__(256)
void verify_trailer_kernel(const InflateJobDev *__restrict__ given, const uint32_t *__restrict__ head,
                           const uint32_t *__restrict__ checks, uint32_t njobs, int format, uint32_t *__restrict__ results) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= njobs) return;
    uint32_t out_len = results[4 * i], used = results[4 * i + 1], status = results[4 * i + 2], msg = results[4 * i + 3];
    const uint32_t hdr = head[2 * i], hmsg = head[2 * i + 1];
    if (hmsg != kMsgNone) {
        out_len = 0;
        used = 0;
        msg = hmsg;
        status = hmsg == kMsgStarved ? (uint32_t)-5 : (uint32_t)-3;
    } else if ((int32_t)status == 1) {
        const uint32_t tail = wrap_tail(format);
        const uint64_t at = (uint64_t)hdr + used;
        if (at + tail > given[i].in_len) {               // the stream ends before its trailer
            status = (uint32_t)-5;
            msg = kMsgStarved;
            used = (uint32_t)given[i].in_len;
        } else {
            const uint8_t *t = given[i].in + at;
            if (format == 1) {
                const uint32_t stored = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
                if (stored != checks[2 * i]) { status = (uint32_t)-3; msg = kMsgDataCheck; }
            } else if (format == 2) {
                const uint32_t crc = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
                const uint32_t isize = t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
                if (crc != checks[2 * i + 1]) { status = (uint32_t)-3; msg = kMsgDataCheck; }
                else if (isize != out_len) { status = (uint32_t)-3; msg = kMsgLengthCheck; }
            }
            used = (uint32_t)(at + tail);
        }
    } else {
        used += hdr;
    }
    results[4 * i] = out_len;
    results[4 * i + 1] = used;
    results[4 * i + 2] = status;
    results[4 * i + 3] = msg;
}

}  // namespace zr

using namespace zr;

extern "C" {

size_t zng_rocm_compress_streams_bound(size_t source_len, int format) {
    return zng_rocm_deflate_quick_bound(source_len) + (format ? kWrapHead + wrap_tail(format) + 4 : 0);
}

int zng_rocm_compress_streams_dev(int format, const zng_rocm_stream_job *jobs, size_t njobs, uint32_t *d_results, void *stream) {
    if (!ctx()) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    if (!njobs) return ZNG_ROCM_OK;
    if (!jobs || !d_results || format < 0 || format > 2 || njobs > 0x7fffffffull) return ZNG_ROCM_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard dev;
    Workspace *ws = workspace_for(st);
    if (!ws) return ZNG_ROCM_ENOMEM;
    const uint32_t head = format ? kWrapHead : 0u, tail = wrap_tail(format);
    uint32_t *d_quick = nullptr, *d_checks = nullptr;
    FrameJob *d_fj = nullptr, *h_fj = nullptr;
    std::vector<zng_rocm_stream_job> inner(njobs);
    std::vector<zng_rocm_check_job> cj(format == 2 ? njobs : 0);
    {
        std::lock_guard<std::mutex> use(ws->mu);
        if (int rc = scratch_reserve(ws, kScrFrameWords, njobs * 4 * sizeof(uint32_t), false, (void **)&d_quick)) return rc;
        d_checks = d_quick + 2 * njobs;
        if (int rc = scratch_reserve(ws, kScrFrameJobs, njobs * sizeof(FrameJob), false, (void **)&d_fj)) return rc;
        if (int rc = host_tables_acquire(ws)) return rc;
        if (int rc = scratch_reserve(ws, kScrFrameJobsHost, njobs * sizeof(FrameJob), true, (void **)&h_fj)) return rc;
        for (size_t i = 0; i < njobs; ++i) {
            const zng_rocm_stream_job &j = jobs[i];
            if (!j.out || ((uintptr_t)j.out & 3) || j.out_cap < zng_rocm_compress_streams_bound(j.in_len, format) ||
                (format && (j.dict_len || j.flags))) {
                set_error("job %zu: out must be 4-byte aligned with out_cap >= zng_rocm_compress_streams_bound(); a wrapped "
                          "stream takes neither a dictionary nor block flags", i);
                return ZNG_ROCM_EINVAL;
            }
            inner[i] = j;
            inner[i].out = (uint8_t *)j.out + head;
            inner[i].out_cap = j.out_cap - head - tail;
            h_fj[i] = FrameJob{(uint8_t *)j.out, j.in_len};
            if (format == 2) cj[i] = zng_rocm_check_job{j.in, j.in_len, 1u, 0u};
        }
        ZR_HIP(hipMemcpyAsync(d_fj, h_fj, njobs * sizeof(FrameJob), hipMemcpyHostToDevice, st));
        if (int rc = host_tables_release(ws, st)) return rc;
    }
    // the two passes take the stream's workspace themselves
    if (int rc = zng_rocm_deflate_quick_dev(inner.data(), njobs, d_quick, st)) return rc;
    if (format == 2)
        if (int rc = zng_rocm_checksums_dev(2, cj.data(), njobs, d_checks, st)) return rc;
    hipLaunchKernelGGL(frame_compress_kernel, dim3((unsigned)((njobs + 255) / 256)), dim3(256), 0, st, d_fj, (uint32_t)njobs,
                       format, d_quick, d_checks, d_results);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

int zng_rocm_uncompress_streams_dev(int format, const zng_rocm_inflate_dev_job *jobs, size_t njobs, uint32_t *d_results,
                                    void *stream) {
    Context *c = ctx();
    if (!c) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    if (!njobs) return ZNG_ROCM_OK;
    if (!jobs || !d_results || format < 0 || format > 2 || njobs > 0x7fffffffull) return ZNG_ROCM_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard dev;
    Workspace *ws = workspace_for(st);
    if (!ws) return ZNG_ROCM_ENOMEM;
    std::lock_guard<std::mutex> use(ws->mu);
    InflateJobDev *d_given = nullptr, *h_given = nullptr, *d_patched = nullptr;
    uint32_t *d_words = nullptr;
    uint8_t *d_msg = nullptr;
    Partial *d_part = nullptr;
    if (int rc = scratch_reserve(ws, kScrInflateDevJobs, 2 * njobs * sizeof(InflateJobDev), false, (void **)&d_given)) return rc;
    d_patched = d_given + njobs;
    if (int rc = scratch_reserve(ws, kScrFrameWords, njobs * 4 * sizeof(uint32_t), false, (void **)&d_words)) return rc;
    uint32_t *d_head = d_words, *d_checks = d_words + 2 * njobs;
    if (int rc = scratch_reserve(ws, kScrCheckMessages, njobs * (sizeof(StreamArgs) + sizeof(FinalArgs)), false, (void **)&d_msg)) return rc;
    if (int rc = scratch_reserve(ws, kScrCheckPartials, njobs * sizeof(Partial), false, (void **)&d_part)) return rc;
    if (int rc = host_tables_acquire(ws)) return rc;
    if (int rc = scratch_reserve(ws, kScrInflateDevJobsHost, njobs * sizeof(InflateJobDev), true, (void **)&h_given)) return rc;
    for (size_t i = 0; i < njobs; ++i) {
        const zng_rocm_inflate_dev_job &j = jobs[i];
        if ((j.in_len && !j.in) || (j.out_cap && !j.out) || j.in_len > 0x7fffffffull || j.out_cap > 0x7fffffffull ||
            j.dict_len || j.flags) {
            set_error("job %zu: null buffer, a stream or output of 2 GiB and more, or dict_len / flags (raw streams only: "
                      "zng_rocm_inflate_streams_dev)", i);
            return ZNG_ROCM_EINVAL;
        }
        h_given[i] = InflateJobDev{(const uint8_t *)j.in, (uint8_t *)j.out, j.in_len, j.out_cap, 0u, 0u};
    }
    ZR_HIP(hipMemcpyAsync(d_given, h_given, njobs * sizeof(InflateJobDev), hipMemcpyHostToDevice, st));
    if (int rc = host_tables_release(ws, st)) return rc;
    const dim3 grid((unsigned)((njobs + 255) / 256)), block(256);
    hipLaunchKernelGGL(parse_header_kernel, grid, block, 0, st, d_given, (uint32_t)njobs, format, c->tables, d_patched, d_head);
    ZR_HIP(hipGetLastError());
    if (int rc = launch_inflate_streams_device(d_patched, njobs, d_results, st)) return rc;
    if (format) {
        StreamArgs *d_sa = reinterpret_cast<StreamArgs *>(d_msg);
        FinalArgs *d_fa = reinterpret_cast<FinalArgs *>(d_msg + njobs * sizeof(StreamArgs));
        const bool adler = format == 1, crc = format == 2;
        hipLaunchKernelGGL(fill_check_args_kernel, grid, block, 0, st, d_patched, d_results, (uint32_t)njobs, c->tables,
                           adler ? 1 : 0, crc ? 1 : 0, d_sa, d_fa);
        ZR_HIP(hipGetLastError());
        if (int rc = launch_checksum_batch_device(adler, crc, d_sa, d_fa, d_part, njobs, d_checks, st)) return rc;
    }
    hipLaunchKernelGGL(verify_trailer_kernel, grid, block, 0, st, d_given, d_head, d_checks, (uint32_t)njobs, format, d_results);
    ZR_HIP(hipGetLastError());
    return ZNG_ROCM_OK;
}

}  // extern "C"
