// oneshot.hip -- compress2 / uncompress2 class front ends over the device paths:
// zlib (RFC 1950) and gzip (RFC 1952) framing around the raw deflate / inflate kernels, with the trailer
// checksum computed ON DEVICE by the streaming checksum kernel.
//
// Reference behaviour mirrored (zlib-ng 2.2.2):
//   compress2   compress.c:31-69      header deflate.c:868-892 (zlib) / :902-1031 (gzip), trailer :1091-1103
//   uncompress2 uncompr.c:25-76       header checks inflate.c:509-555 ("incorrect header check",
//               "unknown compression method", "invalid window size"), trailer inflate.c:1105-1147
//               ("incorrect data check", "incorrect length check"); a preset dictionary or an incomplete stream
//               is Z_DATA_ERROR for the one-shot caller (uncompr.c:70-75)
// `format`: 0 = raw (windowBits -15), 1 = zlib (windowBits 15), 2 = gzip (windowBits 31).
#include "context.h"

#include <string.h>

extern "C" int zng_rocm_deflate_dev(int level, const uint8_t *d_in, size_t in_len, uint8_t *d_out, size_t out_cap,
                                    size_t *out_len, void *stream);
extern "C" size_t zng_rocm_deflate_bound(size_t source_len);
extern "C" int zng_rocm_inflate_raw_ex(const uint8_t *src, size_t src_len, uint8_t *d_dst, size_t dst_cap,
                                       uint64_t *out_len, size_t *in_used, void *stream);

namespace zr {

enum { Z_OK_ = 0, Z_STREAM_END_ = 1, Z_STREAM_ERROR_ = -2, Z_DATA_ERROR_ = -3, Z_MEM_ERROR_ = -4, Z_BUF_ERROR_ = -5 };

// adler32 / crc32 of a device buffer, synchronously (the one-shot front ends return a status)
static int device_checks(const uint8_t *d_buf, size_t len, hipStream_t st, uint32_t out[2]) {
    Workspace *ws = workspace_for(st);
    if (!ws) return ZNG_ROCM_ENOMEM;
    int rc = launch_checksum(true, true, 1u, 0u, d_buf, nullptr, len, ws->result, ws->result + 1, st);
    if (rc) return rc;
    ZR_HIP(hipMemcpyAsync(ws->pinned, ws->result, 8, hipMemcpyDeviceToHost, st));
    ZR_HIP(hipStreamSynchronize(st));
    out[0] = ws->pinned[0];
    out[1] = ws->pinned[1];
    return ZNG_ROCM_OK;
}

}  // namespace zr

using namespace zr;

extern "C" {

size_t zng_rocm_compress_bound(size_t source_len, int format) {
    // compressBound (compress.c:81-98): raw bound + wrapper (ZLIB_WRAPLEN 6, GZIP_WRAPLEN 18; zutil.h:68-69)
    return zng_rocm_deflate_bound(source_len) + (format == 1 ? 6 : format == 2 ? 18 : 0);
}

int zng_rocm_compress2_dev(uint8_t *d_dst, size_t *dst_len, const uint8_t *d_src, size_t src_len, int level,
                           int format, void *stream) {
    if (!ctx()) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    DeviceGuard dev;
    if (!d_dst || !dst_len || (!d_src && src_len) || format < 0 || format > 2) return ZNG_ROCM_EINVAL;
    if (level == -1) level = 6;                                   // Z_DEFAULT_COMPRESSION (deflate.c:296)
    if (level < 0 || level > 9) {                                 // deflateInit2: Z_STREAM_ERROR (deflate.c:318-320)
        set_error("level %d is outside -1..9", level);
        return Z_STREAM_ERROR_;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t head = format == 1 ? 2 : format == 2 ? 10 : 0;
    const size_t trail = format == 1 ? 4 : format == 2 ? 8 : 0;
    if (*dst_len < zng_rocm_compress_bound(src_len, format)) {
        set_error("destination smaller than zng_rocm_compress_bound()");
        return Z_BUF_ERROR_;
    }
    uint32_t chk[2] = {1u, 0u};
    if (format) {
        int rc = device_checks(d_src, src_len, st, chk);
        if (rc) return rc;
    }
    size_t body = 0;
    int rc = zng_rocm_deflate_dev(level, d_src, src_len, d_dst + head, *dst_len - head - trail, &body, stream);
    if (rc) return rc;
    uint8_t h[10], t[8];
    if (format == 1) {
        // deflate.c:868-885: CMF/FLG with the level hint, multiple of 31
        unsigned header = (8u + (7u << 4)) << 8;
        const unsigned level_flags = level < 2 ? 0 : level < 6 ? 1 : level == 6 ? 2 : 3;
        header |= level_flags << 6;
        header += 31 - (header % 31);
        h[0] = (uint8_t)(header >> 8);
        h[1] = (uint8_t)header;
        t[0] = (uint8_t)(chk[0] >> 24); t[1] = (uint8_t)(chk[0] >> 16); t[2] = (uint8_t)(chk[0] >> 8); t[3] = (uint8_t)chk[0];
    } else if (format == 2) {
        // deflate.c:902-916: minimal gzip header (no name/extra/comment), XFL by level (:913-914), OS 3 (Unix)
        const uint8_t gh[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, (uint8_t)(level == 9 ? 2 : level < 2 ? 4 : 0), 3};
        memcpy(h, gh, 10);
        for (int i = 0; i < 4; ++i) {
            t[i] = (uint8_t)(chk[1] >> (8 * i));                   // CRC32 then ISIZE, little endian (deflate.c:1091-1096)
            t[4 + i] = (uint8_t)((uint32_t)src_len >> (8 * i));
        }
    }
    if (head) ZR_HIP(hipMemcpyAsync(d_dst, h, head, hipMemcpyHostToDevice, st));
    if (trail) ZR_HIP(hipMemcpyAsync(d_dst + head + body, t, trail, hipMemcpyHostToDevice, st));
    ZR_HIP(hipStreamSynchronize(st));
    *dst_len = head + body + trail;
    return Z_OK_;
}

int zng_rocm_uncompress2_dev(uint8_t *d_dst, size_t *dst_len, const uint8_t *src, size_t *src_len, int format,
                             void *stream) {
    if (!ctx()) {
        set_error("zng_rocm_init() has not succeeded");
        return ZNG_ROCM_ENODEV;
    }
    DeviceGuard dev;
    if (!dst_len || !src_len || (!src && *src_len) || format < 0 || format > 2) return ZNG_ROCM_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const size_t n = *src_len;
    size_t pos = 0;
    bool gz = false;
    if (format == 1) {
        if (n < 2) { set_error("input ended inside the zlib header"); return Z_DATA_ERROR_; }
        const unsigned cmf = src[0], flg = src[1];
        if (((cmf << 8) + flg) % 31) { set_error("incorrect header check"); return Z_DATA_ERROR_; }
        if ((cmf & 15) != 8) { set_error("unknown compression method"); return Z_DATA_ERROR_; }
        if ((cmf >> 4) + 8 > 15) { set_error("invalid window size"); return Z_DATA_ERROR_; }
        if (flg & 0x20) { set_error("preset dictionary required"); return Z_DATA_ERROR_; }   // Z_NEED_DICT -> uncompr.c:72
        pos = 2;
    } else if (format == 2) {
        gz = true;
        if (n < 10) { set_error("input ended inside the gzip header"); return Z_DATA_ERROR_; }
        if (src[0] != 0x1f || src[1] != 0x8b) { set_error("incorrect header check"); return Z_DATA_ERROR_; }
        if (src[2] != 8) { set_error("unknown compression method"); return Z_DATA_ERROR_; }
        const unsigned flags = src[3];
        if (flags & 0xe0) { set_error("unknown header flags set"); return Z_DATA_ERROR_; }
        pos = 10;
        if (flags & 4) {                                         // FEXTRA
            if (pos + 2 > n) { set_error("input ended inside the gzip header"); return Z_DATA_ERROR_; }
            const size_t xlen = src[pos] | (src[pos + 1] << 8);
            pos += 2 + xlen;
        }
        for (int bit = 8; bit <= 16; bit <<= 1) {                // FNAME, FCOMMENT: zero terminated
            if (flags & bit) {
                while (pos < n && src[pos]) ++pos;
                ++pos;
            }
        }
        if (pos > n) { set_error("input ended inside the gzip header"); return Z_DATA_ERROR_; }
        if (flags & 2) {                                         // FHCRC: low 16 bits of the CRC-32 of the header so far
            if (pos + 2 > n) { set_error("input ended inside the gzip header"); return Z_DATA_ERROR_; }
            uint32_t c = 0xffffffffu;                            // inflate.c:686-692 (state->check over the header)
            const uint32_t *bt = ctx()->host_tables.byte_tab;
            for (size_t i = 0; i < pos; ++i) c = bt[(c ^ src[i]) & 0xffu] ^ (c >> 8);
            c = ~c;
            const unsigned want = src[pos] | ((unsigned)src[pos + 1] << 8);
            if (want != (c & 0xffffu)) { set_error("header crc mismatch"); return Z_DATA_ERROR_; }
            pos += 2;
        }
    }
    const size_t trail = format == 1 ? 4 : format == 2 ? 8 : 0;
    uint64_t got = 0;
    size_t used = 0;
    int rc = zng_rocm_inflate_raw_ex(src + pos, n - pos, d_dst, *dst_len, &got, &used, stream);
    if (rc == Z_BUF_ERROR_ && got > *dst_len) return Z_BUF_ERROR_;              // destination too small
    if (rc == Z_DATA_ERROR_) return Z_DATA_ERROR_;                              // message already set (strm->msg text)
    if (rc != Z_STREAM_END_) {
        if (rc == Z_OK_ || rc == Z_BUF_ERROR_) { set_error("incomplete stream"); return Z_DATA_ERROR_; }   // uncompr.c:73-74
        return rc;
    }
    if (pos + used + trail > n) { set_error("incomplete stream"); return Z_DATA_ERROR_; }
    if (format) {
        uint32_t chk[2];
        rc = device_checks(d_dst, (size_t)got, st, chk);
        if (rc) return rc;
        const uint8_t *tp = src + pos + used;
        if (!gz) {
            const uint32_t want = ((uint32_t)tp[0] << 24) | ((uint32_t)tp[1] << 16) | ((uint32_t)tp[2] << 8) | tp[3];
            if (want != chk[0]) { set_error("incorrect data check"); return Z_DATA_ERROR_; }
        } else {
            const uint32_t want = tp[0] | ((uint32_t)tp[1] << 8) | ((uint32_t)tp[2] << 16) | ((uint32_t)tp[3] << 24);
            const uint32_t isize = tp[4] | ((uint32_t)tp[5] << 8) | ((uint32_t)tp[6] << 16) | ((uint32_t)tp[7] << 24);
            if (want != chk[1]) { set_error("incorrect data check"); return Z_DATA_ERROR_; }
            if (isize != (uint32_t)got) { set_error("incorrect length check"); return Z_DATA_ERROR_; }
        }
    }
    *dst_len = (size_t)got;
    *src_len = pos + used + trail;
    return Z_OK_;
}

}  // extern "C"
