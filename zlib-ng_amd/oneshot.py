"""compress2 / uncompress2 class front ends (compress.c, uncompr.c) over the device paths.
format: 0 raw deflate, 1 zlib (RFC 1950), 2 gzip (RFC 1952)."""
import ctypes as C

from . import rocm

RAW, ZLIB, GZIP = 0, 1, 2
Z_OK, Z_DATA_ERROR, Z_BUF_ERROR = 0, -3, -5


def compress_bound(n, fmt=ZLIB):
    return rocm.lib().zng_rocm_compress_bound(n, fmt)


def compress2_dev(src, level=-1, fmt=ZLIB, length=None, stream=None):
    """device plaintext -> (device tensor, compressed length)"""
    import torch
    rocm._need_init()
    n = src.numel() if length is None else length
    cap = compress_bound(n, fmt)
    dst = torch.empty(cap, dtype=torch.uint8, device=src.device)
    dlen = C.c_size_t(cap)
    rc = rocm.lib().zng_rocm_compress2_dev(rocm._dev_ptr(dst), C.byref(dlen), rocm._dev_ptr(src), n, level, fmt,
                                           rocm._stream_ptr(stream))
    rocm._check(rc, "zng_rocm_compress2_dev")
    return dst, dlen.value


def uncompress2_dev(src_bytes, dst, fmt=ZLIB, stream=None):
    """host stream -> plaintext in the CUDA tensor dst; returns (zlib status, produced, consumed, message)"""
    rocm._need_init()
    raw = bytes(src_bytes)
    buf = C.create_string_buffer(raw, max(len(raw), 1))
    dlen = C.c_size_t(dst.numel())
    slen = C.c_size_t(len(raw))
    rc = rocm.lib().zng_rocm_uncompress2_dev(rocm._dev_ptr(dst), C.byref(dlen), C.addressof(buf), C.byref(slen), fmt,
                                             rocm._stream_ptr(stream))
    msg = rocm.lib().zng_rocm_last_error().decode() if rc else ""
    return rc, dlen.value, slen.value, msg
