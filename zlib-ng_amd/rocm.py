"""ctypes binding of include/zng_rocm.h (the C-ABI drop-in boundary)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "libzng_rocm.so")
_lib = None


class ZngRocmError(RuntimeError):
    pass


class Crc32FoldState(C.Structure):
    """struct crc32_fold_s (crc32.h:8-14)"""
    _fields_ = [("fold", C.c_uint8 * 64), ("value", C.c_uint32)]


def lib_path():
    return _LIB_PATH


def build(verbose=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(cmd, stdout=out)
    return _LIB_PATH


_PROTOS = {
    # name: (restype, argtypes)
    "zng_rocm_init": (C.c_int, [C.c_int]),
    "zng_rocm_available": (C.c_int, []),
    "zng_rocm_device_count": (C.c_int, []),
    "zng_rocm_last_error": (C.c_char_p, []),
    "zng_rocm_device_info": (C.c_int, [C.POINTER(C.c_int32)]),
    "zng_rocm_shutdown": (C.c_int, []),
    "zng_rocm_stream_release": (C.c_int, [C.c_void_p]),
    "zng_rocm_trace_stride": (C.c_int, [C.c_int]),
    "zng_rocm_adler32_try": (C.c_int, [C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]),
    "zng_rocm_adler32_fold_copy_try": (C.c_int, [C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]),
    "zng_rocm_crc32_try": (C.c_int, [C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]),
    "zng_rocm_crc32_fold_try": (C.c_int, [C.POINTER(Crc32FoldState), C.c_void_p, C.c_size_t, C.c_uint32]),
    "zng_rocm_crc32_fold_copy_try": (C.c_int, [C.POINTER(Crc32FoldState), C.c_void_p, C.c_void_p, C.c_size_t]),
    "zng_rocm_adler32": (C.c_uint32, [C.c_uint32, C.c_void_p, C.c_size_t]),
    "zng_rocm_adler32_fold_copy": (C.c_uint32, [C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t]),
    "zng_rocm_crc32": (C.c_uint32, [C.c_uint32, C.c_void_p, C.c_size_t]),
    "zng_rocm_crc32_fold_reset": (C.c_uint32, [C.POINTER(Crc32FoldState)]),
    "zng_rocm_crc32_fold": (None, [C.POINTER(Crc32FoldState), C.c_void_p, C.c_size_t, C.c_uint32]),
    "zng_rocm_crc32_fold_copy": (None, [C.POINTER(Crc32FoldState), C.c_void_p, C.c_void_p, C.c_size_t]),
    "zng_rocm_crc32_fold_final": (C.c_uint32, [C.POINTER(Crc32FoldState)]),
    "zng_rocm_adler32_dev": (C.c_int, [C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zng_rocm_crc32_dev": (C.c_int, [C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zng_rocm_adler32_crc32_dev": (C.c_int, [C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zng_rocm_fold_copy_dev": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_void_p, C.c_void_p]),
    "zng_rocm_checksums_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zng_rocm_adler32_combine_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zng_rocm_crc32_combine_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zng_rocm_reserve_cus": (C.c_int, [C.c_int]),
    "zng_rocm_combine_rows_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zng_rocm_slide_hash_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "zng_rocm_compare256_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zng_rocm_update_hash_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zng_rocm_quick_insert_string_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "zng_rocm_insert_string_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "zng_rocm_update_hash_roll_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zng_rocm_quick_insert_string_roll_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                        C.c_void_p]),
    "zng_rocm_insert_string_roll_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p]),
    "zng_rocm_longest_match_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "zng_rocm_longest_match_slow_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p]),
    "zng_rocm_chunkmemset_safe_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_size_t, C.c_void_p]),
    "zng_rocm_chunksize": (C.c_uint32, []),
    "zng_rocm_deflate_bound": (C.c_size_t, [C.c_size_t]),
    "zng_rocm_deflate_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                       C.POINTER(C.c_size_t), C.c_void_p]),
    "zng_rocm_deflate_block_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t,
                                             C.POINTER(C.c_size_t), C.c_void_p]),
    "zng_rocm_deflate_quick_bound": (C.c_size_t, [C.c_size_t]),
    "zng_rocm_deflate_quick_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zng_rocm_deflate_streams_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zng_rocm_inflate_tokens_decode": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "zng_rocm_inflate_tokens_decode_window": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]),
    "zng_rocm_inflate_resolve_window_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                      C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "zng_rocm_inflate_raw_window": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t,
                                              C.POINTER(C.c_uint64), C.POINTER(C.c_size_t), C.c_void_p]),
    "zng_rocm_inflate_tokens_decode_threads": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_void_p]),
    "zng_rocm_inflate_threads_last_parts": (C.c_int, []),
    "zng_rocm_inflate_raw_threads": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t,
                                               C.POINTER(C.c_uint64), C.POINTER(C.c_size_t), C.c_int]),
    "zng_rocm_inflate_many": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int]),
    "zng_rocm_inflate_streams_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zng_rocm_compress_streams_bound": (C.c_size_t, [C.c_size_t, C.c_int]),
    "zng_rocm_compress_streams_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zng_rocm_uncompress_streams_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "zng_rocm_inflate_message": (C.c_char_p, [C.c_uint32]),
    "zng_rocm_inflate_tokens_free": (None, [C.c_void_p]),
    "zng_rocm_inflate_resolve_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                               C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "zng_rocm_inflate_raw": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64),
                                       C.c_void_p]),
    "zng_rocm_inflate_raw_ex": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_size_t), C.c_void_p]),
    "zng_rocm_compress_bound": (C.c_size_t, [C.c_size_t, C.c_int]),
    "zng_rocm_compress2_dev": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                         C.c_void_p]),
    "zng_rocm_uncompress2_dev": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t),
                                           C.c_int, C.c_void_p]),
    "zng_rocm_deflate_async_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t,
                                             C.c_void_p, C.c_void_p]),
    "zng_rocm_inflate_large_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t,
                                             C.POINTER(C.c_uint64), C.POINTER(C.c_size_t), C.c_void_p]),
    "zng_rocm_inflate_large_last_parts": (C.c_int, []),
    "zng_rocm_hook_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "zng_rocm_hook_destroy": (None, [C.c_void_p]),
    "zng_rocm_hook_reset": (C.c_int, [C.c_void_p]),
    "zng_rocm_hook_set_history": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "zng_rocm_hook_get_history": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "zng_rocm_hook_deflate_bound": (C.c_size_t, [C.c_size_t]),
    "zng_rocm_hook_deflate_block": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int,
                                              C.POINTER(C.c_uint32), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "zng_rocm_hook_inflate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                                        C.POINTER(C.c_char_p)]),
    "zng_rocm_trace_begin": (C.c_int, [C.c_int]),
    "zng_rocm_trace_end": (C.c_int, [C.POINTER(C.c_float), C.c_int]),
    "zng_rocm_adler32_combine": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_int64]),
    "zng_rocm_crc32_combine": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_int64]),
    "zng_rocm_crc32_combine_gen": (C.c_uint32, [C.c_int64]),
    "zng_rocm_crc32_combine_op": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32]),
}


def exported_names():
    return sorted(_PROTOS)


def lib():
    """The loaded C-ABI library.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ZngRocmError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % _LIB_PATH)
        # torch ships its own libamdhip64.so.7 (+ HSA runtime); import it FIRST so this library
        # binds to the HIP runtime already in the process instead of loading a second one
        # from /opt/rocm (two HIP runtimes in one process cannot both own the device).
        import torch  # noqa: F401
        handle = C.CDLL(_LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)          # AttributeError = header and library out of sync
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def _check(rc, what):
    if rc != 0:
        raise ZngRocmError("%s failed (%d): %s" % (what, rc, lib().zng_rocm_last_error().decode()))


def device_count():
    return lib().zng_rocm_device_count()


def available():
    return bool(lib().zng_rocm_available())


def init(device=-1):
    _check(lib().zng_rocm_init(device), "zng_rocm_init")


def _need_init():
    if not lib().zng_rocm_available():
        init(-1)


# ---- host-pointer slots, reference names ---------------------------------
def _host_ptr(data):
    """(keepalive, address, nbytes) for bytes / bytearray / numpy / None"""
    if data is None:
        return None, None, 0
    if isinstance(data, (bytes, bytearray, memoryview)):
        raw = bytes(data)
        keep = C.create_string_buffer(raw, max(len(raw), 1))
        return keep, C.addressof(keep), len(raw)
    import numpy as np
    arr = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    return arr, arr.ctypes.data, arr.size


def adler32_z(adler, buf, length=None):
    """zng_adler32_z (adler32.c:15-17) through the `adler32` slot."""
    _need_init()
    keep, ptr, n = _host_ptr(buf)
    n = n if length is None else length
    return lib().zng_rocm_adler32(adler & 0xffffffff, ptr, n)


def crc32_z(crc, buf, length=None):
    """zng_crc32_z (crc32.c:27-31) through the `crc32` slot (NULL -> 0)."""
    _need_init()
    keep, ptr, n = _host_ptr(buf)
    n = n if length is None else length
    return lib().zng_rocm_crc32(crc & 0xffffffff, ptr, n)


def adler32(buf, adler=1):
    return adler32_z(adler, buf)


def crc32(buf, crc=0):
    return crc32_z(crc, buf)


def adler32_fold_copy(adler, buf):
    """slot adler32_fold_copy: returns (checksum, copied bytes)"""
    _need_init()
    keep, ptr, n = _host_ptr(buf)
    dst = C.create_string_buffer(max(n, 1))
    val = lib().zng_rocm_adler32_fold_copy(adler & 0xffffffff, C.addressof(dst), ptr, n)
    return val, dst.raw[:n]


class Crc32Fold:
    """crc32_fold_reset / crc32_fold / crc32_fold_copy / crc32_fold_final"""

    def __init__(self):
        _need_init()
        self.state = Crc32FoldState()
        self.reset()

    def reset(self):
        return lib().zng_rocm_crc32_fold_reset(C.byref(self.state))

    def fold(self, buf, init_crc=0):
        keep, ptr, n = _host_ptr(buf)
        lib().zng_rocm_crc32_fold(C.byref(self.state), ptr, n, init_crc)

    def fold_copy(self, buf):
        keep, ptr, n = _host_ptr(buf)
        dst = C.create_string_buffer(max(n, 1))
        lib().zng_rocm_crc32_fold_copy(C.byref(self.state), C.addressof(dst), ptr, n)
        return dst.raw[:n]

    def final(self):
        return lib().zng_rocm_crc32_fold_final(C.byref(self.state))


def adler32_combine(adler1, adler2, len2):
    return lib().zng_rocm_adler32_combine(adler1, adler2, len2)


def crc32_combine(crc1, crc2, len2):
    return lib().zng_rocm_crc32_combine(crc1, crc2, len2)


def crc32_combine_gen(len2):
    return lib().zng_rocm_crc32_combine_gen(len2)


def crc32_combine_op(crc1, crc2, op):
    return lib().zng_rocm_crc32_combine_op(crc1, crc2, op)


# ---- device-resident entry points (torch tensors carry the memory) --------
def _stream_ptr(stream):
    import torch
    s = torch.cuda.current_stream() if stream is None else stream
    return C.c_void_p(s.cuda_stream)


def _dev_ptr(t, offset=0):
    return C.c_void_p(t.data_ptr() + offset)


def adler32_dev(buf, out, adler=1, length=None, offset=0, stream=None):
    """buf: uint8 CUDA tensor; out: uint32/int32 CUDA tensor (>= 1 element). Async on `stream`."""
    _need_init()
    n = buf.numel() - offset if length is None else length
    _check(lib().zng_rocm_adler32_dev(adler & 0xffffffff, _dev_ptr(buf, offset), n, _dev_ptr(out),
                                      _stream_ptr(stream)), "zng_rocm_adler32_dev")


def crc32_dev(buf, out, crc=0, length=None, offset=0, stream=None):
    _need_init()
    n = buf.numel() - offset if length is None else length
    _check(lib().zng_rocm_crc32_dev(crc & 0xffffffff, _dev_ptr(buf, offset), n, _dev_ptr(out),
                                    _stream_ptr(stream)), "zng_rocm_crc32_dev")


def adler32_crc32_dev(buf, out2, adler=1, crc=0, length=None, offset=0, stream=None):
    _need_init()
    n = buf.numel() - offset if length is None else length
    _check(lib().zng_rocm_adler32_crc32_dev(adler & 0xffffffff, crc & 0xffffffff, _dev_ptr(buf, offset), n,
                                            _dev_ptr(out2), _stream_ptr(stream)), "zng_rocm_adler32_crc32_dev")


def fold_copy_dev(which, dst, src, out2, adler=1, crc=0, length=None, src_offset=0, dst_offset=0, stream=None):
    _need_init()
    n = src.numel() - src_offset if length is None else length
    _check(lib().zng_rocm_fold_copy_dev(which, adler & 0xffffffff, crc & 0xffffffff, _dev_ptr(dst, dst_offset),
                                        _dev_ptr(src, src_offset), n, _dev_ptr(out2), _stream_ptr(stream)),
           "zng_rocm_fold_copy_dev")


class CheckJob(C.Structure):
    """zng_rocm_check_job"""
    _fields_ = [("buf", C.c_void_p), ("len", C.c_uint64), ("adler", C.c_uint32), ("crc", C.c_uint32)]


def checksums_dev(which, buf, offsets, lengths, out2, adlers=None, crcs=None, stream=None):
    """many messages of one uint8 CUDA tensor in one pass: message i = buf[offsets[i] : offsets[i] + lengths[i]];
    out2: int32/uint32 CUDA tensor [n, 2] <- {adler32, crc32} (which: 1 adler, 2 crc, 3 both).  Async on `stream`."""
    _need_init()
    n = len(lengths)
    jobs = (CheckJob * n)()
    base = buf.data_ptr()
    for i in range(n):
        jobs[i].buf = base + int(offsets[i])
        jobs[i].len = int(lengths[i])
        jobs[i].adler = 1 if adlers is None else int(adlers[i]) & 0xffffffff
        jobs[i].crc = 0 if crcs is None else int(crcs[i]) & 0xffffffff
    _check(lib().zng_rocm_checksums_dev(which, C.byref(jobs), n, _dev_ptr(out2), _stream_ptr(stream)),
           "zng_rocm_checksums_dev")


def adler32_combine_dev(checks, lens, out, stream=None):
    _need_init()
    _check(lib().zng_rocm_adler32_combine_dev(_dev_ptr(checks), _dev_ptr(lens), checks.numel(), _dev_ptr(out),
                                              _stream_ptr(stream)), "zng_rocm_adler32_combine_dev")


def reserve_cus(n):
    """leave n CUs out of the persistent checksum grid (for collectives / other streams); 0 = use all"""
    _need_init()
    _check(lib().zng_rocm_reserve_cus(int(n)), "zng_rocm_reserve_cus")


def combine_rows_dev(rows, count, out2, stream=None):
    """rows: device tensor of `count` packed {u32 adler, u32 crc, u64 len} rows (16 bytes each); out2: int32[2]"""
    _need_init()
    _check(lib().zng_rocm_combine_rows_dev(_dev_ptr(rows), count, _dev_ptr(out2), _stream_ptr(stream)),
           "zng_rocm_combine_rows_dev")


def crc32_combine_dev(checks, lens, out, stream=None):
    _need_init()
    _check(lib().zng_rocm_crc32_combine_dev(_dev_ptr(checks), _dev_ptr(lens), checks.numel(), _dev_ptr(out),
                                            _stream_ptr(stream)), "zng_rocm_crc32_combine_dev")


def trace_begin(max_launches):
    _need_init()
    _check(lib().zng_rocm_trace_begin(max_launches), "zng_rocm_trace_begin")


def trace_end(cap=4096):
    """per-launch durations (ms) of the dominant kernel since trace_begin"""
    buf = (C.c_float * cap)()
    n = lib().zng_rocm_trace_end(buf, cap)
    if n < 0:
        _check(n, "zng_rocm_trace_end")
    return [buf[i] for i in range(n)]


# ---- deflate-side primitives on device-resident state --------------------------
class DeflateView(C.Structure):
    """zng_rocm_deflate_view (include/zng_rocm.h): the deflate_state subset the kernels read"""
    _fields_ = [
        ("window", C.c_void_p), ("prev", C.c_void_p), ("head", C.c_void_p),
        ("w_size", C.c_uint32), ("w_mask", C.c_uint32),
        ("lookahead", C.c_uint32), ("strstart", C.c_uint32), ("match_start", C.c_uint32),
        ("prev_length", C.c_uint32), ("max_chain_length", C.c_uint32), ("good_match", C.c_uint32),
        ("nice_match", C.c_int32), ("level", C.c_int32),
    ]


def views_to_device(views):
    """pack a list of DeflateView into one uint8 CUDA tensor"""
    import numpy as np
    import torch
    arr = (DeflateView * len(views))(*views)
    raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
    return torch.from_numpy(raw).cuda()


def slide_hash_dev(d_views, nstreams, stream=None):
    _need_init()
    _check(lib().zng_rocm_slide_hash_dev(_dev_ptr(d_views), nstreams, _stream_ptr(stream)), "zng_rocm_slide_hash_dev")


def compare256_dev(base, off0, off1, out, stream=None):
    _need_init()
    _check(lib().zng_rocm_compare256_dev(_dev_ptr(base), _dev_ptr(off0), _dev_ptr(off1), off0.numel(), _dev_ptr(out),
                                         _stream_ptr(stream)), "zng_rocm_compare256_dev")


def update_hash_dev(val, out, stream=None):
    _need_init()
    _check(lib().zng_rocm_update_hash_dev(_dev_ptr(val), val.numel(), _dev_ptr(out), _stream_ptr(stream)),
           "zng_rocm_update_hash_dev")


def quick_insert_string_dev(d_views, nstreams, strs, head_out, stream=None):
    _need_init()
    _check(lib().zng_rocm_quick_insert_string_dev(_dev_ptr(d_views), nstreams, _dev_ptr(strs), _dev_ptr(head_out),
                                                  _stream_ptr(stream)), "zng_rocm_quick_insert_string_dev")


def insert_string_dev(d_views, nstreams, strs, counts, stream=None):
    _need_init()
    _check(lib().zng_rocm_insert_string_dev(_dev_ptr(d_views), nstreams, _dev_ptr(strs), _dev_ptr(counts),
                                            _stream_ptr(stream)), "zng_rocm_insert_string_dev")


def update_hash_roll_dev(h, val, out, stream=None):
    """update_hash_roll (insert_string_roll.c) elementwise over two uint32 arrays"""
    _need_init()
    _check(lib().zng_rocm_update_hash_roll_dev(_dev_ptr(h), _dev_ptr(val), val.numel(), _dev_ptr(out),
                                               _stream_ptr(stream)), "zng_rocm_update_hash_roll_dev")


def quick_insert_string_roll_dev(d_views, nstreams, strs, ins_h, head_out, stream=None):
    _need_init()
    _check(lib().zng_rocm_quick_insert_string_roll_dev(_dev_ptr(d_views), nstreams, _dev_ptr(strs), _dev_ptr(ins_h),
                                                       _dev_ptr(head_out), _stream_ptr(stream)),
           "zng_rocm_quick_insert_string_roll_dev")


def insert_string_roll_dev(d_views, nstreams, strs, counts, ins_h, stream=None):
    _need_init()
    _check(lib().zng_rocm_insert_string_roll_dev(_dev_ptr(d_views), nstreams, _dev_ptr(strs), _dev_ptr(counts),
                                                 _dev_ptr(ins_h), _stream_ptr(stream)),
           "zng_rocm_insert_string_roll_dev")


def longest_match_dev(d_views, nstreams, cur_match, len_out, start_out, stream=None):
    _need_init()
    _check(lib().zng_rocm_longest_match_dev(_dev_ptr(d_views), nstreams, _dev_ptr(cur_match), _dev_ptr(len_out),
                                            _dev_ptr(start_out), _stream_ptr(stream)), "zng_rocm_longest_match_dev")


def chunkmemset_safe_dev(base, out_off, from_off, lens, lefts, stream=None):
    _need_init()
    _check(lib().zng_rocm_chunkmemset_safe_dev(_dev_ptr(base), _dev_ptr(out_off), _dev_ptr(from_off), _dev_ptr(lens),
                                               _dev_ptr(lefts), out_off.numel(), _stream_ptr(stream)),
           "zng_rocm_chunkmemset_safe_dev")


def chunksize():
    return lib().zng_rocm_chunksize()
