"""oracle inflate wrapper for tests (test infrastructure)"""
import ctypes as C

import oracle_lib


class _Res(C.Structure):
    _fields_ = [("status", C.c_int), ("msg", C.c_char_p), ("out_len", C.c_size_t), ("in_used", C.c_size_t)]


def oracle_inflate(src, cap):
    lib = oracle_lib.load()
    fn = lib.oracle_inflate_raw
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(_Res)]
    sbuf = C.create_string_buffer(bytes(src), max(len(src), 1))
    dbuf = C.create_string_buffer(max(cap, 1))
    res = _Res()
    fn(C.addressof(sbuf), len(src), C.addressof(dbuf), cap, C.byref(res))
    return res.status, (res.msg or b"").decode(), dbuf.raw[:res.out_len], res.in_used


def oracle_inflate_dict(src, dictionary, cap):
    """raw inflate after inflateSetDictionary(dictionary)"""
    lib = oracle_lib.load()
    fn = lib.oracle_inflate_raw_dict
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(_Res)]
    sbuf = C.create_string_buffer(bytes(src), max(len(src), 1))
    kbuf = C.create_string_buffer(bytes(dictionary), max(len(dictionary), 1))
    dbuf = C.create_string_buffer(max(cap, 1))
    res = _Res()
    fn(C.addressof(sbuf), len(src), C.addressof(kbuf), len(dictionary), C.addressof(dbuf), cap, C.byref(res))
    return res.status, (res.msg or b"").decode(), dbuf.raw[:res.out_len], res.in_used


def oracle_block_starts(src, cap):
    """(bit position of BFINAL, BTYPE) of every block of a raw deflate stream, as the oracle walks it"""
    lib = oracle_lib.load()
    lib.oracle_inflate_trace_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.oracle_inflate_traced_blocks.restype = C.c_size_t
    n_max = 1 << 20
    bits = (C.c_uint64 * n_max)()
    types = (C.c_uint8 * n_max)()
    lib.oracle_inflate_trace_blocks(bits, types, n_max)
    try:
        status = oracle_inflate(src, cap)[0]
        n = min(n_max, lib.oracle_inflate_traced_blocks())
    finally:
        lib.oracle_inflate_trace_blocks(None, None, 0)
    return status, [(int(bits[i]), int(types[i])) for i in range(n)]
