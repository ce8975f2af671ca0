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
