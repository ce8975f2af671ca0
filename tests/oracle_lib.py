"""ctypes loader for the CPU oracle (oracle/liboracle.so) -- test infrastructure.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)


class DeflateState(C.Structure):
    """mirror of oracle_deflate_state (oracle/zng_oracle.h)"""
    _fields_ = [
        ("w_size", C.c_uint32), ("w_bits", C.c_uint32), ("w_mask", C.c_uint32),
        ("lookahead", C.c_uint32), ("window_size", C.c_uint32),
        ("window", C.c_void_p), ("prev", C.c_void_p), ("head", C.c_void_p),
        ("strstart", C.c_uint32), ("match_start", C.c_uint32), ("prev_length", C.c_uint32),
        ("max_chain_length", C.c_uint32), ("good_match", C.c_uint32),
        ("nice_match", C.c_int32), ("level", C.c_int32), ("ins_h", C.c_uint32),
    ]


class Crc32Fold(C.Structure):
    _fields_ = [("fold", C.c_uint8 * 64), ("value", C.c_uint32)]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


def _sig(lib, name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


_LIB = None


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    vp, sz, u32, i64 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int64
    _sig(lib, "oracle_adler32", u32, [u32, vp, sz])
    _sig(lib, "oracle_adler32_fold_copy", u32, [u32, vp, vp, sz])
    _sig(lib, "oracle_adler32_combine", u32, [u32, u32, i64])
    _sig(lib, "oracle_crc32_braid", u32, [u32, vp, sz])
    _sig(lib, "oracle_crc32_bytewise", u32, [u32, vp, sz])
    _sig(lib, "oracle_crc32", u32, [u32, vp, sz])
    _sig(lib, "oracle_crc32_fold_reset", u32, [C.POINTER(Crc32Fold)])
    _sig(lib, "oracle_crc32_fold", None, [C.POINTER(Crc32Fold), vp, sz, u32])
    _sig(lib, "oracle_crc32_fold_copy", None, [C.POINTER(Crc32Fold), vp, vp, sz])
    _sig(lib, "oracle_crc32_fold_final", u32, [C.POINTER(Crc32Fold)])
    _sig(lib, "oracle_multmodp", u32, [u32, u32])
    _sig(lib, "oracle_x2nmodp", u32, [i64, C.c_uint])
    _sig(lib, "oracle_crc32_combine", u32, [u32, u32, i64])
    _sig(lib, "oracle_crc32_combine_gen", u32, [i64])
    _sig(lib, "oracle_crc32_combine_op", u32, [u32, u32, u32])
    _sig(lib, "oracle_get_crc_table", C.POINTER(C.c_uint32), [])
    _sig(lib, "oracle_slide_hash", None, [C.POINTER(DeflateState)])
    _sig(lib, "oracle_compare256", u32, [vp, vp])
    _sig(lib, "oracle_update_hash", u32, [u32, u32])
    _sig(lib, "oracle_quick_insert_string", C.c_uint16, [C.POINTER(DeflateState), u32])
    _sig(lib, "oracle_insert_string", None, [C.POINTER(DeflateState), u32, u32])
    _sig(lib, "oracle_update_hash_roll", u32, [u32, u32])
    _sig(lib, "oracle_quick_insert_string_roll", C.c_uint16, [C.POINTER(DeflateState), u32])
    _sig(lib, "oracle_insert_string_roll", None, [C.POINTER(DeflateState), u32, u32])
    _sig(lib, "oracle_longest_match", u32, [C.POINTER(DeflateState), C.c_uint16])
    _sig(lib, "oracle_longest_match_slow", u32, [C.POINTER(DeflateState), C.c_uint16])
    _sig(lib, "oracle_chunksize", u32, [])
    _sig(lib, "oracle_chunkmemset_safe", vp, [vp, vp, C.c_uint, C.c_uint])
    _LIB = lib
    return lib


def load_ref_crc32():
    """zng_crc32_braid compiled from the reference's own source (oracle/_ref)."""
    path = os.path.join(ORACLE_DIR, "_ref", "libzng_ref_crc32.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    fn = lib.zng_crc32_braid
    fn.restype = C.c_uint32
    fn.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
    return fn


def np_ptr(arr):
    """address of a numpy array's first byte (array must stay alive)"""
    return arr.ctypes.data
