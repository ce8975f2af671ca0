"""which true block starts of a CPython level-6 stream does the device finder (inflate_large.hip F1 + F2) report?
Needs the diagnostic library (tools/micro/inflate_stats.sh) for the list of part starts; the truth comes from the oracle's
block trace.   python tools/micro/find_blocks_check.py [MiB]"""
import ctypes as C, importlib, os, sys, zlib
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
importlib.import_module("zlib-ng_amd.rocm")._LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libzng_rocm_stats.so")
import torch, synth
zr = importlib.import_module("zlib-ng_amd"); inf = importlib.import_module("zlib-ng_amd.inflate"); zr.init(0)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
plain = synth.silesia_like(mib << 20, seed=0x5EED0003)
c = zlib.compressobj(6, zlib.DEFLATED, -15); raw = c.compress(plain.tobytes()) + c.flush()
ora = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
cap = 1 << 20
bits = (C.c_uint64 * cap)(); types = (C.c_uint8 * cap)()
ora.oracle_inflate_trace_blocks(bits, types, C.c_size_t(cap))
class R(C.Structure): _fields_ = [("status", C.c_int), ("msg", C.c_char_p), ("out_len", C.c_size_t), ("in_used", C.c_size_t)]
res = R(); out = (C.c_uint8 * (mib << 20))()
ora.oracle_inflate_raw.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(R)]
ora.oracle_inflate_raw(raw, len(raw), out, mib << 20, C.byref(res))
ora.oracle_inflate_traced_blocks.restype = C.c_size_t
n = ora.oracle_inflate_traced_blocks()
ora.oracle_inflate_trace_blocks(None, None, C.c_size_t(0))
true_bits = np.array(bits[:n], dtype=np.uint64).astype(np.int64); true_types = np.array(types[:n])
src = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).cuda()
dst = torch.zeros(plain.size, dtype=torch.uint8, device="cuda")
st, nout, used, parts = inf.inflate_large_dev(src, dst)
L = zr.rocm.lib(); pc = 1 << 16
stt = (C.c_ulonglong * pc)(); rs = (C.c_uint32 * (8 * pc))(); L.zng_rocm_debug_large_parts.restype = C.c_uint
npart = min(pc, L.zng_rocm_debug_large_parts(stt, rs, pc))
found = set(int(v) for v in stt[:npart])
print("status %d, %d blocks in the stream (stored %d, dynamic %d), %d part starts, %d on the chain"
      % (st, n, int((true_types == 0).sum()), int((true_types == 2).sum()), npart, parts))
for ty, name in ((0, "stored"), (2, "dynamic")):
    idx = np.nonzero(true_types == ty)[0]
    hit = sum(1 for i in idx if int(true_bits[i]) in found)
    print("%s: %d of %d starts reported" % (name, hit, len(idx)))
miss = [i for i in range(n) if true_types[i] == 2 and int(true_bits[i]) not in found]
prev = [int(true_types[i - 1]) if i else -1 for i in miss]
print("missed dynamic starts: the block in front is stored for %d, dynamic for %d; first few bit positions mod 8: %s"
      % (prev.count(0), prev.count(2), [int(true_bits[i]) & 7 for i in miss[:12]]))
gap = [int(true_bits[i] - true_bits[i - 1]) // 8 for i in miss if i]
if gap: print("compressed bytes between a missed start and the block in front: median %d, min %d" % (np.median(gap), min(gap)))
print("missed dynamic start bits:", [int(true_bits[i]) for i in miss[:40]])
