"""host-only: ONE stream (256 MiB plaintext, CPython zlib level 6) decoded on T threads into ordinary memory
(zng_rocm_inflate_tokens_decode_threads) -- how far does the cut-search-chain-join scheme scale on this box?"""
import ctypes as C, importlib, os, sys, time, zlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import synth
zr = importlib.import_module("zlib-ng_amd"); inf = importlib.import_module("zlib-ng_amd.inflate")
lib = zr.lib()
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
plain = synth.silesia_like(n, seed=0x5EED0003)
c = zlib.compressobj(6, zlib.DEFLATED, -15)
parts = [c.compress(plain[lo:lo + (32 << 20)].tobytes()) for lo in range(0, n, 32 << 20)] + [c.flush()]
comp = b"".join(parts)
buf = C.create_string_buffer(comp, len(comp))
tk = inf.InflateTokens()
for T in (1, 2, 4, 8, 16, 32, 64):
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        if T == 1:
            st = lib.zng_rocm_inflate_tokens_decode(C.addressof(buf), len(comp), C.byref(tk))
        else:
            st = lib.zng_rocm_inflate_tokens_decode_threads(C.addressof(buf), len(comp), 0, T, C.byref(tk))
        dt = time.perf_counter() - t0
        joined = lib.zng_rocm_inflate_threads_last_parts() if T > 1 else 0
        lib.zng_rocm_inflate_tokens_free(C.byref(tk))
        best = dt if best is None else min(best, dt)
    print("threads %2d: %.3f s  %.2f GB/s of output  status %d  parts joined %d" % (T, best, n / 1e9 / best, st, joined))
