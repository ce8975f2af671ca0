"""host token decoder alone on T threads (no device work): how far does the sequential-decode half of inflate scale
on this box's cores?  64 streams of 4 MiB plaintext (CPython zlib level 6)."""
import ctypes as C, importlib, os, sys, threading, time, zlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import synth
zr = importlib.import_module("zlib-ng_amd"); inf = importlib.import_module("zlib-ng_amd.inflate")
lib = zr.lib()
plain = synth.silesia_like(64 << 20, seed=0x5EED0003, seg_bytes=4 << 20)
each = 4 << 20
parts = []
for k in range(16):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts.append(c.compress(plain[k * each:(k + 1) * each].tobytes()) + c.flush())
parts = parts * 4
bufs = [C.create_string_buffer(p, len(p)) for p in parts]
print("cpus: os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for T in (1, 2, 4, 8, 16, 32, 64):
    nxt = [0]; lock = threading.Lock()
    def work():
        tk = inf.InflateTokens()
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= len(parts): return
            lib.zng_rocm_inflate_tokens_decode(C.addressof(bufs[i]), len(parts[i]), C.byref(tk))
            lib.zng_rocm_inflate_tokens_free(C.byref(tk))
    ths = [threading.Thread(target=work) for _ in range(T)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print("threads %2d: %.2f GB/s of output" % (T, len(parts) * each / 1e9 / dt))
