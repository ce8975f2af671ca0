"""ONE large stream on the device: zng_rocm_inflate_large_dev on the cfg3 stream (256 MiB of the mix, level 6).
  python tools/micro/run_inflate_large.py [MiB] [own|zlib]"""
import importlib, os, sys, time, zlib
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
if os.environ.get("ZR_STATS_LIB"):           # tools/micro/inflate_stats.sh: count the fast loop's exits
    importlib.import_module("zlib-ng_amd.rocm")._LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libzng_rocm_stats.so")
zr = importlib.import_module("zlib-ng_amd"); inf = importlib.import_module("zlib-ng_amd.inflate"); dfl = importlib.import_module("zlib-ng_amd.deflate")
zr.init(0)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = sys.argv[2] if len(sys.argv) > 2 else "own"
plain = synth.silesia_like(mib << 20, seed=0x5EED0003)
d_plain = torch.from_numpy(plain).cuda()
if kind == "own":
    comp, clen = dfl.deflate_dev(d_plain, level=6)
    src = comp[:clen].contiguous()
else:
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = c.compress(plain.tobytes()) + c.flush()
    src = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).cuda()
dst = torch.zeros(plain.size, dtype=torch.uint8, device="cuda")
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    zr.trace_begin(8)
    st, n, used, parts = inf.inflate_large_dev(src, dst)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    k = zr.trace_end(8)
    print("%s stream %d MiB -> %d MiB: status %d parts %d  %.2f ms = %.2f GB/s of output; traced kernels (ms): %s; exact %s"
          % (kind, src.numel() >> 20, n >> 20, st, parts, dt * 1e3, n / 1e9 / dt, ["%.2f" % x for x in k], bool(torch.equal(dst, d_plain))))
print(zr.rocm.lib().zng_rocm_last_error())
if os.environ.get("ZR_STATS_LIB"):
    import ctypes as C
    L = zr.rocm.lib(); st = (C.c_ulonglong * 16)()
    L.zng_rocm_debug_inflate_stats(st, 1); inf.inflate_large_dev(src, dst); torch.cuda.synchronize(); L.zng_rocm_debug_inflate_stats(st, 0)
    cap = 1 << 14
    stt = (C.c_ulonglong * cap)(); rs = (C.c_uint32 * (8 * cap))(); L.zng_rocm_debug_large_parts.restype = C.c_uint
    npart = min(cap, L.zng_rocm_debug_large_parts(stt, rs, cap))
    stt = np.array(stt[:npart], dtype=np.uint64).astype(np.int64); rs = np.array(rs[:8 * npart], dtype=np.uint32).reshape(-1, 8)
    sp = (C.c_ulonglong * (2 * npart))(); L.zng_rocm_debug_inflate_spans(sp, npart)
    a = np.array(sp, dtype=np.uint64).reshape(-1, 2).astype(np.int64)
    dur = a[:, 1] - a[:, 0]
    q = np.percentile(dur, [0, 10, 50, 90, 99, 100])
    print("%d part starts, %d on the chain; s_memtime ticks per part: inside the hand-written loop %.0f, whole decode %.0f"
          % (npart, parts, st[11] / npart, st[12] / npart))
    print("part durations (Mticks): min %.2f p10 %.2f median %.2f p90 %.2f p99 %.2f max %.2f; sum %.0f" % (*(q / 1e6), dur.sum() / 1e6))
    nbits = np.diff(np.append(stt, src.numel() * 8))
    order = np.argsort(-dur)
    t0 = a[:, 0].min(); span = a[:, 1].max() - t0
    print("the launch: first start to last end %.2f Mticks; parts starting in the first 5 %% of it: %d; the last part to START does so at %.2f"
          % (span / 1e6, int((a[:, 0] - t0 < 0.05 * span).sum()), (a[:, 0].max() - t0) / 1e6))
    print("slowest parts: index, Mticks, symbols produced, compressed bytes to the next start, ticks per symbol, message, start and end in the launch (Mticks)")
    for j in order[:8]:
        print("  %5d %7.2f %8d %8d %7.0f %d  %6.2f %6.2f" % (j, dur[j] / 1e6, rs[j, 0], nbits[j] // 8, dur[j] / max(1, rs[j, 0]), rs[j, 4],
                                                  (a[j, 0] - t0) / 1e6, (a[j, 1] - t0) / 1e6))
    last = np.argsort(-a[:, 1])[:8]
    print("last parts to END: index, start, end, symbols")
    for j in last:
        print("  %5d %6.2f %6.2f %8d" % (j, (a[j, 0] - t0) / 1e6, (a[j, 1] - t0) / 1e6, rs[j, 0]))
    print("median part: %d symbols, %d compressed bytes, %.0f ticks per symbol" % (np.median(rs[:, 0]), np.median(nbits) // 8, np.median(dur / np.maximum(1, rs[:, 0]))))
