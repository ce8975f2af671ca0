"""timing / profiling target: zng_rocm_inflate_streams_dev on N x 1 MiB streams.
  python tools/micro/run_inflate_dev.py [nstreams] [encoder: quick | zlib1 | zlib6 | zlib9] [reps]
quick  = the product's level-1 class (one static-Huffman block per stream), compressed on the device;
zlibL  = CPython zlib level L (dynamic blocks) of 64 distinct MiB, repeated to N streams; zlibLwB: with a 2^B window;
ownL   = the product's level-L class (zng_rocm_deflate_dev), one stream per MiB."""
import importlib, os, sys, time, zlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
import torch, synth
if os.environ.get("ZR_STATS_LIB"):           # tools/micro/inflate_stats.sh: count the fast loop's exits
    importlib.import_module("zlib-ng_amd.rocm")._LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libzng_rocm_stats.so")
zr = importlib.import_module("zlib-ng_amd"); dfl = importlib.import_module("zlib-ng_amd.deflate")
inf = importlib.import_module("zlib-ng_amd.inflate"); zr.init(0)
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
enc = sys.argv[2] if len(sys.argv) > 2 else "quick"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
sz = 1 << 20
distinct = min(ns, 64)
plain = synth.silesia_like(distinct * sz, seed=0x5EED0005)
src = torch.from_numpy(plain).cuda().repeat(ns // distinct)
if enc == "quick":
    qb = dfl.QuickBatch(src, [i * sz for i in range(ns)], [sz] * ns)
    qb.run()
    clen = [int(v) for v in qb.results.cpu()[:, 0]]
    comp, in_off = qb.dst, qb.out_off
else:
  if enc.startswith("own"):                  # the product's own level-L class, one stream per MiB
    blobs = []
    for i in range(distinct):
        d, n = dfl.deflate_dev(src[i * sz:(i + 1) * sz].contiguous(), level=int(enc[3:]))
        blobs.append(d[:n].cpu().numpy().tobytes())
  else:
    level, wbits = (int(enc[4:].split("w")[0]), int(enc.split("w")[1])) if "w" in enc else (int(enc[4:]), 15)
    blobs = []
    for i in range(distinct):
        c = zlib.compressobj(level, zlib.DEFLATED, -wbits)
        blobs.append(c.compress(plain[i * sz:(i + 1) * sz].tobytes()) + c.flush())
  if True:
    blobs = blobs * (ns // distinct)
    clen = [len(b) for b in blobs]
    in_off, pos = [], 0
    for b in blobs:
        in_off.append(pos)
        pos += (len(b) + 15) & ~15
    host = np.zeros(pos + 16, dtype=np.uint8)
    for o, b in zip(in_off, blobs):
        host[o:o + len(b)] = np.frombuffer(b, dtype=np.uint8)
    comp = torch.from_numpy(host).cuda()
dst = torch.empty(ns * sz + 64, dtype=torch.uint8, device="cuda")
b = inf.InflateDevBatch(comp, in_off, clen, dst, [i * sz for i in range(ns)], [sz] * ns)
b.run()
torch.cuda.synchronize()
r = b.results.cpu()
assert (r[:, 2] == 1).all() and (r[:, 0] == sz).all(), r[:4]
assert torch.equal(dst[:ns * sz], src)
zr.trace_begin(reps)
t0 = time.perf_counter()
for _ in range(reps):
    b.run()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps
ms = zr.trace_end(reps)
k = sum(ms) / len(ms)
print("%s: %d x 1 MiB, compressed %.1f MiB (ratio %.3f): kernel %.3f ms = %.1f GB/s out, %.1f GB/s in (wall %.3f ms)" %
      (enc, ns, sum(clen) / 2**20, ns * sz / sum(clen), k, ns * sz / 1e9 / (k / 1e3), sum(clen) / 1e9 / (k / 1e3), wall * 1e3))
if os.environ.get("ZR_STATS_LIB"):
    import ctypes as C
    L = zr.rocm.lib(); st = (C.c_ulonglong * 16)()
    L.zng_rocm_debug_inflate_stats(st, 1); b.run(); torch.cuda.synchronize(); L.zng_rocm_debug_inflate_stats(st, 0)
    names = ["loop entries", "0: 64 literals wait", "0: fetched words used up", "0: EOB / long / bad code", "1: words used up",
             "1: long / bad distance code", "2: len > 64", "2: flush or end of out due", "2: source in front of out", "2: overlap (dist < len)", "2: other"]
    print("per MiB of output:", ", ".join("%s %.0f" % (n, st[i] / ns) for i, n in enumerate(names)))
    print("s_memtime ticks per stream: inside the hand-written loop %.0f, whole decode %.0f" % (st[11] / ns, st[12] / ns))
