"""how does the level-6 class kernel's time depend on max_chain (level)?  -> where is the time going"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch, synth
zr = importlib.import_module("zlib-ng_amd"); dfl = importlib.import_module("zlib-ng_amd.deflate"); zr.init(0)
n = 128 << 20
src = torch.from_numpy(synth.silesia_like(n, seed=0x5EED0003)).cuda()
for level in (2, 4, 5, 6, 7):
    dfl.deflate_dev(src, level=level); torch.cuda.synchronize()
    zr.trace_begin(4); dst, clen = dfl.deflate_dev(src, level=level); k = zr.trace_end(4)
    t0 = time.perf_counter(); dfl.deflate_dev(src, level=level); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("level", level, "lz_ms %.1f" % k[0], "total_ms %.1f" % (dt * 1e3), "ratio %.3f" % (n / clen), "GB/s %.2f" % (n / 1e9 / dt))
