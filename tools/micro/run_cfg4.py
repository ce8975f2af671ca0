"""timing target: cfg4 = zng_rocm_deflate_dev level 6 on ONE 256 MiB stream of the mix (device resident).
  python tools/micro/run_cfg4.py [MiB] [level]
prints GB/s of input (median of 5), ratio, the traced kernel times of the last run, and whether CPython's zlib restores it."""
import importlib, os, sys, time, zlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
zr = importlib.import_module("zlib-ng_amd"); dfl = importlib.import_module("zlib-ng_amd.deflate"); zr.init(0)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
plain = synth.silesia_like(mib << 20, seed=0x5EED0003)
src = torch.from_numpy(plain).cuda()
dst, clen = dfl.deflate_dev(src, level=level)
torch.cuda.synchronize()
ts = []
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if rep == 4: zr.trace_begin(16)
    dst, clen = dfl.deflate_dev(src, level=level)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
k = zr.trace_end(16)
ts.sort()
comp = dst[:clen].cpu().numpy().tobytes()
ok = zlib.decompressobj(-15).decompress(comp) == plain.tobytes()
print("level %d, %d MiB: %.2f GB/s (median of 5: %.2f ms, range %.2f..%.2f), ratio %.3f, traced kernels (ms): %s, zlib restores it: %s"
      % (level, mib, (mib << 20) / 1e9 / ts[2], ts[2] * 1e3, ts[0] * 1e3, ts[-1] * 1e3, (mib << 20) / clen, ["%.2f" % x for x in k], ok))
