"""ONE large stream on the device: zng_rocm_inflate_large_dev on the cfg3 stream (256 MiB of the mix, level 6).
  python tools/micro/run_inflate_large.py [MiB] [own|zlib]"""
import importlib, os, sys, time, zlib
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
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
