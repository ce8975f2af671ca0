"""are the _dev entry points graph-capturable (no sync, no allocation once the stream's workspace exists), and what does
a hipGraph replay of the two-launch checksum step cost at 64 MiB?"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
zr = importlib.import_module("zlib-ng_amd"); zr.init(0)
n = 64 << 20
buf = torch.randint(0, 256, (n + 16,), dtype=torch.uint8, device="cuda")
out = torch.zeros(2, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    zr.adler32_crc32_dev(buf, out, length=n, stream=side)          # first use of this stream: workspace gets created
side.synchronize()
want = out.clone()
g = torch.cuda.CUDAGraph()
out.zero_()
with torch.cuda.graph(g, stream=side):
    zr.adler32_crc32_dev(buf, out, length=n, stream=side)
g.replay(); torch.cuda.synchronize()
print("captured, replay value ok:", bool((out == want).all()))
def timeit(fn, reps=300):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
print("eager 2 launches: %.1f us/step" % timeit(lambda: zr.adler32_crc32_dev(buf, out, length=n)))
print("graph replay    : %.1f us/step" % timeit(lambda: g.replay()))
