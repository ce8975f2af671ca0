"""kernel time of crc32 / adler32 over 64 MiB (settled clocks)"""
import importlib, os, statistics, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
zr = importlib.import_module("zlib-ng_amd"); zr.init(0)
n = 64 << 20
big = torch.randint(0, 256, (n + 16,), dtype=torch.uint8, device="cuda")
out = torch.zeros(2, dtype=torch.int32, device="cuda")
for name, fn in (("adler32", lambda: zr.adler32_dev(big, out, length=n)), ("crc32", lambda: zr.crc32_dev(big, out, length=n)),
                 ("fused", lambda: zr.adler32_crc32_dev(big, out, length=n))):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    zr.trace_begin(100)
    for _ in range(100): fn()
    ms = statistics.mean(zr.trace_end(100))
    print(sys.argv[1] if len(sys.argv) > 1 else "", name, "%.2f us" % (ms * 1e3), "frac %.3f" % (n / 1e9 / (ms / 1e3) / 8000))
