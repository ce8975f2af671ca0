"""does the fused checksum kernel slow down under sustained launches (clock give-back), and does a pause help?"""
import importlib, os, sys, statistics, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
zr = importlib.import_module("zlib-ng_amd"); zr.init(0)
n = 1 << 30
buf = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
out = torch.zeros(2, dtype=torch.int32, device="cuda")
for label, fn in (("fused", lambda: zr.adler32_crc32_dev(buf, out)), ("adler", lambda: zr.adler32_dev(buf, out)), ("crc", lambda: zr.crc32_dev(buf, out))):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    zr.trace_begin(200)
    for _ in range(200): fn()
    ms = zr.trace_end(200)
    chunks = [statistics.mean(ms[i:i+20]) for i in range(0, 200, 20)]
    print(label, "GB/s per 20-launch chunk:", [round(n/1e9/(c/1e3)) for c in chunks])
    time.sleep(0.5)
