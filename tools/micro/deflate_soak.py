"""a one-off round-trip soak of zng_rocm_deflate_dev (levels 0..9): sizes around every boundary the emitter has -- batches
(1024), blocks (61440 positions), segments (128 / 256 / 512 KiB) -- and random ones up to 6 MiB, seven kinds of content;
every stream must inflate to its input with CPython's zlib, and, for a sample, with the device inflaters
(zng_rocm_inflate_streams_dev; zng_rocm_inflate_large_dev from 128 KiB of compressed bytes).
   python tools/micro/deflate_soak.py [streams] [seed]"""
import importlib, os, random, sys, time, zlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch, synth
from test_gpu_deflate_fuzz import _content
zr = importlib.import_module("zlib-ng_amd"); dfl = importlib.import_module("zlib-ng_amd.deflate")
inf = importlib.import_module("zlib-ng_amd.inflate"); zr.init(0)
n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
edges = []
for base in (1024, 61440, 2 * 61440, 3 * 61440, 128 << 10, 256 << 10, 512 << 10, (512 << 10) + 61440, 1 << 20, 8 * 61440, 9 * 61440):
    edges += [base - 1, base, base + 1, base + 1023, base + 1025]
sizes = edges + [rnd.randrange(0, 6 << 20) for _ in range(max(0, n_streams - len(edges)))]
bad, t0, total_in, total_out, large, many = [], time.perf_counter(), 0, 0, 0, []
for k, n in enumerate(sizes[:n_streams]):
    data = _content(rnd, n) if n else np.zeros(0, dtype=np.uint8)
    if rnd.random() < 0.15 and n > 200000:                     # a noise stretch inside compressible data: stored blocks between dynamic ones
        a = rnd.randrange(0, n - 150000); data = data.copy(); data[a:a + 130000] = np.frombuffer(rnd.randbytes(130000), dtype=np.uint8)
    level = rnd.choice([0, 1, 2, 3, 4, 5, 6, 6, 6, 7, 8, 9])
    src = torch.from_numpy(np.concatenate([data, np.zeros(16, dtype=np.uint8)])).cuda()
    dst, clen = dfl.deflate_dev(src, level=level, length=n)
    comp = dst[:clen].cpu().numpy().tobytes()
    d = zlib.decompressobj(-15)
    try:
        ok = d.decompress(comp) == data.tobytes() and d.eof and d.unused_data == b""
    except zlib.error as e:
        ok = False
    if not ok:
        bad.append((k, n, level, "zlib"))
        continue
    total_in += n; total_out += clen
    if clen >= (128 << 10) and rnd.random() < 0.5:             # the emitter's blocks through the device part decoder
        out = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        st, got, used, parts = inf.inflate_large_dev(dst[:clen].contiguous(), out)
        large += parts > 0
        if (st, got, used) != (1, n, clen) or not torch.equal(out[:n], src[:n]) or int(out[n:].max()) != 0:
            bad.append((k, n, level, "inflate_large_dev", st, got, used, parts))
    elif n and len(many) < 64:
        many.append((comp, data))
if many:                                                        # and a batch through the one-wave-per-stream inflater
    in_off, pos = [], 0
    for c, _ in many:
        in_off.append(pos); pos += (len(c) + 15) & ~15
    host = np.zeros(pos + 16, dtype=np.uint8)
    for o, (c, _) in zip(in_off, many):
        host[o:o + len(c)] = np.frombuffer(c, dtype=np.uint8)
    out_off, pos = [], 0
    for _, a in many:
        out_off.append(pos); pos += a.size + 16
    d_out = torch.zeros(pos + 64, dtype=torch.uint8, device="cuda")
    b = inf.InflateDevBatch(torch.from_numpy(host).cuda(), in_off, [len(c) for c, _ in many], d_out, out_off, [a.size for _, a in many])
    b.run(); rows = b.rows(); got = d_out.cpu().numpy()
    for i, ((c, a), r, o) in enumerate(zip(many, rows, out_off)):
        if r[0] != 1 or r[1] != a.size or got[o:o + a.size].tobytes() != a.tobytes():
            bad.append((i, a.size, "inflate_streams_dev", r))
print("%d streams, %.0f MiB in, ratio %.2f overall, %d of them back through inflate_large_dev in parts, %d through inflate_streams_dev; %.0f s; FAILURES: %d"
      % (min(n_streams, len(sizes)), total_in / 2**20, total_in / max(1, total_out), large, len(many), time.perf_counter() - t0, len(bad)))
for x in bad[:10]: print("  ", x)
sys.exit(1 if bad else 0)
