"""one-off confidence run (not part of the suite): random streams of random classes and sizes through every device
encoder (level-1 class, chain levels 1..9, with and without dictionary / pigz blocks) and back through the device
inflater and CPython's zlib.   python tools/fuzz_gpu_roundtrip.py [rounds] [seed]"""
import importlib, os, sys, zlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import torch
import synth
zr = importlib.import_module("zlib-ng_amd"); dfl = importlib.import_module("zlib-ng_amd.deflate")
inf = importlib.import_module("zlib-ng_amd.inflate"); zr.init(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def piece(n):
    k = int(rng.integers(0, 7))
    if k == 0:
        return rng.integers(0, 256, size=n, dtype=np.uint8)
    if k == 1:
        return np.zeros(n, dtype=np.uint8)
    if k == 2:
        p = rng.integers(0, 256, size=int(rng.integers(1, 400)), dtype=np.uint8)
        return np.tile(p, n // p.size + 1)[:n]
    if k == 3:
        return rng.integers(0, int(rng.integers(2, 20)), size=n, dtype=np.uint8)
    if k == 4:
        return np.minimum(rng.geometric(0.3, size=n), 255).astype(np.uint8)
    m = max(n, 8192)
    return synth.silesia_like(m, seed=int(rng.integers(0, 1 << 30)), seg_bytes=m)[:n]      # one class per stream


total = 0
for r in range(rounds):
    n = int(rng.integers(40, 400))
    lens = [int(v) for v in np.minimum(rng.geometric(1 / 60000.0, size=n), 3 << 20)]
    lens[0], lens[1] = 0, 1
    offs, pos = [], int(rng.integers(0, 16)) * 16
    for ln in lens:
        offs.append(pos)
        pos += (ln + 15) & ~15
    host = np.zeros(pos + 64, dtype=np.uint8)
    for o, ln in zip(offs, lens):
        host[o:o + ln] = piece(ln)
    src = torch.from_numpy(host).cuda()
    plain = [host[o:o + ln].tobytes() for o, ln in zip(offs, lens)]
    for level in [0] + [int(v) for v in rng.choice(np.arange(1, 10), size=3, replace=False)]:
        if level == 0:
            b = dfl.QuickBatch(src, offs, lens)
            b.run()
            res = b.results.cpu()
            clens = [int(res[i, 0]) for i in range(n)]
            comp_t, comp_off = b.dst, b.out_off
            for i in range(n):
                assert (int(res[i, 1]) & 0xffffffff) == zlib.adler32(plain[i]), ("adler", r, i)
        else:
            b = dfl.StreamsBatch(src, offs, lens)
            clens = b.run(level=level)
            comp_t, comp_off = b.dst, b.out_off
        out = torch.full((pos + 64,), 0x3C, dtype=torch.uint8, device="cuda")
        ib = inf.InflateDevBatch(comp_t, comp_off, clens, out, offs, lens)
        ib.run()
        rows = ib.rows()
        got = out.cpu().numpy()
        for i in range(n):
            assert rows[i] == (1, lens[i], clens[i], ""), (r, level, i, rows[i], lens[i], clens[i])
            assert got[offs[i]:offs[i] + lens[i]].tobytes() == plain[i], (r, level, i)
        for i in rng.choice(n, size=min(n, 12), replace=False):
            c = comp_t[comp_off[i]:comp_off[i] + clens[i]].cpu().numpy().tobytes()
            d = zlib.decompressobj(-15)
            assert d.decompress(c) == plain[i] and d.eof and d.unused_data == b"", (r, level, int(i))
        total += n
    print("round %d: %d streams x 4 encoders ok" % (r, n), flush=True)
print("fuzz ok: %d stream round trips" % total)
