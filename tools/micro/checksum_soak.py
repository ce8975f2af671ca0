"""a one-off soak of the checksum entry points against CPython's zlib.adler32 / zlib.crc32 (classic zlib, independent):
single messages (adler32_dev, crc32_dev, the fused pass, fold_copy with its copy) at random lengths 0 .. 40 MiB, offsets
of any alignment and arbitrary seeds; thousands of messages per call through zng_rocm_checksums_dev; the host-pointer
slots.   python tools/micro/checksum_soak.py [seed]"""
import importlib, os, sys, time, zlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch
zr = importlib.import_module("zlib-ng_amd"); zr.init(0)
r = zr.rocm
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = 96 << 20
host = rng.integers(0, 256, size=N, dtype=np.uint8)
host[10 << 20:30 << 20] = 0xff                      # the sums' worst case
host[40 << 20:50 << 20] = 0
buf = torch.from_numpy(host).cuda()
dst = torch.zeros(N + 64, dtype=torch.uint8, device="cuda")
out = torch.zeros(2, dtype=torch.int32, device="cuda")
mv = memoryview(host)
bad, t0, singles = [], time.perf_counter(), 0
def want(off, n, a, c):
    return zlib.adler32(mv[off:off + n], a) & 0xffffffff, zlib.crc32(mv[off:off + n], c) & 0xffffffff
for k in range(400):
    n = int(rng.choice([0, 1, 15, 16, 17, 4095, 65536, 1 << 20, 8 << 20, 40 << 20]) * rng.uniform(0.3, 1.0))
    off = int(rng.integers(0, N - n - 64))
    a, c = (int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32))) if k % 3 else (1, 0)
    wa, wc = want(off, n, a, c)
    kind = k % 4
    if kind == 0:
        r.adler32_dev(buf, out, adler=a, length=n, offset=off); got = (int(out[0].item()) & 0xffffffff, wc)
    elif kind == 1:
        r.crc32_dev(buf, out, crc=c, length=n, offset=off); got = (wa, int(out[0].item()) & 0xffffffff)
    elif kind == 2:
        r.adler32_crc32_dev(buf, out, adler=a, crc=c, length=n, offset=off); got = tuple(int(v) & 0xffffffff for v in out.tolist())
    else:
        doff = (off & 15) + 16 * int(rng.integers(0, 4))                     # fold_copy wants src and dst congruent mod 16
        dst[doff - 1 if doff else 0] = 0x5a; dst[doff + n] = 0x5a
        r.fold_copy_dev(3, dst, buf, out, adler=a, crc=c, length=n, src_offset=off, dst_offset=doff)
        got = tuple(int(v) & 0xffffffff for v in out.tolist())
        if not torch.equal(dst[doff:doff + n], buf[off:off + n]) or int(dst[doff + n]) != 0x5a or (doff and int(dst[doff - 1]) != 0x5a):
            bad.append(("fold_copy bytes", k, n, off, doff))
    if got != (wa, wc):
        bad.append((("adler32", "crc32", "fused", "fold_copy")[kind], k, n, off, a, c, got, (wa, wc)))
    singles += 1
many = 0
for rep in range(6):                                                         # many messages per call
    m = int(rng.choice([1, 100, 4096, 20000]))
    lens = (rng.choice([0, 1, 100, 5000, 70000, 1 << 20, 20 << 20], size=m, p=[.05, .1, .3, .3, .2, .04, .01]) * rng.uniform(0.2, 1.0, size=m)).astype(np.int64)
    offs = np.array([int(rng.integers(0, N - int(l) - 64)) for l in lens], dtype=np.int64)
    ads = rng.integers(0, 1 << 32, size=m, dtype=np.uint64); crs = rng.integers(0, 1 << 32, size=m, dtype=np.uint64)
    which = int(rng.integers(1, 4))
    out2 = torch.zeros((m, 2), dtype=torch.int32, device="cuda")
    r.checksums_dev(which, buf, offs, lens, out2, adlers=ads, crcs=crs)
    got = out2.cpu().numpy().view(np.uint32)
    for i in range(m):
        wa, wc = want(int(offs[i]), int(lens[i]), int(ads[i]), int(crs[i]))
        if (which & 1 and int(got[i, 0]) != wa) or (which & 2 and int(got[i, 1]) != wc):
            bad.append(("checksums_dev", rep, i, which, int(lens[i]), int(offs[i]) & 15, (int(got[i, 0]), int(got[i, 1])), (wa, wc)))
    many += m
slots = 0
for k in range(60):                                                          # the host-pointer slots (bounded staging)
    n = int(rng.choice([0, 1, 1000, 1 << 20, 17 << 20, 40 << 20]) * rng.uniform(0.5, 1.0))
    off = int(rng.integers(0, N - n - 64))
    a, c = int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 32))
    wa, wc = want(off, n, a, c)
    if r.adler32(host[off:off + n], a) != wa or r.crc32(host[off:off + n], c) != wc:
        bad.append(("slot", k, n, off))
    slots += 1
print("%d single-message calls, %d messages through checksums_dev, %d host-pointer slot calls; %.0f s; FAILURES: %d"
      % (singles, many, slots, time.perf_counter() - t0, len(bad)))
for x in bad[:10]: print("  ", x)
sys.exit(1 if bad else 0)
