"""What is inside the blocks a large-stream call waits for: the copies of the heaviest block of a CPython level-6 stream of the
mix (lengths, distances, how many overlap, how many reach further back than the device's ring holds).  CPU only.
  python tools/micro/block_profile.py [MiB]"""
import os, sys, zlib
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from inflate_util import oracle_block_starts

LBASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEXT = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DBASE = [1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DEXT = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
ORDER = [16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]

class Rd:
    def __init__(self, data, bit): self.d, self.p = data, bit
    def bits(self, n):
        v = 0
        for i in range(n):
            v |= ((self.d[self.p >> 3] >> (self.p & 7)) & 1) << i; self.p += 1
        return v

def table(lens):                                   # canonical code -> {(len, code): symbol}
    cnt = [0] * 16
    for l in lens: cnt[l] += 1
    cnt[0] = 0; code, nxt = 0, [0] * 16
    for l in range(1, 16):
        code = (code + cnt[l - 1]) << 1; nxt[l] = code
    t = {}
    for s, l in enumerate(lens):
        if l: t[(l, nxt[l])] = s; nxt[l] += 1
    return t

def sym(r, t):
    code = 0
    for l in range(1, 16):
        code = (code << 1) | r.bits(1)
        if (l, code) in t: return t[(l, code)]
    raise ValueError("bad code")

def block_tokens(data, bit):
    r = Rd(data, bit); r.bits(1)
    if r.bits(2) != 2: return None
    hlit, hdist, hclen = r.bits(5) + 257, r.bits(5) + 1, r.bits(4) + 4
    cl = [0] * 19
    for i in range(hclen): cl[ORDER[i]] = r.bits(3)
    ct, lens = table(cl), []
    while len(lens) < hlit + hdist:
        s = sym(r, ct)
        if s < 16: lens.append(s)
        elif s == 16: lens += [lens[-1]] * (3 + r.bits(2))
        elif s == 17: lens += [0] * (3 + r.bits(3))
        else: lens += [0] * (11 + r.bits(7))
    lt, dt, toks = table(lens[:hlit]), table(lens[hlit:]), []
    while True:
        s = sym(r, lt)
        if s < 256: toks.append((0, 0))
        elif s == 256: return toks
        else:
            ln = LBASE[s - 257] + r.bits(LEXT[s - 257]); d = sym(r, dt)
            toks.append((ln, DBASE[d] + r.bits(DEXT[d])))

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 32
plain = synth.silesia_like(mib << 20, seed=0x5EED0003)
c = zlib.compressobj(6, zlib.DEFLATED, -15); raw = c.compress(plain.tobytes()) + c.flush()
starts = np.asarray([b for b, _ in oracle_block_starts(np.frombuffer(raw, dtype=np.uint8), plain.size)[1]], dtype=np.int64)
gaps = np.diff(np.append(starts, len(raw) * 8))
print("%d blocks in %d bytes" % (len(starts), len(raw)))
best = None
for j in range(len(starts)):
    if gaps[j] < 20000 * 8 or gaps[j] > 40000 * 8: continue
    toks = block_tokens(raw, int(starts[j]))
    if toks is None: continue
    out = sum(max(1, t[0]) for t in toks)
    if best is None or out > best[0]: best = (out, j, toks)
out, j, toks = best
ln = np.array([t[0] for t in toks]); ds = np.array([t[1] for t in toks]); m = ln > 0
print("block %d at bit %d: %d codes, %d literals, %d copies, %d bytes out" % (j, starts[j], len(toks), (~m).sum(), m.sum(), out))
L, D = ln[m], ds[m]
print("copy lengths: median %d mean %.1f; > 64: %.1f %%; == 258: %.1f %%" % (np.median(L), L.mean(), 100 * (L > 64).mean(), 100 * (L == 258).mean()))
print("distances: median %d; < len (overlap): %.1f %%; < 64: %.1f %%; > 3838 (ring): %.1f %%; > 16384: %.1f %%"
      % (np.median(D), 100 * (D < L).mean(), 100 * (D < 64).mean(), 100 * (D > 3838).mean(), 100 * (D > 16384).mean()))
runs = np.diff(np.flatnonzero(np.concatenate(([True], m, [True])))) - 1
print("literal runs between copies: mean %.2f, zero-length %.1f %%" % (runs.mean(), 100 * (runs == 0).mean()))
h, e = np.histogram(D, bins=[1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32769])
print("distance histogram:", " ".join("%d:%d" % (int(a), int(b)) for a, b in zip(e[:-1], h)))
h, e = np.histogram(L, bins=[3, 4, 8, 16, 32, 64, 128, 258, 259])
print("length histogram:", " ".join("%d:%d" % (int(a), int(b)) for a, b in zip(e[:-1], h)))
