"""a one-off soak of the device inflater's decode loop against the oracle: thousands of streams -- every level and
strategy, window sizes 2^9 .. 2^15, dictionaries, sizes from 0 to ~400 KB -- each also damaged in a random way; status,
message, bytes produced, bytes consumed and the bytes themselves must be the oracle's (as tests/test_gpu_inflate_dev.py's
mutated-stream test demands of its 360).   python tools/micro/inflate_soak.py [streams] [seed]"""
import importlib, os, sys, time, zlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch, synth, inflate_util
zr = importlib.import_module("zlib-ng_amd"); inf = importlib.import_module("zlib-ng_amd.inflate"); zr.init(0)
n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tails = len(sys.argv) > 3 and sys.argv[3] == "tails"      # instead: small streams cut at EVERY length, 16 values of the last byte
rng = np.random.default_rng(seed)
corpus = synth.silesia_like(8 << 20, seed=100 + seed, seg_bytes=128 << 10).tobytes()
strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]
streams, dicts, caps = [], [], []
for k in range(n_streams // 2):
    size = int(rng.choice([0, 1, 50, 3000, 40000, 150000, 400000] if not tails else [1, 10, 50, 200, 400, 60000]) * rng.uniform(0.5, 1.0))
    at = int(rng.integers(0, len(corpus) - size))
    plain = corpus[at:at + size]
    if rng.random() < 0.1:
        plain = bytes(rng.integers(0, 256, size=size, dtype=np.uint8))            # stored blocks
    if rng.random() < 0.1:
        plain = bytes([int(rng.integers(0, 256))]) * size                          # one long run
    d = b""
    if rng.random() < 0.2 and at > 40000:
        d = corpus[at - int(rng.integers(1, 32769)):at]
    wbits = int(rng.integers(9, 16)) if not d else 15
    args = (int(rng.integers(0, 10)), zlib.DEFLATED, -wbits, int(rng.integers(1, 10)), strategies[int(rng.integers(0, 5))])
    c = zlib.compressobj(*args, d) if d else zlib.compressobj(*args)
    s = c.compress(plain) + c.flush()
    streams.append(s); dicts.append(d); caps.append(len(plain) + int(rng.integers(0, 3)) * 7)
    b = bytearray(s)                                                               # and a damaged twin
    kind = int(rng.integers(0, 5))
    if kind == 0 and b:
        b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1 and b:
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
    elif kind == 2:
        b = b[:int(rng.integers(0, len(b) + 1))]
    elif kind == 3 and len(b) > 40:
        p = int(rng.integers(0, len(b) - 32)); b[p:p + 16] = bytes(16)
    else:
        b = b + bytes(rng.integers(0, 256, size=int(rng.integers(0, 9)), dtype=np.uint8))   # trailing bytes
    streams.append(bytes(b)); dicts.append(d)
    caps.append(max(0, len(plain) + int(rng.integers(-40, 400))))
if tails:
    base = [(s, d) for s, d in zip(streams[0::2], dicts[0::2]) if 0 < len(s) <= 300][:400]
    streams, dicts, caps = [], [], []
    for s, d in base:
        for cut in range(1, len(s) + 1):
            for v in rng.integers(0, 256, size=16):
                streams.append(s[:cut - 1] + bytes([int(v)])); dicts.append(d); caps.append(70000)
in_off, pos = [], 1
for s in streams:
    in_off.append(pos); pos += len(s) + int(rng.integers(0, 5))
src = np.zeros(pos + 64, dtype=np.uint8)
for o, s in zip(in_off, streams):
    src[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
out_off, pos = [], 3
for cp, d in zip(caps, dicts):
    pos += len(d); out_off.append(pos); pos += cp + int(rng.integers(0, 9))
dst_host = np.full(pos + 64, 0xA5, dtype=np.uint8)
for o, d in zip(out_off, dicts):
    if d: dst_host[o - len(d):o] = np.frombuffer(d, dtype=np.uint8)
d_src = torch.from_numpy(src).cuda(); d_dst = torch.from_numpy(dst_host).cuda()
t0 = time.perf_counter()
b = inf.InflateDevBatch(d_src, in_off, [len(s) for s in streams], d_dst, out_off, caps, [len(d) for d in dicts])
b.run(); rows = b.rows(); got = d_dst.cpu().numpy()
t_dev = time.perf_counter() - t0
differ, ended, errors, short = [], 0, 0, 0
t0 = time.perf_counter()
for k, (s, r, o, cp, d) in enumerate(zip(streams, rows, out_off, caps, dicts)):
    ost, omsg, oout, oused = (inflate_util.oracle_inflate_dict(s, d, cap=cp) if d else inflate_util.oracle_inflate(s, cap=cp))
    if ost == -5 and r[0] == -5:
        short += 1
        continue                                          # both ran out of input or room; partial output is not compared
    mine = got[o:o + r[1]].tobytes()
    if (r[0], r[3]) != (ost, omsg) or (ost == 1 and (mine != oout or r[2] != oused)):
        differ.append((k, len(s), r, (ost, omsg, len(oout), oused)))
    ended += ost == 1; errors += ost == -3
print("%d streams (%d MB of output room): %d end of stream, %d data errors, %d out of input / room; device %.2f s, oracle %.1f s; DIFFERENCES: %d"
      % (len(streams), sum(caps) >> 20, ended, errors, short, t_dev, time.perf_counter() - t0, len(differ)))
for x in differ[:10]: print("  ", x)
sys.exit(1 if differ else 0)
