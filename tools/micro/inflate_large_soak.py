"""a one-off soak of zng_rocm_inflate_large_dev (block starts found on the device, one wavefront per part, 16-bit symbols,
context chain) against the oracle: streams of 150 KB .. 6 MB of every level, strategy and window size, with and without a
dictionary, each also damaged; status, message, bytes produced / consumed and the bytes must be the oracle's.
   python tools/micro/inflate_large_soak.py [streams] [seed]"""
import importlib, os, sys, time, zlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch, synth, inflate_util
zr = importlib.import_module("zlib-ng_amd"); inf = importlib.import_module("zlib-ng_amd.inflate"); zr.init(0)
n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
corpus = synth.silesia_like(24 << 20, seed=200 + seed, seg_bytes=512 << 10).tobytes()
strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]
differ, on_device, ended, errors, short = [], 0, 0, 0, 0
t0 = time.perf_counter()
for k in range(n_streams):
    size = int(rng.choice([400000, 1 << 20, 3 << 20, 6 << 20]) * rng.uniform(0.5, 1.0))
    at = int(rng.integers(40000, len(corpus) - size))
    plain = corpus[at:at + size]
    d = corpus[at - int(rng.integers(1, 32769)):at] if rng.random() < 0.25 else b""
    wbits = int(rng.integers(10, 16)) if not d else 15
    args = (int(rng.integers(1, 10)), zlib.DEFLATED, -wbits, int(rng.integers(1, 10)), strategies[int(rng.choice([0, 0, 0, 1, 2, 3, 4]))])
    c = zlib.compressobj(*args, d) if d else zlib.compressobj(*args)
    s = bytearray(c.compress(plain) + c.flush())
    if k % 2:                                              # damage every second one
        kind = int(rng.integers(0, 5))
        p = int(rng.integers(0, len(s)))
        if kind == 0: s[p] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1: s[p] = int(rng.integers(0, 256))
        elif kind == 2: s = s[:p]
        elif kind == 3 and p + 16 < len(s): s[p:p + 16] = bytes(16)
        else: s = s + bytes(rng.integers(0, 256, size=5, dtype=np.uint8))
    s = bytes(s)
    if len(s) < (128 << 10):
        continue
    cap = len(plain) + int(rng.choice([0, 0, 64, -1000]))
    ost, omsg, oout, oused = (inflate_util.oracle_inflate_dict(s, d, cap=cap) if d else inflate_util.oracle_inflate(s, cap=cap))
    src = torch.from_numpy(np.frombuffer(s, dtype=np.uint8).copy()).cuda()
    dst = torch.zeros(max(cap, 1) + 64, dtype=torch.uint8, device="cuda")
    win = torch.from_numpy(np.frombuffer(d, dtype=np.uint8).copy()).cuda() if d else None
    st, n, used, parts = inf.inflate_large_dev(src, dst[:max(cap, 1)] if cap > 0 else dst[:1], window=win)
    on_device += parts > 0
    if ost == -5 and st == -5:
        short += 1
        continue
    msg = zr.rocm.lib().zng_rocm_last_error().decode() if st < 0 else ""
    if st != ost or (ost == -3 and msg != omsg) or (ost == 1 and (n != len(oout) or used != oused or dst[:n].cpu().numpy().tobytes() != oout)):
        differ.append((k, len(s), args[0], args[4], len(d), (st, n, used, parts, msg), (ost, omsg, len(oout), oused)))
    elif int(dst[max(cap, 1):].max()) != 0:
        differ.append((k, "wrote behind the destination"))
    ended += ost == 1; errors += ost == -3
print("%d streams: %d end of stream (%d decoded in parts on the device), %d data errors, %d out of input / room; %.0f s; DIFFERENCES: %d"
      % (n_streams, ended, on_device, errors, short, time.perf_counter() - t0, len(differ)))
for x in differ[:10]: print("  ", x)
sys.exit(1 if differ else 0)
