"""What one copy costs the device inflater, by kind: crafted streams (a stored 32 KiB preamble, then ONE fixed-Huffman block of
N identical copies) decoded by a lone wave and by 12 / 16 waves per CU.  Times are per copy, the preamble's time subtracted.
  python tools/micro/copy_cost.py [N]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
import torch
from deflate_craft import Bits, fixed_block, stored_block
zr = importlib.import_module("zlib-ng_amd"); inf = importlib.import_module("zlib-ng_amd.inflate"); zr.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
pre = np.random.default_rng(5).integers(0, 256, 32768, dtype=np.uint8).tobytes()

def stream(tokens):
    b = Bits(); stored_block(b, pre, False); fixed_block(b, tokens, True); b.align()
    return bytes(b.out)

def run(raw, out_len, ns, reps=3):
    host = np.zeros((len(raw) + 31) & ~15, dtype=np.uint8); host[:len(raw)] = np.frombuffer(raw, dtype=np.uint8)
    comp = torch.from_numpy(host).cuda()
    cap = (out_len + 63) & ~63
    dst = torch.empty(ns * cap + 64, dtype=torch.uint8, device="cuda")
    b = inf.InflateDevBatch(comp, [0] * ns, [len(raw)] * ns, dst, [i * cap for i in range(ns)], [cap] * ns)
    b.run(); torch.cuda.synchronize()
    r = b.results.cpu()
    assert (r[:, 2] == 1).all() and (r[:, 0] == out_len).all(), r[:2]
    zr.trace_begin(reps)
    for _ in range(reps): b.run()
    torch.cuda.synchronize()
    return min(zr.trace_end(reps))

cases = [("literals only", [("L", 65)]), ("len 130 dist 1", [("M", 130, 1)]), ("len 258 dist 1", [("M", 258, 1)]), ("len 130 dist 47", [("M", 130, 47)]),
         ("len 130 dist 100", [("M", 130, 100)]), ("len 130 dist 200", [("M", 130, 200)]), ("len 32 dist 200", [("M", 32, 200)]),
         ("len 64 dist 200", [("M", 64, 200)]), ("len 258 dist 300", [("M", 258, 300)]),
         ("len 130 dist 3000", [("M", 130, 3000)]), ("len 130 dist 8000", [("M", 130, 8000)]), ("len 32 dist 8000", [("M", 32, 8000)]),
         ("len 130 dist 30000", [("M", 130, 30000)]), ("literal + len 130 dist 200", [("L", 66), ("M", 130, 200)])]
waves = [1, 12 * 256, 16 * 256]
base = [run(stream([]), len(pre), ns) for ns in waves]
print("preamble alone (ms):", ["%.3f" % x for x in base])
print("%-28s %s" % ("case", "  ".join("%5d streams: ns per unit, MB/s per wave" % w for w in waves)))
for name, unit in cases:
    toks = unit * N
    out = len(pre) + sum(1 if t[0] == "L" else t[1] for t in toks)
    raw = stream(toks)
    row = []
    for ns, b0 in zip(waves, base):
        ms = run(raw, out, ns)
        per = (ms - b0) * 1e6 / N
        row.append("%8.0f ns %7.1f MB/s" % (per, (out - len(pre)) / ((ms - b0) * 1e3)))
    print("%-28s %s" % (name, "   ".join(row)))
