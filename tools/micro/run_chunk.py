"""profiling target: chunkmemset_safe batches moving ~256 MiB, 5 launches of ONE case (so a --pmc pass averages one case):
  python tools/micro/run_chunk.py [len256 | len4096 | mix | mixwin]
mix = lengths uniform 3..258 with sources anywhere in 256 MiB; mixwin = the same lengths, sources <= 32 KiB behind
(bench_configs.py's chunkset rows).  Prints the algorithmic bytes per launch (2 x moved + 24 per copy)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
zr = importlib.import_module("zlib-ng_amd"); zr.init(0)
case = sys.argv[1] if len(sys.argv) > 1 else "len256"
rng = np.random.default_rng(0xC0B1)
src_bytes = 256 << 20
base = torch.randint(0, 256, (2 * src_bytes + 4096,), dtype=torch.uint8, device="cuda")
if case.startswith("len"):
    ln = int(case[3:])
    lens = np.full(src_bytes // ln, ln, dtype=np.uint32)
else:
    lens = rng.integers(3, 259, size=src_bytes // 131, dtype=np.uint32)
rel = np.concatenate(([0], np.cumsum(lens[:-1], dtype=np.uint64))).astype(np.uint64)
out_off = (src_bytes + rel).astype(np.uint64)
if case == "mixwin":
    dist = rng.integers(0, 32768 - 258, size=lens.size, dtype=np.uint64) + lens
    from_off = np.where(rel + 32768 > dist, rel + 32768 - dist, 0).astype(np.uint64)
else:
    from_off = rng.integers(0, src_bytes - 4096, size=lens.size, dtype=np.uint64)
d_out = torch.from_numpy(out_off.view(np.int64)).cuda()
d_from = torch.from_numpy(from_off.view(np.int64)).cuda()
d_len = torch.from_numpy(lens.view(np.int32)).cuda()
for _ in range(5):
    zr.rocm.chunkmemset_safe_dev(base, d_out, d_from, d_len, d_len)
torch.cuda.synchronize()
# 64-byte sectors the sources touch, each counted once per copy (what line granularity alone costs)
first = from_off // 64
last = (from_off + lens - 1) // 64
print("case %s: %d copies, moved %d bytes, algorithmic %d bytes per launch; source bytes at 64-byte granularity %d"
      % (case, lens.size, int(lens.sum()), 2 * int(lens.sum()) + 24 * lens.size, int(((last - first + 1) * 64).sum())))
