"""timing experiment: which phase of the LZ front end dominates (outputs are invalid for ablate != 0)"""
import importlib, os, sys, statistics
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch, numpy as np, synth
zr = importlib.import_module("zlib-ng_amd"); dfl = importlib.import_module("zlib-ng_amd.deflate")
zr.init(0)
n, each = 1024, 1 << 20
base = synth.silesia_like(96 << 20, seed=0x5EED0005, seg_bytes=1 << 20)
host = np.concatenate([base] * 11)[: n * each]
src = torch.from_numpy(host).cuda()
b = dfl.QuickBatch(src, [i * each for i in range(n)], [each] * n)
b.run(); torch.cuda.synchronize()
zr.trace_begin(8)
for _ in range(3): b.run()
ms = zr.trace_end(8)
print("ablate", os.environ.get("ZNG_LZ_ABLATE", "0"), "lz_kernel_ms", [round(m, 1) for m in ms], "GB/s", round(n * each / 1e9 / (statistics.mean(ms) / 1e3), 1))
