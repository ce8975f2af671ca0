"""profiling target: level-1 class on 1024 x 1 MiB streams, 3 runs (for rocprofv3 --pmc passes)"""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch, synth
zr = importlib.import_module("zlib-ng_amd"); dfl = importlib.import_module("zlib-ng_amd.deflate"); zr.init(0)
ns, sz = 1024, 1 << 20
src = torch.from_numpy(synth.silesia_like(ns * sz, seed=0x5EED0005)).cuda()
qb = dfl.QuickBatch(src, [i * sz for i in range(ns)], [sz] * ns)
for _ in range(3):
    qb.run()
torch.cuda.synchronize()
print("ok")
