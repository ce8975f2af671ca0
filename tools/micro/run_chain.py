"""profiling target: level-6 class on one 64 MiB stream, 2 runs (for rocprofv3 --pmc passes)"""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch, synth
zr = importlib.import_module("zlib-ng_amd"); dfl = importlib.import_module("zlib-ng_amd.deflate"); zr.init(0)
src = torch.from_numpy(synth.silesia_like(64 << 20, seed=0x5EED0003)).cuda()
for _ in range(2):
    dfl.deflate_dev(src, level=6)
torch.cuda.synchronize()
print("ok")
