"""where do the cycles of one batch of lz_rows_kernel go?  Needs the diagnostic library (tools/micro/rows_stamps.sh).
Prints, per phase, the cycles between consecutive stamps: average over 32 batches of one workgroup, min / mean / max over
its 16 waves."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
r = importlib.import_module("zlib-ng_amd.rocm")
r._LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libzng_rocm_stamps.so")
zr = importlib.import_module("zlib-ng_amd"); dfl = importlib.import_module("zlib-ng_amd.deflate"); zr.init(0)
n = 64 << 20
off = int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 0          # which class of the mix workgroup 3 lands in
src = torch.from_numpy(synth.silesia_like(n + off, seed=0x5EED0003)[off:]).cuda()
buf = torch.zeros(32 * 16 * 9, dtype=torch.int64, device="cuda")
L = r.lib(); L.zng_rocm_debug_rows_stamps.argtypes = [C.c_void_p]; L.zng_rocm_debug_rows_stamps.restype = None
L.zng_rocm_debug_rows_stamps(C.c_void_p(buf.data_ptr()))
for _ in range(2):
    dst, clen = dfl.deflate_dev(src, level=6)
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(32, 16, 9).astype(np.float64)
# one loop iteration of the two-batch pipeline: the front of batch k (stamps 0, 1, 2), its compares and the parse of batch
# k - 1 in either order (no stamp between them: half the waves take one order, half the other), the finish of batch k - 1
# (stamps 4 .. 7), its tokens out (8).  Stamp 3 is not set.
names = ["A reads + barrier", "insert turns", "C read, compares (k) + parse (k-1)", "exit map + barrier", "stitch chain", "path follow",
         "outputs + hist"]
d = np.diff(s[:, :, [0, 1, 2, 4, 5, 6, 7, 8]], axis=2)   # [batch, wave, phase]
print("ratio %.3f; cycles per phase (mean over 32 batches): min / mean / max over the 16 waves" % (n / clen))
for k, nm in enumerate(names):
    m = d[:, :, k].mean(axis=0)
    print("  %-36s %8.0f %8.0f %8.0f" % (nm, m.min(), m.mean(), m.max()))
tot = (s[1:, :, 0] - s[:-1, :, 0]).mean()
print("  batch to batch          %8.0f" % tot)
