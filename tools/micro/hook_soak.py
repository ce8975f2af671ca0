"""a one-off soak of the coarse boundary (integration/arch/rocm/rocm_deflate.c / rocm_inflate.c driven by
tests/c/coarse_driver.c, as tests/test_coarse_hooks.py does): random level, wrapper, call sizes, output room and flush
rhythm for deflate(); random call sizes and stream sizes on both sides of the 4 MiB device-decode threshold for inflate().
   python tools/micro/hook_soak.py [runs] [seed]"""
import importlib, os, random, subprocess, sys, tempfile, time, zlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import synth
from test_gpu_deflate_fuzz import _content
zr = importlib.import_module("zlib-ng_amd")
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
tmp = tempfile.mkdtemp()
libdir = os.path.dirname(zr.lib_path())
exe = os.path.join(tmp, "coarse_driver")
arch = os.path.join(ROOT, "integration", "arch", "rocm")
subprocess.check_call(["gcc", "-std=c11", "-O2", "-DZNG_ROCM_STANDALONE_CHECK", "-DROCM_MIN_BYTES=1024", "-DROCM_INFLATE_MIN_BYTES=1",
                       "-DROCM_DEFLATE_BLOCK_BYTES=1048576", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "c"),
                       "-I" + arch, os.path.join(ROOT, "tests", "c", "coarse_driver.c")] +
                      [os.path.join(arch, f) for f in ("rocm_deflate.c", "rocm_inflate.c", "rocm_slots.c", "rocm_features.c")] +
                      ["-o", exe, "-L" + libdir, "-lzng_rocm", "-Wl,-rpath," + libdir])
def run(*args, env=None):
    p = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
    return p.returncode, p.stdout.strip()
bad, t0, dev_parts = [], time.perf_counter(), 0
fin, fout = os.path.join(tmp, "in"), os.path.join(tmp, "out")
for k in range(runs):
    n = rnd.choice([0, 1, 1000, 70000, 1 << 20, 3 << 20, 9 << 20, 20 << 20])
    n = int(n * rnd.uniform(0.6, 1.0))
    data = (_content(rnd, n) if n else np.zeros(0, dtype=np.uint8)).tobytes()
    if k % 2 == 0:                                           # deflate() through the hook
        level, wrap = rnd.choice([0, 1, 2, 4, 6, 6, 9]), rnd.randrange(2)
        in_chunk, out_chunk = rnd.choice([1000, 65536, 100000, 1 << 20, 5 << 20]), rnd.choice([100, 4096, 65536, 1 << 20])
        if n > (4 << 20) and out_chunk < 4096: out_chunk = 4096
        sync_every = rnd.choice([0, 0, 1, 3, 7])
        open(fin, "wb").write(data)
        rc, line = run("d", level, wrap, in_chunk, out_chunk, sync_every, fin, fout)
        comp = open(fout, "rb").read() if rc == 0 else b""
        try:
            back = zlib.decompress(comp) if wrap else zlib.decompressobj(-15).decompress(comp)
        except zlib.error as e:
            back = None
        if rc != 0 or back != data or not line.startswith("device %d " % n) and n >= 1024:
            bad.append(("deflate", k, n, level, wrap, in_chunk, out_chunk, sync_every, rc, line))
    else:                                                    # inflate() through the hook
        wrap = rnd.randrange(2)
        c = zlib.compressobj(rnd.choice([1, 6, 9]), zlib.DEFLATED, 15 if wrap else -15)
        comp = c.compress(data) + c.flush()
        in_chunk, out_chunk = rnd.choice([50000, 1 << 20, 1 << 24, 1 << 26]), rnd.choice([8192, 100000, 1 << 20, 1 << 22])
        if len(comp) > (2 << 20): in_chunk = max(in_chunk, 1 << 24)       # (a trickle of a large member is quadratic by design)
        open(fin, "wb").write(comp)
        rc, out = run("i", wrap, in_chunk, out_chunk, fin, fout, n, env={"COARSE_DRIVER_PARTS": "1"})
        lines = out.splitlines()
        got = open(fout, "rb").read() if rc == 0 else b""
        if rc != 0 or got != data or lines[0] != "device %d %d" % (len(comp), n):
            bad.append(("inflate", k, n, wrap, in_chunk, out_chunk, len(comp), rc, out))
        elif len(lines) > 1 and int(lines[1].split()[1]) > 0:
            dev_parts += 1
print("%d runs (%d inflate() calls decoded in parts on the device); %.0f s; FAILURES: %d" % (runs, dev_parts, time.perf_counter() - t0, len(bad)))
for x in bad[:10]: print("  ", x)
sys.exit(1 if bad else 0)
