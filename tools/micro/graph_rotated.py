"""64 MiB checksum steps (stream kernel + finalize launch) on ROTATED buffers, replayed from a hipGraph: what a step costs
when the host is out of the way (bench_configs.py's cfg2 rows launch from Python through ctypes; between two 15 us kernels
the queue runs dry there on some boxes).   python tools/micro/graph_rotated.py [MiB]"""
import ctypes as C, importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
zr = importlib.import_module("zlib-ng_amd"); zr.init(0)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = mib << 20
k_rot = -(-320 // mib) + 1
big = torch.randint(0, 256, (k_rot * n + 16,), dtype=torch.uint8, device="cuda")
dst = torch.empty_like(big)
out = torch.zeros(2, dtype=torch.int32, device="cuda")
L = zr.lib()
side = torch.cuda.Stream()
st = C.c_void_p(side.cuda_stream)
p_out = C.c_void_p(out.data_ptr())
calls = {"adler32": lambda a, d: L.zng_rocm_adler32_dev(1, a, n, p_out, st),
         "crc32": lambda a, d: L.zng_rocm_crc32_dev(0, a, n, p_out, st),
         "fused": lambda a, d: L.zng_rocm_adler32_crc32_dev(1, 0, a, n, p_out, st),
         "fold_copy": lambda a, d: L.zng_rocm_fold_copy_dev(3, 1, 0, d, a, n, p_out, st)}
for _ in range(3000 * 64 // mib // 8):                       # clocks
    calls["fused"](C.c_void_p(big.data_ptr()), None)
side.synchronize()
for name, call in calls.items():
    reps_in_graph = 8 * k_rot
    with torch.cuda.stream(side):
        call(C.c_void_p(big.data_ptr()), C.c_void_p(dst.data_ptr()))
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for i in range(reps_in_graph):
            k = i % k_rot
            call(C.c_void_p(big.data_ptr() + k * n), C.c_void_p(dst.data_ptr() + k * n))
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / (20 * reps_in_graph) * 1e6)
    best.sort()
    traffic = 2 if name == "fold_copy" else 1
    print("%-10s %d MiB x %d slices from a graph: %.2f us per step (stream kernel + finalize), median of 5 [%.2f..%.2f] = %.3f of 8 TB/s"
          % (name, mib, k_rot, best[2], best[0], best[-1], traffic * n / (best[2] * 1e-6) / 8e12))
