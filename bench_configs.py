#!/usr/bin/env python3
"""bench_configs.py -- measures every BASELINE.json config on ONE MI355X and prints one JSON document.

bench.py stays the driver's contract (configs[1], one line); this script is the wider report DESIGN.md
quotes: per config the device-resident rate, the PCIe-inclusive rate where a host buffer is handed over,
parity checks, and a CPU figure beside it (the oracle / oracle/_ref for checksums; CPython's zlib -- classic
zlib 1.2.11, an independent codec, NOT the reference -- for deflate/inflate, since zlib-ng itself cannot be
built under the round rules).

  python bench_configs.py [--quick] [--only cfg5]
"""
import argparse
import importlib
import json
import os
import statistics
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def gpu_time_ms(fn, reps, torch):
    fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        t.append(a.elapsed_time(b))
    return statistics.median(t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="smaller sizes (smoke)")
    ap.add_argument("--only", default="")
    args = ap.parse_args()

    import numpy as np
    import torch
    import oracle_lib
    import synth
    zr = importlib.import_module("zlib-ng_amd")
    dfl = importlib.import_module("zlib-ng_amd.deflate")
    inf = importlib.import_module("zlib-ng_amd.inflate")
    zr.init(0)
    orc = oracle_lib.load()
    ref_crc = oracle_lib.load_ref_crc32()
    report = {"device": torch.cuda.get_device_name(0), "host_cpus": os.cpu_count(),
              "reference_container": {
                  "what": "the REAL reference (zlib-ng 2.2.2, cmake build, AVX-512 dispatch), 1 thread, survey container, "
                          "BASELINE.md section 2 -- quoted beside the CPython-zlib figures below, which are classic zlib "
                          "1.2.11 (slower than zlib-ng) timed on this box; zlib-ng itself cannot be built under the round rules",
                  "adler32_GBps": 7.5, "crc32_GBps": 9.8, "deflate_level1_MBps": 171, "deflate_level1_ratio": 1.91,
                  "deflate_level6_MBps": 44, "deflate_level6_ratio": 2.94, "inflate_out_MBps": [318, 516]}}
    want = lambda name: not args.only or args.only == name

    # ---- cfg1: zng_adler32 over 64 MiB, CPU plumbing + bit-exact check --------------------------------
    if want("cfg1"):
        n = 64 << 20
        host = np.frombuffer(os.urandom(n), dtype=np.uint8)
        t0 = time.perf_counter()
        cpu = orc.oracle_adler32(1, host.ctypes.data, n)
        t_cpu = time.perf_counter() - t0
        t0 = time.perf_counter()
        slot = zr.adler32_z(1, host)                         # host pointer in, PCIe staging inside
        t_slot = time.perf_counter() - t0
        t0 = time.perf_counter()
        slot = zr.adler32_z(1, host)
        t_slot = min(t_slot, time.perf_counter() - t0)
        dev = torch.from_numpy(host).cuda()
        out = torch.zeros(2, dtype=torch.int32, device="cuda")
        ms = gpu_time_ms(lambda: zr.adler32_dev(dev, out), 20, torch)
        report["cfg1"] = {
            "workload": "zng_adler32 over 64 MiB /dev/urandom", "bit_exact": slot == cpu == (out[0].item() & 0xffffffff),
            "cpu_oracle_GBps": round(n / 1e9 / t_cpu, 2), "slot_host_pointer_GBps_pcie_inclusive": round(n / 1e9 / t_slot, 2),
            "device_resident_GBps": round(n / 1e9 / (ms / 1e3), 1), "device_ms": round(ms, 4)}

    # ---- cfg2 sizes: device-resident checksum kernels at 64 MiB / 256 MiB / 1 GiB ----------------------
    if want("cfg2"):
        rows = []
        sizes = [64, 256] if args.quick else [64, 256, 1024]
        big = torch.randint(0, 256, ((max(sizes) << 20) + 16,), dtype=torch.uint8, device="cuda")
        out = torch.zeros(2, dtype=torch.int32, device="cuda")
        dst = torch.empty_like(big)
        # the first kernels of a process run at idle clocks, and a process started within seconds of a heavy GPU job runs
        # 4-5 % slow for about two seconds (bench.py, SETTLE_LAUNCHES): 2.5 s of streaming first.  (The 64 MiB rows are the
        # first ones measured and the most sensitive: behind a 0.5 s settle they read 0.44-0.47 in passes that followed the
        # bench's heavy legs and 0.61 in a pass of their own.)
        for _ in range(15000):
            zr.adler32_crc32_dev(big, out, length=max(sizes) << 20)
        torch.cuda.synchronize()
        for mib in sizes:
            n = mib << 20
            row = {"MiB": mib}
            # the C ABI entry points with their arguments converted once: at 64 MiB a launch is ~13 us of GPU time and the
            # tensor -> pointer conversions of the Python wrappers cost more than that per call, which leaves the GPU idle
            # between kernels (and at a lower clock: the rows then read 1-3 us slower than tools/micro/crc_phases)
            import ctypes as C
            L = zr.lib()
            p_out = C.c_void_p(out.data_ptr())
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            # Infinity Cache control (MI355X_MICROARCH.md, Infinity Cache: a line stays resident while everything touched
            # between two of its uses fits in ~256 MiB): `rotated` walks K distinct n-byte slices of the big buffer, so
            # that (K - 1) * n * traffic >= 320 MiB pass between two reads of a line -- those rows are HBM fractions.
            # `same_buffer` re-reads one slice (at 64 and 256 MiB that is partly an Infinity Cache figure) and is kept
            # beside it.  Each figure: median over 5 passes of {settle, 50 traced launches}, with min and max.
            k_rot = max(1, min((max(sizes) << 20) // n, -(-320 // mib) + 1))
            def ptrs(k):
                return C.c_void_p(big.data_ptr() + k * n), C.c_void_p(dst.data_ptr() + k * n)
            slices = [ptrs(k) for k in range(k_rot)]
            row["rotation"] = {"slices": k_rot, "MiB_between_two_reads_of_a_line": (k_rot - 1) * mib}
            for name, call, traffic in (
                    ("adler32", lambda a, d: L.zng_rocm_adler32_dev(1, a, n, p_out, st), 1),
                    ("crc32", lambda a, d: L.zng_rocm_crc32_dev(0, a, n, p_out, st), 1),
                    ("adler32+crc32 fused", lambda a, d: L.zng_rocm_adler32_crc32_dev(1, 0, a, n, p_out, st), 1),
                    ("fold_copy (adler+crc, 2N traffic)", lambda a, d: L.zng_rocm_fold_copy_dev(3, 1, 0, d, a, n, p_out, st), 2)):
                row[name] = {}
                for mode, nsl in (("rotated", k_rot), ("same_buffer", 1)):
                    if mode == "same_buffer" and k_rot == 1:
                        row[name][mode] = "= rotated (one slice is already past the Infinity Cache)"
                        continue
                    state = {"i": 0}
                    def fn():
                        a, d = slices[state["i"] % nsl]
                        state["i"] += 1
                        call(a, d)
                    passes = []
                    for _ in range(5):
                        for _ in range(max(200, 1000 * 64 // mib)):    # clocks settle
                            fn()
                        torch.cuda.synchronize()
                        zr.trace_begin(50)
                        for _ in range(50):
                            fn()
                        passes.append(statistics.mean(zr.trace_end(50)))
                    ms = statistics.median(passes)
                    row[name][mode] = {"kernel_ms_median": round(ms, 4), "kernel_ms_min": round(min(passes), 4),
                                       "kernel_ms_max": round(max(passes), 4),
                                       "algorithmic_GBps": round(traffic * n / 1e9 / (ms / 1e3), 1),
                                       "frac_of_8TBps": round(traffic * n / 1e9 / (ms / 1e3) / 8000, 3),
                                       "frac_range": [round(traffic * n / 1e9 / (max(passes) / 1e3) / 8000, 3),
                                                      round(traffic * n / 1e9 / (min(passes) / 1e3) / 8000, 3)]}
                a0, d0 = slices[0]
                row[name]["step_ms_incl_finalize"] = round(gpu_time_ms(lambda: call(a0, d0), 10, torch), 4)
            rows.append(row)
        report["cfg2"] = rows
        del big, dst

        # CPU side of cfg2 on this box's host cores: 1 thread, and T threads each on its own
        # contiguous slice, folded with the combine operators.  adler32 = the oracle port of adler32_c.c, crc32 = the
        # reference's crc32_braid_c.c (oracle/_ref) when present, else the oracle port.
        import threading
        n = 256 << 20
        host = np.frombuffer(np.random.default_rng(2).bytes(n), dtype=np.uint8)
        crc_fn = ref_crc if ref_crc is not None else orc.oracle_crc32
        cpu_rows = {}
        for T in (1, min(64, os.cpu_count() or 1)):
            cuts = [n * k // T for k in range(T + 1)]
            part = [None] * T

            def work(k):
                p, ln = host.ctypes.data + cuts[k], cuts[k + 1] - cuts[k]
                part[k] = (orc.oracle_adler32(1, p, ln), crc_fn(0, p, ln), ln)

            best = None
            for _ in range(3):
                ths = [threading.Thread(target=work, args=(k,)) for k in range(T)]
                t0 = time.perf_counter()
                for th in ths:
                    th.start()
                for th in ths:
                    th.join()
                a, c = part[0][0], part[0][1]
                for k in range(1, T):
                    a = zr.rocm.adler32_combine(a, part[k][0], part[k][2])
                    c = zr.rocm.crc32_combine(c, part[k][1], part[k][2])
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            cpu_rows["threads_%d" % T] = {"adler32_plus_crc32_GBps": round(n / 1e9 / best, 2), "adler": a, "crc": c}
        same = len({(v["adler"], v["crc"]) for v in cpu_rows.values()}) == 1
        report["cfg2_cpu"] = {"sample": "256 MiB, adler32 then crc32 per slice, best of 3", "kind":
                              "reference crc32_braid_c.c + oracle adler32" if ref_crc is not None else "oracle port",
                              "slices_combine_to_same_value": same,
                              **{k: v["adler32_plus_crc32_GBps"] for k, v in cpu_rows.items()}}

    # ---- chunkset: the inflate copy primitive as a batch of independent copies (north_star's third kernel) ----
    # Outputs tile one region back to back (as inflate's output does); sources are drawn from another region.
    # Algorithmic bytes per copy: len read + len written + 24 of descriptors.
    if want("chunkset"):
        rows = []
        rng = np.random.default_rng(0xC0B1)
        src_bytes = (64 if args.quick else 256) << 20
        base = torch.randint(0, 256, (2 * src_bytes + 4096,), dtype=torch.uint8, device="cuda")
        for name, lens, window in (
                ("len 258 (the inflate caller's maximum)", np.full(src_bytes // 258, 258, dtype=np.uint32), False),
                ("len 256", np.full(src_bytes // 256, 256, dtype=np.uint32), False),
                ("len uniform 3..258, sources anywhere in 256 MiB", rng.integers(3, 259, size=src_bytes // 131, dtype=np.uint32), False),
                ("len uniform 3..258, sources <= 32 KiB behind (the inflate caller's window)",
                 rng.integers(3, 259, size=src_bytes // 131, dtype=np.uint32), True),
                ("len 4096", np.full(src_bytes // 4096, 4096, dtype=np.uint32), False)):
            ncopy = lens.size
            rel = np.concatenate(([0], np.cumsum(lens[:-1], dtype=np.uint64))).astype(np.uint64)
            out_off = src_bytes + rel
            if window:
                # the source of copy i sits dist in [len, 32768] bytes behind the copy's own position, in the first
                # region (the batch's copies must not depend on each other, so the "history" is a second buffer)
                dist = rng.integers(0, 32768 - 258, size=ncopy, dtype=np.uint64) + lens
                from_off = np.where(rel + 32768 > dist, rel + 32768 - dist, 0).astype(np.uint64)
            else:
                from_off = rng.integers(0, src_bytes - 4096, size=ncopy, dtype=np.uint64)
            d_out = torch.from_numpy(out_off.view(np.int64)).cuda()
            d_from = torch.from_numpy(from_off.view(np.int64)).cuda()
            d_len = torch.from_numpy(lens.view(np.int32)).cuda()
            fn = lambda: zr.rocm.chunkmemset_safe_dev(base, d_out, d_from, d_len, d_len)
            for _ in range(200):                                   # clocks settle
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50):
                fn()
            b.record()
            b.synchronize()
            ms = a.elapsed_time(b) / 50
            moved = int(lens.sum())
            alg = 2 * moved + 24 * ncopy
            # parity on a sample of the copies
            host = base.cpu().numpy()
            ok = all(bytes(host[int(out_off[k]):int(out_off[k]) + int(lens[k])])
                     == bytes(host[int(from_off[k]):int(from_off[k]) + int(lens[k])])
                     for k in rng.integers(0, ncopy, size=200))
            rows.append({"case": name, "copies": ncopy, "bytes_moved_MiB": round(moved / 2**20, 1), "kernel_ms": round(ms, 4),
                         "copied_GBps": round(moved / 1e9 / (ms / 1e3), 1),
                         "algorithmic_GBps": round(alg / 1e9 / (ms / 1e3), 1),
                         "frac_of_8TBps": round(alg / 1e9 / (ms / 1e3) / 8000, 3), "sample_bit_exact": bool(ok)})
        report["chunkset"] = rows
        del base

    # ---- cfg3: raw inflate of a level-6 stream ----------------------------------------------------------
    if want("cfg3"):
        n = (32 if args.quick else 256) << 20
        plain = synth.silesia_like(n, seed=0x5EED0003)
        t0 = time.perf_counter()
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(plain.tobytes()) + c.flush()
        t_comp = time.perf_counter() - t0
        t0 = time.perf_counter()
        back = zlib.decompressobj(-15).decompress(comp)
        t_pyinf = time.perf_counter() - t0
        assert back == plain.tobytes()
        del back
        t0 = time.perf_counter()
        dec = inf.decode_tokens(comp)
        t_dec = time.perf_counter() - t0
        assert dec.status == 1
        d_tok = torch.from_numpy(dec.tokens.view(np.int32)).cuda()
        d_lit = torch.from_numpy(dec.literals).cuda()
        d_seg = torch.from_numpy(dec.segs.view(np.int64)).cuda()
        d_sym = torch.empty(n, dtype=torch.int16, device="cuda")
        d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
        rocm = zr.rocm

        def resolve():
            rocm._check(rocm.lib().zng_rocm_inflate_resolve_dev(
                rocm._dev_ptr(d_tok), dec.tokens.size, rocm._dev_ptr(d_lit), dec.literals.size, rocm._dev_ptr(d_seg),
                dec.nsegs, rocm._dev_ptr(d_sym), rocm._dev_ptr(d_out), n, rocm._stream_ptr(None)), "resolve")
        ms = gpu_time_ms(resolve, 5, torch)
        ok = torch.equal(d_out, torch.from_numpy(plain).cuda())
        t0 = time.perf_counter()
        dst = torch.empty(n, dtype=torch.uint8, device="cuda")
        rc, produced = inf.inflate_raw(comp, dst)
        torch.cuda.synchronize()
        t_e2e = time.perf_counter() - t0
        warm = []                                                             # the same call again: scratch and slots exist now
        hs1 = inf.HostStream(comp)                                            # (and no Python copy of the stream inside the timing)
        for _ in range(3):
            t0 = time.perf_counter()
            rc_w, produced_w = inf.inflate_raw(hs1, dst)
            torch.cuda.synchronize()
            warm.append(time.perf_counter() - t0)
        t_warm = sorted(warm)[1]
        report["cfg3"] = {
            "workload": "raw inflate of a level-6 stream (CPython zlib encoder), %d MiB plaintext, Silesia-like mix" % (n >> 20),
            "compressed_MiB": round(len(comp) / 2**20, 2), "ratio": round(n / len(comp), 3), "bit_exact": bool(ok and rc == 1),
            "tokens": int(dec.tokens.size), "literal_bytes": int(dec.literals.size), "segments": int(dec.nsegs),
            "host_decode_s": round(t_dec, 3), "host_decode_out_MBps": round(n / 1e6 / t_dec, 1),
            "device_resolve_ms": round(ms, 3), "device_resolve_out_GBps": round(n / 1e9 / (ms / 1e3), 2),
            "device_algorithmic_GBps_C_plus_U": round((n + len(comp)) / 1e9 / (ms / 1e3), 2),
            "end_to_end_s_host_stream_to_device_plaintext": round(t_e2e, 3),
            "end_to_end_out_MBps": round(n / 1e6 / t_e2e, 1), "end_to_end_in_MBps": round(len(comp) / 1e6 / t_e2e, 1),
            "end_to_end_warm_s": round(t_warm, 4), "end_to_end_warm_out_MBps": round(n / 1e6 / t_warm, 1),
            "end_to_end_note": "host stream -> device plaintext through zng_rocm_inflate_raw, ONE host thread, PCIe inclusive; the first figure is "
                               "the process's first call (it allocates the scratch, and Python copies the stream once), warm = median of three more",
            "cpu_python_zlib_inflate_out_MBps_1thread": round(n / 1e6 / t_pyinf, 1),
            "cpu_python_zlib_deflate6_in_MBps_1thread": round(n / 1e6 / t_comp, 1)}
        del d_tok, d_lit, d_seg, d_sym, d_out, dst
        # the same ONE stream with its host decode on T threads (parts cut at block boundaries found by search)
        hs = inf.HostStream(comp)
        dst = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        single = {}
        for T in (4, 16, min(64, os.cpu_count() or 1)):
            inf.inflate_raw_threads(hs, dst, nthreads=T)                    # warm: the pooled pinned arrays
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                rc_t, produced_t, _ = inf.inflate_raw_threads(hs, dst, nthreads=T)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            single["threads_%d" % T] = {"end_to_end_s": round(best, 4), "out_GBps": round(n / 1e9 / best, 2),
                                        "in_GBps": round(len(comp) / 1e9 / best, 2),
                                        "parts_joined": rocm.lib().zng_rocm_inflate_threads_last_parts(),
                                        "bit_exact": bool(rc_t == 1 and produced_t == n and torch.equal(dst[:n], torch.from_numpy(plain).cuda()))}
        report["cfg3"]["one_stream_decode_on_T_threads"] = single
        del dst
        # the same ONE stream, resident in HBM, decoded on the device alone (zng_rocm_inflate_large_dev): the
        # CPython stream above, and this library's own level-6 stream of the same plaintext
        d_plain = torch.from_numpy(plain).cuda()
        own, own_len = dfl.deflate_dev(d_plain, level=6)
        on_dev = {}
        for name, d_comp in (("cpython_level6_stream", torch.from_numpy(np.frombuffer(comp, dtype=np.uint8).copy()).cuda()),
                             ("own_level6_stream", own[:own_len].contiguous())):
            dst = torch.zeros(n, dtype=torch.uint8, device="cuda")
            inf.inflate_large_dev(d_comp, dst)                                    # warm: scratch
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st_l, n_l, used_l, parts_l = inf.inflate_large_dev(d_comp, dst)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            med = statistics.median(ts)
            on_dev[name] = {"ms": round(med * 1e3, 2), "ms_range": [round(min(ts) * 1e3, 2), round(max(ts) * 1e3, 2)],
                            "out_GBps": round(n / 1e9 / med, 2), "in_GBps": round(d_comp.numel() / 1e9 / med, 2),
                            "parts": parts_l, "host_threads": 1,
                            "bit_exact": bool(st_l == 1 and n_l == n and used_l == d_comp.numel() and torch.equal(dst, d_plain))}
            del dst, d_comp
        report["cfg3"]["one_stream_on_the_device"] = on_dev
        del d_plain, own
        # many independent streams (the pigz shape): the same 256 MiB as 64 streams of 4 MiB, each compressed by the
        # device's own level-6 class (validated in cfg4), decoded on T host threads, resolved per stream on the device
        each = n // 64
        d_plain = torch.from_numpy(plain).cuda()
        parts = []
        for k in range(64):
            dstk, clk = dfl.deflate_dev(d_plain, level=6, length=each, offset=k * each)
            parts.append(dstk[:clk].cpu().numpy().tobytes())
        outs = [torch.empty(each + 64, dtype=torch.uint8, device="cuda") for _ in range(64)]
        many = {}
        prepared = inf.InflateBatch(parts, outs)
        for T in (1, 4, 16, min(64, os.cpu_count() or 1)):
            prepared.run(T)                                                  # warm: the pooled worker resources
            t0 = time.perf_counter()
            res = prepared.run(T)
            dt = time.perf_counter() - t0
            ok_m = all(r[0] == 1 and r[1] == each for r in res) and all(
                torch.equal(outs[k][:each], d_plain[k * each:(k + 1) * each]) for k in (0, 17, 63))
            many["threads_%d" % T] = {"out_GBps": round(n / 1e9 / dt, 2), "in_GBps": round(sum(map(len, parts)) / 1e9 / dt, 2),
                                      "bit_exact_sample": bool(ok_m)}
        report["cfg3"]["many_streams_64x%dMiB" % (each >> 20)] = many
        del outs, d_plain

    # ---- cfg4: deflate level 6 of one 256 MiB stream -----------------------------------------------------
    if want("cfg4"):
        n = (32 if args.quick else 256) << 20
        plain = synth.silesia_like(n, seed=0x5EED0003)
        src = torch.from_numpy(plain).cuda()
        dst, clen = dfl.deflate_dev(src, level=6)                 # warm-up (allocates the workspaces)
        torch.cuda.synchronize()
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            dst, clen = dfl.deflate_dev(src, level=6)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        zr.trace_begin(4)
        dst, clen = dfl.deflate_dev(src, level=6)
        k1 = zr.trace_end(4)
        comp = dst[:clen].cpu().numpy().tobytes()
        t0 = time.perf_counter()
        back = zlib.decompressobj(-15).decompress(comp)           # independent inflater (classic zlib)
        t_inf = time.perf_counter() - t0
        ok = back == plain.tobytes()
        del back
        dec = inf.decode_tokens(comp)                             # and the product's own inflate path
        ok2 = dec.status == 1 and torch.equal(inf.resolve_dev(dec), src)
        sample = plain[:32 << 20].tobytes()
        t0 = time.perf_counter()
        z6 = len(zlib.compress(sample, 6))
        t_z6 = time.perf_counter() - t0
        step = statistics.median(times)
        report["cfg4"] = {
            "workload": "deflate level-6 class of one %d MiB stream (Silesia-like mix), device resident, raw" % (n >> 20),
            "round_trip_ok_python_zlib": bool(ok), "round_trip_ok_product_inflate": bool(ok2),
            "ratio": round(n / clen, 3), "compressed_MiB": round(clen / 2**20, 2),
            "step_s_incl_host_sync": round(step, 4), "input_GBps": round(n / 1e9 / step, 2),
            "algorithmic_GBps_U_plus_C": round((n + clen) / 1e9 / step, 2),
            "lz_rows_kernel_ms": round(k1[0], 2) if k1 else None,
            "cpu_python_zlib_level6_in_MBps_1thread": round(len(sample) / 1e6 / t_z6, 1),
            "cpu_python_zlib_level6_ratio_on_first_32MiB": round(len(sample) / z6, 3)}
        del src, dst

    # ---- cfg5: 4096 independent 1 MiB streams, level-1 class, one GPU's view ----------------------------
    if want("cfg5"):
        nstreams = 256 if args.quick else 4096
        each = 1 << 20
        base = synth.silesia_like(96 << 20, seed=0x5EED0005, seg_bytes=1 << 20)     # 96 distinct 1 MiB slices
        host = np.concatenate([base] * ((nstreams * each + base.size - 1) // base.size))[:nstreams * each]
        src = torch.from_numpy(host).cuda()
        batch = dfl.QuickBatch(src, [i * each for i in range(nstreams)], [each] * nstreams)
        ms = gpu_time_ms(batch.run, 3, torch)
        zr.trace_begin(4)
        batch.run()
        k1 = zr.trace_end(4)
        res = batch.results.cpu()
        clen = int(res[:, 0].sum())
        sample = [0, 1, 5, 17, nstreams - 1]
        ok = all(zlib.decompressobj(-15).decompress(batch.compressed(i, res)) == host[i * each:(i + 1) * each].tobytes()
                 for i in sample)
        ok = ok and all((int(res[i, 1]) & 0xffffffff) == zlib.adler32(host[i * each:(i + 1) * each].tobytes()) for i in sample)
        t0 = time.perf_counter()
        cl1 = 0
        for i in range(16):
            cl1 += len(zlib.compress(host[i * each:(i + 1) * each].tobytes(), 1))
        t_cpu = time.perf_counter() - t0
        report["cfg5"] = {
            "workload": "%d independent 1 MiB streams, level-1 class (static Huffman), one GPU" % nstreams,
            "round_trip_sample_ok": bool(ok), "ratio": round(nstreams * each / clen, 3),
            "step_ms": round(ms, 2), "input_GBps": round(nstreams * each / 1e9 / (ms / 1e3), 2),
            "lz_kernel_ms": round(k1[0], 2) if k1 else None,
            "cpu_python_zlib_level1_in_MBps_1thread": round(16 * each / 1e6 / t_cpu, 1),
            "cpu_python_zlib_level1_ratio": round(16 * each / cl1, 3)}
        # ... and back on the device (zng_rocm_inflate_streams_dev, one wavefront per stream), every byte compared there;
        # then the same plaintext as CPython level-6 streams (dynamic Huffman blocks), 64 distinct, repeated
        clens = [int(v) for v in res[:, 0]]
        plain = torch.empty(nstreams * each + 64, dtype=torch.uint8, device="cuda")
        ib = inf.InflateDevBatch(batch.dst, batch.out_off, clens, plain, [i * each for i in range(nstreams)], [each] * nstreams)
        back_ms = gpu_time_ms(ib.run, 3, torch)
        rows = ib.results.cpu()
        back_ok = bool((rows[:, 2] == 1).all()) and rows[:, 1].tolist() == clens and torch.equal(plain[:nstreams * each], src)
        blobs = []
        for i in range(64):
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            blobs.append(c.compress(host[i * each:(i + 1) * each].tobytes()) + c.flush())
        offs, pos = [], 0
        for i in range(nstreams):
            offs.append(pos)
            pos += (len(blobs[i % 64]) + 15) & ~15
        packed = np.zeros(pos + 16, dtype=np.uint8)
        for i in range(nstreams):
            b = blobs[i % 64]
            packed[offs[i]:offs[i] + len(b)] = np.frombuffer(b, dtype=np.uint8)
        d_packed = torch.from_numpy(packed).cuda()
        plain.zero_()
        ib6 = inf.InflateDevBatch(d_packed, offs, [len(blobs[i % 64]) for i in range(nstreams)], plain,
                                  [i * each for i in range(nstreams)], [each] * nstreams)
        back6_ms = gpu_time_ms(ib6.run, 3, torch)
        want6 = torch.from_numpy(host[:64 * each]).cuda().repeat(nstreams // 64) if nstreams % 64 == 0 else None
        back6_ok = bool((ib6.results.cpu()[:, 2] == 1).all()) and (want6 is None or torch.equal(plain[:nstreams * each], want6))
        report["cfg5_inflate_dev"] = {
            "workload": "the %d streams above decoded on the device, one wavefront per stream; then the same plaintext "
                        "as CPython zlib level-6 streams" % nstreams,
            "own_streams": {"bit_exact": back_ok, "ms": round(back_ms, 2), "out_GBps": round(nstreams * each / 1e9 / (back_ms / 1e3), 2),
                            "in_GBps": round(clen / 1e9 / (back_ms / 1e3), 2)},
            "zlib_level6_streams": {"bit_exact": back6_ok, "ms": round(back6_ms, 2),
                                    "out_GBps": round(nstreams * each / 1e9 / (back6_ms / 1e3), 2),
                                    "ratio": round(64 * each / sum(len(b) for b in blobs), 3)},
            "cpu_reference_container_out_GBps": [0.318, 0.516]}
        del ib, ib6, d_packed, plain
        # a check value per stream, all in one pass (zng_rocm_checksums_dev)
        ck = torch.zeros((nstreams, 2), dtype=torch.int32, device="cuda")
        ck_off, ck_len = [i * each for i in range(nstreams)], [each] * nstreams
        ck_ms = {w: gpu_time_ms(lambda w=w: zr.checksums_dev(w, src, ck_off, ck_len, ck), 3, torch) for w in (1, 2, 3)}
        ck_kernel = {}
        for w in (1, 2, 3):
            zr.trace_begin(3)
            for _ in range(3):
                zr.checksums_dev(w, src, ck_off, ck_len, ck)
            ck_kernel[w] = statistics.mean(zr.trace_end(3))
        ckh = ck.cpu().numpy().astype(np.int64) & 0xffffffff
        ck_ok = all(ckh[i, 0] == zlib.adler32(host[i * each:(i + 1) * each].tobytes()) and
                    ckh[i, 1] == zlib.crc32(host[i * each:(i + 1) * each].tobytes()) for i in sample)
        report["cfg5_checksums"] = {
            "workload": "adler32 / crc32 / both of each of the %d streams, one pass for all (zng_rocm_checksums_dev; the "
                        "time includes building and uploading the %d descriptors)" % (nstreams, nstreams),
            "bit_exact_sample": bool(ck_ok),
            "adler32": {"ms": round(ck_ms[1], 3), "kernel_ms": round(ck_kernel[1], 3),
                        "kernel_frac_of_8TBps": round(nstreams * each / 1e9 / (ck_kernel[1] / 1e3) / 8000, 3)},
            "crc32": {"ms": round(ck_ms[2], 3), "kernel_ms": round(ck_kernel[2], 3),
                      "kernel_frac_of_8TBps": round(nstreams * each / 1e9 / (ck_kernel[2] / 1e3) / 8000, 3)},
            "both": {"ms": round(ck_ms[3], 3), "kernel_ms": round(ck_kernel[3], 3),
                     "kernel_frac_of_8TBps": round(nstreams * each / 1e9 / (ck_kernel[3] / 1e3) / 8000, 3)}}
        # the same streams as gzip members, framed and verified on the device (framing_dev.hip)
        wb = dfl.WrappedBatch(src, [i * each for i in range(nstreams)], [each] * nstreams, 2)
        gz_ms = gpu_time_ms(wb.run, 3, torch)
        wres = wb.results.cpu()
        totals = [int(v) for v in wres[:, 0]]
        gplain = torch.empty(nstreams * each + 64, dtype=torch.uint8, device="cuda")
        gb = inf.InflateDevBatch(wb.dst, wb.out_off, totals, gplain, [i * each for i in range(nstreams)], [each] * nstreams)
        gun_ms = gpu_time_ms(lambda: gb.run_wrapped(2), 3, torch)
        grows = gb.results.cpu()
        import gzip as _gzip
        report["cfg5_gzip"] = {
            "workload": "the %d streams as gzip members: level-1 class + CRC-32 of every plaintext + headers / trailers "
                        "(zng_rocm_compress_streams_dev), then header parse + inflate + CRC-32 of every output + trailer "
                        "comparison (zng_rocm_uncompress_streams_dev); nothing on the host in between" % nstreams,
            "compress_ms": round(gz_ms, 2), "compress_in_GBps": round(nstreams * each / 1e9 / (gz_ms / 1e3), 2),
            "uncompress_ms": round(gun_ms, 2), "uncompress_out_GBps": round(nstreams * each / 1e9 / (gun_ms / 1e3), 2),
            "bit_exact": bool((grows[:, 2] == 1).all()) and grows[:, 1].tolist() == totals and torch.equal(gplain[:nstreams * each], src),
            "python_gzip_reads_member_0": _gzip.decompress(wb.compressed(0, wres)) == host[:each].tobytes()}
        del wb, gb, gplain
        # the same many-stream job at pigz's default level: zng_rocm_deflate_streams_dev (chain walk + dynamic Huffman)
        sb = dfl.StreamsBatch(src, [i * each for i in range(nstreams)], [each] * nstreams)
        t0 = time.perf_counter()
        cl6 = sb.run(level=6)
        torch.cuda.synchronize()
        t6 = time.perf_counter() - t0
        t0 = time.perf_counter()
        cl6 = sb.run(level=6)
        torch.cuda.synchronize()
        t6 = min(t6, time.perf_counter() - t0)
        ok6 = all(zlib.decompressobj(-15).decompress(sb.compressed(i)) == host[i * each:(i + 1) * each].tobytes() for i in sample)
        report["cfg5_level6"] = {
            "workload": "%d independent 1 MiB streams at level 6 (zng_rocm_deflate_streams_dev), one GPU" % nstreams,
            "round_trip_sample_ok": bool(ok6), "ratio": round(nstreams * each / sum(cl6), 3), "ms": round(t6 * 1e3, 1),
            "input_GBps": round(nstreams * each / 1e9 / t6, 2), "cpu_reference_container_GBps": 0.044}

    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
