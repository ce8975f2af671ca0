#!/usr/bin/env python3
"""bench.py -- headline measurement of the arch/rocm hot path on MI355X.

Workload (BASELINE.json configs[1]): crc32 + adler32 over a 1 GiB synthetic buffer that is
already resident in HBM, block-parallel with on-device combine.  One "step" = one fused pass
of `zng_rocm_adler32_crc32_dev` over the rank's 1 GiB shard (both checksums, bytes read once).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

N > 1: weak scaling -- every rank owns its own 1 GiB shard of one N GiB logical buffer; the only
exchange is the all-gather of {adler, crc, len} (12 bytes per rank over RCCL) followed by the
ordered on-device combine.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import importlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

METRIC = "GB/s input throughput (adler32, crc32, deflate lvl6, inflate) @1/2/4/8 GPU vs CPU ref"
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SHARD_BYTES = 1 << 30           # cfg2: 1 GiB per GPU
SEED = 0x5EED0002


def cpu_baseline(host_view, reps=3):
    """oracle (port of adler32_c) + oracle/_ref (the reference's crc32_braid_c.c) on ONE host core,
    over the same bytes the GPU step reads."""
    import oracle_lib
    orc = oracle_lib.load()
    ref_crc = oracle_lib.load_ref_crc32()
    crc_fn = ref_crc if ref_crc is not None else orc.oracle_crc32_braid
    n = host_view.size
    ptr = host_view.ctypes.data
    pair, ta, tc = [], [], []
    a = c = 0
    for _ in range(reps):
        t0 = time.perf_counter()
        a = orc.oracle_adler32(1, ptr, n)
        t1 = time.perf_counter()
        c = crc_fn(0, ptr, n)
        t2 = time.perf_counter()
        ta.append(t1 - t0)
        tc.append(t2 - t1)
        pair.append(t2 - t0)
    gb = n / 1e9
    return {
        "value": round(gb / statistics.median(pair), 3),
        "unit": "GB/s",
        "cores": 1,
        "kind": "port",
        "sample": ("%d MiB of the step's own buffer, %d reps, median; one thread runs adler32 (oracle port of "
                   "adler32_c.c) then crc32 (%s)" % (n >> 20, reps,
                                                     "reference crc32_braid_c.c via oracle/_ref" if ref_crc is not None
                                                     else "oracle port of crc32_braid_c.c")),
        "adler32_GBps": round(gb / statistics.median(ta), 3),
        "crc32_GBps": round(gb / statistics.median(tc), 3),
    }, (a, c)


def pmc_traffic(kernel, nbytes):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r*_pmc_summary.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 FETCH_SIZE x2 correction applied).
    Counters cannot be read live from inside the timed run; null when no summary matches this workload."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if "_pmc_summary" in name and name.endswith(".json"):      # later files (sorted) win: newest build last
            try:
                doc = json.load(open(os.path.join(pdir, name)))
            except ValueError:
                continue
            k = doc.get("kernels", {}).get(kernel)
            if k and doc.get("bytes_per_launch_algorithmic") == nbytes and "hbm_bytes_per_launch_corrected" in k:
                best = {"bytes": k["hbm_bytes_per_launch_corrected"], "source": "profiles/" + name}
    return best


def run_streams(args, zr, par, torch, dist, dev, world, rank, rehearse):
    """BASELINE.json configs[4]: `--streams` independent 1 MiB streams, level-1 class, sharded over the ranks
    (strong scaling: the total is fixed).  Step = every rank compresses its shard (K1 parse + K2 static emit) and
    the {clen, adler32, ulen} table is all-gathered (RCCL, 24 bytes per stream) and prefix-summed on every rank."""
    import zlib

    import numpy as np
    import synth
    dfl = importlib.import_module("zlib-ng_amd.deflate")
    each = 1 << 20
    first, count = par.shard_streams(args.streams, world, rank)
    # stream i = slice (i mod 96) of a 96 MiB six-class mix (seed 0x5EED0005): every rank can rebuild any stream
    base = synth.silesia_like(96 << 20, seed=0x5EED0005, seg_bytes=1 << 20)
    idx = (np.arange(first, first + count) % 96)
    host = np.concatenate([base[i * each:(i + 1) * each] for i in idx]) if count else np.zeros(16, dtype=np.uint8)
    src = torch.from_numpy(host).to(dev)
    batch = dfl.QuickBatch(src, [i * each for i in range(count)], [each] * count)
    ulen = torch.full((count,), each, dtype=torch.int64, device=dev)
    state = {}

    def step():
        batch.run()
        res = batch.results.to(torch.int64) & 0xffffffff
        local = torch.stack([res[:, 0], res[:, 1], ulen], dim=1) if count else torch.zeros((0, 3), dtype=torch.int64, device=dev)
        if world > 1:
            tab = local.cpu() if rehearse else local
            state["table"] = par.gather_stream_table(tab, args.streams)
        else:
            state["table"] = (local, torch.cumsum(local[:, 0], 0) - local[:, 0], None)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        step()
    fence()
    zr.trace_begin(args.steps)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = zr.trace_end(args.steps)
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    table = state["table"][0].cpu()
    # parity on this rank: first and last local stream round-trip through an independent inflater
    res = batch.results.cpu()
    for i in ([0, count - 1] if count else []):
        assert zlib.decompressobj(-15).decompress(batch.compressed(i, res)) == host[i * each:(i + 1) * each].tobytes()
    if rank == 0:
        total_in = args.streams * each
        total_out = int(table[:, 0].sum())
        k_ms = statistics.mean(kernel_ms) if kernel_ms else float("nan")
        local_bytes = count * each + int(res[:, 0].to(torch.int64).sum())
        line = {
            "metric": METRIC, "value": round(total_in / 1e9 / (elapsed / args.steps), 2), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[4]: %d independent 1 MiB streams, deflate level-1 class (static Huffman), "
                                   "sharded %d-way, {clen, adler32, ulen} table all-gathered" % (args.streams, world),
                       "streams_per_gpu": count, "ratio": round(total_in / total_out, 3),
                       "parallelism": "streams/%d+allgather(24B/stream)" % world, "rehearsal_same_gpu": rehearse},
            "roofline": {"bound": "hbm", "kernel": "zr::lz_parse_kernel (scalar-issue and barrier bound LZ77 front end, see DESIGN.md 3.4)",
                         "achieved": round(local_bytes / 1e9 / (k_ms / 1e3), 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(local_bytes / 1e9 / (k_ms / 1e3) / HBM_PEAK_GBPS, 4), "traffic": None,
                         "algorithmic_bytes_per_launch": local_bytes, "avg_kernel_ms": round(k_ms, 3)},
        }
        if not args.no_cpu:
            t0 = time.perf_counter()
            done = 0
            while time.perf_counter() - t0 < 10 and done < count:
                zlib.compress(host[done * each:(done + 1) * each].tobytes(), 1)
                done += 1
            dt = time.perf_counter() - t0
            line["cpu_baseline"] = {"value": round(done * each / 1e9 / dt, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                                    "sample": "%d of this rank's streams through CPython zlib level 1 (classic zlib 1.2.11, "
                                              "an independent codec: zlib-ng itself cannot be built here), one thread" % done}
        print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=200)   # LDS-heavy kernels reach their steady clock after ~100 launches (tools/micro/fused_steady.py)
    ap.add_argument("--shard-mib", type=int, default=SHARD_BYTES >> 20)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--workload", choices=("checksum", "streams"), default="checksum",
                    help="checksum = BASELINE.json configs[1] (default, the driver's contract); "
                         "streams = configs[4]: 4096 x 1 MiB level-1 class deflate sharded over the ranks")
    ap.add_argument("--streams", type=int, default=4096)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    # ZNG_BENCH_REHEARSE=1: every rank on cuda:0 with the gloo backend -- lets the N>1 control flow be
    # rehearsed on a one-GPU box (RCCL refuses two ranks on one device); numbers from it mean nothing.
    rehearse = os.environ.get("ZNG_BENCH_REHEARSE") == "1"
    # ZNG_BENCH_FORCE_DIST=1: take the N > 1 code path (RCCL all-gather on the side stream, combine_rows) with a
    # one-rank process group -- the way to exercise that path on a one-GPU box
    force_dist = os.environ.get("ZNG_BENCH_FORCE_DIST") == "1" and world == 1
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist:
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    zr = importlib.import_module("zlib-ng_amd")      # raises if libzng_rocm.so is missing: no fallback
    zr.init(local_rank)
    par = importlib.import_module("zlib-ng_amd.parallel")

    if args.workload == "streams":
        run_streams(args, zr, par, torch, dist, dev, world, rank, rehearse)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    n = args.shard_mib << 20
    gen = torch.Generator(device=dev)
    gen.manual_seed(SEED + rank)
    buf = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev, generator=gen)
    out = torch.zeros(2, dtype=torch.int32, device=dev)
    # N > 1: each rank's {adler, crc, len} row (16 bytes, zng_rocm_check_row) is all-gathered and folded in rank order
    # by ONE device launch pair (zng_rocm_combine_rows_dev).  The exchange of step k runs on a side stream while the
    # main stream is already in step k+1's checksum kernel: the collective (latency bound: 16 bytes per rank over
    # xGMI) never sits on the critical path.  Rows rotate through k_slots buffers; a slot is reused only after the
    # side stream has released it.  The checksum kernel writes its two words straight into the row.
    multi = world > 1 or force_dist
    k_slots = 8
    rows_local = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(k_slots)]
    for r in rows_local:
        lo = n & 0xffffffff
        r[2] = lo if lo < (1 << 31) else lo - (1 << 32)
        r[3] = n >> 32
    rows_all = [torch.zeros(world * 4, dtype=torch.int32, device=dev) for _ in range(k_slots)]
    totals = [torch.zeros(2, dtype=torch.int32, device=dev) for _ in range(k_slots)]
    side = torch.cuda.Stream(device=dev, priority=-1)    # its own hardware queue: must not share the main one
    # Cross-stream synchronisation is per GROUP of k_group steps, not per step: every step still has its own all-gather
    # and its own combine, but an event-record packet in front of every checksum launch costs ~10 us of launch gap
    # (traced), so the exchanges of a group are issued together behind one event.
    k_group = 4
    n_groups = k_slots // k_group
    produced = [torch.cuda.Event() for _ in range(n_groups)]
    released = [torch.cuda.Event() for _ in range(n_groups)]
    state = {"k": 0, "last": 0}

    # Host cost matters as much as device cost here: a step is 0.18 ms of kernel, and every Python-side stream switch
    # or tensor op in the loop is tens of microseconds.  So inside the loop torch's *current* stream IS the exchange
    # stream (the collective and the combine pick it up implicitly) and the checksum kernel is launched on `main`
    # explicitly -- no per-step stream context, no tensor arithmetic.
    main = torch.cuda.current_stream()
    # ZNG_BENCH_SIMPLE_EXCHANGE=1: no second stream -- kernel, all-gather and combine in order on the main stream
    # (the plain form of the same step; kept as a switch so the overlapped form can be compared against it)
    simple = os.environ.get("ZNG_BENCH_SIMPLE_EXCHANGE") == "1"
    if multi and not rehearse and not simple:
        torch.cuda.set_stream(side)
        # the checksum grid is one workgroup per CU for the whole pass: leave a few CUs to the exchange stream, so
        # that the collective's workgroups do not have to displace one the pass then waits for
        zr.reserve_cus(int(os.environ.get("ZNG_BENCH_RESERVE_CUS", "4")))

    def exchange(group, count):
        """the exchanges of `count` finished steps of this group: one all-gather + one combine each"""
        produced[group].record(main)
        side.wait_event(produced[group])
        for j in range(count):
            slot = group * k_group + j
            par.gather_check_rows(rows_local[slot], rows_all[slot])      # RCCL all-gather on the exchange stream, 16 B per rank
            zr.combine_rows_dev(rows_all[slot], world, totals[slot], stream=side)
        released[group].record(side)

    def step():
        if not multi:
            zr.adler32_crc32_dev(buf, out, adler=1, crc=0)
            return
        slot = state["k"] % k_slots
        group, pos = divmod(slot, k_group)
        row = rows_local[slot]
        if pos == 0 and state["k"] >= k_slots and not rehearse and not simple:
            main.wait_event(released[group])             # the exchange stream is done with this group's slots
        zr.adler32_crc32_dev(buf, row, adler=1, crc=0, stream=main)       # row[0:2] <- {adler, crc}
        state["last"] = slot
        state["k"] += 1
        if rehearse:
            # gloo has no device tensors: same payload and fold, synchronously through the host
            g = par.gather_check_rows(row.cpu()).to(dev)
            zr.combine_rows_dev(g, world, totals[slot], stream=main)
        elif simple:
            par.gather_check_rows(row, rows_all[slot])   # current stream = main
            zr.combine_rows_dev(rows_all[slot], world, totals[slot], stream=main)
        elif pos == k_group - 1:
            exchange(group, k_group)

    def flush():
        """exchanges of a partly filled group (step counts that are not a multiple of k_group)"""
        pending = state["k"] % k_group
        if multi and not rehearse and not simple and pending:
            exchange((state["k"] % k_slots) // k_group, pending)
            state["k"] += k_group - pending              # the next step starts a fresh group

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    flush()
    fence()
    zr.trace_begin(args.steps)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    flush()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = zr.trace_end(args.steps)

    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()

    torch.cuda.set_stream(main)
    last = state["last"] if multi else 0
    result = [v & 0xffffffff for v in (rows_local[last][0:2] if multi else out).tolist()]
    if multi:
        # the on-device ordered combine must equal the host fold of the gathered rows
        mine = rows_local[last]
        g = par.gather_check_rows(mine.cpu()) if rehearse else par.gather_check_rows(mine)
        rows = par.check_rows_to_list(g.cpu())
        folded = par.fold_checksums(rows)
        got_total = [v & 0xffffffff for v in totals[last].tolist()]
        assert got_total == [folded[0], folded[1]], (got_total, folded)
        assert rows[rank][:2] == result and all(r[2] == n for r in rows)
    # separate single-checksum timings (outside the timed region, informational)
    extra = {}
    for name, fn in (("adler32", lambda: zr.adler32_dev(buf, out)), ("crc32", lambda: zr.crc32_dev(buf, out))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        zr.trace_begin(20)
        for _ in range(20):
            fn()
        ms = zr.trace_end(20)
        extra[name + "_kernel_GBps"] = round(n / 1e9 / (statistics.mean(ms) / 1e3), 1)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * n / 1e9 / (elapsed / args.steps)
        k_avg_ms = statistics.mean(kernel_ms) if kernel_ms else float("nan")
        achieved = n / 1e9 / (k_avg_ms / 1e3)
        line = {
            "metric": METRIC,
            "value": round(value, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: crc32 + adler32 over a %d MiB synthetic buffer per GPU, HBM-resident, "
                            "fused single pass + on-device combine" % (n >> 20),
                "bytes_per_gpu": n,
                "parallelism": ("shard%d+allgather(16B/rank, side stream, overlapped with the next step)+ordered-combine" % world)
                if multi else "single",
                "checksums": ["%08x" % result[0], "%08x" % result[1]],
                "rehearsal_same_gpu": rehearse,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "zr::stream_kernel<adler,crc> (fused pass)",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": (pmc_traffic("zr::stream_kernel<true, true, false>", n) or {}).get("bytes"),
                "traffic_source": (pmc_traffic("zr::stream_kernel<true, true, false>", n) or {}).get("source"),
                "algorithmic_bytes_per_launch": n,
                "avg_kernel_ms": round(k_avg_ms, 5),
                "launches_timed": len(kernel_ms),
            },
        }
        line.update(extra)
        if not args.no_cpu:
            host = buf.cpu().numpy()
            cb, (a, c) = cpu_baseline(host)
            assert [a, c] == result, "GPU result differs from the CPU checker: %r vs %r" % (result, [a, c])
            line["cpu_baseline"] = cb
        print(json.dumps(line), flush=True)

    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
