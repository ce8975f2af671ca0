"""world_size-2 gloo runs (CPU) of the N>1 logic: the sharded-checksum aggregate and the multi-stream table.
The exchanged payloads are the real ones ({adler, crc, len} rows; {clen, check, ulen} tables); the per-rank
checksums / compressed lengths are computed with CPython's zlib here because there is no GPU -- the collective,
the ordering and the combine are what is under test."""
import importlib
import os
import socket
import zlib

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        importlib.import_module("zlib-ng_amd")
        par = importlib.import_module("zlib-ng_amd.parallel")
        # --- sharded single checksum: ragged shards, rank order matters
        rng = np.random.default_rng(5)
        whole = rng.integers(0, 256, size=3_000_001, dtype=np.uint8).tobytes()
        cuts = [0, 1_234_567, len(whole)] if world == 2 else np.linspace(0, len(whole), world + 1).astype(int).tolist()
        mine = whole[cuts[rank]:cuts[rank + 1]]
        got = par.aggregate_checksums(zlib.adler32(mine), zlib.crc32(mine), len(mine))
        ok1 = got == (zlib.adler32(whole), zlib.crc32(whole), len(whole))
        # the packed 16-byte row form bench.py sends ({u32 adler, u32 crc, u64 len} as int32[4]); a length above 2^32
        # exercises both halves of the u64
        big = (5 << 30) + rank
        a, c = zlib.adler32(mine), zlib.crc32(mine)
        row = torch.tensor([a - (1 << 32) if a >> 31 else a, c - (1 << 32) if c >> 31 else c,
                            (big & 0xffffffff) - (1 << 32) if (big >> 31) & 1 else big & 0xffffffff, big >> 32],
                           dtype=torch.int32)
        rows = par.check_rows_to_list(par.gather_check_rows(row))
        ok1 = ok1 and rows[rank] == [a, c, big] and [r[2] for r in rows] == [(5 << 30) + k for k in range(world)]
        rows_true = par.check_rows_to_list(par.gather_check_rows(torch.tensor(
            [row[0].item(), row[1].item(), len(mine), 0], dtype=torch.int32)))
        ok1 = ok1 and par.fold_checksums(rows_true) == (zlib.adler32(whole), zlib.crc32(whole), len(whole))
        # --- multi-stream table
        nstreams = 7
        first, count = par.shard_streams(nstreams, world, rank)
        streams = [bytes([i]) * (1000 + 37 * i) for i in range(nstreams)]
        rows = [[len(zlib.compress(s, 1)), zlib.adler32(s), len(s)] for s in streams[first:first + count]]
        local = torch.tensor(rows, dtype=torch.int64).view(-1, 3)
        table, offsets, totals = par.gather_stream_table(local, nstreams)
        want = [[len(zlib.compress(s, 1)), zlib.adler32(s), len(s)] for s in streams]
        ok2 = table.tolist() == want and offsets.tolist() == np.concatenate(([0], np.cumsum([w[0] for w in want])[:-1])).tolist()
        ok2 = ok2 and totals == (sum(w[0] for w in want), sum(w[2] for w in want))
        q.put((rank, ok1, ok2))
    finally:
        dist.destroy_process_group()


def _run_world(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True, True) for r in range(world)]


def test_world_size_2_gloo():
    _run_world(2)


def test_world_size_4_gloo():
    """the driver's scaling run goes to 8 ranks; 4 here (8 CPUs in this container) covers a rank count at which the
    stream shards are uneven (7 streams over 4 ranks) and the order-preserving fold has more than one interior rank"""
    _run_world(4)


def test_shard_streams_covers_everything():
    par = importlib.import_module("zlib-ng_amd.parallel")
    for n in (0, 1, 7, 4096):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                first, count = par.shard_streams(n, world, r)
                seen += list(range(first, first + count))
            assert seen == list(range(n))
