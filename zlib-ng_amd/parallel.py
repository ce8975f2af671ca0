"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the box,
"gloo" in CPU tests).  The hot path shards two ways and neither moves bulk data:

  * one big checksum: rank r owns the r-th contiguous range; the ONLY exchange is an all-gather of
    {adler, crc, len} (16 bytes per rank, zng_rocm_check_row) followed by the ordered combine (adler32_combine / crc32_combine
    are associative but NOT commutative, so rank order is kept);
  * independent streams (pigz-style): stream i lives on rank i // per_rank; the exchange is an all-gather of
    the per-stream table {clen, check, ulen}, after which every rank can place any stream in a global archive
    (exclusive prefix sum of clen).
"""
import torch
import torch.distributed as dist

from . import rocm


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def gather_rows(row, group=None):
    """all-gather one int64 row per rank -> [world, len(row)] tensor on the row's device"""
    world, _ = _world(group)
    row = row.contiguous()
    if world == 1:
        return row.view(1, -1)
    out = torch.empty((world * row.numel(),), dtype=row.dtype, device=row.device)
    dist.all_gather_into_tensor(out, row, group=group)
    return out.view(world, row.numel())


def gather_check_rows(row, out=None, group=None):
    """all-gather one packed checksum row per rank: `row` = int32[4] holding a zng_rocm_check_row
    {u32 adler, u32 crc, u64 len} (16 bytes).  Returns int32[world * 4] in rank order -- what
    zng_rocm_combine_rows_dev folds.  `out` (preallocated, same device) makes the call allocation-free, so it can sit
    on a side stream behind the checksum kernel (bench.py)."""
    world, _ = _world(group)
    row = row.contiguous()
    if out is None:
        out = torch.empty((world * 4,), dtype=torch.int32, device=row.device)
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        out.copy_(row)
        return out
    dist.all_gather_into_tensor(out, row, group=group)
    return out


def check_rows_to_list(rows_i32):
    """int32[world * 4] packed rows -> [[adler, crc, len], ...] (host ints)"""
    out = []
    for a, c, lo, hi in rows_i32.view(-1, 4).tolist():
        out.append([a & 0xffffffff, c & 0xffffffff, (lo & 0xffffffff) | ((hi & 0xffffffff) << 32)])
    return out


def fold_checksums(rows):
    """ordered left fold of [[adler, crc, len], ...] with the reference's combine functions
    (zng_adler32_combine adler32.c:66-68, zng_crc32_combine crc32_braid_comb.c:44-46)"""
    adler, crc, total = 1, 0, 0
    for a, c, n in rows:
        adler = rocm.adler32_combine(adler, int(a) & 0xffffffff, int(n))
        crc = rocm.crc32_combine(crc, int(c) & 0xffffffff, int(n))
        total += int(n)
    return adler, crc, total


def aggregate_checksums(local_adler, local_crc, local_len, device="cpu", group=None):
    """every rank gets (adler32, crc32, total_len) of the concatenation of all ranks' ranges"""
    row = torch.tensor([int(local_adler) & 0xffffffff, int(local_crc) & 0xffffffff, int(local_len)],
                       dtype=torch.int64, device=device)
    rows = gather_rows(row, group).cpu().tolist()
    return fold_checksums(rows)


def shard_streams(nstreams, world, rank):
    """contiguous ranges of streams per rank (stream i -> rank i // ceil(n/world)): (first, count)"""
    per = (nstreams + world - 1) // world
    first = min(rank * per, nstreams)
    return first, max(0, min(per, nstreams - first))


def gather_stream_table(local_table, nstreams, group=None):
    """local_table: int64 [count, 3] = {clen, check, ulen} of this rank's streams (contiguous shard).
    Returns (table [nstreams, 3], offsets [nstreams] = exclusive prefix sum of clen, totals (sum clen, sum ulen))."""
    world, rank = _world(group)
    per = (nstreams + world - 1) // world
    padded = torch.zeros((per, 3), dtype=torch.int64, device=local_table.device)
    padded[:local_table.shape[0]] = local_table
    allrows = gather_rows(padded.view(-1), group).view(world * per, 3)[:nstreams]
    offsets = torch.cumsum(allrows[:, 0], 0) - allrows[:, 0]
    return allrows, offsets, (int(allrows[:, 0].sum()), int(allrows[:, 2].sum()))
