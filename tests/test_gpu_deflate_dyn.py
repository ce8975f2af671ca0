"""GPU: level-6 class deflate of one large stream (segments in parallel, dynamic Huffman on device).
Validity = round trip: CPython's zlib (independent inflater), the oracle inflater and the product's own inflate
path must all restore the input bit-exactly; ratio must be in the level-6 class."""
import importlib
import zlib

import numpy as np
import pytest

import inflate_util
import synth
from gpu_common import product, torch_mod

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    zr = product()
    zr.init()
    return zr, importlib.import_module("zlib-ng_amd.deflate"), importlib.import_module("zlib-ng_amd.inflate")


def _cases():
    rng = np.random.default_rng(23)
    mix = synth.silesia_like(3 << 20, seed=41, seg_bytes=512 << 10).tobytes()
    return {
        "mix3MiB": mix,                                   # six classes, six segments
        "empty": b"",
        "one": b"q",
        "tiny": b"abcabcabcabc",
        "zeros": b"\0" * 1500000,                         # 258-byte matches across segment borders
        "random": rng.integers(0, 256, size=700001, dtype=np.uint8).tobytes(),
        "twosyms": bytes(rng.integers(0, 2, size=600000, dtype=np.uint8) * 255),
        "skewed": bytes(np.minimum(rng.geometric(0.5, size=800000), 255).astype(np.uint8)),   # deep Huffman tree
        "segedge": mix[:(512 << 10) + 1],
        "far": rng.integers(0, 256, size=32000, dtype=np.uint8).tobytes() * 40,
    }


@pytest.mark.parametrize("level", [2, 6, 9])
def test_round_trip(mods, level):
    zr, dfl, inf = mods
    torch = torch_mod()
    for name, data in _cases().items():
        if level != 6 and name not in ("mix3MiB", "tiny", "zeros"):
            continue
        src = torch.from_numpy(np.frombuffer(data + b"\0" * 16, dtype=np.uint8).copy()).cuda()
        dst, clen = dfl.deflate_dev(src, level=level, length=len(data))
        comp = dst[:clen].cpu().numpy().tobytes()
        assert clen <= dfl.deflate_bound(len(data))
        d = zlib.decompressobj(-15)
        assert d.decompress(comp) == data and d.eof and d.unused_data == b"", name
        st, msg, out, used = inflate_util.oracle_inflate(comp, cap=len(data) + 16)
        assert st == 1 and out == data and used == len(comp), (name, st, msg)
        dec = inf.decode_tokens(comp)
        assert dec.status == 1
        assert inf.resolve_dev(dec).cpu().numpy().tobytes() == data, name


def test_ratio_is_level6_class(mods):
    """the six-class mix: zlib level 6 reaches ~2.65 (bench_configs cfg3); the device matcher must be close"""
    zr, dfl, inf = mods
    torch = torch_mod()
    data = synth.silesia_like(12 << 20, seed=0x5EED0003, seg_bytes=2 << 20)
    src = torch.from_numpy(data).cuda()
    dst, clen = dfl.deflate_dev(src, level=6)
    comp = dst[:clen].cpu().numpy().tobytes()
    assert zlib.decompressobj(-15).decompress(comp) == data.tobytes()
    z6 = len(zlib.compress(data.tobytes(), 6))
    z1 = len(zlib.compress(data.tobytes(), 1))
    print("ratio dev6 %.3f  zlib6 %.3f  zlib1 %.3f" % (data.size / clen, data.size / z6, data.size / z1))
    assert clen < 1.12 * z6


def test_block_type_choice(mods):
    """zng_tr_flush_block's choice (trees.c:660-719) per block: incompressible bytes are STORED (5 bytes per block),
    a few bytes take the STATIC code set, ordinary data a dynamic block.  BTYPE is bits 1-2 of a block's first byte."""
    zr, dfl, inf = mods
    torch = torch_mod()
    rng = np.random.default_rng(77)

    def run(data):
        src = torch.from_numpy(np.frombuffer(data + b"\0" * 16, dtype=np.uint8).copy()).cuda()
        dst, clen = dfl.deflate_dev(src, level=6, length=len(data))
        comp = dst[:clen].cpu().numpy().tobytes()
        d = zlib.decompressobj(-15)
        assert d.decompress(comp) == data and d.eof
        dec = inf.decode_tokens(comp)
        assert dec.status == 1 and inf.resolve_dev(dec).cpu().numpy().tobytes() == data
        return comp

    noise = rng.integers(0, 256, size=(1 << 20) + 777, dtype=np.uint8).tobytes()     # several segments, a block per 60 KiB
    comp = run(noise)
    pos = payload = nblocks = 0                                         # nothing but stored blocks ...
    while pos < len(comp) - 2:
        ln = int.from_bytes(comp[pos + 1:pos + 3], "little")
        assert comp[pos] == 0 and int.from_bytes(comp[pos + 3:pos + 5], "little") == ln ^ 0xffff and ln <= 65535
        pos += 5 + ln
        payload += ln
        nblocks += 1
    assert payload == len(noise) and comp[pos:] == b"\x03\x00"           # ... and the final empty static block
    # a block per 61440 positions, each followed by the empty stored block that byte-aligns it
    assert nblocks <= 2 * (len(noise) // 61440 + len(noise) // (128 << 10) + 2)
    assert comp[1:3] == (61440).to_bytes(2, "little")
    for data in (b"", b"q", b"abcabcabcabc"):
        comp = run(data)
        assert (comp[0] >> 1) & 3 == 1, data                              # static
        assert len(comp) <= len(zlib.compress(data, 6)) + 8
    text = (b"the quick brown fox jumps over the lazy dog, " * 3000)
    comp = run(text)
    assert (comp[0] >> 1) & 3 == 2                                        # dynamic
    # a stream whose segments differ: noise, then text, then noise
    mixed = noise[:512 << 10] + (text * 4)[:512 << 10] + noise[:300000]
    comp = run(mixed)
    assert len(comp) < len(mixed) - 400000


def test_three_gib_stream(mods):
    """32-bit positions: one stream of 3 GiB + 12345 bytes (6145 segments) round-trips; 4 GiB is refused.  The check
    is size-independent: CRC-32 and length of what CPython's zlib inflates == the device CRC-32 of the source."""
    zr, dfl, inf = mods
    torch = torch_mod()
    block = torch.from_numpy(synth.silesia_like(64 << 20, seed=0x5EED0033, seg_bytes=4 << 20)).cuda()
    n = (3 << 30) + 12345
    src = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
    for lo in range(0, n + 16, block.numel()):
        hi = min(n + 16, lo + block.numel())
        src[lo:hi] = block[:hi - lo]
    src[1 << 31] ^= 0x5a                                    # the copies are not all identical
    dst, clen = dfl.deflate_dev(src, level=4, length=n)
    out2 = torch.zeros(2, dtype=torch.int32, device="cuda")
    zr.crc32_dev(src, out2, length=n)
    want_crc = out2[0].item() & 0xffffffff
    comp = dst[:clen].cpu().numpy()
    d = zlib.decompressobj(-15)
    crc, total, pos = 0, 0, 0
    while pos < comp.size:                                  # streamed: never holds the 3 GiB at once
        piece = d.decompress(comp[pos:pos + (8 << 20)].tobytes())
        crc = zlib.crc32(piece, crc)
        total += len(piece)
        pos += 8 << 20
    tail = d.flush()
    crc = zlib.crc32(tail, crc)
    total += len(tail)
    assert d.eof and total == n and crc == want_crc
    assert clen < n // 2
    with pytest.raises(Exception):
        dfl.deflate_dev(src, level=4, length=(1 << 32) - (64 << 10))      # beyond the 32-bit position format


def test_many_streams_at_a_chain_level(mods):
    """zng_rocm_deflate_streams_dev: ragged streams in one call (levels 1, 6, 9), each restored by CPython's zlib, by the
    oracle inflater and -- all at once -- by the product's device inflater; byte-identical to the one-stream entry point
    when the segment size agrees; a dictionary in front of a stream; pigz-style blocks of one input concatenating into
    one stream"""
    zr, dfl, inf = mods
    torch = torch_mod()
    rng = np.random.default_rng(77)
    mix = synth.silesia_like(5 << 20, seed=43, seg_bytes=256 << 10).tobytes()
    pieces = [mix[:1 << 20], mix[1 << 20:(1 << 20) + 300001], b"", b"z", mix[2 << 20:(2 << 20) + 70000],
              b"\0" * 200000, rng.integers(0, 256, size=150000, dtype=np.uint8).tobytes(), mix[3 << 20:(3 << 20) + (700 << 10)]]
    offs, pos = [], 0
    for p in pieces:
        offs.append(pos)
        pos += (len(p) + 15) & ~15
    host = np.zeros(pos + 16, dtype=np.uint8)
    for o, p in zip(offs, pieces):
        host[o:o + len(p)] = np.frombuffer(p, dtype=np.uint8)
    src = torch.from_numpy(host).cuda()
    for level in (1, 6, 9):
        b = dfl.StreamsBatch(src, offs, [len(p) for p in pieces])
        clens = b.run(level=level)
        comps = [b.compressed(i) for i in range(len(pieces))]
        for p, c in zip(pieces, comps):
            d = zlib.decompressobj(-15)
            assert d.decompress(c) == p and d.eof and d.unused_data == b""
            st, msg, out, used = inflate_util.oracle_inflate(c, cap=len(p) + 16)
            assert (st, out, used) == (1, p, len(c)), msg
        # back through the device inflater, all streams in one launch
        plain = torch.full((pos + 64,), 0x77, dtype=torch.uint8, device="cuda")
        ib = inf.InflateDevBatch(b.dst, b.out_off, clens, plain, offs, [len(p) for p in pieces])
        ib.run()
        rows = ib.rows()
        assert all(r == (1, len(p), c, "") for r, p, c in zip(rows, pieces, clens)), rows
        got = plain.cpu().numpy()
        assert all(got[o:o + len(p)].tobytes() == p for o, p in zip(offs, pieces))
    # ratio in the level-6 class
    b = dfl.StreamsBatch(src, offs[:1], [len(pieces[0])])
    assert len(pieces[0]) / b.run(level=6)[0] > 2.0
    # pigz shape: 128 KiB blocks of one input, each primed with the 32 KiB before it, sync-flushed, concatenated
    whole = mix[:1 << 20]
    blk = 128 << 10
    w = torch.from_numpy(np.frombuffer(whole + b"\0" * 16, dtype=np.uint8).copy()).cuda()
    nb = len(whole) // blk
    flags = [dfl.BLOCK_NOT_FINAL | dfl.BLOCK_SYNC_FLUSH] * (nb - 1) + [0]
    pb = dfl.StreamsBatch(w, [i * blk for i in range(nb)], [blk] * nb, dict_len=[min(32768, i * blk) for i in range(nb)],
                          flags=flags)
    pb.run(level=6)
    joined = b"".join(pb.compressed(i) for i in range(nb))
    d = zlib.decompressobj(-15)
    assert d.decompress(joined) == whole and d.eof


def test_async_entry_point_enqueues_without_synchronising():
    """zng_rocm_deflate_async_dev: the whole level-6 pipeline enqueued on a caller's stream, the size
    in device memory; two calls on two streams overlap, both streams' outputs restore their inputs"""
    import importlib, zlib
    import torch
    zr = importlib.import_module("zlib-ng_amd")
    dfl = importlib.import_module("zlib-ng_amd.deflate")
    zr.init(0)
    plains = [synth.silesia_like(24 << 20, seed=s) for s in (101, 202)]
    srcs = [torch.from_numpy(p).cuda() for p in plains]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    dsts = [torch.empty(dfl.deflate_bound(p.size), dtype=torch.uint8, device="cuda") for p in plains]
    ress = [torch.full((2,), -1, dtype=torch.int64, device="cuda") for _ in plains]
    torch.cuda.synchronize()
    for s, d, r, st in zip(srcs, dsts, ress, streams):
        dfl.deflate_async_dev(s, d, r, level=6, stream=st)
    for st in streams:
        st.synchronize()
    for p, d, r in zip(plains, dsts, ress):
        clen, over = (int(v) for v in r.tolist())
        assert over == 0 and 0 < clen < p.size
        assert zlib.decompressobj(-15).decompress(d[:clen].cpu().numpy().tobytes()) == p.tobytes()
