"""GPU: zlib / gzip framing for many device-resident streams (zng_rocm_compress_streams_dev,
zng_rocm_uncompress_streams_dev): every step on the device.

  * what the device writes must be what ANY zlib / gzip reader accepts: CPython's zlib.decompress (wbits 15 / 31)
    restores every stream, and the trailer is the stream's own Adler-32 / CRC-32 + ISIZE;
  * what any zlib / gzip WRITER produces must come back: CPython-made zlib and gzip streams (levels 1/6/9, gzip
    headers with FEXTRA / FNAME / FCOMMENT / FHCRC) through the device path, bit-exact, bytes consumed = stream length;
  * the reference's messages for damaged wrappers (inflate.c:509-555, :686-692, :1105-1147);
  * the reference's own .gz fixtures (tests/golden/ref_fixtures)."""
import gzip
import importlib
import io
import struct
import zlib

import numpy as np
import pytest

import ref_fixtures
import synth
from gpu_common import product, torch_mod

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    zr = product()
    zr.init()
    return zr, importlib.import_module("zlib-ng_amd.deflate"), importlib.import_module("zlib-ng_amd.inflate")


def _pieces():
    rng = np.random.default_rng(91)
    mix = synth.silesia_like(2 << 20, seed=5, seg_bytes=256 << 10).tobytes()
    return [mix[:1 << 20], mix[(1 << 20):(1 << 20) + 123457], b"", b"a", b"\0" * 100000,
            rng.integers(0, 256, size=70001, dtype=np.uint8).tobytes(), mix[300000:300000 + 5000], b"abc" * 11111]


def _pack(blobs, pad=3):
    offs, pos = [], 1
    for b in blobs:
        offs.append(pos)
        pos += len(b) + pad
    host = np.zeros(pos + 64, dtype=np.uint8)
    for o, b in zip(offs, blobs):
        host[o:o + len(b)] = np.frombuffer(bytes(b), dtype=np.uint8)
    return host, offs


def _uncompress(inf, blobs, caps, fmt):
    torch = torch_mod()
    host, offs = _pack(blobs)
    out_off, pos = [], 7
    for c in caps:
        out_off.append(pos)
        pos += c + 9
    dst = torch.full((pos + 64,), 0xA5, dtype=torch.uint8, device="cuda")
    b = inf.InflateDevBatch(torch.from_numpy(host).cuda(), offs, [len(x) for x in blobs], dst, out_off, caps)
    b.run_wrapped(fmt)
    rows = b.rows()
    got = dst.cpu().numpy()
    for o, c in zip(out_off, caps):
        assert got[o + c:o + c + 9].tolist() == [0xA5] * 9
    return rows, [got[o:o + r[1]].tobytes() for o, r in zip(out_off, rows)]


@pytest.mark.parametrize("fmt", [1, 2])
def test_device_written_wrappers_are_read_by_zlib_and_by_the_device(mods, fmt):
    zr, dfl, inf = mods
    torch = torch_mod()
    pieces = _pieces()
    offs, pos = [], 0
    for p in pieces:
        offs.append(pos)
        pos += (len(p) + 15) & ~15
    host = np.zeros(pos + 16, dtype=np.uint8)
    for o, p in zip(offs, pieces):
        host[o:o + len(p)] = np.frombuffer(p, dtype=np.uint8)
    wb = dfl.WrappedBatch(torch.from_numpy(host).cuda(), offs, [len(p) for p in pieces], fmt)
    wb.run()
    res = wb.results.cpu()
    blobs = [wb.compressed(i, res) for i in range(len(pieces))]
    for p, c, r in zip(pieces, blobs, res.tolist()):
        d = zlib.decompressobj(15 if fmt == 1 else 31)
        assert d.decompress(c) == p and d.eof and d.unused_data == b""
        check = r[1] & 0xffffffff
        if fmt == 1:
            assert c[:2] == b"\x78\x01" and check == zlib.adler32(p) and c[-4:] == struct.pack(">I", check)
        else:
            assert check == zlib.crc32(p) and c[-8:] == struct.pack("<II", check, len(p) & 0xffffffff)
            assert gzip.decompress(c) == p
    # ... and back, entirely on the device
    rows, outs = _uncompress(inf, blobs, [len(p) for p in pieces], fmt)
    for p, c, r, o in zip(pieces, blobs, rows, outs):
        assert r == (1, len(p), len(c), "") and o == p


def test_streams_of_other_writers(mods):
    zr, dfl, inf = mods
    pieces = _pieces()
    zl = [zlib.compress(p, lvl) for p in pieces for lvl in (1, 6, 9)]
    want = [p for p in pieces for _ in range(3)]
    rows, outs = _uncompress(inf, zl, [len(p) for p in want], 1)
    for p, c, r, o in zip(want, zl, rows, outs):
        assert r == (1, len(p), len(c), "") and o == p
    gz = []
    for k, p in enumerate(pieces):
        buf = io.BytesIO()
        with gzip.GzipFile(filename="name-%d.bin" % k if k % 2 else "", mode="wb", fileobj=buf, compresslevel=6, mtime=1234) as f:
            f.write(p)
        gz.append(buf.getvalue())
    # hand-made headers with FEXTRA + FNAME + FCOMMENT + FHCRC in front of a raw stream
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = raw.compress(pieces[1]) + raw.flush()
    hdr = bytes([0x1f, 0x8b, 8, 4 | 8 | 16 | 2, 1, 2, 3, 4, 0, 3]) + struct.pack("<H", 5) + b"extra" + b"file.txt\0" + b"a comment\0"
    hdr += struct.pack("<H", zlib.crc32(hdr) & 0xffff)
    full = hdr + body + struct.pack("<II", zlib.crc32(pieces[1]), len(pieces[1]))
    gz.append(full)
    want = pieces + [pieces[1]]
    rows, outs = _uncompress(inf, gz, [len(p) for p in want], 2)
    for p, c, r, o in zip(want, gz, rows, outs):
        assert r == (1, len(p), len(c), "") and o == p
    # trailing bytes behind a complete member are not consumed
    rows, outs = _uncompress(inf, [gz[0] + b"more bytes"], [len(pieces[0])], 2)
    assert rows[0] == (1, len(pieces[0]), len(gz[0]), "")


def test_damaged_wrappers_give_the_reference_messages(mods):
    zr, dfl, inf = mods
    p = synth.silesia_like(50000, seed=8).tobytes()
    z, g = bytearray(zlib.compress(p, 6)), bytearray(gzip.compress(p, 6, mtime=0))

    def mut(b, i, v):
        c = bytearray(b)
        c[i] = v
        return bytes(c)

    cases = [
        (1, mut(z, 1, z[1] ^ 1), -3, "incorrect header check"),
        (1, mut(z, 0, 0x79) if ((0x79 << 8) | z[1]) % 31 == 0 else mut(mut(z, 0, 0x79), 1, (31 - ((0x79 << 8) % 31)) % 31), -3,
         "unknown compression method"),
        (1, bytes([0x88, 0x1c]) + bytes(z[2:]), -3, "invalid window size"),
        (1, bytes([0x78, 0xbb]) + bytes(z[2:]), -3, "need dictionary"),
        (1, mut(z, len(z) - 1, z[-1] ^ 0x10), -3, "incorrect data check"),
        (1, bytes(z[:-2]), -5, "input ended before the final block"),
        (1, b"\x78", -5, "input ended before the final block"),
        (2, mut(g, 0, 0x1e), -3, "incorrect header check"),
        (2, mut(g, 2, 7), -3, "unknown compression method"),
        (2, mut(g, 3, 0x80), -3, "incorrect header check"),
        (2, mut(g, len(g) - 6, g[-6] ^ 1), -3, "incorrect data check"),
        (2, mut(g, len(g) - 1, g[-1] ^ 1), -3, "incorrect length check"),
        (2, bytes(g[:-3]), -5, "input ended before the final block"),
        (2, mut(g, 20, g[20] ^ 0x40), None, None),          # damage inside the deflate data: some data error
    ]
    assert ((0x88 << 8) | 0x1c) % 31 == 0 and ((0x78 << 8) | 0xbb) % 31 == 0
    for fmt in (1, 2):
        sel = [c for c in cases if c[0] == fmt]
        rows, _ = _uncompress(inf, [c[1] for c in sel], [len(p)] * len(sel), fmt)
        for c, r in zip(sel, rows):
            if c[2] is None:
                assert r[0] in (-3, -5), r
            else:
                assert (r[0], r[3]) == (c[2], c[3]), (c[3], r)
    # FHCRC that does not match
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = raw.compress(p) + raw.flush()
    hdr = bytes([0x1f, 0x8b, 8, 2, 0, 0, 0, 0, 0, 3])
    bad = hdr + struct.pack("<H", (zlib.crc32(hdr) ^ 1) & 0xffff) + body + struct.pack("<II", zlib.crc32(p), len(p))
    rows, _ = _uncompress(inf, [bad], [len(p)], 2)
    assert (rows[0][0], rows[0][3]) == (-3, "header crc mismatch")


def test_reference_gz_fixtures(mods):
    zr, dfl, inf = mods
    blobs, fmts, expect = [], [], []
    for entry, data in ref_fixtures.compressed():
        blobs.append(data)
        fmts.append(2 if entry["format"] == "gzip" else 1)
        expect.append(entry)
    for fmt in (1, 2):
        sel = [(b, e) for b, f, e in zip(blobs, fmts, expect) if f == fmt]
        if not sel:
            continue
        rows, outs = _uncompress(inf, [b for b, _ in sel], [1 << 20] * len(sel), fmt)
        for (b, e), r, o in zip(sel, rows, outs):
            if e["expect"] == "Z_OK":
                d = zlib.decompressobj(31 if fmt == 2 else 15)
                want = d.decompress(b)
                assert r[0] == 1 and o == want and r[2] == len(b) - len(d.unused_data), (e["file"], r)
            else:
                assert r[0] == -3 and r[3] == e["msg"], (e["file"], r)


def test_wrapped_round_trip_at_cfg5_size(mods):
    """BASELINE.json configs[4] with gzip framing: 4096 x 1 MiB compressed and framed on the device, parsed, inflated and
    verified on the device; every byte and every status compared there"""
    zr, dfl, inf = mods
    torch = torch_mod()
    n, per = 4096, 1 << 20
    base = torch.from_numpy(synth.silesia_like(64 << 20, seed=33)).cuda()
    src = base.repeat(n * per // base.numel())
    wb = dfl.WrappedBatch(src, [i * per for i in range(n)], [per] * n, 2)
    wb.run()
    res = wb.results.cpu()
    total = [int(v) for v in res[:, 0]]
    dst = torch.empty(n * per + 64, dtype=torch.uint8, device="cuda")
    b = inf.InflateDevBatch(wb.dst, wb.out_off, total, dst, [i * per for i in range(n)], [per] * n)
    b.run_wrapped(2)
    torch.cuda.synchronize()
    r = b.results.cpu()
    assert (r[:, 2] == 1).all() and (r[:, 0] == per).all() and r[:, 1].tolist() == total
    assert torch.equal(dst[:n * per], src)
    c0 = wb.compressed(0, res)
    assert gzip.decompress(c0) == src[:per].cpu().numpy().tobytes()
