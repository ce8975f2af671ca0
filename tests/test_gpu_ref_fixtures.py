"""GPU: the data files the reference's own tests hold (tests/golden/ref_fixtures, MANIFEST.json cites the cmake
lines) through the product's C ABI.

  * every compressed fixture through zng_rocm_uncompress2_dev: the pigz tarball (test/pigz/CMakeLists.txt:202-204)
    and packobj (test/cmake/test-issues.cmake:99-115) must come back Z_OK with the trailer -- the file's own
    CRC-32 + ISIZE / Adler-32 -- verified on the device; the four CVE streams (test/cmake/test-cves.cmake:3-12:
    "exit 0 or 1", i.e. an error or a clean end, never a fault) must give the reference's message and agree with
    the oracle inflater;
  * every plain fixture through both deflate classes at the levels the cmake files name, round-tripped through
    the oracle inflater and CPython's zlib (validity = round trip, as test/cmake/compress-and-verify.cmake:186-200)."""
import importlib
import zlib

import numpy as np
import pytest

import inflate_util
import ref_fixtures as rf
from gpu_common import product, torch_mod

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    zr = product()
    zr.init()
    return (zr, importlib.import_module("zlib-ng_amd.oneshot"), importlib.import_module("zlib-ng_amd.deflate"),
            importlib.import_module("zlib-ng_amd.inflate"))


def _dev(data, pad=16):
    torch = torch_mod()
    return torch.from_numpy(np.frombuffer(data + b"\0" * pad, dtype=np.uint8).copy()).cuda()


@pytest.mark.parametrize("entry,data", rf.compressed(), ids=lambda v: v["file"] if isinstance(v, dict) else "")
def test_compressed_fixture_on_device(entry, data, mods, oracle):
    zr, one, dfl, inf = mods
    torch = torch_mod()
    fmt = one.GZIP if entry["format"] == "gzip" else one.ZLIB
    dst = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    rc, produced, consumed, msg = one.uncompress2_dev(data, dst, fmt=fmt)
    if entry["expect"] == "Z_OK":
        assert rc == 0, msg
        d = zlib.decompressobj(31 if fmt == one.GZIP else 15)
        want = d.decompress(data)
        # packobj is a git pack object: its zlib stream is followed by 20 bytes that are not part of it
        assert d.eof and consumed == len(data) - len(d.unused_data)
        data = data[:consumed]
        got = dst[:produced].cpu().numpy().tobytes()
        assert got == want
        # the trailer the file carries, checked once more here against the device checksum kernels directly
        out = torch.zeros(2, dtype=torch.int32, device="cuda")
        zr.adler32_crc32_dev(dst, out, length=produced)
        a, c = (v & 0xffffffff for v in out.tolist())
        if fmt == one.GZIP:
            assert c == int.from_bytes(data[-8:-4], "little") and produced == int.from_bytes(data[-4:], "little")
        else:
            assert a == int.from_bytes(data[-4:], "big")
        # a damaged trailer of the same file must be refused with the reference's message (inflate.c:1127-1147)
        bad = data[:-1] + bytes([data[-1] ^ 0x55])
        rc2, _, _, msg2 = one.uncompress2_dev(bad, dst, fmt=fmt)
        assert rc2 == -3 and msg2 == ("incorrect length check" if fmt == one.GZIP else "incorrect data check")
    else:
        assert rc == -3 and msg == entry["msg"], (rc, msg)
        pos, _ = rf.gzip_payload(data)
        st, omsg, _, _ = inflate_util.oracle_inflate(data[pos:], cap=1 << 20)
        assert (st, omsg) == (-3, msg)


def test_gzip_header_crc_is_checked(mods):
    """FHCRC (RFC 1952 2.3.1; inflate.c:686-692 "header crc mismatch"): a header with the flag set is accepted when
    its CRC16 is right and refused when it is not"""
    zr, one, dfl, inf = mods
    torch = torch_mod()
    body = zlib.compressobj(6, zlib.DEFLATED, -15)
    payload = b"header crc test " * 100
    raw = body.compress(payload) + body.flush()
    hdr = bytes([0x1f, 0x8b, 8, 2 | 8, 0, 0, 0, 0, 0, 3]) + b"name\0"
    crc16 = (zlib.crc32(hdr) & 0xffff).to_bytes(2, "little")
    trailer = zlib.crc32(payload).to_bytes(4, "little") + len(payload).to_bytes(4, "little")
    dst = torch.zeros(len(payload) + 64, dtype=torch.uint8, device="cuda")
    good = hdr + crc16 + raw + trailer
    assert zlib.decompress(good, 31) == payload
    rc, produced, consumed, msg = one.uncompress2_dev(good, dst, fmt=one.GZIP)
    assert rc == 0 and produced == len(payload) and consumed == len(good), msg
    bad = hdr + bytes([crc16[0] ^ 1, crc16[1]]) + raw + trailer
    rc, _, _, msg = one.uncompress2_dev(bad, dst, fmt=one.GZIP)
    assert (rc, msg) == (-3, "header crc mismatch")


def _check_round_trip(comp, data, fmt, one):
    if fmt == one.GZIP:
        assert zlib.decompress(comp, 31) == data
        pos, _ = rf.gzip_payload(comp)
        raw, trail = comp[pos:], 8
    elif fmt == one.ZLIB:
        assert zlib.decompress(comp) == data
        raw, trail = comp[2:], 4
    else:
        d = zlib.decompressobj(-15)
        assert d.decompress(comp) == data and d.eof
        raw, trail = comp, 0
    st, msg, out, used = inflate_util.oracle_inflate(raw, cap=len(data) + 64)
    assert st == 1 and out == data and used == len(raw) - trail, (st, msg)


@pytest.mark.parametrize("entry,data", rf.plain(), ids=lambda v: v["file"] if isinstance(v, dict) else "")
def test_plain_fixture_through_both_deflate_classes(entry, data, mods):
    zr, one, dfl, inf = mods
    torch = torch_mod()
    fmt = {"gzip": one.GZIP, "raw": one.RAW}[entry["format"]]
    src = _dev(data)
    for level in entry["levels"]:
        # one-stream class (segment-parallel; level 0 = stored, 1 = single probe, 2..9 = chain walk)
        dst, clen = one.compress2_dev(src, level=level, fmt=fmt, length=len(data))
        comp = dst[:clen].cpu().numpy().tobytes()
        assert clen <= one.compress_bound(len(data), fmt)
        _check_round_trip(comp, data, fmt, one)
        if fmt == one.GZIP:
            assert comp[8] == (2 if level == 9 else 4 if level < 2 else 0)             # XFL, deflate.c:913-914
        if level == 0:
            nblk = max(1, -(-len(data) // 65535))
            body = comp[10:-8] if fmt == one.GZIP else comp
            assert len(body) == len(data) + 5 * nblk                                  # deflate_stored.c:46-95
        if level == 1 or entry.get("fixed_huffman"):
            # many-stream class: deflate_quick's static-Huffman block (what -F / Z_FIXED asks for, deflate_quick.c:47-130)
            batch = dfl.QuickBatch(src, [0], [len(data)])
            batch.run()
            torch.cuda.synchronize()
            res = batch.results.cpu()
            q = batch.compressed(0, res)
            _check_round_trip(q, data, one.RAW, one)
            assert (q[0] & 7) == 3                                                     # BFINAL = 1, BTYPE = 01
            assert (int(res[0, 1]) & 0xffffffff) == zlib.adler32(data)
    # the product's own inflate of its own level-6 stream
    dst, clen = dfl.deflate_dev(src, level=6, length=len(data))
    dec = inf.decode_tokens(dst[:clen].cpu().numpy().tobytes())
    assert dec.status == 1 and inf.resolve_dev(dec).cpu().numpy().tobytes() == data


def test_compress2_level_validation(mods):
    """deflateInit2 refuses levels outside 0..9 with Z_STREAM_ERROR (deflate.c:318-320); compress2 passes it on"""
    zr, one, dfl, inf = mods
    src = _dev(b"abc" * 100)
    for bad in (-2, 10, 99):
        with pytest.raises(zr.ZngRocmError) as e:
            one.compress2_dev(src, level=bad, fmt=one.ZLIB, length=300)
        assert "(-2)" in str(e.value)
    for level, flevel in ((0, 0), (1, 0), (2, 1), (5, 1), (6, 2), (7, 3), (9, 3)):      # FLEVEL, deflate.c:871-880
        dst, clen = one.compress2_dev(src, level=level, fmt=one.ZLIB, length=300)
        comp = dst[:clen].cpu().numpy().tobytes()
        assert (comp[1] >> 6) == flevel and ((comp[0] << 8) | comp[1]) % 31 == 0
        assert zlib.decompress(comp) == b"abc" * 100
