"""GPU: compress2 / uncompress2 class front ends -- zlib and gzip framing with on-device trailer checksums.
Cross-checked both ways against CPython's zlib/gzip (independent codec) and the reference's messages."""
import gzip
import importlib
import zlib

import numpy as np
import pytest

import synth
from gpu_common import product, torch_mod

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def one():
    zr = product()
    zr.init()
    return importlib.import_module("zlib-ng_amd.oneshot")


def _data():
    return {
        "mix": synth.silesia_like(2 << 20, seed=5, seg_bytes=512 << 10).tobytes(),
        "empty": b"",
        "short": b"The quick brown fox jumped over the lazy dog",
        "zeros": b"\0" * 700000,
    }


def test_compress2_is_readable_by_python(one):
    torch = torch_mod()
    for name, data in _data().items():
        src = torch.from_numpy(np.frombuffer(data + b"\0" * 16, dtype=np.uint8).copy()).cuda()
        for fmt, level in ((one.ZLIB, -1), (one.ZLIB, 9), (one.GZIP, 6), (one.RAW, 2)):
            dst, clen = one.compress2_dev(src, level=level, fmt=fmt, length=len(data))
            comp = dst[:clen].cpu().numpy().tobytes()
            assert clen <= one.compress_bound(len(data), fmt)
            if fmt == one.ZLIB:
                assert zlib.decompress(comp) == data, name
                assert comp[0] == 0x78 and ((comp[0] << 8) | comp[1]) % 31 == 0
            elif fmt == one.GZIP:
                assert gzip.decompress(comp) == data, name
            else:
                assert zlib.decompressobj(-15).decompress(comp) == data, name


def test_uncompress2_reads_python_streams(one):
    torch = torch_mod()
    for name, data in _data().items():
        dst = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
        for fmt, comp in ((one.ZLIB, zlib.compress(data, 6)), (one.GZIP, gzip.compress(data, 6)),
                          (one.ZLIB, zlib.compress(data, 1) + b"trailing garbage")):
            rc, produced, consumed, msg = one.uncompress2_dev(comp, dst, fmt=fmt)
            assert rc == 0, (name, msg)
            assert produced == len(data) and dst[:produced].cpu().numpy().tobytes() == data
            assert consumed == len(comp) - (16 if comp.endswith(b"trailing garbage") else 0)


def test_uncompress2_errors_use_reference_messages(one):
    """inflate.c:528-543,1127-1147 message texts; uncompr.c:70-75 status mapping"""
    torch = torch_mod()
    data = _data()["mix"][:300000]
    dst = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
    z = bytearray(zlib.compress(data, 6))
    g = bytearray(gzip.compress(data, 6))
    bad = bytes(z[:-1]) + bytes([z[-1] ^ 1])
    assert one.uncompress2_dev(bad, dst, fmt=one.ZLIB)[::3] == (-3, "incorrect data check")
    bad = bytes(g[:-5]) + bytes([g[-5] ^ 1]) + bytes(g[-4:])
    assert one.uncompress2_dev(bad, dst, fmt=one.GZIP)[::3] == (-3, "incorrect data check")
    bad = bytes(g[:-1]) + bytes([g[-1] ^ 1])
    assert one.uncompress2_dev(bad, dst, fmt=one.GZIP)[::3] == (-3, "incorrect length check")
    bad = bytes([z[0], z[1] ^ 1]) + bytes(z[2:])
    assert one.uncompress2_dev(bad, dst, fmt=one.ZLIB)[::3] == (-3, "incorrect header check")
    flg = next(f for f in range(32) if ((0x79 << 8) | f) % 31 == 0)
    bad = bytes([0x79, flg]) + bytes(z[2:])            # CM = 9, header check still a multiple of 31
    assert one.uncompress2_dev(bad, dst, fmt=one.ZLIB)[::3] == (-3, "unknown compression method")
    assert one.uncompress2_dev(bytes(z[:len(z) // 2]), dst, fmt=one.ZLIB)[0] == -3      # incomplete stream
    small = torch.zeros(100, dtype=torch.uint8, device="cuda")
    assert one.uncompress2_dev(bytes(z), small, fmt=one.ZLIB)[0] == -5                  # Z_BUF_ERROR
    # the reference's own zlib KAT (test/test_inflate_adler32.cc)
    import json, os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "inflate_kat.json")))["zlib_stream"]
    rc, produced, consumed, msg = one.uncompress2_dev(bytes.fromhex(kat["hex"]), dst, fmt=one.ZLIB)
    assert rc == 0 and dst[:produced].cpu().numpy().tobytes() == kat["plaintext"].encode()


def test_a_large_member_is_inflated_on_the_device(one):
    """uncompress2 / inflate_raw of a member with 4 MiB and more of compressed bytes: up over PCIe once, then
    inflate_large.hip (block starts found on the device, one wavefront per part); trailer checks, too-small destinations
    and damaged streams answer as the sequential decoder does"""
    torch = torch_mod()
    zr = product()
    inf = importlib.import_module("zlib-ng_amd.inflate")
    plain = synth.silesia_like(40 << 20, seed=31).tobytes()
    gz = gzip.compress(plain, 6)
    assert len(gz) > (6 << 20)
    dst = torch.zeros(len(plain) + 64, dtype=torch.uint8, device="cuda")
    rc, produced, consumed, msg = one.uncompress2_dev(gz, dst, fmt=one.GZIP)
    assert (rc, produced, consumed) == (0, len(plain), len(gz)), (rc, msg)
    assert zr.rocm.lib().zng_rocm_inflate_large_last_parts() >= 32             # decoded in parts on the device
    assert dst[:produced].cpu().numpy().tobytes() == plain
    bad = bytearray(gz)
    bad[-6] ^= 1                                                              # CRC-32 in the trailer
    assert one.uncompress2_dev(bytes(bad), dst, fmt=one.GZIP)[::3] == (-3, "incorrect data check")
    small = torch.zeros(len(plain) - 1, dtype=torch.uint8, device="cuda")
    assert one.uncompress2_dev(gz, small, fmt=one.GZIP)[0] == -5               # Z_BUF_ERROR
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    stream = raw.compress(plain) + raw.flush()
    rc, produced = inf.inflate_raw(stream[:len(stream) * 3 // 4], dst)         # truncated: the sequential decoder's answer
    assert rc == -5 and zr.rocm.lib().zng_rocm_last_error().decode() == "input ended before the final block"
    rc, produced = inf.inflate_raw(stream, dst)
    assert (rc, produced) == (1, len(plain)) and dst[:produced].cpu().numpy().tobytes() == plain
