"""CPU: the oracle and the product's host token decoder against the data files the reference's own tests hold
(tests/golden/ref_fixtures, MANIFEST.json).  The .gz / zlib fixtures carry their own answer -- a CRC-32 + ISIZE or
an Adler-32 trailer written by whoever made them -- so they pin the oracle's inflate AND its crc32 / adler32 on
real data; the CVE streams pin the error path (test/cmake/test-cves.cmake:3-12: an error or a clean end, no crash)."""
import importlib
import zlib

import numpy as np
import pytest

import inflate_util
import ref_fixtures as rf
from test_inflate_host import replay


def _raw_and_trailer(entry, data):
    if entry["format"] == "gzip":
        pos, _ = rf.gzip_payload(data)
        return data[pos:], 8
    assert data[0] == 0x78 and ((data[0] << 8) | data[1]) % 31 == 0
    return data[2:], 4


@pytest.mark.parametrize("entry,data", rf.compressed(), ids=lambda v: v["file"] if isinstance(v, dict) else "")
def test_compressed_fixture_oracle_and_host_decoder(entry, data, oracle, refcrc):
    raw, trail = _raw_and_trailer(entry, data)
    st, msg, out, used = inflate_util.oracle_inflate(raw, cap=1 << 20)
    importlib.import_module("zlib-ng_amd")
    inf = importlib.import_module("zlib-ng_amd.inflate")
    dec = inf.decode_tokens(raw)                               # the product's host stage (no GPU needed)
    assert (dec.status, dec.msg) == (st, msg)
    if entry["expect"] == "Z_OK":
        assert st == 1 and msg == "", (st, msg)
        assert replay(dec) == out and dec.in_used == used
        t = raw[used:used + trail]
        arr = np.frombuffer(out, dtype=np.uint8)
        if entry["format"] == "gzip":
            want_crc = int.from_bytes(t[:4], "little")
            assert int.from_bytes(t[4:8], "little") == len(out)                    # ISIZE
            assert oracle.oracle_crc32(0, arr.ctypes.data, arr.size) == want_crc   # the file's own CRC-32
            assert oracle.oracle_crc32_braid(0, arr.ctypes.data, arr.size) == want_crc
            if refcrc is not None:
                assert refcrc(0, arr.ctypes.data, arr.size) == want_crc
            assert zlib.decompress(data, 31) == out
        else:
            assert oracle.oracle_adler32(1, arr.ctypes.data, arr.size) == int.from_bytes(t[:4], "big")
            assert zlib.decompress(data) == out
    else:
        assert st == -3 and msg == entry["msg"], (st, msg)
        with pytest.raises(zlib.error) as e:
            zlib.decompress(data, 31)
        assert entry["msg"] in str(e.value)                    # classic zlib shares the message texts


@pytest.mark.parametrize("entry,data", rf.plain(), ids=lambda v: v["file"] if isinstance(v, dict) else "")
def test_plain_fixture_checksums_oracle_vs_independent(entry, data, oracle, refcrc):
    arr = np.frombuffer(data, dtype=np.uint8)
    assert oracle.oracle_adler32(1, arr.ctypes.data, arr.size) == zlib.adler32(data)
    assert oracle.oracle_crc32(0, arr.ctypes.data, arr.size) == zlib.crc32(data)
    if refcrc is not None:
        assert refcrc(0, arr.ctypes.data, arr.size) == zlib.crc32(data)
