"""GPU parity: inflate copy-resolution stage (device) behind the host token decoder.
Bit-exact vs the plaintext / the oracle inflater; streams come from CPython's zlib (an independent
RFC 1951 encoder) at several levels, plus the reference's infcover streams."""
import importlib
import json
import os
import zlib

import numpy as np
import pytest

import inflate_util
import synth
from gpu_common import product, torch_mod

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "inflate_kat.json")))


@pytest.fixture(scope="module")
def inf():
    zr = product()
    zr.init()
    return importlib.import_module("zlib-ng_amd.inflate")


def _raw_deflate(data, level, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    return c.compress(data) + c.flush()


def test_infcover_streams_on_device(inf):
    for r in KAT["rows"]:
        src = bytes(int(t, 16) for t in r["hex"].split())
        dec = inf.decode_tokens(src)
        ost, omsg, oout, _ = inflate_util.oracle_inflate(src, cap=70000)
        got = inf.resolve_dev(dec).cpu().numpy().tobytes()
        assert (dec.status, dec.msg) == (ost, omsg)
        assert got == oout, r


def test_corpus_levels(inf):
    rng = np.random.default_rng(3)
    cases = {
        "mix": synth.silesia_like(3 << 20, seed=7, seg_bytes=512 << 10).tobytes(),
        "zeros": b"\0" * 900000,                                   # distance-1 runs across many segments
        "period7": (b"abcdefg" * 40000),
        "random": rng.integers(0, 256, size=500000, dtype=np.uint8).tobytes(),
        "empty": b"",
        "one": b"x",
        "far": (rng.integers(0, 256, size=32768, dtype=np.uint8).tobytes()) * 9,   # distance 32768 everywhere
    }
    for name, data in cases.items():
        for level in (0, 1, 6, 9):
            comp = _raw_deflate(data, level)
            dec = inf.decode_tokens(comp)
            assert dec.status == 1 and dec.out_len == len(data)
            got = inf.resolve_dev(dec).cpu().numpy().tobytes()
            assert got == data, (name, level)
    comp = _raw_deflate(cases["mix"], 6, zlib.Z_FIXED)
    assert inf.resolve_dev(inf.decode_tokens(comp)).cpu().numpy().tobytes() == cases["mix"]


def test_context_chain_across_groups(inf):
    """Worst case for the two-level context chain: after the first 32 KiB every byte is a distance-32768 (or
    distance-1 / distance-7) reference, so every symbol of every segment tail stays a reference until the
    chain has walked all the way back to segment 0, across several groups of 32 segments."""
    rng = np.random.default_rng(5)
    block = rng.integers(0, 256, size=32768, dtype=np.uint8).tobytes()
    cases = {
        "far": block * 420,                                       # 13 MiB, > 100 segments, 4 groups
        "near": block + b"\x07" * (9 << 20) + b"abcdefg" * 600000,
        "ragged": (block + b"tail") * 300 + block[:12345],
    }
    for name, data in cases.items():
        for level in (1, 6):
            comp = _raw_deflate(data, level)
            dec = inf.decode_tokens(comp)
            assert dec.status == 1 and dec.out_len == len(data) and dec.nsegs > 64, name
            got = inf.resolve_dev(dec).cpu().numpy().tobytes()
            assert got == data, (name, level)


def test_one_shot_inflate_raw_and_errors(inf):
    torch = torch_mod()
    data = synth.silesia_like(2 << 20, seed=11, seg_bytes=256 << 10).tobytes()
    comp = _raw_deflate(data, 6)
    dst = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
    rc, n = inf.inflate_raw(comp, dst)
    assert rc == 1 and n == len(data)
    assert dst[:n].cpu().numpy().tobytes() == data
    # too-small destination
    small = torch.zeros(1000, dtype=torch.uint8, device="cuda")
    rc, n = inf.inflate_raw(comp, small)
    assert rc == -5
    # corrupt stream: status, and the bytes before the bad symbol, match the oracle
    bad = bytearray(comp)
    bad[len(bad) // 2] ^= 0x10
    ost, omsg, oout, _ = inflate_util.oracle_inflate(bytes(bad), cap=4 * len(data))
    dst2 = torch.zeros(4 * len(data), dtype=torch.uint8, device="cuda")
    rc, n = inf.inflate_raw(bytes(bad), dst2)
    assert rc == ost
    if ost in (1, -3):
        assert n == len(oout) and dst2[:n].cpu().numpy().tobytes() == oout
    # truncated stream
    rc, n = inf.inflate_raw(comp[:len(comp) // 2], dst)
    assert rc == -5 and data.startswith(dst[:n].cpu().numpy().tobytes())


def test_cfg3_scale_stream(inf):
    """BASELINE.json configs[2] shape at 64 MiB of plaintext (the 256 MiB run lives in bench_configs.py):
    level-6 raw stream of the Silesia-like mix, bit-exact against the plaintext on device."""
    torch = torch_mod()
    plain = synth.silesia_like(64 << 20, seed=0x5EED0003, seg_bytes=4 << 20)
    comp = _raw_deflate(plain.tobytes(), 6)
    dec = inf.decode_tokens(comp)
    assert dec.status == 1 and dec.out_len == plain.size and dec.nsegs >= 500
    got = inf.resolve_dev(dec)
    want = torch.from_numpy(plain).cuda()
    assert torch.equal(got, want)
    # size-independent property: checksum of the device output == checksum of the plaintext (device adler/crc)
    zr = product()
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    zr.adler32_crc32_dev(got, out)
    assert [v & 0xffffffff for v in out.tolist()] == [zlib.adler32(plain.tobytes()), zlib.crc32(plain.tobytes())]


def test_output_beyond_4gib(inf):
    """64-bit output positions: a stream that inflates to 4.5 GiB + 321 bytes (about 140 k segments, 2200 groups of
    the context chain).  Size-independent check: device CRC-32 + Adler-32 of the plaintext on the device against the
    same checksums accumulated on the host while the stream is built."""
    torch = torch_mod()
    zr = product()
    rng = np.random.default_rng(91)
    piece = bytes(rng.integers(0, 4, size=64 << 10, dtype=np.uint8)) + b"\0" * ((4 << 20) - (64 << 10))   # 4 MiB
    n_pieces, tail = 1152, b"tail" * 80 + b"!"
    c = zlib.compressobj(1, zlib.DEFLATED, -15)
    comp, crc, adler = [], 0, 1
    for _ in range(n_pieces):
        comp.append(c.compress(piece))
        crc, adler = zlib.crc32(piece, crc), zlib.adler32(piece, adler)
    comp.append(c.compress(tail))
    comp.append(c.flush())
    crc, adler = zlib.crc32(tail, crc), zlib.adler32(tail, adler)
    comp = b"".join(comp)
    total = n_pieces * len(piece) + len(tail)
    assert total > (1 << 32)
    dec = inf.decode_tokens(comp)
    assert dec.status == 1 and dec.out_len == total and dec.in_used == len(comp)
    out = inf.resolve_dev(dec)
    chk = torch.zeros(2, dtype=torch.int32, device="cuda")
    zr.adler32_crc32_dev(out, chk, adler=1, crc=0, length=total)
    assert [v & 0xffffffff for v in chk.tolist()] == [adler, crc]
    assert out[-len(tail):].cpu().numpy().tobytes() == tail


def test_inflate_many_threads(inf):
    """zng_rocm_inflate_many: independent streams decoded on several host threads, each resolved on its own HIP
    stream (the pigz shape).  Mixed bag on purpose: empty, tiny, multi-segment, stored, a preset dictionary, a
    corrupted stream, a truncated one and a destination that is too small -- every job reports its own status, and
    the result does not depend on the number of threads."""
    torch = torch_mod()
    rng = np.random.default_rng(1234)
    mix = synth.silesia_like(24 << 20, seed=77, seg_bytes=1 << 20).tobytes()
    plains = [b"", b"x", mix[:70000], mix[1 << 20:(1 << 20) + (5 << 20)], bytes(rng.integers(0, 256, size=300000, dtype=np.uint8)),
              mix[7 << 20:(7 << 20) + (3 << 20)], mix[11 << 20:(11 << 20) + 1234567], b"ab" * 400000]
    plains += [mix[(12 + k) << 20:(13 + k) << 20] for k in range(10)]
    streams, windows, expect = [], [], []
    for i, p in enumerate(plains):
        streams.append(_raw_deflate(p, 1 + i % 9))
        windows.append(None)
        expect.append((1, p))
    dictionary = mix[:32768]
    c = zlib.compressobj(6, zlib.DEFLATED, -15, zdict=dictionary)
    streams.append(c.compress(mix[40000:900000]) + c.flush())
    windows.append(torch.from_numpy(np.frombuffer(dictionary, dtype=np.uint8).copy()).cuda())
    expect.append((1, mix[40000:900000]))
    bad = bytearray(streams[3])
    bad[len(bad) // 2] ^= 0x10
    ost, omsg, oout, _ = inflate_util.oracle_inflate(bytes(bad), cap=8 << 20)
    streams.append(bytes(bad))
    windows.append(None)
    expect.append((ost, oout))
    streams.append(streams[5][:len(streams[5]) // 3])                    # truncated
    windows.append(None)
    expect.append((-5, None))
    dsts = [torch.zeros(max(len(e[1]) if e[1] is not None else 4 << 20, 1) + 64, dtype=torch.uint8, device="cuda")
            for e in expect]
    streams.append(streams[2])                                           # destination too small
    windows.append(None)
    expect.append((-5, None))
    dsts.append(torch.zeros(1000, dtype=torch.uint8, device="cuda"))
    for nthreads in (1, 3, 8):
        for d in dsts:
            d.zero_()
        res = inf.inflate_many(streams, dsts, windows, nthreads=nthreads)
        for i, ((st, want), (got_st, out_len, in_used, msg)) in enumerate(zip(expect, res)):
            assert got_st == st, (nthreads, i, got_st, st, msg)
            if st == 1:
                assert out_len == len(want) and in_used == len(streams[i])
                assert dsts[i][:out_len].cpu().numpy().tobytes() == want, (nthreads, i)
            elif st == -3:
                assert msg == omsg and dsts[i][:out_len].cpu().numpy().tobytes() == want


def test_one_stream_on_many_host_threads(inf):
    """zng_rocm_inflate_raw_threads: ONE stream, host decode cut into parts at block boundaries found by search and run
    on several threads, one device pass over the joined token stream.  Bit-exact against the plaintext; irregular
    streams take the one-thread path and report what it reports."""
    torch = torch_mod()
    zr = product()
    plain = synth.silesia_like(48 << 20, seed=0x5EED0003, seg_bytes=4 << 20).tobytes()
    want = torch.from_numpy(np.frombuffer(plain, dtype=np.uint8).copy()).cuda()
    dst = torch.zeros(len(plain) + 64, dtype=torch.uint8, device="cuda")
    for level in (6, 1):
        comp = _raw_deflate(plain, level)
        for T in (2, 8, 0):
            dst.zero_()
            rc, produced, used = inf.inflate_raw_threads(comp, dst, nthreads=T)
            assert (rc, produced, used) == (1, len(plain), len(comp)), (level, T)
            assert zr.lib().zng_rocm_inflate_threads_last_parts() >= 2
            assert torch.equal(dst[:produced], want)
    # continuing a window; a window that is too short (found when the parts are joined -> the reference's message)
    dictionary = plain[:32768]
    c = zlib.compressobj(6, zlib.DEFLATED, -15, zdict=dictionary)
    body = plain[10000:10000 + (20 << 20)]
    comp = c.compress(body) + c.flush()
    d_win = want[:32768].contiguous()
    rc, produced, used = inf.inflate_raw_threads(comp, dst, window=d_win, nthreads=8)
    assert (rc, produced) == (1, len(body)) and dst[:produced].cpu().numpy().tobytes() == body
    rc, _, _ = inf.inflate_raw_threads(comp, dst, window=d_win[-50:].contiguous(), nthreads=8)
    assert rc == -3 and b"invalid distance too far back" in zr.lib().zng_rocm_last_error()
    # a damaged stream and a truncated one: same status and output as the one-thread call
    comp = bytearray(_raw_deflate(plain[:16 << 20], 6))
    comp[len(comp) // 2] ^= 0x04
    ref = torch.zeros(len(plain) + 64, dtype=torch.uint8, device="cuda")
    for bad in (bytes(comp), bytes(comp[:len(comp) // 3])):
        rc1, n1 = inf.inflate_raw(bad, ref)
        rc2, n2, _ = inf.inflate_raw_threads(bad, dst, nthreads=8)
        assert (rc1, n1) == (rc2, n2) and torch.equal(ref[:n1], dst[:n2])
    small = torch.zeros(1000, dtype=torch.uint8, device="cuda")
    assert inf.inflate_raw_threads(_raw_deflate(plain[:8 << 20], 6), small, nthreads=8)[0] == -5
