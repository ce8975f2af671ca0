"""The end of a truncated or damaged stream, three ways (no GPU): small streams of every kind cut at EVERY length with several
values of the last byte, decoded by the oracle (oracle/inflate_oracle.c), by the product's host decoder (inflate_host.cpp)
and by CPython's zlib -- classic zlib, an independent implementation of inflate.c's algorithm -- which is the arbiter here:
"needs more input", "stream end" or the data error and its text must be the same from all three.  (Found with
tools/micro/inflate_soak.py: the reference sees the unused entry of an incomplete code set after ONE bit, inftrees.c:286-293,
and asks for input before it reports a repeat code without a previous length, inflate.c:856-864.)"""
import importlib
import zlib

import numpy as np

import inflate_util
import synth


def _small_streams():
    rng = np.random.default_rng(77)
    corpus = synth.silesia_like(1 << 20, seed=77, seg_bytes=64 << 10).tobytes()
    strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]
    out = []
    for k in range(60):
        size = int(rng.choice([1, 10, 50, 200, 400]) * rng.uniform(0.5, 1.0))
        at = int(rng.integers(0, len(corpus) - size))
        plain = corpus[at:at + size]
        if k % 7 == 0:
            plain = bytes(rng.integers(0, 256, size=size, dtype=np.uint8))
        if k % 9 == 0:
            plain = bytes([int(rng.integers(0, 256))]) * (size * 40)          # one distance code: an incomplete set
        c = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, -int(rng.integers(9, 16)), int(rng.integers(1, 10)),
                             strategies[int(rng.integers(0, 5))])
        out.append(c.compress(plain) + c.flush())
    return [s for s in out if len(s) <= 300], rng


def test_every_truncation_agrees_with_classic_zlib():
    inf = importlib.import_module("zlib-ng_amd.inflate")
    seeds, rng = _small_streams()
    checked, errors = 0, 0
    for s in seeds:
        for cut in range(1, len(s) + 1):
            for v in [s[cut - 1]] + [int(x) for x in rng.integers(0, 256, size=7)]:
                t = s[:cut - 1] + bytes([v])
                z = zlib.decompressobj(-15)
                try:
                    plain = z.decompress(t)
                    want = (1 if z.eof else -5, "")
                except zlib.error as e:
                    plain, want = None, (-3, str(e).split(": ", 1)[1])
                ost, omsg, oout, _ = inflate_util.oracle_inflate(t, cap=70000)
                h = inf.decode_tokens(t)
                assert ost == want[0] and h.status == want[0], (t.hex(), want, (ost, omsg), (h.status, h.msg))
                if want[0] == -3:
                    errors += 1
                    assert omsg == want[1] and h.msg == want[1], (t.hex(), want, omsg, h.msg)
                if want[0] == 1:
                    assert oout == plain and h.out_len == len(plain), t.hex()
                checked += 1
    assert checked > 20000 and errors > 500, (checked, errors)
