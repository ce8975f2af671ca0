"""The device inflater's hand-written decode loop (inflate_dev.hip, ZR_INFLATE_FAST_LOOP; the loop it stands for is
inffast_tpl.h:151-298) at the edges of its fast paths, on streams built token by token (tests/deflate_craft.py): copies
of every interesting length at every interesting distance -- the ring's reach (kNear = 3838), the flush threshold (2048
unflushed bytes), 64 and 65 bytes, overlap by one, the very first byte, one byte too far -- literal runs around 64, output
room that ends inside a copy, a dictionary in front of the output; and, for part mode, copies that lie in front of the
part, cross into it, or touch its first byte.  Checked against CPython's zlib (an independent inflater) and the oracle."""
import importlib
import zlib

import numpy as np
import pytest

import deflate_craft as craft
import inflate_util
from test_gpu_inflate_dev import _run, inf  # noqa: F401  (the fixture and the batch runner)

pytestmark = pytest.mark.gpu


def _filler(rng, n):
    """n literal tokens of text-like bytes"""
    return [("L", int(b)) for b in rng.integers(97, 123, size=n)]


def _stream(tokens):
    b = craft.Bits()
    craft.fixed_block(b, tokens, True)
    b.align()
    return bytes(b.out)


def _cases():
    rng = np.random.default_rng(0xC0FFEE)
    cases = []
    lens = [3, 4, 15, 16, 17, 63, 64, 65, 66, 127, 128, 257, 258]
    # 1. every length at distances around it (overlap by one, exact fit, one more), at 1 and 2 (runs)
    for ln in lens:
        for dist in sorted({1, 2, ln - 1, ln, ln + 1, 300}):
            if dist >= 1:
                cases.append(_filler(rng, 400) + [("M", ln, dist)] + _filler(rng, 5))
    # 2. distances around the ring's reach and the window's end, after enough output
    for dist in (3837, 3838, 3839, 3840, 4095, 4096, 4097, 8191, 8192, 20000, 32767, 32768):
        for ln in (3, 64, 65, 258):
            cases.append(_filler(rng, 33000) + [("M", ln, dist)] + _filler(rng, 70) + [("M", ln, dist)])
    # 3. the very first byte, and literal runs of 63 / 64 / 65 / 128 / 129 in front of a copy
    for run in (1, 2, 3, 62, 63, 64, 65, 127, 128, 129, 200):
        cases.append(_filler(rng, run) + [("M", min(run, 3) if run >= 3 else 3, run)] if run >= 3 else _filler(rng, 3) + [("M", 3, 3)])
        cases.append(_filler(rng, run) + [("M", 64, 1)] + _filler(rng, run) + [("M", 64, run)])
    # 4. the flush threshold: a copy that starts 0..3 bytes around 2048 / 4096 unflushed bytes, near and far sources
    for at in (2040, 2047, 2048, 2049, 4095, 4096, 4097, 6143, 6144):
        for ln, dist in ((64, 100), (64, 4000), (258, 100), (3, 1), (65, 64)):
            if dist <= at:
                cases.append(_filler(rng, at) + [("M", ln, dist)] + _filler(rng, 300) + [("M", ln, dist)])
    # 5. long chains of copies with no literal between them (the waiting run is empty at every copy)
    cases.append(_filler(rng, 5000) + [("M", int(l), int(d)) for l, d in zip(rng.integers(3, 259, 400), rng.integers(1, 5000, 400))])
    return cases


def test_copies_and_literal_runs_at_the_edges_of_the_fast_paths(inf):
    cases = _cases()
    streams = [_stream(t) for t in cases]
    plains = [craft.replay(t) for t in cases]
    for s, p in zip(streams[::17], plains[::17]):                    # the builder itself, against an independent inflater
        assert zlib.decompressobj(-15).decompress(s) == p
    rows, outs = _run(inf, streams, [len(p) for p in plains])
    for k, (r, o, p, s) in enumerate(zip(rows, outs, plains, streams)):
        assert r[0] == 1 and r[1] == len(p) and r[3] == "", (k, r)
        assert o == p, (k, cases[k][-3:])
    # the same streams with room that ends inside / right behind the last copy: status and bytes written as the oracle's
    short = [max(0, len(p) - 1 - (k % 70)) for k, p in enumerate(plains)]
    rows, outs = _run(inf, streams, short)
    for k, (r, o, p) in enumerate(zip(rows, outs, plains)):
        assert r[0] == -5 and r[3] == "output buffer too small", (k, r)
        assert r[1] <= short[k] and o == p[:r[1]], k


def test_one_byte_too_far_back_is_the_references_error(inf):
    rng = np.random.default_rng(5)
    streams, expect = [], []
    for n in (0, 1, 63, 64, 65, 2048, 3838, 5000):
        toks = _filler(rng, n) + [("M", 3, n + 1)]
        streams.append(_stream(toks))
        expect.append(inflate_util.oracle_inflate(streams[-1], cap=n + 100)[:2])
    rows, _ = _run(inf, streams, [10000] * len(streams))
    for r, e in zip(rows, expect):
        assert (r[0], r[3]) == e == (-3, "invalid distance too far back"), (r, e)


def test_copies_out_of_a_dictionary_and_across_its_end(inf):
    rng = np.random.default_rng(6)
    dic = bytes(rng.integers(65, 91, size=32768, dtype=np.uint8))
    cases = []
    for n in (0, 1, 10, 64, 100, 3000):                              # n bytes of own output, then copies reaching past them
        for ln in (3, 64, 65, 258):
            for back in (1, ln - 1, ln, ln + 1, 5000, 32768 - n):     # how far in front of the output the source starts
                if 1 <= back and n + back <= 32768 + n and back <= 32768:
                    cases.append(_filler(rng, n) + [("M", ln, n + back)] + _filler(rng, 3) + [("M", ln, min(n + back + ln + 3, 32768))])
    streams = [_stream(t) for t in cases]
    plains = [craft.replay(t, history=dic) for t in cases]
    z = zlib.decompressobj(-15, zdict=dic)
    assert z.decompress(streams[5]) == plains[5]
    rows, outs = _run(inf, streams, [len(p) for p in plains], dicts=[dic] * len(streams))
    for k, (r, o, p) in enumerate(zip(rows, outs, plains)):
        assert r[0] == 1 and r[1] == len(p), (k, r)
        assert o == p, k


def test_part_mode_copies_in_front_of_across_and_behind_the_part_start():
    """ONE large stream (zng_rocm_inflate_large_dev): stored blocks of noise -- every one a part start -- and, behind a
    sync marker, fixed-Huffman blocks whose copies reach in front of their own part by every kind of margin"""
    torch = importlib.import_module("torch")
    zr = importlib.import_module("zlib-ng_amd")
    zr.init(0)
    inflate = importlib.import_module("zlib-ng_amd.inflate")
    rng = np.random.default_rng(7)
    b = craft.Bits()
    plain = bytearray()

    def noise_block(n):
        data = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
        craft.stored_block(b, data, False)
        plain.extend(data)

    def crafted(tokens):
        craft.stored_block(b, b"", False)                            # 00 00 ff ff: the block behind it is a part start
        craft.fixed_block(b, tokens, False)
        plain.extend(craft.replay(tokens, history=bytes(plain[-32768:])))

    for _ in range(3):
        noise_block(40000)
    for n_own in (0, 1, 5, 63, 64, 65, 300, 2047, 2048, 2049, 5000):
        toks = _filler(rng, n_own)
        for ln in (3, 64, 65, 258):
            for margin in (-1, 0, 1, 40):                            # source ends margin bytes in front of the part's first byte
                dist = n_own + ln + margin                           # margin < 0: the copy crosses into the part
                if 1 <= dist <= 32768:
                    toks += [("M", ln, dist)] + _filler(rng, 2)
                    n_own += ln + 2
        toks += [("M", 258, 32768), ("M", 64, 32768), ("M", 3, n_own + 3 + 258 + 64 + 1)]
        crafted(toks)
        noise_block(3000)
    # far and near copies deep inside a part, in 16-bit symbols, around its flush threshold
    toks = _filler(rng, 2000)
    for at in range(40):
        toks += [("M", 64, 1900 + at), ("M", 65, 3838 + at - 20), ("M", 258, 4200)] + _filler(rng, 61 + at % 5)
    crafted(toks)
    for _ in range(2):
        noise_block(40000)
    craft.stored_block(b, b"the end", True)
    plain.extend(b"the end")
    comp = bytes(b.out)
    assert len(comp) >= (128 << 10)
    assert zlib.decompressobj(-15).decompress(comp) == bytes(plain)  # the builder, against an independent inflater
    src = torch.from_numpy(np.frombuffer(comp, dtype=np.uint8).copy()).cuda()
    dst = torch.zeros(len(plain) + 64, dtype=torch.uint8, device="cuda")
    st, n, used, parts = inflate.inflate_large_dev(src, dst)
    assert (st, n, used) == (1, len(plain), len(comp)), (st, n, used, zr.rocm.lib().zng_rocm_last_error())
    assert parts >= 20, parts                                         # cut at the stored blocks and behind the markers
    assert dst[:n].cpu().numpy().tobytes() == bytes(plain)
