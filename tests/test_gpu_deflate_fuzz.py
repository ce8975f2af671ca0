"""Seeded round-trip stress of both device deflate classes: ragged sizes (0 .. 3 MiB, segment and batch edges +-1),
six kinds of content, every chain level; each stream must inflate to the input with CPython's zlib (an independent
inflater) and with the product's own inflate pipeline."""
import importlib
import random
import zlib

import numpy as np
import pytest

import synth
from gpu_common import product, torch_mod

pytestmark = pytest.mark.gpu


def _content(rnd, n):
    kind = rnd.randrange(6)
    if kind == 0:
        return np.frombuffer(rnd.randbytes(n), dtype=np.uint8)
    if kind == 1:
        return np.zeros(n, dtype=np.uint8) + rnd.randrange(256)
    if kind == 2:
        period = rnd.choice([1, 2, 3, 7, 63, 64, 65, 255, 256, 257, 1000, 32768, 40000])
        base = np.frombuffer(rnd.randbytes(period), dtype=np.uint8)
        return np.resize(base, n)
    if kind == 3:
        return synth.silesia_like(max(n, 16), seed=rnd.randrange(1 << 30), seg_bytes=64 << 10)[:n]
    if kind == 4:
        words = [rnd.randbytes(rnd.randrange(2, 9)) for _ in range(50)]
        out = bytearray()
        while len(out) < n:
            out += rnd.choice(words) + b" "
        return np.frombuffer(bytes(out[:n]), dtype=np.uint8)
    a = np.frombuffer(rnd.randbytes(n), dtype=np.uint8) & 3
    return a.astype(np.uint8)


def test_round_trips():
    zr = product()
    zr.init()
    dfl = importlib.import_module("zlib-ng_amd.deflate")
    inf = importlib.import_module("zlib-ng_amd.inflate")
    torch = torch_mod()
    rnd = random.Random(0xD5F1A7E)
    edges = [0, 1, 2, 3, 4, 5, 255, 256, 257, 258, 259, 1023, 1024, 1025, 4095, 4096, 4097, 32767, 32768, 32769,
             65535, 65536, (128 << 10) - 1, 128 << 10, (128 << 10) + 1, (512 << 10) + 1]
    sizes = edges + [rnd.randrange(0, 3 << 20) for _ in range(30)]
    # level-6 class, one stream each
    for n in sizes:
        data = _content(rnd, n) if n else np.zeros(0, dtype=np.uint8)
        level = rnd.choice([2, 3, 4, 5, 6, 7, 8, 9])
        src = torch.from_numpy(np.concatenate([data, np.zeros(16, dtype=np.uint8)])).cuda()
        dst, clen = dfl.deflate_dev(src, level=level, length=n)
        comp = dst[:clen].cpu().numpy().tobytes()
        d = zlib.decompressobj(-15)
        assert d.decompress(comp) == data.tobytes() and d.eof and d.unused_data == b"", (n, level)
        if n <= (1 << 20):
            dec = inf.decode_tokens(comp)
            assert dec.status == 1 and inf.resolve_dev(dec).cpu().numpy().tobytes() == data.tobytes(), (n, level)
    # level-1 class: all streams of the list in one batch
    datas = [(_content(rnd, n) if n else np.zeros(0, dtype=np.uint8)) for n in sizes]
    offs, flat, at = [], [], 0
    for a in datas:
        offs.append(at)
        pad = (-a.size) % 16 + 16
        flat += [a, np.zeros(pad, dtype=np.uint8)]
        at += a.size + pad
    src = torch.from_numpy(np.concatenate(flat)).cuda()
    qb = dfl.QuickBatch(src, offs, [a.size for a in datas])
    qb.run()
    torch.cuda.synchronize()
    res = qb.results.cpu().numpy()
    for i, a in enumerate(datas):
        comp = qb.compressed(i, res)
        d = zlib.decompressobj(-15)
        assert d.decompress(comp) == a.tobytes() and d.eof, (i, a.size)
        assert (int(res[i, 1]) & 0xffffffff) == zlib.adler32(a.tobytes()), i
