"""GPU: the stream-level API a real DEFLATE_HOOK / pigz-style driver needs.

  * blocks of ONE input compressed independently -- each primed with the 32 KiB before it (dictionary,
    deflate.c:456-531), BFINAL off, ended by the Z_SYNC_FLUSH marker (deflate.c:1064-1076) -- concatenate into ONE
    valid raw deflate stream: pigz's scheme (test/pigz/CMakeLists.txt is the reference's only many-block test).  The
    concatenation is inflated as a single stream by CPython's zlib, by the oracle and by the product itself.
  * inflate of a stream that continues history: a preset dictionary, and the second block of such a stream given the
    last 32 KiB of the first block's plaintext as its window (inflate.c:325-378, :1214-1261)."""
import importlib
import zlib

import numpy as np
import pytest

import inflate_util
import synth
from gpu_common import product, torch_mod

pytestmark = pytest.mark.gpu
BLOCK = 128 << 10


@pytest.fixture(scope="module")
def mods():
    zr = product()
    zr.init()
    return zr, importlib.import_module("zlib-ng_amd.deflate"), importlib.import_module("zlib-ng_amd.inflate")


@pytest.fixture(scope="module")
def corpus():
    return synth.silesia_like(16 << 20, seed=0xB10C, seg_bytes=1 << 20)


def _check_one_stream(comp, data, inf):
    d = zlib.decompressobj(-15)
    assert d.decompress(comp) == data and d.eof and d.unused_data == b""
    st, msg, out, used = inflate_util.oracle_inflate(comp, cap=len(data) + 64)
    assert (st, used) == (1, len(comp)) and out == data, (st, msg)
    torch = torch_mod()
    dst = torch.empty(len(data) + 64, dtype=torch.uint8, device="cuda")
    rc, produced = inf.inflate_raw(comp, dst)
    assert rc == 1 and produced == len(data) and dst[:produced].cpu().numpy().tobytes() == data


def test_pigz_blocks_level1_class(mods, corpus):
    zr, dfl, inf = mods
    torch = torch_mod()
    n = corpus.size
    nblk = n // BLOCK
    src = torch.from_numpy(corpus).cuda()
    offs = [i * BLOCK for i in range(nblk)]
    dicts = [min(32768, o) for o in offs]
    flags = [dfl.BLOCK_NOT_FINAL | dfl.BLOCK_SYNC_FLUSH] * (nblk - 1) + [0]
    batch = dfl.QuickBatch(src, offs, [BLOCK] * nblk, dict_len=dicts, flags=flags)
    batch.run()
    torch.cuda.synchronize()
    res = batch.results.cpu()
    parts = [batch.compressed(i, res) for i in range(nblk)]
    raw = corpus.tobytes()
    for i in (0, 1, nblk - 1):                                  # the per-block rows: Adler-32 of the block alone
        assert (int(res[i, 1]) & 0xffffffff) == zlib.adler32(raw[offs[i]:offs[i] + BLOCK])
    for i in range(nblk - 1):
        assert parts[i].endswith(b"\x00\x00\xff\xff") and (parts[i][0] & 7) == 2       # marker; BFINAL 0, static
    assert (parts[-1][0] & 7) == 3
    _check_one_stream(b"".join(parts), raw, inf)
    # the dictionary is worth something: the same blocks without priming come out larger
    plain = dfl.QuickBatch(src, offs, [BLOCK] * nblk, flags=flags)
    plain.run()
    torch.cuda.synchronize()
    assert int(plain.results.cpu()[:, 0].sum()) > int(res[:, 0].sum())
    _check_one_stream(b"".join(plain.compressed(i) for i in range(nblk)), raw, inf)   # Z_FULL_FLUSH-style blocks
    # a dictionary that is not a multiple of anything, on a stream that does not start aligned
    odd = dfl.QuickBatch(src, [70001], [300007], dict_len=[12345], flags=[0])
    odd.run()
    torch.cuda.synchronize()
    comp = odd.compressed(0)
    st, msg, out, used = inflate_util.oracle_inflate_dict(comp, raw[70001 - 12345:70001], cap=300007 + 16)
    assert st == 1 and out == raw[70001:70001 + 300007]


@pytest.mark.parametrize("level", [0, 1, 6])
def test_pigz_blocks_chain_class(mods, corpus, level):
    zr, dfl, inf = mods
    torch = torch_mod()
    raw = corpus[:4 << 20].tobytes()
    src = torch.from_numpy(corpus[:4 << 20].copy()).cuda()
    big = 1 << 20                                               # 1 MiB blocks: several segments each
    parts = []
    for i, off in enumerate(range(0, len(raw), big)):
        last = off + big >= len(raw)
        dst, clen = dfl.deflate_dev(src, level=level, length=big, offset=off, dict_len=min(32768, off),
                                    flags=0 if last else dfl.BLOCK_NOT_FINAL | dfl.BLOCK_SYNC_FLUSH)
        part = dst[:clen].cpu().numpy().tobytes()
        if not last:
            assert part.endswith(b"\x00\x00\xff\xff")
        parts.append(part)
    _check_one_stream(b"".join(parts), raw, inf)


def test_inflate_continues_a_window(mods, corpus):
    zr, dfl, inf = mods
    torch = torch_mod()
    raw = corpus[:2 << 20].tobytes()
    # (1) preset dictionary, stream made by CPython (zdict)
    dictionary, data = raw[:32768], raw[100000:900000]
    c = zlib.compressobj(6, zlib.DEFLATED, -15, zdict=dictionary)
    comp = c.compress(data) + c.flush()
    d_win = torch.from_numpy(np.frombuffer(dictionary, dtype=np.uint8).copy()).cuda()
    dst = torch.empty(len(data) + 64, dtype=torch.uint8, device="cuda")
    rc, produced, used = inf.inflate_raw_window(comp, d_win, dst)
    assert (rc, produced, used) == (1, len(data), len(comp))
    assert dst[:produced].cpu().numpy().tobytes() == data
    st, msg, out, _ = inflate_util.oracle_inflate_dict(comp, dictionary, cap=len(data) + 16)
    assert st == 1 and out == data
    rc, _, _ = inf.inflate_raw_window(comp, None, dst)          # without it: the reference's message
    assert rc == -3 and b"invalid distance too far back" in zr.lib().zng_rocm_last_error()
    # a short window, and a single-segment stream (< 32 KiB of output) that reaches into it
    c = zlib.compressobj(9, zlib.DEFLATED, -15, zdict=raw[5000:6000])
    comp = c.compress(raw[5000:5900] * 3) + c.flush()
    rc, produced, used = inf.inflate_raw_window(comp, d_win[5000:6000].contiguous(), dst)
    assert rc == 1 and dst[:produced].cpu().numpy().tobytes() == raw[5000:5900] * 3
    # (2) the second block of a two-block stream, resumed with the first block's last 32 KiB as its window
    src = torch.from_numpy(corpus[:2 << 20].copy()).cuda()
    half = 1 << 20
    b2, clen2 = dfl.deflate_dev(src, level=6, length=half, offset=half, dict_len=32768, flags=0)
    comp2 = b2[:clen2].cpu().numpy().tobytes()
    small = dst
    dst = torch.empty(half + 64, dtype=torch.uint8, device="cuda")
    assert inf.inflate_raw_window(comp2, src[half - 32768:half].contiguous(), small)[0] == -5    # Z_BUF_ERROR: too small
    rc, produced, used = inf.inflate_raw_window(comp2, src[half - 32768:half].contiguous(), dst)
    assert (rc, produced, used) == (1, half, len(comp2))
    assert torch.equal(dst[:half], src[half:])
