"""GPU: every BASELINE.json config at its STATED size.  The smaller-size parity tests live next
to each component; these are the full-size runs, checked bit-exactly where the host can hold the answer and through
size-independent properties otherwise.

  configs[1]  crc32 + adler32 over 1 GiB                      -> tests/test_gpu_checksums.py (1 GiB and 5 GiB)
  configs[2]  raw inflate of a level-6 stream, 256 MiB plain  -> test_cfg3_inflate_256mib
  configs[3]  deflate level 6 of 256 MiB, round trip          -> test_cfg4_deflate_level6_256mib
  configs[4]  4096 x 1 MiB streams, level 1                   -> test_cfg5_4096_streams_one_batch (one GPU's view; the
              sharding over ranks is tests/test_distributed_gloo.py and bench.py --gpus N)"""
import importlib
import zlib

import numpy as np
import pytest

import synth
from gpu_common import product, torch_mod

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    zr = product()
    zr.init()
    return zr, importlib.import_module("zlib-ng_amd.deflate"), importlib.import_module("zlib-ng_amd.inflate")


@pytest.fixture(scope="module")
def plain256():
    """cfg3 / cfg4 plaintext = 256 MiB Silesia-like six-class mix, seed 0x5EED0003, 8 MiB segments"""
    return synth.silesia_like(256 << 20, seed=0x5EED0003)


def test_cfg3_inflate_256mib(mods, plain256):
    """BASELINE.json configs[2]: raw inflate of a pre-built level-6 stream whose plaintext is 256 MiB; the device
    output must equal the plaintext byte for byte.  The stream is built by CPython's zlib (classic zlib, an
    independent encoder), streamed so that the host never holds two copies."""
    zr, dfl, inf = mods
    torch = torch_mod()
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = []
    for lo in range(0, plain256.size, 32 << 20):
        parts.append(c.compress(plain256[lo:lo + (32 << 20)].tobytes()))
    parts.append(c.flush())
    comp = b"".join(parts)
    del parts
    dst = torch.empty(plain256.size + 64, dtype=torch.uint8, device="cuda")
    rc, produced = inf.inflate_raw(comp, dst)
    assert rc == 1 and produced == plain256.size
    assert torch.equal(dst[:produced], torch.from_numpy(plain256).cuda())
    # the same stream with its host decode spread over the host's threads
    dst.zero_()
    rc, produced, used = inf.inflate_raw_threads(comp, dst, nthreads=0)
    assert (rc, produced, used) == (1, plain256.size, len(comp))
    assert zr.lib().zng_rocm_inflate_threads_last_parts() >= 2
    assert torch.equal(dst[:produced], torch.from_numpy(plain256).cuda())
    # and the token route the one-shot uses, with its segment count (>= 32 KiB of output per segment)
    dec = inf.decode_tokens(comp)
    assert dec.status == 1 and dec.in_used == len(comp) and 8000 <= dec.nsegs <= 8192


def test_cfg4_deflate_level6_256mib(mods, plain256):
    """BASELINE.json configs[3]: level 6 on the 256 MiB mix; the stream must inflate back to the identical bytes
    through an independent inflater (CPython zlib, streamed) AND through the product's own inflate; ratio stays in
    the level-6 class (classic zlib -6 reaches 2.65 on this mix)."""
    zr, dfl, inf = mods
    torch = torch_mod()
    src = torch.from_numpy(plain256).cuda()
    dst, clen = dfl.deflate_dev(src, level=6)
    comp = dst[:clen].cpu().numpy()
    assert plain256.size / clen > 2.45, plain256.size / clen
    d = zlib.decompressobj(-15)
    pos = total = 0
    while pos < comp.size:
        piece = d.decompress(comp[pos:pos + (8 << 20)].tobytes())
        assert piece == plain256[total:total + len(piece)].tobytes()
        total += len(piece)
        pos += 8 << 20
    tail = d.flush()
    assert tail == plain256[total:total + len(tail)].tobytes()
    assert d.eof and total + len(tail) == plain256.size
    out = torch.empty(plain256.size + 64, dtype=torch.uint8, device="cuda")
    rc, produced = inf.inflate_raw(comp.tobytes(), out)
    assert rc == 1 and produced == plain256.size and torch.equal(out[:produced], src)


def test_cfg5_4096_streams_one_batch(mods):
    """BASELINE.json configs[4]: 4096 independent 1 MiB streams, level-1 class, in ONE batch.  Stream i is slice
    (i mod 96) of a 96 MiB six-class mix (seed 0x5EED0005) -- the same layout bench.py --workload streams uses -- so
    every one of the 4096 {clen, adler32} rows has a reference: the Adler-32 of its slice (CPython zlib), and the
    compressed length and bytes of the first stream built from the same slice.  A strided sample (every 41st stream)
    plus the last one is inflated by an independent inflater and compared with its slice."""
    zr, dfl, inf = mods
    torch = torch_mod()
    each, nstreams, distinct = 1 << 20, 4096, 96
    base = synth.silesia_like(distinct << 20, seed=0x5EED0005, seg_bytes=1 << 20)
    d_base = torch.from_numpy(base).cuda()
    reps = -(-nstreams // distinct)
    src = d_base.repeat(reps)[:nstreams * each].contiguous()
    del d_base
    batch = dfl.QuickBatch(src, [i * each for i in range(nstreams)], [each] * nstreams)
    batch.run()
    torch.cuda.synchronize()
    res = batch.results.cpu().numpy().astype(np.int64) & 0xffffffff
    raw = [base[k * each:(k + 1) * each].tobytes() for k in range(distinct)]
    adlers = np.array([zlib.adler32(r) for r in raw], dtype=np.int64)
    idx = np.arange(nstreams) % distinct
    assert np.array_equal(res[:, 1], adlers[idx])                       # every Adler-32 row
    assert np.array_equal(res[:, 0], res[idx, 0])                       # equal input -> equal compressed length
    assert np.all(res[:, 0] <= np.array(batch.bounds)) and np.all(res[:, 0] > 0)
    total_out = int(res[:, 0].sum())
    assert 1.7 < nstreams * each / total_out < 2.3                      # reference level 1 on text-like data: 1.91
    for i in list(range(0, nstreams, 41)) + [nstreams - 1]:
        comp = batch.dst[batch.out_off[i]:batch.out_off[i] + int(res[i, 0])].cpu().numpy().tobytes()
        d = zlib.decompressobj(-15)
        assert d.decompress(comp) == raw[i % distinct] and d.eof, i
        first = i % distinct                                            # same slice, first occurrence: same bytes
        if first != i:
            ref = batch.dst[batch.out_off[first]:batch.out_off[first] + int(res[first, 0])]
            assert torch.equal(ref, batch.dst[batch.out_off[i]:batch.out_off[i] + int(res[i, 0])])
