"""The COARSE boundary as code: integration/arch/rocm/rocm_deflate.{h,c} and rocm_inflate.{h,c} -- the
DEFLATE_HOOK / INFLATE_TYPEDO_HOOK backend a maintainer drops into the reference tree (deflate.c:72-106, :1039-1083;
inflate_p.h:11-41, inflate.c:728; template arch/s390/dfltcc_*.h) -- compiled as strict C11 against a test-only restatement
of the declarations they take from the reference (tests/c/zlibng_coarse_min.h), linked against libzng_rocm.so and driven by
tests/c/coarse_driver.c, which restates deflate()'s / inflate()'s control flow around the hook macros.

Without a GPU the hooks must answer "not ours" before a byte is consumed (the software path continues: "fallback").  With
one: several deflate(Z_NO_FLUSH) calls with small avail_out, Z_SYNC_FLUSH in between, Z_FINISH; the stream is read back by
CPython's zlib (classic zlib, checks the Adler-32 trailer the hook maintained) AND by the oracle inflater; inflate with
avail_in trickled and small avail_out returns the plaintext; a damaged stream returns the reference's message."""
import importlib
import os
import subprocess
import zlib

import numpy as np
import pytest

import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    zr = importlib.import_module("zlib-ng_amd")
    libdir = os.path.dirname(zr.lib_path())
    exe = str(tmp_path_factory.mktemp("coarse") / "coarse_driver")
    arch = os.path.join(ROOT, "integration", "arch", "rocm")
    cmd = ["gcc", "-std=c11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O2", "-DZNG_ROCM_STANDALONE_CHECK",
           "-DROCM_MIN_BYTES=1024", "-DROCM_INFLATE_MIN_BYTES=1", "-DROCM_DEFLATE_BLOCK_BYTES=1048576",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "c"), "-I" + arch,
           os.path.join(ROOT, "tests", "c", "coarse_driver.c")] + \
          [os.path.join(arch, f) for f in ("rocm_deflate.c", "rocm_inflate.c", "rocm_slots.c", "rocm_features.c")] + \
          ["-o", exe, "-L" + libdir, "-lzng_rocm", "-Wl,-rpath," + libdir]
    subprocess.check_call(cmd)
    return exe


def _run(exe, *args, env=None):
    p = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **env) if env else None)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    return p.stdout.strip()


def test_hooks_fall_back_without_a_gpu(driver, tmp_path):
    zr = importlib.import_module("zlib-ng_amd")
    if zr.device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu-marked tests")
    plain = synth.silesia_like(1 << 20, seed=7).tobytes()
    (tmp_path / "in.bin").write_bytes(plain)
    assert _run(driver, "d", 6, 1, 100000, 4096, 3, tmp_path / "in.bin", tmp_path / "out.z") == "fallback"
    (tmp_path / "in.z").write_bytes(zlib.compress(plain, 6))
    assert _run(driver, "i", 1, 5000, 8192, tmp_path / "in.z", tmp_path / "out.bin", len(plain)) == "fallback"


@pytest.mark.gpu
@pytest.mark.parametrize("level,wrap,in_chunk,out_chunk,sync_every", [
    (6, 1, 100000, 4096, 7),          # zlib wrapper, many small calls, a sync flush every 7th, tiny avail_out
    (1, 0, 1 << 20, 1 << 16, 0),      # raw, level 1, no flushes
    (9, 1, 300000, 1000, 2),
    (0, 0, 65536, 65536, 5),          # deflate_stored through the hook
    (6, 1, 5 << 20, 1 << 20, 0),      # calls larger than the gather block
])
def test_deflate_hook_stream_round_trips(driver, tmp_path, level, wrap, in_chunk, out_chunk, sync_every):
    import inflate_util
    plain = synth.silesia_like(6 << 20, seed=11 + level).tobytes()
    (tmp_path / "in.bin").write_bytes(plain)
    line = _run(driver, "d", level, wrap, in_chunk, out_chunk, sync_every, tmp_path / "in.bin", tmp_path / "out.z")
    assert line.startswith("device %d " % len(plain)), line
    comp = (tmp_path / "out.z").read_bytes()
    assert len(comp) == int(line.split()[2])
    if wrap:
        assert zlib.decompress(comp) == plain                          # header, Adler-32 trailer and all
        raw = comp[2:-4]
    else:
        assert zlib.decompressobj(-15).decompress(comp) == plain
        raw = comp
    st, msg, data, used = inflate_util.oracle_inflate(raw, cap=len(plain))
    assert (st, data) == (1, plain) and used == len(raw), (st, msg)
    if level:
        assert len(comp) < len(plain) * 0.6


@pytest.mark.gpu
@pytest.mark.parametrize("wrap,in_chunk,out_chunk", [(1, 50000, 8192), (0, 1 << 22, 1 << 20), (1, 1 << 24, 100000)])
def test_inflate_hook_trickled_input(driver, tmp_path, wrap, in_chunk, out_chunk):
    plain = synth.silesia_like(4 << 20, seed=5).tobytes()
    c = zlib.compressobj(6, zlib.DEFLATED, 15 if wrap else -15)
    comp = c.compress(plain) + c.flush()
    (tmp_path / "in.z").write_bytes(comp)
    line = _run(driver, "i", wrap, in_chunk, out_chunk, tmp_path / "in.z", tmp_path / "out.bin", len(plain))
    assert line == "device %d %d" % (len(comp), len(plain)), line
    assert (tmp_path / "out.bin").read_bytes() == plain


@pytest.mark.gpu
def test_inflate_hook_decodes_a_large_member_on_the_device(driver, tmp_path):
    """a member of more than 4 MiB of compressed bytes goes through inflate_large.hip (block starts found on the device,
    one wavefront per part), with the history and the check value kept by the hook as before; a damaged one falls to the
    sequential decoder and reports what the reference would"""
    plain = synth.silesia_like(48 << 20, seed=21).tobytes()
    comp = zlib.compress(plain, 6)
    assert len(comp) > (8 << 20)
    (tmp_path / "big.z").write_bytes(comp)
    out = _run(driver, "i", 1, 1 << 26, 1 << 22, tmp_path / "big.z", tmp_path / "big.bin", len(plain), env={"COARSE_DRIVER_PARTS": "1"})
    lines = out.splitlines()
    assert lines[0] == "device %d %d" % (len(comp), len(plain)), lines
    assert int(lines[1].split()[1]) >= 64, lines                       # really decoded in parts on the device
    assert (tmp_path / "big.bin").read_bytes() == plain
    bad = bytearray(comp)
    bad[len(bad) // 2] ^= 0x10
    try:
        zlib.decompress(bytes(bad))
        expected = None
    except zlib.error as e:
        expected = str(e)
    (tmp_path / "bad.z").write_bytes(bytes(bad))
    got = _run(driver, "i", 1, 1 << 26, 1 << 22, tmp_path / "bad.z", tmp_path / "o", len(plain))
    assert got.startswith("data error: "), got
    if expected:                                                       # classic zlib words the same checks the same way
        assert got[len("data error: "):] in expected, (got, expected)


@pytest.mark.gpu
def test_inflate_hook_reports_the_references_messages(driver, tmp_path):
    plain = synth.silesia_like(1 << 20, seed=6).tobytes()
    comp = bytearray(zlib.compress(plain, 6))
    comp[-2] ^= 0x55                                                    # trailer
    (tmp_path / "bad1.z").write_bytes(bytes(comp))
    assert _run(driver, "i", 1, 1 << 24, 1 << 20, tmp_path / "bad1.z", tmp_path / "o", len(plain)) == "data error: incorrect data check"
    (tmp_path / "bad2.z").write_bytes(b"\x78\x9c" + bytes([0x06]) + b"\0" * 64)      # block type 3, infcover's "invalid block type"
    assert _run(driver, "i", 1, 1 << 24, 1 << 20, tmp_path / "bad2.z", tmp_path / "o", 1024) == "data error: invalid block type"
