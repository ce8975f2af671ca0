"""CPU checks of the C-ABI boundary: the library loads without a GPU and exports every
symbol include/zng_rocm.h declares; host-side scalar logic agrees with the oracle."""
import importlib
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _product():
    return importlib.import_module("zlib-ng_amd")


def test_header_symbols_are_exported():
    zr = _product()
    hdr = open(os.path.join(ROOT, "include", "zng_rocm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    declared = set(re.findall(r"\b(zng_rocm_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    handle = zr.lib()
    for name in sorted(declared):
        assert hasattr(handle, name), "declared in the header but not exported: " + name
    assert declared == set(zr.rocm.exported_names()), declared ^ set(zr.rocm.exported_names())


def test_loads_and_reports_without_gpu():
    zr = _product()
    n = zr.device_count()
    assert n >= 0
    if n == 0:
        assert not zr.available()
        try:
            zr.init()
        except zr.ZngRocmError as e:
            assert "no HIP device" in str(e)
        else:
            raise AssertionError("init must fail loudly without a device")


def test_host_combine_matches_oracle(oracle):
    zr = _product()
    rng = np.random.default_rng(0)
    for _ in range(200):
        a1, a2, c1, c2 = (int(v) for v in rng.integers(0, 2**32, size=4, dtype=np.uint64))
        n = int(rng.integers(0, 2**40))
        assert zr.adler32_combine(a1, a2, n) == oracle.oracle_adler32_combine(a1, a2, n)
        assert zr.crc32_combine(c1, c2, n) == oracle.oracle_crc32_combine(c1, c2, n)
        op = zr.crc32_combine_gen(n)
        assert op == oracle.oracle_crc32_combine_gen(n)
        assert zr.crc32_combine_op(c1, c2, op) == oracle.oracle_crc32_combine_op(c1, c2, op)
    assert zr.adler32_combine(1, 1, -5) == 0xffffffff
