"""tests/test_gpu_inflate_dev.py's mutated-stream corpus ONCE through the diagnostic library whose table builder checks
every stream-derived index (tools/micro/inflate_bounds.sh); prints the number of violations (expected: 0)."""
import ctypes as C, importlib, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
r = importlib.import_module("zlib-ng_amd.rocm")
r._LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libzng_rocm_bounds.so")
import pytest
rc = pytest.main(["-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_inflate_dev.py"), "-k", "mutated or infcover or fixtures"])
L = r.lib(); L.zng_rocm_debug_inflate_bounds.restype = C.c_uint
print("pytest rc", int(rc), "index violations", L.zng_rocm_debug_inflate_bounds())
