"""zlib-ng hot path on MI355X (`arch/rocm`): Python host mirror over the C ABI.

The product is `lib/libzng_rocm.so` (hand-written HIP for gfx950, built from
`csrc/` by `csrc/Makefile`; declared in `include/zng_rocm.h`).  This package is
only the thin host layer tests and bench use to reach it: ctypes bindings with
the reference's names and argument meaning (`adler32`, `crc32`,
`adler32_combine`, ... -- zlib-ng.h.in / functable.h:26-42), plus torch for
device memory, streams and `torch.distributed`.

There is no CPU fallback here: every call goes through the C ABI, and importing
`rocm` raises if the shared library is missing.

The directory name has a hyphen (it is the project's name), so import it with
`importlib.import_module("zlib-ng_amd")`.
"""
from . import rocm  # noqa: F401
from .rocm import (  # noqa: F401
    ZngRocmError, lib, lib_path, build, init, available, device_count,
    adler32, crc32, adler32_z, crc32_z, adler32_fold_copy,
    adler32_combine, crc32_combine, crc32_combine_gen, crc32_combine_op,
    adler32_dev, crc32_dev, adler32_crc32_dev, fold_copy_dev, checksums_dev,
    adler32_combine_dev, crc32_combine_dev, combine_rows_dev, reserve_cus, Crc32Fold, trace_begin, trace_end,
)
