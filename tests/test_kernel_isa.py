"""Build check (no GPU): the stream-level kernels must compile without scratch memory and without out-of-line calls.
Round 2 found a build of inflate_streams_kernel in which one lambda was called out of line: its by-reference captures
(the whole bit-parse state) went through the stack, and that build faulted on the device.  hipcc cross-compiles here."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zlib-ng_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("source", ["inflate_dev.hip", "deflate_stream.hip"])
def test_no_scratch_no_calls(source):
    tmp = tempfile.mkdtemp(prefix="zng_isa_")
    try:
        out = os.path.join(tmp, "k.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-S",
                        "--cuda-device-only", "-o", out, os.path.join(CSRC, source)], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open(out).read()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    assert "s_swappc_b64" not in text, "a device function or lambda is called out of line"
    sizes = [int(v) for v in re.findall(r"\.private_segment_fixed_size:\s*(\d+)", text)]
    spills = [int(v) for v in re.findall(r"\.vgpr_spill_count:\s*(\d+)", text)]
    assert sizes and all(v == 0 for v in sizes), sizes
    assert all(v == 0 for v in spills), spills
