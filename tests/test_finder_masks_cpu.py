"""The block finder's range tests (zlib-ng_amd/csrc/inflate_large.hip, find_headers_kernel<false>) are evaluated for the eight
bit positions of a byte at once, as bit masks over a 32-bit window.  This pins the mask algebra against the rules it stands
for -- BTYPE == 2 (inflate.c:748-757), HLIT <= 29 and HDIST <= 29, i.e. nlen <= 286 and ndist <= 30 (inflate.c:808-813) --
for every 20-bit window (position k looks at bits k .. k+12, k < 8).  CPU only; the device code itself is covered by
tests/test_gpu_inflate_large.py::test_nearly_every_block_start_becomes_a_part."""
import numpy as np


def test_masks_agree_with_the_field_tests_for_every_window():
    w = np.arange(1 << 20, dtype=np.uint32)
    hl = (w >> 4) & (w >> 5) & (w >> 6) & (w >> 7)
    hd = (w >> 9) & (w >> 10) & (w >> 11) & (w >> 12)
    left = (~w >> 1) & (w >> 2) & ~hl & ~hd & np.uint32(0xFF)
    for k in range(8):
        h = (w >> k) & 8191
        btype = (h >> 1) & 3
        hlit = (h >> 3) & 31
        hdist = (h >> 8) & 31
        want = (btype == 2) & (hlit <= 29) & (hdist <= 29)
        got = ((left >> k) & 1).astype(bool)
        assert np.array_equal(got, want), k
