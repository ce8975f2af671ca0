"""GPU parity: longest_match_slow_dev vs the oracle, bit-exact (length and match_start)."""
import ctypes as C

import numpy as np
import pytest

from deflate_state_util import HostState, W_SIZE, texty
from gpu_common import product, torch_mod
from test_gpu_deflate_prims import DevState

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("level", [7, 8, 9])
def test_longest_match_slow(oracle, level):
    zr = product()
    zr.init()
    torch = torch_mod()
    # 32000 positions: no two share a prev[] slot (see test_oracle_match_slow.py)
    data = texty(32000, 500 + level, alphabet=5, words=50)
    hs = HostState(data)
    hs.set_level(level)
    if level == 9:                                  # lm_init binds the rolling hash (deflate.c:1223-1234)
        oracle.oracle_insert_string_roll(hs.ref(), 0, 31700)
    else:
        oracle.oracle_insert_string(hs.ref(), 0, 31700)
    dev = DevState(zr, hs)
    rng = np.random.default_rng(level)
    views, curs, wants = [], [], []
    for strstart in rng.integers(300, 31000, size=700).tolist():
        cur = int(hs.prev[strstart & hs.st.w_mask])
        if cur == 0 or cur >= strstart or strstart - cur > W_SIZE - 262:
            continue
        for prev_length, lookahead in ((0, 400), (3, 400), (5, 400), (12, 400), (30, 400), (4, 7), (6, 262)):
            hs.st.strstart, hs.st.prev_length, hs.st.lookahead = strstart, prev_length, lookahead
            hs.st.match_start = 0x123456
            want_len = oracle.oracle_longest_match_slow(hs.ref(), cur)
            wants.append((want_len, hs.st.match_start))
            v = zr.rocm.DeflateView()
            C.memmove(C.byref(v), C.byref(dev.view), C.sizeof(v))
            v.strstart, v.prev_length, v.lookahead, v.match_start = strstart, prev_length, lookahead, 0x123456
            v.max_chain_length, v.good_match, v.nice_match, v.level = (hs.st.max_chain_length, hs.st.good_match,
                                                                       hs.st.nice_match, hs.st.level)
            views.append(v)
            curs.append(cur)
    assert len(views) > 1000
    d_views = zr.rocm.views_to_device(views)
    d_cur = torch.from_numpy(np.array(curs, dtype=np.uint16).view(np.int16)).cuda()
    lens = torch.zeros(len(views), dtype=torch.int32, device="cuda")
    starts = torch.zeros(len(views), dtype=torch.int32, device="cuda")
    zr.rocm._check(zr.rocm.lib().zng_rocm_longest_match_slow_dev(
        zr.rocm._dev_ptr(d_views), len(views), zr.rocm._dev_ptr(d_cur), zr.rocm._dev_ptr(lens),
        zr.rocm._dev_ptr(starts), zr.rocm._stream_ptr(None)), "zng_rocm_longest_match_slow_dev")
    got = list(zip(lens.cpu().tolist(), starts.cpu().tolist()))
    assert got == wants
