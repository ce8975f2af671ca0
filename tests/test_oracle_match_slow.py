"""oracle_longest_match_slow (match_tpl.h with LONGEST_MATCH_SLOW): the reference has no vectors for it
 -> "parity unpinned"; checked here for the contract of match_tpl.h:16-24: the returned
length never exceeds lookahead, and whenever it beats prev_length the bytes at match_start really match."""
import numpy as np
import pytest

from deflate_state_util import HostState, W_SIZE, texty


@pytest.mark.parametrize("level", [7, 8, 9])
def test_contract(oracle, level):
    # 32000 positions: no two share a prev[] slot, so every chain entry is a true predecessor whatever strstart is
    # (positions at or after strstart are in the table too; the template ignores heads that are not below cur_match)
    data = texty(32000, 300 + level, alphabet=5, words=40)
    hs = HostState(data)
    hs.set_level(level)
    if level == 9:                                  # lm_init binds the rolling hash (deflate.c:1223-1234)
        oracle.oracle_insert_string_roll(hs.ref(), 0, 31700)
    else:
        oracle.oracle_insert_string(hs.ref(), 0, 31700)
    rng = np.random.default_rng(level)
    better_than_fast = checked = 0
    for strstart in rng.integers(300, 31000, size=900).tolist():
        cur = int(hs.prev[strstart & hs.st.w_mask])
        if cur == 0 or cur >= strstart or strstart - cur > W_SIZE - 262:
            continue
        for prev_length, lookahead in ((0, 400), (3, 400), (5, 400), (20, 400), (4, 7)):
            hs.st.strstart, hs.st.prev_length, hs.st.lookahead = strstart, prev_length, lookahead
            hs.st.match_start = 0xABCDEF
            got = oracle.oracle_longest_match_slow(hs.ref(), cur)
            ms = hs.st.match_start
            assert got <= max(lookahead, prev_length if prev_length else 2)
            if got > (prev_length if prev_length else 2):
                assert ms != 0xABCDEF and ms < strstart
                n = min(got, lookahead)
                assert (hs.window[ms:ms + n] == hs.window[strstart:strstart + n]).all(), (strstart, prev_length)
            hs.st.match_start = 0xABCDEF
            fast = oracle.oracle_longest_match(hs.ref(), cur)
            better_than_fast += got > fast
            checked += 1
    assert checked > 1000
