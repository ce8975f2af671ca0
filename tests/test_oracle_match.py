"""Pin the oracle's deflate-side / inflate-side primitives.

The reference has direct tests only for compare256 (test/test_compare256.cc:25-51, a property test re-run
here).  slide_hash / insert_string / longest_match / chunkmemset_safe have NO reference vectors (SURVEY.md
section 4): they are checked against the values SURVEY.md 9.1 recorded from the real reference and against
independent pure-Python models of the documented semantics -- "parity unpinned" beyond that.
"""
import ctypes as C

import numpy as np
import pytest

from deflate_state_util import HostState, W_SIZE, texty


def test_compare256_property(oracle):
    # test_compare256.cc:25-51: strings of 'a', one differing byte at every index 0..256
    for i in range(257):
        a = np.full(257 + 8, ord("a"), dtype=np.uint8)
        b = a.copy()
        if i < 257:
            b[i] = ord("b") if i < 256 else ord("a")
        want = min(i, 256) if i < 256 else 256
        assert oracle.oracle_compare256(a.ctypes.data, b.ctypes.data) == want


def test_recorded_reference_values(oracle):
    assert oracle.oracle_update_hash(0, 0x64636261) == 25357
    assert oracle.oracle_update_hash(12345, 0x64636261) == 25357    # h is unused by the multiplicative hash
    buf = np.zeros(64, dtype=np.uint8)
    buf[:3] = np.frombuffer(b"abc", dtype=np.uint8)
    end = oracle.oracle_chunkmemset_safe(buf.ctypes.data + 3, buf.ctypes.data, 10, 40)
    assert end == buf.ctypes.data + 13
    assert buf[:13].tobytes() == b"abcabcabcabca"


def test_slide_hash_definition(oracle):
    rng = np.random.default_rng(1)
    for w_size in (256, 1024, 4096, 32768):
        hs = HostState(np.zeros(16, dtype=np.uint8), w_size=w_size)
        hs.head[:] = rng.integers(0, 65536, size=65536, dtype=np.uint16)
        hs.prev[:] = rng.integers(0, 65536, size=w_size, dtype=np.uint16)
        h0, p0 = hs.head.astype(np.int64), hs.prev.astype(np.int64)
        oracle.oracle_slide_hash(hs.ref())
        assert (hs.head == np.where(h0 >= w_size, h0 - w_size, 0)).all()
        assert (hs.prev == np.where(p0 >= w_size, p0 - w_size, 0)).all()


def _py_insert(window, head, prev, w_mask, start, count):
    for pos in range(start, start + count):
        val = int.from_bytes(window[pos:pos + 4].tobytes(), "little")
        h = ((val * 2654435761) & 0xffffffff) >> 16
        old = int(head[h])
        idx = pos & 0xffff
        if old != idx:
            prev[idx & w_mask] = old
            head[h] = idx


def test_insert_string_model(oracle):
    data = texty(50000, 3)
    hs = HostState(data)
    head2, prev2 = hs.head.copy(), hs.prev.copy()
    for (start, count) in ((0, 1), (1, 63), (64, 64), (128, 65), (193, 1000), (1193, 0), (1193, 30000)):
        oracle.oracle_insert_string(hs.ref(), start, count)
        _py_insert(hs.window, head2, prev2, W_SIZE - 1, start, count)
        assert (hs.head == head2).all() and (hs.prev == prev2).all()
    got = oracle.oracle_quick_insert_string(hs.ref(), 40000)
    val = int.from_bytes(hs.window[40000:40004].tobytes(), "little")
    assert got == head2[((val * 2654435761) & 0xffffffff) >> 16]


def _py_longest(hs, strstart, cur, prev_length, level):
    """independent model of match_tpl.h for level >= 5: first strictly longer candidate wins, chain order"""
    st = hs.st
    win = hs.window
    best = prev_length if prev_length else 2
    chain = st.max_chain_length >> 2 if best >= st.good_match else st.max_chain_length
    max_dist = st.w_size - 262
    limit = strstart - max_dist if strstart > max_dist else 0
    ms = None
    while cur < strstart:
        n = 0
        while n < 258 and win[cur + n] == win[strstart + n]:
            n += 1
        if n > best and n >= 2 + 0:
            # the reference only evaluates candidates whose first two bytes match; n > best >= 2 implies that
            ms = cur
            if n > st.lookahead:
                return st.lookahead, ms
            best = n
            if best >= st.nice_match:
                return best, ms
        chain -= 1
        if chain == 0:
            break
        cur = int(hs.prev[cur & st.w_mask])
        if cur <= limit:
            break
    return best, ms


@pytest.mark.parametrize("level", [5, 6, 9])
def test_longest_match_model(oracle, level):
    data = texty(60000, 11, alphabet=6, words=60)
    hs = HostState(data)
    hs.set_level(level)
    rng = np.random.default_rng(level)
    pos = 0
    checked = 0
    for strstart in sorted(rng.integers(10, 58000, size=300).tolist()):
        oracle.oracle_insert_string(hs.ref(), pos, strstart - pos)      # chains for everything before strstart
        pos = strstart
        val = int.from_bytes(hs.window[strstart:strstart + 4].tobytes(), "little")
        cur = int(hs.head[((val * 2654435761) & 0xffffffff) >> 16])
        max_dist = W_SIZE - 262
        if cur == 0 or strstart - cur > max_dist:
            continue
        for prev_length in (0, 3, 9):
            hs.st.strstart = strstart
            hs.st.lookahead = min(len(data) - strstart, 400) if checked % 7 else 5
            hs.st.prev_length = prev_length
            hs.st.match_start = 0xABCDEF
            got = oracle.oracle_longest_match(hs.ref(), cur)
            want, ms = _py_longest(hs, strstart, cur, prev_length, level)
            assert got == want, (strstart, cur, prev_length)
            if ms is not None:
                assert hs.st.match_start == ms
            else:
                assert hs.st.match_start == 0xABCDEF
            checked += 1
    assert checked > 100


def test_chunkmemset_safe_is_lz77_copy(oracle):
    rng = np.random.default_rng(5)
    for dist in list(range(1, 40)) + [255, 256, 257, 1000]:
        for length in (1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 64, 100, 258):
            for left in (length, length + 50, max(1, length - 3)):
                buf = rng.integers(0, 256, size=dist + 400, dtype=np.uint8)
                model = buf.copy()
                n = min(length, left)
                for i in range(n):
                    model[dist + i] = model[i]
                end = oracle.oracle_chunkmemset_safe(buf.ctypes.data + dist, buf.ctypes.data, length, left)
                assert end == buf.ctypes.data + dist + n
                assert (buf == model).all()
    assert oracle.oracle_chunksize() == 8


def test_rolling_hash_variants(oracle):
    """insert_string_roll.c (level 9).  The reference holds no vectors for it -> "parity unpinned"; checked against
    an independent Python model of the template and against the closed form the device kernel uses (after three
    bytes the key holds exactly those three: (b0 << 10 ^ b1 << 5 ^ b2) & 32767)."""
    assert oracle.oracle_update_hash_roll(0, 0x61) == 0x61
    assert oracle.oracle_update_hash_roll(0x7fff, 0x1ff) == ((0x7fff << 5) ^ 0xff) & 32767   # only the low byte counts
    data = texty(50000, 11)
    hs = HostState(data)
    head2, prev2 = hs.head.copy(), hs.prev.copy()
    ins_h = 0x1234
    hs.st.ins_h = ins_h
    for (start, count) in ((0, 1), (1, 1), (2, 63), (65, 64), (129, 1000), (1129, 0), (1129, 30000)):
        oracle.oracle_insert_string_roll(hs.ref(), start, count)
        for pos in range(start, start + count):
            ins_h = ((ins_h << 5) ^ int(hs.window[pos + 2])) & 32767
            if pos >= 2:
                b = hs.window[pos:pos + 3].astype(np.uint32)
                assert ins_h == ((int(b[0]) << 10) ^ (int(b[1]) << 5) ^ int(b[2])) & 32767
            idx = pos & 0xffff
            old = int(head2[ins_h])
            if old != idx:
                prev2[idx & (W_SIZE - 1)] = old
                head2[ins_h] = idx
        assert hs.st.ins_h == ins_h
        assert (hs.head == head2).all() and (hs.prev == prev2).all()
    got = oracle.oracle_quick_insert_string_roll(hs.ref(), 40000)
    want_key = ((ins_h << 5) ^ int(hs.window[40002])) & 32767
    assert hs.st.ins_h == want_key and got == head2[want_key]
    assert hs.head[want_key] == 40000 and hs.prev[40000 & (W_SIZE - 1)] == got
