"""GPU parity: deflate-side functable primitives and the inflate match-copy primitive vs the oracle.
Bit-exact; every call goes through the C ABI (zng_rocm_*_dev)."""
import ctypes as C

import numpy as np
import pytest

from deflate_state_util import HostState, PAD, W_SIZE, texty
from gpu_common import product, to_dev, torch_mod

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zr():
    m = product()
    m.init()
    return m


class DevState:
    """device copies of a HostState's slabs + the matching zng_rocm_deflate_view"""

    def __init__(self, zr, hs):
        torch = torch_mod()
        self.window = to_dev(hs.window)
        self.prev = torch.from_numpy(hs.prev.view(np.int16)).cuda()
        self.head = torch.from_numpy(hs.head.view(np.int16)).cuda()
        v = zr.rocm.DeflateView()
        v.window, v.prev, v.head = self.window.data_ptr(), self.prev.data_ptr(), self.head.data_ptr()
        v.w_size, v.w_mask = hs.st.w_size, hs.st.w_mask
        for f in ("lookahead", "strstart", "match_start", "prev_length", "max_chain_length", "good_match",
                  "nice_match", "level"):
            setattr(v, f, getattr(hs.st, f))
        self.view = v

    def head_np(self):
        return self.head.cpu().numpy().view(np.uint16)

    def prev_np(self):
        return self.prev.cpu().numpy().view(np.uint16)


def test_slide_hash(zr, oracle):
    rng = np.random.default_rng(21)
    states, devs = [], []
    for w_size in (256, 512, 4096, 32768, 32768, 32768):
        hs = HostState(np.zeros(16, dtype=np.uint8), w_size=w_size)
        hs.head[:] = rng.integers(0, 65536, size=65536, dtype=np.uint16)
        hs.prev[:] = rng.integers(0, 65536, size=w_size, dtype=np.uint16)
        hs.head[:5] = (0, w_size - 1, w_size, w_size + 1, 65535)
        states.append(hs)
        devs.append(DevState(zr, hs))
    d_views = zr.rocm.views_to_device([d.view for d in devs])
    zr.rocm.slide_hash_dev(d_views, len(devs))
    for hs, d in zip(states, devs):
        oracle.oracle_slide_hash(hs.ref())
        assert (d.head_np() == hs.head).all()
        assert (d.prev_np() == hs.prev).all()


def test_compare256(zr, oracle):
    torch = torch_mod()
    # the reference's property test (test_compare256.cc:25-51) ...
    n = 257
    buf = np.full((2 * n + 2) * 272, ord("a"), dtype=np.uint8)
    off0 = np.arange(n, dtype=np.uint64) * 544
    off1 = off0 + 272
    for i in range(256):
        buf[int(off1[i]) + i] = ord("b")
    # ... plus random unaligned pairs over random data with long common prefixes
    rng = np.random.default_rng(8)
    rnd = rng.integers(0, 4, size=60000, dtype=np.uint8)
    rnd[30000:60000] = rnd[0:30000]
    cut = rng.integers(30000, 59000, size=2000)
    rnd[cut] ^= 1
    base = np.concatenate([buf, rnd])
    r0 = rng.integers(0, 29000, size=500).astype(np.uint64)
    off0 = np.concatenate([off0, r0 + buf.size])
    off1 = np.concatenate([off1, r0 + buf.size + 30000])
    d_base = to_dev(base)
    out = torch.zeros(off0.size, dtype=torch.int32, device="cuda")
    zr.rocm.compare256_dev(d_base, torch.from_numpy(off0.view(np.int64)).cuda(),
                           torch.from_numpy(off1.view(np.int64)).cuda(), out)
    got = out.cpu().numpy()
    for i in range(off0.size):
        want = oracle.oracle_compare256(base.ctypes.data + int(off0[i]), base.ctypes.data + int(off1[i]))
        assert got[i] == want, i
    assert got[:257].tolist() == list(range(256)) + [256]


def test_update_hash(zr, oracle):
    torch = torch_mod()
    rng = np.random.default_rng(4)
    vals = rng.integers(0, 2**32, size=5000, dtype=np.uint64).astype(np.uint32)
    vals[0] = 0x64636261
    out = torch.zeros(vals.size, dtype=torch.int32, device="cuda")
    zr.rocm.update_hash_dev(torch.from_numpy(vals.view(np.int32)).cuda(), out)
    got = out.cpu().numpy().view(np.uint32)
    assert got[0] == 25357
    for i in range(0, vals.size, 7):
        assert got[i] == oracle.oracle_update_hash(0, int(vals[i]))


def test_insert_string_and_quick_insert(zr, oracle):
    torch = torch_mod()
    plans = [(0, 1), (1, 63), (64, 64), (128, 65), (193, 1000), (1193, 0), (1193, 30000), (31193, 2)]
    states = [HostState(texty(60000, s, alphabet=a, words=w)) for s, a, w in ((3, 24, 400), (4, 3, 20), (5, 26, 3000))]
    states.append(HostState(np.zeros(60000, dtype=np.uint8)))       # every position collides
    devs = [DevState(zr, hs) for hs in states]
    d_views = zr.rocm.views_to_device([d.view for d in devs])
    for (start, count) in plans:
        strs = torch.full((len(devs),), start, dtype=torch.int32, device="cuda")
        cnts = torch.full((len(devs),), count, dtype=torch.int32, device="cuda")
        zr.rocm.insert_string_dev(d_views, len(devs), strs, cnts)
        for hs, d in zip(states, devs):
            oracle.oracle_insert_string(hs.ref(), start, count)
            assert (d.head_np() == hs.head).all(), (start, count)
            assert (d.prev_np() == hs.prev).all(), (start, count)
    for pos in (40000, 40001, 40000, 12):           # re-inserting the same position is a no-op
        strs = torch.full((len(devs),), pos, dtype=torch.int32, device="cuda")
        heads = torch.zeros(len(devs), dtype=torch.int16, device="cuda")
        zr.rocm.quick_insert_string_dev(d_views, len(devs), strs, heads)
        got = heads.cpu().numpy().view(np.uint16)
        for i, (hs, d) in enumerate(zip(states, devs)):
            assert got[i] == oracle.oracle_quick_insert_string(hs.ref(), pos)
            assert (d.head_np() == hs.head).all() and (d.prev_np() == hs.prev).all()


def test_rolling_hash_insert_family(zr, oracle):
    """insert_string_roll.c (level 9): update_hash_roll, insert_string_roll, quick_insert_string_roll with the
    running key s->ins_h carried per stream, bit-exact against the oracle (head, prev and the key)."""
    torch = torch_mod()
    rng = np.random.default_rng(21)
    hv = rng.integers(0, 2**32, size=4000, dtype=np.uint64).astype(np.uint32)
    vv = rng.integers(0, 2**32, size=4000, dtype=np.uint64).astype(np.uint32)
    out = torch.zeros(hv.size, dtype=torch.int32, device="cuda")
    zr.rocm.update_hash_roll_dev(torch.from_numpy(hv.view(np.int32)).cuda(), torch.from_numpy(vv.view(np.int32)).cuda(), out)
    got = out.cpu().numpy().view(np.uint32)
    for i in range(0, hv.size, 5):
        assert got[i] == oracle.oracle_update_hash_roll(int(hv[i]), int(vv[i]))

    plans = [(0, 1), (1, 1), (2, 1), (3, 61), (64, 64), (128, 65), (193, 1000), (1193, 0), (1193, 30000), (31193, 2)]
    states = [HostState(texty(60000, s, alphabet=a, words=w)) for s, a, w in ((3, 24, 400), (4, 3, 20), (5, 26, 3000))]
    states.append(HostState(np.zeros(60000, dtype=np.uint8)))       # every position collides
    seeds = [0, 0x7fff, 0x1234, 0x5a5a]
    for hs, k in zip(states, seeds):
        hs.st.ins_h = k
    devs = [DevState(zr, hs) for hs in states]
    d_views = zr.rocm.views_to_device([d.view for d in devs])
    ins_h = torch.from_numpy(np.array(seeds, dtype=np.uint32).view(np.int32)).cuda()
    for (start, count) in plans:
        strs = torch.full((len(devs),), start, dtype=torch.int32, device="cuda")
        cnts = torch.full((len(devs),), count, dtype=torch.int32, device="cuda")
        zr.rocm.insert_string_roll_dev(d_views, len(devs), strs, cnts, ins_h)
        keys = ins_h.cpu().numpy().view(np.uint32)
        for i, (hs, d) in enumerate(zip(states, devs)):
            oracle.oracle_insert_string_roll(hs.ref(), start, count)
            assert keys[i] == hs.st.ins_h, (start, count, i)
            assert (d.head_np() == hs.head).all(), (start, count)
            assert (d.prev_np() == hs.prev).all(), (start, count)
    for pos in (40000, 40001, 40000, 12):
        strs = torch.full((len(devs),), pos, dtype=torch.int32, device="cuda")
        heads = torch.zeros(len(devs), dtype=torch.int16, device="cuda")
        zr.rocm.quick_insert_string_roll_dev(d_views, len(devs), strs, ins_h, heads)
        got = heads.cpu().numpy().view(np.uint16)
        keys = ins_h.cpu().numpy().view(np.uint32)
        for i, (hs, d) in enumerate(zip(states, devs)):
            assert got[i] == oracle.oracle_quick_insert_string_roll(hs.ref(), pos)
            assert keys[i] == hs.st.ins_h
            assert (d.head_np() == hs.head).all() and (d.prev_np() == hs.prev).all()


@pytest.mark.parametrize("level", [1, 3, 4, 6, 9])
def test_longest_match(zr, oracle, level):
    torch = torch_mod()
    data = texty(64000, 100 + level, alphabet=5, words=50)
    hs = HostState(data)
    hs.set_level(level if level > 1 else 2)
    hs.st.level = level
    oracle.oracle_insert_string(hs.ref(), 0, 63000)
    # queries share the (read-only) slabs; chains contain positions beyond strstart too, which
    # longest_match skips through `cur_match >= strstart` exactly like the reference would
    rng = np.random.default_rng(level)
    views, curs, wants = [], [], []
    dev = DevState(zr, hs)
    for strstart in rng.integers(300, 62000, size=400).tolist():
        val = int.from_bytes(hs.window[strstart:strstart + 4].tobytes(), "little")
        cur = int(hs.prev[strstart & hs.st.w_mask])        # the chain as it was when strstart was inserted
        if cur == 0 or cur >= strstart or strstart - cur > W_SIZE - 262:
            continue
        for prev_length, lookahead in ((0, 400), (3, 400), (12, 400), (0, 6), (5, 262)):
            hs.st.strstart, hs.st.prev_length, hs.st.lookahead = strstart, prev_length, lookahead
            hs.st.match_start = 0x123456
            want_len = oracle.oracle_longest_match(hs.ref(), cur)
            wants.append((want_len, hs.st.match_start))
            v = zr.rocm.DeflateView()
            C.memmove(C.byref(v), C.byref(dev.view), C.sizeof(v))
            v.strstart, v.prev_length, v.lookahead, v.match_start = strstart, prev_length, lookahead, 0x123456
            v.max_chain_length, v.good_match, v.nice_match, v.level = (hs.st.max_chain_length, hs.st.good_match,
                                                                       hs.st.nice_match, hs.st.level)
            views.append(v)
            curs.append(cur)
    assert len(views) > 500
    d_views = zr.rocm.views_to_device(views)
    d_cur = torch.from_numpy(np.array(curs, dtype=np.uint16).view(np.int16)).cuda()
    lens = torch.zeros(len(views), dtype=torch.int32, device="cuda")
    starts = torch.zeros(len(views), dtype=torch.int32, device="cuda")
    zr.rocm.longest_match_dev(d_views, len(views), d_cur, lens, starts)
    got = list(zip(lens.cpu().tolist(), starts.cpu().tolist()))
    assert got == wants


def test_chunkmemset_safe(zr, oracle):
    torch = torch_mod()
    rng = np.random.default_rng(9)
    cases = []
    slot = 34000
    dists = list(range(1, 65)) + [255, 256, 257, 32768]
    lens = list(range(1, 20)) + [31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 258, 271, 272, 273, 600, 1000, 1200]
    for d in dists:
        for ln in lens:
            cases.append((d, ln, ln + int(rng.integers(0, 40))))
    cases += [(7, 100, 60), (1, 258, 1), (300, 258, 200)]                 # left < len truncates
    ahead = [(-1, 1, 8), (-5, 5, 10), (-5, 20, 30), (-64, 258, 300), (-3, 100, 100), (-300, 258, 258)]  # from ahead of out
    n = len(cases) + len(ahead)
    base = rng.integers(0, 256, size=n * slot + 1024, dtype=np.uint8)
    out_off = np.zeros(n, dtype=np.uint64)
    from_off = np.zeros(n, dtype=np.uint64)
    ln_a = np.zeros(n, dtype=np.uint32)
    left_a = np.zeros(n, dtype=np.uint32)
    for i, (d, ln, left) in enumerate(cases + ahead):
        o = i * slot + 33000 if d > 0 else i * slot + 100
        out_off[i], from_off[i], ln_a[i], left_a[i] = o, o - d, ln, left
    model = base.copy()
    for i in range(n):
        end = oracle.oracle_chunkmemset_safe(model.ctypes.data + int(out_off[i]), model.ctypes.data + int(from_off[i]),
                                             int(ln_a[i]), int(left_a[i]))
        assert end == model.ctypes.data + int(out_off[i]) + min(int(ln_a[i]), int(left_a[i]))
    d_base = to_dev(base)
    zr.rocm.chunkmemset_safe_dev(d_base, torch.from_numpy(out_off.view(np.int64)).cuda(),
                                 torch.from_numpy(from_off.view(np.int64)).cuda(),
                                 torch.from_numpy(ln_a.view(np.int32)).cuda(),
                                 torch.from_numpy(left_a.view(np.int32)).cuda())
    got = d_base.cpu().numpy()
    bad = np.nonzero(got != model)[0]
    assert bad.size == 0, (bad[:5], (cases + ahead)[int(bad[0]) // slot])
    assert zr.rocm.chunksize() == 16
