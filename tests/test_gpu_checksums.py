"""GPU parity: HIP Adler-32 / CRC-32 (through the C ABI) vs the CPU oracle.

Bit-exact is the bar (integer work).  Layers:
  * the reference's own KAT tables through the host-pointer functable slots,
  * seeded random buffers, every 16-byte phase, ragged lengths, odd seeds, through the
    device-resident entry points,
  * BASELINE.json sizes (64 MiB, 1 GiB) against the oracle and through the size-independent
    property "combine of per-block checksums == one-shot" (fuzzer_checksum.c:33-75).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from gpu_common import product, seeded_bytes, to_dev, torch_mod

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def zr():
    m = product()
    m.init()
    assert m.available()
    return m


def _rows(name):
    return json.load(open(os.path.join(HERE, "golden", name)))["rows"]


def test_adler32_slot_reference_kats(zr):
    for r in _rows("adler32_kat.json"):
        data = None if r["data_hex"] is None else bytes.fromhex(r["data_hex"])
        assert zr.adler32_z(r["seed"], data, r["len"]) == r["expect"], r


def test_crc32_slot_reference_kats(zr):
    for r in _rows("crc32_kat.json"):
        data = None if r["data_hex"] is None else bytes.fromhex(r["data_hex"])
        if data is not None and r["len"] == 0:
            got = zr.crc32_z(r["seed"], data, 0)
        else:
            got = zr.crc32_z(r["seed"], data, r["len"])
        assert got == r["expect"], r


SIZES = [0, 1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 65, 255, 4095, 4096, 5551, 5552, 5553,
         16383, 16384, 16385, 16384 * 3 + 5, 65536, 100000, 16384 * 256, 16384 * 257 + 4093,
         (1 << 22) + 1, 16384 * 1300 + 77]
SEEDS = [(1, 0), (0, 0xffffffff), (0xdeadc0de, 0xdeadbeef), (0xffffffff, 0x12345678)]


def test_dev_random_sizes_phases_seeds(zr, oracle):
    torch = torch_mod()
    big = seeded_bytes(16384 * 1300 + 77 + 64, 42)
    dbig = to_dev(big)
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    for i, n in enumerate(SIZES):
        for phase in sorted({0, 1, 7, 15, (i * 5) % 16}):
            for (sa, sc) in SEEDS[: 2 if n > 100000 else 4]:
                host_ptr = big.ctypes.data + phase
                want_a = oracle.oracle_adler32(sa, host_ptr, n) if n != 1 or True else 0
                want_c = oracle.oracle_crc32_braid(sc, host_ptr, n)
                zr.adler32_dev(dbig, out, adler=sa, length=n, offset=phase)
                got_a = out[0].item() & 0xffffffff
                zr.crc32_dev(dbig, out, crc=sc, length=n, offset=phase)
                got_c = out[0].item() & 0xffffffff
                zr.adler32_crc32_dev(dbig, out, adler=sa, crc=sc, length=n, offset=phase)
                both = [v & 0xffffffff for v in out.tolist()]
                assert got_a == want_a, (n, phase, hex(sa))
                assert got_c == want_c, (n, phase, hex(sc))
                assert both == [want_a, want_c], (n, phase)


def test_dev_structured_inputs(zr, oracle):
    """all-zero, all-0xff and ramp inputs (overflow corners of the lane sums)"""
    torch = torch_mod()
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    n = 16384 * 520 + 123
    for fill in ("zero", "ff", "ramp"):
        if fill == "zero":
            arr = np.zeros(n, dtype=np.uint8)
        elif fill == "ff":
            arr = np.full(n, 255, dtype=np.uint8)
        else:
            arr = (np.arange(n) % 251).astype(np.uint8)
        d = to_dev(arr)
        zr.adler32_crc32_dev(d, out, adler=1, crc=0)
        got = [v & 0xffffffff for v in out.tolist()]
        assert got == [oracle.oracle_adler32(1, arr.ctypes.data, n), oracle.oracle_crc32_braid(0, arr.ctypes.data, n)]


def test_fold_copy_dev(zr, oracle):
    torch = torch_mod()
    src = seeded_bytes(16384 * 70 + 999, 7)
    dsrc = to_dev(src)
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    for n, phase in ((0, 0), (5, 3), (16384 * 70 + 900, 9), (16384 * 64, 0), (40000, 15)):
        ddst = torch.full((n + 64,), 0xAA, dtype=torch.uint8, device="cuda")
        zr.fold_copy_dev(3, ddst, dsrc, out, adler=1, crc=0, length=n, src_offset=phase, dst_offset=16 + phase)
        got = [v & 0xffffffff for v in out.tolist()]
        hp = src.ctypes.data + phase
        assert got == [oracle.oracle_adler32(1, hp, n), oracle.oracle_crc32_braid(0, hp, n)]
        back = ddst.cpu().numpy()
        assert (back[16 + phase:16 + phase + n] == src[phase:phase + n]).all()
        assert (back[:16 + phase] == 0xAA).all() and (back[16 + phase + n:] == 0xAA).all()
    # host-pointer slots
    val, copied = zr.adler32_fold_copy(1, src[:50001])
    assert val == oracle.oracle_adler32(1, src.ctypes.data, 50001) and copied == src[:50001].tobytes()
    st = zr.Crc32Fold()
    st.fold(src[:777])
    assert st.fold_copy(src[777:9000]) == src[777:9000].tobytes()
    assert st.final() == oracle.oracle_crc32_braid(0, src.ctypes.data, 9000)


def test_combine_dev_matches_oracle_fold(zr, oracle):
    torch = torch_mod()
    rng = np.random.default_rng(3)
    data = seeded_bytes(3_000_000, 11)
    for count in (0, 1, 2, 7, 1024, 1025, 3000):
        cuts = np.sort(rng.integers(0, data.size + 1, size=max(count - 1, 0)))
        edges = np.concatenate(([0], cuts, [data.size])) if count else np.array([0], dtype=np.int64)
        lens = np.diff(edges).astype(np.uint64) if count else np.zeros(0, dtype=np.uint64)
        a_chk = np.zeros(count, dtype=np.uint32)
        c_chk = np.zeros(count, dtype=np.uint32)
        for i in range(count):
            p = data.ctypes.data + int(edges[i])
            a_chk[i] = oracle.oracle_adler32(1, p, int(lens[i]))
            c_chk[i] = oracle.oracle_crc32_braid(0, p, int(lens[i]))
        # oracle left fold
        wa, wc = 1, 0
        for i in range(count):
            wa = oracle.oracle_adler32_combine(wa, int(a_chk[i]), int(lens[i]))
            wc = oracle.oracle_crc32_combine(wc, int(c_chk[i]), int(lens[i]))
        out = torch.zeros(2, dtype=torch.int32, device="cuda")
        d_lens = torch.from_numpy(lens.view(np.int64)).cuda()
        zr.adler32_combine_dev(torch.from_numpy(a_chk.view(np.int32)).cuda(), d_lens, out)
        ga = out[0].item() & 0xffffffff
        zr.crc32_combine_dev(torch.from_numpy(c_chk.view(np.int32)).cuda(), d_lens, out)
        gc = out[0].item() & 0xffffffff
        assert (ga, gc) == (wa, wc), count
        if count:
            assert wa == oracle.oracle_adler32(1, data.ctypes.data, data.size)
            assert wc == oracle.oracle_crc32_braid(0, data.ctypes.data, data.size)
        # host scalar forms agree with the oracle's
        if count >= 2:
            assert zr.adler32_combine(int(a_chk[0]), int(a_chk[1]), int(lens[1])) == \
                oracle.oracle_adler32_combine(int(a_chk[0]), int(a_chk[1]), int(lens[1]))
            assert zr.crc32_combine(int(c_chk[0]), int(c_chk[1]), int(lens[1])) == \
                oracle.oracle_crc32_combine(int(c_chk[0]), int(c_chk[1]), int(lens[1]))


@pytest.mark.parametrize("mib", [64, 1024])
def test_baseline_sizes(zr, oracle, mib):
    """cfg1 (64 MiB) and cfg2 (1 GiB): one-shot vs oracle, and block-combine == one-shot."""
    torch = torch_mod()
    n = mib << 20
    g = torch.Generator(device="cuda")
    g.manual_seed(0x5EED0002)
    d = torch.randint(0, 256, (n + 16,), dtype=torch.uint8, device="cuda", generator=g)
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    for phase in (0, 3):
        zr.adler32_crc32_dev(d, out, adler=1, crc=0, length=n, offset=phase)
        got = [v & 0xffffffff for v in out.tolist()]
        # property check, independent of the oracle: 37 ragged blocks, combine on device
        edges = np.linspace(0, n, 38).astype(np.int64)
        edges[1:-1] += np.arange(1, 37) * 7
        a_chk = torch.zeros(37, dtype=torch.int32, device="cuda")
        c_chk = torch.zeros(37, dtype=torch.int32, device="cuda")
        tmp = torch.zeros(2, dtype=torch.int32, device="cuda")
        for i in range(37):
            zr.adler32_crc32_dev(d, tmp, length=int(edges[i + 1] - edges[i]), offset=phase + int(edges[i]))
            a_chk[i] = tmp[0]
            c_chk[i] = tmp[1]
        d_lens = torch.from_numpy(np.diff(edges)).cuda()
        zr.adler32_combine_dev(a_chk, d_lens, tmp)
        ca = tmp[0].item() & 0xffffffff
        zr.crc32_combine_dev(c_chk, d_lens, tmp)
        cc = tmp[0].item() & 0xffffffff
        assert [ca, cc] == got
        if phase == 0:
            host = d.cpu().numpy()
            assert got[0] == oracle.oracle_adler32(1, host.ctypes.data, n)
            assert got[1] == oracle.oracle_crc32_braid(0, host.ctypes.data, n)
            del host


def test_slots_from_concurrent_host_threads(zr, oracle):
    """The reference's slots are called from any number of application threads at once (its
    test_deflate_concurrency.cc drives two streams from two threads).  Each host thread gets its own HIP
    stream and workspace inside the library; ctypes drops the GIL for the call, so these really overlap."""
    import threading
    bufs = [seeded_bytes((1 << 20) * (k + 1) + 17 * k, seed=100 + k) for k in range(4)]
    want = [(oracle.oracle_adler32(1, b.ctypes.data, b.size), oracle.oracle_crc32(0, b.ctypes.data, b.size))
            for b in bufs]
    errors = []

    def worker(k):
        try:
            for _ in range(12):
                a = zr.adler32_z(1, bufs[k])
                c = zr.crc32_z(0, bufs[k])
                if (a, c) != want[k]:
                    errors.append((k, a, c, want[k]))
                    return
        except Exception as e:      # noqa: BLE001 - surfaced through the assert below
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_beyond_4gib(zr, oracle):
    """size_t lengths: 5 GiB + 12345 bytes in one call (every 32-bit byte count or offset would wrap).  Checked
    against the oracle on the host and, independently, against the device combine of three ragged parts."""
    torch = torch_mod()
    n = (5 << 30) + 12345
    g = torch.Generator(device="cuda")
    g.manual_seed(0x5EED0009)
    d = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
    step = 1 << 30
    for lo in range(0, n + 16, step):                       # filled in 1 GiB pieces: randint temporaries stay small
        hi = min(n + 16, lo + step)
        d[lo:hi] = torch.randint(0, 256, (hi - lo,), dtype=torch.uint8, device="cuda", generator=g)
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    zr.adler32_crc32_dev(d, out, adler=1, crc=0, length=n, offset=5)
    got = [v & 0xffffffff for v in out.tolist()]
    edges = [0, (1 << 32) + 77, (1 << 32) + 78, n]          # one part longer than 4 GiB, one of a single byte
    a_chk = torch.zeros(3, dtype=torch.int32, device="cuda")
    c_chk = torch.zeros(3, dtype=torch.int32, device="cuda")
    tmp = torch.zeros(2, dtype=torch.int32, device="cuda")
    for i in range(3):
        zr.adler32_crc32_dev(d, tmp, length=edges[i + 1] - edges[i], offset=5 + edges[i])
        a_chk[i], c_chk[i] = tmp[0], tmp[1]
    d_lens = torch.tensor(np.diff(np.array(edges, dtype=np.int64)), device="cuda")
    zr.adler32_combine_dev(a_chk, d_lens, tmp)
    ca = tmp[0].item() & 0xffffffff
    zr.crc32_combine_dev(c_chk, d_lens, tmp)
    assert [ca, tmp[0].item() & 0xffffffff] == got
    host = d[5:5 + n].cpu().numpy()
    assert got[0] == oracle.oracle_adler32(1, host.ctypes.data, n)
    assert got[1] == oracle.oracle_crc32(0, host.ctypes.data, n)
    # host-pointer slot with a > 4 GiB buffer (PCIe staging inside)
    assert zr.crc32_z(0, host) == got[1]


def test_combine_rows_dev(zr, oracle):
    """zng_rocm_combine_rows_dev: packed {adler, crc, len} rows (the multi-GPU aggregate's payload) folded in order
    on the device == the oracle's left fold with adler32_combine / crc32_combine; lengths above 2^32 included."""
    torch = torch_mod()
    rng = np.random.default_rng(31)
    for count in (1, 2, 3, 8, 64, 1000, 1025, 3000):
        rows = np.zeros(count, dtype=[("adler", "<u4"), ("crc", "<u4"), ("len", "<u8")])
        pieces = [bytes(rng.integers(0, 256, size=int(rng.integers(0, 40)), dtype=np.uint8)) for _ in range(count)]
        for i, pbytes in enumerate(pieces):
            arr = np.frombuffer(pbytes, dtype=np.uint8)
            rows["adler"][i] = oracle.oracle_adler32(1, arr.ctypes.data if arr.size else None, arr.size)
            rows["crc"][i] = oracle.oracle_crc32(0, arr.ctypes.data if arr.size else None, arr.size)
            rows["len"][i] = arr.size
        whole = np.frombuffer(b"".join(pieces) + b"\0", dtype=np.uint8)
        n = whole.size - 1
        want = [oracle.oracle_adler32(1, whole.ctypes.data, n), oracle.oracle_crc32(0, whole.ctypes.data, n)]
        d_rows = torch.from_numpy(rows.view(np.int32).copy()).cuda()
        out = torch.zeros(2, dtype=torch.int32, device="cuda")
        zr.combine_rows_dev(d_rows, count, out)
        assert [v & 0xffffffff for v in out.tolist()] == want, count
    # lengths beyond 2^32: against the oracle's scalar combine
    rows = np.zeros(3, dtype=[("adler", "<u4"), ("crc", "<u4"), ("len", "<u8")])
    rows["adler"] = [0x12345678 % 0xfff1fff1, 0x0badf00d, 0x00010001]
    rows["adler"][0] = (0x1234 << 16) | 0x0567
    rows["adler"][1] = (0xfff0 << 16) | 0xfff0
    rows["crc"] = [0xdeadbeef, 0x01234567, 0xffffffff]
    rows["len"] = [(1 << 33) + 5, 7, (3 << 32) + 11]
    a, c = 1, 0
    for r in rows:
        a = oracle.oracle_adler32_combine(a, int(r["adler"]), int(r["len"]))
        c = oracle.oracle_crc32_combine(c, int(r["crc"]), int(r["len"]))
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    zr.combine_rows_dev(torch.from_numpy(rows.view(np.int32).copy()).cuda(), 3, out)
    assert [v & 0xffffffff for v in out.tolist()] == [a, c]


def test_try_forms_and_bounded_staging(zr, oracle):
    """zng_rocm_<slot>_try (the error channel the reference-side adapter binds) returns the slot's value; the
    host-pointer path goes through one bounded 16 MiB staging chunk with the seed chained on the device, so sizes
    around and above the chunk -- and non-canonical seeds across chunk borders -- must still be bit-exact."""
    import ctypes as C
    h = zr.lib()
    out = C.c_uint32(0)
    chunk = 16 << 20
    for n in (0, 1, 15, chunk - 1, chunk, chunk + 1, 2 * chunk + 12345, 3 * chunk):
        buf = seeded_bytes(n + 1, seed=4000 + n % 97)[:n]
        p = buf.ctypes.data if n else seeded_bytes(1, 1).ctypes.data
        for seed_a, seed_c in ((1, 0), (0xdeadc0de, 0xffffffff), (0xffffffff, 0x12345678)):
            assert h.zng_rocm_adler32_try(seed_a, p, n, C.byref(out)) == 0
            assert out.value == oracle.oracle_adler32(seed_a, p, n), (n, hex(seed_a))
            assert h.zng_rocm_crc32_try(seed_c, p, n, C.byref(out)) == 0
            assert out.value == oracle.oracle_crc32(seed_c, p, n), (n, hex(seed_c))
    n = 2 * chunk + 777
    src = seeded_bytes(n, seed=77)
    dst = np.zeros(n + 32, dtype=np.uint8)
    dst[:] = 0xA5
    assert h.zng_rocm_adler32_fold_copy_try(1, dst.ctypes.data + 16, src.ctypes.data, n, C.byref(out)) == 0
    assert out.value == oracle.oracle_adler32(1, src.ctypes.data, n)
    assert bytes(dst[16:16 + n]) == bytes(src) and set(dst[:16]) == {0xA5} and set(dst[16 + n:]) == {0xA5}
    st = zr.rocm.Crc32FoldState()
    h.zng_rocm_crc32_fold_reset(C.byref(st))
    assert h.zng_rocm_crc32_fold_try(C.byref(st), src.ctypes.data, chunk + 5, 0) == 0
    dst[:] = 0
    assert h.zng_rocm_crc32_fold_copy_try(C.byref(st), dst.ctypes.data, src.ctypes.data + chunk + 5, n - chunk - 5) == 0
    assert h.zng_rocm_crc32_fold_final(C.byref(st)) == oracle.oracle_crc32(0, src.ctypes.data, n)
    assert bytes(dst[:n - chunk - 5]) == bytes(src[chunk + 5:])
    assert h.zng_rocm_adler32_try(1, src.ctypes.data, n, None) == -3


def test_slots_from_a_thread_that_never_selected_the_device(zr, oracle):
    """HIP's current device is per host thread; a slot call from a fresh thread must still land on the backend's
    device.  On a one-GPU box the observable part is that fresh threads work and release their
    per-thread stream when they end."""
    import threading
    buf = seeded_bytes((5 << 20) + 3, seed=9)
    want = oracle.oracle_crc32(0, buf.ctypes.data, buf.size)
    got = []
    for _ in range(6):
        t = threading.Thread(target=lambda: got.append(zr.crc32_z(0, buf)))
        t.start()
        t.join()
    assert got == [want] * 6


def test_many_messages_in_one_pass(zr, oracle):
    """zng_rocm_checksums_dev: ragged messages at every 16-byte phase, empty and tiny ones, non-canonical seeds, more
    messages than one grid holds rows, one message large enough for several workgroups -- each value bit-exact against
    the oracle (adler32_c.c / crc32_braid_c.c restated) and CPython's zlib"""
    import zlib
    torch = torch_mod()
    rng = np.random.default_rng(404)
    total = 96 << 20
    host = rng.integers(0, 256, size=total + 64, dtype=np.uint8)
    dev = torch.from_numpy(host).cuda()
    lens = [0, 1, 2, 15, 16, 17, 31, 4095, 16383, 16384, 16385, 65536 + 7, (1 << 20) + 3, (1 << 20), 300001]
    lens += [int(v) for v in rng.integers(0, 70000, size=200)]
    lens += [(40 << 20) + 11]                                           # several workgroups for this one
    offs, pos = [], 3
    for i, n in enumerate(lens):
        offs.append(pos)
        pos += n + int(rng.integers(0, 37))                             # every alignment
    assert pos < total
    adlers = [1] * len(lens)
    crcs = [0] * len(lens)
    for i in range(0, len(lens), 7):                                    # chained / non-canonical seeds
        adlers[i] = int(rng.integers(0, 1 << 32))
        crcs[i] = int(rng.integers(0, 1 << 32))
    for which in (1, 2, 3):
        out = torch.full((len(lens), 2), -1, dtype=torch.int32, device="cuda")
        zr.checksums_dev(which, dev, offs, lens, out, adlers, crcs)
        got = out.cpu().numpy().astype(np.int64) & 0xffffffff
        for i, (o, n) in enumerate(zip(offs, lens)):
            if which & 1:
                assert got[i, 0] == oracle.oracle_adler32(adlers[i], host.ctypes.data + o, n), (which, i, n)
            else:
                assert got[i, 0] == 0xffffffff                          # untouched
            if which & 2:
                assert got[i, 1] == oracle.oracle_crc32_braid(crcs[i], host.ctypes.data + o, n), (which, i, n)
            else:
                assert got[i, 1] == 0xffffffff
    i = lens.index((1 << 20) + 3)
    seg = host[offs[i]:offs[i] + lens[i]].tobytes()
    assert got[i, 0] == zlib.adler32(seg, adlers[i]) and got[i, 1] == zlib.crc32(seg, crcs[i])
    # more rows than one launch holds (32768): 40000 messages of 4 KiB
    n = 40000
    offs2 = [i * 2048 + (i % 13) for i in range(n)]
    out = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
    zr.checksums_dev(3, dev, offs2, [4096] * n, out)
    got = out.cpu().numpy().astype(np.int64) & 0xffffffff
    for i in (0, 1, 32767, 32768, 39999):
        seg = host[offs2[i]:offs2[i] + 4096].tobytes()
        assert got[i, 0] == zlib.adler32(seg) and got[i, 1] == zlib.crc32(seg), i
