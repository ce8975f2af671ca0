"""CPU: inflate of a stream that continues history -- a preset dictionary (inflateSetDictionary on a raw stream,
inflate.c:1214-1261) or the window an earlier call left (inflate.c:325-378).  The oracle's dictionary form is pinned
against CPython's zlib (zdict=), then the product's host token decoder (window_len) against the oracle."""
import importlib
import zlib

import numpy as np
import pytest

import inflate_util
import synth


def _inf():
    importlib.import_module("zlib-ng_amd")
    return importlib.import_module("zlib-ng_amd.inflate")


def replay_window(dec, window):
    """token interpreter with `window` in front of the output (test-side checker)"""
    out = bytearray(window)
    lit, lp = dec.literals.tobytes(), 0
    for tok in dec.tokens.tolist():
        if tok >> 31:
            ln, dist = ((tok >> 16) & 0xff) + 3, (tok & 0xffff) + 1
            assert dist <= len(out)
            for _ in range(ln):
                out.append(out[-dist])
        else:
            out += lit[lp:lp + tok]
            lp += tok
    return bytes(out[len(window):])


@pytest.mark.parametrize("dict_len", [0, 1, 258, 4000, 32768, 50000])
def test_dictionary_streams(dict_len):
    inf = _inf()
    text = synth.silesia_like(400000, seed=321, seg_bytes=100000).tobytes()
    dictionary, data = text[:dict_len], text[60000:260000]
    if dict_len:
        c = zlib.compressobj(6, zlib.DEFLATED, -15, zdict=dictionary)
    else:
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    st, msg, out, used = inflate_util.oracle_inflate_dict(comp, dictionary, cap=len(data) + 16)
    assert (st, out, used) == (1, data, len(comp)), (st, msg)
    window = dictionary[-32768:]
    dec = inf.decode_tokens(comp, window_len=len(window))
    assert dec.status == 1 and dec.out_len == len(data) and replay_window(dec, window) == data
    if dict_len >= 4000:
        # the same stream without its dictionary: the reference's message, from oracle and product alike
        st, msg, _, _ = inflate_util.oracle_inflate(comp, cap=len(data) + 16)
        dec = inf.decode_tokens(comp)
        assert (st, msg) == (-3, "invalid distance too far back") == (dec.status, dec.msg)
        d = zlib.decompressobj(-15)
        with pytest.raises(zlib.error):
            d.decompress(comp)


def test_window_length_is_a_limit():
    """a distance one byte beyond the window is refused (inffast_tpl.h:198-226, whave = window_len)"""
    inf = _inf()
    dictionary = bytes(np.random.default_rng(5).integers(0, 256, size=1000, dtype=np.uint8))
    c = zlib.compressobj(9, zlib.DEFLATED, -15, zdict=dictionary)
    comp = c.compress(dictionary[:600]) + c.flush()          # a literal, then matches at distance 1000
    dec = inf.decode_tokens(comp, window_len=1000)
    assert dec.status == 1 and dec.tokens[0] == 1 and (int(dec.tokens[1]) & 0xffff) + 1 == 1000
    assert inf.decode_tokens(comp, window_len=999).status == 1       # 1 byte produced + 999 of window: just reachable
    dec = inf.decode_tokens(comp, window_len=998)
    assert (dec.status, dec.msg) == (-3, "invalid distance too far back")
