"""Loader for tests/golden/ref_fixtures: data files the reference's own tests hold (MANIFEST.json cites the
cmake lines that use each).  Test infrastructure."""
import hashlib
import json
import os

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixtures")
MANIFEST = json.load(open(os.path.join(DIR, "MANIFEST.json")))


def read(entry):
    data = open(os.path.join(DIR, entry["file"]), "rb").read()
    assert len(data) == entry["bytes"] and hashlib.sha256(data).hexdigest() == entry["sha256"], entry["file"]
    return data


def compressed():
    return [(e, read(e)) for e in MANIFEST["compressed"]]


def plain():
    return [(e, read(e)) for e in MANIFEST["plain"]]


def gzip_payload(data):
    """(offset of the raw deflate stream, header flags) of a gzip member (RFC 1952 2.3)"""
    assert data[:3] == b"\x1f\x8b\x08"
    flags, pos = data[3], 10
    if flags & 4:
        pos += 2 + (data[pos] | (data[pos + 1] << 8))
    for bit in (8, 16):
        if flags & bit:
            pos = data.index(b"\0", pos) + 1
    if flags & 2:
        pos += 2
    return pos, flags
