"""Deterministic synthetic corpora: a "Silesia-like" mix of six classes,
generated with numpy from a seed so multi-hundred-MiB inputs never need committing."""
import numpy as np

CLASSES = ("text", "records", "dna", "sparse", "random", "xml")


def _from_vocab(vocab, ids):
    """concatenate vocab[ids[0]] + vocab[ids[1]] + ... without a Python loop over ids"""
    lens = np.array([len(w) for w in vocab], dtype=np.int64)
    starts = np.concatenate(([0], np.cumsum(lens)[:-1]))
    blob = np.frombuffer(b"".join(vocab), dtype=np.uint8)
    l = lens[ids]
    total = int(l.sum())
    out_start = np.concatenate(([0], np.cumsum(l)[:-1]))
    idx = np.arange(total, dtype=np.int64) - np.repeat(out_start, l) + np.repeat(starts[ids], l)
    return blob[idx]


def segment(kind, n, rng):
    if kind == "text":
        words = [bytes(rng.integers(97, 123, size=int(k), dtype=np.uint8)) + b" "
                 for k in np.clip(rng.geometric(0.25, size=5000), 1, 14)]
        words[0:40] = [b"the ", b"of ", b"and ", b"to ", b"in ", b"a ", b"is ", b"that ", b"for ", b"it ",
                       b"with ", b"as ", b"was ", b"on ", b"be ", b"by ", b"at ", b"this ", b"have ", b"from ",
                       b"or ", b"had ", b"not ", b"but ", b"what ", b"all ", b"were ", b"when ", b"we ", b"there ",
                       b"can ", b"an ", b"your ", b"which ", b"their ", b"said ", b"if ", b"do ", b".\n", b", "]
        ids = (rng.zipf(1.25, size=n // 3 + 64) - 1) % len(words)
        return _from_vocab(words, ids)[:n]
    if kind == "records":
        m = n // 16 + 1
        rec = np.zeros((m, 4), dtype=np.int32)
        rec[:, 0] = np.cumsum(rng.integers(0, 5, size=m))
        rec[:, 1] = 1000 + np.cumsum(rng.integers(-2, 3, size=m))
        rec[:, 2] = rng.integers(0, 16, size=m)
        rec[:, 3] = rng.integers(0, 2, size=m) * 0x01010101
        return rec.view(np.uint8).reshape(-1)[:n]
    if kind == "dna":
        return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)]
    if kind == "sparse":
        out = np.zeros(n, dtype=np.uint8)
        k = n // 200
        out[rng.integers(0, n, size=k)] = rng.integers(1, 256, size=k, dtype=np.uint8)
        runs = rng.integers(0, n - 600, size=n // 5000 + 1)
        for r in runs[:2000]:
            out[r:r + 500] = 0x20
        return out
    if kind == "random":
        return rng.integers(0, 256, size=n, dtype=np.uint8)
    if kind == "xml":
        tags = [b"<item id=\"", b"\">", b"</item>\n", b"<name>", b"</name>", b"<value>", b"</value>",
                b"<list>\n", b"</list>\n", b"  ", b"true", b"false", b"null"] + \
               [str(int(v)).encode() for v in rng.integers(0, 100000, size=300)] + \
               [bytes(rng.integers(97, 123, size=6, dtype=np.uint8)) for _ in range(200)]
        ids = rng.integers(0, len(tags), size=n // 4 + 64)
        ids[::3] = rng.integers(0, 10, size=ids[::3].size)
        return _from_vocab(tags, ids)[:n]
    raise ValueError(kind)


def silesia_like(nbytes, seed=0x5EED0003, seg_bytes=8 << 20):
    """nbytes of the six-class mix, segments cycling through CLASSES"""
    rng = np.random.default_rng(seed)
    parts, have, i = [], 0, 0
    while have < nbytes:
        n = min(seg_bytes, nbytes - have)
        part = segment(CLASSES[i % len(CLASSES)], n, rng)
        if part.size < n:                      # vocab assembly may come up a little short
            part = np.concatenate([part, np.zeros(n - part.size, dtype=np.uint8)])
        parts.append(part)
        have += n
        i += 1
    return np.ascontiguousarray(np.concatenate(parts)[:nbytes])
