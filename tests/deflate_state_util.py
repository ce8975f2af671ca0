"""Builders for deflate_state-shaped test inputs (host numpy + oracle view, device view)."""
import ctypes as C

import numpy as np

import oracle_lib

W_BITS = 15
W_SIZE = 1 << W_BITS
PAD = 512   # readable bytes past 2*w_size (the reference pads the window for compare overreads)

LEVEL_PARAMS = {   # deflate.c:142-168 configuration_table {good, lazy, nice, chain}
    1: (0, 0, 0, 0), 2: (4, 4, 8, 4), 3: (4, 6, 16, 6), 4: (4, 12, 32, 24), 5: (8, 16, 32, 32),
    6: (8, 16, 128, 128), 7: (8, 32, 128, 256), 8: (32, 128, 258, 1024), 9: (32, 258, 258, 4096),
}


def texty(n, seed, alphabet=24, words=400):
    """compressible pseudo-text: Zipf-ish draws from a small dictionary of random words"""
    rng = np.random.default_rng(seed)
    vocab = [bytes(rng.integers(97, 97 + alphabet, size=int(rng.integers(2, 9)), dtype=np.uint8)) for _ in range(words)]
    ranks = rng.zipf(1.3, size=n // 3 + 16) % words
    out = bytearray()
    for r in ranks:
        out += vocab[r] + b" "
        if len(out) >= n:
            break
    return np.frombuffer(bytes(out[:n]), dtype=np.uint8).copy()


class HostState:
    """numpy-backed window/prev/head + the oracle's struct over them"""

    def __init__(self, data, w_size=W_SIZE):
        self.w_size = w_size
        self.window = np.zeros(2 * w_size + PAD, dtype=np.uint8)
        n = min(len(data), 2 * w_size)
        self.window[:n] = data[:n]
        self.filled = n
        self.prev = np.zeros(w_size, dtype=np.uint16)
        self.head = np.zeros(65536, dtype=np.uint16)
        self.st = oracle_lib.DeflateState()
        self.st.w_size = w_size
        self.st.w_bits = int(np.log2(w_size))
        self.st.w_mask = w_size - 1
        self.st.window_size = 2 * w_size
        self.st.window = self.window.ctypes.data
        self.st.prev = self.prev.ctypes.data
        self.st.head = self.head.ctypes.data

    def set_level(self, level):
        good, lazy, nice, chain = LEVEL_PARAMS[level]
        self.st.level = level
        self.st.good_match = good
        self.st.nice_match = nice
        self.st.max_chain_length = chain

    def ref(self):
        return C.byref(self.st)
