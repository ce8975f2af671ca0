"""Raw deflate streams built token by token (test infrastructure): fixed-Huffman and stored blocks from explicit
('L', byte) / ('M', length, distance) tokens, so that a test can put a copy of a chosen length at a chosen distance at a
chosen output position.  RFC 1951 3.2.5 / 3.2.6 (inftrees.c:38-49 and inffixed_tbl.h hold the same tables)."""

_LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
_LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
_DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
              8193, 12289, 16385, 24577]
_DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


class Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value, nbits):                   # LSB first (extra bits, header fields)
        self.acc |= value << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def put_code(self, code, nbits):               # Huffman codes go in most significant bit first
        rev = int(format(code, "0%db" % nbits)[::-1], 2)
        self.put(rev, nbits)

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)

    def bit_length(self):
        return 8 * len(self.out) + self.n


def _fixed_litlen(sym):
    if sym < 144:
        return 0x30 + sym, 8
    if sym < 256:
        return 0x190 + sym - 144, 9
    if sym < 280:
        return sym - 256, 7
    return 0xC0 + sym - 280, 8


def fixed_block(bits, tokens, final):
    bits.put(1 if final else 0, 1)
    bits.put(1, 2)
    for t in tokens:
        if t[0] == "L":
            bits.put_code(*_fixed_litlen(t[1]))
            continue
        _, length, dist = t
        k = max(i for i in range(29) if _LEN_BASE[i] <= length) if length < 258 else 28
        bits.put_code(*_fixed_litlen(257 + k))
        bits.put(length - _LEN_BASE[k], _LEN_EXTRA[k])
        d = max(i for i in range(30) if _DIST_BASE[i] <= dist)
        bits.put_code(d, 5)
        bits.put(dist - _DIST_BASE[d], _DIST_EXTRA[d])
    bits.put_code(*_fixed_litlen(256))


def stored_block(bits, data, final):
    assert len(data) <= 65535
    bits.put(1 if final else 0, 1)
    bits.put(0, 2)
    bits.align()
    bits.put(len(data), 16)
    bits.put(len(data) ^ 0xFFFF, 16)
    bits.out += data                               # aligned here


def replay(tokens, history=b""):
    """the bytes the tokens produce (history = what precedes the output)"""
    buf = bytearray(history)
    for t in tokens:
        if t[0] == "L":
            buf.append(t[1])
        else:
            _, length, dist = t
            assert dist <= len(buf)
            for _ in range(length):
                buf.append(buf[-dist])
    return bytes(buf[len(history):])
