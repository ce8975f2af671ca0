"""Host mirror of the inflate path of include/zng_rocm.h (token decode on the host, copy
resolution on the device).  Names follow the reference's API: `inflate_raw` is a one-shot
`zng_inflate` of a raw (windowBits < 0) stream, Z_* codes as in zlib-ng.h.in:180-188."""
import ctypes as C

from . import rocm

Z_OK, Z_STREAM_END, Z_DATA_ERROR, Z_MEM_ERROR, Z_BUF_ERROR = 0, 1, -3, -4, -5


class InflateTokens(C.Structure):
    """zng_rocm_inflate_tokens"""
    _fields_ = [
        ("tokens", C.POINTER(C.c_uint32)), ("ntokens", C.c_size_t),
        ("literals", C.POINTER(C.c_uint8)), ("nliterals", C.c_size_t),
        ("segs", C.POINTER(C.c_uint64)), ("nsegs", C.c_size_t),
        ("out_len", C.c_uint64), ("in_used", C.c_size_t),
        ("status", C.c_int), ("msg", C.c_char_p),
    ]


class DecodedStream:
    """numpy views of one decoded stream (copies; the C buffers are freed immediately)"""

    def __init__(self, src, window_len=0, nthreads=1):
        import numpy as np
        lib = rocm.lib()
        raw = bytes(src)
        buf = C.create_string_buffer(raw, max(len(raw), 1))
        tk = InflateTokens()
        if nthreads == 1:
            self.status = lib.zng_rocm_inflate_tokens_decode_window(C.addressof(buf), len(raw), window_len, C.byref(tk))
        else:
            self.status = lib.zng_rocm_inflate_tokens_decode_threads(C.addressof(buf), len(raw), window_len, nthreads,
                                                                     C.byref(tk))
        self.msg = (tk.msg or b"").decode()
        self.out_len = tk.out_len
        self.in_used = tk.in_used
        self.tokens = np.ctypeslib.as_array(tk.tokens, shape=(tk.ntokens,)).copy() if tk.ntokens else \
            np.zeros(0, dtype=np.uint32)
        self.literals = np.ctypeslib.as_array(tk.literals, shape=(tk.nliterals,)).copy() if tk.nliterals else \
            np.zeros(0, dtype=np.uint8)
        nseg = tk.nsegs
        self.nsegs = nseg
        self.segs = np.ctypeslib.as_array(tk.segs, shape=((nseg + 1) * 3,)).copy() if tk.segs else \
            np.zeros(3, dtype=np.uint64)
        lib.zng_rocm_inflate_tokens_free(C.byref(tk))


def decode_tokens(src, window_len=0, nthreads=1):
    """nthreads != 1: the multi-threaded decode of ONE stream (0 = one thread per hardware thread)"""
    return DecodedStream(src, window_len, nthreads)


def resolve_dev(dec, stream=None):
    """run the device stage on a DecodedStream; returns a uint8 CUDA tensor with the plaintext"""
    import numpy as np
    import torch
    rocm._need_init()
    n = int(dec.out_len)
    out = torch.empty(max(n, 1), dtype=torch.uint8, device="cuda")
    if n == 0:
        return out[:0]
    d_tok = torch.from_numpy(dec.tokens.view(np.int32)).cuda()
    d_lit = torch.from_numpy(dec.literals).cuda() if dec.literals.size else torch.zeros(1, dtype=torch.uint8,
                                                                                        device="cuda")
    d_seg = torch.from_numpy(dec.segs.view(np.int64)).cuda()
    d_sym = torch.empty(n, dtype=torch.int16, device="cuda")
    rocm._check(rocm.lib().zng_rocm_inflate_resolve_dev(
        rocm._dev_ptr(d_tok), dec.tokens.size, rocm._dev_ptr(d_lit), dec.literals.size, rocm._dev_ptr(d_seg),
        dec.nsegs, rocm._dev_ptr(d_sym), rocm._dev_ptr(out), n, rocm._stream_ptr(stream)),
        "zng_rocm_inflate_resolve_dev")
    torch.cuda.current_stream().synchronize()
    return out[:n]


def inflate_raw(src, dst, stream=None):
    """one-shot: host bytes (or a HostStream, which saves the Python copy) in, plaintext into the CUDA tensor `dst`;
    returns (zlib status, bytes produced)"""
    rocm._need_init()
    hs = src if isinstance(src, HostStream) else HostStream(src)
    produced = C.c_uint64(0)
    rc = rocm.lib().zng_rocm_inflate_raw(C.addressof(hs.buf), hs.n, rocm._dev_ptr(dst), dst.numel(),
                                         C.byref(produced), rocm._stream_ptr(stream))
    return rc, produced.value


def inflate_raw_window(src, window, dst, stream=None):
    """one-shot raw inflate of a stream that continues `window` (uint8 CUDA tensor, <= 32768 bytes: a preset
    dictionary or the tail of earlier output); returns (zlib status, bytes produced, input bytes used)"""
    rocm._need_init()
    raw = bytes(src)
    buf = C.create_string_buffer(raw, max(len(raw), 1))
    produced, used = C.c_uint64(0), C.c_size_t(0)
    wl = 0 if window is None else window.numel()
    rc = rocm.lib().zng_rocm_inflate_raw_window(C.addressof(buf), len(raw), None if not wl else rocm._dev_ptr(window), wl,
                                                rocm._dev_ptr(dst), dst.numel(), C.byref(produced), C.byref(used),
                                                rocm._stream_ptr(stream))
    return rc, produced.value, used.value


class InflateJob(C.Structure):
    """zng_rocm_inflate_job"""
    _fields_ = [("src", C.c_void_p), ("src_len", C.c_size_t), ("d_dst", C.c_void_p), ("dst_cap", C.c_size_t),
                ("d_window", C.c_void_p), ("window_len", C.c_uint32), ("status", C.c_int), ("out_len", C.c_uint64),
                ("in_used", C.c_size_t), ("msg", C.c_char_p)]


class InflateBatch:
    """a prepared zng_rocm_inflate_many call: the job array and the host copies of the streams are built once"""

    def __init__(self, streams, dsts, windows=None):
        rocm._need_init()
        self.n = len(streams)
        self._keep = [C.create_string_buffer(bytes(s), max(len(s), 1)) for s in streams]
        self._dsts, self._windows = dsts, windows
        self.jobs = (InflateJob * self.n)()
        for i in range(self.n):
            self.jobs[i].src = C.addressof(self._keep[i])
            self.jobs[i].src_len = len(streams[i])
            self.jobs[i].d_dst = dsts[i].data_ptr()
            self.jobs[i].dst_cap = dsts[i].numel()
            w = None if windows is None else windows[i]
            self.jobs[i].d_window = None if w is None or not w.numel() else w.data_ptr()
            self.jobs[i].window_len = 0 if w is None else w.numel()

    def run(self, nthreads=0):
        rc = rocm.lib().zng_rocm_inflate_many(C.byref(self.jobs), self.n, nthreads)
        rocm._check(rc, "zng_rocm_inflate_many")
        return [(self.jobs[i].status, self.jobs[i].out_len, self.jobs[i].in_used, (self.jobs[i].msg or b"").decode())
                for i in range(self.n)]


def inflate_many(streams, dsts, windows=None, nthreads=0):
    """streams: list of bytes-like raw deflate streams (host); dsts: list of uint8 CUDA tensors; windows: optional list
    of uint8 CUDA tensors (or None) holding each stream's history.  Returns [(status, out_len, in_used, msg), ...]."""
    return InflateBatch(streams, dsts, windows).run(nthreads)


class HostStream:
    """a host copy of one compressed stream, made once (keeps the timing of repeated calls free of Python copies)"""

    def __init__(self, src):
        raw = bytes(src)
        self.n = len(raw)
        self.buf = C.create_string_buffer(raw, max(len(raw), 1))


def inflate_raw_threads(src, dst, window=None, nthreads=0):
    """one raw stream, host decode on `nthreads` threads (0 = all hardware threads), device resolve; `src` bytes-like or
    a HostStream.  Returns (zlib status, bytes produced, input bytes used)"""
    rocm._need_init()
    hs = src if isinstance(src, HostStream) else HostStream(src)
    produced, used = C.c_uint64(0), C.c_size_t(0)
    wl = 0 if window is None else window.numel()
    rc = rocm.lib().zng_rocm_inflate_raw_threads(C.addressof(hs.buf), hs.n, None if not wl else rocm._dev_ptr(window), wl,
                                                 rocm._dev_ptr(dst), dst.numel(), C.byref(produced), C.byref(used), nthreads)
    return rc, produced.value, used.value


class InflateDevJob(C.Structure):
    """zng_rocm_inflate_dev_job"""
    _fields_ = [("in_ptr", C.c_void_p), ("out_ptr", C.c_void_p), ("in_len", C.c_uint64), ("out_cap", C.c_uint64),
                ("dict_len", C.c_uint32), ("flags", C.c_uint32)]


def inflate_message(msg_id):
    return (rocm.lib().zng_rocm_inflate_message(int(msg_id)) or b"").decode()


class InflateDevBatch:
    """Many raw deflate streams that already sit in device memory, decoded on the device (zng_rocm_inflate_streams_dev).

    src: uint8 CUDA tensor; stream i = src[in_off[i] : in_off[i] + in_len[i]].
    dst: uint8 CUDA tensor; stream i's plaintext goes to dst[out_off[i] : out_off[i] + out_cap[i]], with dict_len[i]
         bytes of history directly in front of out_off[i] (inside dst).
    results: int32 CUDA tensor [n, 4] = {bytes produced, input bytes used, zlib status, message id}."""

    def __init__(self, src, in_off, in_len, dst, out_off, out_cap, dict_len=None):
        import torch
        rocm._need_init()
        self.src, self.dst = src, dst
        self.n = len(in_len)
        self.results = torch.zeros((self.n, 4), dtype=torch.int32, device=src.device)
        self.jobs = (InflateDevJob * self.n)()
        bi, bo = src.data_ptr(), dst.data_ptr()
        for i in range(self.n):
            d = 0 if dict_len is None else int(dict_len[i])
            if d > int(out_off[i]):
                raise ValueError("the history must lie inside dst, in front of the stream's output")
            if int(in_off[i]) + int(in_len[i]) > src.numel() or int(out_off[i]) + int(out_cap[i]) > dst.numel():
                raise ValueError("stream %d does not fit its tensor" % i)
            self.jobs[i].in_ptr = bi + int(in_off[i])
            self.jobs[i].out_ptr = bo + int(out_off[i])
            self.jobs[i].in_len = int(in_len[i])
            self.jobs[i].out_cap = int(out_cap[i])
            self.jobs[i].dict_len = d
            self.jobs[i].flags = 0

    def run(self, stream=None):
        """asynchronous on `stream`"""
        rocm._check(rocm.lib().zng_rocm_inflate_streams_dev(C.byref(self.jobs), self.n, rocm._dev_ptr(self.results),
                                                            rocm._stream_ptr(stream)), "zng_rocm_inflate_streams_dev")

    def run_wrapped(self, fmt, stream=None):
        """the same streams with their zlib (1) / gzip (2) wrapper: header parse, inflate, check values of the outputs and
        trailer comparison, all on the device (zng_rocm_uncompress_streams_dev)"""
        rocm._check(rocm.lib().zng_rocm_uncompress_streams_dev(fmt, C.byref(self.jobs), self.n, rocm._dev_ptr(self.results),
                                                               rocm._stream_ptr(stream)), "zng_rocm_uncompress_streams_dev")

    def rows(self):
        """[(status, out_len, in_used, message)] (synchronises)"""
        r = self.results.cpu().tolist()
        return [(row[2], row[0], row[1], inflate_message(row[3])) for row in r]


def inflate_large_dev(src_dev, dst, window=None, stream=None):
    """zng_rocm_inflate_large_dev: ONE large raw stream that is already in device memory (`src_dev`: uint8 CUDA tensor),
    cut into parts and decoded on the device; plaintext into the CUDA tensor `dst`, optional history `window` (CUDA tensor,
    <= 32768 bytes).  Returns (zlib status, bytes produced, compressed bytes used, parts on the chain -- 0 when the
    sequential decoder did it)."""
    rocm._need_init()
    lib = rocm.lib()
    out_len, in_used = C.c_uint64(0), C.c_size_t(0)
    wl = 0 if window is None else int(window.numel())
    st = lib.zng_rocm_inflate_large_dev(rocm._dev_ptr(src_dev), int(src_dev.numel()),
                                        rocm._dev_ptr(window) if wl else None, wl, rocm._dev_ptr(dst), int(dst.numel()),
                                        C.byref(out_len), C.byref(in_used), rocm._stream_ptr(stream))
    return st, int(out_len.value), int(in_used.value), int(lib.zng_rocm_inflate_large_last_parts())
