"""Host mirror of the multi-stream deflate path of include/zng_rocm.h.

`deflate_quick_batch` compresses many independent streams (pigz-style, BASELINE.json configs[4]) that are
already resident in HBM; names follow the reference (`deflateBound`-style bound, level-1 = deflate_quick,
deflate.c:142-168)."""
import ctypes as C

from . import rocm


class StreamJob(C.Structure):
    """zng_rocm_stream_job"""
    _fields_ = [("in_ptr", C.c_void_p), ("out_ptr", C.c_void_p), ("in_len", C.c_uint32), ("out_cap", C.c_uint32),
                ("dict_len", C.c_uint32), ("flags", C.c_uint32)]


BLOCK_NOT_FINAL, BLOCK_SYNC_FLUSH = 1, 2


def deflate_quick_bound(n):
    return rocm.lib().zng_rocm_deflate_quick_bound(n)


class QuickBatch:
    """A fixed set of streams laid out in two device tensors.

    src:  uint8 CUDA tensor holding the streams; stream i = src[in_off[i] : in_off[i] + in_len[i]],
          every in_off a multiple of 16 and the tensor padded to a multiple of 16.
    The compressed streams land in `dst` at out_off[i] (bound-sized slots); `results` is an int32
    CUDA tensor [n, 2] = {compressed length, adler32 of the input}.
    """

    def __init__(self, src, in_off, in_len, dict_len=None, flags=None):
        """dict_len[i] bytes in front of stream i (inside `src`) prime its hash; flags[i] = BLOCK_* bits"""
        import torch
        rocm._need_init()
        self.src = src
        self.n = len(in_len)
        self.in_off = [int(v) for v in in_off]
        self.in_len = [int(v) for v in in_len]
        self.bounds = [deflate_quick_bound(v) for v in self.in_len]
        self.out_off = [0] * self.n
        total = 0
        for i, b in enumerate(self.bounds):
            self.out_off[i] = total
            total += b
        self.dst = torch.empty(max(total, 16), dtype=torch.uint8, device=src.device)
        self.results = torch.zeros((self.n, 2), dtype=torch.int32, device=src.device)
        jobs = (StreamJob * self.n)()
        base_in, base_out = src.data_ptr(), self.dst.data_ptr()
        for i in range(self.n):
            jobs[i].dict_len = 0 if dict_len is None else int(dict_len[i])
            jobs[i].flags = 0 if flags is None else int(flags[i])
            if jobs[i].dict_len > self.in_off[i]:
                raise ValueError("the dictionary must lie inside src, in front of the stream")
            jobs[i].in_ptr = base_in + self.in_off[i]
            jobs[i].out_ptr = base_out + self.out_off[i]
            jobs[i].in_len = self.in_len[i]
            jobs[i].out_cap = self.bounds[i]
        self.jobs = jobs

    def run(self, stream=None):
        """asynchronous on `stream`: K1 match/parse + K2 static-Huffman emit for every stream"""
        rocm._check(rocm.lib().zng_rocm_deflate_quick_dev(C.byref(self.jobs), self.n, rocm._dev_ptr(self.results),
                                                          rocm._stream_ptr(stream)), "zng_rocm_deflate_quick_dev")

    def compressed(self, i, results_host=None):
        """bytes of stream i (synchronises)"""
        res = self.results.cpu() if results_host is None else results_host
        clen = int(res[i, 0])
        o = self.out_off[i]
        return self.dst[o:o + clen].cpu().numpy().tobytes()


class WrappedBatch:
    """zng_rocm_compress_streams_dev: many streams, level-1 class, with their zlib (fmt 1) / gzip (fmt 2) wrapper written on
    the device.  Layout as QuickBatch; results: int32 CUDA tensor [n, 2] = {total bytes, check value}."""

    def __init__(self, src, in_off, in_len, fmt):
        import torch
        rocm._need_init()
        self.src, self.fmt = src, fmt
        self.n = len(in_len)
        self.bounds = [(rocm.lib().zng_rocm_compress_streams_bound(int(v), fmt) + 15) & ~15 for v in in_len]
        self.out_off, total = [], 0
        for b in self.bounds:
            self.out_off.append(total)
            total += b
        self.dst = torch.zeros(max(total, 16), dtype=torch.uint8, device=src.device)
        self.results = torch.zeros((self.n, 2), dtype=torch.int32, device=src.device)
        self.jobs = (StreamJob * self.n)()
        bi, bo = src.data_ptr(), self.dst.data_ptr()
        for i in range(self.n):
            self.jobs[i].in_ptr = bi + int(in_off[i])
            self.jobs[i].out_ptr = bo + self.out_off[i]
            self.jobs[i].in_len = int(in_len[i])
            self.jobs[i].out_cap = self.bounds[i]
            self.jobs[i].dict_len = 0
            self.jobs[i].flags = 0

    def run(self, stream=None):
        rocm._check(rocm.lib().zng_rocm_compress_streams_dev(self.fmt, C.byref(self.jobs), self.n, rocm._dev_ptr(self.results),
                                                             rocm._stream_ptr(stream)), "zng_rocm_compress_streams_dev")

    def compressed(self, i, results_host=None):
        res = self.results.cpu() if results_host is None else results_host
        o = self.out_off[i]
        return self.dst[o:o + int(res[i, 0])].cpu().numpy().tobytes()


def deflate_bound(n):
    return rocm.lib().zng_rocm_deflate_bound(n)


class StreamsBatch:
    """Many independent device-resident streams at a chain level (zng_rocm_deflate_streams_dev).  Layout as QuickBatch:
    stream i = src[in_off[i] : in_off[i] + in_len[i]] with dict_len[i] bytes of history in front of it; the compressed
    streams land in `dst` at out_off[i] (bound-sized slots); run() returns the list of compressed lengths."""

    def __init__(self, src, in_off, in_len, dict_len=None, flags=None):
        import torch
        rocm._need_init()
        self.src = src
        self.n = len(in_len)
        self.in_len = [int(v) for v in in_len]
        self.bounds = [(deflate_bound(v) + 15) & ~15 for v in self.in_len]
        self.out_off, total = [], 0
        for b in self.bounds:
            self.out_off.append(total)
            total += b
        self.dst = torch.empty(max(total, 16), dtype=torch.uint8, device=src.device)
        self.jobs = (StreamJob * self.n)()
        bi, bo = src.data_ptr(), self.dst.data_ptr()
        for i in range(self.n):
            self.jobs[i].dict_len = 0 if dict_len is None else int(dict_len[i])
            self.jobs[i].flags = 0 if flags is None else int(flags[i])
            if self.jobs[i].dict_len > int(in_off[i]):
                raise ValueError("the dictionary must lie inside src, in front of the stream")
            self.jobs[i].in_ptr = bi + int(in_off[i])
            self.jobs[i].out_ptr = bo + self.out_off[i]
            self.jobs[i].in_len = self.in_len[i]
            self.jobs[i].out_cap = self.bounds[i]
        self.out_lens = (C.c_size_t * self.n)()

    def run(self, level=6, stream=None):
        rocm._check(rocm.lib().zng_rocm_deflate_streams_dev(level, C.byref(self.jobs), self.n, C.byref(self.out_lens),
                                                            rocm._stream_ptr(stream)), "zng_rocm_deflate_streams_dev")
        return [int(v) for v in self.out_lens]

    def compressed(self, i):
        o = self.out_off[i]
        return self.dst[o:o + int(self.out_lens[i])].cpu().numpy().tobytes()


def deflate_dev(src, level=6, length=None, offset=0, stream=None, dict_len=0, flags=0):
    """one large device-resident stream (or, with dict_len / flags, one BLOCK of a longer stream whose dict_len
    bytes of history sit in src in front of `offset`) -> (uint8 CUDA tensor with raw deflate, compressed length).
    level 0 = stored, 1 = single probe, 2..9 = chain walk."""
    import torch
    rocm._need_init()
    n = src.numel() - offset if length is None else length
    cap = deflate_bound(n)
    dst = torch.empty(cap, dtype=torch.uint8, device=src.device)
    out_len = C.c_size_t(0)
    rc = rocm.lib().zng_rocm_deflate_block_dev(level, rocm._dev_ptr(src, offset), n, dict_len, flags, rocm._dev_ptr(dst),
                                               cap, C.byref(out_len), rocm._stream_ptr(stream))
    rocm._check(rc, "zng_rocm_deflate_block_dev")
    return dst, out_len.value


def deflate_async_dev(src, dst, result, level=6, length=None, offset=0, stream=None, dict_len=0, flags=0):
    """zng_rocm_deflate_async_dev: as deflate_dev, but nothing is synchronised -- `dst` (uint8 CUDA tensor of at least
    deflate_bound(n) bytes) and `result` (int64 CUDA tensor, 2 elements: {compressed size, does-not-fit flag}) are filled
    when `stream` gets there."""
    rocm._need_init()
    n = src.numel() - offset if length is None else length
    rocm._check(rocm.lib().zng_rocm_deflate_async_dev(level, rocm._dev_ptr(src, offset), n, dict_len, flags, rocm._dev_ptr(dst),
                                                      dst.numel(), rocm._dev_ptr(result), rocm._stream_ptr(stream)),
                "zng_rocm_deflate_async_dev")
