"""RLGR entropy stage: drop-in for the reference's ``rlgr`` pybind module (vendored PyRLGR,
reference python/PyRLGR/src/libs/rlgr/bindings.cpp:34-57; call sites python/encode_3dgs.py:229-245),
backed by the byte-exact host coder in libraht_hip.so (csrc/rlgr.hip).

    m = membuf(); ns = m.rlgrWrite(seq, 1); m.close(); buf = m.get_buffer()
    ns, out = membuf(buf).rlgrRead(N, 1)

``seq`` may be a Python list (as in the reference) or -- much faster -- a numpy int32/int64 array.
``encode_channels`` / ``decode_channels`` code all columns of an (N, D) int32 matrix at once on host
threads, without the per-channel tensor -> list -> vector copies of the reference driver.
"""
import ctypes as C
import time

import numpy as np

_STAGING = {}

from . import _lib
from ._lib import check


def _as_i32(seq):
    a = np.asarray(seq)
    if a.dtype != np.int32:
        if a.size and (a.min() < -2 ** 31 or a.max() > 2 ** 32 - 1):
            raise OverflowError("RLGR symbols must fit 32 bits")
        a = a.astype(np.int64).astype(np.uint32).view(np.int32) if a.size and a.max() > 2 ** 31 - 1 else a.astype(np.int32)
    return np.ascontiguousarray(a)


class membuf:
    """Same methods as the reference's ``rlgr.membuf``."""

    def __init__(self, in_buf=None):
        self._write = in_buf is None
        self._buf = np.zeros(0, np.uint8) if in_buf is None else np.ascontiguousarray(np.asarray(in_buf, dtype=np.uint8))

    def rlgrWrite(self, seq, flagSigned=1):
        a = _as_i32(seq)
        cap = int(_lib.lib().raht_rlgr_bound(a.shape[0]))
        out = np.empty(cap, np.uint8)
        n = C.c_int64()
        t0 = time.perf_counter_ns()
        check(_lib.lib().raht_rlgr_encode(a.ctypes.data_as(C.c_void_p), a.shape[0], 1, int(flagSigned),
                                          out.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        ns = time.perf_counter_ns() - t0
        self._buf = out[: n.value].copy()
        return ns

    def rlgrRead(self, N, flagSigned=1):
        out = np.empty(int(N), np.int32)
        t0 = time.perf_counter_ns()
        check(_lib.lib().raht_rlgr_decode(self._buf.ctypes.data_as(C.c_void_p), self._buf.shape[0], int(N),
                                          int(flagSigned), out.ctypes.data_as(C.c_void_p), 1))
        ns = time.perf_counter_ns() - t0
        if not flagSigned:
            return ns, out.view(np.uint32).astype(np.int64).tolist()
        return ns, out.tolist()

    def close(self):
        pass                                   # the stream is padded to a byte boundary by rlgrWrite

    def get_buffer(self):
        return self._buf.tolist()              # the reference returns a Python list of ints

    def get_array(self):
        return self._buf                       # numpy view (no copy) for callers that can use it

    def buffer_size(self):
        return int(self._buf.shape[0])


def encode_channels(Q, flag_signed=1, nthreads=0, channel_major=False):
    """Q: (N, D) int32 numpy array, or (D, N) when channel_major (what ``transpose_on_device`` yields)
    -> (list of D uint8 arrays, seconds)."""
    Q = np.asarray(Q)
    if Q.dtype != np.int32 or Q.ndim != 2 or Q.strides[1] != 4:
        Q = np.ascontiguousarray(Q, dtype=np.int32)
    if channel_major:
        D, N = Q.shape
        ss, cs = 1, Q.strides[0] // 4
    else:
        N, D = Q.shape
        ss, cs = Q.strides[0] // 4, 1
    cap = int(_lib.lib().raht_rlgr_bound(N))
    out = np.empty((D, cap), np.uint8)
    nb = np.empty(D, np.int64)
    t0 = time.perf_counter()
    check(_lib.lib().raht_rlgr_encode_channels(Q.ctypes.data_as(C.c_void_p), N, D, ss, cs, int(flag_signed),
                                               out.ctypes.data_as(C.c_void_p), cap, nb.ctypes.data_as(C.c_void_p),
                                               int(nthreads)))
    dt = time.perf_counter() - t0
    return [out[c, : nb[c]].copy() for c in range(D)], dt


def decode_channels(streams, N, flag_signed=1, nthreads=0, channel_major=False):
    """streams: list of D uint8 arrays -> ((N, D) int32 array, or (D, N) when channel_major; seconds)."""
    D = len(streams)
    cap = max(1, max(int(np.asarray(s).shape[0]) for s in streams))
    bufs = np.zeros((D, cap), np.uint8)
    nb = np.empty(D, np.int64)
    for c, s in enumerate(streams):
        s = np.asarray(s, dtype=np.uint8)
        bufs[c, : s.shape[0]] = s
        nb[c] = s.shape[0]
    Q = np.empty((D, int(N)) if channel_major else (int(N), D), np.int32)
    ss, cs = (1, int(N)) if channel_major else (D, 1)
    t0 = time.perf_counter()
    check(_lib.lib().raht_rlgr_decode_channels(bufs.ctypes.data_as(C.c_void_p), cap, nb.ctypes.data_as(C.c_void_p), int(N), D,
                                               int(flag_signed), Q.ctypes.data_as(C.c_void_p), ss, cs, int(nthreads)))
    return Q, time.perf_counter() - t0


class ChannelCoder:
    """All channels of a frame through the host coder WITHOUT per-call glue: one arena for the D streams, kept across
    calls (worst case 13 bytes per symbol: virtual memory -- only what a stream really takes is ever touched), streams
    handed out as views, decoding straight from the arena into a caller-supplied (e.g. page-locked) array, and a threaded
    round-trip check. On the MI355X box the coder itself takes ~25 ms per direction for 3 M x 56 symbols; the per-call
    numpy allocations / copies / array_equal of encode_channels + decode_channels were another 110 ms."""

    def __init__(self, N, D, flag_signed=1, nthreads=0):
        self.N, self.D, self.flag, self.nthreads = int(N), int(D), int(flag_signed), int(nthreads)
        self.cap = int(_lib.lib().raht_rlgr_bound(self.N))
        self.arena = np.empty((self.D, self.cap), np.uint8)
        self.nb = np.zeros(self.D, np.int64)

    def encode(self, Qcm):
        """Qcm: (D, N) int32, channel-major contiguous -> seconds; streams in self.arena / self.nb"""
        if Qcm.dtype != np.int32 or Qcm.shape != (self.D, self.N) or not Qcm.flags.c_contiguous:
            raise ValueError("ChannelCoder.encode: expected a contiguous (D, N) int32 array")
        t0 = time.perf_counter()
        check(_lib.lib().raht_rlgr_encode_channels(Qcm.ctypes.data_as(C.c_void_p), self.N, self.D, 1, self.N, self.flag,
                                                   self.arena.ctypes.data_as(C.c_void_p), self.cap, self.nb.ctypes.data_as(C.c_void_p), self.nthreads))
        return time.perf_counter() - t0

    @property
    def size_bytes(self):
        return int(self.nb.sum())

    def streams(self):
        return [self.arena[c, : self.nb[c]] for c in range(self.D)]

    def decode(self, out):
        """-> seconds; out: (D, N) int32 contiguous (filled)"""
        if out.dtype != np.int32 or out.shape != (self.D, self.N) or not out.flags.c_contiguous:
            raise ValueError("ChannelCoder.decode: expected a contiguous (D, N) int32 array")
        t0 = time.perf_counter()
        check(_lib.lib().raht_rlgr_decode_channels(self.arena.ctypes.data_as(C.c_void_p), self.cap, self.nb.ctypes.data_as(C.c_void_p), self.N, self.D,
                                                   self.flag, out.ctypes.data_as(C.c_void_p), 1, self.N, self.nthreads))
        return time.perf_counter() - t0


def arrays_equal(a, b, nthreads=0):
    """threaded equality of two contiguous int32 arrays (the drivers' round-trip assertion, encode_3dgs.py:242-245)"""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or a.dtype != np.int32 or b.dtype != np.int32 or not a.flags.c_contiguous or not b.flags.c_contiguous:
        return bool(np.array_equal(a, b))
    first = C.c_int64()
    check(_lib.lib().raht_i32_equal(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), a.size, int(nthreads), C.byref(first)))
    return first.value < 0


def pinned_like(shape, dtype, key):
    """a cached page-locked host tensor of at least this size (the codec's staging buffers), viewed as `shape`"""
    import torch
    n = int(np.prod(shape))
    buf = _STAGING.get(key)
    if buf is None or buf.numel() < n or buf.dtype != dtype:
        buf = torch.empty(max(n, 1), dtype=dtype, pin_memory=True)
        _STAGING[key] = buf
    return buf[:n].view(shape)


class SegmentedCoder:
    """The RLGR stage on the GPU (include/raht.h: raht_rlgr_seg_*): every channel in segments of ``seg_len`` symbols, every
    segment an independent RLGR stream -- byte-identical to the reference coder's output for that slice -- one lane per
    segment. ``encode`` takes the channel-major (D, N) int32 device tensor (``transpose_on_device`` of the quantized
    coefficients) and leaves the streams on the device; ``container()`` is what goes on the wire; ``decode`` rebuilds the
    (D, N) tensor on the device. Nothing but the compressed bytes ever crosses PCIe."""
    MAGIC = b"RLGS0001"

    def __init__(self, N, D, seg_len=2048, flag_signed=1, device="cuda", payload_cap=None):
        import torch
        self.N, self.D, self.S, self.flag = int(N), int(D), int(seg_len), int(flag_signed)
        self.nseg = (self.N + self.S - 1) // self.S
        self.G = self.nseg * self.D
        self.device = torch.device(device)
        self.seg_bytes = torch.empty(self.G, dtype=torch.int32, device=self.device)
        self.seg_off = torch.empty(self.G + 1, dtype=torch.int32, device=self.device)
        # encoder: what a raw dump would take, + slack (grown on demand); decoder (from_container): the payload it was handed
        self.cap = 4 * self.N * self.D + 64 * self.G if payload_cap is None else max(16, int(payload_cap))
        self.out = torch.empty(self.cap, dtype=torch.uint8, device=self.device)
        self.bad = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.total = 0

    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _strides(self, Q, what):
        """(sym_stride, chan_stride) of a channel-major (D, N) or a row-major (N, D) int32 CUDA tensor"""
        import torch
        if not Q.is_cuda or Q.dtype != torch.int32 or Q.dim() != 2 or Q.stride(1) != 1:
            raise ValueError(f"SegmentedCoder.{what}: expected a 2-D int32 CUDA tensor with unit column stride")
        if tuple(Q.shape) == (self.D, self.N) and (self.D != self.N or Q.stride(0) >= self.N):
            return 1, Q.stride(0)
        if tuple(Q.shape) == (self.N, self.D):
            return Q.stride(0), 1
        raise ValueError(f"SegmentedCoder.{what}: expected a (D, N) channel-major or an (N, D) row-major tensor")

    def encode(self, Q):
        """Q: (D, N) channel-major OR (N, D) row-major int32 CUDA tensor (the latter: the quantized coefficients as forward_quant
        returns them, no transpose) -> total container payload bytes (synchronises: it returns a size). Same container either way."""
        import torch
        sym, chan = self._strides(Q, "encode")
        Qcm = Q
        tot = C.c_int64()
        for attempt in (0, 1):
            with torch.cuda.device(self.device):
                rc = _lib.lib().raht_rlgr_seg_encode_strided(C.c_void_p(Qcm.data_ptr()), self.N, self.D, sym, chan, self.S, self.flag,
                                                             C.c_void_p(self.seg_bytes.data_ptr()), C.c_void_p(self.seg_off.data_ptr()),
                                                             C.c_void_p(self.out.data_ptr()), self.cap, C.byref(tot), self._stream())
            if rc == _lib.RAHT_OK or attempt == 1 or tot.value <= self.cap:
                check(rc)
                break
            self.cap = int(tot.value) + 64                    # incompressible data: the exact size is known now
            self.out = torch.empty(self.cap, dtype=torch.uint8, device=self.device)
        self.total = int(tot.value)
        return self.total

    @property
    def size_bytes(self):
        """bytes of the container: header + length table + streams"""
        return len(self.MAGIC) + 5 * 8 + 4 * self.G + self.total

    def container_parts(self):
        """-> (header bytes, uint32 lengths (G,), payload uint8 (total,)): the container without joining it. The payload comes down
        through the cached page-locked staging buffer (57 GB/s) and aliases it until the next ``to_host`` of a uint8 tensor."""
        hdr = self.MAGIC + np.array([self.N, self.D, self.S, self.flag, self.total], np.int64).tobytes()
        lens = to_host(self.seg_bytes).view(np.uint32).copy()
        return hdr, lens, to_host(self.out[: self.total])

    def container(self):
        """-> bytes: magic | N, D, seg_len, flag, payload bytes (int64 each) | uint32 length of every segment | the streams, each in
        a 4-byte slot. Only the compressed bytes come down from the device."""
        hdr, lens, payload = self.container_parts()
        return hdr + lens.tobytes() + payload.tobytes()

    @classmethod
    def from_container(cls, blob, device="cuda", max_symbols=None):
        """max_symbols: refuse containers whose header announces more than this many symbols (N x D): decode() allocates
        4 N D bytes for them, and the header comes off the wire."""
        import torch
        m = len(cls.MAGIC)
        if blob[:m] != cls.MAGIC:
            raise ValueError("not a segmented RLGR container")
        if len(blob) < m + 40:
            raise ValueError("segmented RLGR container: truncated header")
        N, D, S, flag, total = [int(x) for x in np.frombuffer(blob, np.int64, 5, m)]
        # the header comes off the wire: sizes are checked against the blob BEFORE anything is allocated from them
        if not (1 <= N < 2 ** 31 and 1 <= D <= 65536 and 64 <= S < 2 ** 31 and flag in (0, 1) and 0 <= total <= len(blob)):
            raise ValueError("segmented RLGR container: implausible header")
        G = ((N + S - 1) // S) * D
        if len(blob) < m + 40 + 4 * G + total:
            raise ValueError("segmented RLGR container: shorter than its header says")
        if max_symbols is not None and N * D > int(max_symbols):
            raise ValueError(f"segmented RLGR container: {N} x {D} symbols, more than the caller allows ({max_symbols})")
        sc = cls(N, D, S, flag, device, payload_cap=total)     # (the payload only: a decoder never needs the encoder's raw-size buffer)
        lens = np.frombuffer(blob, np.uint32, sc.G, m + 40).astype(np.int64)
        if total % 4 or total > len(blob) - (m + 40 + 4 * sc.G) or int(((lens + 3) // 4 * 4).sum()) != total:
            raise ValueError("segmented RLGR container: inconsistent length table")
        off = np.concatenate([[0], np.cumsum((lens + 3) // 4 * 4)])
        sc.seg_bytes.copy_(torch.from_numpy(lens.astype(np.int32)))
        sc.seg_off.copy_(torch.from_numpy(off.astype(np.int64).astype(np.int32)))
        if total > sc.cap:
            sc.cap, sc.out = total, torch.empty(total, dtype=torch.uint8, device=sc.device)
        sc.out[:total].copy_(torch.from_numpy(np.frombuffer(blob, np.uint8, total, m + 40 + 4 * sc.G).copy()))
        sc.total = total
        return sc

    def decode(self, out=None, row_major=False):
        """-> (D, N) int32 CUDA tensor, or (N, D) with ``row_major=True`` (what dequant_inverse takes: no transpose behind the
        decoder); enqueued on the current stream, no synchronisation"""
        import torch
        if out is None:
            out = torch.empty((self.N, self.D) if row_major else (self.D, self.N), dtype=torch.int32, device=self.device)
        sym, chan = self._strides(out, "decode")
        with torch.cuda.device(self.device):
            check(_lib.lib().raht_rlgr_seg_decode_strided(C.c_void_p(self.out.data_ptr()), (self.total + 3) // 4 * 4, C.c_void_p(self.seg_off.data_ptr()),
                                                          C.c_void_p(self.seg_bytes.data_ptr()), self.N, self.D, self.S, self.flag,
                                                          C.c_void_p(out.data_ptr()), sym, chan, C.c_void_p(self.bad.data_ptr()), self._stream()))
        return out

    BATCH_MAX = 12                                            # RAHT_RLGR_BATCH_MAX

    @staticmethod
    def _same_shape(coders, what):
        c0 = coders[0]
        for c in coders[1:]:
            if (c.N, c.D, c.S, c.flag, c.device) != (c0.N, c0.D, c0.S, c0.flag, c0.device):
                raise ValueError(f"SegmentedCoder.{what}: the coders of a batch share N, D, seg_len, flag and device")
        return c0

    @classmethod
    def encode_batch(cls, coders, Qs):
        """The quantization steps of a frame coded together (raht_rlgr_seg_encode_batch): ``coders[j].encode(Qs[j])`` for every j,
        in ONE set of launches -- k times the independent streams of one frame, which is what the coder's speed depends on. Same
        containers, byte for byte. All Qs in the same layout ((N, D) row-major or (D, N) channel-major) with the same strides.
        -> list of container payload sizes (synchronises)."""
        import torch
        if len(coders) != len(Qs) or not coders:
            raise ValueError("SegmentedCoder.encode_batch: one input per coder")
        c0 = cls._same_shape(coders, "encode_batch")
        st = [c._strides(Q, "encode_batch") for c, Q in zip(coders, Qs)]
        if any(x != st[0] for x in st):
            raise ValueError("SegmentedCoder.encode_batch: the inputs of a batch share their layout and strides")
        sym, chan = st[0]
        L = _lib.lib()
        for lo in range(0, len(coders), cls.BATCH_MAX):
            cs, qs = coders[lo: lo + cls.BATCH_MAX], Qs[lo: lo + cls.BATCH_MAX]
            k = len(cs)
            VP, I64 = C.c_void_p * k, C.c_int64 * k
            tot = I64()
            for attempt in (0, 1):
                with torch.cuda.device(c0.device):
                    rc = L.raht_rlgr_seg_encode_batch(k, VP(*[q.data_ptr() for q in qs]), c0.N, c0.D, sym, chan, c0.S, c0.flag,
                                                      VP(*[c.seg_bytes.data_ptr() for c in cs]), VP(*[c.seg_off.data_ptr() for c in cs]),
                                                      VP(*[c.out.data_ptr() for c in cs]), I64(*[c.cap for c in cs]), tot, c0._stream())
                if rc == _lib.RAHT_OK or attempt == 1 or all(int(tot[j]) <= cs[j].cap for j in range(k)):
                    check(rc)
                    break
                for j, c in enumerate(cs):                    # incompressible data: the exact sizes are known now
                    if int(tot[j]) > c.cap:
                        c.cap = int(tot[j]) + 64
                        c.out = torch.empty(c.cap, dtype=torch.uint8, device=c.device)
            for j, c in enumerate(cs):
                c.total = int(tot[j])
        return [c.total for c in coders]

    @classmethod
    def decode_batch(cls, coders, outs=None, row_major=False, expect=None):
        """``coders[j].decode()`` for every j in ONE launch (raht_rlgr_seg_decode_batch) -> list of (D, N) int32 CUDA tensors ((N, D)
        with ``row_major=True``); enqueued on the current stream, no synchronisation. ``coders[0].bad`` collects the frames whose
        tables reached outside their payload (bit j of a chunk of 12).
        ``expect`` (row-major only): one (N, D) tensor per coder, what the frame should decode to -- compared inside the decoder
        (raht_rlgr_seg_decode_batch_check: the drivers' round-trip assertion without a pass of its own); bit 16 + j of
        ``coders[0].bad`` reports frame j of a chunk; ``roundtrip_failed(coders)`` reads it."""
        import torch
        if not coders:
            return []
        c0 = cls._same_shape(coders, "decode_batch")
        if outs is None:
            outs = [torch.empty((c0.N, c0.D) if row_major else (c0.D, c0.N), dtype=torch.int32, device=c0.device) for _ in coders]
        st = [c._strides(o, "decode_batch") for c, o in zip(coders, outs)]
        if len(outs) != len(coders) or any(x != st[0] for x in st):
            raise ValueError("SegmentedCoder.decode_batch: one output per coder, all in the same layout")
        sym, chan = st[0]
        if expect is not None:
            if not row_major and chan != 1:
                raise ValueError("SegmentedCoder.decode_batch: expect= goes with row-major frames")
            if len(expect) != len(coders) or any(c._strides(e, "decode_batch") != (sym, chan) for c, e in zip(coders, expect)):
                raise ValueError("SegmentedCoder.decode_batch: one expected frame per coder, in the layout and strides of the outputs")
        L = _lib.lib()
        for lo in range(0, len(coders), cls.BATCH_MAX):
            cs, os_ = coders[lo: lo + cls.BATCH_MAX], outs[lo: lo + cls.BATCH_MAX]
            k = len(cs)
            VP, I64 = C.c_void_p * k, C.c_int64 * k
            args = (k, VP(*[c.out.data_ptr() for c in cs]), I64(*[(c.total + 3) // 4 * 4 for c in cs]),
                    VP(*[c.seg_off.data_ptr() for c in cs]), VP(*[c.seg_bytes.data_ptr() for c in cs]),
                    c0.N, c0.D, c0.S, c0.flag, VP(*[o.data_ptr() for o in os_]))
            bad = cs[0].bad                                       # (every chunk reports into its own first coder)
            with torch.cuda.device(c0.device):
                if expect is None:
                    check(L.raht_rlgr_seg_decode_batch(*args, sym, chan, C.c_void_p(bad.data_ptr()), c0._stream()))
                else:
                    ex = expect[lo: lo + cls.BATCH_MAX]
                    check(L.raht_rlgr_seg_decode_batch_check(*args, VP(*[e.data_ptr() for e in ex]), sym, chan, C.c_void_p(bad.data_ptr()), c0._stream()))
        return outs

    @classmethod
    def roundtrip_failed(cls, coders):
        """after ``decode_batch(..., expect=...)``: -> list of the indices of the frames that did not decode to what was expected, or
        whose tables reached outside their payload (synchronises)"""
        out = []
        for lo in range(0, len(coders), cls.BATCH_MAX):
            w = int(coders[lo].bad.item()) & 0xffffffff
            out += [lo + j for j in range(min(cls.BATCH_MAX, len(coders) - lo)) if (w >> j) & 1 or (w >> (16 + j)) & 1]
        return out

    def segment(self, c, s):
        """the bytes of segment s of channel c (host copy; tests)"""
        g = c * self.nseg + s
        off, nb = int(self.seg_off[g].item()), int(self.seg_bytes[g].item())
        return self.out[off: off + nb].cpu().numpy()


def transpose_on_device(Q):
    """(rows, cols) int32 CUDA tensor -> (cols, rows) contiguous, with the HIP LDS-tile transpose
    (row-major N x D quantized coefficients -> channel-major D x N for the entropy stage, or back)."""
    import torch
    if not Q.is_cuda or Q.dtype != torch.int32 or Q.dim() != 2 or Q.stride(1) != 1:
        raise ValueError("expected a 2-D int32 CUDA tensor with unit column stride")
    rows, cols = Q.shape
    out = torch.empty((cols, rows), dtype=torch.int32, device=Q.device)
    with torch.cuda.device(Q.device):
        check(_lib.lib().raht_transpose_i32(C.c_void_p(Q.data_ptr()), Q.stride(0), rows, cols, C.c_void_p(out.data_ptr()),
                                            rows, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def to_host(t):
    """Device tensor -> numpy array through a cached page-locked staging buffer: 57 GB/s instead of the
    6.7 GB/s of a pageable ``.cpu()`` on the MI355X box (708 MB of quantized coefficients: 12 ms
    instead of 106 ms; tools/probe_d2h.py). The array aliases the staging buffer of its dtype and stays
    valid until the next ``to_host`` call with that dtype."""
    import torch
    if not t.is_cuda:
        return t.contiguous().numpy()
    t = t.contiguous()
    n = t.numel()
    buf = _STAGING.get(t.dtype)
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, 1), dtype=t.dtype, pin_memory=True)
        _STAGING[t.dtype] = buf
    out = buf[:n].view(t.shape)
    out.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return out.numpy()
