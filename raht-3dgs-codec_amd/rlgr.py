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


_STAGING = {}


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
