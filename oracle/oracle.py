"""ctypes binding of the CPU oracle (oracle/raht_oracle.c). TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package never does.  Parity status: PINNED by ``tests/golden/*.npz``
(generated from the reference's own Python by ``tests/golden/gen_golden.py``).

The functions mirror the reference operators one-to-one (numpy in, numpy out):

    raht_param(V, minV, width, depth) -> (List, Flags, weights, order)   # RAHT_param.py:190-279
    raht_fwd(C, param)  -> (T, w)                                        # RAHT.py:252-336
    raht_inv(T, param)  -> C                                             # iRAHT.py:40-114
    morton(Vint, J)     -> uint64[N]                                     # voxelize_pc.py:25-59
    voxelize(PC, J, vmin=None, width=None) -> dict                       # voxelize_pc.py:62-172
    quant_reorder / dequant_unreorder                                    # encode_3dgs.py:204-268
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libraht_oracle.so")
CPU_TWINS_SO = os.path.join(_HERE, "_build", "libraht_cpu.so")     # raht_cpu_*: host twins of the product's C ABI


def build(force=False):
    """Compile the oracle with gcc (seconds). Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("raht_oracle.c", "raht_oracle.h", "raht_cpu.c", "raht_cpu.h")]
    outs = [_SO, CPU_TWINS_SO]
    if (not force and all(os.path.exists(o) for o in outs)
            and min(os.path.getmtime(o) for o in outs) >= max(os.path.getmtime(f) for f in srcs)):
        return _SO
    subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
    L.orc_morton.argtypes = [vp, i64, i32, vp]
    L.orc_param_build.argtypes = [vp, i64, vp, dbl, i32, i32, C.POINTER(vp)]
    L.orc_param_levels.argtypes = [vp]
    L.orc_param_level_len.argtypes = [vp, i32]
    L.orc_param_level_len.restype = i64
    for f in (L.orc_param_list, L.orc_param_flags, L.orc_param_weights):
        f.argtypes = [vp, i32]
        f.restype = vp
    L.orc_param_order_len.argtypes = [vp]
    L.orc_param_order_len.restype = i64
    L.orc_param_order.argtypes = [vp]
    L.orc_param_order.restype = vp
    L.orc_param_morton.argtypes = [vp]
    L.orc_param_morton.restype = vp
    L.orc_param_free.argtypes = [vp]
    L.orc_param_free.restype = None
    L.orc_raht_fwd.argtypes = [vp, i64, i32, vp, vp, vp]
    L.orc_raht_inv.argtypes = [vp, i64, i32, vp, vp]
    L.orc_quant_reorder.argtypes = [vp, i64, i32, dbl, vp, vp]
    L.orc_dequant_unreorder.argtypes = [vp, i64, i32, dbl, vp, vp]
    L.orc_voxelize.argtypes = [vp, i64, i32, vp, dbl, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.orc_voxel_residuals.argtypes = [vp, i64, i32, vp, vp, i64, vp, vp, dbl, vp, vp]
    L.orc_rlgr_encode.argtypes = [vp, i64, i32, vp, i64]
    L.orc_rlgr_encode.restype = i64
    L.orc_rlgr_decode.argtypes = [vp, i64, i64, i32, vp]
    L.orc_merge_clusters.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _view(addr, n, dtype):
    if n <= 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class Param:
    """Owns an ``orc_param`` (the reference's List/Flags/weights/order_RAGFT)."""

    def __init__(self, handle):
        self._h = handle
        L = lib()
        self.nlevels = L.orc_param_levels(handle)
        self.List, self.Flags, self.weights = [], [], []
        for l in range(self.nlevels):
            n = L.orc_param_level_len(handle, l)
            self.List.append(_view(L.orc_param_list(handle, l), n, np.int64))
            self.Flags.append(_view(L.orc_param_flags(handle, l), n, np.uint8).astype(bool))
            self.weights.append(_view(L.orc_param_weights(handle, l), n, np.int64))
        n = L.orc_param_order_len(handle)
        self.order = None if n < 0 else _view(L.orc_param_order(handle), n, np.int64)
        self.N = int(self.List[0].shape[0])
        self.morton = _view(L.orc_param_morton(handle), self.N, np.uint64)

    def __del__(self):
        try:
            if self._h:
                lib().orc_param_free(self._h)
                self._h = None
        except Exception:
            pass


def morton(Vint, J):
    Vint = np.ascontiguousarray(Vint, dtype=np.int64)
    out = np.empty(Vint.shape[0], dtype=np.uint64)
    rc = lib().orc_morton(_ptr(Vint), Vint.shape[0], int(J), _ptr(out))
    if rc:
        raise ValueError("orc_morton failed")
    return out


def raht_param(V, minV, width, depth, ref_quirks=True):
    V = np.ascontiguousarray(V, dtype=np.float64)
    minV = np.ascontiguousarray(minV, dtype=np.float64)
    h = C.c_void_p()
    rc = lib().orc_param_build(_ptr(V), V.shape[0], _ptr(minV), float(width), int(depth),
                               1 if ref_quirks else 0, C.byref(h))
    if rc:
        raise ValueError("orc_param_build failed")
    return Param(h)


def raht_fwd(Cmat, param):
    Cmat = np.ascontiguousarray(Cmat, dtype=np.float64)
    N, D = Cmat.shape
    T = np.empty_like(Cmat)
    w = np.empty(N, dtype=np.float64)
    rc = lib().orc_raht_fwd(_ptr(Cmat), N, D, param._h, _ptr(T), _ptr(w))
    if rc:
        raise ValueError("orc_raht_fwd failed")
    return T, w.reshape(N, 1)


def raht_inv(T, param):
    T = np.ascontiguousarray(T, dtype=np.float64)
    N, D = T.shape
    out = np.empty_like(T)
    rc = lib().orc_raht_inv(_ptr(T), N, D, param._h, _ptr(out))
    if rc:
        raise ValueError("orc_raht_inv failed")
    return out


def quant_reorder(T, step, order):
    T = np.ascontiguousarray(T, dtype=np.float64)
    order = np.ascontiguousarray(order, dtype=np.int64)
    N, D = T.shape
    Q = np.empty((N, D), dtype=np.int32)
    lib().orc_quant_reorder(_ptr(T), N, D, float(step), _ptr(order), _ptr(Q))
    return Q


def dequant_unreorder(Q, step, order):
    Q = np.ascontiguousarray(Q, dtype=np.int32)
    order = np.ascontiguousarray(order, dtype=np.int64)
    N, D = Q.shape
    T = np.empty((N, D), dtype=np.float64)
    lib().orc_dequant_unreorder(_ptr(Q), N, D, float(step), _ptr(order), _ptr(T))
    return T


def voxelize(PC, J, vmin=None, width=None):
    PC = np.ascontiguousarray(PC, dtype=np.float32)
    N, ld = PC.shape
    d = ld - 3
    keys = np.empty(N, dtype=np.uint64)
    idx = np.empty(N, dtype=np.int64)
    vi = np.empty(N, dtype=np.int64)
    pcv = np.empty((N, ld), dtype=np.float32)
    vvox = np.empty((N, 3), dtype=np.int64)
    nvox = C.c_int64()
    vmin_out = np.empty(3, dtype=np.float32)
    w_out, vs_out = C.c_double(), C.c_double()
    vmin_a = None if vmin is None else np.ascontiguousarray(vmin, dtype=np.float32)
    rc = lib().orc_voxelize(_ptr(PC), N, d, None if vmin_a is None else _ptr(vmin_a),
                            -1.0 if width is None else float(width), int(J), _ptr(keys), _ptr(idx),
                            _ptr(vi), _ptr(pcv), _ptr(vvox), C.byref(nvox), _ptr(vmin_out),
                            C.byref(w_out), C.byref(vs_out))
    if rc:
        raise ValueError("orc_voxelize failed")
    n = nvox.value
    return dict(keys_sorted=keys, sort_idx=idx, voxel_indices=vi[:n].copy(), PCvox=pcv[:n].copy(),
                Vvox=vvox[:n].copy(), Nvox=n, vmin=vmin_out, width=w_out.value,
                voxel_size=vs_out.value)


def voxel_residuals(PC, r):
    """(PCsorted, DeltaPC) of voxelize_pc_batched (voxelize_pc.py:103-111, 147-156) from voxelize()'s result dict r."""
    PC = np.ascontiguousarray(PC, dtype=np.float32)
    N, ld = PC.shape
    pcs, dl = np.empty((N, ld), np.float32), np.empty((N, ld), np.float32)
    si = np.ascontiguousarray(r["sort_idx"], dtype=np.int64)
    vi = np.ascontiguousarray(r["voxel_indices"], dtype=np.int64)
    pcv = np.ascontiguousarray(r["PCvox"], dtype=np.float32)
    vm = np.ascontiguousarray(r["vmin"], dtype=np.float32)
    lib().orc_voxel_residuals(_ptr(PC), N, ld - 3, _ptr(si), _ptr(vi), vi.shape[0], _ptr(pcv), _ptr(vm), float(r["voxel_size"]),
                              _ptr(pcs), _ptr(dl))
    return pcs, dl


def rlgr_encode(seq, flag_signed=1):
    """bytes of membuf().rlgrWrite(seq, flag); close(); get_buffer()  (PyRLGR membuf.cpp:340-423)."""
    seq = np.ascontiguousarray(seq, dtype=np.int64)
    cap = 16 + 9 * seq.shape[0]
    out = np.empty(cap, dtype=np.uint8)
    n = lib().orc_rlgr_encode(_ptr(seq), seq.shape[0], int(flag_signed), _ptr(out), cap)
    if n < 0:
        raise ValueError("orc_rlgr_encode: buffer too small")
    return out[:n].copy()


def rlgr_decode(buf, N, flag_signed=1):
    """membuf(buf).rlgrRead(N, flag)  (PyRLGR membuf.cpp:258-338)."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    seq = np.empty(N, dtype=np.int64)
    lib().orc_rlgr_decode(_ptr(buf), buf.shape[0], int(N), int(flag_signed), _ptr(seq))
    return seq


def merge_clusters(cluster_indices, cluster_offsets, means, quats, scales, opacities, colors, weight_by_opacity=True):
    """merge_weighted_mean_kernel restated from cuda/merge_cluster.cu:2-111 (PARITY UNPINNED)."""
    ci = np.ascontiguousarray(cluster_indices, dtype=np.int32)
    co = np.ascontiguousarray(cluster_offsets, dtype=np.int32)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)   # noqa: E731
    means, quats, scales, opacities, colors = f(means), f(quats), f(scales), f(opacities), f(colors)
    K, cd = co.shape[0] - 1, colors.shape[1]
    out = [np.zeros((K, 3), np.float32), np.zeros((K, 4), np.float32), np.zeros((K, 3), np.float32),
           np.zeros((K,), np.float32), np.zeros((K, cd), np.float32)]
    lib().orc_merge_clusters(_ptr(ci), _ptr(co), K, _ptr(means), _ptr(quats), _ptr(scales), _ptr(opacities), _ptr(colors),
                             cd, 1 if weight_by_opacity else 0, *[_ptr(o) for o in out])
    return tuple(out)
