"""Host-side mirror of the reference's operator interface for the RAHT hot path.

Same names, argument meaning and return shapes as the reference's L3 operators, so that the
three-entry dispatch table of its drivers (reference python/encode_3dgs.py:23-27) can be replaced by

    from raht_3dgs_codec_amd import raht_fn      # {"RAHT", "iRAHT", "RAHT_param"}

Everything below runs on the MI355X through the C ABI of include/raht.h (libraht_hip.so); torch
is only plumbing (device memory, current stream). There is no CPU path: CPU tensors raise.

Differences from the reference, all deliberate:
  * ``RAHT_param_reorder_fast`` returns one opaque plan token per list instead of 3J tensors per
    list; the tokens are real device tensors, so the drivers' ``[t.to(device) for t in ListC]``
    keeps working, and they are handed back opaquely to RAHT / iRAHT exactly as before.
    ``plan_of(ListC).export_lists()`` materialises the reference-shaped List/Flags/weights.
  * unsorted / duplicate / out-of-range voxels raise ``RahtError`` (the reference silently
    mis-pairs them); N == 1 returns order_RAGFT = [0] (the reference returns None).
  * compute dtype follows the input: float32 in -> float32 kernels (the fast path), float64 in ->
    float64 kernels (the reference's own precision). The reference always returns float64.
"""
import collections
import ctypes as C

import torch

from . import _lib
from ._lib import RahtError, check  # noqa: F401

_MAGIC = 0x52414854_504C414E  # "RAHTPLAN"
_plans = collections.OrderedDict()   # id -> RahtPlan (strong refs to the most recent plans)
_PLAN_CACHE = 16
_next_id = [1]


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"raht-3dgs-codec_amd: {name} must be a CUDA (HIP) tensor; this package has "
                           "no CPU path for the RAHT hot path")


_VDT = {torch.float32: _lib.RAHT_F32, torch.float64: _lib.RAHT_F64, torch.int32: _lib.RAHT_I32,
        torch.int64: _lib.RAHT_I64}


class RahtPlan:
    """Owns a ``raht_plan*`` (device-resident lvl / wl / wr / order_RAGFT and tile schedules)."""

    def __init__(self, handle, device):
        self._h = handle
        self.device = device
        L = _lib.lib()
        self.N = int(L.raht_plan_size(handle))
        self.nbits = int(L.raht_plan_nbits(handle))
        self.id = _next_id[0]
        _next_id[0] += 1
        self._order = None
        self._inv_order = None
        self._roots = None
        self.map_rows = None             # rows of the matrices of a row-mapped plan (set_row_map)

    @property
    def inv_order(self):
        """Inverse permutation of order_RAGFT: position of row i in the reordered coefficient list."""
        if self._inv_order is None:
            o = self.order_RAGFT
            inv = torch.empty_like(o)
            inv[o] = torch.arange(self.N, dtype=torch.int64, device=o.device)
            self._inv_order = inv
        return self._inv_order

    # -- construction ---------------------------------------------------------------------------
    @staticmethod
    def from_coords(V, minV, width, depth):
        _need_cuda(V, "V")
        if V.dim() != 2 or V.shape[1] != 3:
            raise ValueError("V must be (N, 3)")
        if V.dtype not in _VDT:
            V = V.to(torch.float64)
        V = V.contiguous()
        if isinstance(minV, torch.Tensor):
            mv = [float(x) for x in minV.detach().cpu().reshape(-1).tolist()]
        else:
            mv = [float(x) for x in minV]
        if len(mv) != 3:
            raise ValueError("minV must have 3 entries")
        arr = (C.c_double * 3)(*mv)
        h = C.c_void_p()
        with torch.cuda.device(V.device):
            check(_lib.lib().raht_plan_create(C.c_void_p(V.data_ptr()), _VDT[V.dtype], V.shape[0], arr,
                                              float(width), int(depth), _stream(), C.byref(h)))
        return RahtPlan(h, V.device)

    @staticmethod
    def from_keys(keys_sorted, nbits, leaf_weights=None, top_level=None, borrow=False):
        """borrow=True: the plan reads `keys_sorted` in place instead of copying it (the plan keeps a reference to the
        tensor; the caller must not modify it while the plan lives)."""
        _need_cuda(keys_sorted, "keys_sorted")
        k = keys_sorted.contiguous()
        if k.dtype not in (torch.int64, torch.uint64):
            raise ValueError("keys must be int64/uint64")
        lw = None
        if leaf_weights is not None:
            _need_cuda(leaf_weights, "leaf_weights")
            lw = leaf_weights.to(torch.int64).contiguous()
        h = C.c_void_p()
        fn = _lib.lib().raht_plan_create_from_keys_borrowed if borrow else _lib.lib().raht_plan_create_from_keys
        with torch.cuda.device(k.device):
            check(fn(C.c_void_p(k.data_ptr()), k.shape[0], int(nbits),
                     C.c_void_p(lw.data_ptr()) if lw is not None else None, _stream(), C.byref(h)))
        p = RahtPlan(h, k.device)
        if borrow:
            p._keys_ref = k                   # keeps the borrowed array alive as long as the plan
        if top_level is not None:
            p.set_top_level(top_level)
        return p

    # -- truncated trees / roots (Morton-prefix sharded scenes) -----------------------------------
    def set_top_level(self, top_level):
        """Butterflies at binary levels >= top_level are left to the caller's top stage."""
        with torch.cuda.device(self.device):
            check(_lib.lib().raht_plan_set_top_level(self._h, int(top_level), _stream()))
        self._roots = None

    @property
    def root_rows(self):
        """int64 device tensor: rows that still carry a low-pass value after a forward transform."""
        if getattr(self, "_roots", None) is None:
            n = C.c_int64()
            check(_lib.lib().raht_plan_roots(self._h, C.byref(n), None, None))
            r = torch.empty(n.value, dtype=torch.int64, device=self.device)
            with torch.cuda.device(self.device):
                check(_lib.lib().raht_plan_roots(self._h, C.byref(n), C.c_void_p(r.data_ptr()), _stream()))
            self._roots = r
        return self._roots

    @property
    def n_roots(self):
        return int(self.root_rows.shape[0])

    def _set_roots_buffer(self, buf, D, dtype):
        if buf is None:
            check(_lib.lib().raht_plan_set_root_buffer(self._h, None))
            return
        _need_cuda(buf, "roots")
        if buf.dtype != dtype or tuple(buf.shape) != (self.n_roots, D) or not buf.is_contiguous():
            raise ValueError(f"roots buffer must be a contiguous ({self.n_roots}, {D}) {dtype} tensor")
        check(_lib.lib().raht_plan_set_root_buffer(self._h, C.c_void_p(buf.data_ptr())))

    def __del__(self):
        try:
            if self._h:
                _lib.lib().raht_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- views ----------------------------------------------------------------------------------
    @property
    def order_RAGFT(self):
        if self._order is None:
            o = torch.empty(self.N, dtype=torch.int64, device=self.device)
            with torch.cuda.device(self.device):
                check(_lib.lib().raht_plan_order(self._h, C.c_void_p(o.data_ptr()), _stream()))
            self._order = o
        return self._order

    @property
    def levels(self):
        return int(_lib.lib().raht_plan_levels(self._h))

    def set_engine(self, engine="tile", tile_rows=0, tail_rows=0, tail_channels=0, final_rows=0):
        e = {"tile": _lib.ENGINE_TILE, "level": _lib.ENGINE_LEVEL}[engine]
        check(_lib.lib().raht_plan_set_engine(self._h, e, int(tile_rows)))
        check(_lib.lib().raht_plan_set_tail_tile(self._h, int(tail_rows), int(tail_channels), int(final_rows)))

    def export_lists(self):
        """Reference-shaped (List, Flags, weights) as CPU tensors (RAHT_param.py:190-279 outputs)."""
        import numpy as np
        L = _lib.lib()
        List, Flags, weights = [], [], []
        for l in range(self.levels):
            n = C.c_int64()
            check(L.raht_plan_export_level(self._h, l, None, None, None, C.byref(n)))
            a = np.empty(n.value, np.int64)
            f = np.empty(n.value, np.uint8)
            w = np.empty(n.value, np.int64)
            check(L.raht_plan_export_level(self._h, l, a.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p),
                                           w.ctypes.data_as(C.c_void_p), C.byref(n)))
            List.append(torch.from_numpy(a))
            Flags.append(torch.from_numpy(f.astype(bool)))
            weights.append(torch.from_numpy(w))
        return List, Flags, weights

    def keys_tensor(self):
        """The plan's sorted Morton keys as an int64 device tensor (a copy)."""
        t = torch.empty(self.N, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            check(_lib.lib().raht_plan_copy_array(self._h, 0, C.c_void_p(t.data_ptr()), _stream()))
        return t

    def arrays(self):
        """(keys, lvl, wl, wr) copied to CPU numpy arrays (inspection / tests)."""
        import numpy as np
        out = []
        with torch.cuda.device(self.device):
            for which, dt in enumerate((torch.int64, torch.uint8, torch.int32, torch.int32)):
                t = torch.empty(self.N, dtype=dt, device=self.device)
                check(_lib.lib().raht_plan_copy_array(self._h, which, C.c_void_p(t.data_ptr()), _stream()))
                out.append(t.cpu().numpy())
        out[0] = out[0].view(np.uint64)
        return tuple(out)

    def stage_stats(self, elem_size=4, D=59):
        n = C.c_int()
        rows = (C.c_int64 * 32)()
        tr = C.c_int()
        check(_lib.lib().raht_plan_stage_stats(self._h, elem_size, D, C.byref(n), rows, 32, C.byref(tr)))
        k = abs(n.value)
        return dict(valid=n.value > 0, tile_rows=tr.value, rows_per_stage=[int(rows[i]) for i in range(k)])

    # -- transforms -----------------------------------------------------------------------------
    def _xform(self, X, inverse, want_w=False, roots=None, out=None):
        _need_cuda(X, "C" if not inverse else "T")
        rows = self.N if self.map_rows is None else self.map_rows
        if X.dim() != 2 or X.shape[0] != rows:
            raise ValueError(f"expected ({rows}, D) tensor, got {tuple(X.shape)}")
        if X.dtype not in (torch.float32, torch.float64):
            X = X.to(torch.float32)
        if X.stride(1) != 1 or X.stride(0) < X.shape[1]:
            X = X.contiguous()
        D = X.shape[1]
        if out is None:
            out = torch.empty((rows, D), dtype=X.dtype, device=X.device)
        elif out.dtype != X.dtype or tuple(out.shape) != (rows, D) or not out.is_contiguous() or out.device != X.device:
            raise ValueError(f"out must be a contiguous ({rows}, {D}) {X.dtype} tensor on {X.device}")
        w = torch.empty((self.N, 1), dtype=X.dtype, device=X.device) if want_w else None
        L = _lib.lib()
        f64 = X.dtype == torch.float64
        self._set_roots_buffer(roots, D, X.dtype)
        try:
            self._run(L, X, out, w, D, inverse, f64)
        finally:
            if roots is not None:
                self._set_roots_buffer(None, D, X.dtype)
        return (out, w) if want_w else out

    def _run(self, L, X, out, w, D, inverse, f64):
        with torch.cuda.device(X.device):
            if not inverse:
                fn = L.raht_fwd_f64 if f64 else L.raht_fwd
                check(fn(self._h, C.c_void_p(X.data_ptr()), X.stride(0), D, C.c_void_p(out.data_ptr()), D,
                         C.c_void_p(w.data_ptr()) if w is not None else None, _stream()))
            else:
                fn = L.raht_inv_f64 if f64 else L.raht_inv
                check(fn(self._h, C.c_void_p(X.data_ptr()), X.stride(0), D, C.c_void_p(out.data_ptr()), D,
                         _stream()))

    def forward(self, Cmat, want_w=True, roots=None, out=None):
        """roots: optional (n_roots, D) output buffer receiving the rows that still carry a low-pass;
        out: optional preallocated result matrix."""
        return self._xform(Cmat, False, want_w, roots, out)

    def inverse(self, T, roots=None, out=None):
        """roots: optional (n_roots, D) buffer the root rows are read from instead of T."""
        return self._xform(T, True, False, roots, out)

    def prepare(self, D, dtype=torch.float32):
        """Pre-build schedule + workspaces so later calls only enqueue kernels (hipGraph-safe)."""
        es = 8 if dtype == torch.float64 else 4
        with torch.cuda.device(self.device):
            check(_lib.lib().raht_plan_prepare(self._h, es, int(D), _stream()))

    def set_row_map(self, row_map, n_matrix_rows):
        """Plan row i lives in row ``row_map[i]`` of the (n_matrix_rows, D) matrices given to forward / inverse
        (small plans: the replicated top tree of a sharded scene works in place on the padded all-gather buffer)."""
        with torch.cuda.device(self.device):
            if row_map is None:
                check(_lib.lib().raht_plan_set_row_map(self._h, None, 0, _stream()))
                self.map_rows = None
                return
            m = row_map.to(device=self.device, dtype=torch.int64).contiguous()
            check(_lib.lib().raht_plan_set_row_map(self._h, C.c_void_p(m.data_ptr()), int(n_matrix_rows), _stream()))
        self.map_rows = int(n_matrix_rows)

    def set_concurrent_directions(self, on=True):
        """One workspace set per direction: a forward-direction and an inverse-direction call of this plan may then run at the same
        time on two streams (the drivers' loop over quantization steps: forward of step s + 1 next to the inverse of step s)."""
        with torch.cuda.device(self.device):
            check(_lib.lib().raht_plan_set_concurrent_directions(self._h, 1 if on else 0))

    def set_max_stages(self, max_stages):
        """Bound on the launches per direction of the tile schedule; above it the level engine runs."""
        with torch.cuda.device(self.device):
            check(_lib.lib().raht_plan_set_max_stages(self._h, int(max_stages)))

    def _f64_quant_call(self, fn, src, D, steps, dst):
        st = _steps64(steps, D)
        with torch.cuda.device(src.device):
            check(fn(self._h, C.c_void_p(src.data_ptr()), src.stride(0), D, st, len(st), C.c_void_p(dst.data_ptr()),
                     dst.stride(0), _stream()))
        return dst

    def forward_quant(self, Cmat, steps, roots=None):
        """Forward RAHT + quantize + reorder -> int32 Q. float32 (default): ONE fused pass, T is never
        materialised. float64 input: the same at the reference's precision (encode_3dgs.py:82-83,204): the float64 tile
        kernels with the float64 quantizer in their write-back."""
        _need_cuda(Cmat, "C")
        if Cmat.dtype == torch.float64:
            X = Cmat if (Cmat.stride(1) == 1 and Cmat.stride(0) >= Cmat.shape[1]) else Cmat.contiguous()
            Q = torch.empty((self.N, X.shape[1]), dtype=torch.int32, device=X.device)
            self._set_roots_buffer(roots, X.shape[1], torch.float64)
            try:
                return self._f64_quant_call(_lib.lib().raht_fwd_quant_f64, X, X.shape[1], steps, Q)
            finally:
                if roots is not None:
                    self._set_roots_buffer(None, X.shape[1], torch.float64)
        X = Cmat.to(torch.float32)
        if X.stride(1) != 1 or X.stride(0) < X.shape[1]:
            X = X.contiguous()
        D = X.shape[1]
        st = _steps(steps, D)
        Q = torch.empty((self.N, D), dtype=torch.int32, device=X.device)
        self._set_roots_buffer(roots, D, torch.float32)
        try:
            with torch.cuda.device(X.device):
                check(_lib.lib().raht_fwd_quant(self._h, C.c_void_p(X.data_ptr()), X.stride(0), D, st, len(st),
                                                C.c_void_p(Q.data_ptr()), D, _stream()))
        finally:
            if roots is not None:
                self._set_roots_buffer(None, D, torch.float32)
        return Q

    def dequant_inverse(self, Q, steps, roots=None, dtype=torch.float32):
        """Un-reorder + dequantize + inverse RAHT -> C in ONE fused pass (dtype=torch.float64: at the reference's
        precision)."""
        _need_cuda(Q, "Q")
        Q = Q.to(torch.int32).contiguous()
        D = Q.shape[1]
        if dtype == torch.float64:
            out = torch.empty((self.N, D), dtype=torch.float64, device=Q.device)
            self._set_roots_buffer(roots, D, torch.float64)
            try:
                return self._f64_quant_call(_lib.lib().raht_dequant_inv_f64, Q, D, steps, out)
            finally:
                if roots is not None:
                    self._set_roots_buffer(None, D, torch.float64)
        st = _steps(steps, D)
        out = torch.empty((self.N, D), dtype=torch.float32, device=Q.device)
        self._set_roots_buffer(roots, D, torch.float32)
        try:
            with torch.cuda.device(Q.device):
                check(_lib.lib().raht_dequant_inv(self._h, C.c_void_p(Q.data_ptr()), D, D, st, len(st),
                                                  C.c_void_p(out.data_ptr()), D, _stream()))
        finally:
            if roots is not None:
                self._set_roots_buffer(None, D, torch.float32)
        return out

    def forward_quant_multi(self, Cmat, steps):
        """ONE forward pass, one quantization per (scalar) step: -> [Q_0, ..., Q_{k-1}], each bit-identical to
        ``forward_quant(C, steps[i])`` (the drivers quantize one coefficient matrix at nine steps, python/encode_3dgs.py:28,199-217)."""
        _need_cuda(Cmat, "C")
        X = Cmat.to(torch.float32)
        if X.stride(1) != 1 or X.stride(0) < X.shape[1]:
            X = X.contiguous()
        D = X.shape[1]
        k = len(steps)
        st = (C.c_float * k)(*[float(s) for s in steps])
        Qs = [torch.empty((self.N, D), dtype=torch.int32, device=X.device) for _ in range(k)]
        ptrs = (C.c_void_p * k)(*[q.data_ptr() for q in Qs])
        with torch.cuda.device(X.device):
            check(_lib.lib().raht_fwd_quant_multi(self._h, C.c_void_p(X.data_ptr()), X.stride(0), D, st, k, ptrs, D, _stream()))
        return Qs

    def dequant_inverse_sqdiff(self, Q, steps, C_ref, want_rec=True):
        """Un-reorder + dequantize + inverse RAHT (float32, fused) that also compares its output with the original attributes on the
        way out: -> (C_rec or None, float64[D] per-column sums of (C_rec - C_ref)^2). What the drivers' five PSNR columns are made of
        (python/encode_3dgs.py:274,298-310) without a pass of its own; with ``want_rec=False`` C_rec is never written."""
        _need_cuda(Q, "Q")
        _need_cuda(C_ref, "C_ref")
        Q = Q.to(torch.int32).contiguous()
        D = Q.shape[1]
        X = C_ref.to(torch.float32)
        if X.stride(1) != 1 or X.stride(0) < D:
            X = X.contiguous()
        st = _steps(steps, D)
        out = torch.empty((self.N, D), dtype=torch.float32, device=Q.device) if want_rec else None
        ssd = torch.empty(D, dtype=torch.float64, device=Q.device)
        with torch.cuda.device(Q.device):
            check(_lib.lib().raht_dequant_inv_sqdiff(self._h, C.c_void_p(Q.data_ptr()), D, D, st, len(st), C.c_void_p(X.data_ptr()), X.stride(0),
                                                     C.c_void_p(out.data_ptr()) if want_rec else None, D, C.c_void_p(ssd.data_ptr()), _stream()))
        return out, ssd

    def forward_quant_mixed(self, Cmat, steps, n_wide=3):
        """Forward RAHT + quantize + reorder of a float32 matrix whose first ``n_wide`` channels (the xyz columns of a
        59-column frame, python/voxelize_pc.py:155) are carried in float64 -- the reference's precision
        (python/encode_3dgs.py:82-83,204) where float32 cannot hold the quotient -- in the same launches as the float32
        channels. Wide columns: bit-identical to the float64 kernels; the others: bit-identical to ``forward_quant``."""
        _need_cuda(Cmat, "C")
        X = Cmat.to(torch.float32)
        if X.stride(1) != 1 or X.stride(0) < X.shape[1]:
            X = X.contiguous()
        D = X.shape[1]
        st = _steps64(steps, D)
        Q = torch.empty((self.N, D), dtype=torch.int32, device=X.device)
        with torch.cuda.device(X.device):
            check(_lib.lib().raht_fwd_quant_mixed(self._h, C.c_void_p(X.data_ptr()), X.stride(0), D, st, len(st), int(n_wide),
                                                  C.c_void_p(Q.data_ptr()), D, _stream()))
        return Q

    def dequant_inverse_mixed(self, Q, steps, n_wide=3, out=None):
        """Un-reorder + dequantize + inverse RAHT -> float32 C, the first ``n_wide`` channels computed in float64 and
        rounded once on output (counterpart of ``forward_quant_mixed``)."""
        _need_cuda(Q, "Q")
        Q = Q.to(torch.int32).contiguous()
        D = Q.shape[1]
        st = _steps64(steps, D)
        if out is None:
            out = torch.empty((self.N, D), dtype=torch.float32, device=Q.device)
        with torch.cuda.device(Q.device):
            check(_lib.lib().raht_dequant_inv_mixed(self._h, C.c_void_p(Q.data_ptr()), D, D, st, len(st), int(n_wide),
                                                    C.c_void_p(out.data_ptr()), out.stride(0), _stream()))
        return out

    def mixed_stats(self, D=59, n_wide=3):
        """Tile rows and rows per stage of the mixed-precision schedule (tile_rows 0: the shape takes the two-pass path)."""
        tr, ns = C.c_int(0), C.c_int(0)
        rows = (C.c_int64 * 64)()
        with torch.cuda.device(self.device):
            check(_lib.lib().raht_plan_mixed_stats(self._h, int(D), int(n_wide), C.byref(tr), C.byref(ns), rows, 64))
        return {"tile_rows": tr.value, "rows_per_stage": [int(rows[i]) for i in range(ns.value)]}

    def quant_reorder(self, T, steps):
        """int32 Q[k] = floor(T[order[k]] / step + 0.5)  (encode_3dgs.py:204,210,215)."""
        _need_cuda(T, "T")
        if T.dtype == torch.float64:
            T = T.contiguous()
            Q = torch.empty((self.N, T.shape[1]), dtype=torch.int32, device=T.device)
            return self._f64_quant_call(_lib.lib().raht_quant_reorder_f64, T, T.shape[1], steps, Q)
        T = T.to(torch.float32).contiguous()
        D = T.shape[1]
        st = _steps(steps, D)
        Q = torch.empty((self.N, D), dtype=torch.int32, device=T.device)
        with torch.cuda.device(T.device):
            check(_lib.lib().raht_quant_reorder(self._h, C.c_void_p(T.data_ptr()), D, D, st, len(st),
                                                C.c_void_p(Q.data_ptr()), D, _stream()))
        return Q

    def dequant_unreorder(self, Q, steps, dtype=torch.float32):
        """T[order[k]] = Q[k] * step  (encode_3dgs.py:261,267-268); float32, or float64 like the reference."""
        _need_cuda(Q, "Q")
        Q = Q.to(torch.int32).contiguous()
        D = Q.shape[1]
        if dtype == torch.float64:
            T = torch.empty((self.N, D), dtype=torch.float64, device=Q.device)
            return self._f64_quant_call(_lib.lib().raht_dequant_unreorder_f64, Q, D, steps, T)
        st = _steps(steps, D)
        T = torch.empty((self.N, D), dtype=torch.float32, device=Q.device)
        with torch.cuda.device(Q.device):
            check(_lib.lib().raht_dequant_unreorder(self._h, C.c_void_p(Q.data_ptr()), D, D, st, len(st),
                                                    C.c_void_p(T.data_ptr()), D, _stream()))
        return T


def _steps(steps, D):
    if isinstance(steps, (int, float)):
        steps = [float(steps)]
    steps = [float(s) for s in steps]
    if len(steps) not in (1, D):
        raise ValueError("steps must be a scalar or have D entries")
    return (C.c_float * len(steps))(*steps)


def _steps64(steps, D):
    if isinstance(steps, (int, float)):
        steps = [float(steps)]
    steps = [float(s) for s in steps]
    if len(steps) not in (1, D):
        raise ValueError("steps must be a scalar or have D entries")
    return (C.c_double * len(steps))(*steps)


# ---------------------------------------------------------------------------------------------------
# several scenes in one set of launches (include/raht.h: raht_*_batch)
# ---------------------------------------------------------------------------------------------------
def _batch_arrays(plans, mats, dtype, name):
    n = len(plans)
    if n < 1 or len(mats) != n:
        raise ValueError("batch: one matrix per plan")
    D = int(mats[0].shape[1])
    ms = []
    for p, m in zip(plans, mats):
        _need_cuda(m, name)
        if m.dim() != 2 or m.shape[0] != p.N or m.shape[1] != D:
            raise ValueError(f"batch: expected ({p.N}, {D}) matrices, got {tuple(m.shape)}")
        if m.dtype != dtype:
            m = m.to(dtype)
        if m.stride(1) != 1 or m.stride(0) < D:
            m = m.contiguous()
        ms.append(m)
    hp = (C.c_void_p * n)(*[p._h for p in plans])
    mp = (C.c_void_p * n)(*[m.data_ptr() for m in ms])
    ld = (C.c_int64 * n)(*[m.stride(0) for m in ms])
    return n, D, ms, hp, mp, ld


def _batch_out(plans, D, dtype, device):
    outs = [torch.empty((p.N, D), dtype=dtype, device=device) for p in plans]
    return outs, (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs]), (C.c_int64 * len(outs))(*[D] * len(outs))


def forward_batch(plans, Cs):
    """[T_i] = forward RAHT of every scene (plans[i], Cs[i]) in one set of launches; float32."""
    n, D, ms, hp, mp, ld = _batch_arrays(plans, Cs, torch.float32, "C")
    outs, op, old = _batch_out(plans, D, torch.float32, ms[0].device)
    with torch.cuda.device(ms[0].device):
        check(_lib.lib().raht_fwd_batch(n, hp, mp, ld, D, op, old, _stream()))
    return outs


def inverse_batch(plans, Ts):
    n, D, ms, hp, mp, ld = _batch_arrays(plans, Ts, torch.float32, "T")
    outs, op, old = _batch_out(plans, D, torch.float32, ms[0].device)
    with torch.cuda.device(ms[0].device):
        check(_lib.lib().raht_inv_batch(n, hp, mp, ld, D, op, old, _stream()))
    return outs


def forward_quant_batch(plans, Cs, steps):
    """[Q_i] = forward RAHT + quantize + reorder of every scene in one set of launches (bit-identical to
    plans[i].forward_quant(Cs[i], steps))."""
    n, D, ms, hp, mp, ld = _batch_arrays(plans, Cs, torch.float32, "C")
    st = _steps(steps, D)
    outs, op, old = _batch_out(plans, D, torch.int32, ms[0].device)
    with torch.cuda.device(ms[0].device):
        check(_lib.lib().raht_fwd_quant_batch(n, hp, mp, ld, D, st, len(st), op, old, _stream()))
    return outs


def dequant_inverse_batch(plans, Qs, steps):
    n, D, ms, hp, mp, ld = _batch_arrays(plans, Qs, torch.int32, "Q")
    st = _steps(steps, D)
    outs, op, old = _batch_out(plans, D, torch.float32, ms[0].device)
    with torch.cuda.device(ms[0].device):
        check(_lib.lib().raht_dequant_inv_batch(n, hp, mp, ld, D, st, len(st), op, old, _stream()))
    return outs


def quant_rows(X, steps, pos, Q):
    """Q[pos[i], :] = floor(X[i, :] / step + 0.5) in place (X float32 (n, D), pos int64 (n,), Q int32)."""
    _need_cuda(X, "X")
    if X.shape[0] == 0:
        return Q
    X = X.to(torch.float32).contiguous()
    n, D = X.shape
    st = _steps(steps, D)
    pos = pos.to(torch.int64).contiguous()
    if Q.dtype != torch.int32 or Q.stride(1) != 1:
        raise ValueError("Q must be an int32 matrix with contiguous rows")
    with torch.cuda.device(X.device):
        check(_lib.lib().raht_quant_rows(C.c_void_p(X.data_ptr()), D, n, D, st, len(st), C.c_void_p(pos.data_ptr()),
                                         C.c_void_p(Q.data_ptr()), Q.stride(0), _stream()))
    return Q


def dequant_rows(Q, steps, pos, out=None):
    """-> float32 (n, D): Q[pos[i], :] * step (written into ``out`` when given: a contiguous float32 (n, D) view)."""
    _need_cuda(Q, "Q")
    if Q.dtype != torch.int32 or Q.stride(1) != 1:
        raise ValueError("Q must be an int32 matrix with contiguous rows")
    D = Q.shape[1]
    st = _steps(steps, D)
    pos = pos.to(torch.int64).contiguous()
    X = torch.empty((pos.shape[0], D), dtype=torch.float32, device=Q.device) if out is None else out
    if X.dtype != torch.float32 or tuple(X.shape) != (pos.shape[0], D) or not X.is_contiguous():
        raise ValueError("out must be a contiguous float32 (n, D) tensor")
    with torch.cuda.device(Q.device):
        check(_lib.lib().raht_dequant_rows(C.c_void_p(Q.data_ptr()), Q.stride(0), C.c_void_p(pos.data_ptr()), pos.shape[0], D,
                                           st, len(st), C.c_void_p(X.data_ptr()), D, _stream()))
    return X


def _rows_move(fn, src, pos, dst):
    _need_cuda(src, "src")
    if pos.shape[0] == 0:                     # nothing to move (an empty rank of a sharded scene); empty tensors have no strides to check
        return dst
    if src.dtype != dst.dtype or src.dtype not in (torch.float32, torch.float64, torch.int32) or src.stride(1) != 1 or dst.stride(1) != 1:
        raise ValueError("rows: float32 / float64 / int32 matrices with contiguous rows of one dtype")
    if pos.dtype != torch.int64 or not pos.is_contiguous():
        pos = pos.to(torch.int64).contiguous()
    with torch.cuda.device(src.device):
        check(fn(C.c_void_p(src.data_ptr()), src.stride(0), C.c_void_p(pos.data_ptr()), pos.shape[0], src.shape[1],
                 src.element_size(), C.c_void_p(dst.data_ptr()), dst.stride(0), _stream()))
    return dst


def rows_gather(src, pos, out):
    """out[i, :] = src[pos[i], :] in one launch (pos int64 (n,))."""
    return _rows_move(_lib.lib().raht_rows_gather, src, pos, out)


def rows_scatter(src, pos, out):
    """out[pos[i], :] = src[i, :] in one launch."""
    return _rows_move(_lib.lib().raht_rows_scatter, src, pos, out)


# ---------------------------------------------------------------------------------------------------
# plan tokens: what RAHT_param hands back in place of the reference's List / Flags / weights
# ---------------------------------------------------------------------------------------------------
def _token(plan):
    t = torch.tensor([_MAGIC, plan.id], dtype=torch.int64, device=plan.device)
    t._raht_plan = plan
    return t


def plan_of(lst):
    """Recover the RahtPlan from what the driver passes back as List (or Flags / weights)."""
    if isinstance(lst, RahtPlan):
        return lst
    tok = lst[0] if isinstance(lst, (list, tuple)) else lst
    p = getattr(tok, "_raht_plan", None)
    if p is not None:
        return p
    if isinstance(tok, torch.Tensor) and tok.numel() == 2 and tok.dtype == torch.int64:
        magic, pid = tok.cpu().tolist()          # token was copied: fall back to its contents
        if magic == _MAGIC and pid in _plans:
            return _plans[pid]
    raise RuntimeError("raht-3dgs-codec_amd: List/Flags/weights must be the plan tokens returned by "
                       "RAHT_param_reorder_fast of this package (reference-shaped lists are not accepted; "
                       "build the plan from V with RAHT_param_reorder_fast)")


@torch.no_grad()
def RAHT_param_reorder_fast(V, minV, width, depth):
    """Drop-in for reference python/RAHT_param.py:190-279 (call site encode_3dgs.py:149).

    V: (N, 3) tensor holding integer voxel coordinates, Morton-sorted, unique, on the GPU.
    Returns (List, Flags, weights, order_RAGFT); the three lists hold one opaque plan token each.
    """
    plan = RahtPlan.from_coords(V, minV, width, depth)
    _plans[plan.id] = plan
    while len(_plans) > _PLAN_CACHE:
        _plans.popitem(last=False)
    return [_token(plan)], [_token(plan)], [_token(plan)], plan.order_RAGFT


@torch.no_grad()
def RAHT2_optimized(Cmat, List, Flags, weights, one_based=False):
    """Drop-in for reference python/RAHT.py:252-336 (call site encode_3dgs.py:159). Returns (T, w)."""
    if one_based:
        raise ValueError("one_based lists are a MATLAB convention; plan tokens are index-free")
    return plan_of(List).forward(Cmat, want_w=True)


@torch.no_grad()
def inverse_RAHT_optimized(T, List, Flags, weights, one_based=False):
    """Drop-in for reference python/iRAHT.py:40-114 (call site encode_3dgs.py:274). Returns C."""
    if one_based:
        raise ValueError("one_based lists are a MATLAB convention; plan tokens are index-free")
    return plan_of(List).inverse(T)


raht_fn = {
    "RAHT": RAHT2_optimized,
    "iRAHT": inverse_RAHT_optimized,
    "RAHT_param": RAHT_param_reorder_fast,
}


# ---------------------------------------------------------------------------------------------------
# voxelizer (reference python/voxelize_pc.py)
# ---------------------------------------------------------------------------------------------------
@torch.no_grad()
def get_morton_code(V, J):
    """Drop-in for reference python/voxelize_pc.py:25-59. V: (N, 3) integer tensor on the GPU."""
    _need_cuda(V, "V")
    V = V.to(torch.int64).contiguous()
    out = torch.empty(V.shape[0], dtype=torch.int64, device=V.device)
    with torch.cuda.device(V.device):
        check(_lib.lib().raht_morton(C.c_void_p(V.data_ptr()), V.shape[0], int(J), C.c_void_p(out.data_ptr()),
                                     _stream()))
    return out


@torch.no_grad()
def voxel_keys(PC, vmin, width, J):
    """Unsorted 3J-bit Morton keys (int64) of the points of PC (n, >= 3) float32 for a given bounding box: the
    voxelizer's first phase (reference python/voxelize_pc.py:92-100)."""
    _need_cuda(PC, "PC")
    if PC.dtype != torch.float32 or PC.stride(1) != 1:
        PC = PC.to(torch.float32).contiguous()
    out = torch.empty(PC.shape[0], dtype=torch.int64, device=PC.device)
    vm = (C.c_float * 3)(*[float(x) for x in vmin])
    with torch.cuda.device(PC.device):
        check(_lib.lib().raht_voxel_keys(C.c_void_p(PC.data_ptr()), PC.stride(0), PC.shape[0], vm, float(width), int(J),
                                         C.c_void_p(out.data_ptr()), _stream()))
    return out


@torch.no_grad()
def sort_keys(keys, nbits=64):
    """Stable LSD radix sort of Morton keys on device -> (keys_sorted, idx)."""
    _need_cuda(keys, "keys")
    k = keys.contiguous()
    ko = torch.empty_like(k)
    idx = torch.empty(k.shape[0], dtype=torch.int64, device=k.device)
    with torch.cuda.device(k.device):
        check(_lib.lib().raht_sort_keys(C.c_void_p(k.data_ptr()), k.shape[0], int(nbits), C.c_void_p(ko.data_ptr()),
                                        C.c_void_p(idx.data_ptr()), _stream()))
    return ko, idx


_VMIN_CACHE = {}


def _vmin_tensor(vmin, dev):
    """info['vmin'] as a device tensor (the reference's is one). The frames of a sequence share their bounding box: the 12-byte
    upload (a synchronising copy, ~30 us of a 0.7 ms call) is made once per (device, box), the tensor handed out is a clone."""
    key = (str(dev), vmin)
    t = _VMIN_CACHE.get(key)
    if t is None:
        if len(_VMIN_CACHE) > 64:
            _VMIN_CACHE.clear()
        t = _VMIN_CACHE[key] = torch.tensor(list(vmin), dtype=torch.float32, device=dev)
    return t.clone()


@torch.no_grad()
def voxelize_pc_batched(PC, vmin=None, width=None, J=10, device="cuda", residuals=True, sorted_points=True):
    """Drop-in for reference python/voxelize_pc.py:62-172.

    Returns (PCvox, PCsorted, voxel_indices, DeltaPC, info) like the reference. PCvox,
    voxel_indices, info['sort_idx'] and the sorted Morton keys (info['keys_sorted'], extra) come
    from the HIP voxelizer; PCsorted / DeltaPC are its secondary outputs, produced by the same call (``raht_voxelize_all``:
    the pass that forms the voxel means writes them too; ``residuals=False`` skips DeltaPC; ``sorted_points=False`` skips
    PCsorted and returns None in its place; the codec path only consumes PCvox).
    """
    PC = PC.to(device)
    _need_cuda(PC, "PC")
    PC = PC.to(torch.float32).contiguous()
    N, ld = PC.shape
    d = ld - 3
    dev = PC.device
    keys = torch.empty(N, dtype=torch.int64, device=dev)
    idx = torch.empty(N, dtype=torch.int64, device=dev)
    vidx = torch.empty(N, dtype=torch.int64, device=dev)
    pcv = torch.empty((N, ld), dtype=torch.float32, device=dev)
    nvox = C.c_int64()
    vmin_out = (C.c_float * 3)()
    w_out, vs_out = C.c_double(), C.c_double()
    vm = None
    if vmin is not None:
        vm = (C.c_float * 3)(*[float(x) for x in (vmin.detach().cpu().tolist() if isinstance(vmin, torch.Tensor) else vmin)])
    # the secondary outputs (voxelize_pc.py:103-111, 147-156) come out of the same call: the pass that forms the voxel means
    # has every point's row in registers anyway (raht_voxelize_all; raht_voxelize + raht_voxelize_residuals give the same bits)
    PCsorted = torch.empty((N, ld), dtype=torch.float32, device=dev) if sorted_points else None
    DeltaPC = torch.empty((N, ld), dtype=torch.float32, device=dev) if residuals else None
    with torch.cuda.device(dev):
        check(_lib.lib().raht_voxelize_all(C.c_void_p(PC.data_ptr()), ld, N, d, vm, -1.0 if width is None else float(width),
                                           int(J), C.c_void_p(keys.data_ptr()), C.c_void_p(idx.data_ptr()),
                                           C.c_void_p(vidx.data_ptr()), C.c_void_p(pcv.data_ptr()), None,
                                           C.c_void_p(PCsorted.data_ptr()) if PCsorted is not None else None,
                                           C.c_void_p(DeltaPC.data_ptr()) if DeltaPC is not None else None, C.byref(nvox),
                                           vmin_out, C.byref(w_out), C.byref(vs_out), _stream()))
    nv = nvox.value
    voxel_indices = vidx[:nv]
    PCvox = pcv[:nv]
    vmin_t = _vmin_tensor(tuple(vmin_out), dev)
    info = {"Nvox": nv, "voxel_size": vs_out.value, "vmin": vmin_t, "width": w_out.value, "N": N,
            "sort_idx": idx, "keys_sorted": keys}
    return PCvox, PCsorted, voxel_indices, DeltaPC, info


@torch.no_grad()
def voxelize_merge(G, vmin=None, width=None, J=10, device="cuda", weight_by_opacity=True, want_means=True):
    """N whole Gaussians [xyz | quat(4) | scale(3) | opacity | colour(cd)] -> the voxelized frame in ONE call and one pass over the
    rows (raht_voxelize_merge): voxelizer + the reference's per-voxel opacity-weighted merge (python/test_voxelize_3dgs.py:203-257,
    cuda/merge_cluster.cu:2-111), bit-identical to ``voxelize_pc_batched`` followed by ``merge_gaussian_clusters_with_indices``.
    Returns (Gvox (Nvox, 11 + cd): integer voxel coordinates as floats + merged attributes, info: Nvox, voxel_size, vmin, width,
    N, sort_idx, keys_sorted, voxel_indices, merged_means)."""
    G = G.to(device)
    _need_cuda(G, "G")
    G = G.to(torch.float32).contiguous()
    N, ld = G.shape
    if ld < 11:
        raise ValueError("voxelize_merge: rows are [xyz | quat(4) | scale(3) | opacity | colours]: at least 11 columns")
    dev = G.device
    keys = torch.empty(N, dtype=torch.int64, device=dev)
    idx = torch.empty(N, dtype=torch.int64, device=dev)
    vidx = torch.empty(N, dtype=torch.int64, device=dev)
    gv = torch.empty((N, ld), dtype=torch.float32, device=dev)
    mm = torch.empty((N, 3), dtype=torch.float32, device=dev) if want_means else None
    nvox = C.c_int64()
    vmin_out = (C.c_float * 3)()
    w_out, vs_out = C.c_double(), C.c_double()
    vm = None
    if vmin is not None:
        vm = (C.c_float * 3)(*[float(x) for x in (vmin.detach().cpu().tolist() if isinstance(vmin, torch.Tensor) else vmin)])
    with torch.cuda.device(dev):
        check(_lib.lib().raht_voxelize_merge(C.c_void_p(G.data_ptr()), ld, N, ld - 11, 1 if weight_by_opacity else 0, vm,
                                             -1.0 if width is None else float(width), int(J), C.c_void_p(keys.data_ptr()), C.c_void_p(idx.data_ptr()),
                                             C.c_void_p(vidx.data_ptr()), C.c_void_p(gv.data_ptr()), C.c_void_p(mm.data_ptr()) if mm is not None else None,
                                             C.byref(nvox), vmin_out, C.byref(w_out), C.byref(vs_out), _stream()))
    nv = nvox.value
    info = {"Nvox": nv, "voxel_size": vs_out.value, "vmin": _vmin_tensor(tuple(vmin_out), dev), "width": w_out.value, "N": N,
            "sort_idx": idx, "keys_sorted": keys, "voxel_indices": vidx[:nv], "merged_means": None if mm is None else mm[:nv]}
    return gv[:nv], info


@torch.no_grad()
def voxelize_plan(PC, vmin=None, width=None, J=10, device="cuda"):
    """Unsorted cloud -> (PCvox, plan, info): the voxelizer's own sorted voxel keys go STRAIGHT into the RAHT plan
    (reference: voxelize_pc_batched, python/voxelize_pc.py:62-172, then -- one script later, through a PLY file --
    RAHT_param_reorder_fast on the voxel coordinates, python/RAHT_param.py:190-279, which re-derives the same Morton keys).
    No key recomputation, no key copy (the plan borrows the key tensor and keeps it alive), one plan build per frame, ONE call
    through the C ABI (raht_voxelize_plan). PCvox[:, 3:] is the attribute matrix in the plan's row order. info: Nvox,
    voxel_size, vmin (HOST tensor), width, N, voxel_keys, voxel_indices."""
    PC = PC.to(device)
    _need_cuda(PC, "PC")
    PC = PC.to(torch.float32).contiguous()
    N, ld = PC.shape
    dev = PC.device
    vkeys = torch.empty(N, dtype=torch.int64, device=dev)
    vidx = torch.empty(N, dtype=torch.int64, device=dev)
    pcv = torch.empty((N, ld), dtype=torch.float32, device=dev)
    nvox = C.c_int64()
    vmin_out = (C.c_float * 3)()
    w_out, vs_out = C.c_double(), C.c_double()
    vm = None
    if vmin is not None:
        vm = (C.c_float * 3)(*[float(x) for x in (vmin.detach().cpu().tolist() if isinstance(vmin, torch.Tensor) else vmin)])
    h = C.c_void_p()
    with torch.cuda.device(dev):
        check(_lib.lib().raht_voxelize_plan(C.c_void_p(PC.data_ptr()), ld, N, ld - 3, vm, -1.0 if width is None else float(width),
                                            int(J), C.c_void_p(vkeys.data_ptr()), C.c_void_p(vidx.data_ptr()), C.c_void_p(pcv.data_ptr()),
                                            C.byref(nvox), vmin_out, C.byref(w_out), C.byref(vs_out), _stream(), C.byref(h)))
    nv = nvox.value
    plan = RahtPlan(h, dev)
    vkeys = vkeys[:nv]
    plan._keys_ref = vkeys                                # borrowed by the plan: alive as long as it is
    info = {"Nvox": nv, "voxel_size": vs_out.value, "vmin": torch.tensor(list(vmin_out), dtype=torch.float32), "width": w_out.value,
            "N": N, "voxel_keys": vkeys, "voxel_indices": vidx[:nv]}
    return pcv[:nv], plan, info
