"""Test-only reference implementation of the shard-local plan interface used by
raht_3dgs_codec_amd.sharded (same methods as ops.RahtPlan), in numpy on CPU tensors.
It lets the world_size-2 gloo tests exercise the host-side sharding logic without a GPU.
Written from the list-free formulation; float64."""
import numpy as np
import torch


def _msb(x):
    m = np.zeros(x.shape, dtype=np.int64)
    t = x.copy()
    for s in (32, 16, 8, 4, 2, 1):
        big = t >= (np.uint64(1) << np.uint64(s))
        m[big] += s
        t[big] >>= np.uint64(s)
    return m


class NumpyPlan:
    def __init__(self, keys, nbits, top_level=None, leaf_weights=None):
        k = np.ascontiguousarray(keys.cpu().numpy()).astype(np.uint64)
        N = k.shape[0]
        assert N >= 1 and (N == 1 or np.all(k[1:] > k[:-1]))
        self.N, self.nbits = N, nbits
        self.top_level = 64 if top_level is None else int(top_level)
        lvl = np.full(N, 255, dtype=np.int64)
        wl = np.zeros(N, dtype=np.int64)
        wr = np.zeros(N, dtype=np.int64)
        if N > 1:
            lvl[1:] = _msb(k[1:] ^ k[:-1])
            l = lvl[1:].astype(np.uint64)
            i = np.arange(1, N)
            wl[1:] = i - np.searchsorted(k, (k[:-1] >> l) << l, side="left")
            wr[1:] = np.searchsorted(k, ((k[1:] >> l) + np.uint64(1)) << l, side="left") - i
        self.lvl, self.wl, self.wr = lvl, wl, wr
        w = np.ones(N, dtype=np.int64) if leaf_weights is None else leaf_weights.cpu().numpy().astype(np.int64)
        self.S = np.concatenate([[0], np.cumsum(w)])
        roots = np.nonzero((np.arange(N) == 0) | ((lvl >= self.top_level) & (lvl < 255)))[0]
        self.root_rows = torch.from_numpy(roots.astype(np.int64))
        bucket = np.where(np.arange(N) == 0, 0, 1 + (20 - np.minimum(lvl, 62) // 3))
        self.order_RAGFT = torch.from_numpy(np.argsort(bucket, kind="stable").astype(np.int64))
        inv = np.empty(N, dtype=np.int64)
        inv[self.order_RAGFT.numpy()] = np.arange(N)
        self.inv_order = torch.from_numpy(inv)

    @property
    def n_roots(self):
        return int(self.root_rows.shape[0])

    def _levels(self):
        top = min(self.top_level, 64)
        return [l for l in range(top) if np.any(self.lvl == l)]

    def _ab(self, rows):
        i0 = rows - self.wl[rows]
        w0 = (self.S[rows] - self.S[i0]).astype(np.float64)
        w1 = (self.S[rows + self.wr[rows]] - self.S[rows]).astype(np.float64)
        return i0, np.sqrt(w0 / (w0 + w1))[:, None], np.sqrt(w1 / (w0 + w1))[:, None]

    def forward(self, C, want_w=False, roots=None):
        T = C.cpu().numpy().astype(np.float64).copy()
        for l in self._levels():
            rows = np.nonzero(self.lvl == l)[0]
            i0, a, b = self._ab(rows)
            x0, x1 = T[i0].copy(), T[rows].copy()
            T[i0] = a * x0 + b * x1
            T[rows] = a * x1 - b * x0
        if roots is not None:
            roots.copy_(torch.from_numpy(T[self.root_rows.numpy()]).to(roots.dtype))
        out = torch.from_numpy(T).to(C.dtype)
        return (out, None) if want_w else out

    def inverse(self, T, roots=None):
        X = T.cpu().numpy().astype(np.float64).copy()
        if roots is not None:
            X[self.root_rows.numpy()] = roots.cpu().numpy()
        for l in reversed(self._levels()):
            rows = np.nonzero(self.lvl == l)[0]
            i0, a, b = self._ab(rows)
            t0, t1 = X[i0].copy(), X[rows].copy()
            X[i0] = a * t0 - b * t1
            X[rows] = b * t0 + a * t1
        return torch.from_numpy(X).to(T.dtype)

    def forward_quant(self, C, step, roots=None):
        T = self.forward(C, roots=roots)
        Q = torch.floor(T[self.order_RAGFT] / step + 0.5).to(torch.int32)
        return Q

    def dequant_inverse(self, Q, step, roots=None):
        T = torch.empty((self.N, Q.shape[1]), dtype=torch.float64)
        T[self.order_RAGFT] = Q.to(torch.float64) * step
        return self.inverse(T, roots=roots)


class NumpyLocalOps:
    quant_dtype = torch.float64

    @staticmethod
    def make_plan(keys, nbits, top_level=None, leaf_weights=None):
        return NumpyPlan(keys, nbits, top_level=top_level, leaf_weights=leaf_weights)
