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
        self._keys = torch.from_numpy(k.view(np.int64).copy())
        self.row_map, self.map_rows = None, None

    def keys_tensor(self):
        return self._keys.clone()

    def set_row_map(self, row_map, n_matrix_rows):
        """plan row i lives in matrix row row_map[i] (ops.RahtPlan.set_row_map)"""
        self.row_map = None if row_map is None else row_map.cpu().numpy().astype(np.int64)
        self.map_rows = None if row_map is None else int(n_matrix_rows)

    def _take(self, M):
        X = M.cpu().numpy().astype(np.float64)
        if self.row_map is not None:
            assert X.shape[0] == self.map_rows
            X = X[self.row_map]
        return X.copy()

    def _give(self, X, like, out):
        if self.row_map is None:
            res = torch.from_numpy(X).to(like.dtype)
            if out is not None:
                out.copy_(res)
                return out
            return res
        if out is None:
            out = torch.zeros((self.map_rows, X.shape[1]), dtype=like.dtype)
        out[torch.from_numpy(self.row_map)] = torch.from_numpy(X).to(like.dtype)
        return out

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

    def forward(self, C, want_w=False, roots=None, out=None):
        T = self._take(C)
        for l in self._levels():
            rows = np.nonzero(self.lvl == l)[0]
            i0, a, b = self._ab(rows)
            x0, x1 = T[i0].copy(), T[rows].copy()
            T[i0] = a * x0 + b * x1
            T[rows] = a * x1 - b * x0
        if roots is not None:
            roots.copy_(torch.from_numpy(T[self.root_rows.numpy()]).to(roots.dtype))
        res = self._give(T, C, out)
        return (res, None) if want_w else res

    def inverse(self, T, roots=None, out=None):
        X = self._take(T)
        if roots is not None:
            X[self.root_rows.numpy()] = roots.cpu().numpy()
        for l in reversed(self._levels()):
            rows = np.nonzero(self.lvl == l)[0]
            i0, a, b = self._ab(rows)
            t0, t1 = X[i0].copy(), X[rows].copy()
            X[i0] = a * t0 - b * t1
            X[rows] = b * t0 + a * t1
        return self._give(X, T, out)

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

    # a few rows at explicit positions (ops.quant_rows / dequant_rows / rows_gather / rows_scatter)
    @staticmethod
    def quant_rows(X, step, pos, Q):
        Q[pos] = torch.floor(X.to(torch.float64) / torch.as_tensor(step, dtype=torch.float64) + 0.5).to(torch.int32)
        return Q

    @staticmethod
    def dequant_rows(Q, step, pos, out):
        out.copy_((Q[pos].to(torch.float64) * step).to(out.dtype))
        return out

    @staticmethod
    def rows_gather(src, pos, out):
        out.copy_(src[pos])
        return out

    @staticmethod
    def rows_scatter(src, pos, out):
        out[pos] = src
        return out

    # front end (sharded.exchange_by_prefix): the oracle's voxelizer arithmetic
    @staticmethod
    def voxel_keys(PC, vmin, width, J):
        P = PC.numpy().astype(np.float32)
        vs = np.float32(width / float(1 << J))
        V0 = P[:, :3] - np.asarray(vmin, dtype=np.float32)[None, :]
        q = np.clip(np.floor(V0 / vs), 0, (1 << J) - 1).astype(np.uint64)          # voxelize_pc.py:92-98
        k = np.zeros(P.shape[0], dtype=np.uint64)
        for i in range(J):
            s = np.uint64(i)
            k |= (((q[:, 2] >> s) & np.uint64(1)) | (((q[:, 1] >> s) & np.uint64(1)) << np.uint64(1))
                  | (((q[:, 0] >> s) & np.uint64(1)) << np.uint64(2))) << np.uint64(3 * i)
        return torch.from_numpy(k.view(np.int64).copy())

    @staticmethod
    def sort_keys(keys, nbits):
        k = keys.numpy()
        idx = np.argsort(k, kind="stable")
        return torch.from_numpy(k[idx].copy()), torch.from_numpy(idx.astype(np.int64))

    @staticmethod
    def voxelize(PC, vmin, width, J):
        from oracle import oracle as orc
        r = orc.voxelize(PC.numpy(), J, vmin=vmin, width=width)
        keys = r["keys_sorted"][r["voxel_indices"]]
        return (torch.from_numpy(r["PCvox"]), torch.from_numpy(keys.view(np.int64).copy()),
                torch.from_numpy(r["voxel_indices"]), {"sort_idx": torch.from_numpy(r["sort_idx"])})
