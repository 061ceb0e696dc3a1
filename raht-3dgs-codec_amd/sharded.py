"""Morton-prefix sharded RAHT (BASELINE.json configs[4]).

A large scene is partitioned across the GPUs of one node by contiguous ranges of the top
``prefix_bits`` (= 9: three octree levels) bits of the Morton key. Every butterfly below those
levels pairs rows that share the prefix, so it is shard-local; each shard is left with one low-pass
row per non-empty prefix node (<= 512 rows in total over all shards). ONE small all-gather of those
rows (<= 512 x D floats ~ 121 KB at D = 59: latency-bound on xGMI) lets every GPU redundantly run
the top 9 binary levels on a weighted <= 512-row tree and keep the coefficients of its own prefixes.
The inverse mirrors it: gather the top coefficients, invert the top tree, continue shard-locally.

One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).
The shard-local work goes through the C ABI (``ops.RahtPlan``); ``local_ops`` lets the CPU tests
inject a reference implementation of the same interface -- the product default has no CPU path.
"""
import torch


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def balanced_prefix_cuts(keys_sorted, nbits, world, prefix_bits=9):
    """Where to cut a Morton-sorted scene into `world` shards: row boundaries
    [0 = b_0 <= b_1 <= ... <= b_world = N] such that every cut falls between two different values of
    the top ``prefix_bits`` key bits, balanced by the 2^prefix_bits-bin population histogram. Shard r =
    rows [b_r, b_{r+1}); hand each to one rank's ``ShardedRaht``. Works on CPU or GPU tensors."""
    N = int(keys_sorted.shape[0])
    nb = 1 << prefix_bits
    pref = (keys_sorted.to(torch.int64) >> (nbits - prefix_bits)).clamp_(0, nb - 1)
    cum = torch.cumsum(torch.bincount(pref, minlength=nb), 0).cpu().tolist()     # rows with prefix <= p
    cuts, lo = [0], 0
    for r in range(1, world):
        target = N * r / world
        # the prefix boundary whose row count is closest to the target, never moving backwards
        best, best_err = lo, None
        for p in range(lo, nb):
            rows_below = cum[p - 1] if p > 0 else 0
            err = abs(rows_below - target)
            if best_err is None or err < best_err:
                best, best_err = p, err
            if rows_below > target:
                break
        lo = best
        cuts.append(cum[best - 1] if best > 0 else 0)
    cuts.append(N)
    return cuts


class HipLocalOps:
    """Shard-local operations on the MI355X through libraht_hip.so."""
    quant_dtype = torch.float32          # arithmetic type of the fused quantize / dequantize kernels

    @staticmethod
    def make_plan(keys, nbits, top_level=None, leaf_weights=None):
        from .ops import RahtPlan
        return RahtPlan.from_keys(keys, nbits, leaf_weights=leaf_weights, top_level=top_level)

    # the few top coefficients, quantized into / dequantized out of their places in Q: one launch each
    @staticmethod
    def quant_rows(X, step, pos, Q):
        from .ops import quant_rows
        return quant_rows(X, step, pos, Q)

    @staticmethod
    def dequant_rows(Q, step, pos):
        from .ops import dequant_rows
        return dequant_rows(Q, step, pos)


class ShardedRaht:
    def __init__(self, keys_sorted, nbits, prefix_bits=9, group=None, local_ops=None):
        """keys_sorted: this rank's sorted, unique Morton keys (int64 tensor). Ranks must own disjoint,
        increasing ranges of the top ``prefix_bits`` bits (rank 0 the lowest prefixes)."""
        self.ops = local_ops or HipLocalOps
        self.qdt = getattr(self.ops, "quant_dtype", torch.float32)
        self.dist = _dist()
        self.group = group
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self.nbits, self.prefix_bits = int(nbits), int(prefix_bits)
        if not (0 < self.prefix_bits < self.nbits):
            raise ValueError("prefix_bits must be in (0, nbits)")
        self.N = int(keys_sorted.shape[0])
        dev = keys_sorted.device
        self.device = dev
        # shard-local plan: butterflies at levels >= nbits - prefix_bits are left to the top stage
        self.plan = self.ops.make_plan(keys_sorted, self.nbits, top_level=self.nbits - self.prefix_bits)
        self.root_rows = self.plan.root_rows                    # first row of every local prefix node
        self.n_roots = int(self.root_rows.shape[0])
        pref = (keys_sorted[self.root_rows] >> (self.nbits - self.prefix_bits)).to(torch.int64)
        ends = torch.cat([self.root_rows[1:], torch.tensor([self.N], dtype=torch.int64, device=dev)])
        counts = (ends - self.root_rows).to(torch.int64)
        # exchange the root directory once (prefix id + leaf count per root)
        self.max_roots = 1 << self.prefix_bits
        self._pads, self._valid_idx, self._root_pos = {}, None, None
        sizes = self._all_gather_rows(torch.tensor([[self.n_roots]], dtype=torch.int64, device=dev), 1).reshape(-1)
        self.sizes = [int(x) for x in sizes.tolist()]
        self.offset = sum(self.sizes[:self.rank])
        allpref = self._gather_var(pref.reshape(-1, 1)).reshape(-1)
        allcnt = self._gather_var(counts.reshape(-1, 1)).reshape(-1)
        if allpref.numel() > 1 and not bool((allpref[1:] > allpref[:-1]).all()):
            raise ValueError("shards must own disjoint, increasing Morton-prefix ranges")
        self.total_rows = int(allcnt.sum().item())
        # the top tree: <= 2^prefix_bits weighted leaves, replicated on every rank
        self.top = self.ops.make_plan(allpref.contiguous(), self.prefix_bits, leaf_weights=allcnt.contiguous())

    # ---- collectives ------------------------------------------------------------------------------
    def _all_gather_rows(self, x, rows):
        """all-gather a (rows, cols) tensor -> (world * rows, cols)."""
        if self.world == 1:
            return x
        if x.is_cuda and self.dist.get_backend(self.group) == "gloo":
            # test / debugging configuration (several ranks sharing one GPU): stage through the host
            xc = x.contiguous().cpu()
            outc = torch.empty((self.world * rows, x.shape[1]), dtype=x.dtype)
            self.dist.all_gather_into_tensor(outc, xc, group=self.group)
            return outc.to(x.device)
        out = torch.empty((self.world * rows, x.shape[1]), dtype=x.dtype, device=x.device)
        self.dist.all_gather_into_tensor(out, x.contiguous(), group=self.group)
        return out

    def _gather_var(self, x):
        """all-gather per-rank row blocks of different heights: pad to the largest block, ONE
        all-gather, then one index_select with a precomputed index picks the valid rows."""
        if self.world == 1:
            return x
        m = max(self.sizes)
        key = (x.dtype, x.shape[1])
        pad = self._pads.get(key)
        if pad is None:
            pad = torch.zeros((m, x.shape[1]), dtype=x.dtype, device=x.device)
            self._pads[key] = pad
        pad[: x.shape[0]].copy_(x)
        allp = self._all_gather_rows(pad, m)
        if self._valid_idx is None:
            idx = [r * m + i for r in range(self.world) for i in range(self.sizes[r])]
            self._valid_idx = torch.tensor(idx, dtype=torch.int64, device=x.device)
        return allp.index_select(0, self._valid_idx)

    def _mine(self, allrows):
        return allrows[self.offset: self.offset + self.n_roots].contiguous()

    # ---- transforms ------------------------------------------------------------------------------
    def forward(self, C):
        """-> T_local: every row holds its coefficient of the WHOLE scene's RAHT."""
        roots = torch.empty((self.n_roots, C.shape[1]), dtype=C.dtype, device=C.device)
        T = self.plan.forward(C, want_w=False, roots=roots)
        top = self.top.forward(self._gather_var(roots), want_w=False)
        T[self.root_rows] = self._mine(top)
        return T

    def inverse(self, T):
        low = self.top.inverse(self._gather_var(T[self.root_rows].contiguous()))
        return self.plan.inverse(T, roots=self._mine(low))

    def forward_quant(self, C, step):
        """-> Q_local (int32, rank-local order_RAGFT order); the top coefficients are quantized too."""
        roots = torch.empty((self.n_roots, C.shape[1]), dtype=self.qdt, device=C.device)
        Q = self.plan.forward_quant(C, step, roots=roots)
        top = self._mine(self.top.forward(self._gather_var(roots), want_w=False))
        if hasattr(self.ops, "quant_rows"):
            self.ops.quant_rows(top, step, self._root_positions(), Q)
        else:
            # a tensor divisor: torch turns division by a Python scalar into a multiplication by 1 / step
            # on the GPU, which rounds differently from the kernels' IEEE division next to a tie
            st = torch.as_tensor(step, dtype=top.dtype, device=top.device)
            Q[self._root_positions()] = torch.floor(top / st + 0.5).to(torch.int32)
        return Q

    def _root_positions(self):
        if self._root_pos is None:
            self._root_pos = self.plan.inv_order[self.root_rows].contiguous()
        return self._root_pos

    def dequant_inverse(self, Q, step):
        if hasattr(self.ops, "dequant_rows"):
            roots_c = self.ops.dequant_rows(Q, step, self._root_positions())
        else:
            roots_c = (Q[self._root_positions()].to(self.qdt) * step).contiguous()
        low = self.top.inverse(self._gather_var(roots_c))
        return self.plan.dequant_inverse(Q, step, roots=self._mine(low))

    # ---- bench helpers -----------------------------------------------------------------------------
    def step(self, C, quant_step=None):
        if quant_step is None:
            return self.inverse(self.forward(C))
        return self.dequant_inverse(self.forward_quant(C, quant_step), quant_step)

    def roundtrip_error(self, C):
        R = self.inverse(self.forward(C))
        return float(((R - C).abs().max() / C.abs().max()).item())
