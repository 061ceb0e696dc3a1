"""Morton-prefix sharded RAHT (BASELINE.json configs[4]).

A large scene is partitioned across the GPUs of one node by contiguous ranges of the top
``prefix_bits`` (= 9: three octree levels) bits of the Morton key. Every butterfly below those
levels pairs rows that share the prefix, so it is shard-local; each shard is left with one low-pass
row per non-empty prefix node (<= 512 rows in total over all shards). ONE small all-gather of those
rows (<= 512 x D floats ~ 121 KB at D = 59: latency-bound on xGMI) lets every GPU redundantly run
the top 9 binary levels on a weighted <= 512-row tree and keep the coefficients of its own prefixes.
The inverse mirrors it: gather the top coefficients, invert the top tree, continue shard-locally.

What one direction of a step enqueues besides the shard-local transform -- nothing else, no torch ops:

    forward   local kernels write their root rows STRAIGHT into this rank's slot of the send buffer
              -> all_gather_into_tensor (padded slots of max_roots rows per rank)
              -> ONE launch of the replicated top tree, in place on the padded gather buffer (row map)
              -> ONE launch that quantizes this rank's top coefficients into their places in Q
                 (or scatters them into T)
    inverse   ONE launch dequantizes (gathers) this rank's top coefficients into the send buffer
              -> all_gather_into_tensor -> ONE launch of the top tree
              -> local kernels read their roots straight from this rank's slot of the result

One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).
The shard-local work goes through the C ABI (``ops.RahtPlan``); ``local_ops`` lets the CPU tests
inject a reference implementation of the same interface -- the product default has no CPU path.

Un-partitioned input (every rank holds an arbitrary part of the cloud): ``exchange_by_prefix`` is the
distributed counterpart of the reference's single ``torch.sort`` (python/voxelize_pc.py:97-118): global
bounding box, 512-bin prefix histogram, balanced cuts, ONE all-to-all of the points by destination rank,
then the local voxelizer (stable radix sort + per-voxel mean) on what arrived.
"""
import torch


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def cuts_from_histogram(hist, world):
    """Prefix boundaries [0 = c_0 <= c_1 <= ... <= c_world = len(hist)] that balance the row counts of
    ``world`` contiguous prefix ranges: rank r owns prefixes [c_r, c_{r+1})."""
    cum = torch.cumsum(hist.to(torch.int64), 0).cpu().tolist()                    # rows with prefix <= p
    nb = len(cum)
    N = cum[-1] if cum else 0
    cuts, lo = [0], 0
    for r in range(1, world):
        target = N * r / world
        # the prefix boundary whose row count is closest to the target, never moving backwards
        best, best_err = lo, None
        for p in range(lo, nb + 1):
            rows_below = cum[p - 1] if p > 0 else 0
            err = abs(rows_below - target)
            if best_err is None or err < best_err:
                best, best_err = p, err
            if rows_below > target:
                break
        lo = best
        cuts.append(best)
    cuts.append(nb)
    return cuts


def balanced_prefix_cuts(keys_sorted, nbits, world, prefix_bits=9):
    """Where to cut a Morton-sorted scene into `world` shards: row boundaries
    [0 = b_0 <= b_1 <= ... <= b_world = N] such that every cut falls between two different values of
    the top ``prefix_bits`` key bits, balanced by the 2^prefix_bits-bin population histogram. Shard r =
    rows [b_r, b_{r+1}); hand each to one rank's ``ShardedRaht``. Works on CPU or GPU tensors."""
    N = int(keys_sorted.shape[0])
    nb = 1 << prefix_bits
    pref = (keys_sorted.to(torch.int64) >> (nbits - prefix_bits)).clamp_(0, nb - 1)
    hist = torch.bincount(pref, minlength=nb)
    pc = cuts_from_histogram(hist, world)
    cum = [0] + torch.cumsum(hist, 0).cpu().tolist()
    out = [cum[p] for p in pc]
    out[-1] = N
    return out


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
    def dequant_rows(Q, step, pos, out):
        from .ops import dequant_rows
        return dequant_rows(Q, step, pos, out=out)

    @staticmethod
    def rows_gather(src, pos, out):
        from .ops import rows_gather
        return rows_gather(src, pos, out)

    @staticmethod
    def rows_scatter(src, pos, out):
        from .ops import rows_scatter
        return rows_scatter(src, pos, out)

    # front end (exchange_by_prefix)
    @staticmethod
    def voxel_keys(PC, vmin, width, J):
        from .ops import voxel_keys
        return voxel_keys(PC, vmin, width, J)

    @staticmethod
    def sort_keys(keys, nbits):
        from .ops import sort_keys
        return sort_keys(keys, nbits)

    @staticmethod
    def voxelize(PC, vmin, width, J):
        from .ops import voxelize_pc_batched
        PCvox, _, vidx, _, info = voxelize_pc_batched(PC, vmin, width, J, device=PC.device, residuals=False, sorted_points=False)
        return PCvox, info["keys_sorted"][vidx], vidx, info


class _DeviceArray:
    """a raw device pointer as something torch.as_tensor can wrap without copying (__cuda_array_interface__)"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class ShardedRaht:
    def __init__(self, keys_sorted, nbits, prefix_bits=9, group=None, local_ops=None, force_collectives=False, direct=False):
        """keys_sorted: this rank's sorted, unique Morton keys (int64 tensor). Ranks must own disjoint,
        increasing ranges of the top ``prefix_bits`` bits (rank 0 the lowest prefixes).
        force_collectives: issue the all-gathers even in a one-rank group (exercises the RCCL path on one GPU).
        direct: the two all-gathers of a step as DIRECT writes into the peers' gather buffers (include/raht.h, raht_xchg_*:
        one launch per direction, every rank writes its <= 15 KB slot to each peer over its own xGMI link and raises a flag;
        SURVEY.md 5 / 8e) instead of RCCL's all_gather_into_tensor. Opt-in and EXPERIMENTAL: hipIpc + fine-grained memory +
        peer mapping have only been exercised by ranks sharing ONE GPU (tests/test_gpu_sharded.py); no run over xGMI exists.
        The process group is still used once per (D, dtype) to exchange the hipIpc handles. A peer that never arrives ends a
        bounded wait, not a hung GPU -- and leaves a status word: check_exchange() (called by close(), roundtrip_error(),
        check_against_unsharded()) raises on it, and every later direct gather of the object refuses to run. Call close()
        (collective) before dropping the object."""
        self.ops = local_ops or HipLocalOps
        self.force = bool(force_collectives)
        self.direct = bool(direct)
        self._direct_failed = False
        self._xchg_bases = []
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
        self._ev = None                                         # collective timing (time_collectives)
        if self.N > 0:
            # shard-local plan: butterflies at levels >= nbits - prefix_bits are left to the top stage
            self.plan = self.ops.make_plan(keys_sorted, self.nbits, top_level=self.nbits - self.prefix_bits)
            self.root_rows = self.plan.root_rows                # first row of every local prefix node
        else:
            # a rank whose prefix range holds no point (balanced cuts of a very uneven scene, or fewer occupied prefixes
            # than ranks): no local plan, no roots -- but it takes part in every collective like the others
            self.plan = None
            self.root_rows = torch.empty(0, dtype=torch.int64, device=dev)
        self.n_roots = int(self.root_rows.shape[0])
        pref = (keys_sorted[self.root_rows] >> (self.nbits - self.prefix_bits)).to(torch.int64)
        ends = torch.cat([self.root_rows[1:], torch.tensor([self.N], dtype=torch.int64, device=dev)])
        counts = (ends - self.root_rows).to(torch.int64)
        # ---- one-off exchange of the root directory: (prefix id, leaf count) per root, padded slots ----
        sizes = self._all_gather(torch.tensor([[self.n_roots]], dtype=torch.int64, device=dev)).reshape(-1)
        self.sizes = [int(x) for x in sizes.tolist()]
        if max(self.sizes) == 0:
            raise ValueError("ShardedRaht: the scene is empty on every rank")
        self.slot = (max(self.sizes) + 3) // 4 * 4              # rows per rank in the gather buffers (whole 16-byte units per slot)
        self.offset = sum(self.sizes[:self.rank])               # this rank's first entry in the top tree
        directory = torch.zeros((self.slot, 2), dtype=torch.int64, device=dev)
        directory[: self.n_roots, 0] = pref
        directory[: self.n_roots, 1] = counts
        alld = self._all_gather(directory)                      # (world * slot, 2)
        valid = [r * self.slot + i for r in range(self.world) for i in range(self.sizes[r])]
        vidx = torch.tensor(valid, dtype=torch.int64, device=dev)
        allpref, allcnt = alld[vidx, 0].contiguous(), alld[vidx, 1].contiguous()
        if allpref.numel() > 1 and not bool((allpref[1:] > allpref[:-1]).all()):
            raise ValueError("shards must own disjoint, increasing Morton-prefix ranges")
        self.total_rows = int(allcnt.sum().item())
        # ---- the top tree: <= 2^prefix_bits weighted leaves, replicated on every rank, working IN PLACE on the
        # padded gather buffer: entry e of the tree lives in buffer row vidx[e] ----
        self.top = self.ops.make_plan(allpref, self.prefix_bits, leaf_weights=allcnt)
        self.gather_rows = self.world * self.slot
        if self.gather_rows != int(allpref.numel()):          # padded slots: entry e of the tree lives in buffer row vidx[e]
            self.top.set_row_map(vidx, self.gather_rows)
        self._bufs = {}
        self._root_pos = None

    # ---- collectives ------------------------------------------------------------------------------
    def _all_gather(self, x, out=None):
        """all-gather a (rows, cols) tensor -> (world * rows, cols)."""
        if self.world == 1 and not (self.force and self.dist):
            return x
        rows = x.shape[0]
        if x.is_cuda and self.dist.get_backend(self.group) == "gloo":
            # test / debugging configuration (several ranks sharing one GPU): stage through the host
            outc = torch.empty((self.world * rows, x.shape[1]), dtype=x.dtype)
            self.dist.all_gather_into_tensor(outc, x.contiguous().cpu(), group=self.group)
            if out is None:
                return outc.to(x.device)
            out.copy_(outc)
            return out
        if out is None:
            out = torch.empty((self.world * rows, x.shape[1]), dtype=x.dtype, device=x.device)
        self.dist.all_gather_into_tensor(out, x, group=self.group)
        return out

    def _buffers(self, D, dtype):
        """send (slot x D: this rank's roots), recv (world * slot x D: everybody's), res (the top tree's output, same
        layout) -- allocated once per (D, dtype). With one rank the three coincide in size and no collective runs."""
        key = (int(D), dtype)
        b = self._bufs.get(key)
        if b is None:
            mk = lambda rows: torch.zeros((rows, D), dtype=dtype, device=self.device)    # noqa: E731
            send = mk(self.slot)
            recv = mk(self.gather_rows) if (self.world > 1 or self.force) else send
            res = mk(self.gather_rows)
            lo = self.rank * self.slot
            b = dict(send=send, send_roots=send[: self.n_roots], recv=recv, res=res, mine=res[lo: lo + self.n_roots])
            if self.direct and self.dist is not None and send.is_cuda and (self.world > 1 or self.force):
                b["xchg"] = self._open_exchange(send, D, dtype)
            self._bufs[key] = b
        return b

    def _root_positions(self):
        if self._root_pos is None:
            self._root_pos = self.plan.inv_order[self.root_rows].contiguous() if self.plan is not None else self.root_rows
        return self._root_pos

    # ---- timing of the two collectives of a step (bench.py) ---------------------------------------
    def time_collectives(self, on):
        """on: record an event pair on the current stream around each all-gather of the steps that follow."""
        self._ev = {"fwd": [], "inv": []} if on else None

    def collective_ms(self):
        """-> (forward, inverse) mean milliseconds between the events recorded since time_collectives(True)."""
        torch.cuda.synchronize(self.device)
        out = []
        for k in ("fwd", "inv"):
            ev = self._ev[k]
            out.append(sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev)))
            ev.clear()
        return tuple(out)

    # ---- direct exchange (raht_xchg_*): blocks, handles, gathers -----------------------------------
    def _open_exchange(self, send, D, dtype):
        """COLLECTIVE, once per (D, dtype): allocate this rank's exchange block, swap the hipIpc handles through the process
        group, map the peers' blocks. -> dict(base, peers (C array), slot_bytes, seq, views of the two gather buffers)"""
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        slot_bytes = int(send.numel() * send.element_size())
        base, handle = C.c_void_p(), (C.c_ubyte * 64)()
        with torch.cuda.device(self.device):
            _lib.check(L.raht_xchg_alloc(self.world, slot_bytes, C.byref(base), handle))
        handles = [None] * self.world
        self.dist.all_gather_object(handles, bytes(handle), group=self.group)
        peers = (C.c_void_p * self.world)()
        for r in range(self.world):
            if r == self.rank:
                peers[r] = base.value
            else:
                hb, pb = (C.c_ubyte * 64).from_buffer_copy(handles[r]), C.c_void_p()
                with torch.cuda.device(self.device):
                    _lib.check(L.raht_xchg_open(hb, C.byref(pb)))
                peers[r] = pb.value
        typestr = {torch.float32: "<f4", torch.float64: "<f8"}[dtype]
        views = []
        for par in (0, 1):
            p = C.c_void_p()
            _lib.check(L.raht_xchg_buffer(base, self.world, slot_bytes, par, C.byref(p)))
            views.append(torch.as_tensor(_DeviceArray(p.value, (self.gather_rows, D), typestr), device=self.device))
        self.dist.barrier(group=self.group)                    # nobody writes into a block that is not mapped yet
        self._xchg_bases.append(base)
        return dict(base=base, peers=peers, slot_bytes=slot_bytes, seq=0, views=views)

    def _direct_gather(self, b):
        import ctypes as C
        from . import _lib
        if self._direct_failed:
            raise RuntimeError("ShardedRaht(direct=True): an earlier direct gather timed out waiting for a peer; the ranks are out of "
                               "phase and the double buffers no longer mean anything -- rebuild the object (or use the RCCL path)")
        x = b["xchg"]
        x["seq"] += 1
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().raht_xchg_gather(C.c_void_p(b["send"].data_ptr()), x["slot_bytes"], x["peers"], self.rank, self.world,
                                                   x["seq"] & 0xffffffff, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        b["recv"] = x["views"][x["seq"] & 1]

    def exchange_status(self):
        """(synchronises) 0 = every direct gather so far met all of its peers; 1 = a wait timed out"""
        import ctypes as C
        from . import _lib
        worst = 0
        for b in self._bufs.values():
            x = b.get("xchg")
            if x is None:
                continue
            st = C.c_int()
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().raht_xchg_status(x["base"], self.world, x["slot_bytes"], C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), C.byref(st)))
            worst = max(worst, st.value)
        if worst:
            self._direct_failed = True
        return worst

    def check_exchange(self):
        """(synchronises) Raise if a direct gather timed out: the transforms that followed it ran the top tree on a stale or
        partially written gather buffer and returned without an error (the wait is bounded so that a missing peer cannot
        hang the GPU; what it leaves behind is the status word this reads). Called by close(), roundtrip_error() and
        check_against_unsharded(); a caller of the direct path should call it wherever it synchronises anyway. After a
        timeout every further direct gather of this object raises."""
        if self.exchange_status():
            raise RuntimeError("ShardedRaht(direct=True): a direct gather timed out waiting for a peer (raht_xchg_status = 1): "
                               "results since then are not valid")

    def close(self):
        """COLLECTIVE: unmap the peers' exchange blocks and free this rank's (direct=True). A block must outlive every peer's
        last write into it, hence the barriers; nothing to do for the RCCL path."""
        from . import _lib
        if not any("xchg" in b for b in self._bufs.values()):
            return
        failed = bool(self.exchange_status())                  # (synchronises) reported after the blocks are gone
        torch.cuda.synchronize(self.device)
        self.dist.barrier(group=self.group)
        for b in self._bufs.values():
            x = b.pop("xchg", None)
            if x is None:
                continue
            b["recv"] = None
            for r in range(self.world):
                if r != self.rank:
                    _lib.check(_lib.lib().raht_xchg_close(x["peers"][r]))
        self.dist.barrier(group=self.group)
        self._bufs.clear()                                     # the views of the blocks go with them
        # (the blocks themselves: freed when every rank has unmapped them)
        for base in self._xchg_bases:
            _lib.check(_lib.lib().raht_xchg_free(base))
        self._xchg_bases = []
        if failed:
            raise RuntimeError("ShardedRaht(direct=True): a direct gather timed out waiting for a peer; results since then are not valid")

    def _gather_roots(self, b, which):
        gather = (lambda: self._direct_gather(b)) if "xchg" in b else (lambda: self._all_gather(b["send"], out=b["recv"]))
        if self._ev is None or not b["send"].is_cuda:
            return gather()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gather()
        e1.record()
        self._ev[which].append((e0, e1))

    def gathered_bytes_per_step(self, D, elem=4):
        """bytes every rank receives per direction x 2 directions"""
        return 2 * self.gather_rows * D * elem if self.world > 1 else 0

    # ---- transforms ------------------------------------------------------------------------------
    def _top_forward(self, b):
        self._gather_roots(b, "fwd")
        self.top.forward(b["recv"], want_w=False, out=b["res"])

    def _top_inverse(self, b):
        self._gather_roots(b, "inv")
        self.top.inverse(b["recv"], out=b["res"])

    # A rank without rows (self.plan is None) has nothing to transform but still enters both collectives: every
    # method below reaches _top_forward / _top_inverse whatever N is.
    def forward(self, C):
        """-> T_local: every row holds its coefficient of the WHOLE scene's RAHT."""
        b = self._buffers(C.shape[1], C.dtype)
        T = self.plan.forward(C, want_w=False, roots=b["send_roots"]) if self.plan is not None else torch.empty_like(C)
        self._top_forward(b)
        if self.n_roots:
            self.ops.rows_scatter(b["mine"], self.root_rows, T)
        return T

    def inverse(self, T):
        b = self._buffers(T.shape[1], T.dtype)
        if self.n_roots:
            self.ops.rows_gather(T, self.root_rows, b["send_roots"])
        self._top_inverse(b)
        return self.plan.inverse(T, roots=b["mine"]) if self.plan is not None else torch.empty_like(T)

    def forward_quant(self, C, step):
        """-> Q_local (int32, rank-local order_RAGFT order); the top coefficients are quantized too."""
        b = self._buffers(C.shape[1], self.qdt)
        Q = (self.plan.forward_quant(C, step, roots=b["send_roots"]) if self.plan is not None
             else torch.empty(C.shape, dtype=torch.int32, device=C.device))
        self._top_forward(b)
        if self.n_roots:
            self.ops.quant_rows(b["mine"], step, self._root_positions(), Q)
        return Q

    def dequant_inverse(self, Q, step):
        b = self._buffers(Q.shape[1], self.qdt)
        if self.n_roots:
            self.ops.dequant_rows(Q, step, self._root_positions(), b["send_roots"])
        self._top_inverse(b)
        if self.plan is None:
            return torch.empty(Q.shape, dtype=self.qdt, device=Q.device)
        return self.plan.dequant_inverse(Q, step, roots=b["mine"])

    # ---- bench helpers -----------------------------------------------------------------------------
    def step(self, C, quant_step=None):
        if quant_step is None:
            return self.inverse(self.forward(C))
        return self.dequant_inverse(self.forward_quant(C, quant_step), quant_step)

    def local_step(self, C, quant_step=None):
        """The shard-local part of step() alone (bench.py: what is left of a step without the exchange): truncated
        forward (+ quantize) and (dequantize +) inverse, roots through the send / result buffers, NO collective and
        no top tree. Its output is not a transform of anything; it exists to be timed."""
        if self.plan is None:
            return None
        dt = C.dtype if quant_step is None else self.qdt
        b = self._buffers(C.shape[1], dt)
        if quant_step is None:
            return self.plan.inverse(self.plan.forward(C, want_w=False, roots=b["send_roots"]), roots=b["mine"])
        return self.plan.dequant_inverse(self.plan.forward_quant(C, quant_step, roots=b["send_roots"]), quant_step, roots=b["mine"])

    def top_only(self, D, dtype=torch.float32):
        """The replicated top tree of both directions on whatever the buffers hold (bench.py: timed alone)."""
        b = self._buffers(D, dtype)
        self.top.forward(b["recv"], want_w=False, out=b["res"])
        self.top.inverse(b["recv"], out=b["res"])

    def roundtrip_error(self, C):
        R = self.inverse(self.forward(C))
        if self.direct:
            self.check_exchange()
        if self.N == 0:
            return 0.0
        return float(((R - C).abs().max() / C.abs().max()).item())

    def _gather_var(self, x):
        """one-off (verification): all-gather row blocks of different heights -> the concatenation in rank order"""
        if self.world == 1:
            return x
        n = self._all_gather(torch.tensor([[x.shape[0]]], dtype=torch.int64, device=x.device)).reshape(-1).tolist()
        m = int(max(n))
        pad = torch.zeros((m, x.shape[1]), dtype=x.dtype, device=x.device)
        pad[: x.shape[0]] = x
        allp = self._all_gather(pad)
        return torch.cat([allp[r * m: r * m + int(n[r])] for r in range(self.world)], dim=0)

    def check_against_unsharded(self, C, quant_step=None, keys_sorted=None):
        """Correctness gate for a multi-rank run (not on the timed path): gather the whole scene, transform it
        UNSHARDED on this rank, and compare this rank's rows with what the sharded path produced.
        float32: |dT| <= 4e-6 * column max (two float32 transforms, each within SURVEY 8c's 2e-6 of the float64 one: triangle
        inequality; measured 0.0 -- the replicated top tree performs the same butterflies on the same operands); integers: the same
        coefficient-error bound as bench.py's oracle gate."""
        keys = self.plan_keys() if keys_sorted is None else keys_sorted
        allk = self._gather_var(keys.reshape(-1, 1).to(torch.int64)).reshape(-1).contiguous()
        allC = self._gather_var(C)
        n_before = int(self._all_gather(torch.tensor([[self.N]], dtype=torch.int64, device=C.device)).reshape(-1)[: self.rank].sum().item()) if self.world > 1 else 0
        full = self.ops.make_plan(allk, self.nbits)
        Tf = full.forward(allC, want_w=False)
        mine = slice(n_before, n_before + self.N)
        T = self.forward(C)
        colmax = Tf.abs().amax(dim=0).clamp_min(1e-30)
        rel = float(((T - Tf[mine]).abs().amax(dim=0) / colmax).max().item()) if self.N else 0.0
        out = {"kind": "sharded == unsharded (whole scene gathered and transformed on every rank)", "rows_total": int(allk.shape[0]),
               "max_rel_err_T_vs_unsharded": rel, "ok": rel <= 4e-6}
        if quant_step is not None:
            Q = self.forward_quant(C, quant_step)
            R = self.dequant_inverse(Q, quant_step)
            if self.N:
                Tq = torch.empty_like(T)
                Tq[self.plan.order_RAGFT] = Q.to(T.dtype) * quant_step
                # dequantized integers sit within half a step (+ the coefficient error) of the unsharded coefficients
                worst = float(((Tq - Tf[mine]).abs() - 4e-6 * colmax).max().item())
                out["max_dequantized_distance_over_step"] = worst / quant_step
                out["ok"] = out["ok"] and worst <= 0.5 * quant_step * 1.0001
                out["quantized_roundtrip_max_err_over_step"] = float((R - C).abs().max().item()) / quant_step
        if self.direct:
            st = self.exchange_status()
            out["direct_exchange_status"] = st
            out["ok"] = out["ok"] and st == 0
        if self.world > 1:
            flag = torch.tensor([[1 if out["ok"] else 0]], dtype=torch.int64, device=C.device)
            out["ok"] = bool(self._all_gather(flag).min().item() == 1)
        return out

    def plan_keys(self):
        if self.plan is None:
            return torch.empty(0, dtype=torch.int64, device=self.device)
        ks = getattr(self.plan, "keys_tensor", None)
        if ks is None:
            raise ValueError("pass keys_sorted")
        return ks()


# ---------------------------------------------------------------------------------------------------
# front end for un-partitioned input: all-to-all by Morton prefix + local voxelizer
# ---------------------------------------------------------------------------------------------------
def exchange_by_prefix(PC, J, prefix_bits=9, vmin=None, width=None, group=None, local_ops=None):
    """Every rank holds an ARBITRARY part of the cloud: PC (n_r, 3 + d) float32, positions first. Returns this
    rank's Morton-prefix shard, voxelized: (PCvox (Nvox_r, 3 + d), keys (Nvox_r,) sorted unique int64, info).

    The concatenation of the shards over the ranks equals what the single-GPU voxelizer
    (``voxelize_pc_batched``, reference python/voxelize_pc.py:62-172) produces for the concatenation of the
    inputs in rank order, bit for bit: the bounding box is global, every voxel's points end up on one rank in
    their global order (the all-to-all delivers source ranks in order, the radix sort is stable), and the
    per-voxel mean is sequential in that order."""
    ops = local_ops or HipLocalOps
    dist = _dist()
    world = dist.get_world_size(group) if dist else 1
    dev = PC.device
    gloo_staged = bool(dist and PC.is_cuda and dist.get_backend(group) == "gloo")

    def allreduce(t, op):
        if world == 1:
            return t
        if gloo_staged:
            c = t.cpu(); dist.all_reduce(c, op=op, group=group); return c.to(dev)
        dist.all_reduce(t, op=op, group=group)
        return t

    # ---- global bounding box (voxelize_pc.py:87-95: per-axis minimum, ONE scalar width = max extent) ----
    xyz = PC[:, :3]
    if vmin is None:
        lo = xyz.amin(dim=0) if PC.shape[0] else torch.full((3,), float("inf"), device=dev)
        vmin_t = allreduce(lo.clone(), dist.ReduceOp.MIN if dist else None)
    else:
        vmin_t = torch.as_tensor(vmin, dtype=torch.float32, device=dev)
    if width is None:
        hi = (xyz - vmin_t).amax() if PC.shape[0] else torch.tensor(float("-inf"), device=dev)
        width = float(allreduce(hi.reshape(1).clone(), dist.ReduceOp.MAX if dist else None).item())
    vmin_l = [float(v) for v in vmin_t.tolist()]
    nbits = 3 * J
    # ---- keys, 2^prefix_bits-bin population histogram, balanced prefix cuts (same on every rank) ----
    keys = ops.voxel_keys(PC, vmin_l, width, J)
    nb = 1 << prefix_bits
    pref = (keys >> (nbits - prefix_bits)).clamp(0, nb - 1) if nbits > prefix_bits else keys.clamp(0, nb - 1)
    hist = allreduce(torch.bincount(pref, minlength=nb), dist.ReduceOp.SUM if dist else None)
    cuts = cuts_from_histogram(hist, world)                  # rank r owns prefixes [cuts[r], cuts[r+1])
    info = {"vmin": vmin_t, "width": width, "voxel_size": width / (1 << J), "prefix_cuts": cuts, "N_global": int(hist.sum().item())}
    if world == 1:
        mine = PC
    else:
        # ---- bucket the points by destination rank (stable), ONE all-to-all of the rows ----
        dest = torch.bucketize(pref, torch.tensor(cuts[1:-1], dtype=pref.dtype, device=dev), right=True)
        _, perm = ops.sort_keys(dest, max(1, (world - 1).bit_length()))      # stable: keeps the local order per bucket
        send = torch.empty_like(PC)
        ops.rows_gather(PC, perm, send)
        scnt = torch.bincount(dest, minlength=world)
        if gloo_staged:
            scnt_c = scnt.cpu(); rcnt_c = torch.empty_like(scnt_c)
            dist.all_to_all_single(rcnt_c, scnt_c, group=group)
            ss, rs = scnt_c.tolist(), rcnt_c.tolist()
            recv_c = torch.empty((sum(rs), PC.shape[1]), dtype=PC.dtype)
            dist.all_to_all_single(recv_c, send.cpu(), output_split_sizes=rs, input_split_sizes=ss, group=group)
            mine = recv_c.to(dev)
        else:
            rcnt = torch.empty_like(scnt)
            dist.all_to_all_single(rcnt, scnt, group=group)
            ss, rs = scnt.tolist(), rcnt.tolist()
            mine = torch.empty((sum(rs), PC.shape[1]), dtype=PC.dtype, device=dev)
            dist.all_to_all_single(mine, send, output_split_sizes=rs, input_split_sizes=ss, group=group)
        info["sent_rows"], info["received_rows"] = ss, rs
    if mine.shape[0] == 0:
        return mine, torch.empty(0, dtype=torch.int64, device=dev), info
    PCvox, vkeys, _, vinfo = ops.voxelize(mine, vmin_l, width, J)
    info["Nvox_local"] = int(PCvox.shape[0])
    return PCvox, vkeys, info
