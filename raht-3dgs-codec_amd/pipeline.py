"""Codec driver for voxelized 3DGS frames: the repo's counterpart of the reference's
python/encode_3dgs.py main loop (:126-411), logging the same 20-column CSV (:70-76, :402-409) so
that the reference's scripts/summarize_pipeline_runtime.py works on it unchanged.

    prelude (plan) -> forward RAHT -> quantize -> reorder -> [D2H] -> RLGR encode all channels ->
    RLGR decode -> [H2D] -> dequantize -> un-reorder -> inverse RAHT -> PSNR (all / quats / scales /
    opacity / colours) -> CSV row per (frame, step)

``fused=True`` (default) uses raht_fwd_quant / raht_dequant_inv: the Quant / Coeff_reorder columns
are then 0 and their work is inside RAHT_transform_time / iRAHT_time. ``fused=False`` keeps the
reference's stage boundaries one to one. Everything on the transform side runs through the C ABI on
the GPU; the entropy stage is the byte-exact host coder (csrc/rlgr.hip).
"""
import math
import os
import time

import numpy as np
import torch

from . import rlgr as rlgr_mod
from .ops import RAHT_param_reorder_fast, plan_of, raht_fn

CSV_HEADER = ("Frame,Quantization_Step,Rate_bpp,"
              "RAHT_prelude_time,RAHT_transform_time,Quant_time,"
              "Coeff_reorder_enc_time,Entropy_enc_time,"
              "Entropy_dec_time,Dequant_time,"
              "Coeff_reorder_dec_time,iRAHT_time,"
              "Total_enc_time,Total_dec_time,Pipeline_time,"
              "PSNR_all,PSNR_quats,PSNR_scales,PSNR_opacity,PSNR_colors")          # encode_3dgs.py:70-76


def _sync():
    torch.cuda.synchronize()


def _psnr(a, b):
    return -10 * math.log10(torch.mean((a - b) ** 2).item() + 1e-10)               # encode_3dgs.py:298-310


def encode_frame(V_int, attributes, J, steps, frame=1, device="cuda:0", dtype=torch.float32, fused=True,
                 nthreads=0, channel_major=True, overlap=False, entropy="host", seg_len=2048, keep_rec=True, batch_steps=False):
    """One frame through the whole pipeline. Returns a list of dict rows (one per step) with the CSV
    columns plus ``size_bytes`` and ``C_rec`` (last step) for inspection, and the stages the reference's CSV has no
    column for (``Transpose_time``, ``D2H_time``, ``H2D_time``, ``PSNR_time``, ``Step_wall_time``).

    overlap=True (fused path): the steps are software-pipelined -- while the host threads entropy-code step s, the GPU
    already transforms / quantizes / copies step s + 1 and decodes step s - 1 (the reference serialises all of it,
    python/encode_3dgs.py:199-275). Same rows, same bytes; ``Step_wall_time`` is then the pipeline's period.

    keep_rec=False (fused float32 path): the reconstruction is never written -- the decode side is ONE pass that dequantizes,
    inverts and accumulates the PSNR columns' sums of squares (raht_dequant_inv_sqdiff); rows then carry ``C_rec = None``.
    entropy="gpu": the RLGR stage on the device, segmented (rlgr.SegmentedCoder: every ``seg_len`` symbols of a channel an
    independent stream, byte-identical to the reference coder's output for that slice): the integers never leave the GPU, only
    the container's bytes do (``D2H_time``). Rates then count the container (streams + 4 bytes per segment).
    batch_steps=True (entropy="gpu", float32, scalar steps): ALL steps of the frame at once -- one forward transform with every
    step's quantizer (raht_fwd_quant_multi), one set of coder launches for all steps' integers (raht_rlgr_seg_encode_batch: k times
    the independent streams, which is what the coder's speed depends on), one decoder launch; same rows, same bytes. Stage times of
    the shared launches are split evenly over the rows, which carry ``Batched_steps``."""
    if entropy == "gpu":
        if not fused:
            raise ValueError("entropy='gpu' goes with the fused path")
        scalar = all(isinstance(st, (int, float)) for st in steps)
        if batch_steps and scalar and dtype == torch.float32 and len(steps) > 1:
            return _encode_frame_gpu_entropy_batched(V_int, attributes, J, steps, frame, device, seg_len, keep_rec)
        return _encode_frame_gpu_entropy(V_int, attributes, J, steps, frame, device, dtype, seg_len, keep_rec)
    if overlap and fused:
        return _encode_frame_overlapped(V_int, attributes, J, steps, frame, device, dtype, nthreads, keep_rec)
    N = V_int.shape[0]
    C = attributes.to(dtype=dtype).contiguous().to(device)
    _sync()
    V = V_int.to(dtype=torch.float64).to(device)                                    # :139-142
    origin = torch.tensor([0, 0, 0], dtype=V.dtype, device=device)

    t0 = time.time()
    ListC, FlagsC, weightsC, order_RAGFT = RAHT_param_reorder_fast(V, origin, 2 ** J, J)   # :149
    _sync()
    t_prelude = time.time() - t0
    plan = plan_of(ListC)
    use_fused = fused                     # float32 and float64 (the reference's precision) both have fused entry points

    Coeff, t_transform = None, 0.0
    if not use_fused:
        t0 = time.time()
        Coeff, _ = raht_fn["RAHT"](C, ListC, FlagsC, weightsC)                      # :159
        _sync()
        t_transform = time.time() - t0

    rows = []
    coder = None
    for step in steps:
        # a scalar step (the drivers' colorStep entries) or one step per channel (per_attribute_steps below)
        per_channel = not isinstance(step, (int, float))
        if per_channel:
            step_list = [float(x) for x in (step.tolist() if hasattr(step, "tolist") else step)]
            step_t = torch.tensor(step_list, dtype=dtype, device=device)           # tensor divisor: true division on the GPU
            step_arg = step_list
        else:
            step_t, step_arg = step, float(step)
        r = dict(Frame=frame, Quantization_Step="per_attribute" if per_channel else step)
        step = step_t
        t_step0 = time.time()
        # ---------------- encoder ----------------
        if use_fused:
            t0 = time.time()
            coeff_reordered = plan.forward_quant(C, step_arg)                       # :159 + :204 + :210 + :215
            _sync()
            r["RAHT_transform_time"], r["Quant_time"], r["Coeff_reorder_enc_time"] = time.time() - t0, 0.0, 0.0
        else:
            t0 = time.time()
            Coeff_enc = torch.floor(Coeff / step + 0.5)                             # :204
            _sync()
            r["Quant_time"] = time.time() - t0
            t0 = time.time()
            coeff_reordered = Coeff_enc.index_select(0, order_RAGFT).to(torch.int32)   # :210, :215
            _sync()
            r["Coeff_reorder_enc_time"] = time.time() - t0
            r["RAHT_transform_time"] = t_transform
        # device -> host, channel-major so that the entropy coder reads contiguous channels
        t0 = time.time()
        q_dev = rlgr_mod.transpose_on_device(coeff_reordered) if channel_major else coeff_reordered
        _sync()
        r["Transpose_time"] = time.time() - t0
        t0 = time.time()
        q_cpu = rlgr_mod.to_host(q_dev)                                             # :215-217 (pinned staging)
        r["D2H_time"] = time.time() - t0
        if channel_major:
            # all D channels on the host threads, streams in one arena, decoded straight into a page-locked upload buffer
            if coder is None:
                coder = rlgr_mod.ChannelCoder(N, q_cpu.shape[0], 1, nthreads)
            r["Entropy_enc_time"] = coder.encode(q_cpu)                             # :229-234
            size_bytes = coder.size_bytes                                           # :247
            up = rlgr_mod.pinned_like(q_cpu.shape, torch.int32, "up")
            q_back = up.numpy()
            r["Entropy_dec_time"] = coder.decode(q_back)                            # :237-241
            t0 = time.time()
            assert rlgr_mod.arrays_equal(q_back, q_cpu, nthreads), "RLGR roundtrip failed"   # :242-245
            r["Roundtrip_check_time"] = time.time() - t0
            t0 = time.time()
            qd = up.to(device, non_blocking=True)
            _sync()
            r["H2D_time"] = time.time() - t0
        else:
            streams, t_enc = rlgr_mod.encode_channels(q_cpu, 1, nthreads=nthreads, channel_major=channel_major)   # :229-234
            size_bytes = sum(int(s.shape[0]) for s in streams)                          # :247
            r["Entropy_enc_time"] = t_enc
            q_back, t_dec = rlgr_mod.decode_channels(streams, N, 1, nthreads=nthreads, channel_major=channel_major)   # :237-245
            assert np.array_equal(q_back, q_cpu), "RLGR roundtrip failed"               # :242-245
            r["Entropy_dec_time"] = t_dec
            t0 = time.time()
            qd = torch.from_numpy(q_back).to(device)
            _sync()
            r["H2D_time"] = time.time() - t0
        if channel_major:
            t0 = time.time()
            qd = rlgr_mod.transpose_on_device(qd)
            _sync()
            r["Transpose_time"] += time.time() - t0
        measured = False
        if use_fused:
            t0 = time.time()
            C_rec = _decode_and_measure(plan, qd, step_arg, C, dtype, r, keep_rec)  # :261 + :267-268 + :274 (+ :298-310 in the same pass)
            _sync()
            r["iRAHT_time"], r["Dequant_time"], r["Coeff_reorder_dec_time"] = time.time() - t0, 0.0, 0.0
            measured = True
        else:
            t0 = time.time()
            Coeff_dec = qd.to(dtype) * step                                         # :261
            _sync()
            r["Dequant_time"] = time.time() - t0
            t0 = time.time()
            Coeff_dec = Coeff_dec[torch.argsort(order_RAGFT), :]                     # :267-268
            _sync()
            r["Coeff_reorder_dec_time"] = time.time() - t0
            t0 = time.time()
            C_rec = raht_fn["iRAHT"](Coeff_dec, ListC, FlagsC, weightsC)            # :274
            _sync()
            r["iRAHT_time"] = time.time() - t0
        # ---------------- bookkeeping (:279-310) ----------------
        r["RAHT_prelude_time"] = t_prelude
        r["Total_enc_time"] = r["RAHT_transform_time"] + r["Quant_time"] + r["Coeff_reorder_enc_time"] + r["Entropy_enc_time"]
        r["Total_dec_time"] = r["Entropy_dec_time"] + r["Dequant_time"] + r["Coeff_reorder_dec_time"] + r["iRAHT_time"]
        r["Pipeline_time"] = t_prelude + r["Total_enc_time"] + r["Total_dec_time"]
        r["Rate_bpp"] = size_bytes * 8 / N                                          # :403
        r["size_bytes"] = size_bytes
        t0 = time.time()
        if not measured:
            _psnr_columns(r, C, C_rec)
        r["PSNR_time"] = time.time() - t0
        r["Step_wall_time"] = time.time() - t_step0
        r["C_rec"] = C_rec
        rows.append(r)
    return rows


def _encode_frame_gpu_entropy(V_int, attributes, J, steps, frame, device, dtype, seg_len, keep_rec=True):
    """encode_frame with the entropy stage on the device (entropy="gpu"): forward RAHT + quantize + reorder -> segmented RLGR
    encode of the row-major integers (device) -> [container bytes to the host: the codec's output] -> segmented RLGR decode, row-major
    (device) -> round-trip check (device) -> dequantize + un-reorder + inverse RAHT -> PSNR. No transpose on either side."""
    N = V_int.shape[0]
    dev = torch.device(device)
    C = attributes.to(dtype=dtype).contiguous().to(dev)
    _sync()
    V = V_int.to(dtype=torch.float64).to(dev)
    origin = torch.tensor([0, 0, 0], dtype=V.dtype, device=dev)
    t0 = time.time()
    ListC, FlagsC, weightsC, order_RAGFT = RAHT_param_reorder_fast(V, origin, 2 ** J, J)
    _sync()
    t_prelude = time.time() - t0
    plan = plan_of(ListC)
    coder = rlgr_mod.SegmentedCoder(N, C.shape[1], seg_len, 1, dev)
    rows = []
    for step in steps:
        per_channel = not isinstance(step, (int, float))
        step_arg = [float(x) for x in (step.tolist() if hasattr(step, "tolist") else step)] if per_channel else float(step)
        r = dict(Frame=frame, Quantization_Step="per_attribute" if per_channel else step)
        t_step0 = time.time()
        t0 = time.time()
        coeff_reordered = plan.forward_quant(C, step_arg)
        _sync()
        r["RAHT_transform_time"], r["Quant_time"], r["Coeff_reorder_enc_time"] = time.time() - t0, 0.0, 0.0
        q_dev = coeff_reordered                               # row-major, as the transform leaves it: the coder's lanes are neighbouring
        r["Transpose_time"] = 0.0                             # channels, every step of a wave reads one piece of a row (no transpose)
        t0 = time.time()
        coder.encode(q_dev)                                   # (returns the size: synchronises)
        r["Entropy_enc_time"] = time.time() - t0
        t0 = time.time()
        hdr, lens, payload = coder.container_parts()          # what goes on the wire (header, length table, streams): page-locked D2H
        r["D2H_time"] = time.time() - t0
        size_bytes = len(hdr) + lens.nbytes + payload.nbytes
        assert size_bytes == coder.size_bytes
        t0 = time.time()
        qd = coder.decode(row_major=True)                     # row-major, as the inverse takes it: the symbol-synchronous decoder (every
        _sync()                                               # iteration of a wave is symbol i of all its lanes = one piece of a row):
        r["Entropy_dec_time"] = time.time() - t0              # 2.1 ms against 2.9 + 0.3 ms for channel-major + transpose
        t0 = time.time()
        assert torch.equal(qd, q_dev) and int(coder.bad.item()) == 0, "RLGR roundtrip failed"    # encode_3dgs.py:242-245
        r["Roundtrip_check_time"] = time.time() - t0
        r["H2D_time"] = 0.0
        t0 = time.time()
        C_rec = _decode_and_measure(plan, qd, step_arg, C, dtype, r, keep_rec)      # inverse + the PSNR columns' sums, one pass
        _sync()
        r["iRAHT_time"], r["Dequant_time"], r["Coeff_reorder_dec_time"] = time.time() - t0, 0.0, 0.0
        r["RAHT_prelude_time"] = t_prelude
        r["Total_enc_time"] = r["RAHT_transform_time"] + r["Entropy_enc_time"]
        r["Total_dec_time"] = r["Entropy_dec_time"] + r["iRAHT_time"]
        r["Pipeline_time"] = t_prelude + r["Total_enc_time"] + r["Total_dec_time"]
        r["Rate_bpp"] = size_bytes * 8 / N
        r["size_bytes"] = size_bytes
        r["PSNR_time"] = 0.0                                   # (inside iRAHT_time: the fused inverse measures on its way out)
        r["Step_wall_time"] = time.time() - t_step0
        r["C_rec"] = C_rec
        rows.append(r)
    return rows


def _encode_frame_gpu_entropy_batched(V_int, attributes, J, steps, frame, device, seg_len, keep_rec=True):
    """encode_frame(entropy="gpu", batch_steps=True): the loop of python/encode_3dgs.py:199-275 turned inside out -- every stage
    runs ONCE for all steps (in chunks of the 12 a call takes): forward RAHT with k quantizers -> k containers from one set of
    coder launches -> [container bytes to the host] -> k frames decoded by one launch -> round-trip check -> per step the fused
    dequantize + inverse RAHT + PSNR sums."""
    N = V_int.shape[0]
    dev = torch.device(device)
    dtype = torch.float32
    C = attributes.to(dtype=dtype).contiguous().to(dev)
    _sync()
    V = V_int.to(dtype=torch.float64).to(dev)
    origin = torch.tensor([0, 0, 0], dtype=V.dtype, device=dev)
    t0 = time.time()
    ListC, FlagsC, weightsC, order_RAGFT = RAHT_param_reorder_fast(V, origin, 2 ** J, J)
    _sync()
    t_prelude = time.time() - t0
    plan = plan_of(ListC)
    D = C.shape[1]
    rows = []
    KMAX = rlgr_mod.SegmentedCoder.BATCH_MAX
    for lo in range(0, len(steps), KMAX):
        chunk = [float(x) for x in steps[lo: lo + KMAX]]
        k = len(chunk)
        t_chunk0 = time.time()
        t0 = time.time()
        Qs = plan.forward_quant_multi(C, chunk)                 # one forward pass, k quantizations (row-major (N, D) each)
        _sync()
        t_fwd = time.time() - t0
        coders = [rlgr_mod.SegmentedCoder(N, D, seg_len, 1, dev) for _ in chunk]
        _sync()
        t0 = time.time()
        rlgr_mod.SegmentedCoder.encode_batch(coders, Qs)        # (returns the sizes: synchronises)
        t_enc = time.time() - t0
        t0 = time.time()
        sizes = []
        for c in coders:                                        # what goes on the wire, frame by frame: page-locked D2H
            hdr, lens, payload = c.container_parts()
            sizes.append(len(hdr) + lens.nbytes + payload.nbytes)
            assert sizes[-1] == c.size_bytes
        t_d2h = time.time() - t0
        t0 = time.time()
        # row-major (see _encode_frame_gpu_entropy: no transpose behind it), and the round-trip assertion of encode_3dgs.py:242-245
        # inside the decoder: every decoded symbol is compared with what was encoded on its way out (raht_rlgr_seg_decode_batch_check)
        qds = rlgr_mod.SegmentedCoder.decode_batch(coders, row_major=True, expect=Qs)
        _sync()
        t_dec = time.time() - t0
        t_tr = 0.0
        t0 = time.time()
        assert not rlgr_mod.SegmentedCoder.roundtrip_failed(coders), "RLGR roundtrip failed"
        t_chk = time.time() - t0
        del Qs, coders
        t_shared = time.time() - t_chunk0
        for j, step in enumerate(chunk):
            r = dict(Frame=frame, Quantization_Step=steps[lo + j], Batched_steps=k)
            t_step0 = time.time()
            r["RAHT_transform_time"], r["Quant_time"], r["Coeff_reorder_enc_time"] = t_fwd / k, 0.0, 0.0
            r["Transpose_time"], r["Entropy_enc_time"], r["D2H_time"], r["Entropy_dec_time"] = t_tr / k, t_enc / k, t_d2h / k, t_dec / k
            r["Roundtrip_check_time"], r["H2D_time"] = t_chk / k, 0.0
            t0 = time.time()
            C_rec = _decode_and_measure(plan, qds[j], step, C, dtype, r, keep_rec)
            _sync()
            r["iRAHT_time"], r["Dequant_time"], r["Coeff_reorder_dec_time"] = time.time() - t0, 0.0, 0.0
            qds[j] = None
            r["RAHT_prelude_time"] = t_prelude
            r["Total_enc_time"] = r["RAHT_transform_time"] + r["Entropy_enc_time"]
            r["Total_dec_time"] = r["Entropy_dec_time"] + r["iRAHT_time"]
            r["Pipeline_time"] = t_prelude + r["Total_enc_time"] + r["Total_dec_time"]
            r["Rate_bpp"] = sizes[j] * 8 / N
            r["size_bytes"] = sizes[j]
            r["PSNR_time"] = 0.0
            r["Step_wall_time"] = t_shared / k + (time.time() - t_step0)
            r["C_rec"] = C_rec if (keep_rec and lo + j == len(steps) - 1) else None   # (k reconstructions of 0.7 GB each: the last one only)
            rows.append(r)
    return rows


def _psnr_columns(r, C, C_rec):
    """The five PSNR columns of encode_3dgs.py:298-310 from ONE pass over C and C_rec: per-column sums of squared differences on
    the device (raht_sqdiff_columns, float64 sums), grouped on the host -- instead of five torch.mean reductions with a host round
    trip each (2.0 -> 0.2 ms per step on a 3 M x 56 frame)."""
    if C.is_cuda and C.dtype in (torch.float32, torch.float64) and C.dim() == 2 and C.shape == C_rec.shape and C.dtype == C_rec.dtype \
            and C.stride(1) == 1 and C_rec.stride(1) == 1:
        import ctypes as _C
        from . import _lib
        N, D = C.shape
        out = torch.empty(D, dtype=torch.float64, device=C.device)
        with torch.cuda.device(C.device):
            _lib.check(_lib.lib().raht_sqdiff_columns(_C.c_void_p(C.data_ptr()), C.stride(0), _C.c_void_p(C_rec.data_ptr()), C_rec.stride(0), N, D,
                                                      _lib.RAHT_F32 if C.dtype == torch.float32 else _lib.RAHT_F64, _C.c_void_p(out.data_ptr()),
                                                      _C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        _psnr_from_ssd(r, out, N)
        return
    r["PSNR_all"] = _psnr(C, C_rec)                                                 # encode_3dgs.py:298-310
    r["PSNR_quats"] = _psnr(C[:, 0:4], C_rec[:, 0:4])
    r["PSNR_scales"] = _psnr(C[:, 4:7], C_rec[:, 4:7])
    r["PSNR_opacity"] = _psnr(C[:, 7], C_rec[:, 7])
    r["PSNR_colors"] = _psnr(C[:, 8:], C_rec[:, 8:])


def _psnr_from_ssd(r, ssd_dev, N):
    """the five PSNR columns from per-column sums of squared differences (float64, on the device); an empty column group (frames
    with fewer than 9 attribute columns) gives nan, as torch.mean of an empty slice does in the reference (encode_3dgs.py:298-310)"""
    ssd = ssd_dev.cpu().numpy()                               # (the one synchronisation)
    D = ssd.shape[0]

    def col(lo, hi):
        hi = min(hi, D)
        if hi <= lo:
            return float("nan")
        return -10 * math.log10(float(ssd[lo:hi].sum()) / (N * (hi - lo)) + 1e-10)
    r["PSNR_all"] = col(0, D)
    r["PSNR_quats"], r["PSNR_scales"], r["PSNR_opacity"], r["PSNR_colors"] = col(0, 4), col(4, 7), col(7, 8), col(8, D)


def _decode_and_measure(plan, qd, step_arg, C, dtype, r, keep_rec=True):
    """dequantize + un-reorder + inverse RAHT + the five PSNR columns (encode_3dgs.py:261-275, 298-310). float32 frames: ONE fused
    pass (raht_dequant_inv_sqdiff: the inverse's stage-0 kernel compares its rows with C on the way out; C_rec is written only when
    asked for); float64: the float64 fused inverse, then one pass over C and C_rec."""
    if dtype == torch.float32 and C.is_cuda and C.dtype == torch.float32:
        C_rec, ssd = plan.dequant_inverse_sqdiff(qd, step_arg, C, want_rec=keep_rec)
        _psnr_from_ssd(r, ssd, C.shape[0])
        return C_rec
    C_rec = plan.dequant_inverse(qd, step_arg, dtype=dtype)
    _psnr_columns(r, C, C_rec)
    return C_rec


def _encode_frame_overlapped(V_int, attributes, J, steps, frame, device, dtype, nthreads, keep_rec=True):
    """The same frame, software-pipelined over the quantization steps (encode_frame, overlap=True).

    GPU (this thread):   step s: forward RAHT + quantize + reorder (fused) -> channel-major transpose -> device-to-host copy into
                         a page-locked buffer on a copy stream;   step s - 1 (once its bytes are back): host-to-device copy ->
                         transpose -> dequantize + un-reorder + inverse RAHT (fused) -> PSNR
    host (worker thread; the coder itself runs the channels on all cores): entropy-code step s, decode it again, check the
                         round trip (encode_3dgs.py:229-245), hand the decoded integers back in a second page-locked buffer
    Two buffers per direction: step s + 2 may only reuse what step s has finished with, which the order of this loop
    guarantees. What the reference runs back to back per step -- GPU stages, copies, coder -- costs max(host, GPU) here."""
    from concurrent.futures import ThreadPoolExecutor
    N = V_int.shape[0]
    dev = torch.device(device)
    C = attributes.to(dtype=dtype).contiguous().to(dev)
    _sync()
    V = V_int.to(dtype=torch.float64).to(dev)
    origin = torch.tensor([0, 0, 0], dtype=V.dtype, device=dev)
    t0 = time.time()
    ListC, FlagsC, weightsC, order_RAGFT = RAHT_param_reorder_fast(V, origin, 2 ** J, J)
    _sync()
    t_prelude = time.time() - t0
    plan = plan_of(ListC)
    D = C.shape[1]
    pin_dn = [torch.empty((D, N), dtype=torch.int32, pin_memory=True) for _ in range(2)]
    pin_up = [torch.empty((D, N), dtype=torch.int32, pin_memory=True) for _ in range(2)]
    copy_stream = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    steps = list(steps)
    n = len(steps)

    def step_args(step):
        if isinstance(step, (int, float)):
            return float(step), step
        lst = [float(x) for x in (step.tolist() if hasattr(step, "tolist") else step)]
        return lst, "per_attribute"

    coder = rlgr_mod.ChannelCoder(N, D, 1, nthreads)

    def host_job(i, ev_dn):
        ev_dn.synchronize()                                   # step i's integers have arrived
        q_cpu = pin_dn[i % 2].numpy()
        t_enc = coder.encode(q_cpu)
        size_bytes = coder.size_bytes
        q_back = pin_up[i % 2].numpy()
        t_dec = coder.decode(q_back)                          # straight into the page-locked upload buffer
        assert rlgr_mod.arrays_equal(q_back, q_cpu, nthreads), "RLGR roundtrip failed"   # encode_3dgs.py:242-245
        return size_bytes, t_enc, t_dec, time.time()

    rows = [None] * n
    jobs = [None] * n
    t_enc_gpu = [0.0] * n
    t_mark = time.time()
    with ThreadPoolExecutor(max_workers=1) as pool:
        for i in range(n + 1):
            if i < n:
                sa, _ = step_args(steps[i])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main)
                q_dev = rlgr_mod.transpose_on_device(plan.forward_quant(C, sa))
                e1.record(main)
                copy_stream.wait_stream(main)
                with torch.cuda.stream(copy_stream):
                    pin_dn[i % 2].copy_(q_dev, non_blocking=True)
                    q_dev.record_stream(copy_stream)
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                jobs[i] = pool.submit(host_job, i, ev)
                t_enc_gpu[i] = (e0, e1)
            if i >= 1:
                k = i - 1
                size_bytes, t_enc, t_dec, t_done = jobs[k].result()
                sa, label = step_args(steps[k])
                r = dict(Frame=frame, Quantization_Step=label if label == "per_attribute" else steps[k])
                f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.time()
                qd = pin_up[k % 2].to(dev, non_blocking=True)
                f0.record(main)
                qt = rlgr_mod.transpose_on_device(qd)
                C_rec = _decode_and_measure(plan, qt, sa, C, dtype, r, keep_rec)     # (synchronises: reads the sums back)
                f1.record(main)
                f1.synchronize()
                t_gpu_dec = time.time() - t0
                e0, e1 = t_enc_gpu[k]
                r["RAHT_transform_time"] = e0.elapsed_time(e1) * 1e-3          # forward + quantize + reorder + transpose (GPU time)
                r["iRAHT_time"] = f0.elapsed_time(f1) * 1e-3
                r["Quant_time"] = r["Coeff_reorder_enc_time"] = r["Dequant_time"] = r["Coeff_reorder_dec_time"] = 0.0
                r["Entropy_enc_time"], r["Entropy_dec_time"] = t_enc, t_dec
                r["RAHT_prelude_time"] = t_prelude
                r["Total_enc_time"] = r["RAHT_transform_time"] + r["Entropy_enc_time"]
                r["Total_dec_time"] = r["Entropy_dec_time"] + r["iRAHT_time"]
                now = time.time()
                r["Step_wall_time"] = now - t_mark             # the pipeline's period: what a step costs end to end here
                t_mark = now
                r["Pipeline_time"] = t_prelude + r["Step_wall_time"]
                r["GPU_decode_side_time"] = t_gpu_dec
                r["Rate_bpp"] = size_bytes * 8 / N
                r["size_bytes"] = size_bytes
                r["C_rec"] = C_rec
                rows[k] = r
    return rows


CSV_HEADER_PLY = ("Frame,Quantization_Step,Rate_bpp,RAHT_prelude_time,RAHT_transform_time,Quant_time,"
                  "Entropy_enc_time,Entropy_dec_time,Dequant_time,iRAHT_time,psnr")           # encode_ply.py:57


def rgb_to_yuv(rgb):
    """RGB -> YUV, BT.709 full range with 128/255 chroma offsets, clipped to [0, 255], float64
    (same arithmetic as reference python/utils.py:4-33)."""
    if rgb.ndim != 2 or rgb.shape[1] != 3:
        raise ValueError("Expected Nx3 tensor")
    rgb = rgb.to(torch.float64)
    rgb1 = torch.hstack((rgb / 255.0, torch.ones((rgb.size(0), 1), dtype=torch.float64, device=rgb.device)))
    Q = torch.tensor([[0.21260000, -0.114572, 0.5],
                      [0.71520000, -0.385428, -0.454153],
                      [0.07220000, 0.5, -0.045847],
                      [0.0, 0.50196078, 0.50196078]], dtype=torch.float64, device=rgb.device)
    return torch.clamp(rgb1 @ Q, 0.0, 1.0) * 255.0


def encode_ply_frame(V_int, Crgb, J, steps=(1, 2, 4, 6, 8, 12, 16, 20, 24, 32, 64), frame=1, device="cuda:0",
                     dtype=torch.float64, nthreads=0):
    """RGB point cloud counterpart of the reference's encode_ply.py loop (:102-222): YUV, plan,
    forward RAHT, per step quantize / Y-PSNR from the coefficients (:148-152, Parseval) / reorder /
    RLGR Y,U,V / dequantize / un-reorder + inverse RAHT. Returns dict rows (CSV columns of :57 plus
    ``size_bytes`` and ``C_rec``). float64 by default like the reference (:83 of encode_3dgs.py)."""
    N = V_int.shape[0]
    Cyuv = rgb_to_yuv(Crgb.to(torch.float64)).to(dtype).contiguous()
    C = Cyuv.to(device)
    V = V_int.to(dtype=torch.float64).to(device)
    origin = torch.tensor([0, 0, 0], dtype=V.dtype, device=device)
    t0 = time.time()
    ListC, FlagsC, weightsC, order_RAGFT = RAHT_param_reorder_fast(V, origin, 2 ** J, J)
    _sync()
    t_prelude = time.time() - t0
    t0 = time.time()
    Coeff, _ = raht_fn["RAHT"](C, ListC, FlagsC, weightsC)
    _sync()
    t_transform = time.time() - t0
    rows = []
    for step in steps:
        r = dict(Frame=frame, Quantization_Step=step, RAHT_prelude_time=t_prelude, RAHT_transform_time=t_transform)
        t0 = time.time()
        Coeff_enc = torch.floor(Coeff / step + 0.5)                                 # :148
        _sync()
        r["Quant_time"] = time.time() - t0
        Y_hat = Coeff_enc[:, 0] * step
        mse = (torch.linalg.norm(Coeff[:, 0].double() - Y_hat.double()) ** 2) / (N * 255 ** 2)     # :150-151
        r["psnr"] = float(-10 * torch.log10(mse))
        q_dev = Coeff_enc.index_select(0, order_RAGFT).to(torch.int32)              # :156-157
        q_cpu = rlgr_mod.to_host(rlgr_mod.transpose_on_device(q_dev))
        streams, r["Entropy_enc_time"] = rlgr_mod.encode_channels(q_cpu, 1, nthreads=nthreads, channel_major=True)   # :164-176
        q_back, r["Entropy_dec_time"] = rlgr_mod.decode_channels(streams, N, 1, nthreads=nthreads, channel_major=True)
        assert np.array_equal(q_back, q_cpu), "RLGR roundtrip failed"               # :184-187
        r["size_bytes"] = sum(int(s.shape[0]) for s in streams)
        r["Rate_bpp"] = r["size_bytes"] * 8 / N
        qd = rlgr_mod.transpose_on_device(torch.from_numpy(q_back).to(device))
        t0 = time.time()
        Coeff_dec = qd.to(dtype) * step                                             # :202
        _sync()
        r["Dequant_time"] = time.time() - t0
        t0 = time.time()
        Coeff_dec = Coeff_dec[torch.argsort(order_RAGFT), :]                         # :206-208
        r["C_rec"] = raht_fn["iRAHT"](Coeff_dec, ListC, FlagsC, weightsC)
        _sync()
        r["iRAHT_time"] = time.time() - t0
        rows.append(r)
    return rows


def format_row_ply(r):
    """One CSV line, same columns as encode_ply.py:217-220."""
    return (f"{r['Frame']},{r['Quantization_Step']},{r['Rate_bpp']:.6f},{r['RAHT_prelude_time']:.6f},{r['RAHT_transform_time']:.6f},"
            f"{r['Quant_time']:.6f},{r['Entropy_enc_time']:.6f},"
            f"{r['Entropy_dec_time']:.6f},{r['Dequant_time']:.6f},{r['iRAHT_time']:.6f},{r['psnr']:.6f}")


def compress_to_nvox(means, quats, scales, opacities, colors, J=10, device="cuda:0", output_ply=None, fused=True):
    """N Gaussians -> Nvox voxelized Gaussians: counterpart of the reference's compress_to_nvox
    (python/test_voxelize_3dgs.py:160-288): voxelize the means (:203), use the sort permutation and
    the voxel starts as cluster indices / offsets (:225-233), merge all attributes per voxel with
    opacity weights (:247-257), and optionally save the voxelized frame with integer voxel
    coordinates and the voxel_size / vmin header comments (:281-288) -- the input format of
    ``encode_3dgs``. Returns (V_int int64 (Nvox,3), attributes float32 (Nvox, 8 + color_dim), info).

    fused=True: ONE call, one pass over the rows (``ops.voxelize_merge`` / raht_voxelize_merge: the kernel that walks every voxel's
    members merges them on the way); fused=False: the reference's two steps (voxelizer, then the merge kernel). Same bits."""
    from .merge import merge_gaussian_clusters_with_indices
    from .ops import voxelize_merge, voxelize_pc_batched
    from .ply_io import save_ply
    means = means.to(device).float().contiguous()
    N = means.shape[0]
    if fused:
        G = torch.cat([means, quats.to(device).float(), scales.to(device).float(), opacities.to(device).float().reshape(-1, 1),
                       colors.to(device).float().reshape(N, -1)], dim=1).contiguous()
        Gvox, info = voxelize_merge(G, J=J, device=device, weight_by_opacity=True)
        V_int = Gvox[:, :3].long()                                                 # :267
        attributes = Gvox[:, 3:]                                                   # layout of data_util.py:366
        mq, ms, mo, mc = Gvox[:, 3:7], Gvox[:, 7:10], Gvox[:, 10], Gvox[:, 11:]
        if output_ply is not None:
            save_ply(output_ply, Gvox[:, :3], mq, ms, mo, mc, voxel_size=info["voxel_size"], vmin=info["vmin"])
        cluster_offsets = torch.cat([info["voxel_indices"], torch.tensor([N], dtype=torch.int64, device=means.device)]).int()
        info = dict(info, cluster_indices=info["sort_idx"].int(), cluster_offsets=cluster_offsets)
        return V_int, attributes, info
    PCvox, _, voxel_indices, _, info = voxelize_pc_batched(means, J=J, device=device, residuals=False, sorted_points=False)
    cluster_indices = info["sort_idx"].int()                                       # :225-226
    cluster_offsets = torch.cat([voxel_indices, torch.tensor([N], dtype=torch.int64, device=means.device)]).int()   # :230-233
    mm, mq, ms, mo, mc = merge_gaussian_clusters_with_indices(means, quats.to(device), scales.to(device),
                                                              opacities.to(device), colors.to(device),
                                                              cluster_indices, cluster_offsets, weight_by_opacity=True)
    V_int = PCvox[:, :3].long()                                                    # :267
    attributes = torch.cat([mq, ms, mo.unsqueeze(1), mc], dim=1)                   # layout of data_util.py:366
    if output_ply is not None:
        save_ply(output_ply, PCvox[:, :3], mq, ms, mo, mc, voxel_size=info["voxel_size"], vmin=info["vmin"])
    info = dict(info, merged_means=mm, cluster_indices=cluster_indices, cluster_offsets=cluster_offsets)
    return V_int, attributes, info


def format_row(r):
    """One CSV line, same formatting as encode_3dgs.py:402-409."""
    return (f"{r['Frame']},{r['Quantization_Step']},{r['Rate_bpp']:.6f},"
            f"{r['RAHT_prelude_time']:.6f},{r['RAHT_transform_time']:.6f},{r['Quant_time']:.6f},"
            f"{r['Coeff_reorder_enc_time']:.6f},{r['Entropy_enc_time']:.6f},"
            f"{r['Entropy_dec_time']:.6f},{r['Dequant_time']:.6f},"
            f"{r['Coeff_reorder_dec_time']:.6f},{r['iRAHT_time']:.6f},"
            f"{r['Total_enc_time']:.6f},{r['Total_dec_time']:.6f},{r['Pipeline_time']:.6f},"
            f"{r['PSNR_all']:.6f},{r['PSNR_quats']:.6f},{r['PSNR_scales']:.6f},{r['PSNR_opacity']:.6f},{r['PSNR_colors']:.6f}")


def encode_3dgs(ply_list, J=10, colorStep=(1, 4, 8, 12, 16, 20, 24, 32, 64), csv_path="../results/runtime_3dgs.csv",
                device="cuda:0", dtype=torch.float32, fused=True, warmup=True):
    """Counterpart of the reference script: PLY list in, CSV out (defaults: encode_3dgs.py:29-33,60)."""
    from .ply_io import read_compressed_3dgs_ply
    d = os.path.dirname(csv_path)
    if d:
        os.makedirs(d, exist_ok=True)
    lines = [CSV_HEADER]
    for idx, path in enumerate(ply_list):
        res = read_compressed_3dgs_ply(path)
        if res is None:
            raise RuntimeError(f"Failed to load frame from {path}")
        V_int, attributes, _, _ = res
        if warmup and idx == 0:                                                   # :88-118
            encode_frame(V_int, attributes, J, colorStep[:1], frame=0, device=device, dtype=dtype, fused=fused)
        for r in encode_frame(V_int, attributes, J, colorStep, frame=idx + 1, device=device, dtype=dtype, fused=fused):
            lines.append(format_row(r))
    with open(csv_path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return lines


# ---- per-attribute quantization policy ------------------------------------------------
ATTRIBUTE_IMPORTANCE = {"quats": 1.0 / 21.93, "scales": 1.0 / 26.36, "opacity": 1.0 / 42.22, "colors": 1.0 / 38.67}


def per_attribute_steps(Coeff, total_levels_budget=1024, importance=None):
    """Per-channel quantization steps from the visual-importance policy of the reference's debug driver
    (reference python/encode_3dgs_debug.py:326-369): channels are grouped quats(4) / scales(3) /
    opacity(1) / colors(rest); each group gets `budget * importance / sum(importance)` quantization
    levels (at least 2, truncated to int) and the step range / (levels - 1), floored at 1e-6, where
    range = max - min of the group's RAHT coefficients. Returns a float64 tensor with one step per
    channel (the driver's Python floats; the float32 entry points round them once), usable with every quantize entry point (`forward_quant`, `quant_reorder`, ...), and the
    per-group table {name: dict(step, levels, range, channels)}.

    Coeff: (N, n_channels) RAHT coefficients (any device). The ranges need the coefficients, so the
    policy costs one un-fused forward transform per frame; the quantization itself is
    `plan.quant_reorder(Coeff, steps)` (or `plan.forward_quant(C, steps)` on the next, similar frame)."""
    imp = dict(ATTRIBUTE_IMPORTANCE if importance is None else importance)
    n_channels = int(Coeff.shape[1])
    ranges = {"quats": (0, 4), "scales": (4, 7), "opacity": (7, 8), "colors": (8, n_channels)}
    total = sum(imp.values())
    steps = torch.ones(n_channels, dtype=torch.float64)
    table = {}
    for name, (c0, c1) in ranges.items():
        if c0 >= n_channels:
            continue                                       # :353-354
        c1 = min(c1, n_channels)
        blk = Coeff[:, c0:c1]
        rng = blk.max() - blk.min()                        # :357-358
        levels = max(int(total_levels_budget * imp[name] / total), 2)       # :361-362
        step = rng / max(levels - 1, 1)                    # :365
        step = float(max(step.item(), 1e-6))               # :366 (a zero range must not divide by zero)
        steps[c0:c1] = step
        table[name] = dict(step=step, levels=levels, range=float(rng.item()), channels=(c0, c1))
    return steps, table
