"""The RLGR stage on the GPU, segmented (csrc/rlgr_seg.hip; SURVEY.md 8f-1 "GPU-segmented"): every segment must be
byte-identical to the host coder's stream for the same slice -- and the host coder is pinned byte for byte by streams built with
the reference's own PyRLGR (tests/test_rlgr.py) -- so that any RLGR decoder reads a segment."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[-1, 0, 1, 2, 4], ids=["out_auto", "out_word", "out_vec", "out_lds", "rows_per_lane"], autouse=True)
def _decoder_output_mode(request):
    """every test with each of the ways the decoders' symbols leave the lanes (raht_debug_rlgr_decode_out): these frames are far
    too small for the default to pick the LDS columns by itself. rows_per_lane: row-major output from the per-lane decoder with
    strided stores instead of the symbol-synchronous one (the default for that layout)"""
    from raht_3dgs_codec_amd import _lib
    prev = _lib.lib().raht_debug_rlgr_decode_out(request.param)
    prev_e = _lib.lib().raht_debug_rlgr_encode_out(request.param)          # (the batched encoder: words / LDS columns)
    yield
    _lib.lib().raht_debug_rlgr_decode_out(prev)
    _lib.lib().raht_debug_rlgr_encode_out(prev_e)


def _cases():
    rng = np.random.default_rng(11)
    lap = lambda n, b: np.rint(rng.laplace(0, b, size=n)).astype(np.int64)          # noqa: E731
    N = 70000
    chans = [lap(N, 0.3), lap(N, 3.0), lap(N, 200.0), np.zeros(N, np.int64),                          # sparse / dense / wide / all zero
             rng.integers(-2 ** 31, 2 ** 31 - 1, size=N),                                             # escapes: |u| >= 2^32 >> k
             np.where(rng.random(N) < 0.001, lap(N, 50.0), 0),                                        # long zero runs
             np.concatenate([np.zeros(N // 2, np.int64), lap(N - N // 2, 20.0)]),                     # a run across segment borders
             np.arange(N) % 7 - 3]
    return np.stack(chans).astype(np.int32)


@pytest.mark.parametrize("seg_len", [64, 1000, 4096, 100000])
def test_every_segment_is_the_host_coders_stream(seg_len):
    import torch
    from raht_3dgs_codec_amd import rlgr
    Q = _cases()
    D, N = Q.shape
    sc = rlgr.SegmentedCoder(N, D, seg_len)
    Qd = torch.from_numpy(Q).cuda()
    total = sc.encode(Qd)
    lens = sc.seg_bytes.cpu().numpy()
    offs = sc.seg_off.cpu().numpy()
    assert offs[0] == 0 and offs[-1] == total and np.all(np.diff(offs) == (lens + 3) // 4 * 4)
    blob = sc.out[:total].cpu().numpy()
    for c in range(D):
        for s in range(sc.nseg):
            sl = Q[c, s * seg_len: (s + 1) * seg_len]
            m = rlgr.membuf()
            m.rlgrWrite(sl, 1)                                                      # host coder == reference coder (tests/test_rlgr.py)
            ref = m.get_array()
            g = c * sc.nseg + s
            assert lens[g] == ref.shape[0], (c, s, lens[g], ref.shape[0])
            assert np.array_equal(blob[offs[g]: offs[g] + lens[g]], ref), (c, s)
            pad = blob[offs[g] + lens[g]: offs[g + 1]]
            assert not pad.any()
    # GPU decode of the GPU streams
    back = sc.decode()
    assert torch.equal(back, Qd) and int(sc.bad.item()) == 0
    # ... and of a container that went over the wire
    blob = sc.container()
    sc2 = rlgr.SegmentedCoder.from_container(blob)
    assert torch.equal(sc2.decode(), Qd)
    assert sc.size_bytes == len(blob)
    for bad in (blob[:30], blob[:-5], b"XXXXXXXX" + blob[8:], blob[:8] + np.array([-3, D, seg_len, 1, 0], np.int64).tobytes() + blob[48:]):
        with pytest.raises(ValueError):
            rlgr.SegmentedCoder.from_container(bad)
    # host decoder on GPU-made segments (any RLGR decoder reads a segment)
    for c, s in ((0, 0), (4, sc.nseg - 1), (6, sc.nseg // 2)):
        n = min(seg_len, N - s * seg_len)
        _, vals = rlgr.membuf(sc.segment(c, s)).rlgrRead(n, 1)
        assert np.array_equal(np.asarray(vals, dtype=np.int64), Q[c, s * seg_len: s * seg_len + n].astype(np.int64))


def test_unsigned_flag_and_strided_channels():
    import torch
    from raht_3dgs_codec_amd import rlgr
    rng = np.random.default_rng(3)
    N, D = 9000, 5
    big = torch.from_numpy(rng.integers(0, 40, size=(D, N + 37)).astype(np.int32)).cuda()
    Qv = big[:, :N]                                             # channel stride N + 37
    sc = rlgr.SegmentedCoder(N, D, 512, flag_signed=0)
    sc.encode(Qv)
    for c in (0, 3):
        m = rlgr.membuf(); m.rlgrWrite(Qv[c, :512].cpu().numpy(), 0)
        assert np.array_equal(sc.segment(c, 0), m.get_array())
    out = torch.full((D, N + 5), -7, dtype=torch.int32, device="cuda")
    sc.decode(out=out[:, :N])
    assert torch.equal(out[:, :N], Qv) and bool((out[:, N:] == -7).all())


def test_incompressible_data_grows_the_buffer_and_corrupt_tables_do_not_fault():
    import torch
    from raht_3dgs_codec_amd import rlgr
    rng = np.random.default_rng(9)
    N, D = 20000, 3
    Q = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, size=(D, N)).astype(np.int32)).cuda()
    sc = rlgr.SegmentedCoder(N, D, 1024)
    total = sc.encode(Q)                                         # ~8 bytes per symbol: more than the raw-dump estimate
    assert total > 4 * N * D and sc.cap >= total
    assert torch.equal(sc.decode(), Q)
    # offsets / lengths pointing outside the buffer: those segments decode as empty streams, the flag is raised, nothing faults
    sc.seg_off[5] = 2 ** 31 - 4
    sc.seg_bytes[7] = 2 ** 31 - 1
    out = sc.decode()
    torch.cuda.synchronize()
    assert int(sc.bad.item()) == 1
    nseg = sc.nseg
    ok = torch.ones(D * nseg, dtype=torch.bool)
    ok[5] = ok[7] = False
    got = out.view(D, -1)
    for g in np.nonzero(ok.numpy())[0][:40]:
        c, s = divmod(int(g), nseg)
        assert torch.equal(got[c, s * 1024: (s + 1) * 1024], Q[c, s * 1024: (s + 1) * 1024])


def test_cost_of_restarting_the_coder_per_segment():
    """The state restarts in every segment: what that costs in bytes against ONE stream per channel (the reference's format), on
    quantized RAHT coefficients of a 3DGS-like frame."""
    import torch
    import raht_3dgs_codec_amd as R
    from raht_3dgs_codec_amd import rlgr, synth
    V, keys, C = synth.scene(300000, 10, 56, seed=4)
    plan = R.RahtPlan.from_keys(torch.from_numpy(keys.view(np.int64)).cuda(), 30)
    for step in (0.02, 0.2):
        Qcm = rlgr.transpose_on_device(plan.forward_quant(torch.from_numpy(C).cuda(), step))
        streams, _ = rlgr.encode_channels(rlgr.to_host(Qcm), 1, channel_major=True)
        one = sum(int(s.shape[0]) for s in streams)
        for S, bar in ((4096, 1.02), (1024, 1.06)):
            sc = rlgr.SegmentedCoder(Qcm.shape[1], Qcm.shape[0], S)
            sc.encode(Qcm)
            ratio = sc.size_bytes / one
            print(f"[rlgr_seg] step {step} seg_len {S}: {sc.size_bytes} bytes against {one} as one stream per channel: x{ratio:.4f}")
            assert ratio <= bar, (step, S, ratio)
            assert torch.equal(sc.decode(), Qcm)


@pytest.mark.parametrize("seg_len,ld_extra", [(64, 0), (1000, 5), (4096, 0)])
def test_row_major_input_gives_the_same_container(seg_len, ld_extra):
    """raht_rlgr_seg_encode_strided / _decode_strided on ROW-MAJOR coefficients (symbol n of channel c at Q[n * ld + c], what the
    transform kernels write and read): the same segments, offsets and bytes as the channel-major call -- no transpose on either
    side of the coder -- and the decoder writes rows straight back."""
    import torch
    from raht_3dgs_codec_amd import rlgr
    Qcm = torch.from_numpy(_cases()).cuda()                     # (D, N)
    D, N = Qcm.shape
    big = torch.full((N, D + ld_extra), 12345, dtype=torch.int32, device="cuda")
    big[:, :D] = Qcm.t()
    Qrm = big[:, :D]                                            # (N, D), row stride D + ld_extra
    a, b = rlgr.SegmentedCoder(N, D, seg_len), rlgr.SegmentedCoder(N, D, seg_len)
    ta, tb = a.encode(Qcm), b.encode(Qrm)
    assert ta == tb and torch.equal(a.seg_bytes, b.seg_bytes) and torch.equal(a.seg_off, b.seg_off)
    assert torch.equal(a.out[:ta], b.out[:tb])
    assert a.container() == b.container()
    out = torch.full((N, D + ld_extra), -9, dtype=torch.int32, device="cuda")
    b.decode(out=out[:, :D])
    assert torch.equal(out[:, :D], Qrm) and bool((out[:, D:] == -9).all()) and int(b.bad.item()) == 0
    assert torch.equal(b.decode(row_major=True), Qrm.contiguous())
    assert torch.equal(a.decode(), Qcm)                         # either decoder layout from either encoder's container


def test_container_header_limits():
    """a decoder allocates from the payload it was handed, not from the header's raw size, and may cap the symbol count"""
    import torch
    from raht_3dgs_codec_amd import rlgr
    Q = torch.zeros((4, 5000), dtype=torch.int32, device="cuda")
    sc = rlgr.SegmentedCoder(5000, 4, 512)
    sc.encode(Q)
    blob = sc.container()
    d = rlgr.SegmentedCoder.from_container(blob)
    assert d.cap <= max(16, sc.total) + 16 and torch.equal(d.decode(), Q)
    with pytest.raises(ValueError):
        rlgr.SegmentedCoder.from_container(blob, max_symbols=1000)
    # the 32-bit container: inputs whose worst case could reach 4 GiB are refused before anything runs
    from raht_3dgs_codec_amd import _lib
    import ctypes as C
    tot = C.c_int64()
    rc = _lib.lib().raht_rlgr_seg_encode_strided(C.c_void_p(Q.data_ptr()), 600_000_000, 1, 1, 600_000_000, 2048, 1, C.c_void_p(sc.seg_bytes.data_ptr()),
                                                 C.c_void_p(sc.seg_off.data_ptr()), C.c_void_p(sc.out.data_ptr()), sc.cap, C.byref(tot), None)
    assert rc == -1 and b"4 GiB" in _lib.lib().raht_last_error()


def _step_frames(k, N=30000, D=8, seed=5):
    """k 'quantization steps' of one frame: the same Laplacian coefficients divided by growing steps (sparser and sparser)"""
    rng = np.random.default_rng(seed)
    base = rng.laplace(0, 40.0, size=(N, D)) * np.linspace(0.2, 3.0, D)[None, :]
    return [np.floor(base / (1.0 + 1.7 * j) + 0.5).astype(np.int32) for j in range(k)]


@pytest.mark.parametrize("k,seg_len,row_major", [(1, 1000, True), (3, 64, True), (9, 2048, True), (4, 1000, False), (13, 4096, True), (12, 100000, False),
                                                   (3, 1001, False), (2, 333, True)])       # (segment starts off the 16-byte grid: the word paths)
def test_batch_of_frames_is_every_frame_alone(k, seg_len, row_major):
    """raht_rlgr_seg_encode_batch / _decode_batch: the steps of a frame coded by one set of launches -- same tables and containers
    as one call per frame, byte for byte (k = 13: two chunks of the 12 a call takes)"""
    import torch
    from raht_3dgs_codec_amd import rlgr
    frames = _step_frames(k)
    N, D = frames[0].shape
    Qs = [torch.from_numpy(f).cuda() if row_major else torch.from_numpy(np.ascontiguousarray(f.T)).cuda() for f in frames]
    alone = [rlgr.SegmentedCoder(N, D, seg_len) for _ in range(k)]
    for c, Q in zip(alone, Qs):
        c.encode(Q)
    batch = [rlgr.SegmentedCoder(N, D, seg_len) for _ in range(k)]
    totals = rlgr.SegmentedCoder.encode_batch(batch, Qs)
    assert totals == [c.total for c in alone]
    for a, b in zip(alone, batch):
        assert torch.equal(a.seg_bytes, b.seg_bytes) and torch.equal(a.seg_off, b.seg_off)
        assert torch.equal(a.out[: a.total], b.out[: b.total])
        assert a.container() == b.container()
    for rm in (False, True):
        outs = rlgr.SegmentedCoder.decode_batch(batch, row_major=rm)
        for o, f in zip(outs, frames):
            assert torch.equal(o, torch.from_numpy(f if rm else np.ascontiguousarray(f.T)).cuda())
    assert int(batch[0].bad.item()) == 0
    # decoders built from the wire format, decoded together
    wire = [rlgr.SegmentedCoder.from_container(b.container()) for b in batch]
    for o, f in zip(rlgr.SegmentedCoder.decode_batch(wire, row_major=True), frames):
        assert torch.equal(o, torch.from_numpy(f).cuda())


def test_batch_with_an_incompressible_frame_and_bad_arguments():
    """one frame of a batch outgrows its slots (raw 32-bit noise): the call falls back to the exact passes frame by frame and the
    buffers grow; corrupt tables of ONE frame set ITS bit; argument checks"""
    import ctypes as C
    import torch
    from raht_3dgs_codec_amd import _lib, rlgr
    frames = _step_frames(3, N=20000, D=6)
    rng = np.random.default_rng(2)
    frames[1] = rng.integers(-2 ** 31, 2 ** 31 - 1, size=frames[1].shape).astype(np.int32)
    N, D = frames[0].shape
    Qs = [torch.from_numpy(f).cuda() for f in frames]
    batch = [rlgr.SegmentedCoder(N, D, 512) for _ in frames]
    rlgr.SegmentedCoder.encode_batch(batch, Qs)
    assert batch[1].total > 4 * N * D                                   # the escapes cost more than a raw dump
    for c, Q in zip(batch, Qs):
        alone = rlgr.SegmentedCoder(N, D, 512)
        alone.encode(Q)
        assert alone.container() == c.container()
    outs = rlgr.SegmentedCoder.decode_batch(batch, row_major=True)
    assert all(torch.equal(o, Q) for o, Q in zip(outs, Qs)) and int(batch[0].bad.item()) == 0
    batch[2].seg_off[5] = 2 ** 31 - 4                                   # a table entry of frame 2 points far outside its payload
    rlgr.SegmentedCoder.decode_batch(batch, row_major=True)
    torch.cuda.synchronize()
    assert int(batch[0].bad.item()) == 1 << 2
    with pytest.raises(ValueError):
        rlgr.SegmentedCoder.encode_batch(batch, Qs[:2])
    with pytest.raises(ValueError):
        rlgr.SegmentedCoder.encode_batch(batch[:2] + [rlgr.SegmentedCoder(N, D, 1024)], Qs)
    with pytest.raises(ValueError):
        rlgr.SegmentedCoder.encode_batch(batch, [Qs[0], Qs[1], Qs[2].t().contiguous()])
    L = _lib.lib()
    VP, I64 = C.c_void_p * 1, C.c_int64 * 1
    tot = I64()
    c0 = batch[0]
    args = lambda k, seg: (k, VP(Qs[0].data_ptr()), N, D, D, 1, seg, 1, VP(c0.seg_bytes.data_ptr()), VP(c0.seg_off.data_ptr()), VP(c0.out.data_ptr()),   # noqa: E731
                           I64(c0.cap), tot, None)
    assert L.raht_rlgr_seg_encode_batch(*args(0, 512)) == -1                # RAHT_ERR_INVALID
    assert L.raht_rlgr_seg_encode_batch(*args(13, 512)) == -1                # RAHT_ERR_INVALID
    assert L.raht_rlgr_seg_encode_batch(*args(1, 8)) == -1                # RAHT_ERR_INVALID


def test_round_trip_assertion_inside_the_batch_decoder():
    """decode_batch(expect=...): raht_rlgr_seg_decode_batch_check compares every decoded symbol with what it should be on its way out
    (python/encode_3dgs.py:242-245 without a pass of its own); one flipped symbol in one frame is found, in that frame only"""
    import torch
    from raht_3dgs_codec_amd import rlgr
    frames = _step_frames(13, N=9000, D=7)
    N, D = frames[0].shape
    Qs = [torch.from_numpy(f).cuda() for f in frames]
    coders = [rlgr.SegmentedCoder(N, D, 512) for _ in frames]
    rlgr.SegmentedCoder.encode_batch(coders, Qs)
    outs = rlgr.SegmentedCoder.decode_batch(coders, row_major=True, expect=Qs)
    assert rlgr.SegmentedCoder.roundtrip_failed(coders) == []
    assert all(torch.equal(o, q) for o, q in zip(outs, Qs))
    wrong = [q.clone() for q in Qs]
    wrong[4][8191, 3] += 1                                              # (the last symbol of a segment)
    wrong[12][0, 0] -= 2                                                # (second chunk of the 12 a call takes)
    fresh = [rlgr.SegmentedCoder.from_container(c.container()) for c in coders]
    outs = rlgr.SegmentedCoder.decode_batch(fresh, row_major=True, expect=wrong)
    assert rlgr.SegmentedCoder.roundtrip_failed(fresh) == [4, 12]
    assert all(torch.equal(o, q) for o, q in zip(outs, Qs))            # (what was decoded is what was encoded, whatever was expected)
    with pytest.raises(ValueError):
        rlgr.SegmentedCoder.decode_batch(coders, row_major=False, expect=Qs)
    with pytest.raises(ValueError):
        rlgr.SegmentedCoder.decode_batch(coders, row_major=True, expect=Qs[:3])


@pytest.mark.parametrize("N,D,seg_len", [(5000, 1, 64), (777, 130, 128), (64, 3, 64), (4097, 59, 4096)])
def test_odd_shapes_in_both_layouts(N, D, seg_len):
    """one channel, more channels than a wave has lanes, one segment, a last segment of one symbol: single-frame and batched calls,
    channel-major and row-major, against each other and against the input"""
    import torch
    from raht_3dgs_codec_amd import rlgr
    rng = np.random.default_rng(N + D)
    frames = [np.rint(rng.laplace(0, 3.0 + 20.0 * j, size=(N, D))).astype(np.int32) * (rng.random((N, D)) < 0.6) for j in range(3)]
    Qs = [torch.from_numpy(f).cuda() for f in frames]
    coders = [rlgr.SegmentedCoder(N, D, seg_len) for _ in frames]
    rlgr.SegmentedCoder.encode_batch(coders, Qs)
    for c, Q, f in zip(coders, Qs, frames):
        alone = rlgr.SegmentedCoder(N, D, seg_len)
        alone.encode(torch.from_numpy(np.ascontiguousarray(f.T)).cuda())          # channel-major input, one frame
        assert alone.container() == c.container()
        assert torch.equal(c.decode(row_major=True), Q) and torch.equal(c.decode(), Q.t())
    for rm in (True, False):
        outs = rlgr.SegmentedCoder.decode_batch(coders, row_major=rm)
        assert all(torch.equal(o, Q if rm else Q.t()) for o, Q in zip(outs, Qs))
    rlgr.SegmentedCoder.decode_batch(coders, row_major=True, expect=Qs)
    assert rlgr.SegmentedCoder.roundtrip_failed(coders) == []
