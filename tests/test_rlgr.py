"""RLGR entropy stage. Host code: runs without a GPU.

Golden streams come from the REFERENCE's own coder (python/PyRLGR, built from its sources by
`make -C oracle ref`; tests/golden/gen_golden.py::rlgr_cases). Bar: byte-exact streams, exact
round trips (the reference asserts the same, encode_3dgs.py:242-245)."""
import os
import sys

import numpy as np
import pytest

from .conftest import ROOT, load_golden

G = load_golden("rlgr_streams")
CASES = sorted(k[:-3] for k in G if k.endswith("__x"))


@pytest.fixture(scope="module")
def rl():
    import raht_3dgs_codec_amd as R
    if not os.path.exists(R.SO_PATH):
        R.build()
    from raht_3dgs_codec_amd import rlgr
    return rlgr


@pytest.mark.parametrize("name", CASES)
def test_oracle_restatement_matches_reference_streams(oracle, name):
    x, flag, ref = G[name + "__x"], int(G[name + "__flag"]), G[name + "__bytes"]
    assert np.array_equal(oracle.rlgr_encode(x, flag), ref)
    assert np.array_equal(oracle.rlgr_decode(ref, len(x), flag), x)


@pytest.mark.parametrize("name", CASES)
def test_product_coder_is_byte_exact(rl, name):
    x, flag, ref = G[name + "__x"], int(G[name + "__flag"]), G[name + "__bytes"]
    m = rl.membuf()
    m.rlgrWrite(x, flag)                     # numpy fast path
    m.close()
    assert m.buffer_size() == len(ref) and np.array_equal(m.get_array(), ref)
    m2 = rl.membuf()
    m2.rlgrWrite([int(v) for v in x], flag)  # Python list, as the reference driver passes it
    assert m2.get_buffer() == ref.tolist()
    ns, back = rl.membuf(ref.tolist()).rlgrRead(len(x), flag)
    assert back == x.tolist() and ns >= 0


def test_channels_api_strided_and_threaded(rl, oracle):
    rng = np.random.default_rng(9)
    N, D = 20000, 7
    Q = np.empty((N, D), np.int32)
    for c in range(D):
        Q[:, c] = (rng.laplace(0, 0.3 * 4 ** c, N)).astype(np.int32)
    Q[:, 3] = 0
    streams, _ = rl.encode_channels(Q, 1, nthreads=3)
    for c in range(D):
        assert np.array_equal(streams[c], oracle.rlgr_encode(Q[:, c].astype(np.int64), 1)), c
    back, _ = rl.decode_channels(streams, N, 1, nthreads=2)
    assert np.array_equal(back, Q)
    # padded rows (row stride > D)
    big = np.zeros((N, 12), np.int32)
    big[:, :D] = Q
    s2, _ = rl.encode_channels(big[:, :D], 1, nthreads=1)
    assert all(np.array_equal(a, b) for a, b in zip(streams, s2))


@pytest.mark.parametrize("seed", range(6))
def test_random_round_trips_against_oracle(rl, oracle, seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(1, 5000))
    scale = float(10 ** rng.uniform(-1, 5))
    x = np.clip(rng.laplace(0, scale, n), -2 ** 31, 2 ** 31 - 1).astype(np.int64)
    x[rng.random(n) < rng.uniform(0, 0.9)] = 0
    m = rl.membuf()
    m.rlgrWrite(x.astype(np.int32), 1)
    assert np.array_equal(m.get_array(), oracle.rlgr_encode(x, 1))
    _, back = rl.membuf(m.get_array()).rlgrRead(n, 1)
    assert back == x.tolist()


def test_truncated_stream_does_not_crash(rl):
    x = np.arange(-500, 500, dtype=np.int32)
    m = rl.membuf()
    m.rlgrWrite(x, 1)
    buf = m.get_array()[: m.buffer_size() // 2]
    _, back = rl.membuf(buf).rlgrRead(len(x), 1)      # garbage tail, but bounded and no fault
    assert len(back) == len(x)


def test_reference_build_agrees_when_present(rl):
    """If oracle/_ref/rlgr*.so was built here (container with /root/reference), cross-check live."""
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    if not any(f.startswith("rlgr") for f in (os.listdir(ref_dir) if os.path.isdir(ref_dir) else [])):
        pytest.skip("reference rlgr build not present")
    sys.path.insert(0, ref_dir)
    import rlgr as ref
    rng = np.random.default_rng(77)
    x = (rng.laplace(0, 40, 30000)).astype(np.int64)
    x[rng.random(30000) < 0.5] = 0
    a = ref.membuf(); a.rlgrWrite(x.tolist(), 1); a.close()
    b = rl.membuf(); b.rlgrWrite(x, 1)
    assert a.get_buffer() == b.get_buffer()


def test_random_streams_round_trip_and_match_the_reference_build_when_present():
    """300 random sequences (all-zero, sparse, dense, full int32 range, one very long run): exact round
    trip; byte-identical to the reference's own coder when its build is available (oracle/_ref, built
    by `make -C oracle ref` in the build container)."""
    import importlib
    import os
    import sys
    from raht_3dgs_codec_amd import rlgr
    ref = None
    refdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
    if os.path.isdir(refdir):
        sys.path.insert(0, refdir)
        try:
            ref = importlib.import_module("rlgr")
            if not hasattr(ref, "membuf") or ref is rlgr:
                ref = None
        except Exception:
            ref = None
        finally:
            sys.path.remove(refdir)
    rng = np.random.default_rng(7)
    cases = []
    for t in range(300):
        n = int(rng.integers(0, 3000))
        kind = t % 6
        if kind == 0:
            x = np.zeros(n, dtype=np.int32)
        elif kind == 1:
            x = np.round(rng.laplace(0, 0.05, n)).astype(np.int32)
        elif kind == 2:
            x = np.round(rng.laplace(0, 30, n)).astype(np.int32)
        elif kind == 3:
            x = rng.integers(-2 ** 31 + 1, 2 ** 31 - 1, n).astype(np.int32)
        elif kind == 4:
            x = np.zeros(n, dtype=np.int32)
            if n:
                x[rng.integers(0, n, max(1, n // 500))] = rng.integers(-2 ** 30, 2 ** 30, max(1, n // 500))
        else:
            x = np.round(rng.normal(0, 10 ** rng.uniform(-1, 6), n)).astype(np.int32)
        cases.append(x)
    cases.append(np.zeros(300_000, dtype=np.int32))                 # one very long run: the run parameter keeps growing
    checked = 0
    for x in cases:
        streams, _ = rlgr.encode_channels(x.reshape(1, -1), 1, nthreads=1, channel_major=True)
        back, _ = rlgr.decode_channels(streams, x.shape[0], 1, nthreads=1, channel_major=True)
        assert np.array_equal(back.reshape(-1), x)
        if ref is not None and x.shape[0] <= 3000:
            m = ref.membuf()
            m.rlgrWrite(x.tolist(), 1)
            m.close()
            assert bytes(bytearray(m.get_buffer())) == streams[0].tobytes()
            checked += 1
    assert ref is None or checked == 300


def test_channel_coder_keeps_one_arena_and_checks_round_trips():
    """ChannelCoder (the pipeline's glue-free path): same bytes as encode_channels, decode into a caller's array, threaded
    equality with the position of the first difference."""
    from raht_3dgs_codec_amd import rlgr
    rng = np.random.default_rng(5)
    N, D = 50000, 7
    Q = np.rint(rng.laplace(0, 3.0, size=(D, N))).astype(np.int32)
    Q[3] = 0
    cc = rlgr.ChannelCoder(N, D, 1, nthreads=3)
    for rep in range(2):                                      # the arena is reused
        cc.encode(Q)
        ref, _ = rlgr.encode_channels(Q, 1, nthreads=2, channel_major=True)
        assert [bytes(s) for s in cc.streams()] == [bytes(s) for s in ref]
        assert cc.size_bytes == sum(len(s) for s in ref)
        out = np.empty_like(Q)
        cc.decode(out)
        assert rlgr.arrays_equal(out, Q, 4)
        if rep == 0:
            Q = Q[:, ::-1].copy()
    out[5, 1234] += 1
    assert not rlgr.arrays_equal(out, Q, 4)
    import ctypes as C
    from raht_3dgs_codec_amd import _lib
    first = C.c_int64()
    _lib.check(_lib.lib().raht_i32_equal(out.ctypes.data_as(C.c_void_p), Q.ctypes.data_as(C.c_void_p), out.size, 3, C.byref(first)))
    assert first.value == 5 * N + 1234
