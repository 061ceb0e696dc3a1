"""PLY I/O parity with the reference writer / reader: the fixture holds the exact
bytes the reference's save_ply produced and what its reader returned for them."""
import os

import numpy as np
import torch

from .conftest import load_golden


def test_reader_and_writer_match_reference_bytes(tmp_path):
    from raht_3dgs_codec_amd import ply_io
    g = load_golden("pipeline_small")
    path = os.path.join(tmp_path, "ref.ply")
    open(path, "wb").write(g["ply_bytes"].tobytes())
    V, A, vs, vmin = ply_io.read_compressed_3dgs_ply(path)
    assert V.dtype == torch.int64 and np.array_equal(V.numpy(), g["V"])
    assert A.dtype == torch.float32 and np.array_equal(A.numpy(), g["A"])           # quats, scales, opacity, colours
    assert vs == float(g["voxel_size"]) and np.array_equal(vmin.numpy(), g["vmin"])
    out = os.path.join(tmp_path, "sub", "mine.ply")
    ply_io.save_ply(out, V.float(), A[:, 0:4], A[:, 4:7], A[:, 7], A[:, 8:], voxel_size=vs, vmin=vmin)
    assert open(out, "rb").read() == g["ply_bytes"].tobytes()                       # byte-identical file


def test_reader_error_contract(tmp_path):
    import warnings
    from raht_3dgs_codec_amd import ply_io
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert ply_io.read_compressed_3dgs_ply(os.path.join(tmp_path, "missing.ply")) is None   # data_util.py:375-377
        assert w
