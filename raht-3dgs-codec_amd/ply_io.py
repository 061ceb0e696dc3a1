"""3DGS PLY I/O for the codec drivers.

``read_compressed_3dgs_ply`` mirrors reference python/data_util.py:272-382 and ``save_ply`` mirrors
reference python/quality_eval.py:18-117 (same header text, same vertex layout, so files are
byte-identical and interchangeable):

    62 float32 per vertex: x y z | nx ny nz | 48 SH colours | opacity | scale_0..2 | rot_0..3
    header comments:       "comment voxel_size <f>" and "comment vmin <x> <y> <z>"

The reader returns ``(V_int int64 (N,3), attributes float32 (N,56) = [quats4, scales3, opacity1,
colours48], voxel_size, vmin)`` exactly like the reference (`:336-368`). Host-side numpy; vectorised
(the reference writes vertex by vertex in a Python loop, `quality_eval.py:103-115`).
"""
import os
import warnings

import numpy as np
import torch


_HEADER_LIMIT = 1 << 20          # bytes: a header that has not ended by then is not a PLY header


def _parse_header(f):
    """Reads up to and including `end_header`; returns the fields the codec needs as a dict:
    format (str), vertices (int), float_props (int), voxel_size (float | None), vmin (3 floats | None)."""
    blob = bytearray()
    while not blob.endswith(b"end_header\n") and not blob.endswith(b"end_header\r\n"):
        chunk = f.readline()
        if not chunk or len(blob) > _HEADER_LIMIT:
            raise ValueError("PLY header does not end")
        blob += chunk
    words = [ln.split() for ln in blob.decode("ascii").splitlines()]
    hdr = {"format": "", "vertices": 0, "float_props": 0, "voxel_size": None, "vmin": None}
    for w in words:
        if len(w) >= 2 and w[0] == "format":
            hdr["format"] = w[1]
        elif w[:2] == ["element", "vertex"]:
            hdr["vertices"] = int(w[2])
        elif w[:2] == ["property", "float"]:
            hdr["float_props"] += 1
        elif w[:2] == ["comment", "voxel_size"]:
            hdr["voxel_size"] = float(w[2])
        elif w[:2] == ["comment", "vmin"]:
            hdr["vmin"] = [float(x) for x in w[2:5]]
    return hdr


def read_compressed_3dgs_ply(filename):
    """-> (V_int, attributes, voxel_size, vmin), or None with a warning when the file cannot be used (the reference's
    contract, data_util.py:272-382)."""
    try:
        with open(filename, "rb") as f:
            hdr = _parse_header(f)
            n, nprops = hdr["vertices"], hdr["float_props"]
            problems = {
                "Could not find vertex count in PLY header": n == 0,
                "ASCII format not supported for compressed 3DGS PLY. Use binary format.": not hdr["format"].startswith("binary"),
                f"unexpected vertex layout ({nprops} float properties)": nprops < 15,
            }
            for msg, bad in problems.items():
                if bad:
                    raise ValueError(msg)
            data = np.fromfile(f, dtype="<f4", count=n * nprops).reshape(n, nprops)
    except FileNotFoundError:
        warnings.warn(f"File not found: {filename}")
        return None
    except Exception as e:                                    # same contract as the reference: warn, return None
        warnings.warn(f"Error reading compressed 3DGS PLY {filename}: {e}")
        return None
    voxel_size, vmin = hdr["voxel_size"], hdr["vmin"]
    if voxel_size is None:
        warnings.warn("Could not find voxel_size in PLY header comments")
        voxel_size = 1.0
    if vmin is None:
        warnings.warn("Could not find vmin in PLY header comments")
        vmin = [0.0, 0.0, 0.0]
    # vertex layout: x y z | nx ny nz | ncol colours | opacity | scale_0..2 | rot_0..3   (62 floats -> 48 colours)
    ncol = nprops - 14
    cols = {"xyz": slice(0, 3), "colors": slice(6, 6 + ncol), "opacity": slice(6 + ncol, 7 + ncol),
            "scales": slice(7 + ncol, 10 + ncol), "quats": slice(10 + ncol, 14 + ncol)}
    V_int = torch.from_numpy(data[:, cols["xyz"]].copy()).long()
    attributes = torch.from_numpy(np.concatenate([data[:, cols[k]] for k in ("quats", "scales", "opacity", "colors")],
                                                 axis=1).astype(np.float32))
    return V_int, attributes, voxel_size, torch.tensor(vmin, dtype=torch.float32)


def save_ply(filepath, means, quats, scales, opacities, colors, voxel_size=None, vmin=None):
    def npf(t):
        return (t.detach().cpu().float().numpy() if isinstance(t, torch.Tensor) else np.asarray(t, dtype=np.float32))
    means_np, quats_np, scales_np = npf(means), npf(quats), npf(scales)
    op_np, col_np = npf(opacities).reshape(-1, 1), npf(colors)
    N, color_dim = means_np.shape[0], col_np.shape[1]
    d = os.path.dirname(filepath)
    if d:
        os.makedirs(d, exist_ok=True)
    hdr = ["ply", "format binary_little_endian 1.0"]
    if voxel_size is not None:
        hdr.append(f"comment voxel_size {voxel_size}")
    if vmin is not None:
        v = npf(vmin)
        hdr.append(f"comment vmin {v[0]} {v[1]} {v[2]}")
    hdr.append(f"element vertex {N}")
    hdr += ["property float x", "property float y", "property float z",
            "property float nx", "property float ny", "property float nz"]
    hdr += [f"property float f_dc_{i}" for i in range(color_dim)]          # the reference names all of them f_dc_i
    hdr += ["property float opacity", "property float scale_0", "property float scale_1", "property float scale_2",
            "property float rot_0", "property float rot_1", "property float rot_2", "property float rot_3", "end_header"]
    body = np.concatenate([means_np, np.zeros((N, 3), np.float32), col_np, op_np, scales_np, quats_np], axis=1)
    with open(filepath, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode())
        f.write(np.ascontiguousarray(body, dtype="<f4").tobytes())
