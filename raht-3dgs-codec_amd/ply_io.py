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


def read_compressed_3dgs_ply(filename):
    try:
        with open(filename, "rb") as f:
            lines = []
            while True:
                line = f.readline().decode("ascii").strip()
                lines.append(line)
                if line == "end_header":
                    break
            n, binary, voxel_size, vmin, nprops = 0, False, None, None, 0
            for line in lines:
                if line.startswith("format"):
                    binary = "binary" in line
                elif line.startswith("element vertex"):
                    n = int(line.split()[-1])
                elif line.startswith("comment voxel_size"):
                    voxel_size = float(line.split()[-1])
                elif line.startswith("comment vmin"):
                    p = line.split()
                    vmin = torch.tensor([float(p[2]), float(p[3]), float(p[4])], dtype=torch.float32)
                elif line.startswith("property float"):
                    nprops += 1
            if n == 0:
                raise ValueError("Could not find vertex count in PLY header")
            if not binary:
                raise ValueError("ASCII format not supported for compressed 3DGS PLY. Use binary format.")
            if voxel_size is None:
                warnings.warn("Could not find voxel_size in PLY header comments")
                voxel_size = 1.0
            if vmin is None:
                warnings.warn("Could not find vmin in PLY header comments")
                vmin = torch.zeros(3, dtype=torch.float32)
            if nprops < 15:
                raise ValueError(f"unexpected vertex layout ({nprops} float properties)")
            data = np.fromfile(f, dtype="<f4", count=n * nprops).reshape(n, nprops)
        ncol = nprops - 14                                   # 62 -> 48 SH coefficients
        V_int = torch.from_numpy(data[:, 0:3].copy()).long()
        colors = data[:, 6:6 + ncol]
        opacity = data[:, 6 + ncol:7 + ncol]
        scales = data[:, 7 + ncol:10 + ncol]
        quats = data[:, 10 + ncol:14 + ncol]
        attributes = torch.from_numpy(np.concatenate([quats, scales, opacity, colors], axis=1).astype(np.float32))
        return V_int, attributes, voxel_size, vmin
    except FileNotFoundError:
        warnings.warn(f"File not found: {filename}")
        return None
    except Exception as e:                                    # same contract as the reference: warn, return None
        warnings.warn(f"Error reading compressed 3DGS PLY {filename}: {e}")
        return None


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
