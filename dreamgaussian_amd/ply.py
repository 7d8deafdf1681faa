"""The on-disk format either side of the path: DreamGaussian's Gaussian PLY
(`GaussianModel.save_ply` / `load_ply`, gs_renderer.py:376-462; written by main.py:461-463 via
`save_model`, read by `Renderer.initialize`, gs_renderer.py:680-683).

One `vertex` element, `binary_little_endian 1.0`, every property `float` (f4), in the order of
`construct_list_of_attributes` (gs_renderer.py:376-389):
    x y z  nx ny nz  f_dc_0..2  f_rest_0..(3(K-1)-1)  opacity  scale_0..2  rot_0..3
with the SH blocks stored CHANNEL-major (`transpose(1, 2).flatten(start_dim=1)`, :397-398), normals
all zero, and the RAW (pre-activation) opacity / scaling / rotation. The reference goes through the
`plyfile` package; this is a dependency-free numpy reader/writer of the same bytes (plyfile writes
exactly this header for an all-f4 structured array), and the reader accepts any property order,
comments and the `float32` spelling so that files written by other tools load too."""
from __future__ import annotations

import os
from typing import Dict

import numpy as np
import torch


def attribute_names(n_dc: int, n_rest: int, n_scale: int = 3, n_rot: int = 4):
    """gs_renderer.py:376-389."""
    l = ["x", "y", "z", "nx", "ny", "nz"]
    l += [f"f_dc_{i}" for i in range(n_dc)]
    l += [f"f_rest_{i}" for i in range(n_rest)]
    l.append("opacity")
    l += [f"scale_{i}" for i in range(n_scale)]
    l += [f"rot_{i}" for i in range(n_rot)]
    return l


def _np(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def save_ply(path: str, xyz, features_dc, features_rest, opacity, scaling, rotation) -> None:
    """Arguments are the model's raw parameters: `_xyz [N,3]`, `_features_dc [N,1,3]`,
    `_features_rest [N,K-1,3]`, `_opacity [N,1]`, `_scaling [N,3]`, `_rotation [N,4]`."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    xyz = _np(xyz).astype("<f4")
    N = xyz.shape[0]
    f_dc = np.ascontiguousarray(np.transpose(_np(features_dc), (0, 2, 1))).reshape(N, -1)       # :397
    f_rest = np.ascontiguousarray(np.transpose(_np(features_rest), (0, 2, 1))).reshape(N, -1)   # :398
    cols = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, _np(opacity).reshape(N, -1),
                           _np(scaling).reshape(N, -1), _np(rotation).reshape(N, -1)), axis=1).astype("<f4")
    names = attribute_names(f_dc.shape[1], f_rest.shape[1], _np(scaling).reshape(N, -1).shape[1], _np(rotation).reshape(N, -1).shape[1])
    assert cols.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % N
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as fh:
        fh.write(header.encode("ascii"))
        fh.write(np.ascontiguousarray(cols).tobytes())


_TYPES = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1", "char": "i1",
          "int8": "i1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4",
          "uint": "u4", "uint32": "u4"}


def read_vertex_table(path: str) -> np.ndarray:
    """The first `vertex` element as a structured array (binary little/big endian or ascii; no list properties)."""
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements, cur = None, [], None
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f"{path}: header has no end_header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                cur = {"name": tok[1], "count": int(tok[2]), "props": []}
                elements.append(cur)
            elif tok[0] == "property":
                if tok[1] == "list":
                    if cur is elements[0]:
                        raise ValueError(f"{path}: list properties are not part of the Gaussian PLY")
                    continue                                   # later elements (faces) are not read
                cur["props"].append((tok[2], _TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if not elements or elements[0]["name"] != "vertex":
            raise ValueError(f"{path}: the first element must be 'vertex'")
        el = elements[0]
        if fmt == "ascii":
            rows = np.loadtxt(fh, max_rows=el["count"], ndmin=2)
            out = np.empty(el["count"], dtype=[(n, "<" + t) for n, t in el["props"]])
            for k, (n, _) in enumerate(el["props"]):
                out[n] = rows[:, k]
            return out
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, order + t) for n, t in el["props"]])
        buf = fh.read(dt.itemsize * el["count"])
        if len(buf) != dt.itemsize * el["count"]:
            raise ValueError(f"{path}: truncated vertex data")
        return np.frombuffer(buf, dtype=dt, count=el["count"])


def load_ply(path: str, max_sh_degree: int, device="cpu") -> Dict[str, torch.Tensor]:
    """-> the raw parameters as float32 tensors with the shapes `load_ply` gives them
    (gs_renderer.py:455-460): xyz [N,3], features_dc [N,1,3], features_rest [N,K-1,3], opacity [N,1],
    scaling [N,3], rotation [N,4]."""
    v = read_vertex_table(path)
    names = v.dtype.names
    col = lambda n: np.asarray(v[n], dtype=np.float64)
    xyz = np.stack((col("x"), col("y"), col("z")), axis=1)
    opac = col("opacity")[..., None]
    f_dc = np.stack((col("f_dc_0"), col("f_dc_1"), col("f_dc_2")), axis=1)[:, :, None]          # [N,3,1]  :420-423
    # the reference takes these groups in FILE order (:425, :434, :439); sorting by the numeric suffix is the
    # identity on every file it writes and keeps foreign property orders meaningful
    by_index = lambda prefix: sorted((n for n in names if n.startswith(prefix)), key=lambda n: int(n.rsplit("_", 1)[1]))
    extra = by_index("f_rest_")
    if len(extra) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise AssertionError(f"{path}: {len(extra)} f_rest_* properties, expected {3 * (max_sh_degree + 1) ** 2 - 3}")
    f_rest = np.stack([col(n) for n in extra], axis=1) if extra else np.zeros((xyz.shape[0], 0))
    f_rest = f_rest.reshape(f_rest.shape[0], 3, (max_sh_degree + 1) ** 2 - 1)                     # :431
    scales = np.stack([col(n) for n in by_index("scale_")], axis=1)
    rots = np.stack([col(n) for n in by_index("rot_")], axis=1)
    t = lambda a: torch.tensor(a, dtype=torch.float, device=device)
    return dict(xyz=t(xyz), features_dc=t(f_dc).transpose(1, 2).contiguous(),
                features_rest=t(f_rest).transpose(1, 2).contiguous(), opacity=t(opac), scaling=t(scales), rotation=t(rots))


# ---------------------------------------------------------------------------------------------------
# The slice of the `plyfile` API that gs_renderer.py uses (`PlyElement.describe`, `PlyData([el]).write`,
# `PlyData.read(path).elements[0]` with `[name]` and `.properties[i].name`, gs_renderer.py:414-415, 418-441),
# so that the reference's own `save_ply` / `load_ply` run unmodified where `plyfile` is not installed:
#     import sys, dreamgaussian_amd.ply as p; sys.modules.setdefault("plyfile", p)
# ---------------------------------------------------------------------------------------------------
class PlyProperty:
    def __init__(self, name):
        self.name = name


class PlyElement:
    def __init__(self, name, data):
        self.name, self.data = name, data

    @staticmethod
    def describe(data, name):
        return PlyElement(name, np.asarray(data))

    @property
    def properties(self):
        return [PlyProperty(n) for n in self.data.dtype.names]

    def __getitem__(self, key):
        return self.data[key]

    def __len__(self):
        return len(self.data)


class PlyData:
    def __init__(self, elements=()):
        self.elements = list(elements)

    def write(self, path):
        el = self.elements[0]
        names = el.data.dtype.names
        kinds = {np.dtype(t): n for n, t in (("float", "<f4"), ("double", "<f8"), ("uchar", "u1"), ("int", "<i4"))}
        header = "ply\nformat binary_little_endian 1.0\nelement %s %d\n" % (el.name, len(el.data))
        header += "".join(f"property {kinds[el.data.dtype[n].newbyteorder('<')]} {n}\n" for n in names) + "end_header\n"
        with open(path, "wb") as fh:
            fh.write(header.encode("ascii"))
            fh.write(np.ascontiguousarray(el.data.astype(el.data.dtype.newbyteorder("<"))).tobytes())

    @staticmethod
    def read(path):
        return PlyData([PlyElement("vertex", read_vertex_table(path))])
