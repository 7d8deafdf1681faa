"""dreamgaussian_amd/ply.py: the Gaussian PLY of gs_renderer.py:376-462 (host-side data format)."""
import numpy as np
import pytest
import torch

from dreamgaussian_amd import ply


def model(N, deg, seed=0):
    g = torch.Generator().manual_seed(seed)
    K = (deg + 1) ** 2
    return dict(xyz=torch.randn(N, 3, generator=g), features_dc=torch.randn(N, 1, 3, generator=g),
                features_rest=torch.randn(N, K - 1, 3, generator=g), opacity=torch.randn(N, 1, generator=g),
                scaling=torch.randn(N, 3, generator=g), rotation=torch.randn(N, 4, generator=g))


@pytest.mark.parametrize("deg", [0, 3])
def test_round_trip_is_bit_exact_and_header_is_the_reference_attribute_list(tmp_path, deg):
    m = model(37, deg)
    path = str(tmp_path / "sub" / "model.ply")
    ply.save_ply(path, **m)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    K = (deg + 1) ** 2
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(3 * (K - 1))] + \
            ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]           # gs_renderer.py:376-389
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    assert lines[3:] == [f"property float {n}" for n in names]
    assert len(body) == 37 * 4 * len(names)
    back = ply.load_ply(path, deg)
    for k, v in m.items():
        assert back[k].shape == v.shape and torch.equal(back[k], v), k


def test_sh_blocks_are_channel_major_and_normals_zero(tmp_path):
    m = model(5, 1)
    path = str(tmp_path / "m.ply")
    ply.save_ply(path, **m)
    v = ply.read_vertex_table(path)
    # f_rest_i = features_rest.transpose(1,2).flatten(1)[:, i]  (gs_renderer.py:398): channel c, coefficient k -> c*(K-1)+k
    fr = m["features_rest"].numpy()
    for c in range(3):
        for k in range(3):
            assert np.array_equal(v[f"f_rest_{c * 3 + k}"], fr[:, k, c])
    assert not v["nx"].any() and not v["ny"].any() and not v["nz"].any()


def test_reader_accepts_other_writers(tmp_path):
    m = model(4, 0)
    names = ply.attribute_names(3, 0)
    cols = {n: np.arange(4, dtype=np.float32) + i for i, n in enumerate(names)}
    order = list(reversed(names))
    path = str(tmp_path / "other.ply")
    with open(path, "wb") as fh:
        fh.write(b"ply\nformat binary_big_endian 1.0\ncomment made elsewhere\nelement vertex 4\n")
        for n in order:
            fh.write(f"property float32 {n}\n".encode())
        fh.write(b"element face 0\nproperty list uchar int vertex_indices\nend_header\n")
        fh.write(np.stack([cols[n] for n in order], 1).astype(">f4").tobytes())
    back = ply.load_ply(path, 0)
    assert torch.equal(back["xyz"], torch.tensor(np.stack([cols["x"], cols["y"], cols["z"]], 1)))
    assert torch.equal(back["rotation"][:, 3], torch.tensor(cols["rot_3"]))
    with pytest.raises(AssertionError):
        ply.load_ply(path, 1)                   # wrong number of f_rest_* for that degree, as gs_renderer.py:426
