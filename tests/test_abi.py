"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/gsr.h declares; the Python surface has the reference's names and error behaviour
(gs_renderer.py:10-14, 745-760, 800-809). No compute calls."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "gsr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", src)) - {"gsr_stream_t"})


def test_library_exports_every_declared_symbol():
    from dreamgaussian_amd import _lib, build
    build.build(verbose=False)
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 11
    for s in syms:
        assert hasattr(lib, s), f"libgsr.so does not export {s}"
    assert set(syms) == set(_lib.EXPORTS)
    assert b"gfx950" in lib.gsr_version()


def test_struct_layout_matches_header():
    from dreamgaussian_amd import _lib
    assert ctypes.sizeof(_lib.GsrView) == 8 * 4 + 4 * 8 + 2 * 4 + 2 * 8 + 8 + 8
    assert ctypes.sizeof(_lib.GsrAlloc) == 16
    assert ctypes.sizeof(_lib.GsrStats) == 72      # nine int64 (bwd_prepared: ABI 4; speculated, pending: ABI 6)


def test_abi_version_matches_header():
    from dreamgaussian_amd import _lib
    src = open(os.path.join(ROOT, "include", "gsr.h")).read()
    assert int(re.search(r"#define GSR_ABI_VERSION (\d+)", src).group(1)) == _lib.GSR_ABI_VERSION == _lib.load().gsr_abi_version()


def test_c_argument_errors_without_gpu():
    from dreamgaussian_amd import _lib
    lib = _lib.load()
    rc = lib.gsr_forward(None, 0, 0, *([None] * 7), *([None] * 4), _lib.GsrAlloc(), _lib.GsrAlloc(),
                         _lib.GsrAlloc(), None, None)
    assert rc == -1 and b"view is NULL" in lib.gsr_last_error()
    # B cameras in one chain: the view count and the agreement of the views are checked before anything touches the GPU
    views = (_lib.GsrView * 2)()
    for v, w in zip(views, (64, 96)):
        v.image_height, v.image_width, v.tanfovx, v.tanfovy, v.scale_modifier = 64, w, 0.5, 0.5, 1.0
        v.bg = v.viewmatrix = v.projmatrix = v.campos = 0x1000      # never dereferenced on this path
    common = (0, 0, *([None] * 7), *([None] * 4), _lib.GsrAlloc(), _lib.GsrAlloc(), _lib.GsrAlloc(), None, None)
    assert lib.gsr_forward_views(views, 0, *common) == -1 and b"number of views" in lib.gsr_last_error()
    assert lib.gsr_forward_views(views, _lib.GSR_MAX_VIEWS + 1, *common) == -1 and b"number of views" in lib.gsr_last_error()
    assert lib.gsr_forward_views(views, 2, *common) == -1 and b"must agree" in lib.gsr_last_error()
    assert lib.gsr_forward_views(None, 1, *common) == -1 and b"view is NULL" in lib.gsr_last_error()
    assert lib.gsr_dist2(-1, None, None, _lib.GsrAlloc(), None) == -1
    assert lib.gsr_dist2(0, None, None, _lib.GsrAlloc(), None) == 0
    assert lib.gsr_profile_read(0, None, None, None) == 0
    assert lib.gsr_densify_stats(0, None, None, None, None, None, None) == 0
    assert lib.gsr_densify_stats(3, None, None, None, None, None, None) == -1
    a = _lib.GsrAlloc()
    assert lib.gsr_extract_fields(0, *([None] * 4), 128, 8, 16, *([None] * 5), a, None) == -1
    assert b"at least one Gaussian" in lib.gsr_last_error()
    assert lib.gsr_extract_fields(5, *([None] * 4), 128, 8, 16, *([None] * 5), a, None) == -1     # NULL inputs
    assert b"required" in lib.gsr_last_error()


def test_extract_fields_host_errors_without_gpu():
    import dreamgaussian_amd as D
    with pytest.raises(RuntimeError, match="GPU only"):
        D.extract_fields(torch.zeros(4, 3), torch.ones(4, 1), torch.ones(4, 3), torch.ones(4, 4))


def test_dropin_package_names():
    import diff_gaussian_rasterization as dgr
    from simple_knn._C import distCUDA2
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
        "projmatrix", "sh_degree", "campos", "prefiltered", "debug")
    assert callable(distCUDA2)
    r = dgr.GaussianRasterizer(raster_settings=None)
    assert isinstance(r, torch.nn.Module) and hasattr(r, "markVisible")


def _settings():
    from dreamgaussian_amd import synthetic as syn
    return syn.make_settings(syn.orbit_pose(0, 0, 2.0), 32, 32)


def test_exactly_one_of_checks_and_no_cpu_fallback():
    import diff_gaussian_rasterization as dgr
    rast = dgr.GaussianRasterizer(raster_settings=_settings())
    N = 4
    m, o = torch.zeros(N, 3), torch.ones(N, 1)
    sh, col = torch.zeros(N, 1, 3), torch.zeros(N, 3)
    s, q, cov = torch.ones(N, 3), torch.ones(N, 4), torch.ones(N, 6)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rast(means3D=m, means2D=m, opacities=o, shs=sh, colors_precomp=col, scales=s, rotations=q)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rast(means3D=m, means2D=m, opacities=o, scales=s, rotations=q)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=m, means2D=m, opacities=o, shs=sh, scales=s, rotations=q, cov3D_precomp=cov)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=m, means2D=m, opacities=o, shs=sh, scales=s)
    # the product path must fail loudly on CPU tensors: no CPU fallback, no oracle routing
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rast(means3D=m, means2D=m, opacities=o, shs=sh, scales=s, rotations=q)
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        distCUDA2(torch.rand(10, 3))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "dreamgaussian_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{f} imports the oracle"
    for f in ("diff_gaussian_rasterization/__init__.py", "simple_knn/_C.py"):
        assert "oracle" not in open(os.path.join(ROOT, f)).read()


def test_morton_order_is_the_z_curve_sort():
    """dreamgaussian_amd.morton_order against a bit-by-bit interleave on the same quantised coordinates (host tensors)."""
    import torch
    from dreamgaussian_amd.densify import morton_order
    x = torch.rand(500, 3, generator=torch.Generator().manual_seed(4)) * torch.tensor([2.0, 0.5, 1.0]) - 0.7
    x[17] = x[3]                                                        # equal codes: ties stay in index order
    p = morton_order(x).tolist()
    lo, hi = x.min(0).values, x.max(0).values
    q = ((x - lo) / (hi - lo) * 2097151.0).to(torch.int64).clamp(0, 2097151).tolist()

    def code(v):
        c = 0
        for b in range(21):
            for ax in range(3):
                c |= ((v[ax] >> b) & 1) << (3 * b + ax)
        return c
    cs = [code(v) for v in q]
    assert p == sorted(range(500), key=lambda i: (cs[i], i))
    assert p.index(3) + 1 == p.index(17)


def test_torch_binding_builds_loads_and_matches_the_header():
    """dreamgaussian_amd/_gsr_torch.so (csrc/gsr_torch.cpp: the C++ autograd binding over the C ABI) builds with g++ against torch's headers
    (no GPU needed), loads, binds to the in-tree libgsr.so and was compiled against the header's ABI version. It holds no device code
    and computes nothing: CPU tensors raise through it exactly as through the ctypes path."""
    from dreamgaussian_amd import _lib, build as b, rasterizer as R
    assert os.path.exists(b.build_binding(verbose=False))
    assert R.binding_loaded() and R._binding.abi_version() == _lib.GSR_ABI_VERSION
    src = open(os.path.join(ROOT, "dreamgaussian_amd", "csrc", "gsr_torch.cpp")).read()
    assert "oracle" not in src and "__global__" not in src and "hipLaunchKernel" not in src
    m, o, sh, s, q = torch.zeros(4, 3), torch.ones(4, 1), torch.zeros(4, 1, 3), torch.ones(4, 3), torch.ones(4, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback|There is no CPU"):
        R._binding.rasterize(m, m, sh, None, o, s, q, None, None, torch.zeros(3), torch.eye(4), torch.eye(4), torch.zeros(3), 16, 16, 1.0, 1.0, 1.0, 0,
                             False, False, False, 0, False)
