"""The density-grid oracle (oracle/fields_oracle.py) against the reference's OWN output:
tests/golden/reference_fields.npz holds grids produced by the unmodified
`GaussianModel.extract_fields` (gs_renderer.py:218-294) run on the CPU (make_golden.py part 3).
This row of the scope table is therefore PINNED to the reference, unlike the rasterizer."""
import os

import numpy as np
import pytest
import torch

from oracle import fields_oracle as F

SUM_RTOL = 2e-6      # fp32 summation of <= a few thousand positive terms, relative to the grid maximum


@pytest.fixture(scope="module")
def ref(golden_dir):
    return np.load(os.path.join(golden_dir, "reference_fields.npz"))


def case_inputs(z, name):
    R, nb, relax = z[f"{name}_params"]
    return (z[f"{name}_xyz"], z[f"{name}_opacity"], z[f"{name}_scaling"], z[f"{name}_rotation_raw"]), int(R), int(nb), float(relax)


def assert_same_support(ref, got):
    """Grid points no Gaussian of the block reaches are exactly 0 in both; a point may only be 0 on one
    side when its whole sum is below fp32's normal range there (the device exp flushes denormal results)."""
    assert (got[ref == 0] == 0).all(), "a grid point of a block without Gaussians is not 0"
    assert (ref[got == 0] < 1e-35).all(), f"0 where the reference has {ref[got == 0].max():.3e}"


def compare_with_reference(z, name, occ, center, scale, rtol=SUM_RTOL):
    assert np.array_equal(np.asarray(center, np.float32), z[f"{name}_center"]), "center must be bit-identical"
    assert float(scale) == float(z[f"{name}_scale"]), "scale must be bit-identical"
    occ = np.asarray(occ, np.float64)
    if f"{name}_occ" in z.files:
        ref = z[f"{name}_occ"].astype(np.float64)
        got = occ
    else:                                   # the 128^3 case keeps a strided sample + slab sums
        ref = z[f"{name}_occ_stride3"].astype(np.float64)
        got = occ[::3, ::3, ::3]
        slabs = z[f"{name}_occ_slab_sums"]
        assert np.abs(occ.sum(axis=(1, 2)) - slabs).max() <= 1e-6 * slabs.max()
    assert_same_support(ref, got)
    assert np.abs(got - ref).max() <= rtol * ref.max(), f"{name}: {np.abs(got - ref).max():.3e} vs {rtol * ref.max():.3e}"


@pytest.mark.parametrize("name", ["r32", "r64", "r48nb8", "r128"])
def test_oracle_matches_reference_output(ref, name):
    args, R, nb, relax = case_inputs(ref, name)
    occ, center, scale = F.extract_fields(*args, resolution=R, num_blocks=nb, relax_ratio=relax)
    compare_with_reference(ref, name, occ, center, scale)


def test_per_gaussian_quantities_are_the_reference_twins(golden_dir):
    """covariance6 restates build_covariance_from_scaling_rotation operation by operation:
    bit-identical to the reference's output stored in reference_twins.npz."""
    tw = np.load(os.path.join(golden_dir, "reference_twins.npz"))
    cov = F.covariance6(tw["scales"], tw["quat_raw"])
    assert np.array_equal(cov, tw["covariance6_mod1.0"])


def test_host_block_geometry_is_the_reference_construction():
    """dreamgaussian_amd.fields.block_geometry (the product's host side) builds the same fp32 numbers
    as the oracle's restatement of gs_renderer.py:221-225, 251-262."""
    from dreamgaussian_amd.fields import block_geometry
    for R, nb, relax in [(128, 16, 1.5), (64, 16, 1.5), (48, 8, 1.0), (32, 16, 1.5), (256, 16, 1.5)]:
        axis, split, lo, hi = block_geometry(R, nb, relax)
        oaxis, starts, lens, olo, ohi = F.block_boxes(R, nb, relax)
        assert split == R // nb and np.array_equal(axis.numpy(), oaxis)
        assert np.array_equal(lo.numpy(), olo) and np.array_equal(hi.numpy(), ohi)
        assert list(starts) == [i * split for i in range(len(lens))]


def test_low_opacity_gaussians_are_ignored(ref):
    args, R, nb, relax = case_inputs(ref, "r32")
    xyz, op, sc, rot = [a.copy() for a in args]
    base, c0, s0 = F.extract_fields(xyz, op, sc, rot, R, nb, relax)
    # appending filtered-out Gaussians far away changes neither the normalisation nor the grid
    xyz2 = np.concatenate([xyz, np.full((5, 3), 40.0, np.float32)])
    op2 = np.concatenate([op, np.full((5, 1), 0.005, np.float32)])          # == threshold: strict >, dropped
    sc2 = np.concatenate([sc, np.full((5, 3), 0.5, np.float32)])
    rot2 = np.concatenate([rot, np.tile(np.float32([1, 0, 0, 0]), (5, 1))])
    occ2, c2, s2 = F.extract_fields(xyz2, op2, sc2, rot2, R, nb, relax)
    assert np.array_equal(c0, c2) and s0 == s2 and np.array_equal(base, occ2)
