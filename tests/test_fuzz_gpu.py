"""Seeded fuzz of the HIP path against the oracle: random sizes that are not
multiples of the 16-pixel tile, random cameras, active SH degree below the stored one, degenerate
and huge scales, Gaussians behind / beside the camera, opacity extremes, scale_modifier, random
background. Same tolerances as test_parity_gpu.py."""
import math

import numpy as np
import pytest
import torch

from oracle import gs_oracle as O
import util
from util import run_hip, run_oracle, weights_for, assert_forward_close, assert_grads_close, grad_floors

pytestmark = pytest.mark.gpu


def fuzz_case(seed):
    rs = np.random.RandomState(1000 + seed)
    N = int(rs.randint(1, 1500))
    W, H = int(rs.randint(17, 200)), int(rs.randint(17, 200))
    deg_max = int(rs.randint(0, 4))
    deg = int(rs.randint(0, deg_max + 1))                       # active degree <= stored degree
    kind = "trained" if rs.rand() < 0.7 else "blob"
    sc = O.make_scene(N, deg_max, seed, kind)
    g = torch.Generator().manual_seed(seed)
    n_bad = max(1, N // 10)
    idx = torch.randperm(N, generator=g)
    sc["means3D"][idx[:n_bad], 2] += 3.0 + 3.0 * torch.rand(n_bad, generator=g)      # behind a z=+r camera
    sc["means3D"][idx[n_bad:2 * n_bad], 0] += 2.5                                      # far to the side
    sc["scales"][idx[2 * n_bad:3 * n_bad]] = 1e-6                                      # degenerate: 0.3 px floor
    sc["scales"][idx[3 * n_bad:3 * n_bad + 3]] = 0.6                                   # splats larger than the frame
    op = sc["opacities"]
    op[idx[4 * n_bad:5 * n_bad]] = 0.0
    op[idx[5 * n_bad:6 * n_bad]] = 1e-4
    op[idx[6 * n_bad:7 * n_bad]] = 0.999
    el, az, r = rs.uniform(-60, 60), rs.uniform(-180, 180), rs.uniform(1.2, 3.0)
    S = O.make_settings(O.orbit_pose(el, az, r), W, H, fovy_deg=float(rs.uniform(30, 70)), sh_degree=deg,
                        bg=tuple(rs.rand(3)), scale_modifier=float(rs.choice([1.0, 0.5, 1.7])))
    return sc, S, W, H


@pytest.mark.parametrize("seed", range(12))
def test_fuzz(gpu, seed):
    sc, S, W, H = fuzz_case(seed)
    w = weights_for(H, W, seed=seed)
    ho, hg, st = run_hip(sc, S, gpu, w)
    oo, og, aux = run_oracle(sc, S, w, torch.float64)
    assert st["V"] == aux["V"]
    assert_forward_close(ho, oo, aux)
    assert_grads_close(hg, og, aux, floors=grad_floors(sc, og))
    for k in hg:
        assert torch.isfinite(hg[k]).all(), k


def fuzz_case_large(seed):
    """Tens of thousands of Gaussians, frames on both sides of the 1 024-tile line where the forward changes its mode, lists of
    several hundred to several thousand entries per tile (many depth segments, every sort class below the HBM fallback)."""
    rs = np.random.RandomState(5000 + seed)
    N = int(rs.randint(2000, 40000))
    W, H = int(rs.randint(100, 620)), int(rs.randint(100, 560))
    deg_max = int(rs.randint(0, 3))
    deg = int(rs.randint(0, deg_max + 1))
    kind = "trained" if rs.rand() < 0.6 else "blob"
    sc = O.make_scene(N, deg_max, 100 + seed, kind)
    g = torch.Generator().manual_seed(seed)
    idx = torch.randperm(N, generator=g)
    n_bad = N // 20
    sc["means3D"][idx[:n_bad], 2] += 3.0 + 3.0 * torch.rand(n_bad, generator=g)
    sc["scales"][idx[n_bad:2 * n_bad]] *= 3.0                                           # a few big splats: long lists
    sc["opacities"][idx[2 * n_bad:3 * n_bad]] = 0.97
    el, az, r = rs.uniform(-40, 40), rs.uniform(-180, 180), rs.uniform(1.6, 2.6)
    S = O.make_settings(O.orbit_pose(el, az, r), W, H, fovy_deg=float(rs.uniform(35, 60)), sh_degree=deg, bg=tuple(rs.rand(3)))
    return sc, S, W, H


@pytest.mark.parametrize("seed,env", [(0, {}), (1, {}), (2, {}), (3, {}), (4, {"fwd_mode": "seg", "seg_shift": 7}),
                                      (5, {"fwd_mode": "seg", "seg_shift": 8}), (6, {"fwd_mode": "seq"}),
                                      (7, {"fwd_mode": "seq", "seg_shift": 7, "fwd_lists": "block"})],
                         ids=lambda v: "-".join(f"{k}={x}" for k, x in v.items()) or "auto" if isinstance(v, dict) else str(v))
def test_fuzz_large(gpu, hooks, seed, env):
    for k, v in env.items():
        hooks.set(k, v)
    sc, S, W, H = fuzz_case_large(seed)
    w = weights_for(H, W, seed=seed)
    ho, hg, st = run_hip(sc, S, gpu, w)
    oo, og, aux = run_oracle(sc, S, w, torch.float64)
    util.assert_counts_explained(st, aux)
    assert_forward_close(ho, oo, aux)
    assert_grads_close(hg, og, aux, floors=grad_floors(sc, og), og32=util.og32_if_near_opaque(sc, S, w))
    for k in hg:
        assert torch.isfinite(hg[k]).all(), k


def test_large_frame_global_histogram_path(gpu):
    """More than 16384 tiles: the per-tile counters no longer fit the LDS histogram and K1 / K3
    fall back to global atomics (gsr_api.hip kHistLdsMaxTiles)."""
    W = H = 2100                                     # 132 x 132 = 17424 tiles
    sc = O.make_scene(60, 1, 5, "trained")
    sc["scales"] *= 0.4
    S = O.make_settings(O.orbit_pose(10.0, 40.0, 3.5), W, H, sh_degree=1)
    w = weights_for(H, W, seed=2)
    ho, hg, st = run_hip(sc, S, gpu, w)
    oo, og, aux = run_oracle(sc, S, w, torch.float64)
    assert_forward_close(ho, oo, aux)
    assert_grads_close(hg, og, aux, floors=grad_floors(sc, og))


def test_stored_degree_4_coefficients(gpu):
    """shs with 25 stored coefficients per Gaussian (rows of 75 floats: not 16-byte multiples, the
    generic staging path) and active degree 3."""
    sc = O.make_scene(900, 4, 3, "trained")
    S = O.make_settings(O.orbit_pose(-5.0, -30.0, 2.0), 150, 110, sh_degree=3)
    w = weights_for(110, 150, seed=4)
    ho, hg, st = run_hip(sc, S, gpu, w)
    oo, og, aux = run_oracle(sc, S, w, torch.float64)
    assert og["shs"][:, 16:].abs().max() == 0 and hg["shs"][:, 16:].abs().max() == 0
    assert_forward_close(ho, oo, aux)
    assert_grads_close(hg, og, aux, floors=grad_floors(sc, og))
