"""distCUDA2 (simple-knn/simple_knn.cu:185-221 semantics: exact 3-NN mean squared distance,
self excluded by index) against scipy's cKDTree."""
import numpy as np
import pytest
import torch

from oracle import gs_oracle as O
from simple_knn._C import distCUDA2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P", [4, 5, 63, 1000, 5000, 200_000])
def test_matches_kdtree(gpu, P):
    rs = np.random.RandomState(P)
    r = 0.5 * np.cbrt(rs.random_sample(P))
    d = rs.normal(size=(P, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = (r[:, None] * d).astype(np.float32)
    got = distCUDA2(torch.from_numpy(pts).to(gpu)).cpu().numpy()
    ref = O.nn3_mean_sqdist(pts.astype(np.float64))
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-12)


def test_duplicates_clusters_and_degenerate_axes(gpu):
    rs = np.random.RandomState(0)
    a = rs.normal(size=(3000, 3)).astype(np.float32) * 0.01
    b = rs.normal(size=(3000, 3)).astype(np.float32) * 0.01 + 5.0        # two far clusters
    c = np.repeat(rs.normal(size=(10, 3)).astype(np.float32), 5, axis=0)  # exact duplicates
    flat = rs.normal(size=(2000, 3)).astype(np.float32); flat[:, 2] = 0.25   # planar
    for pts in (np.concatenate([a, b]), np.concatenate([a, c]), flat):
        got = distCUDA2(torch.from_numpy(pts).to(gpu)).cpu().numpy()
        ref = O.nn3_mean_sqdist(pts.astype(np.float64))
        np.testing.assert_allclose(got, ref, rtol=3e-5, atol=1e-10)


def test_fewer_than_four_points(gpu):
    # simple_knn.cu:142-182: missing neighbours stay at FLT_MAX
    assert distCUDA2(torch.zeros(0, 3, device=gpu)).shape == (0,)
    out = distCUDA2(torch.tensor([[0.0, 0, 0], [1, 0, 0]], device=gpu)).cpu()
    assert out.shape == (2,) and (out > 1e37).all()
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(5, 2, device=gpu))
