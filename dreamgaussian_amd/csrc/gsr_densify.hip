// gsr_densify.hip -- per-step densification bookkeeping, the consumer of the rasterizer's
// screen-space gradient holder and radii.
//
// Replaces three lines of the training loop and GaussianModel.add_densification_stats
// (main.py:279-281, gs_renderer.py:625-627):
//     max_radii2D[vis] = max(max_radii2D[vis], radii[vis])
//     xyz_gradient_accum[vis] += norm(viewspace_points.grad[vis, :2], dim=-1, keepdim=True)
//     denom[vis] += 1                                   with vis = radii > 0
// In torch every boolean-mask index is a nonzero() with a device->host synchronisation plus a
// gather/scatter pair (about ten launches and three syncs per step); here it is one streaming
// launch over N, no synchronisation, 28 B read + 12 B written per visible Gaussian.
#include "gsr_device.h"

extern "C" __global__ void __launch_bounds__(256)
gsr_densify_stats_kernel(int N, const float* __restrict__ grad_means2D, const int32_t* __restrict__ radii,
                         float* __restrict__ xyz_gradient_accum, float* __restrict__ denom,
                         float* __restrict__ max_radii2D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int32_t r = radii[i];
    if (r <= 0) return;
    const float gx = grad_means2D[3 * i], gy = grad_means2D[3 * i + 1];
    xyz_gradient_accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.f;
    max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
}
