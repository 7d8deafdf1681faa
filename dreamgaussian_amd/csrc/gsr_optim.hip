// gsr_optim.hip -- the optimiser / densification side of DreamGaussian's training step.
//
// (1) gsr_adam_step: torch.optim.Adam(eps=1e-15) over the six parameter groups of GaussianModel.training_setup
//     (gs_renderer.py:356-374) as ONE launch: torch's default path issues ~8 elementwise kernels per group
//     (lerp_, mul_, addcmul_, sqrt, div, add_, addcdiv_ ...) = ~50 launches per step for tensors of a few 10^4
//     elements -- pure launch latency. Same arithmetic in the same order (no fma contraction), so the state is
//     interchangeable with torch's: the reference's optimiser-state surgery in densify_and_prune keeps working.
// (2) gsr_mask_compact + gsr_gather_rows: prune_points / densification_postfix (gs_renderer.py:479-545) index every
//     parameter, both Adam moments and three bookkeeping arrays with the same boolean mask: 21 nonzero() calls
//     (each a device->host synchronisation) + 21 gathers in torch; here one stable compaction of the mask (index
//     list + count, one synchronisation for the new size) and one gather launch for all tensors.
#include "gsr_device.h"

#define GSR_ADAM_MAX_TENSORS 8
struct AdamSeg {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    unsigned long long n;          // elements
    unsigned int first_block;      // blocks [first_block, next.first_block) work on this tensor
    float neg_step_size;           // float(-(lr / (1 - beta1^t))), the quotient taken in double like torch's Python scalars
};
struct AdamArgs {
    AdamSeg seg[GSR_ADAM_MAX_TENSORS];
    int count;
    float beta2, eps;
    float w1, w2;                  // float(1 - beta1), float(1 - beta2): differences taken in double
    float bias_correction2_sqrt;   // float(sqrt(1 - beta2^t))
};
#define GSR_ADAM_PER_BLOCK 2048    // elements per 256-thread block (8 per thread: two float4)

extern "C" __global__ void __launch_bounds__(256)
gsr_adam_kernel(AdamArgs a) {
    // The arithmetic of torch's default (foreach) CUDA path, op by op (torch/optim/adam.py: _multi_tensor_adam):
    //   _foreach_lerp_(exp_avg, grad, 1 - beta1)                    self + weight * (end - self)      (|weight| < 0.5)
    //   _foreach_mul_(exp_avg_sq, beta2); _foreach_addcmul_(exp_avg_sq, grad, grad, 1 - beta2)      self + value * (g * g)
    //   sqrt; _foreach_div_(., sqrt(bias_correction2)); _foreach_add_(., eps)
    //   _foreach_addcdiv_(param, exp_avg, denom, -(lr / bias_correction1))                       self + value * (m / denom)
    // each "self + a * b" written as one fma, as the contracted device code of those kernels computes it.
    int t = 0;
#pragma unroll
    for (int i = 1; i < GSR_ADAM_MAX_TENSORS; ++i) if (i < a.count && blockIdx.x >= a.seg[i].first_block) t = i;
    const AdamSeg s = a.seg[t];
    const unsigned long long base = (unsigned long long)(blockIdx.x - s.first_block) * GSR_ADAM_PER_BLOCK;
    for (int u = 0; u < GSR_ADAM_PER_BLOCK / 256; ++u) {
        const unsigned long long i = base + (unsigned long long)u * 256 + threadIdx.x;
        if (i >= s.n) break;
        const float g = s.grad[i];
        float m = s.exp_avg[i], v = s.exp_avg_sq[i];
        m = fmaf(a.w1, g - m, m);
        v = v * a.beta2;
        v = fmaf(a.w2, g * g, v);
        float denom = sqrtf(v);
        denom = denom / a.bias_correction2_sqrt;
        denom = denom + a.eps;
        s.exp_avg[i] = m; s.exp_avg_sq[i] = v;
        s.param[i] = fmaf(s.neg_step_size, m / denom, s.param[i]);
    }
}

// ---- stable compaction of a byte mask: idx[j] = position of the j-th set element; count[0] = number set ----
extern "C" __global__ void __launch_bounds__(1024)
gsr_mask_count_kernel(int N, const uint8_t* __restrict__ mask, uint32_t* __restrict__ block_count) {
    __shared__ uint32_t wsum[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const unsigned long long b = __ballot(i < N && mask[i] != 0);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t s = 0; for (int w = 0; w < 16; ++w) s += wsum[w]; block_count[blockIdx.x] = s; }
}
extern "C" __global__ void __launch_bounds__(1024)
gsr_mask_scan_kernel(int nblocks, uint32_t* __restrict__ block_count /* in: counts, out: exclusive offsets */,
                     unsigned long long* __restrict__ count) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t c = i < nblocks ? block_count[i] : 0u;
        uint32_t incl = c;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t wbase = carry;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        if (i < nblocks) block_count[i] = wbase + incl - c;
        __syncthreads();
        if (threadIdx.x == 1023) carry = wbase + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) count[0] = carry;
}
extern "C" __global__ void __launch_bounds__(1024)
gsr_mask_write_kernel(int N, const uint8_t* __restrict__ mask, const uint32_t* __restrict__ block_off,
                      uint32_t* __restrict__ idx) {
    __shared__ uint32_t wsum[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const bool set = i < N && mask[i] != 0;
    const unsigned long long b = __ballot(set);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = (uint32_t)__popcll(b);
    __syncthreads();
    uint32_t off = block_off[blockIdx.x];
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (set) idx[off + lanes_below(b)] = (uint32_t)i;
}

// ---- out_t[j, :] = in_t[idx[j], :] for up to 24 tensors of different row widths, one launch -------------------
#define GSR_GATHER_MAX_TENSORS 24
struct GatherSeg { const float* src; float* dst; int width; };
struct GatherArgs { GatherSeg seg[GSR_GATHER_MAX_TENSORS]; int count; int rows; };

extern "C" __global__ void __launch_bounds__(256)
gsr_gather_rows_kernel(GatherArgs a, const uint32_t* __restrict__ idx) {
    const int t = blockIdx.y;                        // tensor
    const GatherSeg s = a.seg[t];
    const long long total = (long long)a.rows * s.width;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int r = (int)(e / s.width), c = (int)(e - (long long)r * s.width);
        s.dst[e] = s.src[(size_t)idx[r] * s.width + c];
    }
}

// ---- dst_t = [a_t ; b_t] (rows_a rows of a, then rows_b rows of b; a NULL source stands for zeros) for up to 24 tensors
// of different row widths, one launch: cat_tensors_to_optimizer / densification_postfix (gs_renderer.py:513-552) -- the six
// parameters extended by the new Gaussians, their twelve Adam moments extended by zeros, the three accumulators reset.
struct ConcatSeg { const float* a; const float* b; float* dst; int width; };
struct ConcatArgs { ConcatSeg seg[GSR_GATHER_MAX_TENSORS]; int count; int rows_a; int rows_b; };

extern "C" __global__ void __launch_bounds__(256)
gsr_concat_rows_kernel(ConcatArgs a) {
    const ConcatSeg s = a.seg[blockIdx.y];
    const long long na = (long long)a.rows_a * s.width, total = na + (long long)a.rows_b * s.width;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256)
        s.dst[e] = e < na ? (s.a ? s.a[e] : 0.f) : (s.b ? s.b[e - na] : 0.f);
}
