// gsr_fields.hip -- density grid of the Gaussians.
//
// Replaces GaussianModel.extract_fields (gs_renderer.py:218-294) and gaussian_3d_coeff
// (gs_renderer.py:64-83): occ[ix,iy,iz] = sum over the Gaussians g "of the block" of
// opacity_g * exp(-1/2 d^T Sigma_g^-1 d), d = grid point - normalised mean, where the grid is
// linspace(-1,1,R)^3 cut into chunks of `split` samples per axis and a Gaussian belongs to a
// block when its mean lies strictly inside the block's sample box grown by block_size*relax.
// The reference walks the 16^3 blocks in a Python loop and materialises [M,L,3] and [M,L,6]
// temporaries per block; here one workgroup owns one block, streams the per-Gaussian block
// ranges once, and keeps its 512 sums in registers.
//
// Numerics: every DISCRETE decision (opacity pre-filter, block membership, `power > 0 -> 0`)
// and every per-Gaussian quantity is computed with the reference's fp32 operations in the
// reference's order, one rounding per operation (`#pragma clang fp contract(off)`: hipcc
// contracts a*b+c by default), so they are bit-identical to the torch kernels the reference
// chains. Only the order in which one grid point adds its contributions differs (index order
// here, 1024-wide .sum(-1) batches there): fp32 summation error, nothing else.
//
// Bound: vector ALU (about 25 fp32 operations + one exp per (grid point, Gaussian) pair);
// HBM traffic is 56 B per Gaussian in, 4 B per grid point out, plus 8 B per (block, Gaussian)
// of range scan served by L2.
#include "gsr_device.h"
#include <float.h>

namespace {

struct FieldRec { float4 a, b, c; };     // a = x y z opacity | b = Ha Hd Hf ib | c = ic ie - -   (H = -0.5 inv)

constexpr int FLD_SCAN = 1024;           // block ranges tested per round (4 per thread)
constexpr int FLD_EVAL = 256;            // records gathered and evaluated per batch
constexpr int FLD_RING = 2048;           // ring of member indices (>= FLD_EVAL - 1 + FLD_SCAN)

__device__ __forceinline__ uint32_t fld_f2ord(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fld_ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

}  // namespace

// bbox[0..2] = min, bbox[3..5] = max of the means that pass the opacity pre-filter
// (gs_renderer.py:230-238), as order-preserving uints; initialised to 0xffffffff / 0.
extern "C" __global__ void __launch_bounds__(256)
gsr_fields_bbox(int N, const float* __restrict__ xyz, const float* __restrict__ opacity, uint32_t* __restrict__ bbox) {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        if (opacity[i] > 0.005f) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { const float v = xyz[3 * i + a]; mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v); }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
        }
    }
    __shared__ float smn[3][4], smx[3][4];
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { smn[a][threadIdx.x >> 6] = mn[a]; smx[a][threadIdx.x >> 6] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        float lo = smn[a][0], hi = smx[a][0];
        for (int w = 1; w < 4; ++w) { lo = fminf(lo, smn[a][w]); hi = fmaxf(hi, smx[a][w]); }
        if (lo <= hi) { atomicMin(&bbox[a], fld_f2ord(lo)); atomicMax(&bbox[3 + a], fld_f2ord(hi)); }
    }
}

// Per Gaussian: normalise to ~[-1,1] (gs_renderer.py:237-243), Sigma = R S S^T R^T
// (gs_renderer.py:85-131), its closed-form inverse (gs_renderer.py:70-77), and the contiguous
// range of chunks per axis whose grown box strictly contains the mean (gs_renderer.py:261-264).
extern "C" __global__ void __launch_bounds__(256)
gsr_fields_prep(int N, const float* __restrict__ xyz, const float* __restrict__ opacity,
                const float* __restrict__ scaling, const float* __restrict__ rot,
                const uint32_t* __restrict__ bbox, int nc, const float* __restrict__ box_lo,
                const float* __restrict__ box_hi, FieldRec* __restrict__ recs, uint2* __restrict__ range,
                float* __restrict__ norm_out) {
#pragma clang fp contract(off)
    const float mnx = fld_ord2f(bbox[0]), mny = fld_ord2f(bbox[1]), mnz = fld_ord2f(bbox[2]);
    const float mxx = fld_ord2f(bbox[3]), mxy = fld_ord2f(bbox[4]), mxz = fld_ord2f(bbox[5]);
    const float cx = (mnx + mxx) / 2.f, cy = (mny + mxy) / 2.f, cz = (mnz + mxz) / 2.f;
    const float extent = fmaxf(fmaxf(mxx - mnx, mxy - mny), mxz - mnz);
    const float s32 = (float)(1.8 / (double)extent);       // python float 1.8 / extent, then fp32 products
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { norm_out[0] = cx; norm_out[1] = cy; norm_out[2] = cz; norm_out[3] = extent; }
    if (i >= N) return;
    const float op = opacity[i];
    uint2 rg = make_uint2(1u, 1u);                          // x0 = 1 > x1 = 0: member of no block
    FieldRec rec;
    rec.a = rec.b = rec.c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (op > 0.005f) {
        const float px = (xyz[3 * i] - cx) * s32, py = (xyz[3 * i + 1] - cy) * s32, pz = (xyz[3 * i + 2] - cz) * s32;
        const float sx = scaling[3 * i] * s32, sy = scaling[3 * i + 1] * s32, sz = scaling[3 * i + 2] * s32;
        const float r0 = rot[4 * i], r1 = rot[4 * i + 1], r2 = rot[4 * i + 2], r3 = rot[4 * i + 3];
        const float norm = sqrtf(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);
        const float r = r0 / norm, x = r1 / norm, y = r2 / norm, z = r3 / norm;
        // L = R diag(s)
        const float L00 = (1.f - 2.f * (y * y + z * z)) * sx, L01 = (2.f * (x * y - r * z)) * sy, L02 = (2.f * (x * z + r * y)) * sz;
        const float L10 = (2.f * (x * y + r * z)) * sx, L11 = (1.f - 2.f * (x * x + z * z)) * sy, L12 = (2.f * (y * z - r * x)) * sz;
        const float L20 = (2.f * (x * z - r * y)) * sx, L21 = (2.f * (y * z + r * x)) * sy, L22 = (1.f - 2.f * (x * x + y * y)) * sz;
        // Sigma = L L^T, three-term sums in k order
        const float a = (L00 * L00 + L01 * L01) + L02 * L02;
        const float b = (L00 * L10 + L01 * L11) + L02 * L12;
        const float c = (L00 * L20 + L01 * L21) + L02 * L22;
        const float d = (L10 * L10 + L11 * L11) + L12 * L12;
        const float e = (L10 * L20 + L11 * L21) + L12 * L22;
        const float f = (L20 * L20 + L21 * L21) + L22 * L22;
        const float det = a * d * f + 2.f * e * c * b - e * e * a - c * c * d - b * b * f + 1e-24f;
        const float inv_det = 1.f / det;
        const float ia = (d * f - e * e) * inv_det, ib = (e * c - b * f) * inv_det, ic = (e * b - c * d) * inv_det;
        const float id = (a * f - c * c) * inv_det, ie = (b * c - e * a) * inv_det, iff = (a * d - b * b) * inv_det;
        rec.a = make_float4(px, py, pz, op);
        rec.b = make_float4(-0.5f * ia, -0.5f * id, -0.5f * iff, ib);
        rec.c = make_float4(ic, ie, 0.f, 0.f);
        int x0 = nc, x1 = -1, y0 = nc, y1 = -1, z0 = nc, z1 = -1;
        for (int bk = 0; bk < nc; ++bk) {
            const float lo = box_lo[bk], hi = box_hi[bk];
            if (px > lo && px < hi) { x0 = min(x0, bk); x1 = bk; }
            if (py > lo && py < hi) { y0 = min(y0, bk); y1 = bk; }
            if (pz > lo && pz < hi) { z0 = min(z0, bk); z1 = bk; }
        }
        if (x1 >= 0 && y1 >= 0 && z1 >= 0)
            rg = make_uint2((uint32_t)x0 | ((uint32_t)x1 << 8) | ((uint32_t)y0 << 16) | ((uint32_t)y1 << 24),
                            (uint32_t)z0 | ((uint32_t)z1 << 8));
    }
    recs[i] = rec;
    range[i] = rg;
}

// One workgroup = up to 512 grid points of block blockIdx.x: 128 "row slots" of FLD_PZ = 4
// z-consecutive samples sharing (x, y), so x^2 Ha + y^2 Hd and x y ib are computed once per four
// points (the same fp32 values the reference computes four times). Both halves of the workgroup
// (waves 0-1 / waves 2-3) hold the same 128 slots and each takes every other staged Gaussian;
// the two partial sums are added at the end (a fixed order: results are bit-repeatable).
// Rounds of FLD_SCAN block ranges are tested, members appended IN INDEX ORDER to an LDS ring;
// whenever FLD_EVAL members are waiting their records are gathered into LDS once (broadcast reads).
constexpr int FLD_PZ = 4;
typedef float f32x2 __attribute__((ext_vector_type(2)));

extern "C" __global__ void __launch_bounds__(256)
gsr_fields_accumulate(int N, const FieldRec* __restrict__ recs, const uint2* __restrict__ range,
                      int R, int nc, int split, const float* __restrict__ axis, float* __restrict__ occ) {
    __shared__ float4 stage[FLD_EVAL * 3];                  // 12 KiB (reused for the final half-sum exchange)
    __shared__ uint32_t ring[FLD_RING];                     // 8 KiB
    __shared__ __attribute__((aligned(16))) uint32_t wcnt[2][4][4];   // [parity][slice][wave]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 1, ht = tid & 127;
    const int blk = blockIdx.x;
    const int bxi = blk / (nc * nc), byi = (blk / nc) % nc, bzi = blk % nc;
    const int x0 = bxi * split, y0 = byi * split, z0 = bzi * split;
    const int lx = min(split, R - x0), ly = min(split, R - y0), lz = min(split, R - z0);
    const int segs = (lz + FLD_PZ - 1) / FLD_PZ;            // row slots per (x, y) row
    const int nslots = lx * ly * segs;
    if ((int)blockIdx.y * 128 >= nslots) return;

    const int slot = blockIdx.y * 128 + ht;
    const bool has_slot = slot < nslots;
    float px = 0.f, py = 0.f, pz[FLD_PZ] = {0.f, 0.f, 0.f, 0.f};
    f32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
    int out_base = 0, nz = 0;
    if (has_slot) {
        const int row = slot / segs, seg = slot - row * segs, ix = row / ly, iy = row - ix * ly;
        px = axis[x0 + ix]; py = axis[y0 + iy];
        nz = min(FLD_PZ, lz - seg * FLD_PZ);
#pragma unroll
        for (int k = 0; k < FLD_PZ; ++k) if (k < nz) pz[k] = axis[z0 + seg * FLD_PZ + k];
        out_base = ((x0 + ix) * R + (y0 + iy)) * R + z0 + seg * FLD_PZ;
    }
    const f32x2 pz01 = {pz[0], pz[1]}, pz23 = {pz[2], pz[3]};
    const bool wave_has_slots = (int)blockIdx.y * 128 + (wave & 1) * 64 < nslots;   // wave-uniform

    uint32_t head = 0, count = 0;                           // ring state, identical in every thread
    int parity = 0;
    uint2 nxt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int idx = k * 256 + tid; nxt[k] = idx < N ? range[idx] : make_uint2(1u, 1u); }

    for (int i0 = 0; i0 < N; i0 += FLD_SCAN) {
        bool hit[4];
        unsigned long long mask[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint2 r = nxt[k];
            hit[k] = (uint32_t)bxi >= (r.x & 255u) && (uint32_t)bxi <= ((r.x >> 8) & 255u) &&
                     (uint32_t)byi >= ((r.x >> 16) & 255u) && (uint32_t)byi <= (r.x >> 24) &&
                     (uint32_t)bzi >= (r.y & 255u) && (uint32_t)bzi <= ((r.y >> 8) & 255u);
            mask[k] = __ballot(hit[k]);
            if (lane == 0) wcnt[parity][k][wave] = (uint32_t)__popcll(mask[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {                       // next round's ranges fly during this round
            const int idx = i0 + FLD_SCAN + k * 256 + tid;
            nxt[k] = idx < N ? range[idx] : make_uint2(1u, 1u);
        }
        __syncthreads();
        uint32_t total = 0;                                 // members ahead of (slice k, wave w) in index order
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4 c = *reinterpret_cast<const uint4*>(&wcnt[parity][k][0]);
            const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (hit[k] && w == wave) ring[(head + count + total + lanes_below(mask[k])) & (FLD_RING - 1)] = (uint32_t)(i0 + k * 256 + tid);
                total += cw[w];
            }
        }
        count += total;
        parity ^= 1;
        const bool last = i0 + FLD_SCAN >= N;
        while (count >= (uint32_t)FLD_EVAL || (last && count > 0u)) {
            const uint32_t n = min(count, (uint32_t)FLD_EVAL);
            __syncthreads();                                // ring writes visible; previous batch fully consumed
            if ((uint32_t)tid < n) {
                const FieldRec* __restrict__ g = recs + ring[(head + tid) & (FLD_RING - 1)];
                stage[3 * tid] = g->a; stage[3 * tid + 1] = g->b; stage[3 * tid + 2] = g->c;
            }
            __syncthreads();
            if (wave_has_slots) {
                // full batches hold FLD_EVAL (even) members: half h owns the members of parity h of the block's list
                for (uint32_t j = (uint32_t)half; j < n; j += 2) {
                    const float4 a = stage[3 * j], b = stage[3 * j + 1], c = stage[3 * j + 2];
                    f32x2 p01, p23;
                    {   // -0.5 (x^2 ia + y^2 id + z^2 if) - x y ib - x z ic - y z ie with the reference's roundings
                        // (gs_renderer.py:79); the -0.5 is folded into b.xyz, exact because it is a power of two.
                        // Two z-neighbours per packed instruction (v_pk_mul_f32 / v_pk_add_f32 round per element).
#pragma clang fp contract(off)
                        const float dx = px - a.x, dy = py - a.y;
                        const float sxy = dx * dx * b.x + dy * dy * b.y;
                        const float xyb = dx * dy * b.w;
                        const f32x2 dz01 = pz01 - a.z, dz23 = pz23 - a.z;
                        p01 = (((sxy + dz01 * dz01 * b.z) - xyb) - dx * dz01 * c.x) - dy * dz01 * c.y;
                        p23 = (((sxy + dz23 * dz23 * b.z) - xyb) - dx * dz23 * c.x) - dy * dz23 * c.y;
                    }
                    const f32x2 e01 = p01 * 1.44269504088896341f, e23 = p23 * 1.44269504088896341f;
                    f32x2 w01, w23;
                    w01.x = __builtin_amdgcn_exp2f(e01.x); w01.y = __builtin_amdgcn_exp2f(e01.y);
                    w23.x = __builtin_amdgcn_exp2f(e23.x); w23.y = __builtin_amdgcn_exp2f(e23.y);
                    // `power > 0 -> weight 0` (gs_renderer.py:81): abnormal and rare, so one test per four points
                    if (fmaxf(fmaxf(p01.x, p01.y), fmaxf(p23.x, p23.y)) > 0.f) {
                        if (p01.x > 0.f) w01.x = 0.f;
                        if (p01.y > 0.f) w01.y = 0.f;
                        if (p23.x > 0.f) w23.x = 0.f;
                        if (p23.y > 0.f) w23.y = 0.f;
                    }
                    acc01 = __builtin_elementwise_fma(f32x2{a.w, a.w}, w01, acc01);
                    acc23 = __builtin_elementwise_fma(f32x2{a.w, a.w}, w23, acc23);
                }
            }
            head += n;
            count -= n;
        }
    }
    // half 1 hands its partial sums to half 0
    __syncthreads();
    if (half == 1) stage[ht] = make_float4(acc01.x, acc01.y, acc23.x, acc23.y);
    __syncthreads();
    if (half == 0 && has_slot) {
        const float4 o = stage[ht];
        const float v[FLD_PZ] = {acc01.x + o.x, acc01.y + o.y, acc23.x + o.z, acc23.y + o.w};
#pragma unroll
        for (int k = 0; k < FLD_PZ; ++k) if (k < nz) occ[out_base + k] = v[k];
    }
}
