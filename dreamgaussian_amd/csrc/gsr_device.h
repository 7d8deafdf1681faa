// gsr_device.h -- shared device-side definitions for the gfx950 splat rasterizer.
// wave = 64 lanes everywhere; no portability shims.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef GSR_MAX_VIEWS
#define GSR_MAX_VIEWS 16       // also in include/gsr.h
#endif
#define GSR_TILE 16          // binning granularity in pixels (numerics: SURVEY 7, hard part 2)
#define GSR_LOG2E 1.4426950408889634f
#define GSR_LN2 0.6931471805599453f

// One 64-byte record per (view, Gaussian), written by K1. The per-tile sorted lists hold Gaussian INDICES; the
// compositing kernels gather the records with vector loads (one record per lane: three 16-byte loads for the
// forward, the fourth dword group is for the backward and the binning).
struct __attribute__((aligned(16))) SplatRec {
    // dwords 0..11: everything the forward compositing needs (three 16-byte loads)
    float x, y;        // projected mean, pixel coordinates
    float qa, qb;      // -0.5*A*log2e, -B*log2e      (conic A,B,C; G = exp2(qa dx^2 + qb dx dy + qc dy^2))
    float qc, opac;    // -0.5*C*log2e, opacity
    float r, g;        // colour (after +0.5 and clamp at 0)
    float b, depth;    // view-space z
    uint32_t bbx;      // int16 xmin | int16 xmax << 16 : pixel columns where alpha can reach 1/255
    uint32_t bby;      // same for rows
    // dwords 12..15: backward / binning only
    uint32_t id;       // Gaussian index
    uint32_t rectx;    // emission tile rect  x0 | x1 << 16   (x1 exclusive)
    uint32_t recty;    //                     y0 | y1 << 16
    uint32_t flags;    // bit0..2 colour channel clamped at 0; bit3 emits instances
};
static_assert(sizeof(SplatRec) == 64, "SplatRec must be 64 bytes");

#define GSR_FLAG_EMIT 8u

// Depth segments: a tile's depth-sorted list is cut every 2^seg_shift positions (64, 128 or 256). Forward AND backward
// run one workgroup per (tile, segment). One 8 KiB record per segment, [plane][256 pixels of the tile]:
//   written by gsr_render_fwd_seg (the segment composited on its own, transmittance 1 coming in):
//     0 T' at the segment's end | 1..3 colour sums | 4 depth sum | 5 alpha sum | 6 list position of the last blended
//     entry (u32, 0 = none) | 7 hint: u16 floor(-256 log2(upper bound of the transmittance after this segment)) | u16 tag << 16
//   rewritten in place by gsr_render_fwd_combine (planes 0..5): the ABSOLUTE state (T, C0, C1, C2, D, A) of the pixel
//     after this segment = what segment + 1 of the backward starts from.
#define GSR_REC_PLANES 8
#define GSR_CKPT_FLOATS (GSR_REC_PLANES * 256)
#define GSR_REC_LAST (6 * 256)
#define GSR_REC_HINT (7 * 256)
// Serial walk with quad lists (gsr_render_fwd_serial<true>): plane 7 holds the four per-quad hit masks of every round of the
// segment instead -- u64 [round in segment (<= 4)][block (4)][quad (4)], bit = lane = list entry of the round -- and word
// GSR_CNT_QMASK of the counters names the views that have them: the backward re-uses them instead of repeating the
// exact ellipse-vs-quad tests (129 vector instructions per wave and round).
#define GSR_CNT_PLAN 4
#define GSR_CNT_QMASK (8 + 2 * GSR_MAX_VIEWS + 1)
#define GSR_REC_SKIPPED (-1.0f)     // plane 0 of a segment gsr_render_fwd_seg did not composite (every pixel had stopped, by its hints)
// Depth-major work list of the segment forward: level c = the c-th segment of every tile that has one; levels
// 0 .. GSR_NLEV-2 hold one segment per item, an item of the last level walks the rest of its tile's list.
#define GSR_NLEV 1023
// hint value from which a pixel has certainly stopped: 2^(-3403/256) = 0.9964e-4 < 1e-4 (margin >> fp32 rounding of the products)
#define GSR_QSAT 3403u

// per-Gaussian streaming side array for the scatter kernel (coalesced 16 B / lane)
struct __attribute__((aligned(16))) EmitRec {
    uint32_t rectx, recty, depth_bits;
    uint32_t mask;   // rects of <= 32 tiles: bit (ty-y0)*(x1-x0)+(tx-x0) set = emit into that tile
};
#define GSR_EMIT_MASK_TILES 32

// screen-space gradient accumulators written by render_bwd (atomics), read by preprocess_bwd
// [0] sum dL/dG*G*(2qa dx + qb dy)   -> mean2D.x   (times ln2 * 0.5 W later)
// [1] sum dL/dG*G*(2qc dy + qb dx)   -> mean2D.y
// [2] dL/dA  [3] dL/dB  [4] dL/dC  (true conic)   [5] dL/dopacity
// [6..8] dL/drgb  [9] dL/ddepth  [10..11] pad
#define GSR_G2D_STRIDE 12

// The backward's "live" flags: one BYTE per (view, Gaussian), set (a plain store of 1: racing writers store the same value; an atomic OR
// on a bit mask cost gsr_render_bwd_q2 13 us at 1M Gaussians) when the compositing kernel adds a non-zero row to the Gaussian's
// accumulators. They sit right behind the accumulators ([views][N][12] floats | [views][GSR_LIVE_BYTES(N)] bytes) and are cleared with them.
#define GSR_LIVE_BYTES(N) (((size_t)(N) + 15) & ~(size_t)15)      // per view, a multiple of 16 bytes
// Caller-owned arrays a kernel clears on the side (by-value kernel argument): float4 slices + up to three floats of tail each.
#define GSR_ZERO_MAX 8
struct ZeroRegions {
    float* p[GSR_ZERO_MAX];
    uint32_t n4[GSR_ZERO_MAX];       // float4s
    uint32_t tail[GSR_ZERO_MAX];     // floats behind them (0..3)
    uint32_t total4;                 // sum of n4
    int count;
};

// ONE caller-owned array the forward's per-tile compositing kernel clears on the side (GsrView.grad_clear: what the backward's
// gradient outputs are carved from): workgroup k stores zeros to slice k of n4 float4s, `per` each (host: ceil(n4 / grid)).
struct ZeroSide {
    float4* p;
    uint32_t n4;
    uint32_t per;
};

// The array the backward's gradient outputs will be carved from (GsrView.grad_clear; 248 MB at 1M Gaussians / SH 3): written
// once here, read by nobody before the per-Gaussian backward overwrites the live rows -- non-temporal stores. Round 5: these zeros
// were stored under gsr_render_bwd_q2 first (+26 us there: it is bound by vector-instruction issue at four waves per SIMD and the
// stores take issue slots from waves that want them); the serial walk has two thirds of its workgroups idle-tiled and its busy waves
// wait for memory most of the time.
template <uint32_t THREADS = 256u>
__device__ __forceinline__ void clear_side(ZeroSide zs) {
    if (zs.n4 == 0u) return;
    typedef float gsr_v4f __attribute__((ext_vector_type(4)));
    const gsr_v4f z = {0.f, 0.f, 0.f, 0.f};
    gsr_v4f* __restrict__ dst = reinterpret_cast<gsr_v4f*>(zs.p);
    const uint32_t lo = min(blockIdx.x * zs.per, zs.n4), hi = min(lo + zs.per, zs.n4);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += THREADS) __builtin_nontemporal_store(z, dst + i);
}

struct ViewConst {           // by-value kernel argument (scalar registers)
    int W, H, gx, gy;
    float tanfovx, tanfovy, focal_x, focal_y;
    float scale_modifier;
    int sh_degree;
    int raw_act;             // inputs are the raw parameters: opacity = sigmoid, scale = exp, rotation = normalise here
    int mat_t;               // bit 0 / 1: `view` / `proj` point at the transposed matrix (GSR_VIEW_VIEWMATRIX_T / _PROJMATRIX_T)
    const float* bg;
    const float* view;
    const float* proj;
    const float* campos;
};

// element i (0..15) of a camera matrix in the layout the kernels compute with, from storage that may hold its transpose
__device__ __forceinline__ int cam_index(int i, int transposed) { return transposed ? ((i & 3) << 2) | (i >> 2) : i; }

// Several cameras in one launch chain (gsr_forward_views / gsr_backward_views): the per-Gaussian kernels take the
// cameras as a by-value table and pick theirs with blockIdx.y; the per-tile kernels run over views * tiles_per_view
// tiles and find the view of a tile by division. One view = the same kernels with a table of one.
struct ViewTab { ViewConst v[GSR_MAX_VIEWS]; };
struct ViewSplit {
    int tiles_per_view;                  // gx * gy
    int N;                               // Gaussians: stride (in records) of the per-view record / gradient arrays
    unsigned long long img_stride;       // floats between two views' per-pixel scratch planes (final_T | n_contrib | totals)
    uint32_t view_mask;                  // forward compositing: the views THIS launch renders (bit v); the rest belong to the other kernel
    const float* bg[GSR_MAX_VIEWS];
};

// ---------------------------------------------------------------------------------------
// wave-level helpers (DPP; gfx9 encodings)
// ---------------------------------------------------------------------------------------

// ---- exact support test ---------------------------------------------------------------
// Largest value over the pixel rectangle [x0,x1] x [y0,y1] of the (concave, log2-scaled) exponent
//   power(p) = qa dx^2 + qb dx dy + qc dy^2,  d = (gx,gy) - p,  qa, qc < 0, 4 qa qc > qb^2.
// 0 if the centre is inside; otherwise the maximum sits on the boundary, where each edge is a
// 1-D concave quadratic maximised in closed form. alpha = o * exp2(power) can reach 1/255
// somewhere in the rectangle only if this value >= -log2(255 o): an exact (not bounding-box)
// cull of (Gaussian, pixel block) pairs that contribute nothing in the reference either.
__device__ __forceinline__ float rect_max_power(float gx, float gy, float qa, float qb, float qc,
                                                float x0, float x1, float y0, float y1) {
    const float dxl = gx - x1, dxh = gx - x0, dyl = gy - y1, dyh = gy - y0;
    const float hx = -0.5f * qb * __builtin_amdgcn_rcpf(qa);   // argmax dx = hx * dy on a dy = const edge
    const float hy = -0.5f * qb * __builtin_amdgcn_rcpf(qc);   // argmax dy = hy * dx on a dx = const edge
    float best = -3.0e38f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float dx = e ? dxh : dxl;
        const float dy = fminf(fmaxf(hy * dx, dyl), dyh);
        best = fmaxf(best, qa * dx * dx + (qb * dx + qc * dy) * dy);
        const float ey = e ? dyh : dyl;
        const float ex = fminf(fmaxf(hx * ey, dxl), dxh);
        best = fmaxf(best, qc * ey * ey + (qb * ey + qa * ex) * ex);
    }
    const bool inside = (dxl <= 0.f) && (dxh >= 0.f) && (dyl <= 0.f) && (dyh >= 0.f);
    return inside ? 0.f : best;
}
// The same test for the four 4x4 quads of an 8x8 pixel block at once: out[(iy << 1) | ix] = rect_max_power over
// [bx + 4 ix, bx + 4 ix + 3] x [by + 4 iy, by + 4 iy + 3]. The sixteen edge segments share their per-line terms
// (one vertical / horizontal line carries two segments), ~2.2x fewer instructions than four separate calls;
// bit-identical results (brute-force checked on the CPU against rect_max_power, 8M random quads).
__device__ __forceinline__ void quad_max_powers(float gx, float gy, float qa, float qb, float qc,
                                                float bx, float by, float out[4]) {
    const float hx = -0.5f * qb * __builtin_amdgcn_rcpf(qa), hy = -0.5f * qb * __builtin_amdgcn_rcpf(qc);
    const float ax = gx - bx, ay = gy - by;            // d = g - p: column c has dx = ax - c, row r has dy = ay - r
    const float dxs[4] = {ax - 3.f, ax, ax - 7.f, ax - 4.f};   // [lo, hi] of quad column 0, then of quad column 1
    const float dys[4] = {ay - 3.f, ay, ay - 7.f, ay - 4.f};
    float vline[4][2], hline[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = dxs[i], A = qa * a * a, B = qb * a, t = hy * a;
#pragma unroll
        for (int iy = 0; iy < 2; ++iy) {
            const float dy = __builtin_amdgcn_fmed3f(t, dys[2 * iy], dys[2 * iy + 1]);
            vline[i][iy] = A + (B + qc * dy) * dy;
        }
        const float e = dys[i], C = qc * e * e, D = qb * e, u = hx * e;
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
            const float dx = __builtin_amdgcn_fmed3f(u, dxs[2 * ix], dxs[2 * ix + 1]);
            hline[i][ix] = C + (D + qa * dx) * dx;
        }
    }
#pragma unroll
    for (int iy = 0; iy < 2; ++iy)
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
            const float best = fmaxf(fmaxf(vline[2 * ix][iy], vline[2 * ix + 1][iy]), fmaxf(hline[2 * iy][ix], hline[2 * iy + 1][ix]));
            const bool inside = (dxs[2 * ix] <= 0.f) && (dxs[2 * ix + 1] >= 0.f) && (dys[2 * iy] <= 0.f) && (dys[2 * iy + 1] >= 0.f);
            out[iy * 2 + ix] = inside ? 0.f : best;
        }
}
// threshold for the test above, with a margin far above fp32 rounding of the in-kernel exponent
__device__ __forceinline__ float min_visible_power(float opac) {
    return -__builtin_amdgcn_logf(255.f * opac) - 1.0e-3f;     // v_log_f32 = log2; NaN/inf for opac <= 0 never passes ">="
}

#define DPP_ROW_ROR(n) (0x120 + (n))
// every lane of a 16-lane row receives the row's total
__device__ __forceinline__ float row_sum16(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(8), 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(4), 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(2), 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(1), 0xf, 0xf, false));
    return v;
}
// number of set bits of `mask` below this lane
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// LDS produced and consumed by the SAME wave: DS operations of a wave execute in order, only the
// compiler must be kept from reordering around the hand-off
__device__ __forceinline__ void wave_lds_handoff() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier for data exchanged through LDS ONLY: __syncthreads() is a fence over every address space, and the release half
// makes a wave wait for its outstanding GLOBAL stores (s_waitcnt vmcnt(0)) in front of the barrier -- a memory round trip in the
// chain wherever results have just been stored. This one orders LDS alone.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = v > o ? v : o;
    }
    return v;
}

__device__ __forceinline__ int32_t unpack_lo16(uint32_t v) { return (int32_t)(int16_t)(v & 0xffffu); }
__device__ __forceinline__ int32_t unpack_hi16(uint32_t v) { return (int32_t)v >> 16; }
__device__ __forceinline__ uint32_t pack16(int32_t lo, int32_t hi) {
    return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
}
