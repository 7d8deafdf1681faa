// gsr_render.hip -- per-tile alpha compositing, forward (K5) and backward (K5b).
//
// Replaces (behaviour, not code) the render half of the external rasterizer reached through
// gs_renderer.py:800-809 and main.py:273. Spec: SURVEY.md Appendix A.5 / A.6.
//
// MI355X design. A 16x16 tile is one 256-thread workgroup = four INDEPENDENT waves, each
// owning an 8x8 pixel block (one pixel per lane). The tile's Gaussians arrive as a
// depth-sorted stream of 64-byte records that every wave reads with SCALAR loads
// (s_load_dwordx16): per-Gaussian data lives in SGPRs, costs no LDS bandwidth and no VALU,
// and the per-Gaussian "does this splat reach my 8x8 block" test runs on the scalar unit
// from the record's alpha>=1/255 bounding box. There is no barrier in either loop; a wave
// leaves as soon as its own 64 pixels are saturated.
// Backward: per-lane gradients are summed across the wave with DPP row shifts/broadcasts and
// leave the wave as ONE fp32 atomic per value per (8x8 block, Gaussian) -- and only for
// Gaussians that touched the block.
#include "gsr_device.h"

namespace {
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
}

extern "C" __global__ void __launch_bounds__(256)
gsr_render_fwd(const uint32_t* __restrict__ tile_off, const SplatRec* __restrict__ recs,
               const float* __restrict__ bg, int W, int H, int gx,
               float* __restrict__ out_color, float* __restrict__ out_depth,
               float* __restrict__ out_alpha, float* __restrict__ final_T,
               uint32_t* __restrict__ n_contrib) {
    const int tile = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int bx = (tile % gx) * GSR_TILE + (wave & 1) * 8;
    const int by = (tile / gx) * GSR_TILE + (wave >> 1) * 8;
    if (bx >= W || by >= H) return;                       // whole block outside the image
    const int px = bx + (lane & 7), py = by + (lane >> 3);
    const bool inside = (px < W) && (py < H);
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t start = tile_off[tile], end = tile_off[tile + 1];

    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f;
    uint32_t last = 0;
    bool done = !inside;

    for (uint32_t i = start; i < end; ++i) {
        if (__ballot(!done) == 0ull) break;
        const SplatRec* __restrict__ g = recs + i;        // wave-uniform address
        const uint32_t bbx = g->bbx, bby = g->bby;
        if (unpack_hi16(bbx) < bx || unpack_lo16(bbx) > bx + 7 ||
            unpack_hi16(bby) < by || unpack_lo16(bby) > by + 7) continue;
        const float dx = g->x - pxf, dy = g->y - pyf;
        const float power = g->qa * dx * dx + g->qc * dy * dy + g->qb * dx * dy;   // log2 units
        const float alpha = fminf(0.99f, g->opac * fast_exp2(power));
        const bool ok = !done && (power <= 0.f) && (alpha >= (1.0f / 255.0f));
        const float test_T = T * (1.f - alpha);
        const bool stop = ok && (test_T < 0.0001f);
        const bool acc = ok && !stop;
        const float w = acc ? alpha * T : 0.f;
        C0 += g->r * w; C1 += g->g * w; C2 += g->b * w;
        D += g->depth * w; A += w;
        T = acc ? test_T : T;
        last = acc ? (i - start + 1) : last;
        done = done || stop;
    }
    if (inside) {
        const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = C0 + T * bg[0];
        out_color[HW + pix] = C1 + T * bg[1];
        out_color[2 * HW + pix] = C2 + T * bg[2];
        out_depth[pix] = D;
        out_alpha[pix] = A;
    }
}

extern "C" __global__ void __launch_bounds__(256)
gsr_render_bwd(const uint32_t* __restrict__ tile_off, const SplatRec* __restrict__ recs,
               const float* __restrict__ bg, int W, int H, int gx,
               const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
               const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
               const float* __restrict__ dL_dalpha, float* __restrict__ g2d) {
    const int tile = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int bx = (tile % gx) * GSR_TILE + (wave & 1) * 8;
    const int by = (tile / gx) * GSR_TILE + (wave >> 1) * 8;
    if (bx >= W || by >= H) return;
    const int px = bx + (lane & 7), py = by + (lane >> 3);
    const bool inside = (px < W) && (py < H);
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t start = tile_off[tile];

    float T_final = 1.f, gC0 = 0.f, gC1 = 0.f, gC2 = 0.f, gD = 0.f, gA = 0.f;
    uint32_t last_contrib = 0;
    if (inside) {
        const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
        T_final = final_T[pix];
        last_contrib = n_contrib[pix];
        gC0 = dL_dcolor[pix]; gC1 = dL_dcolor[HW + pix]; gC2 = dL_dcolor[2 * HW + pix];
        gD = dL_ddepth[pix]; gA = dL_dalpha[pix];
    }
    const float bg_dot = bg[0] * gC0 + bg[1] * gC1 + bg[2] * gC2;
    const uint32_t wave_last = __builtin_amdgcn_readfirstlane(wave_max_u32(last_contrib));

    float T = T_final;
    float accC0 = 0.f, accC1 = 0.f, accC2 = 0.f, accD = 0.f, accA = 0.f;
    float last_alpha = 0.f, lastC0 = 0.f, lastC1 = 0.f, lastC2 = 0.f, lastD = 0.f;

    for (uint32_t k = wave_last; k >= 1; --k) {           // k = 1-based position in the tile list
        const SplatRec* __restrict__ g = recs + (start + k - 1);
        const uint32_t bbx = g->bbx, bby = g->bby;
        if (unpack_hi16(bbx) < bx || unpack_lo16(bbx) > bx + 7 ||
            unpack_hi16(bby) < by || unpack_lo16(bby) > by + 7) continue;
        const float dx = g->x - pxf, dy = g->y - pyf;
        const float power = g->qa * dx * dx + g->qc * dy * dy + g->qb * dx * dy;
        const float G = fast_exp2(power);
        const float alpha = fminf(0.99f, g->opac * G);
        const bool ok = (k <= last_contrib) && (power <= 0.f) && (alpha >= (1.0f / 255.0f));
        if (__ballot(ok) == 0ull) continue;               // nobody in this block blended it

        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, v5 = 0.f, v6 = 0.f, v7 = 0.f, v8 = 0.f, v9 = 0.f;
        if (ok) {
            const float oma_inv = fast_rcp(1.f - alpha);
            T = T * oma_inv;
            const float w = alpha * T;
            accC0 = last_alpha * lastC0 + (1.f - last_alpha) * accC0; lastC0 = g->r;
            accC1 = last_alpha * lastC1 + (1.f - last_alpha) * accC1; lastC1 = g->g;
            accC2 = last_alpha * lastC2 + (1.f - last_alpha) * accC2; lastC2 = g->b;
            accD = last_alpha * lastD + (1.f - last_alpha) * accD; lastD = g->depth;
            accA = last_alpha + (1.f - last_alpha) * accA;
            float dL_dal = (g->r - accC0) * gC0 + (g->g - accC1) * gC1 + (g->b - accC2) * gC2
                         + (g->depth - accD) * gD + (1.f - accA) * gA;
            dL_dal *= T;
            last_alpha = alpha;
            dL_dal += (-T_final * oma_inv) * bg_dot;
            const float dL_dG = g->opac * dL_dal;
            const float gdx = G * dx, gdy = G * dy;
            // mean2D: dG/ddx = -G (A dx + B dy) = ln2 * G (2 qa dx + qb dy); ln2 and 0.5*W applied in K6
            v0 = dL_dG * (2.f * g->qa * gdx + g->qb * gdy);
            v1 = dL_dG * (2.f * g->qc * gdy + g->qb * gdx);
            v2 = -0.5f * gdx * dx * dL_dG;                // dL/dA
            v3 = -gdx * dy * dL_dG;                       // dL/dB (full derivative of -B dx dy)
            v4 = -0.5f * gdy * dy * dL_dG;                // dL/dC
            v5 = G * dL_dal;                              // dL/dopacity
            v6 = w * gC0; v7 = w * gC1; v8 = w * gC2;     // dL/drgb
            v9 = w * gD;                                  // dL/ddepth
        }
        v0 = wave_sum_to_lane63(v0); v1 = wave_sum_to_lane63(v1); v2 = wave_sum_to_lane63(v2);
        v3 = wave_sum_to_lane63(v3); v4 = wave_sum_to_lane63(v4); v5 = wave_sum_to_lane63(v5);
        v6 = wave_sum_to_lane63(v6); v7 = wave_sum_to_lane63(v7); v8 = wave_sum_to_lane63(v8);
        v9 = wave_sum_to_lane63(v9);
        if (lane == 63) {
            float* dst = g2d + (size_t)g->id * GSR_G2D_STRIDE;
            atomicAdd(dst + 0, v0); atomicAdd(dst + 1, v1); atomicAdd(dst + 2, v2);
            atomicAdd(dst + 3, v3); atomicAdd(dst + 4, v4); atomicAdd(dst + 5, v5);
            atomicAdd(dst + 6, v6); atomicAdd(dst + 7, v7); atomicAdd(dst + 8, v8);
            atomicAdd(dst + 9, v9);
        }
    }
}
