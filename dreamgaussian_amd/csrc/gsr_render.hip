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
gsr_render_fwd_v0(const uint32_t* __restrict__ tile_off, const SplatRec* __restrict__ recs,
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
gsr_render_bwd_v0(const uint32_t* __restrict__ tile_off, const SplatRec* __restrict__ recs,
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


// =========================================================================================
// v1: per-wave LDS staging.
//
// v0 above reads one record per iteration with a scalar load and stalls for its full latency
// (measured 290 cycles per list entry at 1M Gaussians). v1 keeps the "four independent waves,
// 8x8 pixels each, no workgroup barrier" structure but moves the list walk off the latency
// path: a wave fetches 64 consecutive list entries with vector loads (one record per lane),
// tests each record's alpha>=1/255 box against its own 8x8 block in parallel, compacts the
// survivors into a private LDS ring (ballot + mbcnt prefix) and then runs the per-pixel loop
// over the survivors only, reading them back with wave-uniform (broadcast) ds_read_b128
// issued one entry ahead of their use (two-entry ping-pong, no register renaming).
// BY_ID: the list holds Gaussian indices and records are gathered from the per-Gaussian array
// (no sorted 64-byte copy per instance); otherwise the list IS the sorted record stream.
// =========================================================================================
#define GSR_RB 64   // list entries fetched per wave per round

template <bool BY_ID>
__device__ __forceinline__ const float4* list_record(const SplatRec* __restrict__ recs,
                                                     const uint32_t* __restrict__ ids, uint32_t i) {
    return reinterpret_cast<const float4*>(BY_ID ? recs + ids[i] : recs + i);
}

template <bool BY_ID, bool SCHED>
__global__ void __launch_bounds__(256)
gsr_render_fwd(const uint32_t* __restrict__ tile_off, const SplatRec* __restrict__ recs,
               const uint32_t* __restrict__ ids,
               const float* __restrict__ bg, int W, int H, int gx,
               float* __restrict__ out_color, float* __restrict__ out_depth,
               float* __restrict__ out_alpha, float* __restrict__ final_T,
               uint32_t* __restrict__ n_contrib, float* __restrict__ totals /*[5][H*W]*/,
               float* __restrict__ ckpt, const uint32_t* __restrict__ tile_seg,
               const uint32_t* __restrict__ tile_order /* heaviest tile first, or NULL */, int exact_cull,
               int seg_shift, uint32_t* __restrict__ tile_last /* max list position blended in the tile */) {
    __shared__ float4 stage[4][3][GSR_RB + 2];             // 12.4 KiB: [wave][field group][slot (+2 pad)]
    const int tile = tile_order ? (int)tile_order[blockIdx.x] : (int)blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int bx = (tile % gx) * GSR_TILE + (wave & 1) * 8;
    const int by = (tile / gx) * GSR_TILE + (wave >> 1) * 8;
    if (bx >= W || by >= H) return;                       // whole block outside the image
    const int px = bx + (lane & 7), py = by + (lane >> 3);
    const bool inside = (px < W) && (py < H);
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t start = tile_off[tile], end = tile_off[tile + 1];
    float4* __restrict__ sa = stage[wave][0];
    float4* __restrict__ sb = stage[wave][1];
    float4* __restrict__ sc = stage[wave][2];

    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f;
    uint32_t last = 0;
    bool done = !inside;

    // ea = x y qa qb | eb = qc opac r g | ec = b depth pos -
#define GSR_FWD_ENTRY(ea, eb, ec, valid)                                                       \
    {                                                                                          \
        const float dx = ea.x - pxf, dy = ea.y - pyf;                                          \
        const float power = ea.z * dx * dx + eb.x * dy * dy + ea.w * dx * dy; /* log2 units */ \
        const float alpha = fminf(0.99f, eb.y * fast_exp2(power));                             \
        const bool ok = (valid) && !done && (power <= 0.f) && (alpha >= (1.0f / 255.0f));      \
        const float test_T = T * (1.f - alpha);                                                \
        const bool stop = ok && (test_T < 0.0001f);                                            \
        const bool acc = ok && !stop;                                                          \
        const float w = acc ? alpha * T : 0.f;                                                 \
        C0 += eb.z * w; C1 += eb.w * w; C2 += ec.x * w;                                        \
        D += ec.y * w; A += w;                                                                 \
        T = acc ? test_T : T;                                                                  \
        last = acc ? __float_as_uint(ec.z) : last;                                             \
        done = done || stop;                                                                   \
    }

    const uint32_t seg_slot0 = tile_seg[tile];
    // padding slots are read (never used): keep them finite so that 0 * garbage stays 0
    for (int q = lane; q < 3 * (GSR_RB + 2); q += 64) (&stage[wave][0][0])[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    wave_lds_handoff();
    // Two-deep fetch pipeline: while round r is composited, the records of round r+1 (whose
    // list entries were fetched during round r-1) and the list entries of round r+2 are in flight.
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc = ra;
    uint32_t id_next = 0;
    if (start + lane < end) {
        const float4* __restrict__ p = list_record<BY_ID>(recs, ids, start + lane);
        ra = p[0]; rb = p[1]; rc = p[2];
    }
    if (BY_ID && start + GSR_RB + lane < end) id_next = ids[start + GSR_RB + lane];
    for (uint32_t base = start; base < end; base += GSR_RB) {
        if (__ballot(!done) == 0ull) break;
        float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na, nc = na;
        uint32_t id_next2 = 0;
        {
            const uint32_t i1 = base + GSR_RB + lane;
            if (i1 < end) {
                const float4* __restrict__ p = reinterpret_cast<const float4*>(BY_ID ? recs + id_next : recs + i1);
                na = p[0]; nb = p[1]; nc = p[2];
            }
            const uint32_t i2 = base + 2 * GSR_RB + lane;
            if (BY_ID && i2 < end) id_next2 = ids[i2];
        }
        const uint32_t rel = base - start;
        if (rel != 0u && (rel & ((1u << seg_shift) - 1u)) == 0u) {    // segment cut: checkpoint for the backward
            float* c = ckpt + (size_t)(seg_slot0 + (rel >> seg_shift) - 1u) * GSR_CKPT_FLOATS + (wave * 64 + lane);
            c[0] = T; c[256] = C0; c[512] = C1; c[768] = C2; c[1024] = D; c[1280] = A;
        }
        const uint32_t i = base + lane;
        bool hit = false;
        if (i < end) {    // can alpha reach 1/255 anywhere in this wave's 8x8 block?
            if (exact_cull) {
                hit = rect_max_power(ra.x, ra.y, ra.z, ra.w, rb.x, (float)bx, (float)(bx + 7), (float)by, (float)(by + 7))
                      >= min_visible_power(rb.y);
            } else {
                const uint32_t bbx = __float_as_uint(rc.z), bby = __float_as_uint(rc.w);
                hit = !(unpack_hi16(bbx) < bx || unpack_lo16(bbx) > bx + 7 ||
                        unpack_hi16(bby) < by || unpack_lo16(bby) > by + 7);
            }
        }
        const unsigned long long mask = __ballot(hit);
        if (mask != 0ull) {
            const int n = __popcll(mask);
            if (hit) {
                const uint32_t pos = lanes_below(mask);
                rc.z = __uint_as_float(i - start + 1);    // 1-based list position replaces the box
                sa[pos] = ra; sb[pos] = rb; sc[pos] = rc;
            }
            wave_lds_handoff();
            float4 e0a = sa[0], e0b = sb[0], e0c = sc[0];
            // two entries per trip, both unconditional (the second is masked off on an odd tail) so
            // that the body stays one basic block and the LDS reads are issued ahead of their use
            for (int j = 0; j < n; j += 2) {              // slots n, n+1 are padding: read, masked
                const float4 e1a = sa[j + 1], e1b = sb[j + 1], e1c = sc[j + 1];   // in flight during entry j
                GSR_FWD_ENTRY(e0a, e0b, e0c, true)
                e0a = sa[j + 2]; e0b = sb[j + 2]; e0c = sc[j + 2];               // in flight during entry j+1
                GSR_FWD_ENTRY(e1a, e1b, e1c, j + 1 < n)
                if constexpr (SCHED) {   // pin the interleave: reads j+1 | math j | reads j+2 | math j+1
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 22, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 22, 0);
                }
            }
            wave_lds_handoff();                           // reads above precede the next round's writes
        }
        ra = na; rb = nb; rc = nc; id_next = id_next2;
    }
#undef GSR_FWD_ENTRY
    {   // how deep the backward has to walk this tile's list
        const uint32_t wl = wave_max_u32(last);
        if (lane == 0 && wl != 0u) atomicMax(&tile_last[tile], wl);
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
        totals[pix] = C0; totals[HW + pix] = C1; totals[2 * HW + pix] = C2;   // sums without background
        totals[3 * HW + pix] = D; totals[4 * HW + pix] = A;
    }
}

// Backward, same staging walked back to front. The ten per-Gaussian sums leave the wave through
// a transposing reduction: v_permlane32_swap / v_permlane16_swap fold ten registers into three
// whose 16-lane rows each carry one quantity, four DPP row rotations finish the rows, and
// three atomic instructions (one lane per row) add them to the Gaussian's accumulator -- 28
// cross-lane instructions and 3 atomics per (block, Gaussian) instead of 60 and 10.
template <bool BY_ID>
__global__ void __launch_bounds__(256)
gsr_render_bwd(const uint32_t* __restrict__ tile_off, const SplatRec* __restrict__ recs,
               const uint32_t* __restrict__ ids,
               const float* __restrict__ bg, int W, int H, int gx,
               const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
               const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
               const float* __restrict__ dL_dalpha, float* __restrict__ g2d) {
    __shared__ float4 stage[4][4][GSR_RB];                 // 16 KiB
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
    float4* __restrict__ sa = stage[wave][0];
    float4* __restrict__ sb = stage[wave][1];
    float4* __restrict__ sc = stage[wave][2];
    float4* __restrict__ sd = stage[wave][3];

    // which accumulator slot each lane's reduced registers carry: push the slot numbers through
    // the same swap network once (robust to the swap direction convention)
    const uint32_t slot0 = tag16(tag32(0u, 5u), tag32(1u, 6u));
    const uint32_t slot1 = tag16(tag32(2u, 7u), tag32(3u, 8u));
    const uint32_t slot2 = tag16(tag32(4u, 9u), tag32(10u, 11u));   // 10, 11: padding slots (zeros)
    const bool row_leader = (lane & 15) == 0;

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
    const int wave_last = (int)__builtin_amdgcn_readfirstlane(wave_max_u32(last_contrib));

    float T = T_final;
    float accC0 = 0.f, accC1 = 0.f, accC2 = 0.f, accD = 0.f, accA = 0.f;
    float last_alpha = 0.f, lastC0 = 0.f, lastC1 = 0.f, lastC2 = 0.f, lastD = 0.f;

    // ea = x y qa qb | eb = qc opac r g | ec = b depth pos - | ed = id - - -
#define GSR_BWD_ENTRY(ea, eb, ec, ed)                                                            \
    {                                                                                            \
        const float qa = ea.z, qb = ea.w, qc = eb.x, opac = eb.y;                                \
        const float cr = eb.z, cg = eb.w, cb = ec.x, cdepth = ec.y;                              \
        const uint32_t kpos = __float_as_uint(ec.z);                                             \
        const float dx = ea.x - pxf, dy = ea.y - pyf;                                            \
        const float power = qa * dx * dx + qc * dy * dy + qb * dx * dy;                          \
        const float G = fast_exp2(power);                                                        \
        const float alpha = fminf(0.99f, opac * G);                                              \
        const bool ok = (kpos <= last_contrib) && (power <= 0.f) && (alpha >= (1.0f / 255.0f));  \
        if (__ballot(ok) != 0ull) {                       /* somebody in this block blended it */ \
            float dL_dal = 0.f, w = 0.f;                  /* stay 0 in lanes that did not blend */  \
            if (ok) {                                                                            \
                const float oma_inv = fast_rcp(1.f - alpha);                                     \
                T = T * oma_inv;                                                                 \
                w = alpha * T;                                                                   \
                accC0 = last_alpha * lastC0 + (1.f - last_alpha) * accC0; lastC0 = cr;           \
                accC1 = last_alpha * lastC1 + (1.f - last_alpha) * accC1; lastC1 = cg;           \
                accC2 = last_alpha * lastC2 + (1.f - last_alpha) * accC2; lastC2 = cb;           \
                accD = last_alpha * lastD + (1.f - last_alpha) * accD; lastD = cdepth;           \
                accA = last_alpha + (1.f - last_alpha) * accA;                                   \
                dL_dal = (cr - accC0) * gC0 + (cg - accC1) * gC1 + (cb - accC2) * gC2            \
                       + (cdepth - accD) * gD + (1.f - accA) * gA;                               \
                dL_dal *= T;                                                                     \
                last_alpha = alpha;                                                              \
                dL_dal += (-T_final * oma_inv) * bg_dot;                                         \
            }                                                                                    \
            const float Gm = ok ? G : 0.f;                /* G may be inf where power > 0 */      \
            const float dL_dG = opac * dL_dal;                                                   \
            const float gdx = Gm * dx, gdy = Gm * dy;                                            \
            const float v0 = dL_dG * (2.f * qa * gdx + qb * gdy);  /* mean2D.x (ln2*0.5W in K6) */ \
            const float v1 = dL_dG * (2.f * qc * gdy + qb * gdx);  /* mean2D.y */                \
            const float v2 = -0.5f * gdx * dx * dL_dG;             /* dL/dA */                   \
            const float v3 = -gdx * dy * dL_dG;                    /* dL/dB */                   \
            const float v4 = -0.5f * gdy * dy * dL_dG;             /* dL/dC */                   \
            const float v5 = Gm * dL_dal;                          /* dL/dopacity */             \
            const float v6 = w * gC0, v7 = w * gC1, v8 = w * gC2;  /* dL/drgb */                 \
            const float v9 = w * gD;                               /* dL/ddepth */               \
            const float t0 = row_sum16(red16(red32(v0, v5), red32(v1, v6)));                     \
            const float t1 = row_sum16(red16(red32(v2, v7), red32(v3, v8)));                     \
            const float t2 = row_sum16(red16(red32(v4, v9), 0.f));                               \
            if (row_leader) {                                                                    \
                const uint32_t gid = __builtin_amdgcn_readfirstlane(__float_as_uint(ed.x));      \
                float* dst = g2d + (size_t)gid * GSR_G2D_STRIDE;                                 \
                atomicAdd(dst + slot0, t0);                                                      \
                atomicAdd(dst + slot1, t1);                                                      \
                if (slot2 < 10u) atomicAdd(dst + slot2, t2);                                     \
            }                                                                                    \
        }                                                                                        \
    }

    for (int hi = wave_last; hi > 0; hi -= GSR_RB) {      // list positions (hi-64, hi], 1-based
        const int k = hi - lane;                          // lane 0 holds the farthest entry
        bool hit = false;
        float4 ra, rb, rc, rd;
        if (k >= 1) {
            const float4* __restrict__ p = list_record<BY_ID>(recs, ids, start + (uint32_t)k - 1u);
            ra = p[0]; rb = p[1]; rc = p[2]; rd = p[3];
            const uint32_t bbx = __float_as_uint(rc.z), bby = __float_as_uint(rc.w);
            hit = !(unpack_hi16(bbx) < bx || unpack_lo16(bbx) > bx + 7 ||
                    unpack_hi16(bby) < by || unpack_lo16(bby) > by + 7);
        }
        const unsigned long long mask = __ballot(hit);
        if (mask == 0ull) continue;
        const int n = __popcll(mask);
        if (hit) {
            const uint32_t pos = lanes_below(mask);       // ascending lane = descending position
            rc.z = __uint_as_float((uint32_t)k);
            sa[pos] = ra; sb[pos] = rb; sc[pos] = rc; sd[pos] = rd;
        }
        wave_lds_handoff();
        float4 e0a = sa[0], e0b = sb[0], e0c = sc[0], e0d = sd[0];
        for (int j = 0; j < n; j += 2) {
            const int j1 = min(j + 1, n - 1);
            const float4 e1a = sa[j1], e1b = sb[j1], e1c = sc[j1], e1d = sd[j1];
            GSR_BWD_ENTRY(e0a, e0b, e0c, e0d)
            const int j2 = min(j + 2, n - 1);
            e0a = sa[j2]; e0b = sb[j2]; e0c = sc[j2]; e0d = sd[j2];
            if (j + 1 < n) GSR_BWD_ENTRY(e1a, e1b, e1c, e1d)
        }
        wave_lds_handoff();
    }
#undef GSR_BWD_ENTRY
}

#define GSR_FWD_INST(B, S) template __global__ void gsr_render_fwd<B, S>(const uint32_t*, const SplatRec*, const uint32_t*, const float*, int, int, int, float*, float*, float*, float*, uint32_t*, float*, float*, const uint32_t*, const uint32_t*, int, int, uint32_t*);
GSR_FWD_INST(false, false) GSR_FWD_INST(true, false) GSR_FWD_INST(false, true) GSR_FWD_INST(true, true)
#undef GSR_FWD_INST
template __global__ void gsr_render_bwd<false>(const uint32_t*, const SplatRec*, const uint32_t*, const float*, int, int, int, const float*, const uint32_t*, const float*, const float*, const float*, float*);
template __global__ void gsr_render_bwd<true>(const uint32_t*, const SplatRec*, const uint32_t*, const float*, int, int, int, const float*, const uint32_t*, const float*, const float*, const float*, float*);


// =========================================================================================
// Backward, FRONT TO BACK and depth-segmented (default).
//
// dL/dalpha_i = T_i (c_i.g) - [ sum_{j>i} w_j (c_j.g) + T_final (bg.g) ] / (1 - alpha_i)
// with c.g the incoming-gradient-weighted scalar cr gC0 + cg gC1 + cb gC2 + depth gD + gA, and
// sum_{j>i} = total - prefix_i - w_i (c_i.g): only FORWARD-running quantities (T and one scalar
// prefix) are needed, the five per-channel "colour behind" recurrences of the back-to-front
// form collapse into one, there is no T/(1-alpha) unrolling, and a tile's list can be cut into
// independent segments: the forward leaves (T, C0, C1, C2, D, A) per pixel at every
// 2^seg_shift-th list position, and workgroup (tile, s) starts from checkpoint s. The longest
// sequential walk drops from the tile's whole list (8.5k entries at 1M Gaussians) to one segment.
// =========================================================================================
template <bool BY_ID, bool ACC_LDS>
__global__ void __launch_bounds__(256, 8)   // <= 64 VGPRs: the kernel leans on 8 waves/SIMD to cover its cross-lane chains
gsr_render_bwd_f2b(const uint32_t* __restrict__ tile_off, const SplatRec* __restrict__ recs,
                   const uint32_t* __restrict__ ids,
                   const float* __restrict__ bg, int W, int H, int gx,
                   const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                   const float* __restrict__ totals, const float* __restrict__ ckpt,
                   const uint32_t* __restrict__ tile_seg,
                   const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                   const float* __restrict__ dL_dalpha, float* __restrict__ g2d, int exact_cull, int seg_shift,
                   const uint32_t* __restrict__ plan_tile, const uint32_t* __restrict__ plan_off,
                   const unsigned long long* __restrict__ plan_total) {
    constexpr bool acc_lds = ACC_LDS;
    __shared__ float4 stage[4][4][GSR_RB + 2];             // 16.5 KiB (+2 pad slots)
    // acc_lds: the four waves of the workgroup first add their per-Gaussian sums into an LDS table
    // [segment position][12] and the table is flushed with coalesced global atomics at the end --
    // one memory-side atomic request per (segment, Gaussian) cache line instead of three per
    // (8x8 block, Gaussian). Measured: the global atomics were 26% of this kernel (agent-scope
    // float atomics execute at the memory side on the multi-XCD part).
    extern __shared__ __attribute__((aligned(16))) float acc[];   // [(1 << seg_shift) * GSR_G2D_STRIDE] when acc_lds
    // work list built by gsr_bwd_plan: entry b = (tile, segment) with at least one blended position
    if (blockIdx.x >= (uint32_t)plan_total[0]) return;
    const int tile = (int)plan_tile[blockIdx.x];
    const uint32_t seg = blockIdx.x - plan_off[tile];
    const uint32_t start = tile_off[tile];
    const uint32_t n = tile_off[tile + 1] - start;
    const uint32_t seg_lo = seg << seg_shift;              // this workgroup: list positions (seg_lo, seg_hi]
    if (seg_lo >= n) return;                              // (block-uniform)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int bx = (tile % gx) * GSR_TILE + (wave & 1) * 8;
    const int by = (tile / gx) * GSR_TILE + (wave >> 1) * 8;
    if (acc_lds) {
        for (int q = threadIdx.x; q < (GSR_G2D_STRIDE << seg_shift); q += 256) acc[q] = 0.f;
        __syncthreads();
    }
    bool active = (bx < W) && (by < H);                   // wave-uniform; no early return: barrier below
    const int px = bx + (lane & 7), py = by + (lane >> 3);
    const bool inside = (px < W) && (py < H);
    const float pxf = (float)px, pyf = (float)py;
    float4* __restrict__ sa = stage[wave][0];
    float4* __restrict__ sb = stage[wave][1];
    float4* __restrict__ sc = stage[wave][2];
    float4* __restrict__ sd = stage[wave][3];

    float T_final = 1.f, gC0 = 0.f, gC1 = 0.f, gC2 = 0.f, gD = 0.f, gA = 0.f, Cg_total = 0.f;
    uint32_t last_contrib = 0;
    if (inside) {
        const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
        T_final = final_T[pix];
        last_contrib = n_contrib[pix];
        gC0 = dL_dcolor[pix]; gC1 = dL_dcolor[HW + pix]; gC2 = dL_dcolor[2 * HW + pix];
        gD = dL_ddepth[pix]; gA = dL_dalpha[pix];
        Cg_total = totals[pix] * gC0 + totals[HW + pix] * gC1 + totals[2 * HW + pix] * gC2
                 + totals[3 * HW + pix] * gD + totals[4 * HW + pix] * gA;
    }
    const uint32_t wave_last = __builtin_amdgcn_readfirstlane(wave_max_u32(last_contrib));
    active = active && (wave_last > seg_lo);              // else nothing in this segment was blended here
    const uint32_t seg_hi = active ? min(seg_lo + (1u << seg_shift), wave_last) : seg_lo;
    // everything behind entry i: (total + T_final bg.g) - prefix_i - w_i (c_i.g)
    const float Cg_behind0 = Cg_total + T_final * (bg[0] * gC0 + bg[1] * gC1 + bg[2] * gC2);

    float T = 1.f, Cgf = 0.f;                             // transmittance and c.g prefix before the segment
    if (seg > 0) {
        const float* c = ckpt + (size_t)(tile_seg[tile] + seg - 1u) * GSR_CKPT_FLOATS + (wave * 64 + lane);
        T = c[0];
        Cgf = c[256] * gC0 + c[512] * gC1 + c[768] * gC2 + c[1024] * gD + c[1280] * gA;
    }

    const uint32_t slot0 = tag16(tag32(0u, 5u), tag32(1u, 6u));
    const uint32_t slot1 = tag16(tag32(2u, 7u), tag32(3u, 8u));
    const uint32_t slot2 = tag16(tag32(4u, 9u), tag32(10u, 11u));   // 10, 11: padding slots (zeros)
    const bool row_leader = (lane & 15) == 0;
    // LDS path: lanes 0,1,2 of every 16-lane row carry t0,t1,t2 -> one ds_add for all ten sums
    const uint32_t l15 = (uint32_t)lane & 15u;
    const uint32_t myslot = l15 == 0u ? slot0 : (l15 == 1u ? slot1 : slot2);
    const bool lds_lane = (l15 < 3u) && (myslot < 10u);

    // ea = x y qa qb | eb = qc opac r g | ec = b depth pos - | ed = id - - -
#define GSR_F2B_ENTRY(ea, eb, ec, ed, valid)                                                     \
    {                                                                                            \
        const float qa = ea.z, qb = ea.w, qc = eb.x, opac = eb.y;                                \
        const uint32_t kpos = __float_as_uint(ec.z);                                             \
        const float dx = ea.x - pxf, dy = ea.y - pyf;                                            \
        const float power = qa * dx * dx + qc * dy * dy + qb * dx * dy;                          \
        const float G = fast_exp2(power);                                                        \
        const float alpha = fminf(0.99f, opac * G);                                              \
        const bool ok = (valid) && (kpos <= last_contrib) && (power <= 0.f) && (alpha >= (1.0f / 255.0f)); \
        if (__ballot(ok) != 0ull) {                       /* somebody in this block blended it */ \
            float dL_dal = 0.f, w = 0.f;                  /* stay 0 in lanes that did not blend */  \
            if (ok) {                                                                            \
                const float cgi = eb.z * gC0 + eb.w * gC1 + ec.x * gC2 + ec.y * gD + gA;         \
                const float oma = 1.f - alpha;                                                   \
                w = alpha * T;                                                                   \
                const float wc = w * cgi;                                                        \
                dL_dal = T * cgi - (Cg_behind0 - Cgf - wc) * fast_rcp(oma);                       \
                Cgf += wc;                                                                       \
                T *= oma;                                                                        \
            }                                                                                    \
            const float Gm = ok ? G : 0.f;                /* G may be inf where power > 0 */      \
            float v0, v1, v2, v3, v4, v5;                                                        \
            if constexpr (ACC_LDS) {   /* raw moments of m = dL/d(exponent); converted once per    \
                                          (segment, Gaussian) before the flush */                \
                v5 = (opac * dL_dal) * Gm;                         /* S_0 */                     \
                v0 = v5 * dx; v1 = v5 * dy;                        /* S_x, S_y */                \
                v2 = v0 * dx; v3 = v0 * dy; v4 = v1 * dy;          /* S_xx, S_xy, S_yy */        \
            } else {                                                                             \
                const float dL_dG = opac * dL_dal;                                               \
                const float gdx = Gm * dx, gdy = Gm * dy;                                        \
                v0 = dL_dG * (2.f * qa * gdx + qb * gdy);          /* mean2D.x (ln2*0.5W in K6) */ \
                v1 = dL_dG * (2.f * qc * gdy + qb * gdx);          /* mean2D.y */                \
                v2 = -0.5f * gdx * dx * dL_dG;                     /* dL/dA */                   \
                v3 = -gdx * dy * dL_dG;                            /* dL/dB */                   \
                v4 = -0.5f * gdy * dy * dL_dG;                     /* dL/dC */                   \
                v5 = Gm * dL_dal;                                  /* dL/dopacity */             \
            }                                                                                    \
            const float v6 = w * gC0, v7 = w * gC1, v8 = w * gC2;  /* dL/drgb */                 \
            const float v9 = w * gD;                               /* dL/ddepth */               \
            const float t0 = row_sum16(red16(red32(v0, v5), red32(v1, v6)));                     \
            const float t1 = row_sum16(red16(red32(v2, v7), red32(v3, v8)));                     \
            const float t2 = row_sum16(red16(red32(v4, v9), 0.f));                               \
            if (acc_lds) {                                                                       \
                const float tv = l15 == 0u ? t0 : (l15 == 1u ? t1 : t2);                          \
                if (lds_lane) atomicAdd(&acc[(kpos - 1u - seg_lo) * GSR_G2D_STRIDE + myslot], tv); \
            } else if (row_leader) {                                                             \
                const uint32_t gid = __builtin_amdgcn_readfirstlane(__float_as_uint(ed.x));      \
                float* dst = g2d + (size_t)gid * GSR_G2D_STRIDE;                                 \
                atomicAdd(dst + slot0, t0);                                                      \
                atomicAdd(dst + slot1, t1);                                                      \
                if (slot2 < 10u) atomicAdd(dst + slot2, t2);                                     \
            }                                                                                    \
        }                                                                                        \
    }

    for (int q = lane; q < 4 * (GSR_RB + 2); q += 64) (&stage[wave][0][0])[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    wave_lds_handoff();
    for (uint32_t pos0 = seg_lo; pos0 < seg_hi; pos0 += GSR_RB) {   // 0-based positions pos0 .. pos0+63
        // (a two-deep prefetch of the next round's records measured no gain here and costs 18 VGPRs)
        const uint32_t i = pos0 + lane;
        bool hit = false;
        float4 ra, rb, rc, rd;
        if (i < seg_hi) {
            const float4* __restrict__ p = list_record<BY_ID>(recs, ids, start + i);
            ra = p[0]; rb = p[1]; rc = p[2]; rd = p[3];
            if (exact_cull) {
                hit = rect_max_power(ra.x, ra.y, ra.z, ra.w, rb.x, (float)bx, (float)(bx + 7), (float)by, (float)(by + 7))
                      >= min_visible_power(rb.y);
            } else {
                const uint32_t bbx = __float_as_uint(rc.z), bby = __float_as_uint(rc.w);
                hit = !(unpack_hi16(bbx) < bx || unpack_lo16(bbx) > bx + 7 ||
                        unpack_hi16(bby) < by || unpack_lo16(bby) > by + 7);
            }
        }
        const unsigned long long mask = __ballot(hit);
        if (mask != 0ull) {
            const int cnt = __popcll(mask);
            if (hit) {
                const uint32_t pos = lanes_below(mask);   // ascending lane = ascending list position
                rc.z = __uint_as_float(i + 1u);           // 1-based list position
                sa[pos] = ra; sb[pos] = rb; sc[pos] = rc; sd[pos] = rd;
            }
            wave_lds_handoff();
            float4 e0a = sa[0], e0b = sb[0], e0c = sc[0], e0d = sd[0];
            for (int j = 0; j < cnt; j += 2) {            // slots cnt, cnt+1 are padding: read, never used
                const float4 e1a = sa[j + 1], e1b = sb[j + 1], e1c = sc[j + 1], e1d = sd[j + 1];
                GSR_F2B_ENTRY(e0a, e0b, e0c, e0d, true)
                e0a = sa[j + 2]; e0b = sb[j + 2]; e0c = sc[j + 2]; e0d = sd[j + 2];
                GSR_F2B_ENTRY(e1a, e1b, e1c, e1d, j + 1 < cnt)
            }
            wave_lds_handoff();
        }
    }
#undef GSR_F2B_ENTRY
    if (acc_lds) {
        __syncthreads();
        const uint32_t len = min(1u << seg_shift, n - seg_lo);
        // raw moments -> the accumulator layout K6 reads (gsr_device.h), once per list position
        for (uint32_t r = threadIdx.x; r < len; r += 256) {
            float* a = acc + r * GSR_G2D_STRIDE;
            const float S0 = a[5];
            if (S0 != 0.f || a[0] != 0.f || a[1] != 0.f || a[2] != 0.f || a[3] != 0.f || a[4] != 0.f) {
                const uint32_t li = start + seg_lo + r;
                const SplatRec* __restrict__ g = BY_ID ? recs + ids[li] : recs + li;
                const float qa = g->qa, qb = g->qb, qc = g->qc, op = g->opac;
                const float Sx = a[0], Sy = a[1];
                a[0] = 2.f * qa * Sx + qb * Sy;
                a[1] = 2.f * qc * Sy + qb * Sx;
                a[2] *= -0.5f; a[3] = -a[3]; a[4] *= -0.5f;
                a[5] = op != 0.f ? S0 / op : 0.f;
            }
        }
        __syncthreads();
        // flush: consecutive threads = consecutive slots of consecutive list positions
        for (uint32_t e = threadIdx.x; e < len * GSR_G2D_STRIDE; e += 256) {
            const float v = acc[e];
            if (v != 0.f) {
                const uint32_t r = e / GSR_G2D_STRIDE, slot = e - r * GSR_G2D_STRIDE;
                const uint32_t li = start + seg_lo + r;
                const uint32_t gid = BY_ID ? ids[li] : recs[li].id;
                atomicAdd(g2d + (size_t)gid * GSR_G2D_STRIDE + slot, v);
            }
        }
    }
}

#define GSR_F2B_INST(B, A) template __global__ void gsr_render_bwd_f2b<B, A>(const uint32_t*, const SplatRec*, const uint32_t*, const float*, int, int, int, const float*, const uint32_t*, const float*, const float*, const uint32_t*, const float*, const float*, const float*, float*, int, int, const uint32_t*, const uint32_t*, const unsigned long long*);
GSR_F2B_INST(false, false) GSR_F2B_INST(true, false) GSR_F2B_INST(false, true) GSR_F2B_INST(true, true)
#undef GSR_F2B_INST
