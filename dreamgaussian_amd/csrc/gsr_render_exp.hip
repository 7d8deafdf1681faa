// gsr_render_exp.hip -- EXPERIMENTAL variants kept for same-box A/B this round (GSR_BWD=quad, GSR_FWD=u4).
// Included after gsr_render.hip (uses its helpers). Not the default path.
//
// STATUS: written at the end of round 1 without GPU time left; it compiles for gfx950 and has NOT run
// on hardware. Default path unchanged. First thing to do with it: GSR_BWD=quad pytest tests/test_parity_gpu.py.
//
// Why (tests/lane_stats.py): in gsr_render_bwd_f2b a wave owns an 8x8 pixel
// block and loops over every list entry that blends in ANY of its 64 pixels; at 1M Gaussians only
// 26 of 64 lanes blend per iteration. Here every 16-lane DPP row owns a 4x4 pixel quad with its
// own list: a fetched entry is tested exactly against each of the four quads, the four lists are
// byte arrays of staged slots in LDS, and the wave loops to the LONGEST of the four lists (1.2-1.5x
// fewer iterations). The ten per-Gaussian sums are reduced inside the row by a transposing DPP
// network (row_mirror, row_half_mirror, two quad_perms: 31 instructions) that leaves ONE quantity
// per lane, and each row adds ten lanes to its own Gaussian's slot of the workgroup's LDS table
// in a single ds_add. Everything else (front-to-back recurrence, checkpoints, work list, LDS table,
// raw moments, flush) is gsr_render_bwd_f2b.
#include "gsr_device.h"

namespace {

#define DPP_ROW_MIRROR 0x140
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_QUAD_XOR2 0x4E      // quad_perm [2,3,0,1]
#define DPP_QUAD_XOR1 0xB1      // quad_perm [1,0,3,2]

// One transposing step inside a 16-lane row: lanes with `hi` keep accumulating b, the others a;
// each adds what its partner lane (the permutation CTRL, an involution that flips `hi`) holds of
// the same quantity.
template <int CTRL>
__device__ __forceinline__ float qred(float a, float b, bool hi) {
    const float mine = hi ? b : a, theirs = hi ? a : b;
    return mine + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(theirs), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ uint32_t qtag(uint32_t a, uint32_t b, bool hi) {   // which quantity a lane ends up holding
    return hi ? b : a;
}

struct QuadSums { float v; };

// ten values per lane -> one value per lane: lane (b3 b2 b1 b0) of a row holds the row total of
// quantity qslot(lane) (>= 10: padding, zero)
__device__ __forceinline__ float quad_reduce10(const float v[10], int l15) {
    const bool b3 = l15 & 8, b2 = l15 & 4, b1 = l15 & 2, b0 = l15 & 1;
    const float r0 = qred<DPP_ROW_MIRROR>(v[0], v[1], b3), r1 = qred<DPP_ROW_MIRROR>(v[2], v[3], b3),
                r2 = qred<DPP_ROW_MIRROR>(v[4], v[5], b3), r3 = qred<DPP_ROW_MIRROR>(v[6], v[7], b3),
                r4 = qred<DPP_ROW_MIRROR>(v[8], v[9], b3);
    const float s0 = qred<DPP_ROW_HALF_MIRROR>(r0, r1, b2), s1 = qred<DPP_ROW_HALF_MIRROR>(r2, r3, b2),
                s2 = qred<DPP_ROW_HALF_MIRROR>(r4, 0.f, b2);
    const float t0 = qred<DPP_QUAD_XOR2>(s0, s1, b1), t1 = qred<DPP_QUAD_XOR2>(s2, 0.f, b1);
    return qred<DPP_QUAD_XOR1>(t0, t1, b0);
}
// the slot (0..9, or >= 10 for padding) of the quantity quad_reduce10 leaves in lane l15
__device__ __forceinline__ uint32_t quad_slot(int l15) {
    const bool b3 = l15 & 8, b2 = l15 & 4, b1 = l15 & 2, b0 = l15 & 1;
    const uint32_t r0 = b3 ? 1u : 0u, r1 = b3 ? 3u : 2u, r2 = b3 ? 5u : 4u, r3 = b3 ? 7u : 6u, r4 = b3 ? 9u : 8u;
    const uint32_t s0 = b2 ? r1 : r0, s1 = b2 ? r3 : r2, s2 = b2 ? 15u : r4;
    const uint32_t t0 = b1 ? s1 : s0, t1 = b1 ? 15u : s2;
    return b0 ? t1 : t0;
}


}  // namespace

__global__ void __launch_bounds__(256)
gsr_render_bwd_f2b_quad(const uint32_t* __restrict__ tile_off, const SplatRec* __restrict__ recs,
                        const uint32_t* __restrict__ ids,
                        const float* __restrict__ bg, int W, int H, int gx,
                        const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
                        const float* __restrict__ totals, const float* __restrict__ ckpt,
                        const uint32_t* __restrict__ tile_seg,
                        const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                        const float* __restrict__ dL_dalpha, float* __restrict__ g2d, int seg_shift,
                        const uint32_t* __restrict__ plan_tile, const uint32_t* __restrict__ plan_off,
                        const unsigned long long* __restrict__ plan_total) {
    __shared__ float4 stage[4][4][GSR_RB + 2];             // staged records, one slot per fetching lane (+2 zero slots)
    __shared__ uint8_t qlist[4][4][GSR_RB + 4];            // [wave][quad][k] = staged slot of the quad's k-th entry
    extern __shared__ __attribute__((aligned(16))) float acc[];   // [(1 << seg_shift) * GSR_G2D_STRIDE]
    if (blockIdx.x >= (uint32_t)plan_total[0]) return;
    const int tile = (int)plan_tile[blockIdx.x];
    const uint32_t seg = blockIdx.x - plan_off[tile];
    const uint32_t start = tile_off[tile];
    const uint32_t n = tile_off[tile + 1] - start;
    const uint32_t seg_lo = seg << seg_shift;
    if (seg_lo >= n) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int row = lane >> 4, l15 = lane & 15;            // row = quad: (row & 1, row >> 1) inside the 8x8 block
    const int bx = (tile % gx) * GSR_TILE + (wave & 1) * 8;
    const int by = (tile / gx) * GSR_TILE + (wave >> 1) * 8;
    for (int q = threadIdx.x; q < (GSR_G2D_STRIDE << seg_shift); q += 256) acc[q] = 0.f;
    for (int q = threadIdx.x; q < 4 * 4 * (GSR_RB + 4); q += 256) (&qlist[0][0][0])[q] = 0;   // stale reads stay inside the stage
    __syncthreads();
    bool active = (bx < W) && (by < H);
    const int lx = (row & 1) * 4 + (l15 & 3), ly = (row >> 1) * 4 + (l15 >> 2);
    const int px = bx + lx, py = by + ly;
    const bool inside = (px < W) && (py < H);
    const float pxf = (float)px, pyf = (float)py;
    const int cidx = wave * 64 + ly * 8 + lx;              // the forward's checkpoint slot of this pixel (row-major 8x8)
    float4* __restrict__ sa = stage[wave][0];
    float4* __restrict__ sb = stage[wave][1];
    float4* __restrict__ sc = stage[wave][2];
    float4* __restrict__ sd = stage[wave][3];
    const uint8_t* __restrict__ ql = qlist[wave][row];

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
    // deepest contributor of each quad (list positions are < 2^24: exact as floats) and of the wave
    const uint32_t row_last = (uint32_t)row_max_f((float)last_contrib);
    const uint32_t ql0 = __builtin_amdgcn_readlane(row_last, 0), ql1 = __builtin_amdgcn_readlane(row_last, 16),
                   ql2 = __builtin_amdgcn_readlane(row_last, 32), ql3 = __builtin_amdgcn_readlane(row_last, 48);
    const uint32_t wave_last = max(max(ql0, ql1), max(ql2, ql3));
    active = active && (wave_last > seg_lo);
    const uint32_t seg_hi = active ? min(seg_lo + (1u << seg_shift), wave_last) : seg_lo;
    const float Cg_behind0 = Cg_total + T_final * (bg[0] * gC0 + bg[1] * gC1 + bg[2] * gC2);

    float T = 1.f, Cgf = 0.f;
    if (seg > 0) {
        const float* c = ckpt + (size_t)(tile_seg[tile] + seg - 1u) * GSR_CKPT_FLOATS + cidx;
        T = c[0];
        Cgf = c[256] * gC0 + c[512] * gC1 + c[768] * gC2 + c[1024] * gD + c[1280] * gA;
    }
    const uint32_t myslot = quad_slot(l15);
    const bool lds_lane = myslot < 10u;
    const float qx0 = (float)(bx), qx1 = (float)(bx + 4), qy0 = (float)(by), qy1 = (float)(by + 4);

    // ea = x y qa qb | eb = qc opac r g | ec = b depth - - ; kpos = 1-based list position (row-uniform)
#define GSR_QUAD_ENTRY(ea, eb, ec, kpos, valid)                                                  \
    {                                                                                            \
        const float qa = ea.z, qb = ea.w, qc = eb.x, opac = eb.y;                                \
        const float dx = ea.x - pxf, dy = ea.y - pyf;                                            \
        const float power = qa * dx * dx + qc * dy * dy + qb * dx * dy;                          \
        const float G = fast_exp2(power);                                                        \
        const float alpha = fminf(0.99f, opac * G);                                              \
        const bool ok = (valid) && ((kpos) <= last_contrib) && (power <= 0.f) && (alpha >= (1.0f / 255.0f)); \
        if (__ballot(ok) != 0ull) {                                                              \
            float dL_dal = 0.f, w = 0.f;                                                         \
            if (ok) {                                                                            \
                const float cgi = eb.z * gC0 + eb.w * gC1 + ec.x * gC2 + ec.y * gD + gA;         \
                const float oma = 1.f - alpha;                                                   \
                w = alpha * T;                                                                   \
                const float wc = w * cgi;                                                        \
                dL_dal = T * cgi - (Cg_behind0 - Cgf - wc) * fast_rcp(oma);                       \
                Cgf += wc;                                                                       \
                T *= oma;                                                                        \
            }                                                                                    \
            const float Gm = ok ? G : 0.f;                                                       \
            float v[10];                                                                         \
            v[5] = (opac * dL_dal) * Gm;                           /* S_0 */                     \
            v[0] = v[5] * dx; v[1] = v[5] * dy;                    /* S_x, S_y */                \
            v[2] = v[0] * dx; v[3] = v[0] * dy; v[4] = v[1] * dy;  /* S_xx, S_xy, S_yy */        \
            v[6] = w * gC0; v[7] = w * gC1; v[8] = w * gC2; v[9] = w * gD;                       \
            const float tv = quad_reduce10(v, l15);                                              \
            if (lds_lane && (valid)) atomicAdd(&acc[((kpos) - 1u - seg_lo) * GSR_G2D_STRIDE + myslot], tv); \
        }                                                                                        \
    }

    for (int q = lane; q < 4 * (GSR_RB + 2); q += 64) (&stage[wave][0][0])[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    wave_lds_handoff();
    for (uint32_t pos0 = seg_lo; pos0 < seg_hi; pos0 += GSR_RB) {
        const uint32_t i = pos0 + lane;
        bool h0 = false, h1 = false, h2 = false, h3 = false;
        float4 ra, rb, rc, rd;
        if (i < seg_hi) {
            const float4* __restrict__ p = reinterpret_cast<const float4*>(recs + ids[start + i]);
            ra = p[0]; rb = p[1]; rc = p[2]; rd = p[3];
            const float thr = min_visible_power(rb.y);
            // exact ellipse-vs-quad support tests; an entry behind a quad's deepest contributor is never blended there
            h0 = (i < ql0) && rect_max_power(ra.x, ra.y, ra.z, ra.w, rb.x, qx0, qx0 + 3.f, qy0, qy0 + 3.f) >= thr;
            h1 = (i < ql1) && rect_max_power(ra.x, ra.y, ra.z, ra.w, rb.x, qx1, qx1 + 3.f, qy0, qy0 + 3.f) >= thr;
            h2 = (i < ql2) && rect_max_power(ra.x, ra.y, ra.z, ra.w, rb.x, qx0, qx0 + 3.f, qy1, qy1 + 3.f) >= thr;
            h3 = (i < ql3) && rect_max_power(ra.x, ra.y, ra.z, ra.w, rb.x, qx1, qx1 + 3.f, qy1, qy1 + 3.f) >= thr;
        }
        const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
        if ((m0 | m1 | m2 | m3) != 0ull) {
            if (h0 | h1 | h2 | h3) { sa[lane] = ra; sb[lane] = rb; sc[lane] = rc; sd[lane] = rd; }
            if (h0) qlist[wave][0][lanes_below(m0)] = (uint8_t)lane;
            if (h1) qlist[wave][1][lanes_below(m1)] = (uint8_t)lane;
            if (h2) qlist[wave][2][lanes_below(m2)] = (uint8_t)lane;
            if (h3) qlist[wave][3][lanes_below(m3)] = (uint8_t)lane;
            const int n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), n3 = __popcll(m3);
            const int nmax = max(max(n0, n1), max(n2, n3));
            const int nmine = row == 0 ? n0 : (row == 1 ? n1 : (row == 2 ? n2 : n3));
            wave_lds_handoff();
            // slots two entries ahead of their use; entries beyond the row's own list read a stale but
            // in-range slot (finite data) and are masked by `valid`
            uint32_t s0 = ql[0], s1 = ql[1];
            float4 e0a = sa[s0], e0b = sb[s0], e0c = sc[s0];
            for (int j = 0; j < nmax; j += 2) {
                const uint32_t s2 = ql[j + 2], s3 = ql[j + 3];
                const float4 e1a = sa[s1], e1b = sb[s1], e1c = sc[s1];
                GSR_QUAD_ENTRY(e0a, e0b, e0c, pos0 + s0 + 1u, j < nmine)
                e0a = sa[s2]; e0b = sb[s2]; e0c = sc[s2];
                GSR_QUAD_ENTRY(e1a, e1b, e1c, pos0 + s1 + 1u, j + 1 < nmine)
                s0 = s2; s1 = s3;
            }
            wave_lds_handoff();
        }
    }
#undef GSR_QUAD_ENTRY
    (void)sd;
    bwd_flush(acc, min(1u << seg_shift, n - seg_lo), start + seg_lo, recs, ids, g2d);
}



// =========================================================================================
// EXPERIMENTAL forward (GSR_FWD=u4; never run on hardware): gsr_render_fwd with FOUR staged entries per
// loop trip. DESIGN.md section 7: the forward runs at ~3 waves/SIMD and half of its time is stall
// (LDS read -> exp -> blend chains); evaluating four alphas before the serial transmittance updates
// gives each wave four independent chains. Everything else is gsr_render_fwd.
// =========================================================================================
__global__ void __launch_bounds__(256)
gsr_render_fwd_u4(const uint32_t* __restrict__ tile_off, const SplatRec* __restrict__ recs,
               const uint32_t* __restrict__ ids,
               const float* __restrict__ bg, int W, int H, int gx,
               float* __restrict__ out_color, float* __restrict__ out_depth,
               float* __restrict__ out_alpha, float* __restrict__ final_T,
               uint32_t* __restrict__ n_contrib, float* __restrict__ totals /*[5][H*W]*/,
               float* __restrict__ ckpt, const uint32_t* __restrict__ tile_seg,
               const uint32_t* __restrict__ tile_order /* heaviest tile first, or NULL */,
               int seg_shift, uint32_t* __restrict__ tile_last /* max list position blended in the tile */,
               const unsigned long long* __restrict__ counters, uint32_t capacity) {
    if (counters[2] > (unsigned long long)capacity) return;
    __shared__ float4 stage[4][3][GSR_RB + 4];             // [wave][field group][slot (+4 zero pad slots)]
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
    // alpha phase: independent of the running transmittance -- four of these are in flight per trip
#define GSR_U4_ALPHA(ea, eb, P, AL)                                                            \
    {                                                                                          \
        const float dx = ea.x - pxf, dy = ea.y - pyf;                                          \
        P = ea.z * dx * dx + eb.x * dy * dy + ea.w * dx * dy;             /* log2 units */     \
        AL = fminf(0.99f, eb.y * fast_exp2(P));                                                \
    }
    // blend phase: the serial part (T, done), in list order
#define GSR_U4_BLEND(eb, ec, P, AL, valid)                                                     \
    {                                                                                          \
        const bool ok = (valid) && !done && (P <= 0.f) && (AL >= (1.0f / 255.0f));             \
        const float test_T = T * (1.f - AL);                                                   \
        const bool stop = ok && (test_T < 0.0001f);                                            \
        const bool acc = ok && !stop;                                                          \
        const float w = acc ? AL * T : 0.f;                                                    \
        C0 += eb.z * w; C1 += eb.w * w; C2 += ec.x * w;                                        \
        D += ec.y * w; A += w;                                                                 \
        T = acc ? test_T : T;                                                                  \
        last = acc ? __float_as_uint(ec.z) : last;                                             \
        done = done || stop;                                                                   \
    }

    const uint32_t seg_slot0 = tile_seg[tile];
    // padding slots are read (never used): keep them finite so that 0 * garbage stays 0
    for (int q = lane; q < 3 * (GSR_RB + 4); q += 64) (&stage[wave][0][0])[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    wave_lds_handoff();
    // Two-deep fetch pipeline: while round r is composited, the records of round r+1 (whose
    // list entries were fetched during round r-1) and the list entries of round r+2 are in flight.
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc = ra;
    uint32_t id_next = 0;
    if (start + lane < end) {
        const float4* __restrict__ p = reinterpret_cast<const float4*>(recs + ids[start + lane]);
        ra = p[0]; rb = p[1]; rc = p[2];
    }
    if (start + GSR_RB + lane < end) id_next = ids[start + GSR_RB + lane];
    for (uint32_t base = start; base < end; base += GSR_RB) {
        if (__ballot(!done) == 0ull) break;
        float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na, nc = na;
        uint32_t id_next2 = 0;
        {
            const uint32_t i1 = base + GSR_RB + lane;
            if (i1 < end) {
                const float4* __restrict__ p = reinterpret_cast<const float4*>(recs + id_next);
                na = p[0]; nb = p[1]; nc = p[2];
            }
            const uint32_t i2 = base + 2 * GSR_RB + lane;
            if (i2 < end) id_next2 = ids[i2];
        }
        const uint32_t rel = base - start;
        if (rel != 0u && (rel & ((1u << seg_shift) - 1u)) == 0u) {    // segment cut: checkpoint for the backward
            float* c = ckpt + (size_t)(seg_slot0 + (rel >> seg_shift) - 1u) * GSR_CKPT_FLOATS + (wave * 64 + lane);
            c[0] = T; c[256] = C0; c[512] = C1; c[768] = C2; c[1024] = D; c[1280] = A;
        }
        const uint32_t i = base + lane;
        bool hit = false;
        if (i < end) {    // can alpha reach 1/255 anywhere in this wave's 8x8 block?
            hit = rect_max_power(ra.x, ra.y, ra.z, ra.w, rb.x, (float)bx, (float)(bx + 7), (float)by, (float)(by + 7))
                  >= min_visible_power(rb.y);
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
            // four entries per trip: the LDS reads and the four exponentials of a trip are independent of
            // the transmittance chain; slots n .. n+3 are zero padding (read, masked)
            float4 e0a = sa[0], e0b = sb[0], e0c = sc[0];
            for (int j = 0; j < n; j += 4) {
                const float4 e1a = sa[j + 1], e1b = sb[j + 1], e1c = sc[j + 1];
                const float4 e2a = sa[j + 2], e2b = sb[j + 2], e2c = sc[j + 2];
                const float4 e3a = sa[j + 3], e3b = sb[j + 3], e3c = sc[j + 3];
                float p0, p1, p2, p3, a0, a1, a2, a3;
                GSR_U4_ALPHA(e0a, e0b, p0, a0)
                GSR_U4_ALPHA(e1a, e1b, p1, a1)
                GSR_U4_ALPHA(e2a, e2b, p2, a2)
                GSR_U4_ALPHA(e3a, e3b, p3, a3)
                GSR_U4_BLEND(e0b, e0c, p0, a0, true)
                e0a = sa[j + 4]; e0b = sb[j + 4]; e0c = sc[j + 4];       // in flight during the other three blends
                GSR_U4_BLEND(e1b, e1c, p1, a1, j + 1 < n)
                GSR_U4_BLEND(e2b, e2c, p2, a2, j + 2 < n)
                GSR_U4_BLEND(e3b, e3c, p3, a3, j + 3 < n)
            }
            wave_lds_handoff();                           // reads above precede the next round's writes
        }
        ra = na; rb = nb; rc = nc; id_next = id_next2;
    }
#undef GSR_U4_ALPHA
#undef GSR_U4_BLEND
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


