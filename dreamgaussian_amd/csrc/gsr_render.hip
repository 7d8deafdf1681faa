// gsr_render.hip -- per-tile alpha compositing, forward (K5) and backward (K5b).
//
// Replaces (behaviour, not code) the render half of the external rasterizer reached through
// gs_renderer.py:800-809 and main.py:273. Spec: SURVEY.md Appendix A.5 / A.6.
//
// MI355X design. A 16x16 bin is one 256-thread workgroup = four INDEPENDENT waves, each owning
// an 8x8 pixel block (one pixel per lane). The bin's depth-sorted list holds Gaussian INDICES;
// a wave fetches 64 list entries per round with vector loads (one 64-byte SplatRec per lane,
// gathered from the per-Gaussian array), tests every record EXACTLY against its pixel block
// (ellipse alpha >= 1/255 vs rectangle, gsr_device.h), compacts the survivors into a private LDS
// stage (ballot + mbcnt) and runs the per-pixel loop over the survivors only, reading them back
// with wave-uniform (broadcast) ds_read_b128. No workgroup barrier in the loops.
//
// Forward AND backward are depth-segmented: one workgroup per (bin, segment of 2^seg_shift list
// entries). The forward composites every segment on its own (K5a) and chains them per pixel (K5b);
// the backward starts each segment from the absolute checkpoint K5b leaves. gsr_render_bwd_q2 walks FOUR
// quad lists per wave (each 16-lane DPP row owns a 4x4 pixel quad) and replaces the cross-lane
// reduction of the ten per-Gaussian sums by a transposition through LDS: pass 1 (lane = pixel)
// stores the two per-(Gaussian, pixel) scalars everything else derives from, pass 2
// (lane = (quad, Gaussian, half)) sums them over the pixels with plain FMAs.
#include "gsr_device.h"

namespace {
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// The exponent of a splat at a pixel (log2 units), with ONE fixed sequence of roundings: the two forward kernels and
// the backward must agree bit for bit on it (the backward re-derives which (pixel, Gaussian) pairs blended from
// `power <= 0 && alpha >= 1/255`), and hipcc's -ffp-contract=fast would otherwise fuse the sum differently per kernel.
// Five operations (round 5; six before: qa dx dx + (qc dy dy + (qb dx) dy)): dx (qa dx + qb dy) + (qc dy) dy -- one vector
// instruction less per (pixel, Gaussian) pair in every compositing kernel.
__device__ __forceinline__ float splat_power(float qa, float qb, float qc, float dx, float dy) {
    return fmaf(fmaf(qa, dx, __fmul_rn(qb, dy)), dx, __fmul_rn(__fmul_rn(qc, dy), dy));
}
}

#define GSR_RB 64   // list entries fetched per wave per round

// Two transcendentals of independent inputs in ADJACENT issue slots. Between plain VALU instructions a v_exp_f32 / v_rcp_f32 costs
// ~16 cycles of a SIMD's issue, behind another transcendental ~8 (tools/valu_rates.hip, profiles/r04_valu_rates.txt), and the
// scheduler spreads them out when left alone. The trailing s_nop covers the trans -> use wait state the hazard recognizer cannot
// see inside the asm.
#define GSR_TRANS_PAIR(op, d0, d1, s0, s1)                                                       \
        asm(op " %0, %2\n\t" op " %1, %3\n\ts_nop 0" : "=&v"(d0), "=&v"(d1) : "v"(s0), "v"(s1));

// agent-scope relaxed accesses (global_load / global_store ... sc1): the hint words other workgroups of the SAME launch
// read; they carry no ordering and need none (every value ever stored at such an address is a valid hint or a tag mismatch)
__device__ __forceinline__ uint32_t hint_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void hint_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long hint_load64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void hint_store64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One blended (pixel, Gaussian) pair of a segment composited on its own. T, C0, C1, C2, D, A, last, done are the lane's
// running state; `gate` = the transmittance in front of the segment (1 in gsr_render_fwd_seg; the pixel's running product
// in the exact walk of gsr_render_fwd_combine): the pixel stops at the first entry with gate * T' (1 - alpha) < 1e-4, which
// is not blended. Both kernels run the SAME sequence of roundings on (T', C', ...) -- the combine's walk
// reproduces the segment kernel's numbers bit for bit up to its own stopping point.
//   ea = x y qa qb | eb = qc opac r g | ec = b depth - - ; kpos = 1-based list position
#define GSR_COMPOSITE_G(ea, eb, ec, power, Gv, kpos, valid, gate, keepT, LASTV)                  \
    {                                                                                          \
        const float alpha = fminf(0.99f, eb.y * Gv);                                           \
        const bool ok = (valid) && !done && (power <= 0.f) && (alpha >= (1.0f / 255.0f));      \
        const float test_T = __fmul_rn(T, 1.f - alpha);                                        \
        const bool stop = ok && (__fmul_rn(gate, test_T) < 0.0001f);                           \
        const bool acc = ok && !stop;                                                          \
        const float w = acc ? alpha * T : 0.f;                                                 \
        C0 = fmaf(eb.z, w, C0); C1 = fmaf(eb.w, w, C1); C2 = fmaf(ec.x, w, C2);                \
        D = fmaf(ec.y, w, D); A += w;                                                          \
        T = (keepT ? acc : ok) ? test_T : T;                                                   \
        LASTV = acc ? (kpos) : LASTV;                                                          \
        done = done || stop;                                                                   \
    }
// Two consecutive list entries: both exponents first, their two v_exp_f32 in adjacent issue slots (GSR_TRANS_PAIR), then the two
// dependent updates in list order; `between` runs after the first (the callers request the next records there). The arithmetic
// per (pixel, entry) is the macro above in every kernel that blends.
#define GSR_COMPOSITE2(e0a, e0b, e0c, k0, v0, e1a, e1b, e1c, k1, v1, gate, keepT, LASTV, between) \
    {                                                                                          \
        const float pw0_ = splat_power(e0a.z, e0a.w, e0b.x, e0a.x - pxf, e0a.y - pyf);         /* log2 units */ \
        const float pw1_ = splat_power(e1a.z, e1a.w, e1b.x, e1a.x - pxf, e1a.y - pyf);         \
        float G0_, G1_;                                                                        \
        GSR_TRANS_PAIR("v_exp_f32", G0_, G1_, pw0_, pw1_)                                      \
        GSR_COMPOSITE_G(e0a, e0b, e0c, pw0_, G0_, k0, v0, gate, keepT, LASTV)                   \
        between                                                                                \
        GSR_COMPOSITE_G(e1a, e1b, e1c, pw1_, G1_, k1, v1, gate, keepT, LASTV)                   \
    }

// =========================================================================================
// K5a: forward, one workgroup per (tile, depth segment).
//
// A tile's depth-sorted list used to be one serial walk per 8x8 pixel block: at most 4 waves per tile, the heaviest
// tile set the kernel's length and a 5k-Gaussian scene kept half the SIMDs empty. Here every segment of 2^seg_shift
// entries is composited on its own workgroup with transmittance 1 coming in and leaves, per pixel, the segment's own
// (T', C', D', A', last); gsr_render_fwd_combine chains the segments per pixel (T = product of the T', C = sum of
// prefix * C') and re-walks, exactly, the ONE segment in which a pixel's running transmittance crosses the 1e-4 stop.
//
// Items run depth-major (gsr_tile_scan): all first segments, then all second segments, ... Most pixels of a dense
// scene stop after a third of their list; a segment behind that point is not composited: plane 7 of a record carries a
// per-pixel HINT, an upper bound of the transmittance after the segment (u16 fixed-point log2 | 16-bit launch tag), built
// from whichever predecessors have already published theirs (agent-scope relaxed loads; a segment that sees nothing
// assumes 1), and a per-(tile, wave) word names the segment from which all 64 pixels have stopped. Hints only ever
// OVER-estimate a transmittance. A wave whose 64 pixels have all stopped writes GSR_REC_SKIPPED instead of compositing.
// Nothing depends on a hint being seen or being right: a pixel that reaches a skipped record alive (stale words of an
// earlier launch with the same 16-bit tag) has that segment walked by the combine kernel -- same result, bit for bit.
//
// QUAD = false: one list per 8x8 block (lane = pixel, row-major). QUAD = true: every 16-lane row owns a 4x4 quad with
// its own list of staged slots (exact ellipse-vs-quad tests, quad_max_powers) -- fewer masked lanes when splats are small
// against the block. Same arithmetic per (pixel, Gaussian), every rounding pinned: bit-identical records.
// =========================================================================================
template <bool QUAD>
__global__ void __launch_bounds__(256)
gsr_render_fwd_seg(const uint4* __restrict__ items, const uint32_t* __restrict__ level_off,
                   const uint32_t* __restrict__ tile_off, const SplatRec* __restrict__ recs_all,
                   const uint32_t* __restrict__ ids, int W, int H, int gx,
                   float* __restrict__ rec_base, int seg_shift, unsigned long long* __restrict__ sat /* [tiles][4] */, uint32_t epoch,
                   int hint_mode /* 0 = on; 1 = off (no hint is read: every segment is composited); 2 = TEST: every
                                    segment behind a tile's first is skipped, the combine kernel walks them all */,
                   const unsigned long long* __restrict__ counters, uint32_t capacity, uint32_t maxc_cap, ViewSplit vs) {
    // lists that do not fit the scratch, or longer than the sort kernels that were launched cover: the host repeats the tail
    if (counters[2] > (unsigned long long)capacity || counters[3] > (unsigned long long)maxc_cap) return;
    __shared__ float4 stage[4][3][GSR_RB + 2];             // [wave][field group][slot (+2 pad)]
    __shared__ __attribute__((aligned(8))) uint8_t qlist[QUAD ? 4 : 1][4][80];   // QUAD: [wave][quad][k] = staged slot of the quad's k-th entry
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int row = lane >> 4, l15 = lane & 15;
    const int lx = QUAD ? (row & 1) * 4 + (l15 & 3) : (lane & 7), ly = QUAD ? (row >> 1) * 4 + (l15 >> 2) : (lane >> 3);
    const int cidx = wave * 64 + ly * 8 + lx;             // record slot of this lane's pixel (row-major 8x8 per wave)
    float4* __restrict__ sa = stage[wave][0];
    float4* __restrict__ sb = stage[wave][1];
    float4* __restrict__ sc = stage[wave][2];
    bool lds_ready = false;
    // One item per workgroup by default (the dispatcher then hands the items out in list order as slots free up, which is what
    // keeps the hints flowing; a fixed stride per workgroup -- test hook "fwd_grid" -- lets the fast ones run ahead of the hints: 2x slower).
    const uint32_t total = level_off[GSR_NLEV];
  for (uint32_t k = blockIdx.x; k < total; k += gridDim.x) {
    const uint4 item = items[k];                          // {tile, list start, segment record, segment << 8 | entries - 1}
    const int tg = (int)item.x;                           // tile among all views' tiles
    const uint32_t start = item.y, c = item.w >> 8;
    const int view = tg / vs.tiles_per_view;
    if (!((vs.view_mask >> view) & 1u)) continue;         // this view composites with the other instantiation
    float* __restrict__ rec0 = rec_base + ((size_t)item.z - c) * GSR_CKPT_FLOATS + cidx;
    // Two thirds of the items of a dense scene lie behind the depth at which their whole tile has stopped: those leave here,
    // after one load (the tile's four per-wave hint words, one per lane group) and the store of the skip marker.
    unsigned long long sw_own = 0ull;
    if (c > 0u && hint_mode == 0) {
        const unsigned long long sw = hint_load64(sat + (size_t)tg * 4 + (lane & 3));
        const bool hit = (uint32_t)(sw >> 32) == epoch && (uint32_t)sw <= c;
        if (__ballot(hit) == ~0ull && c != (uint32_t)(GSR_NLEV - 1)) { rec0[(size_t)c * GSR_CKPT_FLOATS] = GSR_REC_SKIPPED; continue; }
        sw_own = __shfl(sw, wave, 64);
    }
    const int tile = tg - view * vs.tiles_per_view;
    const SplatRec* __restrict__ recs = recs_all + (size_t)view * vs.N;
    const int bx = (tile % gx) * GSR_TILE + (wave & 1) * 8;
    const int by = (tile / gx) * GSR_TILE + (wave >> 1) * 8;
    if (bx >= W || by >= H) continue;                     // whole block outside the image (the combine skips it too)
    const int px = bx + lx, py = by + ly;
    const bool inside = (px < W) && (py < H);
    const float pxf = (float)px, pyf = (float)py;
    // the last level walks the rest of its tile's list (tiles with more than GSR_NLEV segments)
    uint32_t n = (c << seg_shift) + (item.w & 255u) + 1u, c_end = c + 1u;
    if (c == (uint32_t)(GSR_NLEV - 1)) { n = tile_off[tg + 1] - start; c_end = (n + (1u << seg_shift) - 1u) >> seg_shift; }
    unsigned long long* __restrict__ satw = sat + (size_t)tg * 4 + wave;
    const uint32_t tag = epoch << 16;
    const float bx0 = (float)bx, by0 = (float)by;

    for (uint32_t s = c; s < c_end; ++s) {
        float* __restrict__ rec = rec0 + (size_t)s * GSR_CKPT_FLOATS;
        const uint32_t lo = s << seg_shift, hi = min(lo + (1u << seg_shift), n);
        uint32_t qbest = 0;
        if (s > 0 && hint_mode == 2) { rec[0] = GSR_REC_SKIPPED; continue; }
        if (s > 0 && hint_mode == 0) {
            const unsigned long long sw = s == c ? sw_own : hint_load64(satw);
            if ((uint32_t)(sw >> 32) == epoch && (uint32_t)sw <= s) { rec[0] = GSR_REC_SKIPPED; continue; }   // this block stopped earlier
        }
        // the list entries of the first round travel while the hints are read
        uint32_t id0 = 0;
        if (lo + lane < hi) id0 = ids[start + lo + lane];
        if (s > 0 && hint_mode == 0) {
            const uint32_t* hp = reinterpret_cast<const uint32_t*>(rec) + GSR_REC_HINT;
            const uint32_t h1 = hint_load(hp - GSR_CKPT_FLOATS);
            const uint32_t h2 = s >= 2 ? hint_load(hp - 2 * GSR_CKPT_FLOATS) : 0u;
            const uint32_t h3 = s >= 3 ? hint_load(hp - 3 * GSR_CKPT_FLOATS) : 0u;
            qbest = (h1 & 0xffff0000u) == tag ? (h1 & 0xffffu) : 0u;
            qbest = max(qbest, (h2 & 0xffff0000u) == tag ? (h2 & 0xffffu) : 0u);
            qbest = max(qbest, (h3 & 0xffff0000u) == tag ? (h3 & 0xffffu) : 0u);
            if (!inside) qbest = 0xffffu;
            if (__ballot(qbest < GSR_QSAT) == 0ull) {     // every pixel of the block has stopped in front of this segment
                rec[0] = GSR_REC_SKIPPED;
                hint_store(reinterpret_cast<uint32_t*>(rec) + GSR_REC_HINT, tag | qbest);
                if (lane == 0) hint_store64(satw, ((unsigned long long)epoch << 32) | s);
                continue;
            }
        }
        float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f;
        uint32_t last = 0;
        bool done = !inside;
        float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na, nc = na;
        if (lo + lane < hi) {
            const float4* __restrict__ p = reinterpret_cast<const float4*>(recs + id0);
            na = p[0]; nb = p[1]; nc = p[2];
        }
        if (!lds_ready) {   // padding / stale slots are read (never used): keep them finite so that 0 * garbage stays 0
            for (int q = lane; q < 3 * (GSR_RB + 2); q += 64) (&stage[wave][0][0])[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (QUAD) for (int q = lane; q < 4 * 80 / 4; q += 64) reinterpret_cast<uint32_t*>(&qlist[QUAD ? wave : 0][0][0])[q] = 0u;
            wave_lds_handoff();
            lds_ready = true;
        }
        for (uint32_t pos0 = lo; pos0 < hi; pos0 += GSR_RB) {          // 0-based positions pos0 .. pos0+63
            float4 ra = na, rb = nb, rc = nc;
            const uint32_t i = pos0 + lane;
            if (i + GSR_RB < hi) {                                     // next round's records: in flight during this round
                const float4* __restrict__ p = reinterpret_cast<const float4*>(recs + ids[start + i + GSR_RB]);
                na = p[0]; nb = p[1]; nc = p[2];
            }
            const unsigned long long alive = __ballot(!done);
            if (alive == 0ull) break;
            if (!QUAD) {
                bool hit = false;
                if (i < hi)      // can alpha reach 1/255 anywhere in this wave's 8x8 block?
                    hit = rect_max_power(ra.x, ra.y, ra.z, ra.w, rb.x, bx0, bx0 + 7.f, by0, by0 + 7.f) >= min_visible_power(rb.y);
                const unsigned long long mask = __ballot(hit);
                if (mask == 0ull) continue;
                const int nhit = __popcll(mask);
                if (hit) {
                    const uint32_t pos = lanes_below(mask);
                    rc.z = __uint_as_float(i + 1u);       // 1-based list position replaces the box
                    sa[pos] = ra; sb[pos] = rb; sc[pos] = rc;
                }
                wave_lds_handoff();
                float4 e0a = sa[0], e0b = sb[0], e0c = sc[0];
                // two entries per trip, both unconditional (the second is masked off on an odd tail) so
                // that the body stays one basic block and the LDS reads are issued ahead of their use
                for (int j = 0; j < nhit; j += 2) {       // slots nhit, nhit+1 are padding: read, masked
                    const float4 e1a = sa[j + 1], e1b = sb[j + 1], e1c = sc[j + 1];
                    const uint32_t k0 = __float_as_uint(e0c.z);
                    GSR_COMPOSITE2(e0a, e0b, e0c, k0, true, e1a, e1b, e1c, __float_as_uint(e1c.z), j + 1 < nhit, 1.f, false, last,
                                   e0a = sa[j + 2]; e0b = sb[j + 2]; e0c = sc[j + 2];)   // in flight during entry j+1
                }
                wave_lds_handoff();                       // reads above precede the next round's writes
            } else {
                bool h0 = false, h1 = false, h2 = false, h3 = false;
                if (i < hi) {
                    const float thr = min_visible_power(rb.y);
                    float qp[4];
                    quad_max_powers(ra.x, ra.y, ra.z, ra.w, rb.x, bx0, by0, qp);
                    // a quad whose pixels have all stopped takes no more entries
                    h0 = ((alive & 0x000000000000ffffull) != 0ull) && qp[0] >= thr;
                    h1 = ((alive & 0x00000000ffff0000ull) != 0ull) && qp[1] >= thr;
                    h2 = ((alive & 0x0000ffff00000000ull) != 0ull) && qp[2] >= thr;
                    h3 = ((alive & 0xffff000000000000ull) != 0ull) && qp[3] >= thr;
                }
                const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
                if ((m0 | m1 | m2 | m3) == 0ull) continue;
                uint8_t (*qlw)[80] = qlist[QUAD ? wave : 0];
                if (h0 | h1 | h2 | h3) { sa[lane] = ra; sb[lane] = rb; sc[lane] = rc; }
                if (h0) qlw[0][lanes_below(m0)] = (uint8_t)lane;
                if (h1) qlw[1][lanes_below(m1)] = (uint8_t)lane;
                if (h2) qlw[2][lanes_below(m2)] = (uint8_t)lane;
                if (h3) qlw[3][lanes_below(m3)] = (uint8_t)lane;
                const int n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), n3 = __popcll(m3);
                const int nmax = max(max(n0, n1), max(n2, n3));
                const uint32_t pos1 = pos0 + 1u;          // 1-based list position of staged slot 0
                const uint8_t* __restrict__ ql = qlw[row];
                // a list shorter than the longest is padded with slot 64 (an all-zero record: opacity 0 never passes the alpha test) up to the
                // last batch that is read, so that an entry needs no 'inside my list' test; the last blended entry is tracked as a staged
                // slot and turned into a list position once per round (one compare and one add per entry fewer on the longest tiles' path)
                {
                    const int nread = (nmax + 7) & ~7;
                    if (lane >= n0 && lane < nread) qlw[0][lane] = (uint8_t)GSR_RB;
                    if (lane >= n1 && lane < nread) qlw[1][lane] = (uint8_t)GSR_RB;
                    if (lane >= n2 && lane < nread) qlw[2][lane] = (uint8_t)GSR_RB;
                    if (lane >= n3 && lane < nread) qlw[3][lane] = (uint8_t)GSR_RB;
                }
                uint32_t lasts = 0xffffffffu;
                wave_lds_handoff();
                for (int jb = 0; jb < nmax; jb += 8) {
                    const uint2 sl = *reinterpret_cast<const uint2*>(ql + jb);
                    uint32_t slot[8];
#pragma unroll
                    for (int b = 0; b < 8; ++b) slot[b] = ((b < 4 ? sl.x : sl.y) >> (8 * (b & 3))) & 0xffu;
                    // two entries per trip in two fixed register sets; the second is masked off past the row's own
                    // list (nmine <= nmax; stale slots read staged records)
                    float4 e0a = sa[slot[0]], e0b = sb[slot[0]], e0c = sc[slot[0]];
#pragma unroll
                    for (int b = 0; b < 8; b += 2) {
                        if (jb + b < nmax) {              // wave-uniform
                            const float4 e1a = sa[slot[b + 1]], e1b = sb[slot[b + 1]], e1c = sc[slot[b + 1]];
                            GSR_COMPOSITE2(e0a, e0b, e0c, slot[b], true, e1a, e1b, e1c, slot[b + 1], true, 1.f, false, lasts,
                                           if (b + 2 < 8) { e0a = sa[slot[b + 2]]; e0b = sb[slot[b + 2]]; e0c = sc[slot[b + 2]]; })   // in flight during entry b+1
                        }
                    }
                }
                last = lasts != 0xffffffffu ? pos1 + lasts : last;
                wave_lds_handoff();                       // reads above precede the next round's writes
            }
        }
        // ---- the segment's record
        rec[0] = T; rec[256] = C0; rec[512] = C1; rec[768] = C2; rec[1024] = D; rec[1280] = A;
        reinterpret_cast<uint32_t*>(rec)[GSR_REC_LAST] = last;
        // ---- its hint: the best bound of the transmittance in front (as seen now) times this segment's T', rounded UP
        // (v_log_f32 is good to 1 ulp: the 0.02 steps of 1/256 cover it many times over)
        if (hint_mode != 0) continue;
        if (s > 0) {
            const uint32_t h1 = hint_load(reinterpret_cast<const uint32_t*>(rec) + GSR_REC_HINT - GSR_CKPT_FLOATS);
            qbest = max(qbest, (h1 & 0xffff0000u) == tag ? (h1 & 0xffffu) : 0u);
        }
        uint32_t q = 0xffffu;
        if (inside && !done) {
            const float dq = floorf(fmaf(-256.f, __builtin_amdgcn_logf(T), -0.02f));          // T in (0, 1]: dq >= -0.02
            q = min(qbest + (uint32_t)fmaxf(dq, 0.f), 0xffffu);
        }
        hint_store(reinterpret_cast<uint32_t*>(rec) + GSR_REC_HINT, tag | q);
        if (__ballot(q < GSR_QSAT) == 0ull && lane == 0) hint_store64(satw, ((unsigned long long)epoch << 32) | (s + 1u));
    }
  }
}
template __global__ void gsr_render_fwd_seg<false>(const uint4*, const uint32_t*, const uint32_t*, const SplatRec*, const uint32_t*, int, int, int,
                                                   float*, int, unsigned long long*, uint32_t, int, const unsigned long long*, uint32_t, uint32_t, ViewSplit);
template __global__ void gsr_render_fwd_seg<true>(const uint4*, const uint32_t*, const uint32_t*, const SplatRec*, const uint32_t*, int, int, int,
                                                  float*, int, unsigned long long*, uint32_t, int, const unsigned long long*, uint32_t, uint32_t, ViewSplit);

// The backward's per-Gaussian accumulators ([views][N][12] floats inside `geom`) start from zero. The forward's per-tile kernel
// clears them: workgroup k of the launch stores zeros to slice k of `zero_n` float4s BEHIND its own work -- the compositing leaves
// HBM idle, the workgroups finish spread over the kernel's length, and the 10 us fill in front of gsr_render_bwd_q2 (and its launch)
// is gone. zero_n = 0: nothing to clear (GSR_VIEW_NO_BACKWARD, or the other instantiation's launch does it).
// An ASYNCHRONOUS forward (GSR_VIEW_ASYNC_STATS) whose lists turned out not to fit what it was launched for: nobody repeats the tail
// (the host has returned long ago), so the per-tile kernel leaves NaN images instead of unwritten memory -- workgroup k fills tile k
// of the launch (any bijection will do: `order` may be stale) -- and the host reports -6 at the thread's next call.
__device__ __forceinline__ void poison_tile(const ViewSplit& vs, int W, int H, int gx, float* __restrict__ out_color,
                                            float* __restrict__ out_depth, float* __restrict__ out_alpha) {
    if (threadIdx.x >= 256u) return;
    const int tg = (int)blockIdx.x, view = tg / vs.tiles_per_view, tile = tg - view * vs.tiles_per_view;
    const int px = (tile % gx) * GSR_TILE + (int)(threadIdx.x & 15u), py = (tile / gx) * GSR_TILE + (int)(threadIdx.x >> 4);
    if (px >= W || py >= H) return;
    const size_t HW = (size_t)W * H, pix = (size_t)py * W + px;
    const float nan = __uint_as_float(0x7fc00000u);
    out_color[view * 3 * HW + pix] = nan; out_color[view * 3 * HW + HW + pix] = nan; out_color[view * 3 * HW + 2 * HW + pix] = nan;
    out_depth[view * HW + pix] = nan; out_alpha[view * HW + pix] = nan;
}
template <uint32_t THREADS = 256u>      // (the workgroup's size: a constant, not blockDim -- no implicit-argument load in front of the stores)
__device__ __forceinline__ void clear_slice(float4* __restrict__ zero4, uint32_t zero_n, uint32_t per /* float4s per workgroup (host: ceil(zero_n / grid)) */) {
    if (zero_n == 0u) return;
    const uint32_t lo = blockIdx.x * per, hi = min(lo + per, zero_n);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += THREADS) zero4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
// =========================================================================================
// K5s: forward, the serial walk -- for views that fill the chip on their own (fwd_sequential_for in gsr_api.hip).
// One workgroup per tile (heaviest first), four independent waves = four 8x8 blocks. A wave fetches 64 list entries per
// round, three rounds ahead (the gathers miss the XCD's L2 half of the time), tests each record exactly against its
// block (QUAD: against each of its four 4x4 quads, one list of staged slots per 16-lane row), stages the survivors and
// composites them; the per-pixel stop is exact where it happens and a wave leaves when its 64 pixels have stopped: every
// vector instruction of the compositing is needed exactly once. At every segment cut the wave leaves the absolute
// checkpoint the backward's segment starts from (the record K5b writes in the segmented mode); the tile's entries of the
// backward's work list are reserved at the end as in K5b.
// =========================================================================================
template <bool QUAD>
__global__ void __launch_bounds__(256)
gsr_render_fwd_serial(const uint32_t* __restrict__ tile_off, const SplatRec* __restrict__ recs,
                      const uint32_t* __restrict__ ids, int W, int H, int gx,
                      float* __restrict__ out_color, float* __restrict__ out_depth,
                      float* __restrict__ out_alpha, float* __restrict__ final_T,
                      uint32_t* __restrict__ n_contrib, float* __restrict__ totals /*[5][H*W]*/,
                      float* __restrict__ rec_base, const uint32_t* __restrict__ tile_seg,
                      const uint32_t* __restrict__ order, int seg_shift,
                      uint32_t* __restrict__ plan_off, uint4* __restrict__ plan_items,
                      unsigned long long* __restrict__ plan_total, uint32_t plan_cap, unsigned long long qmask_views, uint32_t sink_rec,
                      const unsigned long long* __restrict__ counters, uint32_t capacity, uint32_t maxc_cap, ViewSplit vs,
                      float4* __restrict__ zero4, uint32_t zero_n, uint32_t zero_per, ZeroSide zs) {
    if (counters[2] > (unsigned long long)capacity || counters[3] > (unsigned long long)maxc_cap) {
        if ((qmask_views >> 62) & 1ull) poison_tile(vs, W, H, gx, out_color, out_depth, out_alpha);     // (bit 62: an asynchronous forward)
        return;
    }
    __shared__ float4 stage[4][3][GSR_RB + 2];
    __shared__ __attribute__((aligned(8))) uint8_t qlist[QUAD ? 4 : 1][4][80];
    __shared__ uint32_t wl[4];
    __shared__ uint32_t plan_base;
    // bit 63 of qmask_views: no backward follows this forward (GSR_VIEW_NO_BACKWARD) -- no checkpoints, no quad masks (118 MB of
    // stores at 1M Gaussians), and gsr_backward refuses the state
    const bool keep_state = (qmask_views >> 63) == 0ull;
    if (blockIdx.x == 0 && threadIdx.x == 0) plan_total[GSR_CNT_QMASK - GSR_CNT_PLAN] = qmask_views & ~(1ull << 62);   // the views whose records carry quad masks (| bit 63)
    const int tg = (int)order[blockIdx.x];                // heaviest tiles first
    const int view = tg / vs.tiles_per_view;
    if (!((vs.view_mask >> view) & 1u)) { clear_slice(zero4, zero_n, zero_per); clear_side(zs); return; }   // (workgroup-uniform) this view composites with the other instantiation
    const int tile = tg - view * vs.tiles_per_view;
    const float* __restrict__ bg = vs.bg[view];
    {
        const size_t HWv = (size_t)W * H;
        recs += (size_t)view * vs.N;
        out_color += view * 3 * HWv; out_depth += view * HWv; out_alpha += view * HWv;
        final_T += view * vs.img_stride; n_contrib += view * vs.img_stride; totals += view * vs.img_stride;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int row = lane >> 4, l15 = lane & 15;
    const int bx = (tile % gx) * GSR_TILE + (wave & 1) * 8;
    const int by = (tile / gx) * GSR_TILE + (wave >> 1) * 8;
    const int lx = QUAD ? (row & 1) * 4 + (l15 & 3) : (lane & 7), ly = QUAD ? (row >> 1) * 4 + (l15 >> 2) : (lane >> 3);
    const int px = bx + lx, py = by + ly;
    const bool inside = (px < W) && (py < H);             // false for every lane of a block outside the image
    const float pxf = (float)px, pyf = (float)py;
    const float bx0 = (float)bx, by0 = (float)by;
    const uint32_t start = tile_off[tg];
    const uint32_t tile_n = tile_off[tg + 1] - start;
    const uint32_t n = (bx < W && by < H) ? tile_n : 0u;   // a block outside the image walks nothing
    float* __restrict__ recw = rec_base + (size_t)(keep_state ? tile_seg[tg] : 0u) * GSR_CKPT_FLOATS;     // (wave-uniform) the tile's first segment record
    float* __restrict__ rec0 = recw + (wave * 64 + ly * 8 + lx);
    float4* __restrict__ sa = stage[wave][0];
    float4* __restrict__ sb = stage[wave][1];
    float4* __restrict__ sc = stage[wave][2];
    for (int q = lane; q < 3 * (GSR_RB + 2); q += 64) (&stage[wave][0][0])[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (QUAD) for (int q = lane; q < 4 * 80 / 4; q += 64) reinterpret_cast<uint32_t*>(&qlist[QUAD ? wave : 0][0][0])[q] = 0u;
    wave_lds_handoff();

    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f;
    uint32_t last = 0;
    bool done = !inside;
    // The walk is a chain of dependent gathers (list entry -> record) in front of ~1 000 instructions per round, and in its last
    // third the heaviest tiles run alone on their SIMDs: what a round costs then is what it WAITS for. Three register sets take
    // turns (the loop is unrolled three times, no copies between them): round r composites set r % 3, requested two rounds ago,
    // and requests round r + 2's records and round r + 3's list entries. Every load is branch-free (positions clamped into the
    // list: a load inside a divergent `if` drags a register copy, i.e. a wait, right behind it) and so are the stores (a lane
    // that has nothing to store writes into a spare record): the compiler's waits are static counts, the minimum over all
    // paths, and with the same number of memory operations on every path they leave two rounds of requests outstanding.
    float* const sinkf = rec_base + (size_t)sink_rec * GSR_CKPT_FLOATS + (wave * 64 + lane);
    unsigned long long* const sink64 = reinterpret_cast<unsigned long long*>(rec_base + (size_t)sink_rec * GSR_CKPT_FLOATS + GSR_REC_HINT) + lane;
    if (n > 0u) {                                         // (wave-uniform)
        float4 r0a, r0b, r0c, r1a, r1b, r1c, r2a, r2b, r2c;
        uint32_t idq;                                     // list entries of the round after the youngest requested records
        {
            const uint32_t id0 = ids[start + min((uint32_t)lane, n - 1u)], id1 = ids[start + min(GSR_RB + (uint32_t)lane, n - 1u)];
            idq = ids[start + min(2u * GSR_RB + (uint32_t)lane, n - 1u)];
            const float4* __restrict__ p0 = reinterpret_cast<const float4*>(recs + id0);
            const float4* __restrict__ p1 = reinterpret_cast<const float4*>(recs + id1);
            r0a = p0[0]; r0b = p0[1]; r0c = p0[2];
            r1a = p1[0]; r1b = p1[1]; r1c = p1[2];
            // (the static counts again: the path from here into the loop must hold as many memory operations behind these
            // requests as one trip around the loop does, or the first wait of every round shrinks to what THIS path allows)
#pragma unroll
            for (int d = 1; d <= 16; ++d) sinkf[256 * d] = 0.f;          // planes 1.. of the three spare records
        }
        auto walk_round = [&](const uint32_t rel, float4 ra, float4 rb, float4 rc, float4& da, float4& db, float4& dc) -> bool {
            if (rel >= n) return false;
            const unsigned long long alive = __ballot(!done);
            if (alive == 0ull) return false;
            {
                const float4* __restrict__ p = reinterpret_cast<const float4*>(recs + idq);
                da = p[0]; db = p[1]; dc = p[2];
                idq = ids[start + min(rel + 3u * GSR_RB + (uint32_t)lane, n - 1u)];
            }
            {   // segment cut: checkpoint for the backward
                const bool cut = keep_state && rel != 0u && (rel & ((1u << seg_shift) - 1u)) == 0u;
                float* c = (cut && inside) ? rec0 + (size_t)((rel >> seg_shift) - 1u) * GSR_CKPT_FLOATS : sinkf;
                c[0] = T; c[256] = C0; c[512] = C1; c[768] = C2; c[1024] = D; c[1280] = A;
            }
            const uint32_t i = rel + lane;
            if (!QUAD) {
                bool hit = false;
                if (i < n)      // can alpha reach 1/255 anywhere in this wave's 8x8 block?
                    hit = rect_max_power(ra.x, ra.y, ra.z, ra.w, rb.x, bx0, bx0 + 7.f, by0, by0 + 7.f) >= min_visible_power(rb.y);
                const unsigned long long mask = __ballot(hit);
                if (mask != 0ull) {
                    const int nhit = __popcll(mask);
                    if (hit) {
                        const uint32_t pos = lanes_below(mask);
                        rc.z = __uint_as_float(i + 1u);       // 1-based list position replaces the box
                        sa[pos] = ra; sb[pos] = rb; sc[pos] = rc;
                    }
                    wave_lds_handoff();
                    float4 e0a = sa[0], e0b = sb[0], e0c = sc[0];
                    for (int j = 0; j < nhit; j += 2) {       // slots nhit, nhit+1 are padding: read, masked
                        const float4 e1a = sa[j + 1], e1b = sb[j + 1], e1c = sc[j + 1];
                        const uint32_t k0 = __float_as_uint(e0c.z);
                        GSR_COMPOSITE2(e0a, e0b, e0c, k0, true, e1a, e1b, e1c, __float_as_uint(e1c.z), j + 1 < nhit, 1.f, true, last,
                                       e0a = sa[j + 2]; e0b = sb[j + 2]; e0c = sc[j + 2];)   // in flight during entry j+1
                    }
                    wave_lds_handoff();                       // reads above precede the next round's writes
                }
            } else {
                bool h0 = false, h1 = false, h2 = false, h3 = false;
                if (i < n) {
                    const float thr = min_visible_power(rb.y);
                    float qp[4];
                    quad_max_powers(ra.x, ra.y, ra.z, ra.w, rb.x, bx0, by0, qp);
                    // a quad whose pixels have all stopped takes no more entries
                    h0 = ((alive & 0x000000000000ffffull) != 0ull) && qp[0] >= thr;
                    h1 = ((alive & 0x00000000ffff0000ull) != 0ull) && qp[1] >= thr;
                    h2 = ((alive & 0x0000ffff00000000ull) != 0ull) && qp[2] >= thr;
                    h3 = ((alive & 0xffff000000000000ull) != 0ull) && qp[3] >= thr;
                }
                const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
                {   // the round's four hit masks: the backward's quad tests (GSR_CNT_QMASK). Lanes 0..3 store one each, the others into the sink
                    unsigned long long* mp = reinterpret_cast<unsigned long long*>(recw + (size_t)(rel >> seg_shift) * GSR_CKPT_FLOATS + GSR_REC_HINT)
                                             + (((rel >> 6) & ((1u << (seg_shift - 6)) - 1u)) * 16u + (uint32_t)wave * 4u);
                    *((lane < 4 && keep_state) ? mp + lane : sink64) = lane == 0 ? m0 : (lane == 1 ? m1 : (lane == 2 ? m2 : m3));
                }
                if ((m0 | m1 | m2 | m3) != 0ull) {
                    uint8_t (*qlw)[80] = qlist[QUAD ? wave : 0];
                    if (h0 | h1 | h2 | h3) { sa[lane] = ra; sb[lane] = rb; sc[lane] = rc; }
                    if (h0) qlw[0][lanes_below(m0)] = (uint8_t)lane;
                    if (h1) qlw[1][lanes_below(m1)] = (uint8_t)lane;
                    if (h2) qlw[2][lanes_below(m2)] = (uint8_t)lane;
                    if (h3) qlw[3][lanes_below(m3)] = (uint8_t)lane;
                    const int n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), n3 = __popcll(m3);
                    const int nmax = max(max(n0, n1), max(n2, n3));
                    const uint32_t pos1 = rel + 1u;           // 1-based list position of staged slot 0
                    const uint8_t* __restrict__ ql = qlw[row];
                    // a list shorter than the longest is padded with slot 64 (an all-zero record: opacity 0 never passes the alpha test) up to the
                    // last batch that is read, so that an entry needs no 'inside my list' test; the last blended entry is tracked as a staged
                    // slot and turned into a list position once per round (one compare and one add per entry fewer on the longest tiles' path)
                    {
                        const int nread = (nmax + 7) & ~7;
                        if (lane >= n0 && lane < nread) qlw[0][lane] = (uint8_t)GSR_RB;
                        if (lane >= n1 && lane < nread) qlw[1][lane] = (uint8_t)GSR_RB;
                        if (lane >= n2 && lane < nread) qlw[2][lane] = (uint8_t)GSR_RB;
                        if (lane >= n3 && lane < nread) qlw[3][lane] = (uint8_t)GSR_RB;
                    }
                    uint32_t lasts = 0xffffffffu;
                    wave_lds_handoff();
                    for (int jb = 0; jb < nmax; jb += 8) {
                        const uint2 sl = *reinterpret_cast<const uint2*>(ql + jb);
                        uint32_t slot[8];
#pragma unroll
                        for (int b = 0; b < 8; ++b) slot[b] = ((b < 4 ? sl.x : sl.y) >> (8 * (b & 3))) & 0xffu;
                        float4 e0a = sa[slot[0]], e0b = sb[slot[0]], e0c = sc[slot[0]];
#pragma unroll
                        for (int b = 0; b < 8; b += 2) {
                            if (jb + b < nmax) {              // wave-uniform
                                const float4 e1a = sa[slot[b + 1]], e1b = sb[slot[b + 1]], e1c = sc[slot[b + 1]];
                                GSR_COMPOSITE2(e0a, e0b, e0c, slot[b], true, e1a, e1b, e1c, slot[b + 1], true, 1.f, true, lasts,
                                               if (b + 2 < 8) { e0a = sa[slot[b + 2]]; e0b = sb[slot[b + 2]]; e0c = sc[slot[b + 2]]; })   // in flight during entry b+1
                            }
                        }
                    }
                    last = lasts != 0xffffffffu ? pos1 + lasts : last;
                    wave_lds_handoff();                       // reads above precede the next round's writes
                }
            }
            return true;
        };
        for (uint32_t rel = 0;; rel += 3u * GSR_RB) {
            if (!walk_round(rel, r0a, r0b, r0c, r2a, r2b, r2c)) break;
            if (!walk_round(rel + GSR_RB, r1a, r1b, r1c, r0a, r0b, r0c)) break;
            if (!walk_round(rel + 2u * GSR_RB, r2a, r2b, r2c, r1a, r1b, r1c)) break;
        }
    }
    if (inside) {
        const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = fmaf(T, bg[0], C0);
        out_color[HW + pix] = fmaf(T, bg[1], C1);
        out_color[2 * HW + pix] = fmaf(T, bg[2], C2);
        out_depth[pix] = D;
        out_alpha[pix] = A;
        totals[pix] = C0; totals[HW + pix] = C1; totals[2 * HW + pix] = C2;   // sums without background
        totals[3 * HW + pix] = D; totals[4 * HW + pix] = A;
    }
    // ---- how deep the backward has to walk this tile's list, and its (tile, segment) work items
    {
        const uint32_t wmax = wave_max_u32(inside ? last : 0u);
        if (lane == 0) wl[wave] = wmax;
    }
    lds_barrier();
    if (threadIdx.x == 0) {
        const uint32_t tl = max(max(wl[0], wl[1]), max(wl[2], wl[3]));
        const uint32_t segs = (tl + (1u << seg_shift) - 1u) >> seg_shift;
        const uint32_t base = segs ? (uint32_t)atomicAdd(plan_total, (unsigned long long)segs) : 0u;
        plan_off[tg] = base;
        plan_base = base;
        wl[0] = segs;
    }
    lds_barrier();
    {
        const uint32_t segs = wl[0], base = plan_base;
        for (uint32_t q = threadIdx.x; q < segs; q += 256)
            if (base + q < plan_cap)     // {tile, the segment's record, list start, segment | (entries in it - 1) << 24}: all the backward needs, in one load
                plan_items[base + q] = make_uint4((uint32_t)tg, tile_seg[tg] + q, start, q | ((min(1u << seg_shift, tile_n - (q << seg_shift)) - 1u) << 24));
    }
    clear_slice(zero4, zero_n, zero_per);
    clear_side(zs);
}
template __global__ void gsr_render_fwd_serial<false>(const uint32_t*, const SplatRec*, const uint32_t*, int, int, int, float*, float*, float*, float*,
                                                      uint32_t*, float*, float*, const uint32_t*, const uint32_t*, int, uint32_t*, uint4*,
                                                      unsigned long long*, uint32_t, unsigned long long, uint32_t, const unsigned long long*, uint32_t, uint32_t, ViewSplit, float4*, uint32_t, uint32_t, ZeroSide);
template __global__ void gsr_render_fwd_serial<true>(const uint32_t*, const SplatRec*, const uint32_t*, int, int, int, float*, float*, float*, float*,
                                                     uint32_t*, float*, float*, const uint32_t*, const uint32_t*, int, uint32_t*, uint4*,
                                                     unsigned long long*, uint32_t, unsigned long long, uint32_t, const unsigned long long*, uint32_t, uint32_t, ViewSplit, float4*, uint32_t, uint32_t, ZeroSide);

// =========================================================================================
// K5p: the serial walk with TWO waves per 8x8 block -- the host's choice for ONE view of 1 024 .. 2 047 tiles (finish_impl in
// gsr_api.hip; test hook "fwd_mode" = 3 forces it). Written at the end of round 4, first run in round 5: bit-identical to
// gsr_render_fwd_serial<true> in every case tried, 15 % faster at 250k Gaussians / 512^2 (354 busy tiles), nothing at 800^2.
//
// gsr_render_fwd_serial is latency-bound per wave, and a dense single view offers few waves: 777 busy tiles x 4 at 1M Gaussians /
// 800^2 = 3 per SIMD (two per SIMD: +34 %, profiles/r04_ab_nt.txt). A round of a wave is ~1 250 instructions of which a third
// (the exact ellipse-vs-quad tests of the 64 fetched records, the ballots, the staging and the four quad lists) does not depend on
// the pixels' running state. Here a block has a TESTER wave (waves 4..7: fetches the records, tests them, stages the survivors and
// builds the quad lists of round r + 1 into the other of two LDS buffers) and a BLENDER wave (waves 0..3: owns the 64 pixels,
// writes the checkpoints, composites round r out of its buffer): 8 waves per tile, the blender's chain a third shorter.
//   tester -> blender  ready[block][buf] = ((round + 1) << 8) | longest list (0..64), or | GSR_PAIR_END behind the list's end
//   blender -> tester  freed[block][buf] = round + 1 once the buffer is consumed; GSR_PAIR_STOP when all 64 pixels have stopped;
//                      alive[block] = the pixels still alive (the tester gates the quads with it, a round or two stale: a quad
//                      list then holds entries for pixels that have stopped since -- blended with done = true, i.e. not at all)
// Same arithmetic, same macros as gsr_render_fwd_serial<true>: images, checkpoints and work list are bit-identical to its output;
// the quad masks left for the backward may carry the stale-gate entries above (its own per-pixel tests drop them).
// Spins are bounded: a lost hand-shake (2^20 polls of a flag the partner never sets) ends the walk instead of hanging the GPU, and the
// block's pixels are written as NaN -- visible in the image and in every loss, not a silently shorter walk.
// =========================================================================================
// Wave priorities inside the pair kernel (s_setprio, 0..3). The BLENDER of a block is the chain the kernel's length is made of; its
// tester has slack (it is at most a round ahead and waits for a free buffer). Round 6, same box, 1M Gaussians / 800^2: blender 3 /
// tester 0 takes the forward compositing from 0.1464 to 0.1310 ms -- the blenders issue whenever they can, the testers of all blocks
// fill the slots the blenders leave --, tester 1 (above the finished blocks' output / zero stores, which run at 0) to 0.1277. Measured
// and NOT kept (profiles/r06_ab_pair_priorities.txt): blender 2 = blender 3; tester 2 = tester 1; the blender's level kept through its tail
// 0.1331; levels that rise with the rounds walked 0.134-0.138, or once a blender has composited 300 .. 1 100 staged entries 0.127-0.129;
// a tester that takes the blender's level while the blender waits for it 0.126-0.128; a blender that waits at level 0 0.125-0.126 (and
// 0.089 -> 0.096 at 250k / 512^2); four workgroups per CU (<= 64 VGPRs) 0.155. (Priorities among the waves of the SERIAL walk, where every wave is a chain, did nothing in
// round 4: 0.1536 / 0.1536.)
#ifndef GSR_PAIR_PRIO_B
#define GSR_PAIR_PRIO_B 3
#endif
#ifndef GSR_PAIR_PRIO_T
#define GSR_PAIR_PRIO_T 1
#endif
#ifndef GSR_PAIR_PRIO_TAIL
#define GSR_PAIR_PRIO_TAIL 0      // priority of the blender behind its walk (outputs, work items, the slices of zeros)
#endif
#define GSR_PAIR_END 0xffu
#define GSR_PAIR_STOP 0xffffffffu
#define GSR_PAIR_SPINS (1 << 20)
__device__ __forceinline__ uint32_t lds_flag_load(const uint32_t* p) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void lds_flag_store(uint32_t* p, uint32_t v, int lane) {     // behind everything this wave wrote to LDS
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ void __launch_bounds__(512, 6)      // <= 80 VGPRs: three workgroups of eight waves per CU
gsr_render_fwd_pair(const uint32_t* __restrict__ tile_off, const SplatRec* __restrict__ recs,
                    const uint32_t* __restrict__ ids, int W, int H, int gx,
                    float* __restrict__ out_color, float* __restrict__ out_depth,
                    float* __restrict__ out_alpha, float* __restrict__ final_T,
                    uint32_t* __restrict__ n_contrib, float* __restrict__ totals /*[5][H*W]*/,
                    float* __restrict__ rec_base, const uint32_t* __restrict__ tile_seg,
                    const uint32_t* __restrict__ order, int seg_shift,
                    uint32_t* __restrict__ plan_off, uint4* __restrict__ plan_items,
                    unsigned long long* __restrict__ plan_total, uint32_t plan_cap, unsigned long long qmask_views, uint32_t sink_rec,
                    const unsigned long long* __restrict__ counters, uint32_t capacity, uint32_t maxc_cap, ViewSplit vs,
                    float4* __restrict__ zero4, uint32_t zero_n, uint32_t zero_per, ZeroSide zs) {
    if (counters[2] > (unsigned long long)capacity || counters[3] > (unsigned long long)maxc_cap) {
        if ((qmask_views >> 62) & 1ull) poison_tile(vs, W, H, gx, out_color, out_depth, out_alpha);
        return;
    }
    __shared__ float4 stage[4][2][3][GSR_RB + 2];                            // [block][buffer][a | b | c][slot]; slot 64 = the all-zero record
    __shared__ __attribute__((aligned(8))) uint8_t qlist[4][2][4][80];
    __shared__ uint32_t ready[4][2], freed[4][2];
    __shared__ unsigned long long alive_pub[4];
    __shared__ uint32_t wl[4];
    __shared__ uint32_t plan_base;
    const bool keep_state = (qmask_views >> 63) == 0ull;  // (as in gsr_render_fwd_serial)
    if (blockIdx.x == 0 && threadIdx.x == 0) plan_total[GSR_CNT_QMASK - GSR_CNT_PLAN] = qmask_views & ~(1ull << 62);
    const int tg = (int)order[blockIdx.x];                // heaviest tiles first
    const int view = tg / vs.tiles_per_view;
    if (!((vs.view_mask >> view) & 1u)) { clear_slice<512>(zero4, zero_n, zero_per); clear_side<512>(zs); return; }
    const int tile = tg - view * vs.tiles_per_view;
    const float* __restrict__ bg = vs.bg[view];
    {
        const size_t HWv = (size_t)W * H;
        recs += (size_t)view * vs.N;
        out_color += view * 3 * HWv; out_depth += view * HWv; out_alpha += view * HWv;
        final_T += view * vs.img_stride; n_contrib += view * vs.img_stride; totals += view * vs.img_stride;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = wave & 3;                             // the 8x8 block; waves blk (blender) and blk + 4 (tester) share it
    const bool tester = wave >= 4;
    const int lane = threadIdx.x & 63;
    const int row = lane >> 4, l15 = lane & 15;
    const int bx = (tile % gx) * GSR_TILE + (blk & 1) * 8;
    const int by = (tile / gx) * GSR_TILE + (blk >> 1) * 8;
    const int lx = (row & 1) * 4 + (l15 & 3), ly = (row >> 1) * 4 + (l15 >> 2);
    const int px = bx + lx, py = by + ly;
    const bool inside = (px < W) && (py < H);
    const float pxf = (float)px, pyf = (float)py;
    const float bx0 = (float)bx, by0 = (float)by;
    const uint32_t start = tile_off[tg];
    const uint32_t tile_n = tile_off[tg + 1] - start;
    const uint32_t n = (bx < W && by < H) ? tile_n : 0u;   // a block outside the image walks nothing
    float* __restrict__ recw = rec_base + (size_t)(keep_state ? tile_seg[tg] : 0u) * GSR_CKPT_FLOATS;
    float* __restrict__ rec0 = recw + (blk * 64 + ly * 8 + lx);
    for (int q = threadIdx.x; q < 4 * 2 * 3 * (GSR_RB + 2); q += 512) (&stage[0][0][0][0])[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = threadIdx.x; q < 4 * 2 * 4 * 80 / 4; q += 512) reinterpret_cast<uint32_t*>(&qlist[0][0][0][0])[q] = 0u;
    if (threadIdx.x < 8) { (&ready[0][0])[threadIdx.x] = 0u; (&freed[0][0])[threadIdx.x] = 0u; }
    if (threadIdx.x < 4) alive_pub[threadIdx.x] = ~0ull;   // until the blender publishes: every quad takes entries
    lds_barrier();

    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f;
    uint32_t last = 0;
    bool done = !inside;
    float* const sinkf = rec_base + (size_t)sink_rec * GSR_CKPT_FLOATS + (blk * 64 + lane);
    unsigned long long* const sink64 = reinterpret_cast<unsigned long long*>(rec_base + (size_t)sink_rec * GSR_CKPT_FLOATS + GSR_REC_HINT) + lane;
    bool lost = false;                                    // (blender) the hand-shake broke: the block comes out as NaN, not as a silently shorter walk
    if (n > 0u && tester) {
        if (GSR_PAIR_PRIO_T) __builtin_amdgcn_s_setprio(GSR_PAIR_PRIO_T);
        // ---- the tester: two register sets take turns (round r tests set r & 1, requested a round ago, and requests round r + 1's
        // records and round r + 2's list entries); it is never more than one round ahead of the blender, so its own latencies hide
        // behind the wait for a free buffer
        float4 r0a, r0b, r0c, r1a, r1b, r1c;
        uint32_t idq;
        {
            const uint32_t id0 = ids[start + min((uint32_t)lane, n - 1u)];
            idq = ids[start + min(GSR_RB + (uint32_t)lane, n - 1u)];
            const float4* __restrict__ p0 = reinterpret_cast<const float4*>(recs + id0);
            r0a = p0[0]; r0b = p0[1]; r0c = p0[2];
        }
        unsigned long long alive = __ballot(inside);
        auto wait_free = [&](const uint32_t* f, uint32_t want) -> bool {     // false: the blender has stopped (or the hand-shake is lost)
            uint32_t v = lds_flag_load(f);
#pragma unroll 1
            for (int spin = 0; v != want && v != GSR_PAIR_STOP && spin < GSR_PAIR_SPINS; ++spin) { __builtin_amdgcn_s_sleep(2); v = lds_flag_load(f); }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            return v == want;
        };
        auto test_round = [&](const uint32_t r, float4 ra, float4 rb, float4 rc, float4& da, float4& db, float4& dc) -> bool {
            const uint32_t rel = r * GSR_RB;
            const int buf = (int)(r & 1u);
            if (rel >= n) {                                 // behind the end of the list
                if (r >= 2u && !wait_free(&freed[blk][buf], r - 1u)) return false;
                lds_flag_store(&ready[blk][buf], ((r + 1u) << 8) | GSR_PAIR_END, lane);
                return false;
            }
            {
                const float4* __restrict__ p = reinterpret_cast<const float4*>(recs + idq);
                da = p[0]; db = p[1]; dc = p[2];
                idq = ids[start + min(rel + 2u * GSR_RB + (uint32_t)lane, n - 1u)];
            }
            const uint32_t i = rel + lane;
            bool h0 = false, h1 = false, h2 = false, h3 = false;
            if (i < n) {
                const float thr = min_visible_power(rb.y);
                float qp[4];
                quad_max_powers(ra.x, ra.y, ra.z, ra.w, rb.x, bx0, by0, qp);
                h0 = ((alive & 0x000000000000ffffull) != 0ull) && qp[0] >= thr;
                h1 = ((alive & 0x00000000ffff0000ull) != 0ull) && qp[1] >= thr;
                h2 = ((alive & 0x0000ffff00000000ull) != 0ull) && qp[2] >= thr;
                h3 = ((alive & 0xffff000000000000ull) != 0ull) && qp[3] >= thr;
            }
            const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
            {   // the round's four hit masks for the backward (GSR_CNT_QMASK), as gsr_render_fwd_serial<true> leaves them
                unsigned long long* mp = reinterpret_cast<unsigned long long*>(recw + (size_t)(rel >> seg_shift) * GSR_CKPT_FLOATS + GSR_REC_HINT)
                                         + (((rel >> 6) & ((1u << (seg_shift - 6)) - 1u)) * 16u + (uint32_t)blk * 4u);
                *((lane < 4 && keep_state) ? mp + lane : sink64) = lane == 0 ? m0 : (lane == 1 ? m1 : (lane == 2 ? m2 : m3));
            }
            // the buffer is free once the blender has left round r - 2
            if (r >= 2u) { if (!wait_free(&freed[blk][buf], r - 1u)) return false; }
            else if (lds_flag_load(&freed[blk][buf]) == GSR_PAIR_STOP) return false;
            alive = __hip_atomic_load(&alive_pub[blk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // gate of the NEXT round's quads (a superset of the pixels alive then)
            int nmax = 0;
            if ((m0 | m1 | m2 | m3) != 0ull) {
                float4* __restrict__ sa = stage[blk][buf][0];
                float4* __restrict__ sb = stage[blk][buf][1];
                float4* __restrict__ sc = stage[blk][buf][2];
                uint8_t (*qlw)[80] = qlist[blk][buf];
                if (h0 | h1 | h2 | h3) { sa[lane] = ra; sb[lane] = rb; sc[lane] = rc; }
                if (h0) qlw[0][lanes_below(m0)] = (uint8_t)lane;
                if (h1) qlw[1][lanes_below(m1)] = (uint8_t)lane;
                if (h2) qlw[2][lanes_below(m2)] = (uint8_t)lane;
                if (h3) qlw[3][lanes_below(m3)] = (uint8_t)lane;
                const int n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), n3 = __popcll(m3);
                nmax = max(max(n0, n1), max(n2, n3));
                const int nread = (nmax + 7) & ~7;          // shorter lists end in the all-zero record up to the last batch that is read
                if (lane >= n0 && lane < nread) qlw[0][lane] = (uint8_t)GSR_RB;
                if (lane >= n1 && lane < nread) qlw[1][lane] = (uint8_t)GSR_RB;
                if (lane >= n2 && lane < nread) qlw[2][lane] = (uint8_t)GSR_RB;
                if (lane >= n3 && lane < nread) qlw[3][lane] = (uint8_t)GSR_RB;
            }
            lds_flag_store(&ready[blk][buf], ((r + 1u) << 8) | (uint32_t)nmax, lane);
            return true;
        };
        for (uint32_t r = 0;; r += 2u) {
            if (!test_round(r, r0a, r0b, r0c, r1a, r1b, r1c)) break;
            if (!test_round(r + 1u, r1a, r1b, r1c, r0a, r0b, r0c)) break;
        }
    } else if (n > 0u) {
        // ---- the blender
        __builtin_amdgcn_s_setprio(GSR_PAIR_PRIO_B);
#pragma unroll 1
        for (uint32_t r = 0;; ++r) {
            const uint32_t rel = r * GSR_RB;
            const int buf = (int)(r & 1u);
            uint32_t code = lds_flag_load(&ready[blk][buf]);
#pragma unroll 1
            for (int spin = 0; (code >> 8) != r + 1u && spin < GSR_PAIR_SPINS; ++spin) { __builtin_amdgcn_s_sleep(1); code = lds_flag_load(&ready[blk][buf]); }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            if ((code >> 8) != r + 1u) { lost = true; break; }      // the tester never delivered this round (2^20 polls): see below
            if ((code & 0xffu) == GSR_PAIR_END) break;
            const unsigned long long alive = __ballot(!done);
            if (alive == 0ull) {                            // every pixel has stopped: release the tester, leave
                lds_flag_store(&freed[blk][0], GSR_PAIR_STOP, lane);
                lds_flag_store(&freed[blk][1], GSR_PAIR_STOP, lane);
                break;
            }
            {   // segment cut: checkpoint for the backward
                const bool cut = keep_state && rel != 0u && (rel & ((1u << seg_shift) - 1u)) == 0u;
                float* c = (cut && inside) ? rec0 + (size_t)((rel >> seg_shift) - 1u) * GSR_CKPT_FLOATS : sinkf;
                c[0] = T; c[256] = C0; c[512] = C1; c[768] = C2; c[1024] = D; c[1280] = A;
            }
            const int nmax = (int)(code & 0xffu);
            if (nmax > 0) {
                const float4* __restrict__ sa = stage[blk][buf][0];
                const float4* __restrict__ sb = stage[blk][buf][1];
                const float4* __restrict__ sc = stage[blk][buf][2];
                const uint8_t* __restrict__ ql = qlist[blk][buf][row];
                const uint32_t pos1 = rel + 1u;               // 1-based list position of staged slot 0
                uint32_t lasts = 0xffffffffu;
                for (int jb = 0; jb < nmax; jb += 8) {
                    const uint2 sl = *reinterpret_cast<const uint2*>(ql + jb);
                    uint32_t slot[8];
#pragma unroll
                    for (int b = 0; b < 8; ++b) slot[b] = ((b < 4 ? sl.x : sl.y) >> (8 * (b & 3))) & 0xffu;
                    float4 e0a = sa[slot[0]], e0b = sb[slot[0]], e0c = sc[slot[0]];
#pragma unroll
                    for (int b = 0; b < 8; b += 2) {
                        if (jb + b < nmax) {                  // wave-uniform
                            const float4 e1a = sa[slot[b + 1]], e1b = sb[slot[b + 1]], e1c = sc[slot[b + 1]];
                            GSR_COMPOSITE2(e0a, e0b, e0c, slot[b], true, e1a, e1b, e1c, slot[b + 1], true, 1.f, true, lasts,
                                           if (b + 2 < 8) { e0a = sa[slot[b + 2]]; e0b = sb[slot[b + 2]]; e0c = sc[slot[b + 2]]; })
                        }
                    }
                }
                last = lasts != 0xffffffffu ? pos1 + lasts : last;
            }
            {   // the buffer goes back; the pixels still alive ride along for the tester's quad gate
                const unsigned long long still = __ballot(!done);
                if (lane == 0) __hip_atomic_store(&alive_pub[blk], still, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                lds_flag_store(&freed[blk][buf], r + 1u, lane);
            }
        }
    }
    __builtin_amdgcn_s_setprio(GSR_PAIR_PRIO_TAIL);
    if (lost) { T = C0 = C1 = C2 = D = A = __uint_as_float(0x7fc00000u); }     // (round-5 advisor: a lost hand-shake must not pass for a result)
    if (inside && !tester) {
        const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = fmaf(T, bg[0], C0);
        out_color[HW + pix] = fmaf(T, bg[1], C1);
        out_color[2 * HW + pix] = fmaf(T, bg[2], C2);
        out_depth[pix] = D;
        out_alpha[pix] = A;
        totals[pix] = C0; totals[HW + pix] = C1; totals[2 * HW + pix] = C2;
        totals[3 * HW + pix] = D; totals[4 * HW + pix] = A;
    }
    // ---- how deep the backward has to walk this tile's list, and its (tile, segment) work items
    if (!tester) {
        const uint32_t wmax = wave_max_u32(inside ? last : 0u);
        if (lane == 0) wl[blk] = wmax;
    }
    lds_barrier();
    if (threadIdx.x == 0) {
        const uint32_t tl = max(max(wl[0], wl[1]), max(wl[2], wl[3]));
        const uint32_t segs = (tl + (1u << seg_shift) - 1u) >> seg_shift;
        const uint32_t base = segs ? (uint32_t)atomicAdd(plan_total, (unsigned long long)segs) : 0u;
        plan_off[tg] = base;
        plan_base = base;
        wl[0] = segs;
    }
    lds_barrier();
    {
        const uint32_t segs = wl[0], base = plan_base;
        for (uint32_t q = threadIdx.x; q < segs; q += 512)
            if (base + q < plan_cap)
                plan_items[base + q] = make_uint4((uint32_t)tg, tile_seg[tg] + q, start, q | ((min(1u << seg_shift, tile_n - (q << seg_shift)) - 1u) << 24));
    }
    clear_slice<512>(zero4, zero_n, zero_per);
    clear_side<512>(zs);
}

// The exact walk of list positions [lo, hi) of a tile for the lanes with done == false (lane = pixel, row-major 8x8 block at
// (bx0, by0)), `gate` = their transmittance in front of the segment. Updates T, C0, C1, C2, D, A, last, done: the segment's own
// numbers up to the stopping entry (done is set by a stop only). Same fetch / exact cull / stage / composite as K5a, block lists.
#define GSR_WALK_SEGMENT(lo, hi, gate)                                                                     \
    for (uint32_t pos0 = (lo); pos0 < (hi); pos0 += GSR_RB) {                                              \
        if (__ballot(!done) == 0ull) break;                                                                \
        const uint32_t i = pos0 + lane;                                                                    \
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rc = ra;                                     \
        bool hit = false;                                                                                  \
        if (i < (hi)) {                                                                                    \
            const float4* __restrict__ p = reinterpret_cast<const float4*>(recs + ids[start + i]);         \
            ra = p[0]; rb = p[1]; rc = p[2];                                                               \
            hit = rect_max_power(ra.x, ra.y, ra.z, ra.w, rb.x, bx0, bx0 + 7.f, by0, by0 + 7.f) >= min_visible_power(rb.y); \
        }                                                                                                  \
        const unsigned long long mask = __ballot(hit);                                                     \
        if (mask == 0ull) continue;                                                                        \
        const int nhit = __popcll(mask);                                                                   \
        if (hit) {                                                                                         \
            const uint32_t pos = lanes_below(mask);                                                        \
            rc.z = __uint_as_float(i + 1u);                                                                \
            sa[pos] = ra; sb[pos] = rb; sc[pos] = rc;                                                      \
        }                                                                                                  \
        wave_lds_handoff();                                                                                \
        float4 e0a = sa[0], e0b = sb[0], e0c = sc[0];                                                      \
        for (int j = 0; j < nhit; j += 2) {                                                                \
            const float4 e1a = sa[j + 1], e1b = sb[j + 1], e1c = sc[j + 1];                                \
            const uint32_t k0 = __float_as_uint(e0c.z);                                                    \
            GSR_COMPOSITE2(e0a, e0b, e0c, k0, true, e1a, e1b, e1c, __float_as_uint(e1c.z), j + 1 < nhit, gate, true, last, \
                           e0a = sa[j + 2]; e0b = sb[j + 2]; e0c = sc[j + 2];)                             \
        }                                                                                                  \
        wave_lds_handoff();                                                                                \
    }

#define GSR_DEFERRED 0x80000000u    // n_contrib of a pixel between K5b and K5c: stops inside segment (value & 0x7fffffff)
#define GSR_WALK_SLOTS 32           // walk items a wave of K5b can hand to K5c (more: it walks them itself)

// =========================================================================================
// K5b: forward, chaining the segments. One workgroup per tile, wave = 8x8 block, lane = pixel (row-major).
//
// Per pixel, front to back over the tile's segment records: P = running transmittance (1 at the start).
//   fl(P * T'_s) >= 1e-4 : no entry of segment s stops the pixel (T' only falls inside a segment and fl(P * .) is
//                          monotone): C += P * C'_s, ..., P = fl(P * T'_s)                       [a handful of FMAs]
//   fl(P * T'_s) <  1e-4 : the pixel stops INSIDE segment s. Which entry stops it takes a walk of that segment with P as
//                          the gate of the stop test: deferred to K5c (one item per (block, segment) with such pixels,
//                          handed over through a list; the pixel is tagged GSR_DEFERRED | s in n_contrib) -- walked here,
//                          one after the other by a wave with < 2 neighbours per SIMD, the walks were 2/3 of this kernel
//   record skipped       : walked here, for the whole segment (only reachable through a wrong hint; see K5a)
// and writes back, in place, the pixel's ABSOLUTE state after every segment it passes: the checkpoint segment s + 1 of
// the backward -- and the walk of K5c -- starts from. Then the image outputs of the pixels that never stop, and -- what used
// to be two more kernels in front of the backward -- the tile's entries of the backward's work list (one atomic per tile).
// =========================================================================================
__global__ void __launch_bounds__(256)
gsr_render_fwd_combine(const uint32_t* __restrict__ tile_off, const SplatRec* __restrict__ recs,
                       const uint32_t* __restrict__ ids, int W, int H, int gx,
                       float* __restrict__ out_color, float* __restrict__ out_depth,
                       float* __restrict__ out_alpha, float* __restrict__ final_T,
                       uint32_t* __restrict__ n_contrib, float* __restrict__ totals /*[5][H*W]*/,
                       float* __restrict__ rec_base, const uint32_t* __restrict__ tile_seg,
                       const uint32_t* __restrict__ order, int seg_shift,
                       uint32_t* __restrict__ plan_off, uint4* __restrict__ plan_items,
                       unsigned long long* __restrict__ plan_total, uint32_t plan_cap,
                       uint2* __restrict__ walk_items, unsigned long long* __restrict__ walk_total,
                       const unsigned long long* __restrict__ counters, uint32_t capacity, uint32_t maxc_cap, ViewSplit vs,
                       float4* __restrict__ zero4, uint32_t zero_n, uint32_t zero_per, int poison /* an asynchronous forward */) {
    if (counters[2] > (unsigned long long)capacity || counters[3] > (unsigned long long)maxc_cap) {
        if (poison) poison_tile(vs, W, H, gx, out_color, out_depth, out_alpha);
        return;
    }
    __shared__ float4 stage[4][3][GSR_RB + 2];
    __shared__ uint32_t wl[4];
    __shared__ uint32_t plan_base;
    __shared__ uint16_t wseg[4][GSR_WALK_SLOTS];          // segments this wave hands to K5c
    __shared__ uint32_t wcnt[4], walk_base;
    if (blockIdx.x == 0 && threadIdx.x == 0) plan_total[GSR_CNT_QMASK - GSR_CNT_PLAN] = 0ull;   // plane 7 holds hints here, no quad masks
    const int tg = (int)order[blockIdx.x];                // heaviest tiles first
    const int view = tg / vs.tiles_per_view;
    const int tile = tg - view * vs.tiles_per_view;
    const float* __restrict__ bg = vs.bg[view];
    {
        const size_t HWv = (size_t)W * H;
        recs += (size_t)view * vs.N;
        out_color += view * 3 * HWv; out_depth += view * HWv; out_alpha += view * HWv;
        final_T += view * vs.img_stride; n_contrib += view * vs.img_stride; totals += view * vs.img_stride;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int bx = (tile % gx) * GSR_TILE + (wave & 1) * 8;
    const int by = (tile / gx) * GSR_TILE + (wave >> 1) * 8;
    const int px = bx + (lane & 7), py = by + (lane >> 3);
    const bool inside = (px < W) && (py < H);             // false for every lane of a block outside the image
    const float pxf = (float)px, pyf = (float)py;
    const float bx0 = (float)bx, by0 = (float)by;
    const uint32_t start = tile_off[tg];
    const uint32_t n = tile_off[tg + 1] - start;
    const uint32_t tile_n = n;
    const uint32_t nseg = (n + (1u << seg_shift) - 1u) >> seg_shift;
    float* __restrict__ rec0 = rec_base + (size_t)tile_seg[tg] * GSR_CKPT_FLOATS + (wave * 64 + lane);
    float4* __restrict__ sa = stage[wave][0];
    float4* __restrict__ sb = stage[wave][1];
    float4* __restrict__ sc = stage[wave][2];
    bool lds_ready = false;

    float P = 1.f, S0 = 0.f, S1 = 0.f, S2 = 0.f, SD = 0.f, SA = 0.f;     // the pixel's absolute state
    uint32_t last_abs = 0;                                // deepest blended position (deferred pixels: the end of their segment)
    uint32_t deferred = 0;                                // GSR_DEFERRED | stop segment
    uint32_t carry_last = 0;                              // ... and the pixel's deepest blended position in front of that segment
    uint32_t nwalk = 0;                                   // (wave-uniform) items handed to K5c
    bool alive = inside;
    // The records of the next GSR_CQ segments are in flight while one is chained: a queue in registers, one record
    // requested per segment passed (the chain is a string of dependent ~1 us loads otherwise: 30-80 segments per tile)
#define GSR_CQ 4
    float qT[GSR_CQ], q0[GSR_CQ], q1[GSR_CQ], q2[GSR_CQ], qD[GSR_CQ], qA[GSR_CQ];
    uint32_t qL[GSR_CQ];
#pragma unroll
    for (int j = 0; j < GSR_CQ; ++j) {
        qT[j] = 0.f; q0[j] = 0.f; q1[j] = 0.f; q2[j] = 0.f; qD[j] = 0.f; qA[j] = 0.f; qL[j] = 0u;
        if ((uint32_t)j < nseg && inside) {
            const float* __restrict__ r = rec0 + (size_t)j * GSR_CKPT_FLOATS;
            qT[j] = r[0]; q0[j] = r[256]; q1[j] = r[512]; q2[j] = r[768]; qD[j] = r[1024]; qA[j] = r[1280];
            qL[j] = reinterpret_cast<const uint32_t*>(r)[GSR_REC_LAST];
        }
    }
    for (uint32_t s = 0; s < nseg; ++s) {
        if (__ballot(alive) == 0ull) break;
        float* __restrict__ rec = rec0 + (size_t)s * GSR_CKPT_FLOATS;
        float T = qT[0], C0 = q0[0], C1 = q1[0], C2 = q2[0], D = qD[0], A = qA[0];            // the segment's own (T', C', ...)
        uint32_t last = qL[0];
#pragma unroll
        for (int j = 0; j + 1 < GSR_CQ; ++j) { qT[j] = qT[j + 1]; q0[j] = q0[j + 1]; q1[j] = q1[j + 1]; q2[j] = q2[j + 1]; qD[j] = qD[j + 1]; qA[j] = qA[j + 1]; qL[j] = qL[j + 1]; }
        if (s + GSR_CQ < nseg && inside) {
            const float* __restrict__ r = rec + (size_t)GSR_CQ * GSR_CKPT_FLOATS;
            qT[GSR_CQ - 1] = r[0]; q0[GSR_CQ - 1] = r[256]; q1[GSR_CQ - 1] = r[512]; q2[GSR_CQ - 1] = r[768];
            qD[GSR_CQ - 1] = r[1024]; qA[GSR_CQ - 1] = r[1280];
            qL[GSR_CQ - 1] = reinterpret_cast<const uint32_t*>(r)[GSR_REC_LAST];
        }
        const uint32_t lo = s << seg_shift, hi = min(lo + (1u << seg_shift), n);
        const bool skipped = alive && T < 0.f;            // the record holds nothing: walk it here
        bool defer = alive && !skipped && __fmul_rn(P, T) < 0.0001f;       // the pixel stops inside this segment
        if (__ballot(defer) != 0ull && nwalk >= (uint32_t)GSR_WALK_SLOTS) defer = false;   // no slot left: walked here like a skipped record
        const bool walk = alive && (skipped || (!defer && __fmul_rn(P, T) < 0.0001f));
        if (__ballot(defer) != 0ull) {
            if (lane == 0) wseg[wave][nwalk] = (uint16_t)s;
            ++nwalk;
            if (defer) { deferred = GSR_DEFERRED | s; carry_last = last_abs; last_abs = hi; alive = false; }
        }
        bool done = !walk;                                // lanes that do not walk keep the record's numbers
        if (__ballot(walk) != 0ull) {
            if (!lds_ready) {
                for (int q = lane; q < 3 * (GSR_RB + 2); q += 64) (&stage[wave][0][0])[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                wave_lds_handoff();
                lds_ready = true;
            }
            if (walk) { T = 1.f; C0 = 0.f; C1 = 0.f; C2 = 0.f; D = 0.f; A = 0.f; last = 0u; }
            GSR_WALK_SEGMENT(lo, hi, P)
        }
        const bool stopped = walk && done;                // (done was false for walking lanes and is set by a stop only)
        if (alive) {
            S0 = fmaf(P, C0, S0); S1 = fmaf(P, C1, S1); S2 = fmaf(P, C2, S2);
            SD = fmaf(P, D, SD); SA = fmaf(P, A, SA);
            P = __fmul_rn(P, T);
            last_abs = last != 0u ? last : last_abs;
            alive = !stopped;
        }
        if (inside) {   // the backward's checkpoint after segment s (lanes that have stopped: never read for a blended entry)
            rec[0] = P; rec[256] = S0; rec[512] = S1; rec[768] = S2; rec[1024] = SD; rec[1280] = SA;
        }
    }
    if (inside) {
        const size_t pix = (size_t)py * W + px, HW = (size_t)H * W;
        if (deferred != 0u) { n_contrib[pix] = deferred; final_T[pix] = __uint_as_float(carry_last); }   // K5c finishes this pixel
        else {
            final_T[pix] = P;
            n_contrib[pix] = last_abs;
            out_color[pix] = fmaf(P, bg[0], S0);
            out_color[HW + pix] = fmaf(P, bg[1], S1);
            out_color[2 * HW + pix] = fmaf(P, bg[2], S2);
            out_depth[pix] = SD;
            out_alpha[pix] = SA;
            totals[pix] = S0; totals[HW + pix] = S1; totals[2 * HW + pix] = S2;   // sums without background
            totals[3 * HW + pix] = SD; totals[4 * HW + pix] = SA;
        }
    }
    // ---- how deep the backward has to walk this tile's list (a deferred pixel counts to the end of its segment: the same
    // number of segments), its (tile, segment) work items, and the walk items of K5c
    {
        const uint32_t wmax = wave_max_u32(inside ? last_abs : 0u);
        if (lane == 0) { wl[wave] = wmax; wcnt[wave] = nwalk; }
    }
    lds_barrier();
    if (threadIdx.x == 0) {
        const uint32_t tl = max(max(wl[0], wl[1]), max(wl[2], wl[3]));
        const uint32_t segs = (tl + (1u << seg_shift) - 1u) >> seg_shift;
        const uint32_t base = segs ? (uint32_t)atomicAdd(plan_total, (unsigned long long)segs) : 0u;
        plan_off[tg] = base;
        plan_base = base;
        wl[0] = segs;
        const uint32_t nw = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        walk_base = nw ? (uint32_t)atomicAdd(walk_total, (unsigned long long)nw) : 0u;
    }
    lds_barrier();
    {
        const uint32_t segs = wl[0], base = plan_base;
        for (uint32_t q = threadIdx.x; q < segs; q += 256)
            if (base + q < plan_cap)     // {tile, the segment's record, list start, segment | (entries in it - 1) << 24}: all the backward needs, in one load
                plan_items[base + q] = make_uint4((uint32_t)tg, tile_seg[tg] + q, start, q | ((min(1u << seg_shift, tile_n - (q << seg_shift)) - 1u) << 24));
        uint32_t wb = walk_base;
        for (int w = 0; w < wave; ++w) wb += wcnt[w];
        if ((uint32_t)lane < nwalk) walk_items[wb + lane] = make_uint2((uint32_t)tg * 4u + (uint32_t)wave, (uint32_t)wseg[wave][lane]);
    }
    clear_slice(zero4, zero_n, zero_per);
}

// =========================================================================================
// K5c: forward, the pixels that stop inside a segment. One wave per item (tile, 8x8 block, segment) of K5b's list: the
// lanes tagged GSR_DEFERRED | segment start from the checkpoint in front of the segment (K5b's in-place record), walk the
// segment with that transmittance as the gate of the stop test, and write the pixel's outputs. Every pixel is in exactly one
// item; the walks of a tile run on as many waves as it has items instead of one after the other.
// =========================================================================================
__global__ void __launch_bounds__(256)
gsr_render_fwd_fix(const uint2* __restrict__ walk_items, const unsigned long long* __restrict__ walk_total,
                   const uint32_t* __restrict__ tile_off, const SplatRec* __restrict__ recs_all,
                   const uint32_t* __restrict__ ids, int W, int H, int gx,
                   float* __restrict__ out_color, float* __restrict__ out_depth,
                   float* __restrict__ out_alpha, float* __restrict__ final_T,
                   uint32_t* __restrict__ n_contrib, float* __restrict__ totals /*[5][H*W]*/,
                   const float* __restrict__ rec_base, const uint32_t* __restrict__ tile_seg, int seg_shift,
                   const unsigned long long* __restrict__ counters, uint32_t capacity, uint32_t maxc_cap, ViewSplit vs) {
    if (counters[2] > (unsigned long long)capacity || counters[3] > (unsigned long long)maxc_cap) return;
    __shared__ float4 stage[4][3][GSR_RB + 2];
    const uint32_t nitems = (uint32_t)walk_total[0];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (blockIdx.x * 4u + (uint32_t)wave >= nitems) return;
    float4* __restrict__ sa = stage[wave][0];
    float4* __restrict__ sb = stage[wave][1];
    float4* __restrict__ sc = stage[wave][2];
    for (int q = lane; q < 3 * (GSR_RB + 2); q += 64) (&stage[wave][0][0])[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    wave_lds_handoff();
    for (uint32_t it = blockIdx.x * 4u + (uint32_t)wave; it < nitems; it += gridDim.x * 4u) {
        const uint2 item = walk_items[it];
        const int tg = (int)(item.x >> 2), blk = (int)(item.x & 3u);
        const uint32_t s = item.y;
        const int view = tg / vs.tiles_per_view;
        const int tile = tg - view * vs.tiles_per_view;
        const float* __restrict__ bg = vs.bg[view];
        const SplatRec* __restrict__ recs = recs_all + (size_t)view * vs.N;
        const int bx = (tile % gx) * GSR_TILE + (blk & 1) * 8;
        const int by = (tile / gx) * GSR_TILE + (blk >> 1) * 8;
        const int px = bx + (lane & 7), py = by + (lane >> 3);
        const bool inside = (px < W) && (py < H);
        const float pxf = (float)px, pyf = (float)py;
        const float bx0 = (float)bx, by0 = (float)by;
        const size_t pix = (size_t)py * W + px, HW = (size_t)H * W, plane = (size_t)view * vs.img_stride;
        const bool mine = inside && n_contrib[plane + pix] == (GSR_DEFERRED | s);
        const uint32_t start = tile_off[tg];
        const uint32_t n = tile_off[tg + 1] - start;
        const uint32_t lo = s << seg_shift, hi = min(lo + (1u << seg_shift), n);
        float P = 1.f, S0 = 0.f, S1 = 0.f, S2 = 0.f, SD = 0.f, SA = 0.f;
        if (mine && s > 0u) {
            const float* __restrict__ r = rec_base + ((size_t)tile_seg[tg] + s - 1u) * GSR_CKPT_FLOATS + (blk * 64 + lane);
            P = r[0]; S0 = r[256]; S1 = r[512]; S2 = r[768]; SD = r[1024]; SA = r[1280];
        }
        float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f;
        uint32_t last = 0;
        bool done = !mine;
        GSR_WALK_SEGMENT(lo, hi, P)
        if (mine) {
            S0 = fmaf(P, C0, S0); S1 = fmaf(P, C1, S1); S2 = fmaf(P, C2, S2);
            SD = fmaf(P, D, SD); SA = fmaf(P, A, SA);
            P = __fmul_rn(P, T);
            if (last == 0u) last = __float_as_uint(final_T[plane + pix]);   // nothing blended in this segment: K5b left the position in front
            final_T[plane + pix] = P;
            n_contrib[plane + pix] = last;
            float* __restrict__ oc = out_color + (size_t)view * 3 * HW;
            oc[pix] = fmaf(P, bg[0], S0);
            oc[HW + pix] = fmaf(P, bg[1], S1);
            oc[2 * HW + pix] = fmaf(P, bg[2], S2);
            out_depth[(size_t)view * HW + pix] = SD;
            out_alpha[(size_t)view * HW + pix] = SA;
            float* __restrict__ tt = totals + plane;
            tt[pix] = S0; tt[HW + pix] = S1; tt[2 * HW + pix] = S2;
            tt[3 * HW + pix] = SD; tt[4 * HW + pix] = SA;
        }
    }
}
#undef GSR_WALK_SEGMENT
#undef GSR_COMPOSITE2
#undef GSR_COMPOSITE_G

// =========================================================================================
// Backward, FRONT TO BACK and depth-segmented.
//
// dL/dalpha_i = T_i (c_i.g) - [ sum_{j>i} w_j (c_j.g) + T_final (bg.g) ] / (1 - alpha_i)
// with c.g the incoming-gradient-weighted scalar cr gC0 + cg gC1 + cb gC2 + depth gD + gA, and
// sum_{j>i} = total - prefix_i - w_i (c_i.g): only FORWARD-running quantities (T and one scalar
// prefix) are needed, and a tile's list can be cut into independent segments: the forward leaves
// (T, C0, C1, C2, D, A) per pixel at every 2^seg_shift-th list position, and workgroup (tile, s)
// starts from checkpoint s.
//
// Per (Gaussian, pixel) pair everything the ten per-Gaussian sums need is two scalars:
//   m = opacity * G * dL/dalpha   (dL/d exponent, up to ln2)      w = alpha * T
// The workgroup adds RAW MOMENTS into an LDS table [segment position][12]:
//   [0] S_x = sum m dx  [1] S_y  [2] S_xx  [3] S_xy  [4] S_yy  [5] S_0 = sum m
//   [6..8] sum w gC{0,1,2}  [9] sum w gD
// bwd_flush() converts them to the accumulator layout K6 reads (gsr_device.h) once per
// (segment, Gaussian) and adds them to g2d with coalesced global atomics (agent-scope float
// atomics execute at the memory side on this multi-XCD part: one request per cache line here
// instead of three per (8x8 block, Gaussian)).
// =========================================================================================
#define GSR_BWD_PARAMS                                                                            \
    const uint32_t* __restrict__ tile_off, const SplatRec* __restrict__ recs,                     \
    const uint32_t* __restrict__ ids, int W, int H, int gx,                                       \
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,                    \
    const float* __restrict__ totals, const float* __restrict__ ckpt,                             \
    const uint32_t* __restrict__ tile_seg, const float* __restrict__ dL_dcolor,                   \
    const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dalpha,                     \
    float* __restrict__ g2d, int seg_shift, const uint4* __restrict__ plan_items,                 \
    const uint32_t* __restrict__ plan_off, const unsigned long long* __restrict__ plan_total,     \
    ViewSplit vs, ZeroRegions zr, uint8_t* __restrict__ live /* [views][GSR_LIVE_BYTES(N)], cleared with g2d */,               \
    uint32_t gnull /* bit 0 / 1 / 2: the caller passed no dL_dcolor / dL_ddepth / dL_dalpha (zeros; the pointer is a readable dummy) */, \
    unsigned long long* __restrict__ det64 /* GSR_VIEW_DETERMINISTIC: [views][N][GSR_Q2_ROW] 64-bit fixed-point sums (zeroed), else NULL */, \
    const uint32_t* __restrict__ det_gmax /* ... and the bits of the largest |incoming gradient| sum of any pixel (gsr_grad_absmax) */

// -----------------------------------------------------------------------------------------
// K5b: quad lists + two passes + fixed-point accumulation.
//
// Every 16-lane DPP row of a wave owns a 4x4 pixel quad (row r: quad (r & 1, r >> 1) of the 8x8
// block, lane l15: pixel (l15 & 3, l15 >> 2) of the quad). A fetched record is tested exactly
// against each of the four quads (quad_max_powers); the survivors are staged once (slot =
// fetching lane) and every quad gets its own byte list of staged slots, so a row iterates only
// over Gaussians that can blend in ITS sixteen pixels and the wave loops to the longest of the
// four lists (at 1M Gaussians 1.5x fewer trips than one 8x8 list, tests/lane_stats.py).
//
// pass 1 (lane = pixel), eight list entries per batch: the serial front-to-back recurrence of
//   T and the c.g prefix; (m, w) go to LDS:  mw[wave][quad][cell][8 pixels] float2 -- sixteen (entry k, pixel half h)
//   cells of 16 floats at a 20-float pitch: cell = (k & 3) + 8 (k >> 2) + 4 h. The 16 lanes a ds_read_b128 services per
//   LDS cycle in pass 2 hit 16 distinct 4-bank slots (the pitch), and the two halves of an entry sit 80 floats = 16 banks
//   (mod 32) apart, so a 16-lane row's ds_write_b64 covers the 32 banks once (with h next to k -- pitch 40 / 20 -- the
//   second half overlapped the first one's banks 0..3: a 2-way conflict on every store, 32 extra LDS cycles per batch).
// pass 2 (lane = (quad, half h = l15 >> 3, entry k = l15 & 7)): each lane sums ITS entry over
//   the eight pixels of ITS half -- S_0, S_x, S_y, S_xx, S_xy, S_yy and the four w * g sums are
//   plain FMAs against per-pixel gradients held in registers -- the two halves meet through one
//   DPP row rotation per value, and each lane adds five of the ten totals to the workgroup's
//   table. No transposing reduction, no per-entry cross-lane chain.
//
// The table is 64-bit FIXED POINT. Measured on gfx950 (profiles/r02_ubench_lds_valu.txt):
// ds_add_f32 costs 2.8 LDS cycles PER ACTIVE LANE (177 cycles for a full wave), ds_add_u64 5.9
// cycles per instruction; with float atomics the LDS pipe was 76% busy and this kernel slower
// than the round-1 one. Every contributor of a table row derives the same exponent from data
// all of them see: |m| <= 101 cmax gsum_p (T <= 1, opacity G <= 1, 1/(1 - alpha) <= 100, colour
// components and depth <= cmax -- a per-launch maximum K1 leaves in counters[5] -- and
// gsum_p = |gC0| + |gC1| + |gC2| + |gD| + |gA| of the pixel), at most 256 pixels add into a row,
// and |dx|, |dy| <= R = max(|x - tile centre x|, |y - tile centre y|) + 7.5. With
// 2^e0 * 256 * 101 * cmax * max_tile(gsum) < 2^60 the zeroth moments and the w sums use e0, the
// first moments e0 - eR, the second e0 - 2 eR (2^eR > R): no overflow, resolution below 1e-12 of
// the largest possible row total, and the workgroup's sums no longer depend on the order of the adds.
// -----------------------------------------------------------------------------------------
#define GSR_Q2_BATCH 8
#define GSR_Q2_KOFF(k) (20 * (k) + ((k) >= 4 ? 80 : 0))   // float offset of entry k's half-0 cell inside a quad's batch: cells 0..3 | 8..11
#define GSR_Q2_HSTRIDE 80     // floats from an entry's half 0 to its half 1: cells 4..7 | 12..15
#define GSR_QL_PITCH 80       // bytes per quad list (64 + zero padding, multiple of 8)
#define GSR_Q2_ROW 10         // u64 per table row: S_x S_y S_xx S_xy S_yy | S_0 W0 W1 W2 W3

__device__ __forceinline__ float row_max_f(float v) {       // every lane of a 16-lane row receives the row's maximum
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140 /* row_mirror */, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141 /* row_half_mirror */, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false)));
    return v;
}
// v + (v of the lane eight positions away inside the 16-lane row)
__device__ __forceinline__ float add_other_half(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(8), 0xf, 0xf, true));
}
// v * 2^e as a two's-complement 64-bit integer (|v * 2^e| < 2^62; exact: the split is done on exactly representable floats)
__device__ __forceinline__ unsigned long long to_fixed(float v, int e) {
    const float x = ldexpf(v, e);
    const float hi = floorf(x * 2.3283064365386963e-10f);            // floor(x / 2^32)
    const float lo = fmaf(-hi, 4294967296.f, x);                      // in [0, 2^32)
    return ((unsigned long long)(uint32_t)(int32_t)hi << 32) | (unsigned long long)(uint32_t)lo;
}
__device__ __forceinline__ float from_fixed(unsigned long long v, int e) {
    // two conversions + one fma instead of the 13-instruction i64 -> f32 sequence (double rounding: <= 1 ulp)
    const float hi = (float)(int32_t)(uint32_t)(v >> 32), lo = (float)(uint32_t)v;
    return ldexpf(fmaf(hi, 4294967296.f, lo), -e);
}
// exponent with 2^result > R for the per-row scale of the moments; the SAME expression in pass 2 and in the flush
__device__ __forceinline__ int row_radius_exp(float gx, float gy, float tcx, float tcy) {
    return __builtin_amdgcn_frexp_expf(fmaxf(fabsf(gx - tcx), fabsf(gy - tcy)) + 7.5f);
}

// ---- the deterministic backward (GSR_VIEW_DETERMINISTIC) ------------------------------------------------------------------------
// The default flush adds every workgroup's ten per-Gaussian sums to the accumulators with FLOAT atomics: a Gaussian that lies in k
// tiles receives k additions in whatever order the workgroups finish, and fp32 addition is not associative -- the gradients are
// reproducible up to the last bits only. In this mode the sums travel as 64-bit FIXED-POINT integers whose
// scale every contributor of a Gaussian derives from the same data -- integer addition is associative, so the totals (and every
// gradient behind them) are bit-identical from run to run:
//   e0g  from the LAUNCH's largest per-pixel gradient sum (gsr_grad_absmax, one small kernel in front) and K1's colour / depth
//        bound, exactly as the workgroup table's e0 is derived from the tile's;
//   eT   2^eT >= the tiles of the Gaussian's emission rectangle (at most that many workgroups add to it);
//   eR   2^eR > the largest |dx|, |dy| between the Gaussian's centre and any pixel of that rectangle;
// zeroth moments and colour sums at 2^(e0g - eT), first moments at 2^(e0g - eT - eR), second at 2^(e0g - eT - 2 eR).
// gsr_g2d_from_fixed then turns the totals into the float accumulator layout K6 reads (and sets the live flags).
__device__ __forceinline__ void det_exponents(float gx, float gy, uint32_t rectx, uint32_t recty, int& eT, int& eR) {
    const int x0 = (int)(rectx & 0xffffu), x1 = (int)(rectx >> 16), y0 = (int)(recty & 0xffffu), y1 = (int)(recty >> 16);
    const int nt = max(1, (x1 - x0) * (y1 - y0));
    eT = 32 - __builtin_clz((unsigned)nt);                // 2^eT > nt
    const float rx = fmaxf(fabsf(gx - (float)(x0 * GSR_TILE)), fabsf((float)(x1 * GSR_TILE) - gx));
    const float ry = fmaxf(fabsf(gy - (float)(y0 * GSR_TILE)), fabsf((float)(y1 * GSR_TILE) - gy));
    eR = __builtin_amdgcn_frexp_expf(fmaxf(rx, ry) + 1.f);
}
// det_gmax[0] = bits of the launch's largest per-pixel gradient sum, [1] = of the largest |background| component; cmax_k1 = K1's bound
__device__ __forceinline__ int det_e0(const uint32_t* __restrict__ det_gmax, float cmax_k1) {
    const uint32_t gb = det_gmax[0];
    const float g = gb >= 0x7f800000u ? 1.f : __uint_as_float(gb);
    const float cmax = fmaxf(fmaxf(cmax_k1, 1.f), __uint_as_float(det_gmax[1]));
    return 60 - __builtin_amdgcn_frexp_expf(25856.f * cmax * fmaxf(g, 1e-30f));
}
// bits of max over the pixels of |gC0| + |gC1| + |gC2| + |gD| + |gA| (non-negative floats order like their bits; NaN / inf end up >= 0x7f800000)
extern "C" __global__ void __launch_bounds__(256)
gsr_grad_absmax(const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dalpha,
                uint32_t gnull, size_t HW, int views, ViewSplit vs, uint32_t* __restrict__ out /* [0] gradients, [1] max |background| */) {
    if (blockIdx.x == 0 && (int)threadIdx.x < views) {
        const float* bg = vs.bg[threadIdx.x];
        atomicMax(out + 1, __float_as_uint(fmaxf(fabsf(bg[0]), fmaxf(fabsf(bg[1]), fabsf(bg[2])))));
    }
    uint32_t m = 0u;
    const size_t total = HW * (size_t)views;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < total; i += (size_t)gridDim.x * 256u) {
        const size_t v = i / HW, p = i - v * HW;
        float sgm = 0.f;
        if (!(gnull & 1u)) sgm += fabsf(dL_dcolor[v * 3 * HW + p]) + fabsf(dL_dcolor[v * 3 * HW + HW + p]) + fabsf(dL_dcolor[v * 3 * HW + 2 * HW + p]);
        if (!(gnull & 2u)) sgm += fabsf(dL_ddepth[i]);
        if (!(gnull & 4u)) sgm += fabsf(dL_dalpha[i]);
        m = max(m, __float_as_uint(sgm));
    }
    m = wave_max_u32(m);
    if ((threadIdx.x & 63u) == 0u && m) atomicMax(out, m);
}
// the fixed-point totals -> the accumulator layout of the float flush (out[0..9] there), live flags
extern "C" __global__ void __launch_bounds__(256)
gsr_g2d_from_fixed(int N, int views, const SplatRec* __restrict__ recs, const unsigned long long* __restrict__ det64,
                   const uint32_t* __restrict__ det_gmax, const unsigned long long* __restrict__ plan_total,
                   float* __restrict__ g2d, uint8_t* __restrict__ live) {
    const size_t total = (size_t)N * (size_t)views;
    const uint32_t gbits = det_gmax[0];
    const bool poisoned = gbits >= 0x7f800000u;
    const int e0g = det_e0(det_gmax, __uint_as_float((uint32_t)plan_total[1]));
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < total; i += (size_t)gridDim.x * 256u) {
        const unsigned long long* v = det64 + i * GSR_Q2_ROW;
        unsigned long long w[GSR_Q2_ROW];
        bool any = false;
#pragma unroll
        for (int q = 0; q < GSR_Q2_ROW; ++q) { w[q] = v[q]; any = any || (w[q] != 0ull); }
        if (!any) continue;                               // (accumulators and flags were cleared with the rest)
        const SplatRec r = recs[i];
        int eT, eR;
        det_exponents(r.x, r.y, r.rectx, r.recty, eT, eR);
        const int e = e0g - eT;
        const float Sx = from_fixed(w[0], e - eR), Sy = from_fixed(w[1], e - eR);
        const float Sxx = from_fixed(w[2], e - 2 * eR), Sxy = from_fixed(w[3], e - 2 * eR), Syy = from_fixed(w[4], e - 2 * eR);
        const float S0 = from_fixed(w[5], e);
        float out[GSR_G2D_STRIDE];
        out[0] = 2.f * r.qa * Sx + r.qb * Sy;
        out[1] = 2.f * r.qc * Sy + r.qb * Sx;
        out[2] = -0.5f * Sxx; out[3] = -Sxy; out[4] = -0.5f * Syy;
        out[5] = r.opac != 0.f ? S0 / r.opac : 0.f;
        out[6] = from_fixed(w[6], e); out[7] = from_fixed(w[7], e); out[8] = from_fixed(w[8], e); out[9] = from_fixed(w[9], e);
        out[10] = out[11] = 0.f;
        if (poisoned) {
#pragma unroll
            for (int q = 0; q < 10; ++q) out[q] = __uint_as_float(0x7fc00000u);
        }
        const size_t view = i / (size_t)N;
        float4* o4 = reinterpret_cast<float4*>(g2d + i * GSR_G2D_STRIDE);
        o4[0] = make_float4(out[0], out[1], out[2], out[3]); o4[1] = make_float4(out[4], out[5], out[6], out[7]); o4[2] = make_float4(out[8], out[9], 0.f, 0.f);
        live[view * GSR_LIVE_BYTES(N) + (i - view * (size_t)N)] = (uint8_t)1;
    }
}

__global__ void __launch_bounds__(256, 4)   // <= 128 VGPRs: four workgroups per CU (LDS: 34 KiB + 5 KiB table at 64-entry segments)
gsr_render_bwd_q2(GSR_BWD_PARAMS) {
    __shared__ float4 stage[4][3][GSR_RB + 1];                               // 12 KiB staged records, slot = fetching lane; slot 64 = an all-zero record (opacity 0: never blends)
    __shared__ __attribute__((aligned(8))) uint8_t qlist[4][4][GSR_QL_PITCH]; // 1.25 KiB [wave][quad][k] = staged slot of the quad's k-th entry
    __shared__ __attribute__((aligned(16))) float mw[4][4][GSR_Q2_BATCH * 40];   // 20 KiB: 16 cells of 16 floats at a 20-float pitch
    __shared__ uint32_t gmax_w[4];                                           // per wave: max of gsum over its pixels (bits of a non-negative float)
    extern __shared__ __attribute__((aligned(16))) unsigned long long acc64[];   // [(1 << seg_shift) * GSR_Q2_ROW]
    // Every gradient K6 produces starts from zero and K6 (gsr_preprocess_bwd_compact) then writes only the rows of the Gaussians a
    // pixel gradient reached (a quarter of a dense scene, 6 % of a trained one): the workgroups that HAVE an item clear a slice each
    // -- this kernel is bound by vector-instruction issue and leaves HBM idle; K6 was bound by its stores, 248 bytes per Gaussian and
    // three quarters of them zeros. Slices over the items, not over the launched grid: the grid is an upper bound whose tail (two
    // thirds of it at 1M Gaussians) is handed out when the kernel is nearly over. No item at all: every workgroup clears.
    const uint32_t nitems = (uint32_t)min(plan_total[0], (unsigned long long)gridDim.x);
    const uint32_t nclear = nitems ? nitems : gridDim.x;
    auto clear_rows = [&]() {
        if (zr.total4 == 0u || blockIdx.x >= nclear) return;
        const uint32_t per = (zr.total4 + nclear - 1u) / nclear;
        const uint32_t lo = min(blockIdx.x * per, zr.total4), hi = min(lo + per, zr.total4);   // (per * nclear < 2^32: total4 < 2^31 on the host)
        typedef float gsr_v4f __attribute__((ext_vector_type(4)));
        const gsr_v4f z = {0.f, 0.f, 0.f, 0.f};
        uint32_t first = 0;                               // the regions laid end to end, in float4s
        for (int r = 0; r < zr.count; ++r) {
            const uint32_t rlo = max(lo, first), rhi = min(hi, first + zr.n4[r]);
            gsr_v4f* __restrict__ dst = reinterpret_cast<gsr_v4f*>(zr.p[r]);
            for (uint32_t i = rlo + threadIdx.x; i < rhi; i += 256u) __builtin_nontemporal_store(z, dst + (i - first));
            if (blockIdx.x == 0 && threadIdx.x < zr.tail[r]) zr.p[r][(size_t)zr.n4[r] * 4 + threadIdx.x] = 0.f;
            first += zr.n4[r];
        }
    };
    clear_rows();                                         // (in front of the set-up loads: 2.5 us better than behind the flush, same box)
    if (blockIdx.x >= nitems) return;
    // {tile among all views' tiles, the segment's record, list start, segment | (entries - 1) << 24}: ONE scalar load, and every
    // vector load of the set-up below depends on nothing else (only the records hang off the list entries).
    const uint4 item = plan_items[blockIdx.x];
    const int tg = (int)item.x;
    const uint32_t rec_idx = item.y;
    const uint32_t start = item.z;
    const uint32_t seg = item.w & 0xffffffu;
    const uint32_t len = (item.w >> 24) + 1u;             // entries of this segment
    const int view = tg / vs.tiles_per_view;
    const int tile = tg - view * vs.tiles_per_view;
    const float* __restrict__ bg = vs.bg[view];
    {
        const size_t HWv = (size_t)W * H;
        recs += (size_t)view * vs.N;
        g2d += (size_t)view * vs.N * GSR_G2D_STRIDE;
        live += (size_t)view * GSR_LIVE_BYTES(vs.N);
        dL_dcolor += view * 3 * HWv; dL_ddepth += view * HWv; dL_dalpha += view * HWv;
        final_T += view * vs.img_stride; n_contrib += view * vs.img_stride; totals += view * vs.img_stride;
    }
    const uint32_t seg_lo = seg << seg_shift;
    const uint32_t seg_end = seg_lo + len;
    const bool have_masks = ((plan_total[GSR_CNT_QMASK - GSR_CNT_PLAN] >> view) & 1ull) != 0ull;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int row = lane >> 4, l15 = lane & 15;
    const int tx0 = (tile % gx) * GSR_TILE, ty0 = (tile / gx) * GSR_TILE;
    const int bx = tx0 + (wave & 1) * 8, by = ty0 + (wave >> 1) * 8;
    bool active = (bx < W) && (by < H);                   // wave-uniform; no early return: barriers below
    const int qx = bx + (row & 1) * 4, qy = by + (row >> 1) * 4;         // this row's quad
    const int lx = (row & 1) * 4 + (l15 & 3), ly = (row >> 1) * 4 + (l15 >> 2);
    const int px = bx + lx, py = by + ly;
    const bool inside = (px < W) && (py < H);
    const float pxf = (float)px, pyf = (float)py;
    const int cidx = wave * 64 + ly * 8 + lx;             // the forward's checkpoint slot of this pixel (row-major 8x8)
    // ---- every load of the set-up, issued back to back and BRANCH-FREE (indices clamped into range, the values of lanes that
    // must not see them replaced further down): a load inside a divergent `if` is followed by a register copy, i.e. by a wait
    // for memory right behind it, and a workgroup is a latency chain of which only four fit a CU.
    // (1) list entry of the first round -> record (the only dependent pair); entries past the segment's end repeat its last one
    //     and are masked where they are used
    const uint32_t id_first = ids[start + min(seg_lo + (uint32_t)lane, seg_end - 1u)];
    // (2) the pixel: clamped into the image
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)min(py, H - 1) * W + min(px, W - 1);
    const float l_T = final_T[pix];
    const uint32_t l_last = n_contrib[pix];
    const float l_g0 = dL_dcolor[pix], l_g1 = dL_dcolor[HW + pix], l_g2 = dL_dcolor[2 * HW + pix];
    const float l_gD = dL_ddepth[pix], l_gA = dL_dalpha[pix];
    const float l_t0 = totals[pix], l_t1 = totals[HW + pix], l_t2 = totals[2 * HW + pix], l_t3 = totals[3 * HW + pix], l_t4 = totals[4 * HW + pix];
    // (3) the segment's checkpoint = the state the forward left after the segment in front (segment 0: its own record, unused)
    const float* __restrict__ ckp = ckpt + (size_t)(rec_idx - (seg > 0u ? 1u : 0u)) * GSR_CKPT_FLOATS + cidx;
    const float l_c0 = ckp[0], l_c1 = ckp[256], l_c2 = ckp[512], l_c3 = ckp[768], l_c4 = ckp[1024], l_c5 = ckp[1280];
    // (4) the first round's quad masks (written by the serial forward; garbage without them: unused)
    const unsigned long long* __restrict__ mp0 = reinterpret_cast<const unsigned long long*>(ckpt + (size_t)rec_idx * GSR_CKPT_FLOATS + GSR_REC_HINT) + (uint32_t)wave * 4u;
    const unsigned long long km0 = mp0[0], km1 = mp0[1], km2 = mp0[2], km3 = mp0[3];
    // (5) the records of the first round
    float4 pa, pb, pc;
    {
        const float4* __restrict__ p = reinterpret_cast<const float4*>(recs + id_first);
        pa = p[0]; pb = p[1]; pc = p[2];
    }
    // wave 0 keeps what the flush needs of its first-round records (index, x, y, qa, qb, qc, opacity): table row r of the flush
    // is list entry seg_lo + r, i.e. exactly lane r's record here -- no second trip list -> record at the end of the chain
    const float4 keep_a = pa;
    const float keep_qc = pb.x, keep_op = pb.y;
    for (int q = threadIdx.x; q < (GSR_Q2_ROW << seg_shift); q += 256) acc64[q] = 0ull;
    for (int q = threadIdx.x; q < 4 * 4 * GSR_QL_PITCH / 4; q += 256) reinterpret_cast<uint32_t*>(&qlist[0][0][0])[q] = 0u;   // stale reads stay inside the stage
    // ... and read FINITE numbers: pass 1 is branch-free, a lane past its quad's list runs the arithmetic on whatever its stale slot
    // holds with alpha = 0 (0 * garbage must stay 0). A stale list byte is 0 (cleared above) or a slot an earlier round staged a
    // real record in: slot 0 is the only one that can be read before it was ever written
    if (lane < 3) { stage[wave][lane][0] = make_float4(0.f, 0.f, 0.f, 0.f); stage[wave][lane][GSR_RB] = make_float4(0.f, 0.f, 0.f, 0.f); }
    float4* __restrict__ sa = stage[wave][0];
    float4* __restrict__ sb = stage[wave][1];
    float4* __restrict__ sc = stage[wave][2];
    const uint8_t* __restrict__ ql = qlist[wave][row];

    // a pixel outside the image: transmittance 1, no contributor, zero incoming gradients
    const float T_final = inside ? l_T : 1.f;
    const uint32_t last_contrib = inside ? l_last : 0u;
    const bool use_c = inside && !(gnull & 1u), use_d = inside && !(gnull & 2u), use_a = inside && !(gnull & 4u);
    const float gC0 = use_c ? l_g0 : 0.f, gC1 = use_c ? l_g1 : 0.f, gC2 = use_c ? l_g2 : 0.f, gD = use_d ? l_gD : 0.f, gA = use_a ? l_gA : 0.f;
    const float Cg_total = inside ? l_t0 * gC0 + l_t1 * gC1 + l_t2 * gC2 + l_t3 * gD + l_t4 * gA : 0.f;
    const float ck0 = seg > 0u ? l_c0 : 1.f, ck1 = l_c1, ck2 = l_c2, ck3 = l_c3, ck4 = l_c4, ck5 = l_c5;    // (ck1..5 only read when seg > 0)
    // (scalar loads: requested in front of the barrier, not behind it)
    const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
    const float cmax = fmaxf(fmaxf(__uint_as_float((uint32_t)plan_total[1]), 1.f), fmaxf(fabsf(bg0), fmaxf(fabsf(bg1), fabsf(bg2))));
    float4* __restrict__ gtab = reinterpret_cast<float4*>(&mw[wave][0][0]);   // per-wave, read once below, then the space is pass 1's
    gtab[lane] = make_float4(gC0, gC1, gC2, gD);          // pass 2 reads other lanes' pixels
    {
        const float gsum = fabsf(gC0) + fabsf(gC1) + fabsf(gC2) + fabsf(gD) + fabsf(gA);
        const uint32_t wm = wave_max_u32(__float_as_uint(gsum));          // non-negative floats order like their bits
        if (lane == 0) gmax_w[wave] = wm;
    }
    lds_barrier();
    const uint32_t gmax_bits = max(max(gmax_w[0], gmax_w[1]), max(gmax_w[2], gmax_w[3]));   // (one barrier covers the zeroing above and these)
    const bool poisoned = gmax_bits >= 0x7f800000u;       // an infinite or NaN incoming gradient somewhere in the tile
    const float gmax = poisoned ? 1.f : __uint_as_float(gmax_bits);
    if (!(gmax > 0.f)) return;                            // block-uniform: a zero incoming gradient adds nothing anywhere
    const int e0 = 60 - __builtin_amdgcn_frexp_expf(25856.f * cmax * gmax);     // 2^e0 * (256 * 101 * cmax * gmax) < 2^60
    const float tcx = (float)tx0 + 7.5f, tcy = (float)ty0 + 7.5f;
    // deepest contributor of each quad (list positions are < 2^24: exact as floats) and of the wave
    const uint32_t row_last = (uint32_t)row_max_f((float)last_contrib);
    const uint32_t ql0 = __builtin_amdgcn_readlane(row_last, 0), ql1 = __builtin_amdgcn_readlane(row_last, 16),
                   ql2 = __builtin_amdgcn_readlane(row_last, 32), ql3 = __builtin_amdgcn_readlane(row_last, 48);
    const uint32_t wave_last = max(max(ql0, ql1), max(ql2, ql3));
    active = active && (wave_last > seg_lo);
    const uint32_t seg_hi = active ? min(seg_lo + (1u << seg_shift), wave_last) : seg_lo;
    const float Cg_behind0 = Cg_total + T_final * (bg0 * gC0 + bg1 * gC1 + bg2 * gC2);

    const float T_in = ck0;
    float T = T_in, Cgf = 0.f;
    if (seg > 0) Cgf = ck1 * gC0 + ck2 * gC1 + ck3 * gC2 + ck4 * gD + ck5 * gA;
    // what is composited behind the entry at hand, its own share included: (everything) - (in front of it); every entry takes its
    // share off (one fma) and what is left IS the numerator of its dL/dalpha
    float Rb = Cg_behind0 - Cgf;
    const float bx0 = (float)bx, by0 = (float)by;
    // pass 1 writes (m, w) of pixel l15 of entry k to mw1[k * KSTRIDE]; pass 2 lane (h2, k2) reads mw2[0..15]
    float* __restrict__ mw1 = &mw[wave][row][(l15 >> 3) * GSR_Q2_HSTRIDE + (l15 & 7) * 2];
    const int k2 = l15 & 7, h2 = l15 >> 3;
    const float* __restrict__ mw2 = &mw[wave][row][GSR_Q2_KOFF(k2) + h2 * GSR_Q2_HSTRIDE];
    const float qxf = (float)qx, qyf = (float)(qy + 2 * h2);                 // first pixel of pass 2's half
    float4 g2[8];                                                            // gradients of the 8 pixels of pass 2's half
    wave_lds_handoff();
#pragma unroll
    for (int i2 = 0; i2 < 8; ++i2) g2[i2] = gtab[row * 16 + h2 * 8 + i2];
    wave_lds_handoff();                                   // ... before pass 1 writes over the table

    // ea = x y qa qb | eb = qc opac r g | ec = b depth - - ; kpos = 1-based list position (row-uniform)
    // Pass 1 is BRANCH-FREE and handles two list entries per trip: what does not depend on the running (T, Cgf) -- exponent, G,
    // alpha, 1 / (1 - alpha), the colour dot product -- is computed for both first, so that the transcendentals issue back to back
    // (v_exp_f32 / v_rcp_f32 cost ~16 cycles each between plain VALU instructions and ~8 behind one another: tools/valu_rates.hip),
    // and a lane that does not blend an entry carries a_eff = 0 through the same arithmetic (alpha = 0, 1 - alpha = 1, w = m = 0
    // exactly) instead of sitting out an exec-masked region with its s_and_saveexec / s_cbranch_execz pair.
#define GSR_Q2_POWER(ea, eb, pw_)                                                                \
        const float pw_ = splat_power(ea.z, ea.w, eb.x, ea.x - pxf, ea.y - pyf);
#define GSR_Q2_HEAD(eb, ec, pw_, G_, slot_, ae_, al_, om_, cgi_)                                \
        float ae_, al_, om_, cgi_;                                                                    \
        {                                                                                        \
            const float araw = eb.y * G_;                                                        \
            /* alpha = min(0.99, araw) >= 1/255  <=>  araw >= 1/255: the forward's decision, bit for bit; the list position against */ \
            /* the pixel's last contributor as (staged slot) <= (last - first position of the round - 1): one compare */ \
            /* (plain `&`: no short-circuit, or the compiler rebuilds the exec-masked regions around the exponential) */ \
            const bool ok = (bool)((int)((int)(slot_) <= last_rel) & (int)(pw_ <= 0.f) & (int)(araw >= (1.0f / 255.0f))); \
            ae_ = ok ? araw : 0.f;                                                               \
            al_ = __builtin_amdgcn_fmed3f(ae_, 0.f, 0.99f);   /* min(0.99, a_eff), a_eff >= 0: one instruction, no canonicalising v_max in front */ \
            om_ = 1.f - al_;                                                                     \
            cgi_ = fmaf(eb.z, gC0, fmaf(eb.w, gC1, fmaf(ec.x, gC2, fmaf(ec.y, gD, gA))));        \
        }
#define GSR_Q2_TAIL(ae_, al_, om_, rc_, cgi_, kslot)                                             \
        {                                                                                        \
            const float w = al_ * T;                                                             \
            Rb = fmaf(-w, cgi_, Rb);                          /* behind this entry, without it */ \
            const float dL_dal = fmaf(T, cgi_, -(Rb * rc_));                                     \
            const float m = dL_dal * ae_;                     /* (opacity * dL/dalpha) * G */     \
            T *= om_;                                                                            \
            *reinterpret_cast<float2*>(mw1 + GSR_Q2_KOFF(kslot)) = make_float2(m, w);       \
        }

    for (uint32_t pos0 = seg_lo; pos0 < seg_hi; pos0 += GSR_RB) {   // 0-based positions pos0 .. pos0+63
        const uint32_t i = pos0 + lane;
        bool h0 = false, h1 = false, h2q = false, h3 = false;
        const float4 ra = pa, rb = pb, rc = pc;
        if (pos0 + GSR_RB < seg_hi) {                     // (wave-uniform) next round's records: in flight during this round; entries past the end repeat the last one
            const float4* __restrict__ p = reinterpret_cast<const float4*>(recs + ids[start + min(i + GSR_RB, seg_end - 1u)]);
            pa = p[0]; pb = p[1]; pc = p[2];
        }
        if (have_masks) {   // the forward's serial walk left the round's four quad masks: the same tests, already made
            const unsigned long long* __restrict__ mp = reinterpret_cast<const unsigned long long*>(ckpt + (size_t)rec_idx * GSR_CKPT_FLOATS + GSR_REC_HINT)
                                                        + (((pos0 >> 6) & ((1u << (seg_shift - 6)) - 1u)) * 16u + (uint32_t)wave * 4u);
            unsigned long long k0 = km0, k1 = km1, k2 = km2, k3 = km3;       // first round: requested at the top of the kernel
            if (pos0 != seg_lo) { k0 = mp[0]; k1 = mp[1]; k2 = mp[2]; k3 = mp[3]; }
            if (i < seg_hi) {
                h0 = (i < ql0) && ((k0 >> lane) & 1ull);
                h1 = (i < ql1) && ((k1 >> lane) & 1ull);
                h2q = (i < ql2) && ((k2 >> lane) & 1ull);
                h3 = (i < ql3) && ((k3 >> lane) & 1ull);
            }
        } else if (i < seg_hi) {
            const float thr = min_visible_power(rb.y);
            // exact ellipse-vs-quad support tests; an entry behind a quad's deepest contributor is never blended there
            float qp[4];
            quad_max_powers(ra.x, ra.y, ra.z, ra.w, rb.x, bx0, by0, qp);
            h0 = (i < ql0) && qp[0] >= thr;
            h1 = (i < ql1) && qp[1] >= thr;
            h2q = (i < ql2) && qp[2] >= thr;
            h3 = (i < ql3) && qp[3] >= thr;
        }
        const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2q), m3 = __ballot(h3);
        if ((m0 | m1 | m2 | m3) == 0ull) continue;
        if (h0 | h1 | h2q | h3) { sa[lane] = ra; sb[lane] = rb; sc[lane] = rc; }
        if (h0) qlist[wave][0][lanes_below(m0)] = (uint8_t)lane;
        if (h1) qlist[wave][1][lanes_below(m1)] = (uint8_t)lane;
        if (h2q) qlist[wave][2][lanes_below(m2)] = (uint8_t)lane;
        if (h3) qlist[wave][3][lanes_below(m3)] = (uint8_t)lane;
        const int n0 = __popcll(m0), n1 = __popcll(m1), n2 = __popcll(m2), n3 = __popcll(m3);
        const int nmax = max(max(n0, n1), max(n2, n3));
        const int nmine = row == 0 ? n0 : (row == 1 ? n1 : (row == 2 ? n2 : n3));
        {   // a list shorter than the longest one is padded with the all-zero record up to the last batch the loop below reads: pass 1
            // then needs no "is this entry inside my list" test per entry (opacity 0 fails the alpha test by itself)
            const int nread = (nmax + GSR_Q2_BATCH - 1) & ~(GSR_Q2_BATCH - 1);
            if (lane >= n0 && lane < nread) qlist[wave][0][lane] = (uint8_t)GSR_RB;
            if (lane >= n1 && lane < nread) qlist[wave][1][lane] = (uint8_t)GSR_RB;
            if (lane >= n2 && lane < nread) qlist[wave][2][lane] = (uint8_t)GSR_RB;
            if (lane >= n3 && lane < nread) qlist[wave][3][lane] = (uint8_t)GSR_RB;
        }
        const uint32_t accrow0 = pos0 - seg_lo;           // table row of staged slot 0
        const int last_rel = (int)last_contrib - (int)pos0 - 1;   // staged slot s is list position pos0 + s + 1 (1-based)
        wave_lds_handoff();
        for (int jb = 0; jb < nmax; jb += GSR_Q2_BATCH) {
            // ---- pass 1: entries jb .. jb+7 of every quad list (stale slots beyond a list: in range, masked)
            const uint2 sl = *reinterpret_cast<const uint2*>(ql + jb);
            uint32_t slot[GSR_Q2_BATCH];
#pragma unroll
            for (int b = 0; b < GSR_Q2_BATCH; ++b) slot[b] = ((b < 4 ? sl.x : sl.y) >> (8 * (b & 3))) & 0xffu;
            // the records of a pair are dead once its heads are computed: the next pair's are requested into the same registers right
            // there and travel during the pair's tails (no second register set: the kernel sits at 124 of 128 VGPRs)
            float4 ea = sa[slot[0]], eb = sb[slot[0]], ec = sc[slot[0]];
            float4 fa = sa[slot[1]], fb = sb[slot[1]], fc = sc[slot[1]];
#pragma unroll
            for (int b = 0; b < GSR_Q2_BATCH; b += 2) {
                if (jb + b < nmax) {                      // wave-uniform; the second entry of the pair may lie past every list: masked
                    GSR_Q2_POWER(ea, eb, pw0)
                    GSR_Q2_POWER(fa, fb, pw1)
                    float G0, G1, rc0, rc1;
                    GSR_TRANS_PAIR("v_exp_f32", G0, G1, pw0, pw1)
                    GSR_Q2_HEAD(eb, ec, pw0, G0, slot[b], ae0, al0, om0, cgi0)
                    GSR_Q2_HEAD(fb, fc, pw1, G1, slot[b + 1], ae1, al1, om1, cgi1)
                    if (b + 2 < GSR_Q2_BATCH) {
                        ea = sa[slot[b + 2]]; eb = sb[slot[b + 2]]; ec = sc[slot[b + 2]];
                        fa = sa[slot[b + 3]]; fb = sb[slot[b + 3]]; fc = sc[slot[b + 3]];
                    }
                    GSR_TRANS_PAIR("v_rcp_f32", rc0, rc1, om0, om1)
                    GSR_Q2_TAIL(ae0, al0, om0, rc0, cgi0, b)
                    GSR_Q2_TAIL(ae1, al1, om1, rc1, cgi1, b + 1)
                }
            }
            wave_lds_handoff();
            // ---- pass 2: lane (row, h2, k2) sums entry jb + k2 of its quad over the 8 pixels of half h2
            {
                const bool v2 = jb + k2 < nmine;
                const uint32_t s2 = ql[jb + k2];          // staged slot of this lane's entry (in range even when !v2)
                const float2 gxy = *reinterpret_cast<const float2*>(&sa[s2]);
                const float dxb = gxy.x - qxf, dyb = gxy.y - qyf;
                float S0 = 0.f, Sx = 0.f, Sy = 0.f, Sxx = 0.f, Sxy = 0.f, Syy = 0.f, W0 = 0.f, W1 = 0.f, W2 = 0.f, W3 = 0.f;
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {          // two pixel rows of the half
                    const float dy = dyb - (float)r2;
                    float R0 = 0.f, R1 = 0.f, R2 = 0.f;    // row sums of m, m dx, m dx^2
#pragma unroll
                    for (int c2 = 0; c2 < 4; c2 += 2) {
                        const float4 v = *reinterpret_cast<const float4*>(mw2 + (r2 * 4 + c2) * 2);   // m, w, m', w' of two pixels
                        const float4 g0 = g2[r2 * 4 + c2], g1 = g2[r2 * 4 + c2 + 1];
                        const float dx0 = dxb - (float)c2, dx1 = dxb - (float)(c2 + 1);
                        const float t0 = v.x * dx0, t1 = v.z * dx1;
                        R0 += v.x; R0 += v.z;
                        R1 += t0; R1 += t1;
                        R2 += t0 * dx0; R2 += t1 * dx1;
                        W0 += v.y * g0.x; W1 += v.y * g0.y; W2 += v.y * g0.z; W3 += v.y * g0.w;
                        W0 += v.w * g1.x; W1 += v.w * g1.y; W2 += v.w * g1.z; W3 += v.w * g1.w;
                    }
                    const float R0y = R0 * dy;
                    S0 += R0; Sx += R1; Sxx += R2;
                    Sy += R0y; Sxy += R1 * dy; Syy += R0y * dy;
                }
                // the other half of the same entry sits eight lanes away in the row
                S0 = add_other_half(S0); Sx = add_other_half(Sx); Sy = add_other_half(Sy);
                Sxx = add_other_half(Sxx); Sxy = add_other_half(Sxy); Syy = add_other_half(Syy);
                W0 = add_other_half(W0); W1 = add_other_half(W1); W2 = add_other_half(W2); W3 = add_other_half(W3);
                if (v2) {   // lane h2 = 0 adds table slots 0..4 (moments), lane h2 = 1 slots 5..9 (S_0 and the w sums)
                    const int eR = row_radius_exp(gxy.x, gxy.y, tcx, tcy);
                    const int e1 = h2 ? e0 : e0 - eR, e2 = h2 ? e0 : e0 - 2 * eR;
                    unsigned long long* a = acc64 + (accrow0 + s2) * GSR_Q2_ROW + h2 * 5;
                    atomicAdd(a + 0, to_fixed(h2 ? S0 : Sx, e1));
                    atomicAdd(a + 1, to_fixed(h2 ? W0 : Sy, e1));
                    atomicAdd(a + 2, to_fixed(h2 ? W1 : Sxx, e2));
                    atomicAdd(a + 3, to_fixed(h2 ? W2 : Sxy, e2));
                    atomicAdd(a + 4, to_fixed(h2 ? W3 : Syy, e2));
                }
            }
            wave_lds_handoff();                           // pass 2's reads precede the next batch's / round's writes
        }
    }
#undef GSR_Q2_POWER
#undef GSR_Q2_HEAD
#undef GSR_Q2_TAIL
    // ---- flush: fixed point -> float, raw moments -> the accumulator layout K6 reads, coalesced global atomics
    lds_barrier();
    float out[GSR_G2D_STRIDE];
    uint32_t gid = 0;
    bool any = false;
    if (threadIdx.x < len) {                              // seg_shift <= 8: one table row per thread
        const unsigned long long* a = acc64 + threadIdx.x * GSR_Q2_ROW;
        unsigned long long v[GSR_Q2_ROW];
#pragma unroll
        for (int q = 0; q < GSR_Q2_ROW; ++q) { v[q] = a[q]; any = any || (v[q] != 0ull); }
        any = any || poisoned;                            // non-finite input: the fixed-point sums mean nothing -> NaN out, like float arithmetic
        if (any) {
            float gxr = keep_a.x, gyr = keep_a.y, qa = keep_a.z, qb = keep_a.w, qc = keep_qc, op = keep_op;
            gid = id_first;
            if (threadIdx.x >= GSR_RB) {                  // segments of more than 64 entries: rows beyond the first round
                gid = ids[start + seg_lo + threadIdx.x];
                const SplatRec* __restrict__ g = recs + gid;
                gxr = g->x; gyr = g->y; qa = g->qa; qb = g->qb; qc = g->qc; op = g->opac;
            }
            const int eR = row_radius_exp(gxr, gyr, tcx, tcy);
            const float Sx = from_fixed(v[0], e0 - eR), Sy = from_fixed(v[1], e0 - eR);
            const float Sxx = from_fixed(v[2], e0 - 2 * eR), Sxy = from_fixed(v[3], e0 - 2 * eR), Syy = from_fixed(v[4], e0 - 2 * eR);
            const float S0 = from_fixed(v[5], e0);
            out[0] = 2.f * qa * Sx + qb * Sy;             // mean2D.x (ln2 * 0.5 W applied in K6)
            out[1] = 2.f * qc * Sy + qb * Sx;
            out[2] = -0.5f * Sxx; out[3] = -Sxy; out[4] = -0.5f * Syy;   // true conic A, B, C
            out[5] = op != 0.f ? S0 / op : 0.f;           // opacity
            out[6] = from_fixed(v[6], e0); out[7] = from_fixed(v[7], e0); out[8] = from_fixed(v[8], e0);
            out[9] = from_fixed(v[9], e0);
            out[10] = out[11] = 0.f;
            if (poisoned) {
#pragma unroll
                for (int q = 0; q < 10; ++q) out[q] = __uint_as_float(0x7fc00000u);
            }
        }
    }
    if (det64) {                                          // (uniform) GSR_VIEW_DETERMINISTIC: integer atomics on per-Gaussian scales, see above
        if (threadIdx.x < len && any) {
            const unsigned long long* a = acc64 + threadIdx.x * GSR_Q2_ROW;
            float gxr = keep_a.x, gyr = keep_a.y;
            if (threadIdx.x >= GSR_RB) { const SplatRec* __restrict__ g = recs + gid; gxr = g->x; gyr = g->y; }
            const uint4 tail = reinterpret_cast<const uint4*>(recs + gid)[3];       // id | rectx | recty | flags
            int eT, eRg;
            det_exponents(gxr, gyr, tail.y, tail.z, eT, eRg);
            const int e = det_e0(det_gmax, __uint_as_float((uint32_t)plan_total[1])) - eT;
            const int eR = row_radius_exp(gxr, gyr, tcx, tcy);
            unsigned long long* dst = det64 + ((size_t)view * vs.N + gid) * GSR_Q2_ROW;
            // the workgroup's sums as floats (its own scale), then on the Gaussian's scale: a deterministic function of this workgroup's inputs
            atomicAdd(dst + 0, to_fixed(from_fixed(a[0], e0 - eR), e - eRg));
            atomicAdd(dst + 1, to_fixed(from_fixed(a[1], e0 - eR), e - eRg));
            atomicAdd(dst + 2, to_fixed(from_fixed(a[2], e0 - 2 * eR), e - 2 * eRg));
            atomicAdd(dst + 3, to_fixed(from_fixed(a[3], e0 - 2 * eR), e - 2 * eRg));
            atomicAdd(dst + 4, to_fixed(from_fixed(a[4], e0 - 2 * eR), e - 2 * eRg));
#pragma unroll
            for (int q = 5; q < GSR_Q2_ROW; ++q) atomicAdd(dst + q, to_fixed(from_fixed(a[q], e0), e) + ((poisoned && q == 5) ? 1ull : 0ull));
        }
        return;
    }
    // pass 2 is over (barrier above): its buffer takes the converted rows [len][12] and the Gaussian indices [len] -- a region of
    // its own, so no barrier between reading the table and writing them
    float* accf = &mw[0][0][0] + 256;
    uint32_t* gids = reinterpret_cast<uint32_t*>(&mw[0][0][0]);
    if (threadIdx.x < len) {
        // the Gaussian carries a gradient: K6 takes its work list from these flags
        if (any) live[gid] = (uint8_t)1;
        gids[threadIdx.x] = any ? gid : 0xffffffffu;
#pragma unroll
        for (int q = 0; q < GSR_G2D_STRIDE; ++q) accf[threadIdx.x * GSR_G2D_STRIDE + q] = any ? out[q] : 0.f;
    }
    lds_barrier();
    // consecutive threads = consecutive slots of consecutive list positions
    for (uint32_t e = threadIdx.x; e < len * GSR_G2D_STRIDE; e += 256) {
        const float v = accf[e];
        if (v != 0.f) {
            const uint32_t r = e / GSR_G2D_STRIDE, slot = e - r * GSR_G2D_STRIDE;
            atomicAdd(g2d + (size_t)gids[r] * GSR_G2D_STRIDE + slot, v);
        }
    }
}
