// tools/valu_ceiling.hip -- what the compositing kernels' OWN instruction mix can issue per SIMD (gfx950).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize tools/valu_ceiling.hip -o _exp/valu_ceiling
//   _exp/valu_ceiling                      # wall times per (kernel, waves per SIMD)
//   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- _exp/valu_ceiling
//                                          # instruction counts of the same launches (tools/valu_ceiling.py joins the two)
//
// The round-3 argument "render_bwd / render_fwd sit on their vector-ALU floor" priced a wave64 VALU instruction at 4 cycles (from
// SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU); the guide and tools/ubench.hip say a plain v_fma_f32 issues in 2. What decides is the
// kernels' own mix (transcendentals, DPP, v_cmp + v_cndmask chains, LDS traffic between them), so this file runs exactly that
// mix with NO global memory in the loop:
//   bwd_mix   the inner loop of gsr_render_bwd_q2 (csrc/gsr_render.hip): per batch of 8 entries of four quad lists, pass 1
//             (GSR_Q2_ENTRY, lane = pixel) + pass 2 (lane = (quad, entry, half): moments, DPP half-adds, to_fixed, ds_add_u64);
//             the loop bodies are copied from the kernel, the staged records / quad lists / gradients are synthetic and LDS-resident
//   fwd_mix   the inner loop of gsr_render_fwd_serial<QUAD = true>: 8 entries per trip through GSR_COMPOSITE
// at 1, 2, 3, 4 workgroups per CU = waves per SIMD (the real kernels hold 124 / 128 VGPRs: four is their ceiling), occupancy forced
// through the dynamic-LDS request. Output: ms per launch and batches per wave; tools/valu_ceiling.py divides SQ_INSTS_VALU by the
// time -> wave-instructions per second per SIMD = cycles per instruction at the measured clock.
#include "../dreamgaussian_amd/csrc/gsr_device.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

namespace {
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float splat_power(float qa, float qb, float qc, float dx, float dy) {
    return fmaf(__fmul_rn(qa, dx), dx, fmaf(__fmul_rn(qc, dy), dy, __fmul_rn(__fmul_rn(qb, dx), dy)));
}
__device__ __forceinline__ float add_other_half(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_ROR(8), 0xf, 0xf, false));
}
__device__ __forceinline__ unsigned long long to_fixed(float v, int e) {
    const float x = ldexpf(v, e);
    const float hi = floorf(x * 2.3283064365386963e-10f);
    const float lo = fmaf(-hi, 4294967296.f, x);
    return ((unsigned long long)(uint32_t)(int32_t)hi << 32) | (unsigned long long)(uint32_t)lo;
}
__device__ __forceinline__ int row_radius_exp(float gx, float gy, float tcx, float tcy) {
    return __builtin_amdgcn_frexp_expf(fmaxf(fabsf(gx - tcx), fabsf(gy - tcy)) + 7.5f);
}
}  // namespace

#define GSR_RB 64
#define GSR_Q2_BATCH 8
#define GSR_Q2_KSTRIDE 40
#define GSR_Q2_HSTRIDE 20
#define GSR_QL_PITCH 80
#define GSR_Q2_ROW 10

// synthetic staged records of one wave: 64 splats of sigma ~ 5 px scattered over the wave's 8x8 block (+- 6 px)
__device__ __forceinline__ void fill_stage(float4* sa, float4* sb, float4* sc, int lane, float bx0, float by0) {
    const uint32_t h = (uint32_t)lane * 2654435761u;
    const float gx = bx0 + 3.5f + (float)((int)(h >> 8 & 15) - 8) * 0.8f, gy = by0 + 3.5f + (float)((int)(h >> 12 & 15) - 8) * 0.8f;
    const float sig = 3.5f + (float)(h >> 16 & 7) * 0.5f;
    const float q = -0.5f / (sig * sig) * GSR_LOG2E;
    sa[lane] = make_float4(gx, gy, q, 0.1f * q);
    sb[lane] = make_float4(q * 1.1f, 0.15f + 0.01f * (float)(h >> 20 & 31), 0.3f, 0.6f);
    sc[lane] = make_float4(0.9f, 1.5f + 0.01f * (float)lane, 0.f, 0.f);
}

// ---------------------------------------------------------------------------------------------------------------
// backward: pass 1 + pass 2 of gsr_render_bwd_q2 for `rounds` rounds of `nq` entries per quad list (nq a multiple of 8)
// ---------------------------------------------------------------------------------------------------------------
// OCC = waves per SIMD the kernel is compiled for. OCC > 4 (round 6, the review's experiment (a): what would >= 5 waves per SIMD buy?):
// the per-pixel gradients of pass 2 (`g2`, 32 registers) are read from LDS at their uses instead of held, and -- because the real
// kernel's LDS (9.6 KiB per wave: staged records, quad lists, the (m, w) exchange, the table) caps a CU at 16 waves whatever the
// registers allow -- the staging and exchange buffers are SHARED by wave pairs here: the results are wrong by construction, the
// instruction stream and the LDS traffic are the kernel's. Only the timing is used.
template <int OCC>
__global__ void __launch_bounds__(256, OCC)
bwd_mix_t(float* __restrict__ out, int rounds, int nq) {
    constexpr int NB = OCC > 4 ? 2 : 4;                   // buffer sets per workgroup
    constexpr bool G2_LDS = OCC > 4;
    __shared__ float4 stage[NB][3][GSR_RB];
    __shared__ __attribute__((aligned(8))) uint8_t qlist[4][4][GSR_QL_PITCH];
    __shared__ __attribute__((aligned(16))) float mw[NB][4][GSR_Q2_BATCH * GSR_Q2_KSTRIDE];
    __shared__ __attribute__((aligned(16))) unsigned long long acc64[GSR_Q2_ROW * 64];
    __shared__ float4 gtab[G2_LDS ? 64 : 1];           // (one wave's worth, shared: timing only)
    extern __shared__ __attribute__((aligned(16))) unsigned char pad_[];      // occupancy control only
    const int wave_true = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave_true % NB;
    const int lane = threadIdx.x & 63;
    const int row = lane >> 4, l15 = lane & 15;
    const int bx = (wave & 1) * 8, by = (wave >> 1) * 8;
    const int qx = bx + (row & 1) * 4, qy = by + (row >> 1) * 4;
    const int lx = (row & 1) * 4 + (l15 & 3), ly = (row >> 1) * 4 + (l15 >> 2);
    const float pxf = (float)(bx + lx), pyf = (float)(by + ly);
    float4* __restrict__ sa = stage[wave][0];
    float4* __restrict__ sb = stage[wave][1];
    float4* __restrict__ sc = stage[wave][2];
    fill_stage(sa, sb, sc, lane, (float)bx, (float)by);
    for (int q = threadIdx.x; q < GSR_Q2_ROW * 64; q += 256) acc64[q] = 0ull;
    for (int q = lane; q < 4 * GSR_QL_PITCH; q += 64) qlist[wave_true][q / GSR_QL_PITCH][q % GSR_QL_PITCH] = (uint8_t)((q * 7 + 3 * (q / GSR_QL_PITCH)) & 63);
    if (G2_LDS && threadIdx.x < 64) gtab[threadIdx.x] = make_float4(out[256 + (threadIdx.x & 31)] + 0.3f, out[257] + 0.5f, out[258] + 0.7f, out[259] + 0.2f + 0.001f * lane);
    __syncthreads();
    const uint8_t* __restrict__ ql = qlist[wave_true][row];
    // per-pixel values the real kernel loads: kept opaque to the compiler (the host zeroes `out`; + constants)
    const float z = out[threadIdx.x];
    const float gC0 = z + 0.3f + 0.001f * lane, gC1 = z + 0.5f, gC2 = z + 0.7f - 0.001f * lane, gD = z + 0.2f, gA = z + 0.1f;
    const float Cg_behind0 = z + 3.0f;
    const uint32_t last_contrib = (1u << 20) + (uint32_t)z;
    const int e0 = 30 + (int)z;
    const float tcx = z + 7.5f, tcy = z + 7.5f;
    float* __restrict__ mw1 = &mw[wave][row][(l15 >> 3) * GSR_Q2_HSTRIDE + (l15 & 7) * 2];
    const int k2 = l15 & 7, h2 = l15 >> 3;
    const float* __restrict__ mw2 = &mw[wave][row][k2 * GSR_Q2_KSTRIDE + h2 * GSR_Q2_HSTRIDE];
    const float qxf = (float)qx, qyf = (float)(qy + 2 * h2);
    float4 g2[G2_LDS ? 1 : 8];
    if (!G2_LDS) {
#pragma unroll
        for (int i2 = 0; i2 < (G2_LDS ? 1 : 8); ++i2) g2[i2] = make_float4(out[256 + 4 * i2] + 0.3f + 0.01f * i2, out[257 + 4 * i2] + 0.5f, out[258 + 4 * i2] + 0.7f, out[259 + 4 * i2] + 0.2f + 0.001f * lane);
    }
    const float4* __restrict__ g2l = gtab + (G2_LDS ? row * 16 + h2 * 8 : 0);
    float T = 1.f, Cgf = 0.f;

#define GSR_Q2_ENTRY(ea, eb, ec, kpos, valid, kslot)                                             \
    {                                                                                            \
        const float dx = ea.x - pxf, dy = ea.y - pyf;                                            \
        const float power = splat_power(ea.z, ea.w, eb.x, dx, dy);                               \
        const float G = fast_exp2(power);                                                        \
        const float alpha = fminf(0.99f, eb.y * G);                                              \
        const bool ok = (valid) && ((kpos) <= last_contrib) && (power <= 0.f) && (alpha >= (1.0f / 255.0f)); \
        float m = 0.f, w = 0.f;                                                                  \
        if (ok) {                                                                                \
            const float cgi = eb.z * gC0 + eb.w * gC1 + ec.x * gC2 + ec.y * gD + gA;             \
            const float oma = 1.f - alpha;                                                       \
            w = alpha * T;                                                                       \
            const float wc = w * cgi;                                                            \
            const float dL_dal = T * cgi - (Cg_behind0 - Cgf - wc) * fast_rcp(oma);               \
            m = (eb.y * dL_dal) * G;                                                             \
            Cgf += wc;                                                                           \
            T *= oma;                                                                            \
        }                                                                                        \
        if (valid) *reinterpret_cast<float2*>(mw1 + (kslot) * GSR_Q2_KSTRIDE) = make_float2(m, w); \
    }

    for (int r = 0; r < rounds; ++r) {
        const uint32_t pos0 = (uint32_t)r * 64u;
        const int nmax = nq, nmine = nq - (row & 1);          // one row a little shorter: the masked tail exists as in the kernel
        const uint32_t accrow0 = 0;
        T = 1.f; Cgf = 0.f;
        for (int jb = 0; jb < nmax; jb += GSR_Q2_BATCH) {
            const uint2 sl = *reinterpret_cast<const uint2*>(ql + jb);
            uint32_t slot[GSR_Q2_BATCH];
#pragma unroll
            for (int b = 0; b < GSR_Q2_BATCH; ++b) slot[b] = ((b < 4 ? sl.x : sl.y) >> (8 * (b & 3))) & 0xffu;
            float4 ea = sa[slot[0]], eb = sb[slot[0]], ec = sc[slot[0]];
#pragma unroll
            for (int b = 0; b < GSR_Q2_BATCH; ++b) {
                if (jb + b < nmax) {
                    float4 na = ea, nb = eb, nc = ec;
                    if (b + 1 < GSR_Q2_BATCH) { na = sa[slot[b + 1]]; nb = sb[slot[b + 1]]; nc = sc[slot[b + 1]]; }
                    GSR_Q2_ENTRY(ea, eb, ec, pos0 + slot[b] + 1u, jb + b < nmine, b)
                    ea = na; eb = nb; ec = nc;
                }
            }
            wave_lds_handoff();
            {
                const bool v2 = jb + k2 < nmine;
                const uint32_t s2 = ql[jb + k2];
                const float2 gxy = *reinterpret_cast<const float2*>(&sa[s2]);
                const float dxb = gxy.x - qxf, dyb = gxy.y - qyf;
                float S0 = 0.f, Sx = 0.f, Sy = 0.f, Sxx = 0.f, Sxy = 0.f, Syy = 0.f, W0 = 0.f, W1 = 0.f, W2 = 0.f, W3 = 0.f;
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2) {
                    const float dy = dyb - (float)r2;
                    float R0 = 0.f, R1 = 0.f, R2 = 0.f;
#pragma unroll
                    for (int c2 = 0; c2 < 4; c2 += 2) {
                        const float4 v = *reinterpret_cast<const float4*>(mw2 + (r2 * 4 + c2) * 2);
                        const float4 g0 = G2_LDS ? g2l[r2 * 4 + c2] : g2[G2_LDS ? 0 : r2 * 4 + c2], g1 = G2_LDS ? g2l[r2 * 4 + c2 + 1] : g2[G2_LDS ? 0 : r2 * 4 + c2 + 1];
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
                S0 = add_other_half(S0); Sx = add_other_half(Sx); Sy = add_other_half(Sy);
                Sxx = add_other_half(Sxx); Sxy = add_other_half(Sxy); Syy = add_other_half(Syy);
                W0 = add_other_half(W0); W1 = add_other_half(W1); W2 = add_other_half(W2); W3 = add_other_half(W3);
                if (v2) {
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
            wave_lds_handoff();
        }
    }
#undef GSR_Q2_ENTRY
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = T + Cgf + (float)(acc64[threadIdx.x] & 0xffffu);
}

// ---------------------------------------------------------------------------------------------------------------
// forward: the per-quad-list compositing loop of gsr_render_fwd_serial<true> (GSR_COMPOSITE, two entries per trip)
// ---------------------------------------------------------------------------------------------------------------
#define GSR_COMPOSITE(ea, eb, ec, kpos, valid, gate, keepT)                                     \
    {                                                                                          \
        const float dx = ea.x - pxf, dy = ea.y - pyf;                                          \
        const float power = splat_power(ea.z, ea.w, eb.x, dx, dy);                             \
        const float alpha = fminf(0.99f, eb.y * fast_exp2(power));                             \
        const bool ok = (valid) && !done && (power <= 0.f) && (alpha >= (1.0f / 255.0f));      \
        const float test_T = __fmul_rn(T, 1.f - alpha);                                        \
        const bool stop = ok && (__fmul_rn(gate, test_T) < 0.0001f);                           \
        const bool acc = ok && !stop;                                                          \
        const float w = acc ? alpha * T : 0.f;                                                 \
        C0 = fmaf(eb.z, w, C0); C1 = fmaf(eb.w, w, C1); C2 = fmaf(ec.x, w, C2);                \
        D = fmaf(ec.y, w, D); A += w;                                                          \
        T = (keepT ? acc : ok) ? test_T : T;                                                   \
        last = acc ? (kpos) : last;                                                            \
        done = done || stop;                                                                   \
    }

__global__ void __launch_bounds__(256)
fwd_mix(float* __restrict__ out, int rounds, int nq) {
    __shared__ float4 stage[4][3][GSR_RB + 2];
    __shared__ __attribute__((aligned(8))) uint8_t qlist[4][4][80];
    extern __shared__ __attribute__((aligned(16))) unsigned char pad_[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int row = lane >> 4, l15 = lane & 15;
    const int bx = (wave & 1) * 8, by = (wave >> 1) * 8;
    const int lx = (row & 1) * 4 + (l15 & 3), ly = (row >> 1) * 4 + (l15 >> 2);
    const float pxf = (float)(bx + lx), pyf = (float)(by + ly);
    float4* __restrict__ sa = stage[wave][0];
    float4* __restrict__ sb = stage[wave][1];
    float4* __restrict__ sc = stage[wave][2];
    fill_stage(sa, sb, sc, lane, (float)bx, (float)by);
    for (int q = lane; q < 4 * 80; q += 64) qlist[wave][q / 80][q % 80] = (uint8_t)((q * 7 + 3 * (q / 80)) & 63);
    __syncthreads();
    const uint8_t* __restrict__ ql = qlist[wave][row];
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, A = 0.f;
    uint32_t last = 0;
    bool done = false;
    for (int r = 0; r < rounds; ++r) {
        const uint32_t pos1 = (uint32_t)r * 64u + 1u;
        const int nmax = nq, nmine = nq - (row & 1);
        if ((r & 15) == 0) { T = 1.f; done = false; }           // keep the pixels alive: the mix of a live wave is what is measured
        for (int jb = 0; jb < nmax; jb += 8) {
            const uint2 sl = *reinterpret_cast<const uint2*>(ql + jb);
            uint32_t slot[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) slot[b] = ((b < 4 ? sl.x : sl.y) >> (8 * (b & 3))) & 0xffu;
            float4 e0a = sa[slot[0]], e0b = sb[slot[0]], e0c = sc[slot[0]];
#pragma unroll
            for (int b = 0; b < 8; b += 2) {
                if (jb + b < nmax) {
                    const float4 e1a = sa[slot[b + 1]], e1b = sb[slot[b + 1]], e1c = sc[slot[b + 1]];
                    GSR_COMPOSITE(e0a, e0b, e0c, pos1 + slot[b], jb + b < nmine, 1.f, true)
                    if (b + 2 < 8) { e0a = sa[slot[b + 2]]; e0b = sb[slot[b + 2]]; e0c = sc[slot[b + 2]]; }
                    GSR_COMPOSITE(e1a, e1b, e1c, pos1 + slot[b + 1], jb + b + 1 < nmine, 1.f, true)
                }
            }
        }
        wave_lds_handoff();
    }
    out[blockIdx.x * 256 + threadIdx.x] = T + C0 + C1 + C2 + D + A + (float)last;
}

// plain v_fma_f32, 8 independent chains: the 2-cycle reference point on the same launch geometry
__global__ void __launch_bounds__(256)
fma_ref(float* __restrict__ out, int rounds, int) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pad_[];
    float x0 = threadIdx.x * 0.001f + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const float c = 1.0001f, d = 0.5f;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { x0 = fmaf(x0, c, d); x1 = fmaf(x1, c, d); x2 = fmaf(x2, c, d); x3 = fmaf(x3, c, d); x4 = fmaf(x4, c, d); x5 = fmaf(x5, c, d); x6 = fmaf(x6, c, d); x7 = fmaf(x7, c, d); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <typename K>
int run(const char* name, K kern, size_t static_lds, int occ, int rounds, int nq, float* out) {
    // exactly `occ` workgroups per CU: request so much dynamic LDS that occ + 1 do not fit the CU's 160 KiB
    const size_t cu = 160 * 1024;
    size_t want = cu / (size_t)occ;                       // per-workgroup budget
    size_t dyn = want > static_lds + 1024 ? want - static_lds - 1024 : 0;
    dyn &= ~(size_t)255;
    if (static_lds + dyn > 32 * 1024) CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * occ;
    CHECK(hipMemset(out, 0, 256 * 8 * 256 * 4));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), dyn, 0, out, rounds / 8 + 1, nq);   // warm-up
    CHECK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0.f;
    const int reps = 5;
    for (int i = 0; i < reps; ++i) {
        CHECK(hipMemset(out, 0, 256 * 8 * 256 * 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), dyn, 0, out, rounds, nq);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    // batches of 8 entries (x 4 quad lists) per wave and launch
    const double batches = (double)rounds * (nq / 8);
    printf("{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"grid\": %d, \"rounds\": %d, \"nq\": %d, \"ms_min\": %.4f, \"ms_avg\": %.4f, \"batches_per_wave\": %.0f, "
           "\"ns_per_batch_per_simd\": %.2f}\n", name, occ, grid, rounds, nq, best, sum / reps, batches, best * 1e6 / (batches * occ));
    return 0;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 400;
    const int nq = argc > 2 ? atoi(argv[2]) : 24;
    float* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    const size_t lds_bwd = 4 * 3 * 64 * 16 + 4 * 4 * 80 + 4 * 4 * 8 * 40 * 4 + 10 * 64 * 8;
    const size_t lds_fwd = 4 * 3 * 66 * 16 + 4 * 4 * 80;
    for (int occ = 1; occ <= 4; ++occ) if (run("bwd_mix", bwd_mix_t<4>, lds_bwd, occ, rounds, nq, out)) return 1;
    // experiment (a): the same loops compiled for 5 / 6 waves per SIMD (<= 102 / <= 84 VGPRs, g2 from LDS, shared buffers: timing only)
    const size_t lds_bwd2 = 2 * 3 * 64 * 16 + 4 * 4 * 80 + 2 * 4 * 8 * 40 * 4 + 10 * 64 * 8 + 64 * 16;
    for (int occ = 4; occ <= 5; ++occ) if (run("bwd_mix_occ5", bwd_mix_t<5>, lds_bwd2, occ, rounds, nq, out)) return 1;
    for (int occ = 4; occ <= 6; ++occ) if (run("bwd_mix_occ6", bwd_mix_t<6>, lds_bwd2, occ, rounds, nq, out)) return 1;
    for (int occ = 1; occ <= 4; ++occ) if (run("fwd_mix", fwd_mix, lds_fwd, occ, rounds, nq, out)) return 1;
    for (int occ = 1; occ <= 4; ++occ) if (run("fma_ref", fma_ref, 0, occ, rounds * 8, nq, out)) return 1;
    // the forward has 128 VGPRs at most too, but LDS would let 8 workgroups share a CU: the mix at 8 waves per SIMD if registers allowed
    for (int occ : {6, 8}) if (run("fma_ref", fma_ref, 0, occ, rounds * 8, nq, out)) return 1;
    return 0;
}
