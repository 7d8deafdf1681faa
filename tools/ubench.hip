// tools/ubench.hip -- instruction-cost microbenchmarks for the cost model of the compositing kernels (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o gpurun_out/ubench && gpurun_out/ubench
// Every kernel: grid = 256 CUs x BPC workgroups of 256 threads (BPC = 4 -> 16 waves / CU = 4 waves / SIMD), each
// wave runs ITERS trips of U independent copies of one instruction; reported = cycles per wave-instruction per CU
// pipe (LDS: per CU; VALU: per SIMD), from the wall time at the measured clock (s_memtime delta of one wave).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define AS3 __attribute__((address_space(3)))
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2000;

enum Pat { RD128_BCAST, RD128_ROW4, RD128_SEQ, RD64_SEQ, RD32_SEQ, RD128_ROWSTRIDE, WR64_SEQ, WR128_SEQ, WR32_SEQ, WR8_SEQ,
           ADD_SEQ, ADD_12LANES_4SAME, ADD_PAIRS, ADD_SAME64, RD_U8, BPERMUTE,
           ADDU32_SEQ, ADDU64_SEQ, ADDU32_PAIRS, ADDU64_4SAME, ADDF_16LANES, ADDF_32LANES, RMW_SEQ, V_FMA, V_PKFMA, V_EXP, V_RCP, V_DPPADD, V_PERMLANE32SWAP, V_CNDMASK, V_CMP, V_READLANE, V_MUL, V_FMAC_DEP, NPAT };
const char* names[NPAT] = {"ds_read_b128 broadcast (1 addr)", "ds_read_b128 4 row addrs", "ds_read_b128 lane-consecutive", "ds_read_b64 lane-consecutive",
    "ds_read_b32 lane-consecutive", "ds_read_b128 stride 144B/lane", "ds_write_b64 lane-consecutive", "ds_write_b128 lane-consecutive", "ds_write_b32 lane-consecutive", "ds_write_b8 lane-consecutive",
    "ds_add_f32 64 distinct consecutive", "ds_add_f32 12 lanes, 4 rows same addr (f2b)", "ds_add_f32 64 lanes, pairs same addr", "ds_add_f32 64 lanes one addr", "ds_read_u8 row addrs", "ds_bpermute_b32",
    "ds_add_u32 64 distinct consecutive", "ds_add_u64 64 distinct consecutive", "ds_add_u32 64 lanes, pairs same addr", "ds_add_u64 64 lanes, 4 lanes same addr", "ds_add_f32 16 active lanes distinct", "ds_add_f32 32 active lanes distinct", "ds_read_b32 + v_add + ds_write_b32 (non-atomic RMW)",
    "v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_rcp_f32", "v_add_f32 dpp row_ror", "v_permlane32_swap", "v_cndmask_b32", "v_cmp_lt_f32", "v_readlane_b32", "v_mul_f32", "v_fmac_f32 dependent chain"};

template <int P>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* clk, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x0 = lane * 0.001f + 1.0f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    const int base = wave * 2048;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const int o = (it & 7) * 4;
        if (P == RD128_BCAST) {
            AS3 const f4* p = (AS3 const f4*)(lds + base + o * 4);
            a0 = *(volatile AS3 const f4*)(p + 0); a1 = *(volatile AS3 const f4*)(p + 1); a2 = *(volatile AS3 const f4*)(p + 2); a3 = *(volatile AS3 const f4*)(p + 3);
        } else if (P == RD128_ROW4) {
            AS3 const f4* p = (AS3 const f4*)(lds + base + ((lane >> 4) * 37 + o) * 4);
            a0 = *(volatile AS3 const f4*)(p + 0); a1 = *(volatile AS3 const f4*)(p + 64); a2 = *(volatile AS3 const f4*)(p + 128); a3 = *(volatile AS3 const f4*)(p + 192);
        } else if (P == RD128_SEQ) {
            AS3 const f4* p = (AS3 const f4*)(lds + base + lane * 4 + o * 4);
            a0 = *(volatile AS3 const f4*)(p + 0); a1 = *(volatile AS3 const f4*)(p + 64); a2 = *(volatile AS3 const f4*)(p + 128); a3 = *(volatile AS3 const f4*)(p + 192);
        } else if (P == RD128_ROWSTRIDE) {
            AS3 const f4* p = (AS3 const f4*)(lds + base + lane * 36 + o * 4);
            a0 = *(volatile AS3 const f4*)(p + 0); a1 = *(volatile AS3 const f4*)(p + 1); a2 = *(volatile AS3 const f4*)(p + 2); a3 = *(volatile AS3 const f4*)(p + 3);
        } else if (P == RD64_SEQ) {
            AS3 const f2* p = (AS3 const f2*)(lds + base + lane * 2 + o * 4);
            f2 u0 = *(volatile AS3 const f2*)(p + 0), u1 = *(volatile AS3 const f2*)(p + 64), u2 = *(volatile AS3 const f2*)(p + 128), u3 = *(volatile AS3 const f2*)(p + 192);
            a0.x += u0.x; a1.x += u1.x; a2.x += u2.x; a3.x += u3.x;
        } else if (P == RD32_SEQ) {
            AS3 const float* p = (AS3 const float*)(lds + base + lane + o * 4);
            a0.x += *(volatile AS3 const float*)(p + 0); a1.x += *(volatile AS3 const float*)(p + 64); a2.x += *(volatile AS3 const float*)(p + 128); a3.x += *(volatile AS3 const float*)(p + 192);
        } else if (P == WR64_SEQ) {
            AS3 f2* p = (AS3 f2*)(lds + base + lane * 2 + o * 4);
            *(volatile AS3 f2*)(p + 0) = f2{x0, x1}; *(volatile AS3 f2*)(p + 64) = f2{x2, x3}; *(volatile AS3 f2*)(p + 128) = f2{x4, x5}; *(volatile AS3 f2*)(p + 192) = f2{x6, x7};
        } else if (P == WR128_SEQ) {
            AS3 f4* p = (AS3 f4*)(lds + base + lane * 4 + o * 4);
            *(volatile AS3 f4*)(p + 0) = f4{x0, x1, x2, x3}; *(volatile AS3 f4*)(p + 64) = f4{x2, x3, x4, x5}; *(volatile AS3 f4*)(p + 128) = f4{x4, x5, x6, x7}; *(volatile AS3 f4*)(p + 192) = f4{x6, x7, x0, x1};
        } else if (P == WR32_SEQ) {
            AS3 float* p = (AS3 float*)(lds + base + lane + o * 4);
            *(volatile AS3 float*)(p + 0) = x0; *(volatile AS3 float*)(p + 64) = x1; *(volatile AS3 float*)(p + 128) = x2; *(volatile AS3 float*)(p + 192) = x3;
        } else if (P == WR8_SEQ) {
            AS3 uint8_t* p = (AS3 uint8_t*)(lds + base) + lane + o * 16;
            *(volatile AS3 uint8_t*)(p + 0) = (uint8_t)lane; *(volatile AS3 uint8_t*)(p + 64) = (uint8_t)lane; *(volatile AS3 uint8_t*)(p + 128) = (uint8_t)lane; *(volatile AS3 uint8_t*)(p + 192) = (uint8_t)lane;
        } else if (P == ADD_SEQ) {
            float* p = lds + base + lane + o * 4;
            atomicAdd(p + 0, x0); atomicAdd(p + 64, x1); atomicAdd(p + 128, x2); atomicAdd(p + 192, x3);
        } else if (P == ADD_12LANES_4SAME) {
            if ((lane & 15) < 3) { float* p = lds + base + (lane & 15) + o * 4; atomicAdd(p + 0, x0); atomicAdd(p + 12, x1); atomicAdd(p + 24, x2); atomicAdd(p + 36, x3); }
        } else if (P == ADD_PAIRS) {
            float* p = lds + base + (lane >> 1) + o * 4;
            atomicAdd(p + 0, x0); atomicAdd(p + 64, x1); atomicAdd(p + 128, x2); atomicAdd(p + 192, x3);
        } else if (P == ADD_SAME64) {
            float* p = lds + base + o * 4;
            atomicAdd(p + 0, x0); atomicAdd(p + 1, x1); atomicAdd(p + 2, x2); atomicAdd(p + 3, x3);
        } else if (P == RD_U8) {
            AS3 const uint8_t* p = (AS3 const uint8_t*)(lds + base) + (lane >> 4) * 80 + o;
            a0.x += *(volatile AS3 const uint8_t*)(p + 0); a1.x += *(volatile AS3 const uint8_t*)(p + 1); a2.x += *(volatile AS3 const uint8_t*)(p + 2); a3.x += *(volatile AS3 const uint8_t*)(p + 3);
        } else if (P == BPERMUTE) {
            const int idx = ((lane * 7 + it) & 63) * 4;
            x0 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(x0))); x1 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(x1)));
            x2 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(x2))); x3 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(x3)));
        } else if (P == ADDU32_SEQ) {
            unsigned* p = (unsigned*)(lds + base + lane + o * 4);
            atomicAdd(p + 0, (unsigned)lane); atomicAdd(p + 64, (unsigned)lane); atomicAdd(p + 128, (unsigned)lane); atomicAdd(p + 192, (unsigned)lane);
        } else if (P == ADDU64_SEQ) {
            unsigned long long* p = (unsigned long long*)(lds + base) + lane + o * 2;
            atomicAdd(p + 0, (unsigned long long)lane); atomicAdd(p + 64, (unsigned long long)lane); atomicAdd(p + 128, (unsigned long long)lane); atomicAdd(p + 192, (unsigned long long)lane);
        } else if (P == ADDU32_PAIRS) {
            unsigned* p = (unsigned*)(lds + base + (lane >> 1) + o * 4);
            atomicAdd(p + 0, (unsigned)lane); atomicAdd(p + 64, (unsigned)lane); atomicAdd(p + 128, (unsigned)lane); atomicAdd(p + 192, (unsigned)lane);
        } else if (P == ADDU64_4SAME) {
            unsigned long long* p = (unsigned long long*)(lds + base) + (lane & 15) + o * 2;
            atomicAdd(p + 0, (unsigned long long)lane); atomicAdd(p + 64, (unsigned long long)lane); atomicAdd(p + 128, (unsigned long long)lane); atomicAdd(p + 192, (unsigned long long)lane);
        } else if (P == ADDF_16LANES) {
            if ((lane & 3) == 0) { float* p = lds + base + lane + o * 4; atomicAdd(p + 0, x0); atomicAdd(p + 64, x1); atomicAdd(p + 128, x2); atomicAdd(p + 192, x3); }
        } else if (P == ADDF_32LANES) {
            if ((lane & 1) == 0) { float* p = lds + base + lane + o * 4; atomicAdd(p + 0, x0); atomicAdd(p + 64, x1); atomicAdd(p + 128, x2); atomicAdd(p + 192, x3); }
        } else if (P == RMW_SEQ) {
            AS3 float* p = (AS3 float*)(lds + base + lane + o * 4);
            float r0 = *(volatile AS3 float*)(p + 0), r1 = *(volatile AS3 float*)(p + 64), r2 = *(volatile AS3 float*)(p + 128), r3 = *(volatile AS3 float*)(p + 192);
            *(volatile AS3 float*)(p + 0) = r0 + x0; *(volatile AS3 float*)(p + 64) = r1 + x1; *(volatile AS3 float*)(p + 128) = r2 + x2; *(volatile AS3 float*)(p + 192) = r3 + x3;
        } else if (P == V_FMA) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { x0 = __builtin_fmaf(x0, 1.0001f, 0.5f); x1 = __builtin_fmaf(x1, 1.0001f, 0.5f); x2 = __builtin_fmaf(x2, 1.0001f, 0.5f); x3 = __builtin_fmaf(x3, 1.0001f, 0.5f);
                                         x4 = __builtin_fmaf(x4, 1.0001f, 0.5f); x5 = __builtin_fmaf(x5, 1.0001f, 0.5f); x6 = __builtin_fmaf(x6, 1.0001f, 0.5f); x7 = __builtin_fmaf(x7, 1.0001f, 0.5f); }
        } else if (P == V_MUL) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { x0 *= 1.0001f; x1 *= 1.0001f; x2 *= 1.0001f; x3 *= 1.0001f; x4 *= 1.0001f; x5 *= 1.0001f; x6 *= 1.0001f; x7 *= 1.0001f; }
        } else if (P == V_FMAC_DEP) {
#pragma unroll
            for (int u = 0; u < 32; ++u) x0 = __builtin_fmaf(x0, 1.0001f, 0.5f);
        } else if (P == V_PKFMA) {
            const v2 c = {1.0001f, 1.0001f}, d = {0.5f, 0.5f};
#pragma unroll
            for (int u = 0; u < 8; ++u) { p0 = p0 * c + d; p1 = p1 * c + d; p2 = p2 * c + d; p3 = p3 * c + d; }
        } else if (P == V_EXP) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); x2 = __builtin_amdgcn_exp2f(x2); x3 = __builtin_amdgcn_exp2f(x3);
                                         x4 = __builtin_amdgcn_exp2f(x4); x5 = __builtin_amdgcn_exp2f(x5); x6 = __builtin_amdgcn_exp2f(x6); x7 = __builtin_amdgcn_exp2f(x7); }
        } else if (P == V_RCP) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
                                         x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7); }
        } else if (P == V_DPPADD) {
#define DPPA(v) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false))
#pragma unroll
            for (int u = 0; u < 4; ++u) { DPPA(x0); DPPA(x1); DPPA(x2); DPPA(x3); DPPA(x4); DPPA(x5); DPPA(x6); DPPA(x7); }
        } else if (P == V_PERMLANE32SWAP) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                auto r0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x0), __float_as_uint(x1), false, false); x0 = __uint_as_float(r0[0]); x1 = __uint_as_float(r0[1]);
                auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x2), __float_as_uint(x3), false, false); x2 = __uint_as_float(r1[0]); x3 = __uint_as_float(r1[1]);
                auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x4), __float_as_uint(x5), false, false); x4 = __uint_as_float(r2[0]); x5 = __uint_as_float(r2[1]);
                auto r3 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x6), __float_as_uint(x7), false, false); x6 = __uint_as_float(r3[0]); x7 = __uint_as_float(r3[1]);
            }
        } else if (P == V_CNDMASK) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { const bool c = (lane + it + u) & 1; x0 = c ? x1 : x0; x2 = c ? x3 : x2; x4 = c ? x5 : x4; x6 = c ? x7 : x6; x1 = c ? x2 : x1; x3 = c ? x4 : x3; x5 = c ? x6 : x5; x7 = c ? x0 : x7; }
        } else if (P == V_CMP) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { unsigned long long m = __ballot(x0 < x1 + u) ^ __ballot(x2 < x3 + u) ^ __ballot(x4 < x5 + u) ^ __ballot(x6 < x7 + u); x0 += (float)(m & 1); }
        } else if (P == V_READLANE) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int l = (it + u) & 63; x0 += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x1), l)); x2 += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x3), l));
                                         x4 += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x5), l)); x6 += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x7), l)); }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0.x + a1.y + a2.z + a3.w + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + lds[(lane * 5) & 8191];
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int P> int run(float* out, unsigned long long* clk, int bpc, int insts_per_iter, bool valu) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<P><<<256 * bpc, 256>>>(out, clk, 200);   // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<P><<<256 * bpc, 256>>>(out, clk, ITERS);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h = 0; CHECK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));
    const double mhz = (double)h / (ms * 1e-3) / 1e6;                  // s_memtime ticks of one wave / wall time
    const double waves_per_pipe = valu ? bpc : 4.0 * bpc;                // VALU: per SIMD; LDS: per CU
    const double cyc = (double)h / ((double)ITERS * insts_per_iter * waves_per_pipe);
    const double wall_cyc = ms * 1e-3 * 2.2e9 / ((double)ITERS * insts_per_iter * waves_per_pipe);
    printf("%-52s bpc %d  %8.3f ms  -> %7.2f %s-cycles per wave-instruction (wall clock at 2.2 GHz; wave-0 counter: %.2f, %.0f MHz)\n", names[P], bpc, ms, wall_cyc, valu ? "SIMD" : "LDS", cyc, mhz);
    return 0;
}

int main() {
    float* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * 4)); CHECK(hipMalloc(&clk, 8));
    for (int bpc : {4}) {
        run<ADD_SEQ>(out, clk, bpc, 4, false); run<ADDF_32LANES>(out, clk, bpc, 4, false); run<ADDF_16LANES>(out, clk, bpc, 4, false); run<ADD_12LANES_4SAME>(out, clk, bpc, 4, false);
        run<ADDU32_SEQ>(out, clk, bpc, 4, false); run<ADDU64_SEQ>(out, clk, bpc, 4, false); run<ADDU32_PAIRS>(out, clk, bpc, 4, false); run<ADDU64_4SAME>(out, clk, bpc, 4, false);
        run<RMW_SEQ>(out, clk, bpc, 8, false); run<WR32_SEQ>(out, clk, bpc, 4, false); run<RD32_SEQ>(out, clk, bpc, 4, false);
    }
    return 0;
}
