// tools/valu_rates.hip -- issue rate of single VALU instructions by ENCODING on gfx950 (wave64), and the shader clock under load.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o _exp/valu_rates && _exp/valu_rates
// Each kernel: 256 CUs x `occ` workgroups of 256 threads, every wave runs ROUNDS x 64 copies of one instruction over 8
// independent register chains (inline asm: the encoding is what is written). Reported: wave-instructions per ns per SIMD and
// cycles per instruction at the clock measured inside the same launch (s_memtime ticks of wave 0 / s_memrealtime at 100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define REP8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define BODY8(S) S S S S S S S S
enum { FMAC_E32, FMA_VVV, FMA_SGPR, MUL_E32, ADD_E32, MUL_LIT, PK_FMA, EXP, RCP, CNDMASK, CMP_E32, CMP_E64, MOV, ADD_U32, FMAC_DPP, MAX_E32, CND_E64, CMP_CND, MIN_E32, MED3, BFI, AND_B32, EXECMOV, MIX_CND, MIX_CND64, MIX_CMP, MIX_MAX, MIX_FMAS, MIX_DPP, MIX_EXP, MIX_CMPCND, MIX_BFI, ASHR, NK };
const char* names[NK] = {"v_fmac_f32_e32 v,v,v (VOP2, 4 B)", "v_fma_f32 v,v,v,v (VOP3, 8 B)", "v_fma_f32 v,v,s,0.5 (VOP3, 8 B)", "v_mul_f32_e32 v,v,v (VOP2)", "v_add_f32_e32 v,v,v (VOP2)",
                         "v_mul_f32_e32 v,lit,v (VOP2 + literal, 8 B)", "v_pk_fma_f32 (VOP3P, 8 B)", "v_exp_f32_e32 (VOP1)", "v_rcp_f32_e32 (VOP1)", "v_cndmask_b32_e32 (vcc)",
                         "v_cmp_lt_f32_e32 (vcc)", "v_cmp_lt_f32_e64 (sgpr pair)", "v_mov_b32_e32", "v_add_u32_e32", "v_add_f32_dpp row_ror:8 (8 B)", "v_max_f32_e32",
    "v_cndmask_b32_e64 (sgpr pair)", "v_cmp_e32 vcc + v_cndmask vcc (pair = 2 instr)", "v_min_f32_e32", "v_med3_f32", "v_bfi_b32", "v_and_b32_e32", "s_mov exec + v_mov + s_mov exec (1 valu)",
    "mix 4 fmac + 1 cndmask vcc (5 instr)", "mix 4 fmac + 1 cndmask e64", "mix 4 fmac + 1 v_cmp_e32", "mix 4 fmac + 1 v_max", "mix 4 fmac + 1 v_fma sgpr", "mix 4 fmac + 1 dpp add", "mix 4 fmac + 1 v_exp",
    "mix 4 fmac + v_cmp + v_cndmask (6 instr)", "mix 4 fmac + 1 v_bfi", "v_ashrrev_i32_e32"};

template <int K>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* clk, int rounds) {
    float x0 = threadIdx.x * 0.001f + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float a = 1.0001f + threadIdx.x * 1e-9f, b = 0.5f;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    float sc = 1.0001f;
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sc) : "v"(a));
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
        if (K == FMAC_E32) { asm volatile(BODY8("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n")
                                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)); }
        if (K == FMA_VVV) { asm volatile(BODY8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)); }
        if (K == FMA_SGPR) { asm volatile(BODY8("v_fma_f32 %0, %0, %8, 0.5\n v_fma_f32 %1, %1, %8, 0.5\n v_fma_f32 %2, %2, %8, 0.5\n v_fma_f32 %3, %3, %8, 0.5\n v_fma_f32 %4, %4, %8, 0.5\n v_fma_f32 %5, %5, %8, 0.5\n v_fma_f32 %6, %6, %8, 0.5\n v_fma_f32 %7, %7, %8, 0.5\n")
                                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(sc)); }
        if (K == MUL_E32) { asm volatile(BODY8("v_mul_f32_e32 %0, %8, %0\n v_mul_f32_e32 %1, %8, %1\n v_mul_f32_e32 %2, %8, %2\n v_mul_f32_e32 %3, %8, %3\n v_mul_f32_e32 %4, %8, %4\n v_mul_f32_e32 %5, %8, %5\n v_mul_f32_e32 %6, %8, %6\n v_mul_f32_e32 %7, %8, %7\n")
                                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a)); }
        if (K == ADD_E32) { asm volatile(BODY8("v_add_f32_e32 %0, %8, %0\n v_add_f32_e32 %1, %8, %1\n v_add_f32_e32 %2, %8, %2\n v_add_f32_e32 %3, %8, %3\n v_add_f32_e32 %4, %8, %4\n v_add_f32_e32 %5, %8, %5\n v_add_f32_e32 %6, %8, %6\n v_add_f32_e32 %7, %8, %7\n")
                                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b)); }
        if (K == MAX_E32) { asm volatile(BODY8("v_max_f32_e32 %0, %8, %0\n v_max_f32_e32 %1, %8, %1\n v_max_f32_e32 %2, %8, %2\n v_max_f32_e32 %3, %8, %3\n v_max_f32_e32 %4, %8, %4\n v_max_f32_e32 %5, %8, %5\n v_max_f32_e32 %6, %8, %6\n v_max_f32_e32 %7, %8, %7\n")
                                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b)); }
        if (K == MUL_LIT) { asm volatile(BODY8("v_mul_f32_e32 %0, 0x3f800347, %0\n v_mul_f32_e32 %1, 0x3f800347, %1\n v_mul_f32_e32 %2, 0x3f800347, %2\n v_mul_f32_e32 %3, 0x3f800347, %3\n v_mul_f32_e32 %4, 0x3f800347, %4\n v_mul_f32_e32 %5, 0x3f800347, %5\n v_mul_f32_e32 %6, 0x3f800347, %6\n v_mul_f32_e32 %7, 0x3f800347, %7\n")
                                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (K == PK_FMA) { asm volatile(BODY8("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                                       : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb)); }
        if (K == EXP) { asm volatile(BODY8("v_exp_f32_e32 %0, %0\n v_exp_f32_e32 %1, %1\n v_exp_f32_e32 %2, %2\n v_exp_f32_e32 %3, %3\n v_exp_f32_e32 %4, %4\n v_exp_f32_e32 %5, %5\n v_exp_f32_e32 %6, %6\n v_exp_f32_e32 %7, %7\n")
                                    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (K == RCP) { asm volatile(BODY8("v_rcp_f32_e32 %0, %0\n v_rcp_f32_e32 %1, %1\n v_rcp_f32_e32 %2, %2\n v_rcp_f32_e32 %3, %3\n v_rcp_f32_e32 %4, %4\n v_rcp_f32_e32 %5, %5\n v_rcp_f32_e32 %6, %6\n v_rcp_f32_e32 %7, %7\n")
                                    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (K == CNDMASK) { asm volatile(BODY8("v_cndmask_b32_e32 %0, %8, %0, vcc\n v_cndmask_b32_e32 %1, %8, %1, vcc\n v_cndmask_b32_e32 %2, %8, %2, vcc\n v_cndmask_b32_e32 %3, %8, %3, vcc\n v_cndmask_b32_e32 %4, %8, %4, vcc\n v_cndmask_b32_e32 %5, %8, %5, vcc\n v_cndmask_b32_e32 %6, %8, %6, vcc\n v_cndmask_b32_e32 %7, %8, %7, vcc\n")
                                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b) : "vcc"); }
        if (K == CMP_E32) { asm volatile(BODY8("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cmp_lt_f32_e32 vcc, %1, %2\n v_cmp_lt_f32_e32 vcc, %2, %3\n v_cmp_lt_f32_e32 vcc, %3, %4\n v_cmp_lt_f32_e32 vcc, %4, %5\n v_cmp_lt_f32_e32 vcc, %5, %6\n v_cmp_lt_f32_e32 vcc, %6, %7\n v_cmp_lt_f32_e32 vcc, %7, %0\n")
                                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : : "vcc"); }
        if (K == CMP_E64) { asm volatile(BODY8("v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cmp_lt_f32_e64 s[22:23], %1, %2\n v_cmp_lt_f32_e64 s[24:25], %2, %3\n v_cmp_lt_f32_e64 s[26:27], %3, %4\n v_cmp_lt_f32_e64 s[20:21], %4, %5\n v_cmp_lt_f32_e64 s[22:23], %5, %6\n v_cmp_lt_f32_e64 s[24:25], %6, %7\n v_cmp_lt_f32_e64 s[26:27], %7, %0\n")
                                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"); }
        if (K == MOV) { asm volatile(BODY8("v_mov_b32_e32 %0, %1\n v_mov_b32_e32 %1, %2\n v_mov_b32_e32 %2, %3\n v_mov_b32_e32 %3, %4\n v_mov_b32_e32 %4, %5\n v_mov_b32_e32 %5, %6\n v_mov_b32_e32 %6, %7\n v_mov_b32_e32 %7, %0\n")
                                    : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (K == ADD_U32) { asm volatile(BODY8("v_add_u32_e32 %0, %8, %0\n v_add_u32_e32 %1, %8, %1\n v_add_u32_e32 %2, %8, %2\n v_add_u32_e32 %3, %8, %3\n v_add_u32_e32 %4, %8, %4\n v_add_u32_e32 %5, %8, %5\n v_add_u32_e32 %6, %8, %6\n v_add_u32_e32 %7, %8, %7\n")
                                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b)); }
        if (K == FMAC_DPP) { asm volatile(BODY8("v_add_f32_dpp %0, %1, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %3, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %5, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %7, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                                                "v_add_f32_dpp %1, %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %2, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %4, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %6, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n")
                                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }

#define F4(a0,a1,a2,a3) "v_fmac_f32_e32 %" #a0 ", %8, %9\n v_fmac_f32_e32 %" #a1 ", %8, %9\n v_fmac_f32_e32 %" #a2 ", %8, %9\n v_fmac_f32_e32 %" #a3 ", %8, %9\n"
#define OUTS : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
        if (K == CND_E64) { asm volatile(BODY8("v_cndmask_b32_e64 %0, %8, %0, s[20:21]\n v_cndmask_b32_e64 %1, %8, %1, s[20:21]\n v_cndmask_b32_e64 %2, %8, %2, s[20:21]\n v_cndmask_b32_e64 %3, %8, %3, s[20:21]\n v_cndmask_b32_e64 %4, %8, %4, s[20:21]\n v_cndmask_b32_e64 %5, %8, %5, s[20:21]\n v_cndmask_b32_e64 %6, %8, %6, s[20:21]\n v_cndmask_b32_e64 %7, %8, %7, s[20:21]\n") OUTS : "v"(b) : "s20", "s21"); }
        if (K == CMP_CND) { asm volatile(BODY8("v_cmp_lt_f32_e32 vcc, %8, %0\n v_cndmask_b32_e32 %1, %8, %1, vcc\n v_cmp_lt_f32_e32 vcc, %8, %2\n v_cndmask_b32_e32 %3, %8, %3, vcc\n v_cmp_lt_f32_e32 vcc, %8, %4\n v_cndmask_b32_e32 %5, %8, %5, vcc\n v_cmp_lt_f32_e32 vcc, %8, %6\n v_cndmask_b32_e32 %7, %8, %7, vcc\n") OUTS : "v"(b) : "vcc"); }
        if (K == MIN_E32) { asm volatile(BODY8("v_min_f32_e32 %0, %8, %0\n v_min_f32_e32 %1, %8, %1\n v_min_f32_e32 %2, %8, %2\n v_min_f32_e32 %3, %8, %3\n v_min_f32_e32 %4, %8, %4\n v_min_f32_e32 %5, %8, %5\n v_min_f32_e32 %6, %8, %6\n v_min_f32_e32 %7, %8, %7\n") OUTS : "v"(b)); }
        if (K == MED3) { asm volatile(BODY8("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n") OUTS : "v"(a), "v"(b)); }
        if (K == BFI) { asm volatile(BODY8("v_bfi_b32 %0, %8, %9, %0\n v_bfi_b32 %1, %8, %9, %1\n v_bfi_b32 %2, %8, %9, %2\n v_bfi_b32 %3, %8, %9, %3\n v_bfi_b32 %4, %8, %9, %4\n v_bfi_b32 %5, %8, %9, %5\n v_bfi_b32 %6, %8, %9, %6\n v_bfi_b32 %7, %8, %9, %7\n") OUTS : "v"(a), "v"(b)); }
        if (K == AND_B32) { asm volatile(BODY8("v_and_b32_e32 %0, %8, %0\n v_and_b32_e32 %1, %8, %1\n v_and_b32_e32 %2, %8, %2\n v_and_b32_e32 %3, %8, %3\n v_and_b32_e32 %4, %8, %4\n v_and_b32_e32 %5, %8, %5\n v_and_b32_e32 %6, %8, %6\n v_and_b32_e32 %7, %8, %7\n") OUTS : "v"(a)); }
        if (K == ASHR) { asm volatile(BODY8("v_ashrrev_i32_e32 %0, 1, %0\n v_ashrrev_i32_e32 %1, 1, %1\n v_ashrrev_i32_e32 %2, 1, %2\n v_ashrrev_i32_e32 %3, 1, %3\n v_ashrrev_i32_e32 %4, 1, %4\n v_ashrrev_i32_e32 %5, 1, %5\n v_ashrrev_i32_e32 %6, 1, %6\n v_ashrrev_i32_e32 %7, 1, %7\n") OUTS); }
        if (K == EXECMOV) { asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b32 s22, 0x55555555\n s_mov_b32 s23, 0x55555555\n" BODY8("s_mov_b64 exec, s[22:23]\n v_mov_b32_e32 %0, %8\n s_mov_b64 exec, s[20:21]\n s_mov_b64 exec, s[22:23]\n v_mov_b32_e32 %1, %8\n s_mov_b64 exec, s[20:21]\n s_mov_b64 exec, s[22:23]\n v_mov_b32_e32 %2, %8\n s_mov_b64 exec, s[20:21]\n s_mov_b64 exec, s[22:23]\n v_mov_b32_e32 %3, %8\n s_mov_b64 exec, s[20:21]\n"
                                                "s_mov_b64 exec, s[22:23]\n v_mov_b32_e32 %4, %8\n s_mov_b64 exec, s[20:21]\n s_mov_b64 exec, s[22:23]\n v_mov_b32_e32 %5, %8\n s_mov_b64 exec, s[20:21]\n s_mov_b64 exec, s[22:23]\n v_mov_b32_e32 %6, %8\n s_mov_b64 exec, s[20:21]\n s_mov_b64 exec, s[22:23]\n v_mov_b32_e32 %7, %8\n s_mov_b64 exec, s[20:21]\n") OUTS : "v"(b) : "s20", "s21", "s22", "s23"); }
        if (K == MIX_CND) { asm volatile(BODY8(F4(0,1,2,3) "v_cndmask_b32_e32 %4, %8, %4, vcc\n" F4(5,6,7,0) "v_cndmask_b32_e32 %1, %8, %1, vcc\n") OUTS : "v"(a), "v"(b) : "vcc"); }
        if (K == MIX_CND64) { asm volatile(BODY8(F4(0,1,2,3) "v_cndmask_b32_e64 %4, %8, %4, s[20:21]\n" F4(5,6,7,0) "v_cndmask_b32_e64 %1, %8, %1, s[20:21]\n") OUTS : "v"(a), "v"(b) : "s20", "s21"); }
        if (K == MIX_CMP) { asm volatile(BODY8(F4(0,1,2,3) "v_cmp_lt_f32_e32 vcc, %8, %4\n" F4(5,6,7,0) "v_cmp_lt_f32_e32 vcc, %8, %1\n") OUTS : "v"(a), "v"(b) : "vcc"); }
        if (K == MIX_MAX) { asm volatile(BODY8(F4(0,1,2,3) "v_max_f32_e32 %4, %8, %4\n" F4(5,6,7,0) "v_max_f32_e32 %1, %8, %1\n") OUTS : "v"(a), "v"(b)); }
        if (K == MIX_FMAS) { asm volatile(BODY8(F4(0,1,2,3) "v_fma_f32 %4, %4, s20, 0.5\n" F4(5,6,7,0) "v_fma_f32 %1, %1, s20, 0.5\n") OUTS : "v"(a), "v"(b) : "s20"); }
        if (K == MIX_DPP) { asm volatile(BODY8(F4(0,1,2,3) "v_add_f32_dpp %4, %5, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n" F4(5,6,7,0) "v_add_f32_dpp %1, %2, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n") OUTS : "v"(a), "v"(b)); }
        if (K == MIX_EXP) { asm volatile(BODY8(F4(0,1,2,3) "v_exp_f32_e32 %4, %4\n" F4(5,6,7,0) "v_exp_f32_e32 %1, %1\n") OUTS : "v"(a), "v"(b)); }
        if (K == MIX_CMPCND) { asm volatile(BODY8(F4(0,1,2,3) "v_cmp_lt_f32_e32 vcc, %8, %4\n v_cndmask_b32_e32 %5, %8, %5, vcc\n" F4(6,7,0,1) "v_cmp_lt_f32_e32 vcc, %8, %2\n v_cndmask_b32_e32 %3, %8, %3, vcc\n") OUTS : "v"(a), "v"(b) : "vcc"); }
        if (K == MIX_BFI) { asm volatile(BODY8(F4(0,1,2,3) "v_bfi_b32 %4, %8, %9, %4\n" F4(5,6,7,0) "v_bfi_b32 %1, %8, %9, %1\n") OUTS : "v"(a), "v"(b)); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int K> int run(float* out, unsigned long long* clk, int occ, int rounds, int per_round = 64) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<K><<<256 * occ, 256>>>(out, clk, rounds / 4);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < 3; ++i) {
        CHECK(hipEventRecord(e0));
        k<K><<<256 * occ, 256>>>(out, clk, rounds);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    unsigned long long h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double insts = (double)rounds * per_round * occ;                     // per SIMD
    const double mhz = h[1] ? (double)h[0] / ((double)h[1] / 100.0) : 0.0;   // s_memtime ticks per us of the 100 MHz wall clock
    printf("%-46s %d waves/SIMD  %8.4f ms  %6.3f instr/ns/SIMD  counter %6.0f MHz  -> %5.2f cycles/instr at 2.4 GHz, %5.2f at the counter's rate\n",
           names[K], occ, best, insts / (best * 1e6), mhz, best * 1e-3 * 2.4e9 / insts, best * 1e-3 * mhz * 1e6 / insts);
    return 0;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 4000;
    float* out; unsigned long long* clk;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * 4)); CHECK(hipMalloc(&clk, 16));
    for (int occ : {4}) {
        run<CND_E64>(out, clk, occ, rounds); run<CMP_CND>(out, clk, occ, rounds); run<MIN_E32>(out, clk, occ, rounds); run<MED3>(out, clk, occ, rounds); run<BFI>(out, clk, occ, rounds);
        run<AND_B32>(out, clk, occ, rounds); run<ASHR>(out, clk, occ, rounds); run<EXECMOV>(out, clk, occ, rounds);
        run<MIX_CND>(out, clk, occ, rounds, 80); run<MIX_CND64>(out, clk, occ, rounds, 80); run<MIX_CMP>(out, clk, occ, rounds, 80); run<MIX_MAX>(out, clk, occ, rounds, 80);
        run<MIX_FMAS>(out, clk, occ, rounds, 80); run<MIX_DPP>(out, clk, occ, rounds, 80); run<MIX_EXP>(out, clk, occ, rounds, 80); run<MIX_CMPCND>(out, clk, occ, rounds, 96); run<MIX_BFI>(out, clk, occ, rounds, 80);
    }
    for (int occ : {1, 4, 8}) {
        run<FMAC_E32>(out, clk, occ, rounds); run<FMA_VVV>(out, clk, occ, rounds); run<FMA_SGPR>(out, clk, occ, rounds); run<MUL_E32>(out, clk, occ, rounds);
        run<ADD_E32>(out, clk, occ, rounds); run<MAX_E32>(out, clk, occ, rounds); run<MUL_LIT>(out, clk, occ, rounds); run<PK_FMA>(out, clk, occ, rounds); run<EXP>(out, clk, occ, rounds);
        run<RCP>(out, clk, occ, rounds); run<CNDMASK>(out, clk, occ, rounds); run<CMP_E32>(out, clk, occ, rounds); run<CMP_E64>(out, clk, occ, rounds);
        run<MOV>(out, clk, occ, rounds); run<ADD_U32>(out, clk, occ, rounds); run<FMAC_DPP>(out, clk, occ, rounds);
    }
    return 0;
}
