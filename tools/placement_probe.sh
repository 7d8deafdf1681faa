#!/bin/bash
# Builds _exp/libgsr_place.so: the product sources + a probe in gsr_render_fwd_serial<QUAD> and gsr_render_bwd_q2 that records, per
# wave / per workgroup, where it ran (XCC_ID, HW_ID: SE / SH / CU / SIMD), when it started and ended (100 MHz wall clock) and how long
# its list was. Read back through gsr_debug_placement (exported by the variant only). Never part of the product library.
#   bash tools/placement_probe.sh && GSR_LIB=$PWD/_exp/libgsr_place.so python tools/placement_report.py     (on the GPU box)
set -e
cd "$(dirname "$0")/.."
tools/build_variant.sh place <<'PATCH'
python - <<'PY'
p = 'dreamgaussian_amd/csrc/gsr_render.hip'
s = open(p).read()
# ---- forward: one row per wave
s = s.replace('''namespace {
__device__ __forceinline__ float fast_exp2''', '''#define GSR_DBG_ROWS 16384
__device__ unsigned long long g_dbg_fwd[GSR_DBG_ROWS * 4];
__device__ unsigned long long g_dbg_bwd[65536 * 4];
__device__ __forceinline__ unsigned long long dbg_where() {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    return (unsigned long long)hw | ((unsigned long long)(xcc & 15u) << 32);
}
namespace {
__device__ __forceinline__ float fast_exp2''', 1)
a = s.index('gsr_render_fwd_serial(const uint32_t* __restrict__ tile_off')
b = s.index('template __global__ void gsr_render_fwd_serial<false>')
k = s[a:b]
k = k.replace('''    const int tg = (int)order[blockIdx.x];                // heaviest tiles first''', '''    const unsigned long long dbg_t0 = wall_clock64();
    const int tg = (int)order[blockIdx.x];                // heaviest tiles first''', 1)
assert 'dbg_t0 = wall_clock64' in k
k = k.replace('''    // ---- how deep the backward has to walk this tile's list, and its (tile, segment) work items''', '''    if (lane == 0 && blockIdx.x * 4 + wave < GSR_DBG_ROWS) {
        unsigned long long* d = g_dbg_fwd + (size_t)(blockIdx.x * 4 + wave) * 4;
        d[0] = dbg_t0; d[1] = wall_clock64(); d[2] = dbg_where(); d[3] = (unsigned long long)n | ((unsigned long long)wave_max_u32(inside ? last : 0u) << 32);
    }
    // ---- how deep the backward has to walk this tile's list, and its (tile, segment) work items''', 1)
s = s[:a] + k + s[b:]
# ---- the pair forward (round 6): one row per BLENDER wave (the block's chain), same table
a = s.index('gsr_render_fwd_pair(const uint32_t* __restrict__ tile_off')
b = s.index('// The exact walk of list positions [lo, hi) of a tile')
k = s[a:b]
k = k.replace('''    const int tg = (int)order[blockIdx.x];                // heaviest tiles first''', '''    const unsigned long long dbg_t0 = wall_clock64();
    const int tg = (int)order[blockIdx.x];                // heaviest tiles first''', 1)
assert 'dbg_t0 = wall_clock64' in k
k = k.replace('''    // ---- how deep the backward has to walk this tile's list, and its (tile, segment) work items''', '''    if (!tester && lane == 0 && blockIdx.x * 4 + blk < GSR_DBG_ROWS) {
        unsigned long long* d = g_dbg_fwd + (size_t)(blockIdx.x * 4 + blk) * 4;
        d[0] = dbg_t0; d[1] = wall_clock64(); d[2] = dbg_where(); d[3] = (unsigned long long)n | ((unsigned long long)wave_max_u32(inside ? last : 0u) << 32);
    }
    // ---- how deep the backward has to walk this tile's list, and its (tile, segment) work items''', 1)
s = s[:a] + k + s[b:]
# ---- backward: one row per workgroup (item)
s = s.replace('''    clear_rows();                                         // (in front of the set-up loads: 2.5 us better than behind the flush, same box)
    if (blockIdx.x >= nitems) return;''', '''    const unsigned long long dbg_t0 = wall_clock64();
    clear_rows();                                         // (in front of the set-up loads: 2.5 us better than behind the flush, same box)
    if (blockIdx.x >= nitems) return;''', 1)
s = s.replace('''    // ---- flush: fixed point -> float, raw moments -> the accumulator layout K6 reads, coalesced global atomics
    lds_barrier();''', '''    if (threadIdx.x == 0 && blockIdx.x < 65536) {
        unsigned long long* d = g_dbg_bwd + (size_t)blockIdx.x * 4;
        d[0] = dbg_t0; d[1] = wall_clock64(); d[2] = dbg_where(); d[3] = (unsigned long long)len | ((unsigned long long)tg << 32);
    }
    // ---- flush: fixed point -> float, raw moments -> the accumulator layout K6 reads, coalesced global atomics
    lds_barrier();''', 1)
open(p, 'w').write(s)
p = 'dreamgaussian_amd/csrc/gsr_api.hip'
s = open(p).read()
s += '''
extern "C" int gsr_debug_placement(int which, unsigned long long* host_dst, size_t n_u64) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    hipError_t e = which == 0 ? hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_dbg_fwd), n_u64 * 8) : hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_dbg_bwd), n_u64 * 8);
    return e == hipSuccess ? 0 : -2;
}
'''
open(p, 'w').write(s)
PY
PATCH
