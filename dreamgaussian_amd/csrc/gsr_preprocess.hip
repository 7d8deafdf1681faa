// gsr_preprocess.hip -- per-Gaussian stage, forward (K1) and backward (K6).
//
// Replaces (behaviour, not code) the preprocess half of the external rasterizer that
// gs_renderer.py:800-809 calls; in-tree Python twins of the math: gs_renderer.py:85-132
// (rotation/covariance), sh_utils.py:57-112 + gs_renderer.py:793 (SH -> RGB),
// gs_renderer.py:629-671 (projection conventions). Spec: SURVEY.md Appendix A.3 / A.7.
//
// Both kernels stream the Gaussians from HBM. K1: three phases per wave of 64 Gaussians without a workgroup barrier --
// lane = Gaussian (frustum test, view direction, SH basis), lane = SH coefficient for the loads (16 lanes share a Gaussian: the SH
// rows are read straight from HBM, 768 contiguous bytes per wave instruction, pass through the wave's LDS slice and are evaluated by the lane that owns the Gaussian),
// lane = Gaussian again (projection, conic, radius, exact per-tile support test); per-tile instance counts privatised in an
// LDS histogram and flushed once per workgroup. K6: lane = Gaussian, SH in / dSH out staged through LDS for K > 1 (row pitch
// 3K+1 dwords: odd => conflict-free per-lane row walks); with nothing to stage (K == 1) one launch runs through all the
// cameras of a batch in registers.
#include "gsr_device.h"

namespace {

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f,
                SH_C2_2 = 0.31539156525252005f, SH_C2_3 = -1.0925484305920792f,
                SH_C2_4 = 0.5462742152960396f;
constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f,
                SH_C3_2 = -0.4570457994644658f, SH_C3_3 = 0.3731763325901154f,
                SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                SH_C3_6 = -0.5900435899266435f;

// View-space depth in ONE pinned fp32 evaluation order (no fma contraction): it is the sort
// key of the compositing order, so the oracle evaluates exactly this chain.
// (HIP's __fmul_rn/__fadd_rn are plain * and + and DO contract to v_fma under hipcc's default
// -ffp-contract=fast-honor-pragmas; the pragma is what pins the order.)
__device__ __forceinline__ float view_depth(const float* __restrict__ V, float x, float y, float z) {
#pragma clang fp contract(off)
    return ((V[2] * x + V[6] * y) + V[10] * z) + V[14];
}

// DreamGaussian's parameter activations (gs_renderer.py:134-142), fused when ViewConst.raw_act
__device__ __forceinline__ float act_sigmoid(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float4 act_normalize(float4 q, float* inv_norm) {
    const float inv = 1.f / sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    *inv_norm = inv;
    return make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
}

struct Cov3 { float c0, c1, c2, c3, c4, c5; };   // S00 S01 S02 S11 S12 S22

__device__ __forceinline__ void quat_to_R(const float4 q, float R[9]) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;   // (r,x,y,z), gs_renderer.py:92-105
    R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = (R diag(s)) (R diag(s))^T
__device__ __forceinline__ Cov3 cov3d_from_scale_rot(const float3 s, const float R[9]) {
    float M[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) { M[3 * i] = R[3 * i] * s.x; M[3 * i + 1] = R[3 * i + 1] * s.y; M[3 * i + 2] = R[3 * i + 2] * s.z; }
    Cov3 c;
    c.c0 = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
    c.c1 = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    c.c2 = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
    c.c3 = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    c.c4 = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
    c.c5 = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
    return c;
}

struct Proj2D {
    float tx, ty, tz;       // view-space mean with the 1.3*tanfov clamp applied to x,y
    bool clampx, clampy;
    float T0[3], T1[3];     // rows of J * Wr
    float a, b, c;          // 2D covariance incl. the +0.3 dilation
};

__device__ __forceinline__ Proj2D project_cov(const ViewConst& vc, const float* __restrict__ V,
                                              float3 pv, const Cov3& S) {
    Proj2D o;
    const float limx = 1.3f * vc.tanfovx, limy = 1.3f * vc.tanfovy;
    const float txtz = pv.x / pv.z, tytz = pv.y / pv.z;
    o.clampx = (txtz < -limx) || (txtz > limx);
    o.clampy = (tytz < -limy) || (tytz > limy);
    o.tx = fminf(limx, fmaxf(-limx, txtz)) * pv.z;
    o.ty = fminf(limy, fmaxf(-limy, tytz)) * pv.z;
    o.tz = pv.z;
    const float iz = 1.f / pv.z;
    const float j00 = vc.focal_x * iz, j02 = -(vc.focal_x * o.tx) * iz * iz;
    const float j11 = vc.focal_y * iz, j12 = -(vc.focal_y * o.ty) * iz * iz;
    // Wr[i][k] = V[k][i] = V[4k+i]  (view = p @ V, row-vector convention)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o.T0[k] = j00 * V[4 * k + 0] + j02 * V[4 * k + 2];
        o.T1[k] = j11 * V[4 * k + 1] + j12 * V[4 * k + 2];
    }
    // u = Sigma * T0^T, w = Sigma * T1^T
    const float u0 = S.c0 * o.T0[0] + S.c1 * o.T0[1] + S.c2 * o.T0[2];
    const float u1 = S.c1 * o.T0[0] + S.c3 * o.T0[1] + S.c4 * o.T0[2];
    const float u2 = S.c2 * o.T0[0] + S.c4 * o.T0[1] + S.c5 * o.T0[2];
    const float w0 = S.c0 * o.T1[0] + S.c1 * o.T1[1] + S.c2 * o.T1[2];
    const float w1 = S.c1 * o.T1[0] + S.c3 * o.T1[1] + S.c4 * o.T1[2];
    const float w2 = S.c2 * o.T1[0] + S.c4 * o.T1[1] + S.c5 * o.T1[2];
    o.a = o.T0[0] * u0 + o.T0[1] * u1 + o.T0[2] * u2 + 0.3f;
    o.b = o.T1[0] * u0 + o.T1[1] * u1 + o.T1[2] * u2;
    o.c = o.T1[0] * w0 + o.T1[1] * w1 + o.T1[2] * w2 + 0.3f;
    return o;
}

// SH basis values for unit direction d (sh_utils.py:74-100); nb = (deg+1)^2 entries filled
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float B[16]) {
    B[0] = SH_C0;
    if (deg > 0) {
        B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = SH_C2_0 * xy; B[5] = SH_C2_1 * yz; B[6] = SH_C2_2 * (2.f * zz - xx - yy);
            B[7] = SH_C2_3 * xz; B[8] = SH_C2_4 * (xx - yy);
            if (deg > 2) {
                B[9] = SH_C3_0 * y * (3.f * xx - yy); B[10] = SH_C3_1 * xy * z;
                B[11] = SH_C3_2 * y * (4.f * zz - xx - yy);
                B[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
                B[13] = SH_C3_4 * x * (4.f * zz - xx - yy); B[14] = SH_C3_5 * z * (xx - yy);
                B[15] = SH_C3_6 * x * (xx - 3.f * yy);
            }
        }
    }
}

// K6's results are written once and not read again by this library: non-temporal stores (global_store ... nt). With plain
// stores the 248 MB of a 1M-Gaussian / SH 3 backward sat dirty in the memory-side cache when the kernel ended and were written
// back under whatever ran next: measured on the same box (profiles/r04_ab_nt.txt), K6 0.087 -> 0.091 ms (it now waits for its
// own writes) and the NEXT kernel -- K1 of the following step -- 0.095 -> 0.073; the dL/dSH rows alone: 0.087 -> 0.091 / 0.095 -> 0.076.
// (Loading the backward's accumulators non-temporally on top cost K6 9 us: not done.)
__device__ __forceinline__ void store_once(float4* p, float x, float y, float z, float w) {
    typedef float gsr_v4f __attribute__((ext_vector_type(4)));
    const gsr_v4f v = {x, y, z, w};
    __builtin_nontemporal_store(v, reinterpret_cast<gsr_v4f*>(p));
}

// Stage `cnt` SH rows (each 3K floats, contiguous in HBM) into LDS with pitch 3K+1. `rowlive[r]` = 0: nobody will read row r
// (a Gaussian no pixel gradient reached, three quarters of a dense scene) -- its 192 bytes are not fetched: the request goes to
// the batch's first vector instead (one address for all such lanes, a cache hit; the loads stay branch-free and in flight
// together) and nothing is written to LDS.
__device__ __forceinline__ void stage_rows_in(const float* __restrict__ src, float* lds,
                                              int cnt, int rowlen, const uint8_t* rowlive) {
    const int total = cnt * rowlen;
    const int pitch = rowlen + 1;
    if ((rowlen & 3) == 0) {
        // all of a thread's loads are issued before the first LDS write: one HBM round trip per
        // batch of STAGE_U loads instead of one per load (the simple loop waits on every load)
        constexpr int STAGE_U = 12;
        const float4* s4 = reinterpret_cast<const float4*>(src);
        const int nvec = total / 4, stride = blockDim.x;
        for (int i0 = threadIdx.x; i0 < nvec; i0 += stride * STAGE_U) {
            float4 v[STAGE_U];
            bool keep[STAGE_U];
#pragma unroll
            for (int u = 0; u < STAGE_U; ++u) {
                const int i = min(i0 + u * stride, nvec - 1);
                keep[u] = rowlive[(4 * i) / rowlen] != 0;
                v[u] = s4[keep[u] ? i : 0];
            }
#pragma unroll
            for (int u = 0; u < STAGE_U; ++u) {
                const int i = i0 + u * stride;
                if (i < nvec && keep[u]) {
                    const int e = 4 * i, row = e / rowlen, col = e - row * rowlen;  // rowlen%4==0: no row straddle
                    float* d = lds + row * pitch + col;
                    d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                }
            }
        }
    } else {
        for (int e = threadIdx.x; e < total; e += blockDim.x) {
            const int row = e / rowlen, col = e - row * rowlen;
            if (rowlive[row]) lds[row * pitch + col] = src[e];
        }
    }
}

// split rows: element e of a row lives in `dc` (e < 3) or in `rest` (e >= 3): DreamGaussian's two feature tensors
__device__ __forceinline__ void stage_rows_in_split(const float* __restrict__ dc, const float* __restrict__ rest, float* lds,
                                                    int cnt, int rowlen, const uint8_t* rowlive) {
    const int pitch = rowlen + 1, restlen = rowlen - 3;
    for (int e = threadIdx.x; e < cnt * rowlen; e += blockDim.x) {
        const int row = e / rowlen, col = e - row * rowlen;
        if (rowlive[row]) lds[row * pitch + col] = col < 3 ? dc[row * 3 + col] : rest[(size_t)row * restlen + (col - 3)];
    }
}
__device__ __forceinline__ void stage_rows_out_split(float* __restrict__ dc, float* __restrict__ rest, const float* lds,
                                                     int cnt, int rowlen, int accumulate, const uint8_t* rowlive) {
    const int pitch = rowlen + 1, restlen = rowlen - 3;
    for (int e = threadIdx.x; e < cnt * rowlen; e += blockDim.x) {
        const int row = e / rowlen, col = e - row * rowlen;
        if (accumulate && !rowlive[row]) continue;        // + 0: nothing to add
        const float v = lds[row * pitch + col];
        float* o = col < 3 ? dc + (row * 3 + col) : rest + ((size_t)row * restlen + (col - 3));
        *o = accumulate ? *o + v : v;
    }
}

__device__ __forceinline__ void stage_rows_out(float* __restrict__ dst, const float* lds,
                                               int cnt, int rowlen, int accumulate, const uint8_t* rowlive) {
    const int total = cnt * rowlen;
    const int pitch = rowlen + 1;
    if ((rowlen & 3) == 0) {
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (int i = threadIdx.x; i < total / 4; i += blockDim.x) {
            const int e = 4 * i, row = e / rowlen, col = e - row * rowlen;
            if (accumulate && !rowlive[row]) continue;    // a later view of a batch adds nothing to this row: neither read nor written
            const float* s = lds + row * pitch + col;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (accumulate) o = d4[i];
            if (accumulate) d4[i] = make_float4(o.x + s[0], o.y + s[1], o.z + s[2], o.w + s[3]);
            else store_once(d4 + i, s[0], s[1], s[2], s[3]);
        }
    } else {
        for (int e = threadIdx.x; e < total; e += blockDim.x) {
            const int row = e / rowlen, col = e - row * rowlen;
            if (accumulate && !rowlive[row]) continue;
            const float v = lds[row * pitch + col];
            dst[e] = accumulate ? dst[e] + v : v;
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------
// K1: preprocess forward.  A persistent grid walks batches of 256 Gaussians; a wave's 64 Gaussians only touch that wave's
// slice of LDS (no workgroup barrier inside the loop: the four waves drift apart and cover each other's latency).
//   A (lane = Gaussian)      frustum test, view direction, the SH basis values in registers
//   B                        the wave's SH rows pass through its LDS slice 16 at a time: LOAD with 16 lanes per row, lane =
//                            coefficient, one 12-byte load each (a row is 16 x 12 B contiguous: four rows per wave instruction,
//                            fully coalesced; the two tensors of DreamGaussian's `get_features`, features_dc / features_rest,
//                            gs_renderer.py:209-212, are read where they are: lane 0 the first, lanes 1.. the second -- no
//                            torch.cat copy); EVALUATE in the lane that owns the Gaussian: 3 x nb
//                            multiply-adds in coefficient order. (Round 2 summed 16 coefficient lanes per Gaussian with DPP:
//                            812 vector instructions per batch against 390 now; 12.5 KiB of LDS per workgroup against 21.5.)
//   C (lane = Gaussian)      projection, 2D covariance, conic, radius, tile rectangle, exact tile emission; the wave's 64
//                            records leave through the same LDS slice as four coalesced 1 KiB stores.
// What the kernel's length is made of (knock-outs at 1M Gaussians / SH 3 / 800^2, profiles/r03_k1_knockouts.txt): a batch is a
// CHAIN of memory round trips and a wave's loads and stores share one in-order counter, so a load issued behind the stores is
// delivered behind them. The next batch's per-Gaussian inputs are requested at the top of a batch and its first SH rows right
// before the stores, every wait in the loop is a count that leaves the stores outstanding (the code below keeps the number of
// stores the same on every path for that), the camera sits in LDS, and the grid is sized so that every workgroup walks the same
// number of batches: 0.101 -> 0.083 ms.
// dynamic LDS: [hist: nTiles u32 when hist_in_lds][4 waves x GSR_K1_WSLICE floats]
// ---------------------------------------------------------------------------------------
typedef float gsr_f3 __attribute__((ext_vector_type(3)));
typedef gsr_f3 gsr_f3u __attribute__((aligned(4)));
#define GSR_K1_ROWS 16           // SH rows a wave stages per round
#define GSR_K1_PITCH 49          // floats between staged rows (16 coefficients x 3 channels, + 1: odd)
#define GSR_K1_WSLICE 1024       // floats of LDS per wave: GSR_K1_ROWS staged rows, later the wave's 64 records (4 KiB) on their way out

template <bool RAW>      // RAW: the inputs are DreamGaussian's raw parameters, activations fused (ViewConst.raw_act)
__global__ void __launch_bounds__(256, 4)
gsr_preprocess_fwd(ViewTab views /* camera of this workgroup: views.v[blockIdx.y] */, int N, int K,
                   const float* __restrict__ means3D, const float* __restrict__ shs,
                   const float* __restrict__ shs_rest /* NULL: shs is [N,K,3]; else shs is [N,1,3] and this [N,K-1,3] */,
                   const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
                   const float* __restrict__ scales, const float* __restrict__ rotations,
                   const float* __restrict__ cov3D_precomp,
                   SplatRec* __restrict__ recs, EmitRec* __restrict__ emit,
                   int32_t* __restrict__ radii, uint32_t* __restrict__ tile_count,
                   unsigned long long* __restrict__ block_stats /*[grid][3]: M_ref, V, max(colour, depth) bits per workgroup*/,
                   int hist_in_lds,
                   uint8_t* __restrict__ flags8 /* colour-clamp bits for K6: 1 B instead of a 64-B record line */,
                   uint32_t* __restrict__ zero_base /* tile_count | cursor | counters of ALL views: zeroed here, by workgroup (0, 0) */,
                   uint32_t zero_words, uint32_t flag_word /* (even) index in zero_base of the 64-bit "zeroed" flag */, unsigned long long epoch,
                   uint32_t* __restrict__ wg_hist /* [views][grid][nTiles]: this workgroup's tile histogram, for the group-mate that reserves (LDS-histogram mode, groups of more than one) */,
                   uint32_t* __restrict__ group_base /* [views][groups][nTiles]: where the entries of a GROUP of `group` consecutive workgroups start inside each tile's list */,
                   int group /* workgroups per group: 1, or what one workgroup of gsr_scatter covers (its block = 256 x group threads) */,
                   uint32_t* __restrict__ arrive /* [views][2048] inside the zeroed words: how many workgroups of a group have stored their histogram */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pp[];
    const ViewConst vc = views.v[blockIdx.y];
    // The per-tile counts, the scatter's cursors and the counters start from zero: workgroup (0, 0) -- the first the dispatcher
    // hands out -- clears them (20 KB at 2 500 tiles) and publishes a per-launch tag; every workgroup looks at the tag once, right
    // before its first atomic on a tile count (~80 us later at 1M Gaussians): the hipMemsetAsync in front of K1 (a launch of its
    // own, 6 us on the driver's box) is gone. Stale memory never holds this launch's tag (a process-wide counter).
    unsigned long long* const zflag = reinterpret_cast<unsigned long long*>(zero_base + flag_word);
    __shared__ uint32_t zfail_s;
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        for (uint32_t i = threadIdx.x; i < zero_words; i += blockDim.x)
            if ((i & ~1u) != flag_word) zero_base[i] = 0u;
        // EVERY zeroing thread makes its own stores visible at agent scope before the barrier (the barrier alone does not wait for
        // another wave's stores to reach L2): only then may the tag follow them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(zflag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    // false: the tag never arrived (2^22 polls, seconds) -- the caller must not touch the counters; the workgroup reports it through
    // its M_ref statistic (>= 2^62: gsr_forward returns an error instead of results built on memory that may not be cleared)
    auto wait_zeroed = [&]() -> bool {                    // (block-uniform call sites)
        if (threadIdx.x == 0) {
            int spin = 0;
#pragma unroll 1
            for (; spin < (1 << 22) && __hip_atomic_load(zflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch; ++spin)
                __builtin_amdgcn_s_sleep(8);
            zfail_s = spin >= (1 << 22) ? 1u : 0u;
        }
        __syncthreads();
        // NO acquire fence here. Every access the callers make to the cleared words is an agent-scope atomic read-modify-write, and
        // those execute at the memory side (never out of this XCD's L2 or a vector cache): they are issued after the tag was seen
        // (its value came back before the branch above; the barrier holds the other waves), the zeros reached that point before
        // the tag did (the release in front of it), and the later readers are later kernels. An agent-scope acquire costs every
        // wave of every workgroup a wait for its outstanding record stores plus an invalidate of the XCD's caches: measured round
        // 5, same box, K1 with / without it: 0.048 / 0.037 ms at 250k Gaussians, 0.0918 / 0.0797 at 1M (profiles/r05_ab_round5.txt).
        return zfail_s == 0u;
    };
    bool zeroed_ok = true;
    const int nTiles = vc.gx * vc.gy;
    // 256 bytes behind the statistics that nobody reads: where the lanes past the end of the array store (below)
    char* const sink = reinterpret_cast<char*>(block_stats) + (size_t)gridDim.y * 2048 * 3 * 8;
    {   // per-view outputs: [views][N] records / radii / flags, [views][nTiles] counts, [views][grid][3] statistics
        const size_t vo = (size_t)blockIdx.y * (size_t)N;
        recs += vo; emit += vo; radii += vo; flags8 += vo;
        tile_count += (size_t)blockIdx.y * nTiles;
        block_stats += (size_t)blockIdx.y * gridDim.x * 3;
    }
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem_pp);
    float* stage = reinterpret_cast<float*>(smem_pp + (hist_in_lds ? ((nTiles * 4 + 15) & ~15) : 0));
    const int nb = (vc.sh_degree + 1) * (vc.sh_degree + 1);          // active coefficients (<= 16)
    const bool coop = (shs != nullptr) && nb > 1;                     // phases A/B only for view-dependent colour

    if (hist_in_lds) {
        for (int t = threadIdx.x; t < nTiles; t += blockDim.x) hist[t] = 0;
    } else zeroed_ok = wait_zeroed();                     // tile grids beyond the LDS histogram: every emission is a global atomic
    if (!zeroed_ok) {                                     // (block-uniform)
        if (threadIdx.x == 0) { block_stats[3 * blockIdx.x] = 1ull << 62; block_stats[3 * blockIdx.x + 1] = 0; block_stats[3 * blockIdx.x + 2] = 0; }
        return;
    }
    // The camera sits in LDS: read through its device pointers it comes back as VECTOR-memory loads (the compiler cannot
    // prove the memory read-only), and waiting for one of those at the top of a batch also waits for the previous batch's
    // stores (one in-order counter for a wave's loads and stores).
    __shared__ __attribute__((aligned(16))) float cam[36];
    if (threadIdx.x < 16) cam[threadIdx.x] = vc.view[cam_index(threadIdx.x, vc.mat_t & 1)];
    else if (threadIdx.x < 32) cam[threadIdx.x] = vc.proj[cam_index(threadIdx.x - 16, vc.mat_t & 2)];
    else if (threadIdx.x < 35) cam[threadIdx.x] = vc.campos[threadIdx.x - 32];
    __syncthreads();

    const float* V = cam;
    const float* P = cam + 16;
    const float* campos = cam + 32;
    unsigned long long my_ref = 0, my_vis = 0;
    float my_cmax = 0.f;              // largest colour component / depth of a listed Gaussian (bound used by the backward)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, grp = lane >> 4;

    // A batch is a chain of memory round trips (inputs -> SH rows -> stores), and on this part a wave's stores count in the same
    // in-order counter as its loads: a load issued AFTER the stores cannot be consumed before the stores have completed. So the
    // NEXT batch's per-Gaussian inputs are requested at the top of a batch and its first SH rows right before the stores: the
    // next batch starts on data that is already there, and the first load it issues itself is consumed ~250 instructions later.
    float nmx = 0.f, nmy = 0.f, nmz = 0.f, nop = 0.f, nqw = 1.f, nqx = 0.f, nqy = 0.f, nqz = 0.f, nsx = 1.f, nsy = 1.f, nsz = 1.f;
    // branch-free (index clamped, a dummy source when the 3D covariance is given): a load inside a branch comes with a register
    // copy right behind it, i.e. a wait for memory at the top of the batch
    const float* qbase = cov3D_precomp ? reinterpret_cast<const float*>(recs) : rotations;
    const float* sbase = cov3D_precomp ? reinterpret_cast<const float*>(recs) : scales;
    const int qmul = cov3D_precomp ? 0 : 4, smul = cov3D_precomp ? 0 : 3;
    auto load_inputs = [&](int i) {                       // (culled Gaussians included: no dependent second trip)
        const size_t ic = (size_t)min(i, N - 1);
        const gsr_f3 m = *reinterpret_cast<const gsr_f3u*>(means3D + 3 * ic);
        const float o = opacities[ic];
        const float4 q = *reinterpret_cast<const float4*>(qbase + qmul * ic);
        const gsr_f3 sc = *reinterpret_cast<const gsr_f3u*>(sbase + smul * ic);
        nmx = m.x; nmy = m.y; nmz = m.z; nop = o;
        nqw = q.x; nqx = q.y; nqy = q.z; nqz = q.w; nsx = sc.x; nsy = sc.y; nsz = sc.z;
    };
    // SH rows: 16 lanes per row, lane = coefficient, one 12-byte load each (a row is 16 x 12 B contiguous: four rows per wave
    // instruction, fully coalesced); GSR_K1_ROWS rows per round
    const int ki = min(l15, nb - 1);
    const bool from_rest = shs_rest != nullptr && ki > 0;
    const float* cbase = !coop ? nullptr : (from_rest ? shs_rest + (ki - 1) * 3 : shs + ki * 3);
    const long long rstride = from_rest ? (long long)(K - 1) * 3 : (long long)(shs_rest ? 1 : K) * 3;   // floats per row
    gsr_f3 cf[GSR_K1_ROWS / 4];
    auto load_round = [&](int wbase, int r) {
        const int g0 = wbase + r * GSR_K1_ROWS + grp;
        if (wbase + 64 <= N) {                            // wave-uniform: only the last wave of the array needs the clamp
            const float* p = cbase + (long long)g0 * rstride;
#pragma unroll
            for (int u = 0; u < GSR_K1_ROWS / 4; ++u) cf[u] = *reinterpret_cast<const gsr_f3u*>(p + (long long)(4 * u) * rstride);
        } else {
#pragma unroll
            for (int u = 0; u < GSR_K1_ROWS / 4; ++u)
                cf[u] = *reinterpret_cast<const gsr_f3u*>(cbase + (long long)min(g0 + 4 * u, N - 1) * rstride);
        }
    };
    const int bstride = gridDim.x * 256;
    load_inputs(blockIdx.x * 256 + threadIdx.x);
    asm volatile("" : "+v"(nmx), "+v"(nmy), "+v"(nmz), "+v"(nop), "+v"(nqw), "+v"(nqx), "+v"(nqy), "+v"(nqz), "+v"(nsx), "+v"(nsy), "+v"(nsz));
    if (coop && blockIdx.x * 256 < N) load_round(blockIdx.x * 256 + wave * 64, 0);
    // The waits the compiler places are STATIC counts ("at most n younger operations outstanding"), the minimum over all paths into
    // a point. Inside the loop seven stores sit between the first SH rows' request and their use; the path from here must hold as
    // many, or the loop's wait would be "everything but the four input loads" -- i.e. all of the previous batch's stores.
#pragma unroll
    for (int d = 0; d < 7; ++d) reinterpret_cast<uint32_t*>(sink)[8 * d + 1] = (uint32_t)d;   // (apart: not merged into wider stores)

    for (int base = blockIdx.x * 256; base < N; base += bstride) {
        const int idx = base + threadIdx.x;
        const float mx = nmx, my = nmy, mz = nmz;
        const float4 q_in = make_float4(nqw, nqx, nqy, nqz);
        const gsr_f3 s_in = {nsx, nsy, nsz};
        const float op_in = nop;
        const float depth = idx < N ? view_depth(V, mx, my, mz) : 0.f;
        const bool front = (idx < N) && depth > 0.2f;
        load_inputs(base + bstride + threadIdx.x);        // the next batch's inputs: in flight during this one (index clamped: unconditional)
        float sh_r = 0.f, sh_g = 0.f, sh_b = 0.f;       // view-dependent colour of this lane's Gaussian (before the +0.5 and the clamp)
        if (coop) {
            // A wave stages its own rows in its own slice of LDS: no workgroup barrier, the four waves drift apart and
            // cover each other's memory latency.
            float* wstage = stage + wave * GSR_K1_WSLICE;
            // ---- phase A: basis values of this lane's Gaussian, kept in registers
            float Bv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) Bv[k] = 0.f;
            if (front) {
                float dx = mx - campos[0], dy = my - campos[1], dz = mz - campos[2];
                const float inv = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
                dx *= inv; dy *= inv; dz *= inv;
                sh_basis(vc.sh_degree, dx, dy, dz, Bv);
            }
            // ---- phase B: GSR_K1_ROWS rows per round. Load: 16 lanes per row, lane = coefficient, one 12-byte load each (a row
            // is 16 x 12 B contiguous: four rows per wave instruction, fully coalesced) -> LDS slot [row][3 coef .. +2].
            // Evaluate: the lane that owns the Gaussian reads its row back (odd pitch: conflict-free) and runs the
            // 3 x nb multiply-adds in registers, in coefficient order like the reference's sum (sh_utils.py:74-100).
            const int wbase = base + wave * 64;
#pragma unroll
            for (int r = 0; r < 64 / GSR_K1_ROWS; ++r) {
                float* wr = wstage + grp * GSR_K1_PITCH + 3 * l15;
#pragma unroll
                for (int u = 0; u < GSR_K1_ROWS / 4; ++u) {
                    wr[u * 4 * GSR_K1_PITCH] = cf[u].x; wr[u * 4 * GSR_K1_PITCH + 1] = cf[u].y; wr[u * 4 * GSR_K1_PITCH + 2] = cf[u].z;
                }
                if (r + 1 < 64 / GSR_K1_ROWS) load_round(wbase, r + 1);   // in flight while this round is evaluated
                wave_lds_handoff();
                if (lane / GSR_K1_ROWS == r) {
                    const float* rr = wstage + (lane % GSR_K1_ROWS) * GSR_K1_PITCH;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { sh_r += Bv[k] * rr[3 * k]; sh_g += Bv[k] * rr[3 * k + 1]; sh_b += Bv[k] * rr[3 * k + 2]; }
                    if (nb > 4) {
#pragma unroll
                        for (int k = 4; k < 9; ++k) { sh_r += Bv[k] * rr[3 * k]; sh_g += Bv[k] * rr[3 * k + 1]; sh_b += Bv[k] * rr[3 * k + 2]; }
                    }
                    if (nb > 9) {
#pragma unroll
                        for (int k = 9; k < 16; ++k) { sh_r += Bv[k] * rr[3 * k]; sh_g += Bv[k] * rr[3 * k + 1]; sh_b += Bv[k] * rr[3 * k + 2]; }
                    }
                }
                wave_lds_handoff();
            }
        }
        // ---- phase C (no `continue` for the lanes past the end: one path to the loop's back edge, see the prefetch below)
        SplatRec rec;
        rec.x = rec.y = rec.qa = rec.qb = rec.qc = rec.opac = rec.r = rec.g = rec.b = rec.depth = 0.f;
        rec.id = (uint32_t)idx; rec.bbx = pack16(1, 0); rec.bby = pack16(1, 0);
        rec.rectx = 0; rec.recty = 0; rec.flags = 0;
        EmitRec em; em.rectx = 0; em.recty = 0; em.depth_bits = 0; em.mask = 0;
        int32_t radius_out = 0;

        if (idx < N) {
        float3 pv;
        pv.x = V[0] * mx + V[4] * my + V[8] * mz + V[12];
        pv.y = V[1] * mx + V[5] * my + V[9] * mz + V[13];
        pv.z = depth;
        if (front) {
            const float hx = P[0] * mx + P[4] * my + P[8] * mz + P[12];
            const float hy = P[1] * mx + P[5] * my + P[9] * mz + P[13];
            const float hw = P[3] * mx + P[7] * my + P[11] * mz + P[15];
            const float p_w = 1.0f / (hw + 0.0000001f);
            const float ndcx = hx * p_w, ndcy = hy * p_w;

            Cov3 S;
            if (cov3D_precomp) {
                const float* c = cov3D_precomp + 6 * (size_t)idx;
                S.c0 = c[0]; S.c1 = c[1]; S.c2 = c[2]; S.c3 = c[3]; S.c4 = c[4]; S.c5 = c[5];
            } else {
                float4 q = q_in;
                float3 s = make_float3(s_in.x, s_in.y, s_in.z);
                if (RAW) { float inv; q = act_normalize(q, &inv); s.x = __expf(s.x); s.y = __expf(s.y); s.z = __expf(s.z); }
                s.x *= vc.scale_modifier; s.y *= vc.scale_modifier; s.z *= vc.scale_modifier;
                float R[9];
                quat_to_R(q, R);
                S = cov3d_from_scale_rot(s, R);
            }
            const Proj2D pj = project_cov(vc, V, pv, S);
            const float det = pj.a * pj.c - pj.b * pj.b;
            if (det != 0.f) {
                const float det_inv = 1.f / det;
                const float cA = pj.c * det_inv, cB = -pj.b * det_inv, cC = pj.a * det_inv;
                const float mid = 0.5f * (pj.a + pj.c);
                const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
                const float radius = ceilf(3.f * sqrtf(lam));
                const float px = ((ndcx + 1.0f) * vc.W - 1.0f) * 0.5f;
                const float py = ((ndcy + 1.0f) * vc.H - 1.0f) * 0.5f;
                // reference tile rect (A.3)
                const float fgx = (float)vc.gx, fgy = (float)vc.gy;
                const int rx0 = (int)fminf(fgx, fmaxf(0.f, truncf((px - radius) / GSR_TILE)));
                const int rx1 = (int)fminf(fgx, fmaxf(0.f, truncf((px + radius + (GSR_TILE - 1)) / GSR_TILE)));
                const int ry0 = (int)fminf(fgy, fmaxf(0.f, truncf((py - radius) / GSR_TILE)));
                const int ry1 = (int)fminf(fgy, fmaxf(0.f, truncf((py + radius + (GSR_TILE - 1)) / GSR_TILE)));
                const int area_ref = (rx1 - rx0) * (ry1 - ry0);
                if (area_ref > 0) {
                    radius_out = (int32_t)radius;
                    my_ref += (unsigned long long)area_ref;
                    my_vis += 1;
                    // colour
                    float cr, cg, cb;
                    uint32_t flags = 0;
                    if (colors_precomp) {
                        const gsr_f3 c = *reinterpret_cast<const gsr_f3u*>(colors_precomp + 3 * (size_t)idx);
                        cr = c.x; cg = c.y; cb = c.z;
                    } else {
                        if (coop) {
                            cr = sh_r; cg = sh_g; cb = sh_b;
                        } else {                            // degree 0: one 12-byte load per lane, no direction
                            const gsr_f3 c = *reinterpret_cast<const gsr_f3u*>(shs + (size_t)idx * (shs_rest ? 1 : K) * 3);
                            cr = SH_C0 * c.x; cg = SH_C0 * c.y; cb = SH_C0 * c.z;
                        }
                        cr += 0.5f; cg += 0.5f; cb += 0.5f;
                        if (cr < 0.f) { flags |= 1u; cr = 0.f; }
                        if (cg < 0.f) { flags |= 2u; cg = 0.f; }
                        if (cb < 0.f) { flags |= 4u; cb = 0.f; }
                    }
                    const float op = RAW ? act_sigmoid(op_in) : op_in;
                    rec.x = px; rec.y = py;
                    rec.qa = -0.5f * cA * GSR_LOG2E; rec.qb = -cB * GSR_LOG2E; rec.qc = -0.5f * cC * GSR_LOG2E;
                    rec.opac = op; rec.r = cr; rec.g = cg; rec.b = cb; rec.depth = pv.z;
                    my_cmax = fmaxf(my_cmax, fmaxf(fmaxf(fabsf(cr), fabsf(cg)), fmaxf(fabsf(cb), pv.z)));
                    // Exact no-op culling: a pixel can only pass the alpha >= 1/255 test inside the
                    // axis-aligned box |dx| <= sqrt(2 tau a), |dy| <= sqrt(2 tau c), tau = ln(255 o)
                    // (a 0.2% safety margin dominates fp32 rounding of the in-kernel alpha).
                    int ex0 = 0, ex1 = 0, ey0 = 0, ey1 = 0;
                    const float tau = __logf(255.f * op);
                    if (tau > 0.f) {
                        const float hxw = sqrtf(2.f * tau * pj.a) * 1.002f + 0.01f;
                        const float hyw = sqrtf(2.f * tau * pj.c) * 1.002f + 0.01f;
                        const float bx0 = fminf(fmaxf(ceilf(px - hxw), -32768.f), 32767.f);
                        const float bx1 = fminf(fmaxf(floorf(px + hxw), -32768.f), 32767.f);
                        const float by0 = fminf(fmaxf(ceilf(py - hyw), -32768.f), 32767.f);
                        const float by1 = fminf(fmaxf(floorf(py + hyw), -32768.f), 32767.f);
                        const int ibx0 = (int)bx0, ibx1 = (int)bx1, iby0 = (int)by0, iby1 = (int)by1;
                        rec.bbx = pack16(ibx0, ibx1); rec.bby = pack16(iby0, iby1);
                        if (ibx1 >= 0 && iby1 >= 0 && ibx0 < vc.W && iby0 < vc.H && ibx0 <= ibx1 && iby0 <= iby1) {
                            ex0 = max(rx0, max(ibx0, 0) / GSR_TILE);
                            ex1 = min(rx1, min(ibx1, vc.W - 1) / GSR_TILE + 1);
                            ey0 = max(ry0, max(iby0, 0) / GSR_TILE);
                            ey1 = min(ry1, min(iby1, vc.H - 1) / GSR_TILE + 1);
                        }
                    }
                    if (ex1 > ex0 && ey1 > ey0) {
                        // tile-exact emission for small rects: keep a tile only if alpha can reach
                        // 1/255 on one of its pixels (bit mask travels to the scatter kernel)
                        const bool masked = (ex1 - ex0) * (ey1 - ey0) <= GSR_EMIT_MASK_TILES;
                        const float pmin = min_visible_power(op);
                        uint32_t mask = 0, bit = 1u;
                        for (int ty = ey0; ty < ey1; ++ty)
                            for (int tx = ex0; tx < ex1; ++tx, bit <<= 1) {
                                if (masked) {
                                    const float xa = (float)(tx * GSR_TILE), ya = (float)(ty * GSR_TILE);
                                    const float xb = fminf(xa + (GSR_TILE - 1), (float)(vc.W - 1));
                                    const float yb = fminf(ya + (GSR_TILE - 1), (float)(vc.H - 1));
                                    if (!(rect_max_power(px, py, rec.qa, rec.qb, rec.qc, xa, xb, ya, yb) >= pmin)) continue;
                                    mask |= bit;
                                }
                                const int t = ty * vc.gx + tx;
                                if (hist_in_lds) atomicAdd(&hist[t], 1u);
                                else atomicAdd(&tile_count[t], 1u);
                            }
                        if (!masked || mask != 0u) {
                            flags |= GSR_FLAG_EMIT;
                            rec.rectx = pack16(ex0, ex1); rec.recty = pack16(ey0, ey1);
                            em.rectx = rec.rectx; em.recty = rec.recty; em.depth_bits = __float_as_uint(pv.z);
                            em.mask = masked ? mask : 0xffffffffu;
                        }
                    }
                    rec.flags = flags;
                }
            }
        }
        }
        // The next batch's inputs have long arrived: have the compiler take delivery HERE (an empty asm that "modifies" them), not
        // at the copies it would otherwise place on the loop's back edge -- behind the stores, with a wait for everything.
        asm volatile("" : "+v"(nmx), "+v"(nmy), "+v"(nmz), "+v"(nop), "+v"(nqw), "+v"(nqx), "+v"(nqy), "+v"(nqz), "+v"(nsx), "+v"(nsy), "+v"(nsz));
        if (coop && base + bstride < N) load_round(base + bstride + wave * 64, 0);   // the next batch's first SH rows: requested BEFORE the stores
        // Stores: no branch around them (the same number of stores on every path, see above) -- a lane past the end of the array
        // stores into the sink.
        const bool live = idx < N;
        *(live ? radii + idx : reinterpret_cast<int32_t*>(sink)) = radius_out;
        *(live ? flags8 + idx : reinterpret_cast<uint8_t*>(sink)) = (uint8_t)rec.flags;
        if (base + wave * 64 + 64 <= N) {
            // the wave's 64 records leave as four fully coalesced 1 KiB stores (lane = consecutive 16 bytes) instead of four
            // stores of 64 sixteen-byte pieces 64 bytes apart: transposed through the wave's LDS slice
            uint4* wrec = reinterpret_cast<uint4*>(stage + wave * GSR_K1_WSLICE);
            wave_lds_handoff();
#pragma unroll
            for (int j = 0; j < 4; ++j) wrec[lane * 4 + j] = reinterpret_cast<uint4*>(&rec)[j];
            wave_lds_handoff();
            uint4* dst = reinterpret_cast<uint4*>(recs + (base + wave * 64));
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[j * 64 + lane] = wrec[j * 64 + lane];
            wave_lds_handoff();
        } else {
            uint4* dst = live ? reinterpret_cast<uint4*>(recs + idx) : reinterpret_cast<uint4*>(sink);
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[j] = reinterpret_cast<uint4*>(&rec)[j];
        }
        *(live ? reinterpret_cast<uint4*>(emit) + idx : reinterpret_cast<uint4*>(sink)) = *reinterpret_cast<uint4*>(&em);
    }

    // statistics: wave -> workgroup in LDS -> one plain store per workgroup (summed by tile_scan).
    // Same-address global atomics execute one after the other at the memory side: 2 per wave to two
    // shared counters was ~40 us of serialised tail at 2048 waves.
    __shared__ unsigned long long wstat[3][16];
    const uint32_t wave_cmax = wave_max_u32(__float_as_uint(my_cmax));   // non-negative floats order like their bits
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        my_ref += __shfl_xor(my_ref, off, 64);
        my_vis += __shfl_xor(my_vis, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { wstat[0][threadIdx.x >> 6] = my_ref; wstat[1][threadIdx.x >> 6] = my_vis; wstat[2][threadIdx.x >> 6] = wave_cmax; }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long sum = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w)
            sum = threadIdx.x < 2 ? sum + wstat[threadIdx.x][w] : max(sum, wstat[2][w]);
        block_stats[3 * blockIdx.x + threadIdx.x] = sum;
    }
    if (hist_in_lds) {
        if (!wait_zeroed()) {
            if (threadIdx.x == 0) block_stats[3 * blockIdx.x] = 1ull << 62;
            return;
        }
        // The flush RESERVES: the value the atomic returns is the number of entries others have claimed in that tile's list so far = where
        // the own entries start; gsr_scatter hands positions out from there -- no counting pass and no reservation atomics of its own.
        //
        // GROUPS (round 6). What is reserved is ONE range per tile for `group` consecutive workgroups (the Gaussians ONE workgroup of
        // gsr_scatter emits: its block is 256 x group threads with one LDS cursor per tile). A range per K1 workgroup meant runs of ~4
        // keys (32 B) per (workgroup, tile) at 1M Gaussians; the 128-byte line around a run is shared with three other workgroups'
        // runs which, nine times out of ten, live in ANOTHER XCD's L2 -- every L2 wrote its quarter of the line back on its own
        // and the scatter's HBM writes were 3.2 x its keys (profiles/r05_*_pmc.txt). A group's run is ~16 keys = a whole line written
        // by one workgroup through one L2. The workgroups of a group do not wait for each other: each stores its histogram row
        // (write-through stores), then counts itself in; the one that finds the others already counted reads their rows back
        // (loads that are not served from its own XCD's L2), adds its own and reserves for all. Four times fewer returning atomics
        // per tile counter than one reservation per workgroup.
        const int G = group;
        const int grp = (int)blockIdx.x / G, gfirst = grp * G;
        const int gsize = min(G, (int)gridDim.x - gfirst);
        const int ngroups = ((int)gridDim.x + G - 1) / G;
        const int pitch = (nTiles + 3) & ~3;                              // words between histogram rows (16-byte stores)
        if (gsize > 1) {
            // The hand-off (MI355X guide, inter-workgroup visibility R1): WRITE-THROUGH (sc1) row stores, every storing wave drains its
            // stores, barrier, ONE lane counts the workgroup in with a relaxed agent-scope atomic; the reader uses sc1 loads. No
            // release fence: with it (buffer_wbl2: the XCD's L2 writes back every dirty record line first) K1 was 0.104 ms instead
            // of 0.069 (profiles/r06_group_reservation.txt).
            __shared__ uint32_t last_s;
            uint4* __restrict__ my_row = reinterpret_cast<uint4*>(wg_hist + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * pitch);
            for (int t4 = threadIdx.x; t4 < pitch / 4; t4 += 256) {       // (hist is a multiple of 16 bytes long: the words past nTiles are never read back)
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v = reinterpret_cast<const u32x4*>(hist)[t4];
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(my_row + t4), "v"(v) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0)
                last_s = __hip_atomic_fetch_add(arrive + (size_t)blockIdx.y * 2048 + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)(gsize - 1) ? 1u : 0u;
            __syncthreads();
            if (!last_s) return;
        }
        const uint32_t* __restrict__ rows = wg_hist + ((size_t)blockIdx.y * gridDim.x + gfirst) * pitch;
        const int own = (int)blockIdx.x - gfirst;
        uint32_t* __restrict__ base_row = group_base + ((size_t)blockIdx.y * ngroups + grp) * nTiles;
        // every group starts its flush at a different tile: no burst of atomics on one address
        const int t0 = (int)(((unsigned)grp * 67u) % (unsigned)nTiles);
        for (int i0 = threadIdx.x; i0 < nTiles; i0 += 256 * 4) {          // four returning atomics in flight per thread
            int tt[4]; uint32_t cc[4], got[4], oth[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + 256 * u, nTiles - 1);
                int t = t0 + i; if (t >= nTiles) t -= nTiles;
                tt[u] = t;
                // the group-mates' counts: agent-scope loads written out by hand (sc1: not served from this XCD's L2) -- the compiler
                // waits for an atomic load where it issues it, and these are twelve independent requests. Branch-free: a mate the
                // (short, last) group does not have is read from the own row and dropped.
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int q = k + (k >= own ? 1 : 0);
                    const uint32_t* p = rows + (size_t)(q < gsize ? q : own) * pitch + t;
                    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(oth[u][k]) : "v"(p) : "memory");
                }
            }
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(oth[0][0]), "+v"(oth[0][1]), "+v"(oth[0][2]), "+v"(oth[1][0]), "+v"(oth[1][1]), "+v"(oth[1][2]),
                           "+v"(oth[2][0]), "+v"(oth[2][1]), "+v"(oth[2][2]), "+v"(oth[3][0]), "+v"(oth[3][1]), "+v"(oth[3][2]) :: "memory");
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint32_t c = hist[tt[u]];
#pragma unroll
                for (int k = 0; k < 3; ++k) c += k + (k >= own ? 1 : 0) < gsize ? oth[u][k] : 0u;
                cc[u] = i0 + 256 * u < nTiles ? c : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { got[u] = 0u; if (cc[u]) got[u] = atomicAdd(&tile_count[tt[u]], cc[u]); }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (cc[u]) base_row[tt[u]] = got[u];
        }
    }
}

// ---------------------------------------------------------------------------------------
// K6: preprocess backward.  Exact analytic gradient of K1; recomputes the
// forward intermediates from the inputs instead of re-reading saved state.
// dynamic LDS: blockDim.x * (3K+1) floats when shs (used for SH in, then dSH out).
// ---------------------------------------------------------------------------------------
// One (view, Gaussian) of K6: the gradients of this view's 2D quantities (g2d) carried to the Gaussian's parameters.
// Outputs are THIS view's contributions (exact zeros for a Gaussian the view culled); the callers add views up.
// a pointer every lane holds the same value of, moved to scalar registers (loads through it become scalar loads)
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v), hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}
struct K6Out { float dm[3], dm2[2], dop, dsc[3], dq[4], dcov[6], dcol[3], dsh[3]; };
// What k6_gaussian reads of a Gaussian, requested by the caller at the top of the batch (branch-free, in flight while the SH
// rows are staged): the camera-independent inputs ...
struct K6In { float mx, my, mz, op; float4 q; float sx, sy, sz; };
// ... and the per-view ones: the 2D gradients render_bwd accumulated (three 16-byte loads) and K1's colour-clamp bits
struct K6ViewIn { float4 g0, g1, g2; uint32_t flags; };
template <bool RAW>
__device__ __forceinline__ void k6_gaussian(const ViewConst& vc, const float* cam /* LDS: view[16] | proj[16] | campos[3] */, int idx, int N, int K, bool live,
                                            const K6In& in, const K6ViewIn& vin, const float* __restrict__ shs,
                                            const float* __restrict__ cov3D_precomp,
                                            float* __restrict__ dL_dshs, bool stage, const float* myrow /* staged SH row (input) */,
                                            float* myrow_out /* LDS row of dL/dSH (the same row as the input for one view per launch) */,
                                            bool lds_accum /* several views per launch: add to myrow_out instead of overwriting it */, int accumulate,
                                            bool sh_reg /* K == 1, several views per launch: dL/dSH of this view goes to out.dsh */, K6Out& out,
                                            bool sparse = false /* the caller never writes out the staged row of a Gaussian that is not `live` */) {
    const int rowlen = 3 * K;
    const bool use_sh = (shs != nullptr);
    const float* V = cam;
    const float* P = cam + 16;
    const float* campos = cam + 32;
    const float g[12] = {vin.g0.x, vin.g0.y, vin.g0.z, vin.g0.w, vin.g1.x, vin.g1.y, vin.g1.z, vin.g1.w, vin.g2.x, vin.g2.y, vin.g2.z, vin.g2.w};
    out.dm[0] = out.dm[1] = out.dm[2] = 0.f; out.dm2[0] = out.dm2[1] = 0.f; out.dop = 0.f; out.dsc[0] = out.dsc[1] = out.dsc[2] = 0.f; out.dq[0] = out.dq[1] = out.dq[2] = out.dq[3] = 0.f;
#pragma unroll
    for (int e = 0; e < 6; ++e) out.dcov[e] = 0.f;
    out.dcol[0] = out.dcol[1] = out.dcol[2] = 0.f;
    out.dsh[0] = out.dsh[1] = out.dsh[2] = 0.f;
    if (live) {
        const uint32_t flags = vin.flags;
        const float mx = in.mx, my = in.my, mz = in.mz;
        float3 pv;
        pv.x = V[0] * mx + V[4] * my + V[8] * mz + V[12];
        pv.y = V[1] * mx + V[5] * my + V[9] * mz + V[13];
        pv.z = view_depth(V, mx, my, mz);
        const float hx = P[0] * mx + P[4] * my + P[8] * mz + P[12];
        const float hy = P[1] * mx + P[5] * my + P[9] * mz + P[13];
        const float hw = P[3] * mx + P[7] * my + P[11] * mz + P[15];
        const float m_w = 1.0f / (hw + 0.0000001f);

        // ---- screen-space mean ------------------------------------------------------
        const float gmx = g[0] * (GSR_LN2 * 0.5f * vc.W);   // dL/d ndc.x
        const float gmy = g[1] * (GSR_LN2 * 0.5f * vc.H);
        out.dm2[0] = gmx; out.dm2[1] = gmy;
        const float mul1 = hx * m_w * m_w, mul2 = hy * m_w * m_w;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            out.dm[k] = (P[4 * k] * m_w - P[4 * k + 3] * mul1) * gmx + (P[4 * k + 1] * m_w - P[4 * k + 3] * mul2) * gmy;
        // ---- depth ------------------------------------------------------------------
        const float gdepth = g[9];
        out.dm[0] += V[2] * gdepth; out.dm[1] += V[6] * gdepth; out.dm[2] += V[10] * gdepth;
        // ---- opacity ----------------------------------------------------------------
        out.dop = g[5];
        if (RAW) { const float o = act_sigmoid(in.op); out.dop *= o * (1.f - o); }   // d sigmoid

        // ---- colour -----------------------------------------------------------------
        float gr = g[6], gg = g[7], gb = g[8];
        if (!use_sh) {
            out.dcol[0] = gr; out.dcol[1] = gg; out.dcol[2] = gb;
        } else {
            if (flags & 1u) gr = 0.f;
            if (flags & 2u) gg = 0.f;
            if (flags & 4u) gb = 0.f;
            float dx = mx - campos[0], dy = my - campos[1], dz = mz - campos[2];
            const float len2 = dx * dx + dy * dy + dz * dz;
            const float inv = 1.f / sqrtf(len2);
            const float x = dx * inv, y = dy * inv, z = dz * inv;
            const int deg = vc.sh_degree;
            const int nb = (deg + 1) * (deg + 1);
            float B[16];
            sh_basis(deg, x, y, z, B);
            const float* row = stage ? myrow : (shs + (size_t)idx * rowlen);
            // direction derivative: dRGB/d(x,y,z) contracted with (gr,gg,gb)
            float ddx = 0.f, ddy = 0.f, ddz = 0.f;
            if (deg > 0) {
                float s[16];   // s[k] = sh[k] . grad_rgb
#pragma unroll
                for (int k = 1; k < 16; ++k)
                    s[k] = (k < nb) ? (row[3 * k] * gr + row[3 * k + 1] * gg + row[3 * k + 2] * gb) : 0.f;
                ddx = -SH_C1 * s[3]; ddy = -SH_C1 * s[1]; ddz = SH_C1 * s[2];
                if (deg > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z;
                    ddx += SH_C2_0 * y * s[4] + SH_C2_2 * (-2.f * x) * s[6] + SH_C2_3 * z * s[7] + SH_C2_4 * (2.f * x) * s[8];
                    ddy += SH_C2_0 * x * s[4] + SH_C2_1 * z * s[5] + SH_C2_2 * (-2.f * y) * s[6] + SH_C2_4 * (-2.f * y) * s[8];
                    ddz += SH_C2_1 * y * s[5] + SH_C2_2 * (4.f * z) * s[6] + SH_C2_3 * x * s[7];
                    if (deg > 2) {
                        ddx += SH_C3_0 * (6.f * x * y) * s[9] + SH_C3_1 * (y * z) * s[10] + SH_C3_2 * (-2.f * x * y) * s[11]
                             + SH_C3_3 * (-6.f * x * z) * s[12] + SH_C3_4 * (4.f * zz - 3.f * xx - yy) * s[13]
                             + SH_C3_5 * (2.f * x * z) * s[14] + SH_C3_6 * (3.f * xx - 3.f * yy) * s[15];
                        ddy += SH_C3_0 * (3.f * xx - 3.f * yy) * s[9] + SH_C3_1 * (x * z) * s[10]
                             + SH_C3_2 * (4.f * zz - xx - 3.f * yy) * s[11] + SH_C3_3 * (-6.f * y * z) * s[12]
                             + SH_C3_4 * (-2.f * x * y) * s[13] + SH_C3_5 * (-2.f * y * z) * s[14]
                             + SH_C3_6 * (-6.f * x * y) * s[15];
                        ddz += SH_C3_1 * (x * y) * s[10] + SH_C3_2 * (8.f * y * z) * s[11]
                             + SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy) * s[12] + SH_C3_4 * (8.f * x * z) * s[13]
                             + SH_C3_5 * (xx - yy) * s[14];
                    }
                }
                // through the normalisation: (I - d d^T)/|p| applied to (ddx,ddy,ddz)
                const float dot = x * ddx + y * ddy + z * ddz;
                out.dm[0] += (ddx - x * dot) * inv; out.dm[1] += (ddy - y * dot) * inv; out.dm[2] += (ddz - z * dot) * inv;
            }
            // dL/dSH: write this lane's row (the staged input row is dead now)
            if (sh_reg) { out.dsh[0] = B[0] * gr; out.dsh[1] = B[0] * gg; out.dsh[2] = B[0] * gb; }
            float* o = stage ? myrow_out : (dL_dshs + (size_t)idx * rowlen);
            const bool add = stage ? lds_accum : (accumulate != 0);     // (K == 1 without staging: straight to HBM)
#pragma unroll
            for (int k = 0; k < 16 && !sh_reg; ++k) {
                if (k < K) {
                    const float bk = (k < nb) ? B[k] : 0.f;
                    if (add) { o[3 * k] = o[3 * k] + bk * gr; o[3 * k + 1] = o[3 * k + 1] + bk * gg; o[3 * k + 2] = o[3 * k + 2] + bk * gb; }   // old + new
                    else { o[3 * k] = bk * gr; o[3 * k + 1] = bk * gg; o[3 * k + 2] = bk * gb; }
                }
            }
            if (!sh_reg && !add) for (int e = 48; e < rowlen; ++e) o[e] = 0.f;   // K > 16: inactive coefficients
        }

        // ---- conic -> cov2D -> (Sigma, t) -------------------------------------------
        Cov3 S;
        float R[9];
        float3 s = make_float3(0.f, 0.f, 0.f);
        float4 q = make_float4(1.f, 0.f, 0.f, 0.f);
        float3 s_act = make_float3(1.f, 1.f, 1.f);
        float q_inv_norm = 1.f;
        if (cov3D_precomp) {
            const float* c = cov3D_precomp + 6 * (size_t)idx;
            S.c0 = c[0]; S.c1 = c[1]; S.c2 = c[2]; S.c3 = c[3]; S.c4 = c[4]; S.c5 = c[5];
        } else {
            q = in.q;
            s.x = in.sx; s.y = in.sy; s.z = in.sz;
            if (RAW) { q = act_normalize(q, &q_inv_norm); s.x = __expf(s.x); s.y = __expf(s.y); s.z = __expf(s.z); }
            s_act = s;
            s.x *= vc.scale_modifier; s.y *= vc.scale_modifier; s.z *= vc.scale_modifier;
            quat_to_R(q, R);
            S = cov3d_from_scale_rot(s, R);
        }
        const Proj2D pj = project_cov(vc, V, pv, S);
        const float a = pj.a, b = pj.b, c = pj.c;
        const float det = a * c - b * b;
        const float d2i = 1.f / (det * det + 0.0000001f);   // the external package's guard (denom2inv); det >= 0.09 for PSD covariances
        const float gA = g[2], gB = g[3], gC = g[4];
        const float dLa = d2i * (-c * c * gA + b * c * gB - b * b * gC);
        const float dLb = d2i * (2.f * b * c * gA - (a * c + b * b) * gB + 2.f * a * b * gC);
        const float dLc = d2i * (-b * b * gA + a * b * gB - a * a * gC);

        // G = dLa T0^T T0 + dLb sym(T0^T T1) + dLc T1^T T1   (gradient w.r.t. Sigma entries)
        const float* T0 = pj.T0; const float* T1 = pj.T1;
        float Gm[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                Gm[3 * i + j] = dLa * T0[i] * T0[j] + 0.5f * dLb * (T0[i] * T1[j] + T1[i] * T0[j]) + dLc * T1[i] * T1[j];
        if (cov3D_precomp) {
            out.dcov[0] = Gm[0]; out.dcov[1] = 2.f * Gm[1]; out.dcov[2] = 2.f * Gm[2];
            out.dcov[3] = Gm[4]; out.dcov[4] = 2.f * Gm[5]; out.dcov[5] = Gm[8];
        } else {
            // Sigma = M M^T, M = R diag(s): dL/dM = 2 G M
            float M[9], dM[9];
#pragma unroll
            for (int i = 0; i < 3; ++i) { M[3 * i] = R[3 * i] * s.x; M[3 * i + 1] = R[3 * i + 1] * s.y; M[3 * i + 2] = R[3 * i + 2] * s.z; }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    dM[3 * i + j] = 2.f * (Gm[3 * i] * M[j] + Gm[3 * i + 1] * M[3 + j] + Gm[3 * i + 2] * M[6 + j]);
            const float sv[3] = {s.x, s.y, s.z};
            float dR[9];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                out.dsc[j] = vc.scale_modifier * (dM[j] * R[j] + dM[3 + j] * R[3 + j] + dM[6 + j] * R[6 + j]);
#pragma unroll
                for (int i = 0; i < 3; ++i) dR[3 * i + j] = dM[3 * i + j] * sv[j];
            }
            const float r = q.x, x = q.y, y = q.z, z = q.w;
            out.dq[0] = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
            out.dq[1] = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.f * x * dR[8]);
            out.dq[2] = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.f * y * dR[8]);
            out.dq[3] = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
            if (RAW) {                              // d exp and d normalise
                out.dsc[0] *= s_act.x; out.dsc[1] *= s_act.y; out.dsc[2] *= s_act.z;
                const float qd = q.x * out.dq[0] + q.y * out.dq[1] + q.z * out.dq[2] + q.w * out.dq[3];
                out.dq[0] = (out.dq[0] - q.x * qd) * q_inv_norm; out.dq[1] = (out.dq[1] - q.y * qd) * q_inv_norm;
                out.dq[2] = (out.dq[2] - q.z * qd) * q_inv_norm; out.dq[3] = (out.dq[3] - q.w * qd) * q_inv_norm;
            }
        }
        // dL/dT rows: dT0 = 2 dLa Sigma T0 + dLb Sigma T1 ; dT1 = 2 dLc Sigma T1 + dLb Sigma T0
        const float u0 = S.c0 * T0[0] + S.c1 * T0[1] + S.c2 * T0[2];
        const float u1 = S.c1 * T0[0] + S.c3 * T0[1] + S.c4 * T0[2];
        const float u2 = S.c2 * T0[0] + S.c4 * T0[1] + S.c5 * T0[2];
        const float w0 = S.c0 * T1[0] + S.c1 * T1[1] + S.c2 * T1[2];
        const float w1 = S.c1 * T1[0] + S.c3 * T1[1] + S.c4 * T1[2];
        const float w2 = S.c2 * T1[0] + S.c4 * T1[1] + S.c5 * T1[2];
        const float dT0[3] = {2.f * dLa * u0 + dLb * w0, 2.f * dLa * u1 + dLb * w1, 2.f * dLa * u2 + dLb * w2};
        const float dT1[3] = {2.f * dLc * w0 + dLb * u0, 2.f * dLc * w1 + dLb * u1, 2.f * dLc * w2 + dLb * u2};
        // T = J Wr, Wr[i][k] = V[4k+i]:  dJ[r][i] = sum_k dT[r][k] * V[4k+i]
        float dJ00 = 0.f, dJ02 = 0.f, dJ11 = 0.f, dJ12 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            dJ00 += dT0[k] * V[4 * k + 0]; dJ02 += dT0[k] * V[4 * k + 2];
            dJ11 += dT1[k] * V[4 * k + 1]; dJ12 += dT1[k] * V[4 * k + 2];
        }
        const float tz = 1.f / pj.tz, tz2 = tz * tz, tz3 = tz2 * tz;
        const float dtx = pj.clampx ? 0.f : -vc.focal_x * tz2 * dJ02;
        const float dty = pj.clampy ? 0.f : -vc.focal_y * tz2 * dJ12;
        const float dtz = -vc.focal_x * tz2 * dJ00 - vc.focal_y * tz2 * dJ11
                        + (2.f * vc.focal_x * pj.tx) * tz3 * dJ02 + (2.f * vc.focal_y * pj.ty) * tz3 * dJ12;
        // t_i = sum_k m_k V[4k+i]
#pragma unroll
        for (int k = 0; k < 3; ++k) out.dm[k] += V[4 * k] * dtx + V[4 * k + 1] * dty + V[4 * k + 2] * dtz;
    } else if (idx < N && use_sh && !sh_reg) {
        if (stage) { if (!lds_accum && !sparse) for (int e = 0; e < rowlen; ++e) myrow_out[e] = 0.f; }
        else if (!accumulate) { float* o = dL_dshs + (size_t)idx * rowlen; for (int e = 0; e < rowlen; ++e) o[e] = 0.f; }
    }

}

template <bool RAW, bool MULTI>   // RAW: inputs are the raw parameters (fused sigmoid / exp / normalise backward); MULTI: B > 1
__global__ void __launch_bounds__(256)
gsr_preprocess_bwd(ViewTab tab, int first_view, int B /* B == 1: the view tab.v[0], per-view arrays already offset by the host;
                                                           B > 1: views 0 .. B - 1 of the table, first_view = 0 */, int N, int K,
                   const float* __restrict__ means3D, const float* __restrict__ shs,
                   const float* __restrict__ shs_rest /* split layout (GsrView.shs_rest) or NULL */,
                   float* __restrict__ dL_dshs_rest,
                   const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
                   const float* __restrict__ scales, const float* __restrict__ rotations,
                   const float* __restrict__ cov3D_precomp, const int32_t* __restrict__ radii /* [views][N] */,
                   const uint8_t* __restrict__ flags8 /* [views][N] */, const float* __restrict__ g2d /* [views][N][12] */,
                   float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dmeans2D /* [views][N][3] */,
                   float* __restrict__ dL_dshs, float* __restrict__ dL_dcolors,
                   float* __restrict__ dL_dopac, float* __restrict__ dL_dscales,
                   float* __restrict__ dL_drots, float* __restrict__ dL_dcov3D,
                   int accumulate /* add to the parameter gradients instead of overwriting (views after the first;
                                     dL_dmeans2D is per view and always overwritten) */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pp[];
    float* shbuf = reinterpret_cast<float*>(smem_pp);
    const int rowlen = 3 * K;
    const bool use_sh = (shs != nullptr);
    const bool stage = use_sh && (K > 1);
    // Several views in one launch (B > 1): every Gaussian is read ONCE and the cameras are run through in registers, last
    // view first and `earlier sum + this view` after that -- the additions of B separate launches (= the order autograd
    // accumulates B separate calls) in the same order, without B - 1 read-modify-write passes over the gradients. K == 1
    // (DreamGaussian's sh_degree 0) or precomputed colours: nothing is staged, dL/dSH is three registers. K > 1 (round 4): the SH
    // rows are staged ONCE into the first half of the LDS buffer and stay there for all views; dL/dSH accumulates in a second
    // row per Gaussian (the input row can no longer double as the output row) and leaves through one stage_rows_out -- the
    // 8-view chain at SH degree 3 used to be 8 launches x 524 MB. The cameras go from the by-value table to LDS.
    __shared__ ViewConst sv[MULTI ? GSR_MAX_VIEWS : 1];
    // The cameras' matrices sit in LDS (view[16] | proj[16] | campos[3] per view): read through their device pointers they come
    // back as vector-memory loads, each with a wait for memory right behind it in the middle of the arithmetic.
    __shared__ __attribute__((aligned(16))) float camf[MULTI ? GSR_MAX_VIEWS : 1][36];
    if (MULTI) {
        if (threadIdx.x == 0) {
#pragma unroll
            for (int v = 0; v < GSR_MAX_VIEWS; ++v) sv[MULTI ? v : 0] = tab.v[v];
        }
    }
    for (int e = threadIdx.x; e < 36 * (MULTI ? B : 1); e += blockDim.x) {
        const int v = e / 36, c = e - v * 36;
        const ViewConst& cv = tab.v[MULTI ? first_view + v : 0];
        if (c < 35) camf[MULTI ? v : 0][c] = c < 16 ? cv.view[cam_index(c, cv.mat_t & 1)] : (c < 32 ? cv.proj[cam_index(c - 16, cv.mat_t & 2)] : cv.campos[c - 32]);
    }
    lds_barrier();
    __shared__ uint8_t rowlive_buf[2][256];               // [batch parity][row]: a batch's stage_rows_out may still read while the next one writes
    int parity = 0;
    for (int base = blockIdx.x * blockDim.x; base < N; base += gridDim.x * blockDim.x, parity ^= 1) {
        uint8_t* rowlive = rowlive_buf[parity];
        const int cnt = min((int)blockDim.x, N - base);
        const int idx = base + threadIdx.x;
        // this batch's per-Gaussian inputs: requested first, branch-free (index clamped), in flight while the SH rows are staged
        const size_t ic = (size_t)min(idx, N - 1);
        K6In in;
        {
            // (12-byte vector loads and stores throughout: one request per lane and array instead of three 4-byte ones 12 bytes apart)
            const gsr_f3 m = *reinterpret_cast<const gsr_f3u*>(means3D + 3 * ic);
            in.mx = m.x; in.my = m.y; in.mz = m.z;
            in.op = opacities[ic];
            in.q = make_float4(1.f, 0.f, 0.f, 0.f); in.sx = in.sy = in.sz = 0.f;
            if (!cov3D_precomp) {                         // (uniform)
                in.q = reinterpret_cast<const float4*>(rotations)[ic];
                const gsr_f3 sc3 = *reinterpret_cast<const gsr_f3u*>(scales + 3 * ic);
                in.sx = sc3.x; in.sy = sc3.y; in.sz = sc3.z;
            }
        }
        K6ViewIn vin0;
        int32_t radius0;
        {
            const size_t r0 = (size_t)(MULTI ? first_view + B - 1 : 0) * N + ic;     // the first view of the loop below
            radius0 = radii[r0];
            vin0.flags = flags8[r0];
            const float4* gp = reinterpret_cast<const float4*>(g2d + r0 * GSR_G2D_STRIDE);
            vin0.g0 = gp[0]; vin0.g1 = gp[1]; vin0.g2 = gp[2];
        }
        // A Gaussian no pixel gradient reached (hidden behind the stop, or outside every blended pixel's support: 73 % of the 1M blob,
        // 94 % of the trained-like scene) has an all-zero row of 2D gradients: every output of this kernel is zero for it, and its SH
        // row -- 192 of the 524 bytes the kernel moves per Gaussian at degree 3 -- is not read at all.
        const bool touched = ((__float_as_uint(vin0.g0.x) | __float_as_uint(vin0.g0.y) | __float_as_uint(vin0.g0.z) | __float_as_uint(vin0.g0.w) |
                               __float_as_uint(vin0.g1.x) | __float_as_uint(vin0.g1.y) | __float_as_uint(vin0.g1.z) | __float_as_uint(vin0.g1.w) |
                               __float_as_uint(vin0.g2.x) | __float_as_uint(vin0.g2.y)) & 0x7fffffffu) != 0u;
        if (stage) {
            // (several views: a row is fetched if the Gaussian exists -- which views touch it is known view by view, below)
            rowlive[threadIdx.x] = (idx < N && (MULTI || (radius0 > 0 && touched))) ? 1 : 0;     // (the previous batch's stage_rows_out reads the other copy)
            lds_barrier();
            if (shs_rest) stage_rows_in_split(shs + (size_t)base * 3, shs_rest + (size_t)base * (rowlen - 3), shbuf, cnt, rowlen, rowlive);
            else stage_rows_in(shs + (size_t)base * rowlen, shbuf, cnt, rowlen, rowlive);
            lds_barrier();
        }
        K6Out out, cur;
        float tsh[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 3; ++e) { out.dm[e] = 0.f; out.dsc[e] = 0.f; out.dcol[e] = 0.f; }
        out.dop = 0.f; out.dm2[0] = out.dm2[1] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) out.dq[e] = 0.f;
#pragma unroll
        for (int e = 0; e < 6; ++e) out.dcov[e] = 0.f;
        float* myrow = shbuf + threadIdx.x * (rowlen + 1);
        float* shbuf_out = (MULTI && stage) ? shbuf + blockDim.x * (rowlen + 1) : shbuf;     // several views: input rows stay, output rows apart
        float* myrow_out = shbuf_out + threadIdx.x * (rowlen + 1);
        const bool sh_in_regs = MULTI && !stage;          // (then K == 1: three numbers per view)
        for (int v = MULTI ? first_view + B - 1 : 0; v >= (MULTI ? first_view : 0); --v) {
            // one view per launch (the common case): the camera stays a kernel argument = scalar registers; several: from LDS,
            // made wave-uniform again from lane 0's copy
            ViewConst vc = tab.v[0];
            K6ViewIn vin = vin0;
            int32_t radius = radius0;
            if (MULTI) {
                vc = sv[MULTI ? v : 0];
                vc.W = __builtin_amdgcn_readfirstlane(vc.W); vc.H = __builtin_amdgcn_readfirstlane(vc.H);
                vc.sh_degree = __builtin_amdgcn_readfirstlane(vc.sh_degree);
                if (v != first_view + B - 1) {            // (uniform) the views after the first of the loop
                    const size_t r = (size_t)v * N + ic;
                    radius = radii[r];
                    vin.flags = flags8[r];
                    const float4* gp = reinterpret_cast<const float4*>(g2d + r * GSR_G2D_STRIDE);
                    vin.g0 = gp[0]; vin.g1 = gp[1]; vin.g2 = gp[2];
                }
            }
            const bool live = (idx < N) && (radius > 0) && (MULTI || touched);   // (MULTI: nothing is staged, the views differ)
            k6_gaussian<RAW>(vc, camf[MULTI ? v - first_view : 0], idx, N, K, live, in, vin, shs, cov3D_precomp, dL_dshs, stage, myrow, myrow_out,
                             MULTI && v != first_view + B - 1, accumulate, sh_in_regs, cur);
            if (idx < N) { const gsr_f3 m2 = {cur.dm2[0], cur.dm2[1], 0.f}; __builtin_nontemporal_store(m2, reinterpret_cast<gsr_f3u*>(dL_dmeans2D + ((size_t)v * N + idx) * 3)); }
            // first pass (the last view): 0 + x = x exactly
#pragma unroll
            for (int e = 0; e < 3; ++e) { out.dm[e] = out.dm[e] + cur.dm[e]; out.dsc[e] = out.dsc[e] + cur.dsc[e]; out.dcol[e] = out.dcol[e] + cur.dcol[e]; tsh[e] = tsh[e] + cur.dsh[e]; }
            out.dop = out.dop + cur.dop;
#pragma unroll
            for (int e = 0; e < 4; ++e) out.dq[e] = out.dq[e] + cur.dq[e];
#pragma unroll
            for (int e = 0; e < 6; ++e) out.dcov[e] = out.dcov[e] + cur.dcov[e];
        }
        if (sh_in_regs && use_sh && idx < N) { float* o = dL_dshs + (size_t)idx * rowlen; o[0] = tsh[0]; o[1] = tsh[1]; o[2] = tsh[2]; }

        if (idx < N) {
            if (accumulate) {   // old + new, the order autograd's AccumulateGrad adds a later view's gradient in
                const gsr_f3 om = *reinterpret_cast<const gsr_f3u*>(dL_dmeans3D + 3 * (size_t)idx);
                out.dm[0] = om.x + out.dm[0]; out.dm[1] = om.y + out.dm[1]; out.dm[2] = om.z + out.dm[2];
                out.dop = dL_dopac[idx] + out.dop;
                if (dL_dcolors) { out.dcol[0] = dL_dcolors[3 * idx] + out.dcol[0]; out.dcol[1] = dL_dcolors[3 * idx + 1] + out.dcol[1]; out.dcol[2] = dL_dcolors[3 * idx + 2] + out.dcol[2]; }
                if (dL_dcov3D) {
#pragma unroll
                    for (int e = 0; e < 6; ++e) out.dcov[e] = dL_dcov3D[6 * (size_t)idx + e] + out.dcov[e];
                }
                if (dL_dscales) { const gsr_f3 os = *reinterpret_cast<const gsr_f3u*>(dL_dscales + 3 * (size_t)idx); out.dsc[0] = os.x + out.dsc[0]; out.dsc[1] = os.y + out.dsc[1]; out.dsc[2] = os.z + out.dsc[2]; }
                if (dL_drots) { const float4 o = reinterpret_cast<const float4*>(dL_drots)[idx]; out.dq[0] = o.x + out.dq[0]; out.dq[1] = o.y + out.dq[1]; out.dq[2] = o.z + out.dq[2]; out.dq[3] = o.w + out.dq[3]; }
            }
            { const gsr_f3 o3 = {out.dm[0], out.dm[1], out.dm[2]}; __builtin_nontemporal_store(o3, reinterpret_cast<gsr_f3u*>(dL_dmeans3D + 3 * (size_t)idx)); }
            __builtin_nontemporal_store(out.dop, dL_dopac + idx);
            if (dL_dcolors) { const gsr_f3 o3 = {out.dcol[0], out.dcol[1], out.dcol[2]}; *reinterpret_cast<gsr_f3u*>(dL_dcolors + 3 * (size_t)idx) = o3; }
            if (dL_dcov3D) {
#pragma unroll
                for (int e = 0; e < 6; ++e) dL_dcov3D[6 * (size_t)idx + e] = out.dcov[e];
            }
            if (dL_dscales) { const gsr_f3 o3 = {out.dsc[0], out.dsc[1], out.dsc[2]}; __builtin_nontemporal_store(o3, reinterpret_cast<gsr_f3u*>(dL_dscales + 3 * (size_t)idx)); }
            if (dL_drots) store_once(reinterpret_cast<float4*>(dL_drots) + idx, out.dq[0], out.dq[1], out.dq[2], out.dq[3]);
        }
        if (stage) {
            lds_barrier();
            if (shs_rest) stage_rows_out_split(dL_dshs + (size_t)base * 3, dL_dshs_rest + (size_t)base * (rowlen - 3), shbuf_out, cnt, rowlen, accumulate, rowlive);
            else stage_rows_out(dL_dshs + (size_t)base * rowlen, shbuf_out, cnt, rowlen, accumulate, rowlive);
        }
    }
}

template __global__ void gsr_preprocess_bwd<false, false>(ViewTab, int, int, int, int, const float*, const float*, const float*, float*, const float*, const float*, const float*, const float*, const float*, const int32_t*, const uint8_t*, const float*, float*, float*, float*, float*, float*, float*, float*, float*, int);
template __global__ void gsr_preprocess_bwd<true, false>(ViewTab, int, int, int, int, const float*, const float*, const float*, float*, const float*, const float*, const float*, const float*, const float*, const int32_t*, const uint8_t*, const float*, float*, float*, float*, float*, float*, float*, float*, float*, int);

// ---------------------------------------------------------------------------------------
// K6c: preprocess backward for ONE view, over the Gaussians that HAVE a gradient only.
//
// In a dense scene a quarter of the Gaussians are reached by a pixel gradient (27 % of the 1M blob, 6 % of the trained-like scene:
// the rest is hidden behind the stop or never reaches alpha 1/255), in an order that is random against the index.
// gsr_preprocess_bwd above -- lane = Gaussian -- streams every Gaussian, runs its ~1 900 vector instructions per wave for every wave
// that holds ONE live lane (3.0e7 wave-instructions = 43 us of a 90 us kernel at the rate its mix can issue) and stores 248 bytes
// per Gaussian, three quarters of them zeros. Here nothing is streamed:
//   * gsr_render_bwd_q2 sets one byte per Gaussian it adds a non-zero row to (`live`, cleared with the accumulators) and clears
//     every gradient array on the side (ZeroRegions: a slice of zeros per workgroup, under an issue-bound kernel);
//   A  a persistent 128-thread workgroup turns 256 flags per round (two per lane) into a ring of live indices in LDS (wave scan);
//   B  as soon as the ring holds 128: lane = entry -- inputs, gradient row and SH row of the ENTRY (gathers), the arithmetic of
//      k6_gaussian with every lane busy, the entry's outputs. What is left in the ring is flushed at the end.
// Same arithmetic per Gaussian as gsr_preprocess_bwd<RAW, false> (one function). A Gaussian without a bit is not touched at all:
// its gradients are the zeros the compositing kernel stored.
// Not here (the host falls back to the kernel above): several views per launch, accumulation onto an earlier view's result, the
// split SH layout, gradient arrays that are not 16-byte aligned, a backward without the compositing kernel (no instances).
// dynamic LDS: GSR_K6C_NT * (3K + 1) floats when SH rows are staged (K > 1).
// ---------------------------------------------------------------------------------------
#define GSR_K6C_NT 128                                    // threads per workgroup = queue entries per round of phase B
#define GSR_K6C_PER 2                                     // Gaussians (flag bytes) per lane and round of phase A
#define GSR_K6C_ROUND (GSR_K6C_NT * GSR_K6C_PER)
#define GSR_K6C_RING (GSR_K6C_NT + GSR_K6C_ROUND)         // ring of live indices: < NT waiting + one round of phase A (1.5 KiB: with
                                                          // the 25 KiB of staged rows at 16 coefficients six workgroups share a CU)
static_assert(GSR_K6C_NT == 128 && GSR_K6C_PER == 2, "phase A of gsr_preprocess_bwd_compact: four quarters of 32 lanes x 2 flags");
__device__ __forceinline__ uint32_t k6c_slot(uint32_t head, uint32_t off) { const uint32_t p = head + off; return p >= GSR_K6C_RING ? p - GSR_K6C_RING : p; }   // head, off < RING
template <bool RAW>
__global__ void __launch_bounds__(GSR_K6C_NT, 4)      // <= 128 VGPRs (0.0688 -> 0.0614 ms at 1M against three waves per SIMD and 130 VGPRs)
gsr_preprocess_bwd_compact(ViewTab tab /* the view: tab.v[0] */, int N, int K,
                           const float* __restrict__ means3D, const float* __restrict__ shs,
                           const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
                           const float* __restrict__ scales, const float* __restrict__ rotations,
                           const float* __restrict__ cov3D_precomp,
                           const uint8_t* __restrict__ flags8, const float* __restrict__ g2d /* [N][12] */,
                           const uint8_t* __restrict__ live /* [GSR_LIVE_BYTES(N)] */,
                           float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dmeans2D /* [N][3] */,
                           float* __restrict__ dL_dshs, float* __restrict__ dL_dcolors,
                           float* __restrict__ dL_dopac, float* __restrict__ dL_dscales,
                           float* __restrict__ dL_drots, float* __restrict__ dL_dcov3D) {
    constexpr uint32_t NT = GSR_K6C_NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pp[];
    float* shbuf = reinterpret_cast<float*>(smem_pp);
    const int rowlen = 3 * K, pitch = rowlen + 1;
    const bool use_sh = (shs != nullptr);
    const bool stage = use_sh && (K > 1);
    const ViewConst vc = tab.v[0];
    __shared__ __attribute__((aligned(16))) float camf[36];
    __shared__ uint32_t ring[GSR_K6C_RING];               // indices of live Gaussians
    __shared__ uint32_t wcnt[NT / 64];
    if (threadIdx.x < 35) camf[threadIdx.x] = threadIdx.x < 16 ? vc.view[cam_index(threadIdx.x, vc.mat_t & 1)]
                                              : (threadIdx.x < 32 ? vc.proj[cam_index(threadIdx.x - 16, vc.mat_t & 2)] : vc.campos[threadIdx.x - 32]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t head = 0, qn = 0;                            // (uniform) ring[head ...], wrapping: qn entries waiting
    // Which Gaussians a workgroup looks at: 64 consecutive ones (32 lanes x 2 flags) per QUARTER of a round, the quarters of all
    // workgroups interleaved -- sub-chunk s = (4 round + quarter) * grid + workgroup. Rounds 2-5 handed out 256 consecutive Gaussians per
    // round (2.5 rounds per workgroup at 1M): in a random order every chunk holds its 27 % of live ones, in Z-order (reorder_gaussians)
    // a chunk is all live or all dead and the workgroups' loads spread 0 .. 3x -- the kernel got SLOWER under the order that helps
    // every other kernel (0.0665 -> 0.0697 ms blob, 0.0275 -> 0.0416 trained-like, profiles/r05_bench_order.jsonl). A quarter keeps 64
    // neighbours together (they share the lines of their 12- and 4-byte inputs); a draw from ONE atomic counter was measured 3x slower.
    const int G = (int)gridDim.x;
    const int nsub = (N + 63) >> 6;
    int r4 = 0;                                           // (uniform) 4 * rounds of phase A taken so far
    // the first round's flags travel while the camera is staged; every later round's are requested a round ahead
    // (GSR_K6C_PER = 2 flags = one aligned 16-bit load per lane: the array is padded to a multiple of 16 bytes)
    const unsigned short* live16 = reinterpret_cast<const unsigned short*>(live);
    const int npairs = (int)(GSR_LIVE_BYTES(N) / 2);
    const int quarter = (int)(threadIdx.x >> 5), l31 = (int)(threadIdx.x & 31);
    auto pair_of = [&](int r4_) -> int {                  // this lane's pair of flags in the round that starts at quarter r4_ (may lie past the end)
        const long long sub = (long long)(r4_ + quarter) * G + (int)blockIdx.x;
        return sub < (long long)nsub ? (int)sub * 32 + l31 : npairs;
    };
    uint32_t next_flags = live16[min(pair_of(0), npairs - 1)];
    lds_barrier();
    for (;;) {
        const bool more = (long long)r4 * G + (int)blockIdx.x < (long long)nsub && qn < NT;   // (uniform) a round of phase A only when phase B has nothing full to do
        if (more) {
            // ---- phase A: this round's flags -> ring entries (lane = GSR_K6C_PER consecutive Gaussians)
            uint32_t f = next_flags;
            const int pr = pair_of(r4);
            if (pr >= npairs) f = 0u;                     // (quarters / pairs past the end; bytes past N inside the padding are never set)
            next_flags = live16[min(pair_of(r4 + 4), npairs - 1)];
            const int i0 = pr * GSR_K6C_PER;
            const bool l0 = (f & 0xffu) != 0u, l1 = (f >> 8) != 0u;
            const uint32_t c = (uint32_t)l0 + (uint32_t)l1;
            uint32_t incl = c;                            // inclusive scan of the counts over the wave
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += u; }
            if (lane == 63) wcnt[wave] = incl;
            lds_barrier();
            uint32_t off = incl - c, tot = 0;
#pragma unroll
            for (int w = 0; w < (int)(NT / 64); ++w) { const uint32_t x = wcnt[w]; off += w < wave ? x : 0u; tot += x; }
            if (l0) { ring[k6c_slot(head, qn + off)] = (uint32_t)i0; ++off; }
            if (l1) ring[k6c_slot(head, qn + off)] = (uint32_t)(i0 + 1);
            lds_barrier();                                // the ring's new entries (and wcnt free again)
            qn += tot;
            r4 += 4;
        }
        const bool last = (long long)r4 * G + (int)blockIdx.x >= (long long)nsub;   // (uniform) nothing left to look at: the rest of the ring goes out as it is
        const uint32_t take = qn >= NT ? NT : (last ? qn : 0u);    // a full round of entries, or the rest at the very end
        if (take == 0u) { if (last) break; continue; }
        // ---- phase B: `take` queue entries, lane = entry
        const bool mine = threadIdx.x < take;
        const uint32_t e_idx = ring[k6c_slot(head, min(threadIdx.x, take - 1u))];   // (lanes past the end: a valid index, unused)
        K6In in;
        K6ViewIn vin;
        {
            const size_t ic = (size_t)e_idx;
            const gsr_f3 m = *reinterpret_cast<const gsr_f3u*>(means3D + 3 * ic);
            in.mx = m.x; in.my = m.y; in.mz = m.z;
            in.op = opacities[ic];
            in.q = make_float4(1.f, 0.f, 0.f, 0.f); in.sx = in.sy = in.sz = 0.f;
            if (!cov3D_precomp) {                         // (uniform)
                in.q = reinterpret_cast<const float4*>(rotations)[ic];
                const gsr_f3 sc3 = *reinterpret_cast<const gsr_f3u*>(scales + 3 * ic);
                in.sx = sc3.x; in.sy = sc3.y; in.sz = sc3.z;
            }
            vin.flags = flags8[ic];
            const float4* gp = reinterpret_cast<const float4*>(g2d + ic * GSR_G2D_STRIDE);
            vin.g0 = gp[0]; vin.g1 = gp[1]; vin.g2 = gp[2];
        }
        if (stage) {
            // the entries' SH rows: lane = 16 bytes of a row (a row is rowlen * 4 contiguous bytes), twelve requests in flight per lane
            if ((rowlen & 3) == 0) {
                constexpr int STAGE_U = 12;
                const int nvec = (int)take * rowlen / 4;
                for (int i0 = threadIdx.x; i0 < nvec; i0 += NT * STAGE_U) {
                    float4 v[STAGE_U];
#pragma unroll
                    for (int u = 0; u < STAGE_U; ++u) {
                        const int i = min(i0 + u * NT, nvec - 1);
                        const int e = 4 * i, row = e / rowlen, col = e - row * rowlen;
                        v[u] = *reinterpret_cast<const float4*>(shs + (size_t)ring[k6c_slot(head, (uint32_t)row)] * rowlen + col);
                    }
#pragma unroll
                    for (int u = 0; u < STAGE_U; ++u) {
                        const int i = i0 + u * NT;
                        if (i < nvec) {
                            const int e = 4 * i, row = e / rowlen, col = e - row * rowlen;
                            float* d = shbuf + row * pitch + col;
                            d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                        }
                    }
                }
            } else {
                for (int e = threadIdx.x; e < (int)take * rowlen; e += NT) {
                    const int row = e / rowlen, col = e - row * rowlen;
                    shbuf[row * pitch + col] = shs[(size_t)ring[k6c_slot(head, (uint32_t)row)] * rowlen + col];
                }
            }
            lds_barrier();
        }
        K6Out cur;
        float* myrow = shbuf + threadIdx.x * pitch;
        k6_gaussian<RAW>(vc, camf, mine ? (int)e_idx : N, N, K, mine, in, vin, shs, cov3D_precomp, dL_dshs, stage, myrow, myrow,
                         false, 0, false, cur, true);
        if (mine) {
            const size_t o = (size_t)e_idx;
            { const gsr_f3 o3 = {cur.dm[0], cur.dm[1], cur.dm[2]}; __builtin_nontemporal_store(o3, reinterpret_cast<gsr_f3u*>(dL_dmeans3D + 3 * o)); }
            { const gsr_f3 m2 = {cur.dm2[0], cur.dm2[1], 0.f}; __builtin_nontemporal_store(m2, reinterpret_cast<gsr_f3u*>(dL_dmeans2D + 3 * o)); }
            __builtin_nontemporal_store(cur.dop, dL_dopac + o);
            if (dL_dcolors) { const gsr_f3 o3 = {cur.dcol[0], cur.dcol[1], cur.dcol[2]}; *reinterpret_cast<gsr_f3u*>(dL_dcolors + 3 * o) = o3; }
            if (dL_dcov3D) {
#pragma unroll
                for (int e = 0; e < 6; ++e) dL_dcov3D[6 * o + e] = cur.dcov[e];
            }
            if (dL_dscales) { const gsr_f3 o3 = {cur.dsc[0], cur.dsc[1], cur.dsc[2]}; __builtin_nontemporal_store(o3, reinterpret_cast<gsr_f3u*>(dL_dscales + 3 * o)); }
            if (dL_drots) store_once(reinterpret_cast<float4*>(dL_drots) + o, cur.dq[0], cur.dq[1], cur.dq[2], cur.dq[3]);
        }
        if (stage) {
            lds_barrier();
            if ((rowlen & 3) == 0) {
                for (int i = threadIdx.x; i < (int)take * rowlen / 4; i += NT) {
                    const int e = 4 * i, row = e / rowlen, col = e - row * rowlen;
                    const float* sp = shbuf + row * pitch + col;
                    store_once(reinterpret_cast<float4*>(dL_dshs + (size_t)ring[k6c_slot(head, (uint32_t)row)] * rowlen + col), sp[0], sp[1], sp[2], sp[3]);
                }
            } else {
                for (int e = threadIdx.x; e < (int)take * rowlen; e += NT) {
                    const int row = e / rowlen, col = e - row * rowlen;
                    dL_dshs[(size_t)ring[k6c_slot(head, (uint32_t)row)] * rowlen + col] = shbuf[row * pitch + col];
                }
            }
        }
        lds_barrier();                                    // the rows and the ring's slots are free again
        head = k6c_slot(head, take); qn -= take;
    }
}
template __global__ void gsr_preprocess_bwd_compact<false>(ViewTab, int, int, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const uint8_t*, const float*, const uint8_t*, float*, float*, float*, float*, float*, float*, float*, float*);
template __global__ void gsr_preprocess_bwd_compact<true>(ViewTab, int, int, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const uint8_t*, const float*, const uint8_t*, float*, float*, float*, float*, float*, float*, float*, float*);

// visible[i] = view-space z > 0.2  (frustum rule of A.3)
extern "C" __global__ void __launch_bounds__(256)
gsr_mark_visible_kernel(const float* __restrict__ Vg, int transposed, int N, const float* __restrict__ means3D,
                        uint8_t* __restrict__ visible) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N) return;
    float V[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) V[i] = (i == 2 || i == 6 || i == 10 || i == 14) ? Vg[cam_index(i, transposed)] : 0.f;
    const float z = view_depth(V, means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    visible[idx] = z > 0.2f ? 1 : 0;
}
