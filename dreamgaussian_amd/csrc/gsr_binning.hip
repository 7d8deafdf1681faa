// gsr_binning.hip -- tile binning and per-tile depth sort (K2, K3, K4).
//
// Replaces (behaviour, not code) the duplicate-with-keys / global radix sort / range
// identification stage of the external rasterizer: order inside a
// tile = ascending fp32 depth bits, ties in ascending Gaussian index.
//
// MI355X design: no global M-element sort. Per-tile instance counts come out of K1's LDS
// histograms; K2 scans them into segment offsets; K3 scatters 8-byte (depth,id) keys into the
// tile segments through an LDS-privatised cursor reservation (one returning global atomic per
// (workgroup,tile) instead of one per instance); K4 sorts each segment inside LDS (counting pass
// into ~n/4 depth buckets + rank inside each bucket on the 64-bit key; rank sort for lists of <= 256;
// bitonic network as the skew fallback; a list longer than the 8 192-key buffer is streamed through it in depth-ordered portions) and writes the tile's sorted
// list of Gaussian INDICES; the compositing kernels gather the 64-byte records themselves.
#include "gsr_device.h"

// ---------------------------------------------------------------------------------------
// K2: exclusive scan of the per-tile counts, K1's statistics, and the DEPTH-MAJOR work list of the segment
// forward (single workgroup; T is a few thousand).
// counters[0] = M_ref, [1] = V, [2] = M_emit, [3] = max per-tile count, [4] = backward work items (written by
// gsr_render_fwd_combine), [5] = bits of max(colour, depth), [6] = list capacity (gsr_scatter), [7] = levels | seg_shift << 32,
// [8 + 2v], [9 + 2v] = M_ref, V of view v.
//
// Work list. Tile t has nseg_t = ceil(n_t / 2^seg_shift) segments. `order` = the tiles sorted by
// class = min(nseg, GSR_NLEV) DESCENDING (exact counting sort: one class per value), so the tiles that have a
// segment c (< GSR_NLEV) are exactly order[0 .. S_c) with S_c = #tiles of class > c, and item k of the launch is
// (level c, tile order[k - level_off[c]]) with level_off = exclusive scan of S: all first segments, then all second
// segments, ... -- no list is written, gsr_render_fwd_seg finds its level with two ballots over level_off[0..1023].
// level_off[c] = the item total for every c >= the number of levels.
// ---------------------------------------------------------------------------------------
#define GSR_NCLS (GSR_NLEV + 1)          // 1024 classes

// in-place exclusive scan of a[0 .. n) in LDS by the NT threads of the workgroup (wsum: NT / 64 words); returns the total
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan_lds(uint32_t* a, int n, uint32_t* wsum) {
    const int per = (n + NT - 1) / NT;
    const int beg = min((int)threadIdx.x * per, n), end = min(beg + per, n);
    uint32_t local = 0;
    for (int i = beg; i < end; ++i) local += a[i];
    uint32_t incl = local;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    __syncthreads();                                   // wsum may still be read by the previous scan
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { const uint32_t x = wsum[w]; all += x; base += w < wave ? x : 0u; }
    uint32_t run = base + incl - local;
    for (int i = beg; i < end; ++i) { const uint32_t c = a[i]; a[i] = run; run += c; }
    __syncthreads();
    return all;
}

// LDS of one scan workgroup (static in gsr_tile_scan; inside the dynamic block of gsr_scatter's scan workgroup)
struct TileScanLds {
    unsigned long long wsum[16], sref[16], svis[16], smax[16];
    uint32_t wsegs[16], wmax[16], scratch[16];
    uint32_t cls[GSR_NCLS], tmp[GSR_NCLS];
};

// The body of K2 for a workgroup of NT threads (1 024: the kernel of its own; 256: a workgroup of gsr_scatter, see there).
// `A`: T + 1 words of LDS, or NULL. With it the tile counts are read from HBM ONCE, coalesced and all requests in flight together,
// become the list starts in place, and every later phase (segment classes, the order) takes count and start from LDS; without it
// (tile grids beyond the buffer) each phase reads tile_count again and the order reads tile_off back from memory.
template <int NT>
__device__ __forceinline__ void tile_scan_body(TileScanLds& L, uint32_t* A, const uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_off, int T,
              unsigned long long* __restrict__ counters, uint32_t* __restrict__ tile_seg, int seg_shift,
              const unsigned long long* __restrict__ block_stats, int nblocks /* all views */, int nviews,
              uint32_t* __restrict__ order /* tiles by segment count, descending */, uint2* __restrict__ order_span /* (list start, length) of order[k] */,
              uint32_t* __restrict__ level_off /* [1024] */,
              unsigned long long* __restrict__ host_out /* pinned host memory (device-mapped): the first `host_words` counters land in
                                                           words 0.., the arrival flag in word `host_flag` */,
              int host_words, int host_flag) {
    constexpr int NW = NT / 64;
    // tile_seg[t] = index of tile t's first segment record = exclusive scan of ceil(n_t / 2^seg_shift)
    const uint32_t round = (1u << seg_shift) - 1u;
    const int per = (T + NT - 1) / NT;
    const int beg = min((int)threadIdx.x * per, T), end = min(beg + per, T);
    unsigned long long local = 0;
    uint32_t lmax = 0, lsegs = 0;
    for (int c = threadIdx.x; c < GSR_NCLS; c += NT) L.cls[c] = 0u;
    if (A) {
        for (int t0 = threadIdx.x; t0 < T; t0 += NT * 4) {            // (four requests in flight per thread, indices clamped: branch-free)
            uint32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = tile_count[min(t0 + u * NT, T - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (t0 + u * NT < T) A[t0 + u * NT] = v[u];
        }
        __syncthreads();
    }
    for (int i = beg; i < end; ++i) {
        const uint32_t c = A ? A[i] : tile_count[i];
        local += c; lmax = max(lmax, c); lsegs += (c + round) >> seg_shift;
    }
    // inclusive scan inside the wave
    unsigned long long incl = local;
    uint32_t sincl = lsegs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(incl, off, 64);
        const uint32_t so = __shfl_up(sincl, off, 64);
        if (lane >= off) { incl += o; sincl += so; }
    }
    lmax = wave_max_u32(lmax);
    if (lane == 63) { L.wsum[wave] = incl; L.wsegs[wave] = sincl; }
    if (lane == 0) L.wmax[wave] = lmax;
    __syncthreads();
    unsigned long long wave_base = 0;
    uint32_t seg_base = 0;
    for (int w = 0; w < wave; ++w) { wave_base += L.wsum[w]; seg_base += L.wsegs[w]; }
    unsigned long long run = wave_base + incl - local;
    uint32_t srun = seg_base + sincl - lsegs;
    for (int i = beg; i < end; ++i) {
        const uint32_t c = A ? A[i] : tile_count[i];
        tile_off[i] = (uint32_t)run; if (A) A[i] = (uint32_t)run; run += c;
        tile_seg[i] = srun; srun += (c + round) >> seg_shift;
    }
    {   // K1's per-workgroup statistics (M_ref, V): parallel sum over the workgroups
        unsigned long long a = 0, b = 0, c = 0;
        for (int i = threadIdx.x; i < nblocks; i += NT) { a += block_stats[3 * i]; b += block_stats[3 * i + 1]; c = max(c, block_stats[3 * i + 2]); }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); c = max(c, (unsigned long long)__shfl_xor(c, off, 64)); }
        if (lane == 0) { L.sref[wave] = a; L.svis[wave] = b; L.smax[wave] = c; }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long ta = 0, tb = 0, tc = 0;
            for (int w = 0; w < NW; ++w) { ta += L.sref[w]; tb += L.svis[w]; tc = max(tc, L.smax[w]); }
            counters[0] = ta; counters[1] = tb; counters[5] = tc;
        }
        // per view (block_stats is [view][workgroup][3]): counters[8 + 2v] = M_ref, [9 + 2v] = V -- the host picks the
        // compositing kernel of every view from them exactly as a single-view call would
        const int per_view = nviews > 0 ? nblocks / nviews : 0;
        for (int v = wave; v < nviews; v += NW) {
            unsigned long long va = 0, vb = 0;
            for (int i = lane; i < per_view; i += 64) { va += block_stats[3 * (v * per_view + i)]; vb += block_stats[3 * (v * per_view + i) + 1]; }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { va += __shfl_xor(va, off, 64); vb += __shfl_xor(vb, off, 64); }
            if (lane == 0) { counters[8 + 2 * v] = va; counters[9 + 2 * v] = vb; }
        }
    }
    uint32_t maxc = 0;
    for (int w = 0; w < NW; ++w) maxc = max(maxc, L.wmax[w]);
    if (threadIdx.x == NT - 1) {
        tile_off[T] = (uint32_t)(wave_base + incl);
        if (A) A[T] = (uint32_t)(wave_base + incl);
        counters[2] = wave_base + incl;
        counters[3] = maxc;
        const uint32_t nlev = min((maxc + round) >> seg_shift, (uint32_t)GSR_NLEV);
        counters[7] = (unsigned long long)nlev | ((unsigned long long)seg_shift << 32);
    }
    __syncthreads();                                      // (A: now the list starts, A[T] = M)
    // ---- classes: histogram of min(segments, GSR_NLEV). Empty tiles (two thirds of a 512^2 view) all fall into class 0:
    // one atomic per wave for them, not one per tile
    for (int t0 = 0; t0 < T; t0 += NT) {
        const int t = t0 + (int)threadIdx.x;
        const uint32_t n = t < T ? (A ? A[t + 1] - A[t] : tile_count[t]) : 0u;
        const unsigned long long em = __ballot(t < T && n == 0u);
        if (t < T && n != 0u) atomicAdd(&L.cls[min((n + round) >> seg_shift, (uint32_t)GSR_NLEV)], 1u);
        if (em != 0ull && lane == __builtin_ctzll(em)) atomicAdd(&L.cls[0], (uint32_t)__popcll(em));
    }
    __syncthreads();
    // S[h] = #tiles of class > h = the exclusive scan of the counts in REVERSED class order, read back reversed
    for (int c = threadIdx.x; c < GSR_NCLS; c += NT) L.tmp[c] = L.cls[(GSR_NCLS - 1) - c];
    __syncthreads();
    block_excl_scan_lds<NT>(L.tmp, GSR_NCLS, L.scratch);
    for (int c = threadIdx.x; c < GSR_NCLS; c += NT) L.cls[c] = L.tmp[(GSR_NCLS - 1) - c];      // every count is read: the array becomes S (and then the cursors)
    __syncthreads();
    // level_off[c] = sum of S[c'] for c' < c
    for (int c = threadIdx.x; c < GSR_NCLS; c += NT) L.tmp[c] = L.cls[c];
    __syncthreads();
    block_excl_scan_lds<NT>(L.tmp, GSR_NCLS, L.scratch);
    for (int c = threadIdx.x; c < GSR_NCLS; c += NT) level_off[c] = L.tmp[c];
    // order[S[class]++] = tile: descending classes, the empty tiles last
    for (int t0 = 0; t0 < T; t0 += NT) {
        const int t = t0 + (int)threadIdx.x;
        const uint32_t n = t < T ? (A ? A[t + 1] - A[t] : tile_count[t]) : 0u;
        const bool empty = t < T && n == 0u;
        const unsigned long long em = __ballot(empty);
        // (without A: tile_off was stored by other threads of this workgroup in front of several barriers. The sort kernels take a
        // tile's (start, length) from ONE load at its position in the order instead of order -> tile_off)
        if (t < T && n != 0u) {
            const uint32_t pos = atomicAdd(&L.cls[min((n + round) >> seg_shift, (uint32_t)GSR_NLEV)], 1u);
            order[pos] = (uint32_t)t;
            order_span[pos] = make_uint2(A ? A[t] : __hip_atomic_load(tile_off + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), n);
        }
        if (em != 0ull) {
            const int leader = __builtin_ctzll(em);
            uint32_t b = 0;
            if (lane == leader) b = atomicAdd(&L.cls[0], (uint32_t)__popcll(em));
            b = __builtin_amdgcn_readlane(b, leader);
            if (empty) {
                const uint32_t pos = b + (uint32_t)__popcll(em & ((1ull << lane) - 1ull));
                order[pos] = (uint32_t)t;
                order_span[pos] = make_uint2(0u, 0u);
            }
        }
    }
    // The host sizes the list scratch from the counters (gsr_forward's one round trip). They go straight into its pinned block --
    // the two stream-ordered D2H copies behind this kernel were blit KERNELS of their own (5-8 us each) with an 11 us hole behind
    // them in the rocprof trace of round 4: 23 us of every forward. Counters first, system-scope fence, then the flag the host polls.
    if (host_out) {
        __syncthreads();                                  // every counter above has been stored (by different threads)
        if ((int)threadIdx.x < host_words) {
            host_out[threadIdx.x] = __hip_atomic_load(counters + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
        }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(host_out + host_flag, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

extern "C" __global__ void __launch_bounds__(1024)
gsr_tile_scan(const uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_off, int T,
              unsigned long long* __restrict__ counters, uint32_t* __restrict__ tile_seg, int seg_shift,
              const unsigned long long* __restrict__ block_stats, int nblocks, int nviews,
              uint32_t* __restrict__ order, uint2* __restrict__ order_span, uint32_t* __restrict__ level_off,
              unsigned long long* __restrict__ host_out, int host_words, int host_flag, int lds_words /* dynamic LDS, in words */) {
    __shared__ TileScanLds L;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];      // (T + 1) words when the host found room for them, else nothing
    tile_scan_body<1024>(L, lds_words > T ? reinterpret_cast<uint32_t*>(smem_raw) : nullptr, tile_count, tile_off, T, counters, tile_seg, seg_shift, block_stats, nblocks, nviews, order, order_span, level_off,
                         host_out, host_words, host_flag);
}

// ---------------------------------------------------------------------------------------
// K3: scatter (depth bits, id) keys into the tile segments.
//
// The forward's work items ride along in both kernels (they wait on scattered stores most of their time): one 16-byte record per
// (tile, segment) of the segment forward -- {tile, list start, segment record, segment << 8 | entries - 1} -- in the depth-major
// order gsr_tile_scan defined (item k = level c, tile order[k - level_off[c]]), so that a workgroup of gsr_render_fwd_seg learns
// everything about its item from ONE scalar load instead of a chain of five dependent ones.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void write_forward_items(const uint32_t* __restrict__ level_off, const uint32_t* __restrict__ order,
                                                    const uint32_t* __restrict__ tile_off, const uint32_t* __restrict__ tile_seg,
                                                    int seg_shift, uint4* __restrict__ items, uint32_t items_cap, uint32_t* lev /* LDS [GSR_NLEV + 1] */) {
    for (int i = threadIdx.x; i <= GSR_NLEV; i += (int)blockDim.x) lev[i] = level_off[i];
    lds_barrier();
    const uint32_t total = min(lev[GSR_NLEV], items_cap);
    const uint32_t nthreads = gridDim.x * gridDim.y * blockDim.x, gid = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    for (uint32_t k = gid; k < total; k += nthreads) {
        uint32_t lo = 0, hi = GSR_NLEV;               // lev[lo] <= k < lev[hi]
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (lev[mid] <= k) lo = mid; else hi = mid; }
        const uint32_t t = order[k - lev[lo]];
        const uint32_t s0 = tile_off[t], n = tile_off[t + 1] - s0, first = lo << seg_shift;
        const uint32_t len = min(1u << seg_shift, n - first);
        items[k] = make_uint4(t, s0, tile_seg[t] + lo, (lo << 8) | (len - 1u));
    }
}

// What gsr_scatter's scan workgroup needs to run K2 inside the scatter's launch (by-value kernel argument).
struct ScanFold {
    int on;                                  // 0: gsr_tile_scan ran in front of this launch (tile_off is in memory)
    const uint32_t* tile_count;              // [views * nTiles]
    uint32_t* tile_off_w; uint32_t* tile_seg_w;
    const unsigned long long* block_stats; int nblocks;
    uint32_t* order_w; uint2* order_span; uint32_t* level_off_w;
    unsigned long long* host_out; int host_words, host_flag;
    int lds_words;                           // words of dynamic LDS behind the TileScanLds block (all tiles' counts fit: K2 keeps them there)
};

// LDS-histogram mode (tile grids of up to 16 384 tiles): K1's twin. A workgroup here is S x 256 threads and emits the Gaussians of a
// GROUP of S consecutive K1 workgroups (K1 workgroup w walks batches w, w + k1_grid, ... of 256; slice q = threadIdx.x / 256 of block
// g stands for K1 workgroup g * S + q): K1's histogram flush has reserved, per tile, ONE range for the group's entries (group_base);
// the positions inside it are handed out from LDS. ONE pass, no global atomics. S = 4 at the headline size: what a workgroup writes
// into a tile's list is a run of ~16 keys = a whole 128-byte line through ONE L2 (gsr_preprocess_fwd's flush has the measurement
// this answers).
//
// K2 FOLDED IN (fold.on, round 5; the speculative forward of views whose compositing takes its tiles from `order`): the scatter needs
// of K2 only where every tile's list starts -- an exclusive scan of the tile counts K1 left, which every workgroup here takes for
// itself in LDS (2 500 counts: ~1 us, the counts are L2 hits) -- and the rest of K2 (segment bases, K1's statistics, the tiles
// ordered by length for the sort and the compositing, the counters for the host) is needed by the kernels BEHIND the scatter. So
// workgroup x = 0 of view 0 runs K2's body (it has the scatter's 35 us to hide in) and the scatter workgroups are x = 1 .. groups:
// one launch and 15-18 us of a single-workgroup kernel fewer per forward.
// dynamic LDS: max(nTiles uint32, sizeof(TileScanLds) + (views * nTiles + 1) uint32 when that fits 64 KiB).
#define GSR_SC_R 4      // emission records a thread requests at once (K1's default grid: four batches per workgroup)
template <int S>
__global__ void __launch_bounds__(256 * S)
gsr_scatter(int N, const EmitRec* __restrict__ emit, const uint32_t* __restrict__ tile_off,
            const uint32_t* __restrict__ group_base, unsigned long long* __restrict__ entries,
            int gx, int nTiles, uint32_t capacity, unsigned long long* __restrict__ counters,
            const uint32_t* __restrict__ level_off, const uint32_t* __restrict__ order, const uint32_t* __restrict__ tile_seg,
            int seg_shift, uint4* __restrict__ items, uint32_t items_cap, int k1_grid, ScanFold fold,
            const unsigned long long* __restrict__ k1_stats /* K1's block_stats [views][k1_grid][3] */) {
    constexpr int NT = 256 * S;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t* next = reinterpret_cast<uint32_t*>(smem_raw);      // [nTiles]: the next free position of this group's range in the tile's list
    __shared__ uint32_t lev[GSR_NLEV + 1];
    __shared__ uint32_t red[2 * (NT / 64)];
    int bx = (int)blockIdx.x;
    if (fold.on) {
        if (bx == 0) {
            if (blockIdx.y == 0)
                tile_scan_body<NT>(*reinterpret_cast<TileScanLds*>(smem_raw),
                                   fold.lds_words > nTiles * (int)gridDim.y ? reinterpret_cast<uint32_t*>(smem_raw + ((sizeof(TileScanLds) + 15) & ~(size_t)15)) : nullptr, fold.tile_count, fold.tile_off_w, nTiles * (int)gridDim.y, counters,
                                   fold.tile_seg_w, seg_shift, fold.block_stats, fold.nblocks, (int)gridDim.y, fold.order_w, fold.order_span,
                                   fold.level_off_w, fold.host_out, fold.host_words, fold.host_flag);
            return;
        }
        --bx;
    } else if (counters[2] > (unsigned long long)capacity) return;
    // (the scratch may have been sized BEFORE the host knew M -- gsr_forward: previous call + 25 %. M is on the device: every consumer
    // of the lists leaves at once when they do not fit, and the host repeats the tail)
    // A K1 workgroup of this group gave up waiting for the cleared tile counters (gsr_preprocess_fwd: it reports M_ref >= 2^62 and the
    // host returns an error): the group reserved nothing, its row of group_base is whatever the scratch held -- nothing may be
    // written from it.
    const int ngroups = (k1_grid + S - 1) / S;
#pragma unroll
    for (int q = 0; q < S; ++q)
        if (bx * S + q < k1_grid && k1_stats[3 * ((size_t)blockIdx.y * k1_grid + bx * S + q)] >= (1ull << 62)) return;
    // blockIdx.y = view: its records and its tiles (tile_off holds positions in the one list array of all views)
    emit += (size_t)blockIdx.y * (size_t)N;
    const uint32_t* __restrict__ base_row = group_base + ((size_t)blockIdx.y * ngroups + bx) * nTiles;
    const int stride = k1_grid * 256;
    const int wg = bx * S + ((int)threadIdx.x >> 8);               // the K1 workgroup this slice stands for
    const int first = wg < k1_grid ? wg * 256 + ((int)threadIdx.x & 255) : N;      // (the last group may be short: its spare slices emit nothing)
    // a thread's first records travel while the ranges are loaded
    uint4 em[GSR_SC_R];
#pragma unroll
    for (int r = 0; r < GSR_SC_R; ++r) em[r] = reinterpret_cast<const uint4*>(emit)[min(first + r * stride, N - 1)];   // branch-free (index clamped)
    if (fold.on) {
        // this view's list starts: the counts of the views in front summed, the own view's scanned in LDS; M = the sum over all views
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int before = (int)blockIdx.y * nTiles, all = (int)gridDim.y * nTiles;
        uint32_t s_before = 0, s_after = 0;
        for (int t = threadIdx.x; t < all; t += NT) {
            if (t >= before && t < before + nTiles) continue;
            const uint32_t c = fold.tile_count[t];
            if (t < before) s_before += c; else s_after += c;
        }
        for (int t = threadIdx.x; t < nTiles; t += NT) next[t] = fold.tile_count[before + t];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { s_before += __shfl_xor(s_before, off, 64); s_after += __shfl_xor(s_after, off, 64); }
        if (lane == 0) { red[wave] = s_before; red[NT / 64 + wave] = s_after; }
        __syncthreads();
        const uint32_t own = block_excl_scan_lds<NT>(next, nTiles, lev);     // (lev: free here -- no items in this mode; NT / 64 words used)
        uint32_t lists_before = 0, lists_after = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { lists_before += red[w]; lists_after += red[NT / 64 + w]; }
        const unsigned long long M_all = (unsigned long long)lists_before + own + lists_after;
        if (M_all > (unsigned long long)capacity) return;                        // (the scan workgroup tells the host; the tail is repeated)
        if (bx == 0 && blockIdx.y == 0 && threadIdx.x == 0) counters[6] = capacity;
        for (int t = threadIdx.x; t < nTiles; t += NT) next[t] += lists_before + base_row[t];
    } else {
        if (bx == 0 && blockIdx.y == 0 && threadIdx.x == 0) counters[6] = capacity;    // for a backward called without GsrStats
        if (items_cap) write_forward_items(level_off, order, tile_off, tile_seg, seg_shift, items, items_cap, lev);
        tile_off += (size_t)blockIdx.y * nTiles;
        for (int t0 = threadIdx.x; t0 < nTiles; t0 += NT * 4) {      // (eight loads in flight per thread; entries of tiles nobody here emits into: never used)
            uint32_t a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int t = min(t0 + NT * u, nTiles - 1); a[u] = tile_off[t]; b[u] = base_row[t]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (t0 + NT * u < nTiles) next[t0 + NT * u] = a[u] + b[u];
        }
    }
    lds_barrier();
    for (int base = first; base < N; base += GSR_SC_R * stride) {
#pragma unroll
        for (int r = 0; r < GSR_SC_R; ++r) {
            const int idx = base + r * stride;
            if (idx >= N) break;
            const uint4 e = em[r];
            const int x0 = e.x & 0xffff, x1 = e.x >> 16, y0 = e.y & 0xffff, y1 = e.y >> 16;
            const unsigned long long key = ((unsigned long long)e.z << 32) | (uint32_t)idx;
            const bool masked = (x1 - x0) * (y1 - y0) <= GSR_EMIT_MASK_TILES;
            uint32_t bit = 1u;
            for (int ty = y0; ty < y1; ++ty)
                for (int tx = x0; tx < x1; ++tx, bit <<= 1) {
                    if (masked && !(e.w & bit)) continue;
                    const uint32_t pos = atomicAdd(&next[ty * gx + tx], 1u);
                    if (pos < capacity) entries[pos] = key;
                }
        }
        if (base + GSR_SC_R * stride < N) {           // grids pinned below K1's default: more than GSR_SC_R batches per workgroup
#pragma unroll
            for (int r = 0; r < GSR_SC_R; ++r) em[r] = reinterpret_cast<const uint4*>(emit)[min(base + (GSR_SC_R + r) * stride, N - 1)];
        }
    }
}
template __global__ void gsr_scatter<1>(int, const EmitRec*, const uint32_t*, const uint32_t*, unsigned long long*, int, int, uint32_t, unsigned long long*, const uint32_t*, const uint32_t*, const uint32_t*, int, uint4*, uint32_t, int, ScanFold, const unsigned long long*);
template __global__ void gsr_scatter<4>(int, const EmitRec*, const uint32_t*, const uint32_t*, unsigned long long*, int, int, uint32_t, unsigned long long*, const uint32_t*, const uint32_t*, const uint32_t*, int, uint4*, uint32_t, int, ScanFold, const unsigned long long*);

// Tile grids beyond the LDS histogram (more than 16 384 tiles): one pass, a global cursor per tile.
extern "C" __global__ void __launch_bounds__(256)
gsr_scatter_global(int N, const EmitRec* __restrict__ emit, const uint32_t* __restrict__ tile_off,
                   uint32_t* __restrict__ cursor, unsigned long long* __restrict__ entries,
                   int gx, int nTiles, uint32_t capacity, unsigned long long* __restrict__ counters,
                   const uint32_t* __restrict__ level_off, const uint32_t* __restrict__ order, const uint32_t* __restrict__ tile_seg,
                   int seg_shift, uint4* __restrict__ items, uint32_t items_cap) {
    if (counters[2] > (unsigned long long)capacity) return;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) counters[6] = capacity;    // for a backward called without GsrStats
    __shared__ uint32_t lev[GSR_NLEV + 1];
    if (items_cap) write_forward_items(level_off, order, tile_off, tile_seg, seg_shift, items, items_cap, lev);
    emit += (size_t)blockIdx.y * (size_t)N;
    tile_off += (size_t)blockIdx.y * nTiles;
    cursor += (size_t)blockIdx.y * nTiles;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < N; idx += gridDim.x * blockDim.x) {
        const uint4 em = reinterpret_cast<const uint4*>(emit)[idx];
        const int x0 = em.x & 0xffff, x1 = em.x >> 16, y0 = em.y & 0xffff, y1 = em.y >> 16;
        const unsigned long long key = ((unsigned long long)em.z << 32) | (uint32_t)idx;
        const bool masked = (x1 - x0) * (y1 - y0) <= GSR_EMIT_MASK_TILES;
        uint32_t bit = 1u;
        for (int ty = y0; ty < y1; ++ty)
            for (int tx = x0; tx < x1; ++tx, bit <<= 1) {
                if (masked && !(em.w & bit)) continue;
                const int t = ty * gx + tx;
                const uint32_t pos = tile_off[t] + atomicAdd(&cursor[t], 1u);
                if (pos < capacity) entries[pos] = key;
            }
    }
}

// ---------------------------------------------------------------------------------------
// Bitonic network (fallback of the bucket sort; in place in HBM for lists beyond the LDS classes)
// Bitonic network in the "all ascending" form (first step of each merge mirrors, the rest
// are half-cleaners) so that indices >= n act as +infinity keys and are simply skipped.
// ---------------------------------------------------------------------------------------
template <typename KeyPtr>
__device__ __forceinline__ void bitonic_sort(KeyPtr keys, uint32_t n, int nthreads) {
    uint32_t n2 = 1;
    while (n2 < n) n2 <<= 1;
    const uint32_t pairs = n2 >> 1;
    for (uint32_t k = 2, lk = 1; k <= n2; k <<= 1, ++lk) {
        // mirror step: lo = blk*k + w, hi = blk*k + k-1-w, w < k/2
        for (uint32_t t = threadIdx.x; t < pairs; t += nthreads) {
            const uint32_t w = t & ((k >> 1) - 1), blk = t >> (lk - 1);
            const uint32_t lo = blk * k + w, hi = blk * k + (k - 1 - w);
            if (hi < n) {
                const unsigned long long a = keys[lo], b = keys[hi];
                if (a > b) { keys[lo] = b; keys[hi] = a; }
            }
        }
        __syncthreads();
        for (uint32_t j = k >> 2, lj = lk - 2; j >= 1; j >>= 1, --lj) {
            for (uint32_t t = threadIdx.x; t < pairs; t += nthreads) {
                const uint32_t lo = ((t >> lj) << (lj + 1)) | (t & (j - 1));
                const uint32_t hi = lo + j;
                if (hi < n) {
                    const unsigned long long a = keys[lo], b = keys[hi];
                    if (a > b) { keys[lo] = b; keys[hi] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------
// K4: per-tile BUCKET sort. The bitonic network above costs log^2 passes over LDS
// (105 barrier-separated passes for 8k keys). Depth keys of one tile are spread smoothly
// between the tile's nearest and farthest Gaussian, so one counting pass into ~n/4 buckets
// by linearly quantised depth (monotone => bucket order = depth order) followed by an
// insertion sort of each few-element bucket on the full 64-bit key (depth bits, then index:
// the same total order as the network) is O(n). A tile whose keys pile up in one bucket
// (max bucket > kBucketLimit, e.g. thousands of equal depths) falls back to the network.
// Output: the tile's Gaussian indices in final order (out_ids).
// dynamic LDS: u64 keys[CAP] | u32 off[NBMAX+1] | u32 cur[NBMAX] | u32 red[40]
// ---------------------------------------------------------------------------------------
constexpr uint32_t kBucketLimit = 64;

template <int NT>
__device__ __forceinline__ void block_excl_scan_u32(uint32_t* a, int n, uint32_t* wsum) {
    const int per = (n + NT - 1) / NT;
    const int beg = min((int)threadIdx.x * per, n), end = min(beg + per, n);
    uint32_t local = 0;
    for (int i = beg; i < end; ++i) local += a[i];
    uint32_t incl = local;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    uint32_t run = base + incl - local;
    for (int i = beg; i < end; ++i) { const uint32_t c = a[i]; a[i] = run; run += c; }
    __syncthreads();
}

template <int CAP, int NT, int NBMAX>
__device__ __forceinline__ void sort_one_tile(unsigned long long* keys, uint32_t* off, uint32_t* cur, uint32_t* red,
                                              const unsigned long long* __restrict__ entries, uint32_t* __restrict__ out_ids, uint32_t s, uint32_t n) {
    const unsigned long long* __restrict__ src = entries + s;
    const int lane = threadIdx.x & 63;
    constexpr int PER = CAP / NT;                      // entries per thread: ONE pass over HBM, the keys stay in registers
    unsigned long long mine[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const uint32_t i = threadIdx.x + u * NT;
        mine[u] = i < n ? src[i] : ~0ull;
    }
    if (n <= 256) {
        // rank sort: every key counts the keys below it (keys are distinct: the index is part of them); the
        // reads are wave-uniform (broadcast). One barrier instead of the network's 36 for 256 keys -- the list
        // lengths of DreamGaussian-sized scenes (a few thousand Gaussians on 256 tiles) live here.
#pragma unroll
        for (int u = 0; u < PER; ++u) { const uint32_t i = threadIdx.x + u * NT; if (i < n) keys[i] = mine[u]; }
        __syncthreads();
        if (threadIdx.x < n) {                          // n <= 256 <= NT: one key per thread
            const unsigned long long k = mine[0];
            uint32_t rank = 0;
            for (uint32_t q = 0; q < n; ++q) rank += keys[q] < k ? 1u : 0u;
            out_ids[s + rank] = (uint32_t)k;
        }
        return;
    }
    bool use_network = false;
    {
        uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (threadIdx.x + u * NT < n) { const uint32_t d = (uint32_t)(mine[u] >> 32); mn = min(mn, d); mx = max(mx, d); }
        }
        mx = wave_max_u32(mx);
        mn = ~wave_max_u32(~mn);
        if (threadIdx.x == 0) { red[32] = 0xffffffffu; red[33] = 0u; red[34] = 0u; }
        __syncthreads();
        if (lane == 0) { atomicMin(&red[32], mn); atomicMax(&red[33], mx); }
        __syncthreads();
        const float fmin = __uint_as_float(red[32]), fmax = __uint_as_float(red[33]);
        const int NB = min(NBMAX, max(1, (int)(n >> 2)));
        const float range = fmax - fmin;
        const float scale = range > 0.f ? (float)NB / range : 0.f;
        for (int b = threadIdx.x; b <= NB; b += NT) { off[b] = 0; if (b < NB) cur[b] = 0; }
        __syncthreads();
        int bucket[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            bucket[u] = min((int)((__uint_as_float((uint32_t)(mine[u] >> 32)) - fmin) * scale), NB - 1);
            if (threadIdx.x + u * NT < n) atomicAdd(&off[bucket[u]], 1u);
        }
        __syncthreads();
        uint32_t m = 0;
        for (int b = threadIdx.x; b < NB; b += NT) m = max(m, off[b]);
        m = wave_max_u32(m);
        if (lane == 0) atomicMax(&red[34], m);
        __syncthreads();
        use_network = red[34] > kBucketLimit;             // block-uniform
        if (!use_network) {
            block_excl_scan_u32<NT>(off, NB + 1, red);      // off[NB] = n
#pragma unroll
            for (int u = 0; u < PER; ++u)
                if (threadIdx.x + u * NT < n) keys[off[bucket[u]] + atomicAdd(&cur[bucket[u]], 1u)] = mine[u];
            __syncthreads();
            // final position = bucket start + rank inside the bucket (a few keys): no serial insertion, no
            // second buffer -- the index goes straight to HBM
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                if (threadIdx.x + u * NT < n) {
                    const uint32_t lo = off[bucket[u]], hi = off[bucket[u] + 1];
                    const unsigned long long k = mine[u];
                    uint32_t rank = 0;
                    for (uint32_t q = lo; q < hi; ++q) rank += keys[q] < k ? 1u : 0u;
                    out_ids[s + lo + rank] = (uint32_t)k;
                }
            }
            return;
        }
    }
    // skewed depths (e.g. thousands of equal keys in one bucket): the network
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; ++u) { const uint32_t i = threadIdx.x + u * NT; if (i < n) keys[i] = mine[u]; }
    __syncthreads();
    bitonic_sort(keys, n, NT);
    for (uint32_t i = threadIdx.x; i < n; i += NT) out_ids[s + i] = (uint32_t)keys[i];
}

// A list LONGER than the class's LDS buffer (round 6; was: a 16 384-entry class of its own -- one workgroup per CU, launched on a
// prediction and mostly for nothing, 4.9 us of every step at 1M Gaussians, 27 us BEHIND the main launch where such lists exist -- and
// beyond that a bitonic network in HBM). Same bucket sort, with the keys streamed from the list (an L2-resident re-read) instead of
// held in registers: minimum / maximum, histogram over n / 4 depth buckets (<= NBMAX), scan; then the buckets go through LDS in
// depth-ordered PORTIONS of at most CAP keys -- stream the list, keep the portion's keys, rank inside the bucket, store the indices.
// Any length; two or three portions for the lists the 16 384 class took. Skewed depths (a bucket of more than kLongBucketLimit keys):
// the network, in place in the list (global memory, the workgroup's own L1: the barriers between its passes order it).
constexpr uint32_t kLongBucketLimit = 512;       // (a portion must hold a whole bucket: <= the smallest CAP; the rank loop is as long as the bucket)
template <int CAP, int NT, int NBMAX>
__device__ __forceinline__ void sort_long_tile(unsigned long long* keys, uint32_t* off, uint32_t* cur, uint32_t* red,
                                               unsigned long long* __restrict__ list /* entries + s: may be sorted in place */,
                                               uint32_t* __restrict__ out /* out_ids + s */, uint32_t n) {
    const int lane = threadIdx.x & 63;
    uint32_t mn = 0xffffffffu, mx = 0u;
    for (uint32_t i = threadIdx.x; i < n; i += NT) { const uint32_t d = (uint32_t)(list[i] >> 32); mn = min(mn, d); mx = max(mx, d); }
    mx = wave_max_u32(mx);
    mn = ~wave_max_u32(~mn);
    if (threadIdx.x == 0) { red[32] = 0xffffffffu; red[33] = 0u; red[34] = 0u; }
    __syncthreads();
    if (lane == 0) { atomicMin(&red[32], mn); atomicMax(&red[33], mx); }
    __syncthreads();
    const float fmin = __uint_as_float(red[32]), fmax = __uint_as_float(red[33]);
    const int NB = min(NBMAX, (int)(n >> 2));
    const float range = fmax - fmin;
    const float scale = range > 0.f ? (float)NB / range : 0.f;
    auto bucket_of = [&](unsigned long long k) -> int { return min((int)((__uint_as_float((uint32_t)(k >> 32)) - fmin) * scale), NB - 1); };
    for (int b = threadIdx.x; b <= NB; b += NT) off[b] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += NT) atomicAdd(&off[bucket_of(list[i])], 1u);
    __syncthreads();
    uint32_t m = 0;
    for (int b = threadIdx.x; b < NB; b += NT) m = max(m, off[b]);
    m = wave_max_u32(m);
    if (lane == 0) atomicMax(&red[34], m);
    __syncthreads();
    if (red[34] > kLongBucketLimit) {                     // (block-uniform)
        bitonic_sort(list, n, NT);
        for (uint32_t i = threadIdx.x; i < n; i += NT) out[i] = (uint32_t)list[i];
        return;
    }
    __syncthreads();                                      // (red[34] has been read by everyone: the scan reuses red)
    block_excl_scan_u32<NT>(off, NB + 1, red);            // off[b] = where bucket b starts in the sorted list, off[NB] = n
    int lo = 0;
    while (lo < NB) {                                     // (block-uniform) buckets [lo, hi): the most that fit CAP keys
        const uint32_t base = off[lo];
        int a = lo + 1, z = NB;                           // the largest hi in [lo + 1, NB] with off[hi] - base <= CAP (a bucket holds <= kLongBucketLimit <= CAP keys)
        while (a < z) { const int mid = (a + z + 1) >> 1; if (off[mid] - base <= (uint32_t)CAP) a = mid; else z = mid - 1; }
        const int hi = a;
        const uint32_t cnt = off[hi] - base;
        for (int b = lo + (int)threadIdx.x; b < hi; b += NT) cur[b] = 0u;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += NT) {
            const unsigned long long k = list[i];
            const int b = bucket_of(k);
            if (b >= lo && b < hi) keys[off[b] - base + atomicAdd(&cur[b], 1u)] = k;
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < cnt; j += NT) {
            const unsigned long long k = keys[j];
            const int b = bucket_of(k);
            const uint32_t l0 = off[b] - base, l1 = off[b + 1] - base;
            uint32_t rank = 0;
            for (uint32_t q = l0; q < l1; ++q) rank += keys[q] < k ? 1u : 0u;
            out[base + l0 + rank] = (uint32_t)k;
        }
        __syncthreads();
        lo = hi;
    }
}

template <int CAP, int NT, int NBMAX>
__global__ void __launch_bounds__(NT, CAP == 8192 ? 2 : 1)     // the 8 192 class: two 1 024-thread workgroups per CU (<= 64 VGPRs, 2 x 81 060 B of LDS)
gsr_tile_sort_bucket(const uint32_t* __restrict__ tile_off, unsigned long long* __restrict__ entries,
                     uint32_t* __restrict__ out_ids, uint32_t lo_excl, uint32_t hi_incl /* (unused bound of the classes of rounds 1-5: every list longer than lo_excl is sorted) */,
                     const unsigned long long* __restrict__ counters, uint32_t capacity,
                     const uint2* __restrict__ order_span /* (list start, length) of the tiles, longest lists first (gsr_tile_scan) */,
                     uint32_t ntiles /* entries of order_span */) {
    if (counters[2] > (unsigned long long)capacity) return;            // lists do not fit the scratch (see gsr_scatter)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem_raw);
    uint32_t* off = reinterpret_cast<uint32_t*>(smem_raw + (size_t)CAP * 8);
    uint32_t* cur = off + (NBMAX + 1);
    uint32_t* red = cur + NBMAX;                       // [0..15] wave partials, [32] min, [33] max, [34] max bucket
    // longest lists first: a tile is a latency chain of its own (eight barriers), and with two 80 KiB workgroups per CU the
    // launch is two rounds deep -- the long chains start in the first, the short ones fill in behind them
    const uint2 span = order_span[blockIdx.x];
    if (span.y <= lo_excl) return;
    if (span.y <= (uint32_t)CAP) sort_one_tile<CAP, NT, NBMAX>(keys, off, cur, red, entries, out_ids, span.x, span.y);
    else sort_long_tile<CAP, NT, NBMAX>(keys, off, cur, red, entries + span.x, out_ids + span.x, span.y);
}

template __global__ void gsr_tile_sort_bucket<2048, 256, 512>(const uint32_t*, unsigned long long*, uint32_t*, uint32_t, uint32_t, const unsigned long long*, uint32_t, const uint2*, uint32_t);
template __global__ void gsr_tile_sort_bucket<8192, 1024, 1920>(const uint32_t*, unsigned long long*, uint32_t*, uint32_t, uint32_t, const unsigned long long*, uint32_t, const uint2*, uint32_t);
