// gsr_api.hip -- host side of the C ABI declared in include/gsr.h (single translation unit:
// the kernel files are included so that one hipcc invocation builds libgsr.so for gfx950).
#include "../../include/gsr.h"
#include "gsr_device.h"

#include "gsr_preprocess.hip"
#include "gsr_binning.hip"
#include "gsr_render.hip"
#include "gsr_knn.hip"
#include "gsr_fields.hip"
#include "gsr_densify.hip"
#include "gsr_optim.hip"

#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <mutex>
#include <atomic>

namespace {

thread_local char g_err[512] = "";
thread_local unsigned long long* g_pinned = nullptr;   // (kCounterWords + 1) x u64 host-pinned scratch: the counters, then the arrival flag
constexpr int kCounterWords = 8 + 2 * GSR_MAX_VIEWS;  // device counters the host reads: 8 totals, then (M_ref, V) per view
constexpr int kWalkCounter = kCounterWords;            // device only: walk items handed from the chaining kernel to the fix-up kernel
constexpr int kZeroedFlag = kCounterWords + 4;          // device only: K1's "tile counts / cursors / counters are cleared" tag of the launch (gsr_preprocess.hip)
constexpr int kCounterSlots = kCounterWords + 6;        // u64 words of the counter block
// What this thread's earlier gsr_forward calls of one problem shape left behind: predicts the next one's list sizes. A TABLE keyed by
// (N, H, W, views), least recently used slot replaced: DreamGaussian's own loop renders the 256^2 known view and then a 128^2 / 256^2 /
// 512^2 novel view EVERY iteration (main.py:198-216, :211) -- with one slot (rounds 2-5) every forward of the 70 % of iterations whose
// two resolutions differ found the other shape's hint and took the wait-first path.
struct FwdHint {
    bool valid = false; int N = 0, H = 0, W = 0, B = 0;
    unsigned long long M = 0, maxc = 0;                 // the most recent call
    unsigned long long M_hi = 0, maxc_hi = 0;           // the largest of the recent calls, decaying (what an asynchronous forward sizes from)
    unsigned long long per_view[2 * GSR_MAX_VIEWS] = {};
    unsigned long long stamp = 0;
};
constexpr int kHintSlots = 8;
thread_local FwdHint g_hints[kHintSlots];
thread_local unsigned long long g_hint_clock = 0;
FwdHint* hint_find(int N, int H, int W, int B) {
    for (auto& h : g_hints) if (h.valid && h.N == N && h.H == H && h.W == W && h.B == B) return &h;
    return nullptr;
}
void hint_update(int N, int H, int W, int B, unsigned long long M, unsigned long long maxc, const unsigned long long* per_view) {
    FwdHint* h = hint_find(N, H, W, B);
    if (!h) {                                           // an empty slot, else the least recently used one
        h = &g_hints[0];
        for (auto& c : g_hints) { if (!c.valid) { h = &c; break; } if (c.stamp < h->stamp) h = &c; }
        *h = FwdHint();
        h->valid = true; h->N = N; h->H = H; h->W = W; h->B = B;
    }
    h->M = M; h->maxc = maxc;
    h->M_hi = M > h->M_hi - h->M_hi / 8 ? M : h->M_hi - h->M_hi / 8;              // max(this call, 7/8 of the running maximum)
    h->maxc_hi = maxc > h->maxc_hi - h->maxc_hi / 8 ? maxc : h->maxc_hi - h->maxc_hi / 8;
    for (int v = 0; v < 2 * B; ++v) h->per_view[v] = per_view[v];
    h->stamp = ++g_hint_clock;
}
void hint_drop(int N, int H, int W, int B) { if (FwdHint* h = hint_find(N, H, W, B)) h->valid = false; }
constexpr unsigned long long kFlagSentinel = 0xffffffffffffffffull;

// An ASYNCHRONOUS forward (GSR_VIEW_ASYNC_STATS) returns before its counters have arrived. They land in this thread's pinned block;
// whoever needs them next -- the thread's next gsr_forward (it re-arms the block), gsr_forward_complete -- waits for them then, checks
// the speculation and files the statistics here.
struct PendingFwd {
    bool on = false; long long serial = 0;
    int N = 0, H = 0, W = 0, B = 0;
    unsigned long long cap = 0, capc = 0;
    hipStream_t stream = nullptr;
};
struct CompletedFwd { long long serial = 0; int rc = 0; unsigned long long M_ref = 0, V = 0, M = 0, maxc = 0; };
thread_local PendingFwd g_pending;
thread_local CompletedFwd g_completed;
thread_local long long g_async_serial = 0;

int fail(int code, const char* fmt, const char* a = "", long long b = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, b);
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return fail(-2, "HIP error '%s' at line %lld: " #expr, hipGetErrorString(e_), __LINE__); \
    } while (0)

// ---- per-kernel timing (gsr_profile_*): hipEvent pairs on the caller's stream ------------
struct ProfPending { const char* name; hipEvent_t start, stop; };
struct ProfRow { const char* name; double ms; int launches; };
struct ProfState {
    bool on = false;
    std::vector<hipEvent_t> pool;
    std::vector<ProfPending> pending;
    std::vector<ProfRow> rows;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
};
// process-wide (the backward runs on torch's autograd thread, the forward on the caller's)
ProfState g_prof;
std::mutex g_prof_mu;
thread_local hipEvent_t g_prof_cur = nullptr;

inline void prof_begin(hipStream_t stream) {
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_cur = g_prof.get();
    if (g_prof_cur) (void)hipEventRecord(g_prof_cur, stream);
}
inline void prof_end(hipStream_t stream, const char* name) {
    if (!g_prof.on || !g_prof_cur) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEvent_t stop = g_prof.get();
    if (stop) { (void)hipEventRecord(stop, stream); g_prof.pending.push_back({name, g_prof_cur, stop}); }
    else g_prof.pool.push_back(g_prof_cur);
    g_prof_cur = nullptr;
}

int launch_status(bool debug, hipStream_t stream, const char* name) {
    prof_end(stream, name);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && debug) e = hipStreamSynchronize(stream);
    if (e == hipSuccess) return 0;
    snprintf(g_err, sizeof(g_err), "kernel %s failed: %s", name, hipGetErrorString(e));
    return -3;
}
#define LAUNCH_CHECK(view, stream, name)                                                    \
    do { if (int rc_ = launch_status((view)->debug != 0, stream, name)) return rc_; } while (0)

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Runs `fn` the first time the calling code reaches this call site on each device (function
// attributes such as the > 64 KiB dynamic-LDS opt-in are per device; forward and backward run on
// different host threads).
template <typename F>
int once_per_device(F fn) {
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { HIP_TRY(fn()); return 0; }
    std::lock_guard<std::mutex> lk(mu);
    if (!done[dev]) { HIP_TRY(fn()); done[dev] = true; }
    return 0;
}

// ---- test hooks (gsr_testing_override, include/gsr.h): choices the library makes from the problem shape, forced by a TEST or an
// A/B measurement through an explicit call. Nothing here is read from the process environment: two of them (forward mode, segment
// length) decide where the per-pixel sums are cut, i.e. the rounding of the results, and that must not depend on who started the
// process. -1 = the library decides.
enum { OV_FWD_MODE, OV_SEG_SHIFT, OV_FWD_LISTS, OV_FWD_HINTS, OV_SPECULATE, OV_HIST_MAX, OV_K1_GRID, OV_FWD_GRID, OV_K6_GRID, OV_FWD_LDS_KB, OV_BWD_GRID, OV_K6_COMPACT, OV_SCAN_FOLD, OV_GRAD_CLEAR, OV_K1_GROUP, OV_SORT_KERNEL, OV_COUNT };
const char* const kOvNames[OV_COUNT] = {"fwd_mode", "seg_shift", "fwd_lists", "fwd_hints", "speculate", "hist_max", "k1_grid", "fwd_grid", "k6_grid", "fwd_lds_kb", "bwd_grid", "k6_compact", "scan_fold", "grad_clear", "k1_group", "sort_kernel"};
std::atomic<int> g_ov[OV_COUNT] = {{-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}, {-1}};
inline int ov(int k) { return g_ov[k].load(std::memory_order_relaxed); }

// K1's grid (a persistent grid: every workgroup walks the same number of 256-Gaussian batches; test hook "k1_grid" pins it). The
// scatter kernel runs on the SAME grid with the same Gaussian -> workgroup assignment (it continues K1's per-workgroup list ranges).
int k1_grid_for(int N) {
    if (N <= 0) return 0;
    const int batches = (N + 255) / 256;
    const int pin = ov(OV_K1_GRID) < 0 ? 0 : (ov(OV_K1_GRID) > 2048 ? 2048 : ov(OV_K1_GRID));
    if (pin) return batches < pin ? batches : pin;
    const int cap = 1024, rounds = (batches + cap - 1) / cap;      // block_stats holds 2048 workgroups per view
    return (batches + rounds - 1) / rounds;
}

// How many consecutive K1 workgroups share one reserved range per tile list = how many 256-thread slices a workgroup of gsr_scatter has
// (1 or 4). Four from 256 K1 workgroups on (65k Gaussians): a group's run in a list is then a whole cache line written through one
// L2, and a tile counter sees a quarter of the reserving atomics (gsr_preprocess_fwd's flush; profiles/r06_group_reservation.txt:
// K1 -7 us and the scatter -14 us at 1M Gaussians, K1 -10 us at 250k). Test hook "k1_group" pins it (1 / 4).
int k1_group_for(int N) {
    const int pin = ov(OV_K1_GROUP);
    if (pin == 1 || pin == 4) return pin;
    return k1_grid_for(N) >= 256 ? 4 : 1;
}

// Largest tile grid whose per-tile counters a workgroup of K1 / the scatter keeps in LDS (64 KiB)
static int hist_lds_max_tiles() {
    const int t = ov(OV_HIST_MAX);
    return t < 0 ? 16384 : (t > 16384 ? 16384 : t);
}

struct GeomLayout {
    size_t recs, emit, flags8, block_stats, tile_count, cursor, counters, arrive, tile_off, tile_seg, order, order_span, level_off, sat, plan_off, g2d, wg_base, total;
    int nTiles;        // per view
    int allTiles;      // views * nTiles: the per-tile arrays hold every view's tiles, view-major
};
// B views of the same size share one scratch block: per-view arrays are [B][N] / [B][nTiles], the lists of all
// B * nTiles tiles live in ONE array (BinLayout) addressed through tile_off.
GeomLayout geom_layout(int N, int H, int W, int B = 1, bool with_acc = true /* false: a forward no backward follows (GSR_VIEW_NO_BACKWARD) */) {
    GeomLayout L;
    const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
    L.nTiles = gx * gy;
    L.allTiles = L.nTiles * B;
    const size_t BN = (size_t)B * (size_t)N, BT = (size_t)L.allTiles;
    size_t o = 0;
    L.recs = o; o += align_up(BN * sizeof(SplatRec));
    L.emit = o; o += align_up(BN * sizeof(EmitRec));
    L.flags8 = o; o += align_up(BN);
    L.block_stats = o; o += align_up((size_t)B * 2048 * 3 * 8 + 256);  // K1 grid <= 2048 workgroups per view; + 256 B nobody reads (K1's store sink)
    // tile_count | cursor | counters are contiguous: one memset in front of K1
    L.tile_count = o; o += align_up(BT * 4);
    L.cursor = o; o += align_up(BT * 4);
    L.counters = o; o += align_up((size_t)kCounterSlots * 8);          // totals, (M_ref, V) per view, walk items, quad-mask word, K1's tag
    L.arrive = o; o += align_up((size_t)B * 2048 * 4);                 // K1's flush: workgroups of a group that have stored their histogram (cleared with the counters)
    L.tile_off = o; o += align_up((BT + 1) * 4);
    L.tile_seg = o; o += align_up((BT + 1) * 4);
    L.order = o; o += align_up(BT * 4);
    L.order_span = o; o += align_up(BT * 8);                       // (list start, length) of order[k]: what a sort workgroup needs, in one load
    L.level_off = o; o += align_up((GSR_NLEV + 1) * 4);
    L.sat = o; o += align_up(BT * 4 * 8);                         // hint word per (tile, wave) of the segment forward
    L.plan_off = o; o += align_up(BT * 4);
    // the backward's screen-space gradient accumulators [B][N][12] f32 and, right behind them, the [B][N] byte flags "this Gaussian
    // received a gradient": cleared by the FORWARD (forward_impl) so that the backward starts on its first kernel. A forward that
    // no backward can follow (GSR_VIEW_NO_BACKWARD) leaves them out (with_acc = false).
    L.g2d = o; o += with_acc ? align_up(BN * GSR_G2D_STRIDE * 4 + (size_t)B * GSR_LIVE_BYTES(N)) : 0;
    // LAST (round-5 advisor: its size follows K1's grid and the LDS-histogram limit, both of which a test hook can change between a
    // forward and its backward -- nothing the backward reads may sit behind it): where each of K1's workgroups starts inside every
    // tile's list ([view][workgroup][tile] u32 histogram rows of K1's workgroups, then [view][group][tile] u32 range starts: written
    // by K1's histogram flush, the second read by the scatter) -- only with the tile counters in LDS: larger tile grids use global
    // cursors and never touch it
    L.wg_base = o;
    if (L.nTiles <= hist_lds_max_tiles()) {
        const int grid = k1_grid_for(N), G = k1_group_for(N);
        o += align_up((size_t)B * (size_t)grid * (size_t)((L.nTiles + 3) & ~3) * 4) + align_up((size_t)B * (size_t)((grid + G - 1) / G) * (size_t)L.nTiles * 4);
    }
    L.total = o;
    return L;
}
// where the [view][group][tile] range starts sit behind the histogram rows (geom_layout)
inline size_t group_base_off(const GeomLayout& L, int N, int B) { return L.wg_base + align_up((size_t)B * (size_t)k1_grid_for(N) * (size_t)((L.nTiles + 3) & ~3) * 4); }
struct BinLayout { size_t entries, ids, ckpt, plan_tile, plan_cap, items, item_recs, walk_items, total; };
BinLayout bin_layout(size_t M, int nTiles, int shift, bool records = true /* false: a serial-walk forward no backward follows keeps no segment records */) {
    BinLayout L;
    size_t o = 0;
    // first: the sorted lists (4-byte Gaussian indices) -- the backward finds them at offset 0
    // without knowing M
    L.ids = o; o += align_up(M * 4);
    L.entries = o; o += align_up(M * 8);
    // (tile, segment) items of the forward = segment records: sum over tiles of ceil(n_t >> shift) <= (M >> shift) + nTiles
    L.items = (M >> shift) + (size_t)nTiles;
    L.ckpt = o; o += align_up(((records ? L.items : 0) + 3) * (size_t)GSR_CKPT_FLOATS * 4);   // + 3 spare records: the store sink of the serial walk (gsr_render.hip)
    // backward work list: sum over tiles of ceil(last_t / 2^shift) <= the same bound
    L.plan_cap = L.items + 1;
    L.plan_tile = o; o += align_up(L.plan_cap * 16);      // 16-byte work items of the backward: (tile, the segment's record, list start, segment | (entries - 1) << 24)
    L.item_recs = o; o += align_up(L.items * 16);        // the forward's work items (written by gsr_scatter)
    L.walk_items = o; o += align_up(L.items * 4 * 8);    // (tile, block, segment) items of the fix-up kernel: <= one per block and segment
    L.total = o < 256 ? 256 : o;
    return L;
}

ViewConst make_view(const GsrView* v) {
    ViewConst c;
    c.W = v->image_width; c.H = v->image_height;
    c.gx = (c.W + GSR_TILE - 1) / GSR_TILE; c.gy = (c.H + GSR_TILE - 1) / GSR_TILE;
    c.tanfovx = v->tanfovx; c.tanfovy = v->tanfovy;
    c.focal_x = c.W / (2.0f * v->tanfovx); c.focal_y = c.H / (2.0f * v->tanfovy);
    c.scale_modifier = v->scale_modifier; c.sh_degree = v->sh_degree;
    c.raw_act = v->raw_activations != 0;
    c.mat_t = v->flags & (GSR_VIEW_VIEWMATRIX_T | GSR_VIEW_PROJMATRIX_T);
    c.bg = v->bg; c.view = v->viewmatrix; c.proj = v->projmatrix; c.campos = v->campos;
    return c;
}

int check_view(const GsrView* v) {
    if (!v) return fail(-1, "view is NULL%s", "");
    if (v->image_width <= 0 || v->image_height <= 0) return fail(-1, "image size must be positive%s", "");
    if (v->image_width > 32767 || v->image_height > 32767) return fail(-1, "image size above 32767 is not supported%s", "");
    if (!v->bg || !v->viewmatrix || !v->projmatrix || !v->campos) return fail(-1, "bg/viewmatrix/projmatrix/campos must be device pointers%s", "");
    if (v->sh_degree < 0 || v->sh_degree > 3) return fail(-1, "sh_degree must be in 0..3%s", "");
    return 0;
}

// the B views of one launch chain: same image size, SH degree, activation mode and SH buffers (they share the kernels)
int check_views(const GsrView* views, int B) {
    if (!views) return fail(-1, "view is NULL%s", "");
    if (B < 1 || B > GSR_MAX_VIEWS) return fail(-1, "the number of views per call must be in 1..%s%lld", "", (long long)GSR_MAX_VIEWS);
    for (int v = 0; v < B; ++v) {
        if (int rc = check_view(views + v)) return rc;
        if (views[v].image_width != views[0].image_width || views[v].image_height != views[0].image_height ||
            views[v].sh_degree != views[0].sh_degree || (views[v].raw_activations != 0) != (views[0].raw_activations != 0) ||
            views[v].shs_rest != views[0].shs_rest || views[v].dL_dshs_rest != views[0].dL_dshs_rest)
            return fail(-1, "the views of one call must agree in image size, sh_degree, raw_activations and shs_rest%s", "");
    }
    return 0;
}

ViewSplit make_split(const GsrView* views, int B, int N, int T, int H, int W) {
    ViewSplit vs;
    memset(&vs, 0, sizeof(vs));
    vs.tiles_per_view = T; vs.N = N;
    vs.img_stride = (unsigned long long)(align_up((size_t)H * W * 4) * 7 / 4);
    for (int v = 0; v < B; ++v) vs.bg[v] = views[v].bg;
    return vs;
}

int check_inputs(int N, int K, const GsrView* v, const float* means3D, const float* shs,
                 const float* colors, const float* opac, const float* scales, const float* rots,
                 const float* cov3D) {
    if (N < 0) return fail(-1, "N must be >= 0%s", "");
    if (N == 0) return 0;
    if (!means3D || !opac) return fail(-1, "means3D and opacities are required%s", "");
    if ((shs != nullptr) == (colors != nullptr)) return fail(-1, "Please provide excatly one of either SHs or precomputed colors!%s", "");
    const bool sr = scales != nullptr && rots != nullptr;
    if (sr == (cov3D != nullptr) || ((scales != nullptr) != (rots != nullptr)))
        return fail(-1, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!%s", "");
    if (shs && K < (v->sh_degree + 1) * (v->sh_degree + 1)) return fail(-1, "shs has fewer coefficients than sh_degree needs%s", "");
    if (v->shs_rest && (!shs || K < 2)) return fail(-1, "split SH input (GsrView.shs_rest) needs shs = features_dc and K >= 2%s", "");
    return 0;
}

// How the forward composites. Depth-segmented (K5a + K5b + K5c) where a view has few tiles: there the serial walk of a
// tile's list leaves the chip empty (a 512^2 view has ~350 non-empty tiles = 1.4 waves per SIMD, a 256^2 view 0.6) and
// the forward is a latency chain. The serial walk (K5b alone) where a view fills the chip: every vector instruction of the
// compositing is then needed exactly once, and the segmented path's extra ones (segments composited behind the stop before
// the hint arrived, the stop segment walked a second time, per-item set-up: +40 % at 1M Gaussians / 800^2) cost more than
// its fuller SIMDs win. Measured, forward compositing in us, segmented / serial: 5k-256^2 36 / 63, 250k-512^2 116 / 130,
// 100k-800^2 125 / 89, 1M-800^2 196 / 169, 1M-800^2 anisotropic 152 / 108.
// The choice fixes where the per-pixel sums are cut (the rounding of the results), so -- like the segment length below --
// it must not depend on anything a second call with the same inputs could see differently, nor on how many views share the
// call (a view of a batch renders bit-identically to its single-view call): the per-view tile count only.
bool fwd_sequential_for(int N, int tiles_per_view) {
    (void)N;
    if (ov(OV_FWD_MODE) == 1 || ov(OV_FWD_MODE) == 3) return true;   // test hook: 1 = serial walk, 2 = depth-segmented, 3 = serial walk with two waves per block (experimental)
    if (ov(OV_FWD_MODE) == 2) return false;
    return tiles_per_view >= 1024;
}
// Segment length of one call: the backward's unit of work in both modes (it prefers 64 entries: 0.219 / 0.233 / 0.300 ms
// at 1M for 64 / 128 / 256), the forward's in the segmented mode, where long lists amortise the per-item set-up over 128
// entries (250k-512^2: K5a 97 -> 75 us) and short ones want the finest cut (5k-256^2: 16 / 21 / 32 us). From N and the
// per-view tile count only, for the reason above.
int seg_shift_for(int N, int tiles_per_view) {
    if (ov(OV_SEG_SHIFT) >= 6 && ov(OV_SEG_SHIFT) <= 8) return ov(OV_SEG_SHIFT);   // test hook
    if (fwd_sequential_for(N, tiles_per_view)) return 6;
    const double x = 4.0 * (double)N / (double)(tiles_per_view > 0 ? tiles_per_view : 1);   // ~ list length of an average tile
    return x <= 512.0 ? 6 : 7;
}
// Dynamic LDS requested on top of the serial walk's own 14 KiB. The kernel does not use it: 42 KiB per workgroup keeps the walk at
// THREE workgroups per CU instead of the five its registers allow. A dense single view has fewer busy tiles than the chip has
// slots (1M Gaussians / 800^2: 777 of 2 500 tiles, heaviest first), all of them are placed at once, and with five slots some CUs
// receive four or five of the heavy ones while others hold one or two; with three the 768 heaviest land three per CU. Measured on
// one box, forward compositing, 5 / 4 / 3 / 3 (50 KiB) / 2 per CU: 0.1458 / 0.1445 / 0.1419 / 0.1419 / 0.1949 ms -- two per CU
// starves it (the walk is latency-bound per wave and wants waves). 1M trained-like, 100k / 800^2, 250k / 512^2 at three: unchanged
// (0.1054 / 0.1051, 0.0866 / 0.0869, 0.1066 / 0.1065). Batches of views bring more busy tiles than slots and keep the five
// (not measured with three). Test hook "fwd_lds_kb": 0 = none, n = n KiB.
size_t fwd_serial_lds_pad(int B) {
    const int kb = ov(OV_FWD_LDS_KB);
    if (kb >= 0) return (size_t)(kb > 144 ? 144 : kb) * 1024;
    return B == 1 ? (size_t)28 * 1024 : 0;
}
// forward compositing kernel: 0 = per view (finish_impl), 1 = 8x8 block lists, 2 = quad lists (test hook "fwd_lists")
int fwd_kernel_env() { const int v = ov(OV_FWD_LISTS); return v == 1 || v == 2 ? v : 0; }
// segment forward's hints: 0 = on, 1 = off (every segment composited), 2 = TEST: every segment behind a tile's first is skipped, so
// that the chaining kernel has to walk them all (same results, bit for bit) (test hook "fwd_hints")
int fwd_hint_env() { const int v = ov(OV_FWD_HINTS); return v == 1 || v == 2 ? v : 0; }
std::atomic<uint32_t> g_epoch{0x5eed};                  // launch tag of the segment forward's hints
// launch tag of K1's "counters cleared" flag: process-wide and never repeated, so a stale word in recycled scratch is never it
std::atomic<unsigned long long> g_k1_epoch{0x6b31000000000000ull ^ ((unsigned long long)(uintptr_t)&g_epoch << 8)};

}  // namespace

extern "C" const char* gsr_last_error(void) { return g_err; }

extern "C" int gsr_profile_enable(int on) { g_prof.on = on != 0; return 0; }
extern "C" int gsr_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& p : g_prof.pending) { g_prof.pool.push_back(p.start); g_prof.pool.push_back(p.stop); }
    g_prof.pending.clear(); g_prof.rows.clear();
    return 0;
}
extern "C" int gsr_profile_read(int cap, const char** names, float* total_ms, int* launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& p : g_prof.pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.stop) == hipSuccess && hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            ProfRow* row = nullptr;
            for (auto& r : g_prof.rows) if (r.name == p.name || strcmp(r.name, p.name) == 0) { row = &r; break; }
            if (!row) { g_prof.rows.push_back({p.name, 0.0, 0}); row = &g_prof.rows.back(); }
            row->ms += ms; row->launches += 1;
        }
        g_prof.pool.push_back(p.start); g_prof.pool.push_back(p.stop);
    }
    g_prof.pending.clear();
    int n = 0;
    for (auto& r : g_prof.rows) {
        if (n >= cap) break;
        if (names) names[n] = r.name;
        if (total_ms) total_ms[n] = (float)r.ms;
        if (launches) launches[n] = r.launches;
        ++n;
    }
    return n;
}
extern "C" const char* gsr_version(void) { return "gsr 0.5 (gfx950, wave64, 16x16 bins / 8x8 wave blocks, depth-segmented forward, ABI 6)"; }
extern "C" int gsr_abi_version(void) { return GSR_ABI_VERSION; }
extern "C" int gsr_testing_override(const char* name, int32_t value) {
    if (!name) return fail(-1, "override name is NULL%s", "");
    for (int k = 0; k < OV_COUNT; ++k)
        if (strcmp(name, kOvNames[k]) == 0) { g_ov[k].store(value, std::memory_order_relaxed); return 0; }
    return fail(-1, "unknown override '%s'", name);
}

extern "C" size_t gsr_geom_bytes(int32_t N, int32_t H, int32_t W) { return geom_layout(N, H, W).total; }
// final_T | n_contrib | totals[5] (the five per-pixel sums without background)
extern "C" size_t gsr_img_bytes(int32_t H, int32_t W) { return align_up((size_t)H * W * 4) * 7; }

// ---- forward -----------------------------------------------------------------------------------
// begin : memset, per-Gaussian stage (K1), tile scan + depth-major work list (K2) on `stream`, then an async copy of the
//         counters (M_ref, V, M, longest list, ..., per-view M_ref / V) into this thread's pinned block, followed by a
//         second copy that overwrites the arrival flag.
// finish: binning, sort, segment forward (K5a), segment chaining (K5b) for lists of up to `cap` instances.
// All of it takes B views of the same size (gsr_forward_views): ONE launch of every kernel covers the B cameras, the
// counters are totals over the views.
namespace {
int begin_impl(const GsrView* views, int B, int32_t N, int32_t K,
               const float* means3D, const float* shs, const float* colors_precomp,
               const float* opacities, const float* scales, const float* rotations,
               const float* cov3D_precomp, int32_t* radii, char* gbuf, int shift,
               unsigned long long* host_counters, int counter_words, bool scan_in_scatter, bool with_acc /* geom_layout */, hipStream_t stream) {
    const GsrView* view = views;
    unsigned long long* host_counters_dev = nullptr;       // the pinned block as the device addresses it
    HIP_TRY(hipHostGetDevicePointer((void**)&host_counters_dev, host_counters, 0));
    host_counters[kCounterWords] = kFlagSentinel;          // overwritten by the scan kernel behind the counters (wait_counters polls it)
    __sync_synchronize();
    ViewTab tab;
    memset(&tab, 0, sizeof(tab));
    for (int v = 0; v < B; ++v) tab.v[v] = make_view(views + v);
    const ViewConst& vc = tab.v[0];
    const int H = vc.H, W = vc.W;
    const GeomLayout GL = geom_layout(N, H, W, B, with_acc);
    const int T = GL.nTiles;

    SplatRec* recs = (SplatRec*)(gbuf + GL.recs);
    EmitRec* emit = (EmitRec*)(gbuf + GL.emit);
    uint32_t* tile_count = (uint32_t*)(gbuf + GL.tile_count);
    unsigned long long* counters = (unsigned long long*)(gbuf + GL.counters);

    const int hist_in_lds = T <= hist_lds_max_tiles();
    // tile_count | cursor | counters start from zero: cleared by K1's first workgroup (no launch of its own); without Gaussians
    // there is no K1
    if (N <= 0) {
        prof_begin(stream);
        HIP_TRY(hipMemsetAsync(gbuf + GL.tile_count, 0, GL.tile_off - GL.tile_count, stream));
        prof_end(stream, "memset_fwd");
    }
    const uint32_t zero_words = (uint32_t)((GL.tile_off - GL.tile_count) / 4);
    const uint32_t flag_word = (uint32_t)((GL.counters - GL.tile_count) / 4) + 2u * (uint32_t)kZeroedFlag;
    const unsigned long long epoch = g_k1_epoch.fetch_add(1ull) + 1ull;

    // K1 is a persistent grid (per-workgroup tile histogram + statistics): four workgroups per CU (115 VGPRs), every wave walks
    // batches of 64 Gaussians. The grid is SHRUNK to ceil(batches / rounds) so that no workgroup walks one batch more than the
    // others (at 1M Gaussians: 977 workgroups x 4 batches; 1280 workgroups gave 67 of them a fourth batch and every tile counter
    // 1280 flush atomics instead of 977: 0.102 -> 0.096 ms by the grid alone). Test hook "k1_grid" pins the grid (A/B runs).
    const int grid_pre = k1_grid_for(N);
    if (N > 0) {
        const size_t hist_bytes = hist_in_lds ? (((size_t)T * 4 + 15) & ~(size_t)15) : 0;
        const size_t lds = hist_bytes + (size_t)4 * GSR_K1_WSLICE * 4;      // + each wave's slice: SH staging rows, then its records on their way out
        if (lds > 160 * 1024) return fail(-1, "preprocess needs more than 160 KiB of LDS%s", "");
        auto k1 = vc.raw_act ? gsr_preprocess_fwd<true> : gsr_preprocess_fwd<false>;
        if (lds > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        prof_begin(stream); hipLaunchKernelGGL(k1, dim3(grid_pre, B), dim3(256), lds, stream, tab, N, K, means3D, shs, view->shs_rest,
                           colors_precomp, opacities, scales, rotations, cov3D_precomp, recs, emit, radii,
                           tile_count, (unsigned long long*)(gbuf + GL.block_stats), hist_in_lds, (uint8_t*)(gbuf + GL.flags8),
                           (uint32_t*)(gbuf + GL.tile_count), zero_words, flag_word, epoch, (uint32_t*)(gbuf + GL.wg_base),
                           (uint32_t*)(gbuf + group_base_off(GL, N, B)), k1_group_for(N), (uint32_t*)(gbuf + GL.arrive));
        LAUNCH_CHECK(view, stream, "preprocess_fwd");
    }
    if (scan_in_scatter) return 0;                         // (finish_impl: a workgroup of gsr_scatter runs K2 beside the scatter)
    // one single-workgroup kernel: scan of the counts, K1's statistics, the segment forward's depth-major work list
    // (the tile counts pass through LDS when they fit beside the kernel's 9 KiB: one trip to memory instead of one per phase)
    const int scan_words = (size_t)GL.allTiles + 1 <= (48 * 1024 - sizeof(TileScanLds)) / 4 ? GL.allTiles + 1 : 0;
    prof_begin(stream); hipLaunchKernelGGL(gsr_tile_scan, dim3(1), dim3(1024), (size_t)scan_words * 4, stream, tile_count, (uint32_t*)(gbuf + GL.tile_off), GL.allTiles, counters,
                       (uint32_t*)(gbuf + GL.tile_seg), shift, (const unsigned long long*)(gbuf + GL.block_stats), N > 0 ? grid_pre * B : 0, B,
                       (uint32_t*)(gbuf + GL.order), (uint2*)(gbuf + GL.order_span), (uint32_t*)(gbuf + GL.level_off), host_counters_dev, counter_words, kCounterWords, scan_words);
    LAUNCH_CHECK(view, stream, "tile_scan");
    // (the scan kernel stores the counters and then the arrival flag into the pinned block itself: no copy kernels and no event
    // marker in the stream -- the marker alone was a 5 us hole in front of the scatter)
    return 0;
}

// The per-tile sort takes a list of any length in either of its two kernels (a list beyond the LDS buffer is streamed through it): the
// predicted longest list only picks the faster kernel, it can no longer be WRONG (rounds 2-5: a frame whose longest list outgrew the
// predicted class repeated binning, sort and compositing).
int sort_class(unsigned long long) { return 0; }

// Whether the per-Gaussian backward of this problem visits the live Gaussians only (gsr_preprocess_bwd_compact, backward_impl) -- and
// with it whether somebody has to clear the gradient arrays on the side. Known from the problem's shape: the forward asks too.
bool k6_compact_for(const GsrView* view, int B, int32_t N, int32_t K, bool has_shs) {
    const size_t grad_bytes = (size_t)N * (size_t)(3 * (has_shs ? K : 1) + 14) * 4;
    return B == 1 && !view->shs_rest && (ov(OV_K6_COMPACT) == 1 || (ov(OV_K6_COMPACT) < 0 && grad_bytes >= ((size_t)64 << 20)));
}

// Binning, sort and compositing for lists of up to `cap` instances whose longest is assumed <= `maxc`.
// cap / maxc are either the exact counters (the host has waited for them) or forward_impl's prediction; in the second
// case every kernel that touches the lists checks the true M and the true longest list and leaves when they do not fit.
int finish_impl(const GsrView* views, int B, int32_t N, float* out_color, float* out_depth, float* out_alpha,
                char* gbuf, char* ibuf, GsrAlloc bin, int shift,
                const unsigned long long* per_view /* (M_ref, V) of every view */,
                unsigned long long cap, unsigned long long maxc, bool prepare_bwd, bool scan_in_scatter, int counter_words, hipStream_t stream,
                ZeroSide side = ZeroSide{nullptr, 0u, 0u} /* GsrView.grad_clear, when the serial walk is to clear it (forward_impl) */,
                bool poison = false /* an asynchronous forward: lists that do not fit `cap` / `maxc` leave NaN images, not unwritten ones */) {
    const GsrView* view = views;
    const ViewConst vc = make_view(view);
    const int H = vc.H, W = vc.W;
    const GeomLayout GL = geom_layout(N, H, W, B, prepare_bwd);
    const int T = GL.nTiles, TA = GL.allTiles;
    ViewSplit vs = make_split(views, B, N, T, H, W);

    SplatRec* recs = (SplatRec*)(gbuf + GL.recs);
    EmitRec* emit = (EmitRec*)(gbuf + GL.emit);
    uint32_t* cursor = (uint32_t*)(gbuf + GL.cursor);
    uint32_t* tile_off = (uint32_t*)(gbuf + GL.tile_off);
    uint32_t* tile_seg = (uint32_t*)(gbuf + GL.tile_seg);
    uint32_t* order = (uint32_t*)(gbuf + GL.order);
    unsigned long long* counters = (unsigned long long*)(gbuf + GL.counters);
    float* final_T = (float*)ibuf;
    uint32_t* n_contrib = (uint32_t*)(ibuf + align_up((size_t)H * W * 4));
    float* totals = (float*)(ibuf + 2 * align_up((size_t)H * W * 4));

    const int hist_in_lds = T <= hist_lds_max_tiles();
    const unsigned long long M = cap;
    if (M >= 0xfffffff0ull) return fail(-5, "too many tile instances (%s%lld)", "", (long long)M);
    const uint32_t maxc_cap = 0xffffffffu;              // (the sort takes lists of any length: the compositing kernels no longer leave on a longer one than predicted)

    const bool sequential = fwd_sequential_for(N, T);
    // GSR_VIEW_NO_BACKWARD + the serial walk: no checkpoints, no quad masks, no records in the scratch (the segmented mode composites
    // THROUGH its records and keeps them)
    const bool keep_state = prepare_bwd || !sequential;
    const BinLayout BL = bin_layout((size_t)M, TA, shift, keep_state);
    if (BL.items >= 0x7fffffffull) return fail(-5, "too many depth segments (%s%lld)", "", (long long)BL.items);
    const unsigned long long no_state_bit = (keep_state ? 0ull : (1ull << 63)) | (poison ? (1ull << 62) : 0ull);
    const uint32_t sink_rec = keep_state ? (uint32_t)BL.items : 0u;     // the first of the three spare records: the serial walk's store sink
    char* bbuf = (char*)bin.resize(bin.ctx, BL.total);
    if (!bbuf) return fail(-4, "bin scratch allocation failed%s", "");
    unsigned long long* entries = (unsigned long long*)(bbuf + BL.entries);
    uint32_t* sorted_ids = (uint32_t*)(bbuf + BL.ids);
    float* ckpt = (float*)(bbuf + BL.ckpt);

    if (M > 0) {
        const size_t lds = hist_in_lds ? (size_t)T * 4 : 0;
        const uint32_t* level_off = (const uint32_t*)(gbuf + GL.level_off);
        uint4* items = (uint4*)(bbuf + BL.item_recs);
        const uint32_t items_cap = sequential ? 0u : (uint32_t)BL.items;   // the serial walk takes its tiles from `order`: no work items
        prof_begin(stream);
        if (hist_in_lds) {
            // K1's grid and Gaussian -> workgroup assignment: every workgroup continues the list ranges its K1 twin reserved
            ScanFold fold;
            memset(&fold, 0, sizeof(fold));
            size_t lds_sc = lds;
            if (scan_in_scatter) {
                unsigned long long* host_dev = nullptr;
                HIP_TRY(hipHostGetDevicePointer((void**)&host_dev, g_pinned, 0));
                fold.on = 1; fold.tile_count = (const uint32_t*)(gbuf + GL.tile_count); fold.tile_off_w = tile_off; fold.tile_seg_w = tile_seg;
                fold.block_stats = (const unsigned long long*)(gbuf + GL.block_stats); fold.nblocks = k1_grid_for(N) * B;
                fold.order_w = order; fold.order_span = (uint2*)(gbuf + GL.order_span); fold.level_off_w = (uint32_t*)(gbuf + GL.level_off);
                fold.host_out = host_dev; fold.host_words = counter_words; fold.host_flag = kCounterWords;
                const size_t scan_base = (sizeof(TileScanLds) + 15) & ~(size_t)15;
                if (scan_base + ((size_t)TA + 1) * 4 <= 24 * 1024) fold.lds_words = TA + 1;      // (every workgroup of the launch is given the same dynamic LDS: keep it small)
                if (lds_sc < scan_base + (size_t)fold.lds_words * 4) lds_sc = scan_base + (size_t)fold.lds_words * 4;
            }
            const int G = k1_group_for(N), groups = (k1_grid_for(N) + G - 1) / G;
            auto sc = G == 4 ? gsr_scatter<4> : gsr_scatter<1>;
            if (lds_sc > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)sc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc));
            hipLaunchKernelGGL(sc, dim3(groups + (scan_in_scatter ? 1 : 0), B), dim3(256 * G), lds_sc, stream, N, emit, tile_off, (const uint32_t*)(gbuf + group_base_off(GL, N, B)), entries,
                               vc.gx, T, (uint32_t)M, counters, level_off, order, tile_seg, shift, items, items_cap, k1_grid_for(N), fold,
                               (const unsigned long long*)(gbuf + GL.block_stats));
        } else {
            const int grid_sc = (int)fmin((double)((N + 255) / 256), 512.0);
            hipLaunchKernelGGL(gsr_scatter_global, dim3(grid_sc, B), dim3(256), 0, stream, N, emit, tile_off, cursor, entries,
                               vc.gx, T, (uint32_t)M, counters, level_off, order, tile_seg, shift, items, items_cap);
        }
        LAUNCH_CHECK(view, stream, "scatter");
        // per-tile sort: ONE launch. Lists of up to 2 048 entries sort on 256 threads; when longer ones exist the 1 024-thread kernel takes
        // every list (a short one runs on its first waves; one beyond its 8 192-key LDS buffer streams its keys through it in portions:
        // gsr_binning.hip sort_long_tile). Rounds 1-5 had size classes in launches of their own: latency chains one behind the other --
        // 12 + 32 us at 1M Gaussians against 32 for the merged one -- and a 16 384 class that was launched whenever the previous
        // frame's longest list x 1.25 exceeded 8 192: 4.9 us of every step at the headline scene for a launch that found nothing.
        constexpr size_t lds_s = 2048 * 8 + (512 + 1 + 512 + 40) * 4;
        constexpr size_t lds_m = 8192 * 8 + (1920 + 1 + 1920 + 40) * 4;     // 81 060 B: TWO workgroups per CU (2 048 buckets: 82 084 B, 328 B too many for the second)
        if (int rc = once_per_device([]() -> hipError_t {   // > 64 KiB of dynamic LDS is an opt-in per device
                return hipFuncSetAttribute((const void*)gsr_tile_sort_bucket<8192, 1024, 1920>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);
            })) return rc;
        if (ov(OV_SORT_KERNEL) == 1 || (ov(OV_SORT_KERNEL) != 2 && maxc <= 2048)) {       // (test hook "sort_kernel": 1 / 2 = the 256- / 1 024-thread kernel whatever the lists)
            prof_begin(stream); hipLaunchKernelGGL((gsr_tile_sort_bucket<2048, 256, 512>), dim3(TA), dim3(256), lds_s, stream, tile_off, entries, sorted_ids, 0u, 0xffffffffu, counters, (uint32_t)M, (const uint2*)(gbuf + GL.order_span), (uint32_t)TA);
            LAUNCH_CHECK(view, stream, "tile_sort_small");
        } else {
            prof_begin(stream); hipLaunchKernelGGL((gsr_tile_sort_bucket<8192, 1024, 1920>), dim3(TA), dim3(1024), lds_m, stream, tile_off, entries, sorted_ids, 0u, 0xffffffffu, counters, (uint32_t)M, (const uint2*)(gbuf + GL.order_span), (uint32_t)TA);
            LAUNCH_CHECK(view, stream, "tile_sort");
        }
    }
    // Quad lists pay when splats are small against an 8x8 block (few of its 64 lanes blend a given Gaussian): the scene
    // statistic M_ref / V (reference tiles per visible Gaussian) decides, view by view -- a view renders with the kernel its
    // single-view call would use (the two give the same bits anyway: tests/test_parity_gpu.py).
    uint32_t mask_q = 0;
    const int fk = fwd_kernel_env();
    for (int v = 0; v < B; ++v) {
        const unsigned long long mr = per_view[2 * v], vv = per_view[2 * v + 1];
        if (fk == 2 || (fk == 0 && vv > 0 && mr <= 6ull * vv)) mask_q |= 1u << v;
    }
    const uint32_t mask_all = B >= 32 ? ~0u : ((1u << B) - 1u);
    uint32_t* plan_off = (uint32_t*)(gbuf + GL.plan_off);
    uint4* plan_tile = (uint4*)(bbuf + BL.plan_tile);
    // the backward's accumulators, cleared by the per-tile kernel's workgroups (the first launch of the two instantiations)
    float4* zero4 = (float4*)(gbuf + GL.g2d);
    uint32_t zero_n = prepare_bwd ? (uint32_t)(((size_t)B * (size_t)N * GSR_G2D_STRIDE * 4 + (size_t)B * GSR_LIVE_BYTES(N)) / 16) : 0u;   // accumulators + live flags
    const uint32_t zero_per = (zero_n + (uint32_t)TA - 1u) / (uint32_t)TA;       // float4s per workgroup (grid = TA)
    side.per = (side.n4 + (uint32_t)TA - 1u) / (uint32_t)TA;
    if (sequential) {
        // ---- K5s: the serial walk, one workgroup per tile
        const size_t fwd_lds = fwd_serial_lds_pad(B);
        if (fwd_lds > 32 * 1024) {
            HIP_TRY(hipFuncSetAttribute((const void*)gsr_render_fwd_serial<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds));
            HIP_TRY(hipFuncSetAttribute((const void*)gsr_render_fwd_serial<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds));
        }
        prof_begin(stream);
        // Two waves per 8x8 block (gsr_render_fwd_pair: a tester that fetches, tests and stages round r + 1 while the blender composites
        // round r) where the launch has fewer busy waves than the chip has SIMDs to keep fed: one view of 1 024 .. 2 047 tiles (512^2:
        // ~350 busy tiles x 4 waves = 1.4 per SIMD). Measured round 5, same box, forward compositing: 250k / 512^2 0.1060 -> 0.0897 ms;
        // at 2 500 tiles it does nothing (1M / 800^2 0.1404 -> 0.1385, trained-like 0.1032 -> 0.1033: three waves per SIMD already, and
        // the blender's and the tester's halves of a round are about equally long) and the plain walk stays. Same bits either way
        // (tests/test_parity_gpu.py::test_pair_forward_is_bit_identical_to_the_serial_walk). Test hook "fwd_mode": 1 = never, 3 = always.
        // ... and for a single view whose launch also stores the backward's gradient block (side.n4, forward_impl): with 248 MB of
        // zeros leaving under it the walk's gathers queue behind them and twice the waves hide more of that -- 1M / 800^2, same box,
        // serial -> pair: 0.1522 -> 0.1459 ms (step 0.5575 -> 0.5519); trained-like 0.1100 -> 0.1102.
        const bool pair = ov(OV_FWD_MODE) == 3 || (ov(OV_FWD_MODE) < 0 && B == 1 && (TA < 2048 || side.n4 > 0u));
        if (mask_q && pair) {
            vs.view_mask = mask_q;
            hipLaunchKernelGGL(gsr_render_fwd_pair, dim3(TA), dim3(512), 0, stream, tile_off, recs, sorted_ids, W, H, vc.gx,
                               out_color, out_depth, out_alpha, final_T, n_contrib, totals, ckpt, tile_seg, order, shift,
                               plan_off, plan_tile, counters + 4, (uint32_t)BL.plan_cap, (unsigned long long)mask_q | no_state_bit, sink_rec,
                               counters, (uint32_t)M, maxc_cap, vs, zero4, zero_n, zero_per, side);
            zero_n = 0u; side.n4 = 0u;
        } else if (mask_q) {
            vs.view_mask = mask_q;
            hipLaunchKernelGGL(gsr_render_fwd_serial<true>, dim3(TA), dim3(256), fwd_lds, stream, tile_off, recs, sorted_ids, W, H, vc.gx,
                               out_color, out_depth, out_alpha, final_T, n_contrib, totals, ckpt, tile_seg, order, shift,
                               plan_off, plan_tile, counters + 4, (uint32_t)BL.plan_cap, (unsigned long long)mask_q | no_state_bit, sink_rec,
                               counters, (uint32_t)M, maxc_cap, vs, zero4, zero_n, zero_per, side);
            zero_n = 0u; side.n4 = 0u;
        }
        if (mask_q != mask_all) {
            vs.view_mask = mask_all & ~mask_q;
            hipLaunchKernelGGL(gsr_render_fwd_serial<false>, dim3(TA), dim3(256), fwd_lds, stream, tile_off, recs, sorted_ids, W, H, vc.gx,
                               out_color, out_depth, out_alpha, final_T, n_contrib, totals, ckpt, tile_seg, order, shift,
                               plan_off, plan_tile, counters + 4, (uint32_t)BL.plan_cap, (unsigned long long)mask_q | no_state_bit, sink_rec,
                               counters, (uint32_t)M, maxc_cap, vs, zero4, zero_n, zero_per, side);
        }
        LAUNCH_CHECK(view, stream, "render_fwd");
        return 0;
    }
    if (M > 0) {
        // ---- K5a: every (tile, segment) composited on its own
        const uint32_t epoch = g_epoch.fetch_add(1u) + 1u;
        const int hint_mode = fwd_hint_env();
        unsigned long long* sat = (unsigned long long*)(gbuf + GL.sat);
        const uint32_t* level_off = (const uint32_t*)(gbuf + GL.level_off);
        uint4* items = (uint4*)(bbuf + BL.item_recs);
        // one workgroup per item (test hook "fwd_grid" = n: n workgroups striding through the list, an A/B switch -- measured 2x slower,
        // see the kernel)
        const size_t gmax = ov(OV_FWD_GRID) > 0 ? (size_t)ov(OV_FWD_GRID) : ~(size_t)0;
        const unsigned grid_a = (unsigned)(BL.items < gmax ? BL.items : gmax);
        prof_begin(stream);
        if (mask_q) {
            vs.view_mask = mask_q;
            hipLaunchKernelGGL(gsr_render_fwd_seg<true>, dim3(grid_a), dim3(256), 0, stream, items, level_off, tile_off, recs, sorted_ids, W, H, vc.gx,
                               ckpt, shift, sat, epoch, hint_mode, counters, (uint32_t)M, maxc_cap, vs);
        }
        if (mask_q != mask_all) {
            vs.view_mask = mask_all & ~mask_q;
            hipLaunchKernelGGL(gsr_render_fwd_seg<false>, dim3(grid_a), dim3(256), 0, stream, items, level_off, tile_off, recs, sorted_ids, W, H, vc.gx,
                               ckpt, shift, sat, epoch, hint_mode, counters, (uint32_t)M, maxc_cap, vs);
        }
        LAUNCH_CHECK(view, stream, "render_fwd");
    }
    // ---- K5b: chain the segments per pixel, outputs of the pixels that never stop, the backward's work list, walk items for K5c
    prof_begin(stream);
    uint2* walk_items = (uint2*)(bbuf + BL.walk_items);
    hipLaunchKernelGGL(gsr_render_fwd_combine, dim3(TA), dim3(256), 0, stream, tile_off, recs, sorted_ids, W, H, vc.gx,
                       out_color, out_depth, out_alpha, final_T, n_contrib, totals, ckpt, tile_seg, order, shift,
                       plan_off, plan_tile, counters + 4, (uint32_t)BL.plan_cap, walk_items, counters + kWalkCounter, counters, (uint32_t)M, maxc_cap, vs, zero4, zero_n, zero_per, poison ? 1 : 0);
    LAUNCH_CHECK(view, stream, "render_combine");
    if (M > 0) {   // ---- K5c: the pixels that stop inside a segment, one wave per (block, segment) item
        const unsigned grid_f = (unsigned)(TA < 2048 ? (TA < 64 ? 64 : TA) : 2048);
        prof_begin(stream);
        hipLaunchKernelGGL(gsr_render_fwd_fix, dim3(grid_f), dim3(256), 0, stream, (const uint2*)walk_items, (const unsigned long long*)(counters + kWalkCounter),
                           tile_off, recs, sorted_ids, W, H, vc.gx, out_color, out_depth, out_alpha, final_T, n_contrib, totals,
                           (const float*)ckpt, tile_seg, shift, counters, (uint32_t)M, maxc_cap, vs);
        LAUNCH_CHECK(view, stream, "render_fix");
    }
    return 0;
}

// waits until the scan kernel's counters have landed in the thread's pinned block (it stores the arrival flag behind them)
int wait_counters(volatile unsigned long long* pinned, hipStream_t stream) {
    volatile unsigned long long* f = pinned + kCounterWords;
    bool arrived = false;
    for (long it = 0; it < 20000000L && !arrived; ++it) { arrived = (*f != kFlagSentinel); if (!arrived) __builtin_ia32_pause(); }
    __sync_synchronize();
    if (!arrived) {                                       // (bounded poll: a stream that makes no progress surfaces as an error, not a hang)
        HIP_TRY(hipStreamSynchronize(stream));
        if (*f == kFlagSentinel) return fail(-2, "the forward's counters did not arrive%s", "");
    }
    return 0;
}

// The counters of this thread's pending asynchronous forward: waited for, checked against what its tail was launched for, filed in
// g_completed, and the shape's hint brought up to date. -6: the lists did not fit (the forward's outputs are NaN-filled).
int complete_pending() {
    const PendingFwd p = g_pending;
    g_pending.on = false;
    CompletedFwd c;
    c.serial = p.serial;
    int rc = wait_counters(g_pinned, p.stream);
    if (rc == 0) {
        c.M_ref = g_pinned[0]; c.V = g_pinned[1]; c.M = g_pinned[2]; c.maxc = g_pinned[3];
        if (c.M_ref >= (1ull << 62)) {
            (void)hipStreamSynchronize(p.stream);
            hint_drop(p.N, p.H, p.W, p.B);
            rc = fail(-2, "the per-Gaussian kernel never saw the tile counters cleared (device error)%s", "");
        } else {
            hint_update(p.N, p.H, p.W, p.B, c.M, c.maxc, g_pinned + 8);     // (also after an overflow: the next call then sizes from the true counts)
            if (c.M > p.cap || sort_class(c.maxc) > sort_class(p.capc))
                rc = fail(-6, "an asynchronous forward (GSR_VIEW_ASYNC_STATS) found more tile instances than its speculative capacity (%s%lld): "
                              "its images are NaN-filled and its backward yields zeros; repeat the step", "", (long long)c.M);
        }
    }
    c.rc = rc;
    g_completed = c;
    return rc;
}

int forward_impl(const GsrView* views, int B, int32_t N, int32_t K,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* opacities, const float* scales, const float* rotations,
                 const float* cov3D_precomp,
                 float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                 GsrAlloc geom, GsrAlloc bin, GsrAlloc img,
                 GsrStats* stats, gsr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = check_views(views, B)) return rc;
    const GsrView* view = views;
    if (int rc = check_inputs(N, K, view, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)) return rc;
    if (!out_color || !out_depth || !out_alpha || (N > 0 && !radii)) return fail(-1, "output pointers are required%s", "");
    if (!geom.resize || !bin.resize || !img.resize) return fail(-1, "scratch allocators are required%s", "");
    // the one host round trip of the forward: how many (tile,Gaussian) instances to allocate
    if (!g_pinned) HIP_TRY(hipHostMalloc((void**)&g_pinned, (kCounterWords + 1) * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
    // the previous asynchronous forward of this thread: its counters must have landed before the pinned block is re-armed (the GPU
    // has long passed that point: this is the check, not a wait) -- and ITS failure is reported here, before anything new is enqueued
    if (g_pending.on) { if (int rcp = complete_pending()) return rcp; }
    const ViewConst vcs = make_view(view);
    const bool with_acc = N > 0 && !(view->flags & GSR_VIEW_NO_BACKWARD);
    const GeomLayout GLs = geom_layout(N, vcs.H, vcs.W, B, with_acc);
    const int shift = seg_shift_for(N, GLs.nTiles);
    // The backward accumulates its screen-space gradients with atomics into [B][N][12] floats that must start from zero -- 48 MB at
    // 1M Gaussians, a 10 us fill in front of gsr_render_bwd_q2 in round 3. The forward's compositing leaves HBM idle: every workgroup
    // of its per-tile kernel (gsr_render_fwd_serial / gsr_render_fwd_combine) clears a slice of them behind its own work. (A fill on
    // a second stream was measured first: the two cross-stream waits cost 15 us of bubbles, more than the fill.)
    const bool prepare = N > 0 && !(view->flags & GSR_VIEW_NO_BACKWARD);
    char* gbuf = (char*)geom.resize(geom.ctx, GLs.total);                          // (no backward: the layout holds no accumulators)
    char* ibuf = (char*)img.resize(img.ctx, gsr_img_bytes(vcs.H, vcs.W) * (size_t)B);
    if (!gbuf || !ibuf) return fail(-4, "scratch allocation failed%s", "");
    // GsrView.grad_clear: the array the backward's outputs will be carved from, cleared under the serial walk when the backward would
    // otherwise clear it under ITS compositing kernel (same rule: k6_compact_for). GsrStats.bwd_prepared = 2 tells the caller.
    ZeroSide side{nullptr, 0u, 0u};
    if (prepare && ov(OV_GRAD_CLEAR) != 0 && view->grad_clear && view->grad_clear_floats > 0 && !((uintptr_t)view->grad_clear & 15) && !(view->grad_clear_floats & 3) &&
        (unsigned long long)view->grad_clear_floats / 4 < 0x7fffffffull && k6_compact_for(view, B, N, K, shs != nullptr) &&
        fwd_sequential_for(N, GLs.nTiles)) {
        side.p = reinterpret_cast<float4*>(view->grad_clear);
        side.n4 = (uint32_t)(view->grad_clear_floats / 4);
    }
    // (a COPY of the table entry: the table is updated further down)
    FwdHint hintv;
    if (const FwdHint* hp = (ov(OV_SPECULATE) != 0 && N > 0) ? hint_find(N, vcs.H, vcs.W, B) : nullptr) hintv = *hp;
    const FwdHint& g_hint = hintv;
    const bool spec = g_hint.valid;
    // GSR_VIEW_ASYNC_STATS: a speculative forward returns as soon as everything is enqueued -- no wait for K2's counters in the steady
    // state. The capacity then comes from the LARGEST of the recent calls of the shape + 50 % (a list that still does not
    // fit leaves NaN images and error -6 at the thread's next call: gsr.h).
    const bool async = spec && stats != nullptr && (view->flags & GSR_VIEW_ASYNC_STATS) != 0;
    // K2 (scan of the tile counts, statistics, tile order, the counters for the host) inside the scatter's launch: when the forward is
    // enqueued in one go (speculation), the tile counters sit in LDS (no global cursors) and the compositing takes its tiles from
    // `order` (the depth-major item list of the segmented mode is written BY the scatter FROM K2's results). Test hook "scan_fold" = 0: never.
    // ... and when the forward is long enough for the host: its one wait (for K2's counters) ends a scatter later with the fold, and what it
    // then has to enqueue (the caller's loss, the backward) must still arrive before the compositing ends -- 250k / 512^2 and 100k / 800^2
    // lost 50 us per step to GPU idle time with it (their sort + compositing take ~0.1 ms), 1M / 800^2 gains 6-13: from 2M instances on.
    // (and for up to 4 096 tiles in the launch: K2 on 256 threads is the tail of the scatter beyond that -- 8 x 1 024 tiles: +20 us)
    const bool fold = spec && GLs.nTiles <= hist_lds_max_tiles() && GLs.allTiles <= 4096 && fwd_sequential_for(N, GLs.nTiles) &&
                      (ov(OV_SCAN_FOLD) == 1 || (ov(OV_SCAN_FOLD) < 0 && g_hint.M >= 2000000ull));
    if (int rc = begin_impl(views, B, N, K, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                            radii, gbuf, shift, g_pinned, 8 + 2 * B, fold, with_acc, stream)) {
        (void)hipStreamSynchronize(stream);               // nothing may still write the pinned block when the next call re-arms it
        return rc;
    }
    // Speculation (default; test hook "speculate" = 0 turns it off): the previous call of this thread on the same problem shape predicts
    // M (+25 %), the sort classes and each view's kernel; binning / sort / compositing are enqueued at once -- the GPU runs the
    // forward back to back -- and only then does the host wait for the counters, repeating the tail when the prediction was too
    // small (every kernel that touches the lists leaves when the true M or the true longest list exceeds what was launched for:
    // tests/test_parity_gpu.py::test_speculative_forward_recovers_from_mispredictions). Measured, ms per fwd+bwd, wait-first /
    // speculative with the event / speculative with the flag: 5k-256^2 0.402 / 0.396 / 0.313, 250k-512^2 0.366 / 0.352 / 0.354,
    // 100k-800^2 0.369 / 0.348 / 0.350, 1M-800^2 0.751 / 0.750 / 0.747 (round 2).
    unsigned long long cap = 0, capc = 0;
    int rc = 0;
    if (spec) {
        cap = async ? g_hint.M_hi + g_hint.M_hi / 2 + 4096 : g_hint.M + g_hint.M / 4 + 4096;
        const unsigned long long c = async ? g_hint.maxc_hi + g_hint.maxc_hi / 2 + 64 : g_hint.maxc + g_hint.maxc / 4 + 64;
        capc = c <= 2048 ? 2048 : ~0ull;                      // (which of the two sort kernels; the compositing kernels take lists of any length)
        rc = finish_impl(views, B, N, out_color, out_depth, out_alpha, gbuf, ibuf, bin, shift, g_hint.per_view, cap, capc, prepare, fold, 8 + 2 * B, stream, side, async);
        if (rc && fold) {                                 // K2 never ran: nothing will ever arrive in the pinned block
            (void)hipStreamSynchronize(stream);
            return rc;
        }
        if (async && rc == 0) {                           // the counters are waited for by whoever needs them next (complete_pending)
            g_pending.on = true; g_pending.serial = ++g_async_serial;
            g_pending.N = N; g_pending.H = vcs.H; g_pending.W = vcs.W; g_pending.B = B;
            g_pending.cap = cap; g_pending.capc = capc; g_pending.stream = stream;
            stats->num_instances = -1; stats->num_instances_ref = -1; stats->num_visible = -1; stats->max_tile_count = -1;
            stats->bin_capacity = (int64_t)cap; stats->seg_shift = shift;
            stats->bwd_prepared = prepare ? (side.n4 ? 2 : 1) : -1;
            stats->speculated = 1; stats->pending = g_pending.serial;
            return 0;
        }
    }
    if (int rcw = wait_counters(g_pinned, stream)) return rcw;    // also on a failed tail: nothing may stay pending on the pinned block
    if (rc) return rc;
    const unsigned long long M_ref = g_pinned[0], V = g_pinned[1], M = g_pinned[2], maxc = g_pinned[3];
    if (M_ref >= (1ull << 62)) {                          // a workgroup of K1 gave up waiting for the cleared tile counters (gsr_preprocess.hip)
        (void)hipStreamSynchronize(stream);
        hint_drop(N, vcs.H, vcs.W, B);
        return fail(-2, "the per-Gaussian kernel never saw the tile counters cleared (device error)%s", "");
    }
    bool speculated = spec;                               // the tail enqueued before the wait is the one that counts
    if (spec && (M > cap || sort_class(maxc) > sort_class(capc))) {
        // misprediction: the kernels above left without writing; clear what the scatter accumulates into
        HIP_TRY(hipMemsetAsync(gbuf + GLs.cursor, 0, GLs.counters - GLs.cursor, stream));
        cap = 0;
        speculated = false;
    }
    if (!spec || cap == 0) {
        cap = M;
        rc = finish_impl(views, B, N, out_color, out_depth, out_alpha, gbuf, ibuf, bin, shift, g_pinned + 8, M, maxc, prepare, false, 8 + 2 * B, stream, side);
    }
    if (stats) { stats->num_instances = (int64_t)M; stats->num_instances_ref = (int64_t)M_ref; stats->num_visible = (int64_t)V;
                 stats->max_tile_count = (int64_t)maxc; stats->bin_capacity = (int64_t)cap; stats->seg_shift = shift;
                 stats->bwd_prepared = (prepare && rc == 0) ? (side.n4 ? 2 : 1) : (prepare ? 0 : -1);
                 stats->speculated = speculated ? 1 : 0; stats->pending = 0; }
    if (rc == 0) hint_update(N, vcs.H, vcs.W, B, M, maxc, g_pinned + 8);
    return rc;
}

int backward_impl(const GsrView* views, int B, int32_t N, int32_t K,
                  const float* means3D, const float* shs, const float* colors_precomp,
                  const float* opacities, const float* scales, const float* rotations,
                  const float* cov3D_precomp, const int32_t* radii,
                  const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                  const void* geom, const void* bin, const void* img,
                  const GsrStats* fwd_stats,
                  float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors,
                  float* dL_dopacities, float* dL_dscales, float* dL_drotations,
                  float* dL_dcov3D, GsrAlloc tmp, gsr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int rc = check_views(views, B)) return rc;
    const GsrView* view = views;
    if (int rc = check_inputs(N, K, view, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)) return rc;
    if (N == 0) return 0;
    if (!geom || !bin || !img || !radii) return fail(-1, "forward state (geom/bin/img/radii) is required%s", "");
    if (!dL_dmeans3D || !dL_dmeans2D || !dL_dopacities) return fail(-1, "dL_dmeans3D/dL_dmeans2D/dL_dopacities are required%s", "");
    if (shs && !dL_dshs) return fail(-1, "dL_dshs is required with shs%s", "");
    if (view->shs_rest && (!shs || !view->dL_dshs_rest || K < 2)) return fail(-1, "split SH input needs shs (features_dc), K >= 2 and GsrView.dL_dshs_rest%s", "");
    const ViewConst vc = make_view(view);
    const int H = vc.H, W = vc.W;
    const GeomLayout GL = geom_layout(N, H, W, B);
    const int T = GL.nTiles, TA = GL.allTiles;
    const ViewSplit vs = make_split(views, B, N, T, H, W);
    const char* gbuf = (const char*)geom;
    const char* ibuf = (const char*)img;
    const SplatRec* recs = (const SplatRec*)(gbuf + GL.recs);
    const uint32_t* tile_off = (const uint32_t*)(gbuf + GL.tile_off);
    const unsigned long long* counters = (const unsigned long long*)(gbuf + GL.counters);
    const float* final_T = (const float*)ibuf;
    const uint32_t* n_contrib = (const uint32_t*)(ibuf + align_up((size_t)H * W * 4));
    const float* totals = (const float*)(ibuf + 2 * align_up((size_t)H * W * 4));
    const uint32_t* tile_seg = (const uint32_t*)(gbuf + GL.tile_seg);
    // the layout of `bin` follows the capacity and the segment length of the forward: from the caller's GsrStats, else
    // read back from the device (blocking): counters[6] (left by the forward's scatter kernel) and counters[7]
    unsigned long long M = 0;
    int shift = 6;
    if (fwd_stats) {
        if (fwd_stats->bwd_prepared < 0) return fail(-1, "the forward ran with GSR_VIEW_NO_BACKWARD: it left no state for gsr_backward%s", "");
        // (an asynchronous forward's statistics are still pending: the capacity -- known when it was enqueued -- is all that is needed)
        M = (unsigned long long)((fwd_stats->num_instances > 0 || fwd_stats->pending != 0) ? fwd_stats->bin_capacity : 0);
        shift = (int)fwd_stats->seg_shift;
        if (shift < 6 || shift > 8) return fail(-1, "fwd_stats is not the GsrStats of a gsr_forward of this library version%s", "");
    } else {
        unsigned long long h[kCounterSlots];
        HIP_TRY(hipMemcpyAsync(h, counters, sizeof(h), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (h[2] > 0 && (h[GSR_CNT_QMASK] >> 63)) return fail(-1, "the forward ran with GSR_VIEW_NO_BACKWARD: it left no state for gsr_backward%s", "");
        M = h[2] > 0 ? h[6] : 0;
        shift = (int)(h[7] >> 32);
        if (shift < 6 || shift > 8) return fail(-1, "geom does not hold the state of a forward%s", "");
    }
    const uint32_t* sorted_ids = (const uint32_t*)bin;    // BinLayout.ids == 0

    const size_t g2d_view = (size_t)N * GSR_G2D_STRIDE;
    float* g2d = nullptr;
    const bool grads_cleared = fwd_stats && fwd_stats->bwd_prepared == 2;   // ... and GsrView.grad_clear with them
    // GSR_VIEW_DETERMINISTIC: the compositing kernel's sums meet in 64-bit fixed point ([views][N][10] u64 + the two words of
    // gsr_grad_absmax, from `tmp`, cleared here) and gsr_g2d_from_fixed turns them into the float accumulators: bit-identical
    // gradients from run to run (gsr_render.hip)
    const bool det = M > 0 && (view->flags & GSR_VIEW_DETERMINISTIC) != 0;
    const size_t det_bytes = det ? align_up((size_t)B * (size_t)N * GSR_Q2_ROW * 8 + 256) : 0;
    const bool own_acc = !(fwd_stats && fwd_stats->bwd_prepared >= 1);   // no GsrStats, or a second backward of the same forward
    const size_t acc_bytes = g2d_view * 4 * (size_t)B + (size_t)B * GSR_LIVE_BYTES(N);   // accumulators | live flags
    char* tbuf = nullptr;
    if (own_acc || det) {
        if (!tmp.resize) return fail(-1, "tmp allocator is required%s", "");
        tbuf = (char*)tmp.resize(tmp.ctx, det_bytes + (own_acc ? align_up(acc_bytes) : 0));
        if (!tbuf) return fail(-4, "tmp scratch allocation failed%s", "");
        prof_begin(stream);
        HIP_TRY(hipMemsetAsync(tbuf, 0, det_bytes + (own_acc ? acc_bytes : 0), stream));
        prof_end(stream, "memset_bwd");
    }
    unsigned long long* det64 = det ? (unsigned long long*)(tbuf + 256) : nullptr;
    uint32_t* det_gmax = det ? (uint32_t*)tbuf : nullptr;
    if (!own_acc) g2d = (float*)(const_cast<char*>(gbuf) + GL.g2d);   // cleared by the forward (forward_impl)
    else g2d = (float*)(tbuf + det_bytes);
    uint8_t* live = (uint8_t*)(g2d + g2d_view * (size_t)B);

    // ONE view, concatenated SH layout (gsr_preprocess_bwd_compact): the compositing kernel clears every gradient array on the side and
    // marks the Gaussians it hands a gradient to; K6 then visits those only -- a quarter of the 1M blob scene, 6 % of the trained-like
    // one -- instead of streaming all of them and storing 248 bytes each, most of them zeros.
    ZeroRegions zr;
    memset(&zr, 0, sizeof(zr));
    // Where it pays: the clearing costs the compositing kernel ~8 us per 100 MB (HBM writes under its gathers), so the gradient arrays
    // must be large for the streaming kernel's stores to be the bigger evil. Measured round 5, K6 + render_bwd, streaming -> compact:
    // 1M / SH 3 blob (248 MB of gradients, 27 % live) 0.265 -> 0.258 ms, trained-like (6 % live) 0.175 -> 0.134; 100k / SH 3 (25 MB)
    // 0.131 -> 0.139, 250k / SH 0 (17 MB) 0.093 -> 0.098, 5k 0.028 -> 0.032. Test hook "k6_compact": 0 = never, 1 = whenever possible.
    bool compact = M > 0 && k6_compact_for(view, B, N, K, shs != nullptr);
    if (compact) {
        const struct { float* p; size_t n; } arr[GSR_ZERO_MAX] = {
            {dL_dmeans3D, (size_t)N * 3}, {dL_dmeans2D, (size_t)N * 3}, {dL_dopacities, (size_t)N}, {shs ? dL_dshs : nullptr, (size_t)N * 3 * K},
            {dL_dcolors, (size_t)N * 3}, {dL_dscales, (size_t)N * 3}, {dL_drotations, (size_t)N * 4}, {dL_dcov3D, (size_t)N * 6}};
        size_t total4 = 0;
        for (int r = 0; r < GSR_ZERO_MAX; ++r) {
            if (!arr[r].p) continue;
            if ((uintptr_t)arr[r].p & 15) compact = false;        // (the float4 slices)
            // bwd_prepared == 2: the forward cleared view->grad_clear; every output must lie inside it (the contract of that field)
            if (grads_cleared && (arr[r].p < view->grad_clear || arr[r].p + arr[r].n > view->grad_clear + view->grad_clear_floats))
                return fail(-1, "bwd_prepared = 2 but a gradient output lies outside GsrView.grad_clear%s", "");
            zr.p[zr.count] = arr[r].p; zr.n4[zr.count] = (uint32_t)(arr[r].n / 4); zr.tail[zr.count] = (uint32_t)(arr[r].n & 3);
            total4 += arr[r].n / 4; ++zr.count;
        }
        if (total4 >= 0x7fffffffull) compact = false;
        zr.total4 = (uint32_t)total4;
        if (!compact || grads_cleared) memset(&zr, 0, sizeof(zr));   // (nothing to clear here: the forward's compositing kernel did it)
    }
    // A NULL incoming gradient stands for zeros (ABI 6: DreamGaussian's stage 1 never differentiates depth, main.py:198-275 -- the
    // wrapper used to build a zero image per backward for it). The kernel's loads stay unconditional: they go to a plane of the
    // forward's own per-pixel scratch (`totals`: 5 planes per view, the colour gradient's 3 fit) and the value is masked.
    uint32_t gnull = 0u;
    if (!dL_dcolor) { dL_dcolor = totals; gnull |= 1u; }
    if (!dL_ddepth) { dL_ddepth = totals; gnull |= 2u; }
    if (!dL_dalpha) { dL_dalpha = totals; gnull |= 4u; }
    if (det) {
        const size_t HWp = (size_t)H * W;
        const size_t npx = HWp * (size_t)B;
        prof_begin(stream);
        hipLaunchKernelGGL(gsr_grad_absmax, dim3((unsigned)((npx + 255) / 256 < 1024 ? (npx + 255) / 256 : 1024)), dim3(256), 0, stream,
                           dL_dcolor, dL_ddepth, dL_dalpha, gnull, HWp, B, vs, det_gmax);
        LAUNCH_CHECK(view, stream, "grad_absmax");
    }
    prof_begin(stream);
    if (M > 0) {
        const BinLayout BL = bin_layout((size_t)M, TA, shift);
        const float* ckpt = (const float*)((const char*)bin + BL.ckpt);
        // the work list ((tile, segment) up to each tile's deepest blended position) was left by the forward's chaining kernel
        const uint4* plan_tile = (const uint4*)((const char*)bin + BL.plan_tile);
        const uint32_t* plan_off = (const uint32_t*)(gbuf + GL.plan_off);
        const unsigned long long* plan_total = counters + 4;
        // one workgroup per (tile, segment) of the work list: at most M / 2^shift + tiles of them (M = the exact instance count when
        // the caller handed the forward's statistics back; the list's capacity otherwise)
        size_t gmax = BL.plan_cap;
        if (fwd_stats && fwd_stats->num_instances > 0) gmax = ((size_t)fwd_stats->num_instances >> shift) + (size_t)TA + 1;
        unsigned grid = (unsigned)(gmax < BL.plan_cap ? gmax : BL.plan_cap);
        if (ov(OV_BWD_GRID) > 0 && (unsigned)ov(OV_BWD_GRID) < grid) grid = (unsigned)ov(OV_BWD_GRID);   // (timing experiments only: items beyond the grid are dropped)
        // workgroup table: [2^shift][10] 64-bit fixed-point sums
        const size_t dyn = ((size_t)GSR_Q2_ROW * 8) << shift;
        hipLaunchKernelGGL(gsr_render_bwd_q2, dim3(grid), dim3(256), dyn, stream, tile_off, recs, sorted_ids, W, H, vc.gx,
                           final_T, n_contrib, totals, ckpt, tile_seg, dL_dcolor, dL_ddepth, dL_dalpha, g2d, shift,
                           plan_tile, plan_off, plan_total, vs, zr, live, gnull, det64, det_gmax);
    }
    LAUNCH_CHECK(view, stream, "render_bwd");
    if (det) {
        const size_t rows = (size_t)N * (size_t)B;
        prof_begin(stream);
        hipLaunchKernelGGL(gsr_g2d_from_fixed, dim3((unsigned)((rows + 255) / 256 < 4096 ? (rows + 255) / 256 : 4096)), dim3(256), 0, stream,
                           N, B, recs, det64, det_gmax, counters + 4, g2d, live);
        LAUNCH_CHECK(view, stream, "g2d_from_fixed");
    }

    const int k6_grid = ov(OV_K6_GRID) < 1 ? 2048 : ov(OV_K6_GRID);   // (test hook: A/B runs)
    // K6 runs on 128-thread workgroups when it stages SH rows (25 KiB of LDS each, six per CU): a batch is a chain -- inputs, staged
    // rows, arithmetic, per-Gaussian stores, row stores -- and more, smaller workgroups spread the chains' phases over the CU
    // (0.091 -> 0.088 ms at 1M; 256 threads when nothing is staged)
    const int k6_threads = (shs && K > 1) ? 128 : 256;
    const int grid_n = (int)fmin((double)((N + k6_threads - 1) / k6_threads), (double)k6_grid * (256 / k6_threads));
    const size_t lds = (shs && K > 1) ? (size_t)k6_threads * (3 * K + 1) * 4 : 0;
    if (lds > 160 * 1024) return fail(-1, "preprocess_bwd needs more than 160 KiB of LDS%s", "");
    auto k6 = vc.raw_act ? gsr_preprocess_bwd<true, false> : gsr_preprocess_bwd<false, false>;
    auto k6m = vc.raw_act ? gsr_preprocess_bwd<true, true> : gsr_preprocess_bwd<false, true>;
    if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k6, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // The parameter gradients are the SUM over the views in the order autograd accumulates B separate rasterizer calls (the node
    // created last runs first: last view first, then `earlier sum + this view`), so that the sums are bit-identical to the serial
    // loop; dL_dmeans2D stays per view ([B,N,3]). Nothing staged through LDS (K == 1 or precomputed colours): ONE launch that
    // reads each Gaussian once and runs through the cameras in registers. Otherwise one launch per view, adding to the first.
    ViewTab tab;
    memset(&tab, 0, sizeof(tab));
    const uint8_t* flags8 = (const uint8_t*)(gbuf + GL.flags8);
    // several views in ONE launch: always when nothing is staged; with staged SH rows when input + output rows of a workgroup fit
    // the CU's LDS twice over (two workgroups per CU at least): 50 KiB at 16 coefficients
    const size_t lds_multi = 2 * lds;
    if (compact) {
        tab.v[0] = make_view(views);
        const size_t lds_c = (shs && K > 1) ? (size_t)GSR_K6C_NT * (3 * K + 1) * 4 : 0;
        if (lds_c > 150 * 1024) return fail(-1, "preprocess_bwd needs more than 160 KiB of LDS%s", "");
        auto k6c = vc.raw_act ? gsr_preprocess_bwd_compact<true> : gsr_preprocess_bwd_compact<false>;
        if (lds_c > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k6c, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
        // a persistent grid of 128-thread workgroups, six per CU (25 KiB of staged rows each at 16 coefficients; 168 VGPRs): small
        // chunks balance the random number of live Gaussians a workgroup finds, and six chains overlap their phases
        const int rounds = (N + GSR_K6C_ROUND - 1) / GSR_K6C_ROUND;
        const int gcap = ov(OV_K6_GRID) < 1 ? 1536 : ov(OV_K6_GRID);
        prof_begin(stream);
        hipLaunchKernelGGL(k6c, dim3(rounds < gcap ? rounds : gcap), dim3(GSR_K6C_NT), lds_c, stream, tab, N, K, means3D, shs, colors_precomp, opacities, scales,
                           rotations, cov3D_precomp, flags8, g2d, live, dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors, dL_dopacities,
                           dL_dscales, dL_drotations, dL_dcov3D);
        LAUNCH_CHECK(view, stream, "preprocess_bwd_live");
    } else if (B > 1 && lds_multi <= 80 * 1024) {
        for (int v = 0; v < B; ++v) tab.v[v] = make_view(views + v);
        if (lds_multi > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k6m, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_multi));
        prof_begin(stream);
        hipLaunchKernelGGL(k6m, dim3(grid_n), dim3(k6_threads), lds_multi, stream, tab, 0, B, N, K, means3D, shs, view->shs_rest, view->dL_dshs_rest,
                           colors_precomp, opacities, scales, rotations, cov3D_precomp, radii, flags8, g2d,
                           dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors, dL_dopacities, dL_dscales, dL_drotations, dL_dcov3D, 0);
        LAUNCH_CHECK(view, stream, "preprocess_bwd");
    } else {
        for (int v = B - 1; v >= 0; --v) {
            tab.v[0] = make_view(views + v);
            prof_begin(stream);
            hipLaunchKernelGGL(k6, dim3(grid_n), dim3(k6_threads), lds, stream, tab, 0, 1, N, K, means3D, shs, view->shs_rest, view->dL_dshs_rest,
                               colors_precomp, opacities, scales, rotations, cov3D_precomp, radii + (size_t)v * N, flags8 + (size_t)v * N,
                               g2d + (size_t)v * g2d_view, dL_dmeans3D, dL_dmeans2D + (size_t)v * N * 3, dL_dshs, dL_dcolors, dL_dopacities,
                               dL_dscales, dL_drotations, dL_dcov3D, v == B - 1 ? 0 : 1);
            LAUNCH_CHECK(view, stream, "preprocess_bwd");
        }
    }
    return 0;
}
}  // namespace

extern "C" int gsr_forward(const GsrView* view, int32_t N, int32_t K,
                           const float* means3D, const float* shs, const float* colors_precomp,
                           const float* opacities, const float* scales, const float* rotations,
                           const float* cov3D_precomp,
                           float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                           GsrAlloc geom, GsrAlloc bin, GsrAlloc img,
                           GsrStats* stats, gsr_stream_t stream) {
    return forward_impl(view, 1, N, K, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                        out_color, out_depth, out_alpha, radii, geom, bin, img, stats, stream);
}

extern "C" int gsr_forward_complete(GsrStats* stats) {
    if (!stats) return fail(-1, "stats is NULL%s", "");
    if (stats->pending == 0) return 0;
    if (g_pending.on && g_pending.serial == stats->pending) (void)complete_pending();
    if (g_completed.serial != stats->pending)
        return fail(-1, "gsr_forward_complete: not the most recent asynchronous forward of the calling thread%s", "");
    stats->num_instances = (int64_t)g_completed.M; stats->num_instances_ref = (int64_t)g_completed.M_ref;
    stats->num_visible = (int64_t)g_completed.V; stats->max_tile_count = (int64_t)g_completed.maxc;
    stats->pending = 0;
    if (g_completed.rc == -6) stats->speculated = 0;
    return g_completed.rc;
}

extern "C" int gsr_backward(const GsrView* view, int32_t N, int32_t K,
                            const float* means3D, const float* shs, const float* colors_precomp,
                            const float* opacities, const float* scales, const float* rotations,
                            const float* cov3D_precomp, const int32_t* radii,
                            const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                            const void* geom, const void* bin, const void* img,
                            const GsrStats* fwd_stats,
                            float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors,
                            float* dL_dopacities, float* dL_dscales, float* dL_drotations,
                            float* dL_dcov3D, GsrAlloc tmp, gsr_stream_t stream) {
    return backward_impl(view, 1, N, K, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii,
                         dL_dcolor, dL_ddepth, dL_dalpha, geom, bin, img, fwd_stats, dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors,
                         dL_dopacities, dL_dscales, dL_drotations, dL_dcov3D, tmp, stream);
}

// ---- B cameras, one launch chain (include/gsr.h) -------------------------------------------------
extern "C" int gsr_forward_views(const GsrView* views, int32_t B, int32_t N, int32_t K,
                                 const float* means3D, const float* shs, const float* colors_precomp,
                                 const float* opacities, const float* scales, const float* rotations,
                                 const float* cov3D_precomp,
                                 float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                                 GsrAlloc geom, GsrAlloc bin, GsrAlloc img,
                                 GsrStats* stats, gsr_stream_t stream) {
    return forward_impl(views, B, N, K, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                        out_color, out_depth, out_alpha, radii, geom, bin, img, stats, stream);
}

extern "C" int gsr_backward_views(const GsrView* views, int32_t B, int32_t N, int32_t K,
                                  const float* means3D, const float* shs, const float* colors_precomp,
                                  const float* opacities, const float* scales, const float* rotations,
                                  const float* cov3D_precomp, const int32_t* radii,
                                  const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                                  const void* geom, const void* bin, const void* img,
                                  const GsrStats* fwd_stats,
                                  float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors,
                                  float* dL_dopacities, float* dL_dscales, float* dL_drotations,
                                  float* dL_dcov3D, GsrAlloc tmp, gsr_stream_t stream) {
    return backward_impl(views, B, N, K, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii,
                         dL_dcolor, dL_ddepth, dL_dalpha, geom, bin, img, fwd_stats, dL_dmeans3D, dL_dmeans2D, dL_dshs, dL_dcolors,
                         dL_dopacities, dL_dscales, dL_drotations, dL_dcov3D, tmp, stream);
}

extern "C" int gsr_mark_visible(const GsrView* view, int32_t N, const float* means3D,
                                uint8_t* visible, gsr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!view || !view->viewmatrix) return fail(-1, "viewmatrix is required%s", "");
    if (N < 0) return fail(-1, "N must be >= 0%s", "");
    if (N == 0) return 0;
    if (!means3D || !visible) return fail(-1, "means3D and visible are required%s", "");
    prof_begin(stream); hipLaunchKernelGGL(gsr_mark_visible_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, view->viewmatrix, view->flags & GSR_VIEW_VIEWMATRIX_T, N, means3D, visible);
    LAUNCH_CHECK(view, stream, "mark_visible");
    return 0;
}

extern "C" int gsr_extract_fields(int32_t N, const float* xyz, const float* opacity, const float* scaling,
                                  const float* rotation_raw, int32_t resolution, int32_t split_size,
                                  int32_t num_chunks, const float* axis, const float* box_lo, const float* box_hi,
                                  float* occ, float* norm_out, GsrAlloc tmp, gsr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N <= 0) return fail(-1, "extract_fields needs at least one Gaussian%s", "");
    if (!xyz || !opacity || !scaling || !rotation_raw) return fail(-1, "xyz, opacity, scaling and rotation_raw are required%s", "");
    if (!axis || !box_lo || !box_hi || !occ || !norm_out) return fail(-1, "axis, box_lo, box_hi, occ and norm_out are required%s", "");
    if (resolution < 1 || resolution > 1024) return fail(-1, "resolution must be in [1, 1024]%s", "");
    if (split_size < 1 || num_chunks < 1 || num_chunks > 255 ||
        (long long)split_size * num_chunks < resolution || (long long)split_size * (num_chunks - 1) >= resolution)
        return fail(-1, "num_chunks must equal ceil(resolution / split_size) and be <= 255%s", "");
    if (!tmp.resize) return fail(-1, "tmp allocator is required%s", "");
    size_t o = 0;
    const size_t o_bbox = o; o += align_up(6 * 4);
    const size_t o_rec = o; o += align_up((size_t)N * sizeof(FieldRec));
    const size_t o_rng = o; o += align_up((size_t)N * sizeof(uint2));
    char* buf = (char*)tmp.resize(tmp.ctx, o);
    if (!buf) return fail(-4, "tmp scratch allocation failed%s", "");
    uint32_t* bbox = (uint32_t*)(buf + o_bbox);
    FieldRec* recs = (FieldRec*)(buf + o_rec);
    uint2* range = (uint2*)(buf + o_rng);
    HIP_TRY(hipMemsetAsync(bbox, 0xff, 3 * 4, stream));
    HIP_TRY(hipMemsetAsync(bbox + 3, 0, 3 * 4, stream));
    GsrView dbg; memset(&dbg, 0, sizeof(dbg));
    const int grid_n = (N + 255) / 256;
    prof_begin(stream); hipLaunchKernelGGL(gsr_fields_bbox, dim3(grid_n < 256 ? grid_n : 256), dim3(256), 0, stream, N, xyz, opacity, bbox);
    LAUNCH_CHECK(&dbg, stream, "fields_bbox");
    prof_begin(stream); hipLaunchKernelGGL(gsr_fields_prep, dim3(grid_n), dim3(256), 0, stream, N, xyz, opacity, scaling, rotation_raw,
                       bbox, num_chunks, box_lo, box_hi, recs, range, norm_out);
    LAUNCH_CHECK(&dbg, stream, "fields_prep");
    const int slots = split_size * split_size * ((split_size + 3) / 4);    // rows of 4 z-consecutive samples
    prof_begin(stream); hipLaunchKernelGGL(gsr_fields_accumulate, dim3(num_chunks * num_chunks * num_chunks, (slots + 127) / 128), dim3(256), 0, stream,
                       N, recs, range, resolution, num_chunks, split_size, axis, occ);
    LAUNCH_CHECK(&dbg, stream, "fields_accumulate");
    return 0;
}

extern "C" int gsr_densify_stats(int32_t N, const float* grad_means2D, const int32_t* radii,
                                 float* xyz_gradient_accum, float* denom, float* max_radii2D, gsr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0) return fail(-1, "N must be >= 0%s", "");
    if (N == 0) return 0;
    if (!grad_means2D || !radii || !xyz_gradient_accum || !denom || !max_radii2D)
        return fail(-1, "grad_means2D, radii, xyz_gradient_accum, denom and max_radii2D are required%s", "");
    GsrView dbg; memset(&dbg, 0, sizeof(dbg));
    prof_begin(stream); hipLaunchKernelGGL(gsr_densify_stats_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, N, grad_means2D, radii,
                       xyz_gradient_accum, denom, max_radii2D);
    LAUNCH_CHECK(&dbg, stream, "densify_stats");
    return 0;
}

extern "C" int gsr_adam_step(int32_t count, const GsrAdamTensor* tensors, int32_t step, double beta1, double beta2, double eps,
                             gsr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (count < 0 || count > GSR_ADAM_MAX_TENSORS) return fail(-1, "gsr_adam_step takes 0..8 tensors per call%s", "");
    if (count == 0) return 0;
    if (!tensors || step < 1) return fail(-1, "tensors are required and step counts from 1%s", "");
    AdamArgs a;
    memset(&a, 0, sizeof(a));
    unsigned int blocks = 0;
    int k = 0;
    for (int i = 0; i < count; ++i) {
        const GsrAdamTensor& t = tensors[i];
        if (t.n < 0) return fail(-1, "negative tensor size%s", "");
        if (t.n == 0) continue;
        if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq) return fail(-1, "param, grad, exp_avg and exp_avg_sq are required%s", "");
        a.seg[k].param = t.param; a.seg[k].grad = t.grad; a.seg[k].exp_avg = t.exp_avg; a.seg[k].exp_avg_sq = t.exp_avg_sq;
        a.seg[k].n = (unsigned long long)t.n; a.seg[k].first_block = blocks;
        a.seg[k].neg_step_size = (float)(-((double)t.lr / (1.0 - pow((double)beta1, (double)step))));
        blocks += (unsigned int)((t.n + GSR_ADAM_PER_BLOCK - 1) / GSR_ADAM_PER_BLOCK);
        ++k;
    }
    if (k == 0) return 0;
    // torch takes these differences / quotients on Python floats (doubles) and hands the kernels the rounded result
    a.count = k; a.beta2 = (float)beta2; a.eps = (float)eps;
    a.w1 = (float)(1.0 - beta1); a.w2 = (float)(1.0 - beta2);
    a.bias_correction2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    GsrView dbg; memset(&dbg, 0, sizeof(dbg));
    prof_begin(stream); hipLaunchKernelGGL(gsr_adam_kernel, dim3(blocks), dim3(256), 0, stream, a);
    LAUNCH_CHECK(&dbg, stream, "adam_step");
    return 0;
}

extern "C" int gsr_mask_compact(int32_t N, const uint8_t* mask, uint32_t* idx, uint64_t* count, GsrAlloc tmp, gsr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0) return fail(-1, "N must be >= 0%s", "");
    if (!count) return fail(-1, "count is required%s", "");
    if (N == 0) { HIP_TRY(hipMemsetAsync(count, 0, 8, stream)); return 0; }
    if (!mask || !idx || !tmp.resize) return fail(-1, "mask, idx and the tmp allocator are required%s", "");
    const int nblocks = (N + 1023) / 1024;
    uint32_t* bc = (uint32_t*)tmp.resize(tmp.ctx, align_up((size_t)nblocks * 4));
    if (!bc) return fail(-4, "tmp scratch allocation failed%s", "");
    GsrView dbg; memset(&dbg, 0, sizeof(dbg));
    prof_begin(stream); hipLaunchKernelGGL(gsr_mask_count_kernel, dim3(nblocks), dim3(1024), 0, stream, N, mask, bc);
    LAUNCH_CHECK(&dbg, stream, "mask_count");
    prof_begin(stream); hipLaunchKernelGGL(gsr_mask_scan_kernel, dim3(1), dim3(1024), 0, stream, nblocks, bc, (unsigned long long*)count);
    LAUNCH_CHECK(&dbg, stream, "mask_scan");
    prof_begin(stream); hipLaunchKernelGGL(gsr_mask_write_kernel, dim3(nblocks), dim3(1024), 0, stream, N, mask, bc, idx);
    LAUNCH_CHECK(&dbg, stream, "mask_write");
    return 0;
}

extern "C" int gsr_gather_rows(int32_t count, const GsrGatherTensor* tensors, int32_t rows, const uint32_t* idx, gsr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (count < 0 || count > GSR_GATHER_MAX_TENSORS) return fail(-1, "gsr_gather_rows takes 0..24 tensors per call%s", "");
    if (rows < 0) return fail(-1, "rows must be >= 0%s", "");
    if (count == 0 || rows == 0) return 0;
    if (!tensors || !idx) return fail(-1, "tensors and idx are required%s", "");
    GatherArgs a;
    memset(&a, 0, sizeof(a));
    int wmax = 1;
    for (int i = 0; i < count; ++i) {
        if (!tensors[i].src || !tensors[i].dst || tensors[i].width < 1) return fail(-1, "src, dst and width >= 1 are required%s", "");
        a.seg[i].src = tensors[i].src; a.seg[i].dst = tensors[i].dst; a.seg[i].width = tensors[i].width;
        if (tensors[i].width > wmax) wmax = tensors[i].width;
    }
    a.count = count; a.rows = rows;
    const long long total = (long long)rows * wmax;
    const int gx = (int)fmin((double)((total + 255) / 256), 4096.0);
    GsrView dbg; memset(&dbg, 0, sizeof(dbg));
    prof_begin(stream); hipLaunchKernelGGL(gsr_gather_rows_kernel, dim3(gx, count), dim3(256), 0, stream, a, idx);
    LAUNCH_CHECK(&dbg, stream, "gather_rows");
    return 0;
}

extern "C" int gsr_concat_rows(int32_t count, const GsrConcatTensor* tensors, int32_t rows_a, int32_t rows_b, gsr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (count < 0 || count > GSR_GATHER_MAX_TENSORS) return fail(-1, "gsr_concat_rows takes 0..24 tensors per call%s", "");
    if (rows_a < 0 || rows_b < 0) return fail(-1, "rows must be >= 0%s", "");
    if (count == 0 || rows_a + rows_b == 0) return 0;
    if (!tensors) return fail(-1, "tensors are required%s", "");
    ConcatArgs a;
    memset(&a, 0, sizeof(a));
    int wmax = 1;
    for (int i = 0; i < count; ++i) {
        if (!tensors[i].dst || tensors[i].width < 1) return fail(-1, "dst and width >= 1 are required%s", "");
        a.seg[i].a = tensors[i].a; a.seg[i].b = tensors[i].b; a.seg[i].dst = tensors[i].dst; a.seg[i].width = tensors[i].width;
        if (tensors[i].width > wmax) wmax = tensors[i].width;
    }
    a.count = count; a.rows_a = rows_a; a.rows_b = rows_b;
    const long long total = ((long long)rows_a + rows_b) * wmax;
    const int gx = (int)fmin((double)((total + 255) / 256), 4096.0);
    GsrView dbg; memset(&dbg, 0, sizeof(dbg));
    prof_begin(stream); hipLaunchKernelGGL(gsr_concat_rows_kernel, dim3(gx, count), dim3(256), 0, stream, a);
    LAUNCH_CHECK(&dbg, stream, "concat_rows");
    return 0;
}

extern "C" int gsr_dist2(int32_t P, const float* points, float* out, GsrAlloc tmp, gsr_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (P < 0) return fail(-1, "P must be >= 0%s", "");
    if (P == 0) return 0;
    if (!points || !out) return fail(-1, "points and out are required%s", "");
    if (!tmp.resize) return fail(-1, "tmp allocator is required%s", "");
    // grid resolution from P alone (no host round trip): ~4 points per cell on a filled box
    int G = (int)floor(cbrt((double)P / 4.0));
    if (G < 1) G = 1;
    if (G > 128) G = 128;
    const size_t nCells = (size_t)G * G * G;
    size_t o = 0;
    const size_t o_bbox = o; o += align_up(6 * 4);
    const size_t o_grid = o; o += align_up(sizeof(KnnGrid));
    const size_t o_cnt = o; o += align_up(nCells * 4);
    const size_t o_cur = o; o += align_up(nCells * 4);
    const size_t o_off = o; o += align_up((nCells + 1) * 4);
    const size_t o_cell = o; o += align_up((size_t)P * 4);
    const size_t o_sorted = o; o += align_up((size_t)P * 16);
    char* buf = (char*)tmp.resize(tmp.ctx, o);
    if (!buf) return fail(-4, "tmp scratch allocation failed%s", "");
    uint32_t* bbox = (uint32_t*)(buf + o_bbox);
    KnnGrid* grid = (KnnGrid*)(buf + o_grid);
    uint32_t* cnt = (uint32_t*)(buf + o_cnt);
    uint32_t* cur = (uint32_t*)(buf + o_cur);
    uint32_t* off = (uint32_t*)(buf + o_off);
    uint32_t* cell_of = (uint32_t*)(buf + o_cell);
    float4* sorted = (float4*)(buf + o_sorted);
    HIP_TRY(hipMemsetAsync(bbox, 0xff, 3 * 4, stream));
    HIP_TRY(hipMemsetAsync(bbox + 3, 0, 3 * 4, stream));
    HIP_TRY(hipMemsetAsync(cnt, 0, o_off - o_cnt, stream));   // cnt | cur
    const int grid_p = (int)fmin((double)((P + 255) / 256), 2048.0);
    GsrView dbg; memset(&dbg, 0, sizeof(dbg));
    prof_begin(stream); hipLaunchKernelGGL(gsr_knn_bbox, dim3(grid_p < 256 ? grid_p : 256), dim3(256), 0, stream, P, points, bbox);
    LAUNCH_CHECK(&dbg, stream, "knn_bbox");
    prof_begin(stream); hipLaunchKernelGGL(gsr_knn_grid_setup, dim3(1), dim3(64), 0, stream, bbox, G, grid);
    LAUNCH_CHECK(&dbg, stream, "knn_grid_setup");
    prof_begin(stream); hipLaunchKernelGGL(gsr_knn_count, dim3(grid_p), dim3(256), 0, stream, P, points, grid, cell_of, cnt);
    LAUNCH_CHECK(&dbg, stream, "knn_count");
    prof_begin(stream); hipLaunchKernelGGL(gsr_knn_scan, dim3(1), dim3(1024), 0, stream, cnt, off, (int)nCells);
    LAUNCH_CHECK(&dbg, stream, "knn_scan");
    prof_begin(stream); hipLaunchKernelGGL(gsr_knn_scatter, dim3(grid_p), dim3(256), 0, stream, P, points, cell_of, off, cur, sorted);
    LAUNCH_CHECK(&dbg, stream, "knn_scatter");
    prof_begin(stream); hipLaunchKernelGGL(gsr_knn_search, dim3(grid_p), dim3(256), 0, stream, P, sorted, off, grid, out);
    LAUNCH_CHECK(&dbg, stream, "knn_search");
    return 0;
}
