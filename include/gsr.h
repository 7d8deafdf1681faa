/* gsr.h -- C ABI of libgsr.so, the MI355X (gfx950) Gaussian-splat rasterizer.
 *
 * This is the drop-in boundary for the one hot path of DreamGaussian that this repository
 * replaces: the differentiable rasterizer behind gs_renderer.Renderer.render and the
 * simple-knn distCUDA2 initialiser.  Plain pointers and sizes only -- no torch types.
 * All pointers are DEVICE pointers to contiguous fp32 / int32 data unless marked [host].
 * Every call is asynchronous on `stream` except where stated; nothing here calls
 * hipMalloc: scratch memory is obtained from the caller through GsrAlloc callbacks
 * (the torch caching allocator in the Python host side), the same ownership model as the
 * reference extension's resize lambdas.
 *
 * Reference interfaces each entry point replaces (file:line in /root/reference):
 *   gsr_forward       <- GaussianRasterizer(raster_settings)(means3D, means2D, shs, ...)
 *                        gs_renderer.py:760,800-809  (ext: _C.rasterize_gaussians)
 *   gsr_forward_views / gsr_backward_views <- the serial per-view render loop main.py:219-255 (B cameras, one launch chain)
 *   gsr_backward      <- loss.backward() through that call, main.py:273
 *                        (ext: _C.rasterize_gaussians_backward)
 *   gsr_mark_visible  <- GaussianRasterizer.markVisible (ext: _C.mark_visible; never called
 *                        by DreamGaussian, kept for surface completeness)
 *   gsr_dist2         <- simple_knn._C.distCUDA2(points), gs_renderer.py:341;
 *                        simple-knn/spatial.cu:15-26, simple_knn.cu:185-221
 * and, either side of the path:
 *   gsr_extract_fields <- GaussianModel.extract_fields, gs_renderer.py:218-294 (+ gaussian_3d_coeff :64-83)
 *   gsr_densify_stats  <- main.py:279-281 + GaussianModel.add_densification_stats, gs_renderer.py:625-627
 *   gsr_mask_compact / gsr_gather_rows / gsr_concat_rows <- prune_points, densify_and_clone / _split, cat_tensors_to_optimizer,
 *                        densification_postfix, gs_renderer.py:479-595
 *   GsrView.raw_activations <- the activations Renderer.render applies before the call,
 *                        gs_renderer.py:134-142, 762-766
 */
#ifndef GSR_H
#define GSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* hipStream_t without dragging hip headers into C callers */
typedef void* gsr_stream_t;

/* Per-view constants: the 12 fields of GaussianRasterizationSettings
 * (gs_renderer.py:745-758). Matrices are in the reference's transposed/row-vector layout:
 * x' = m[0]*x + m[4]*y + m[8]*z + m[12]  (gs_renderer.py:662-670). */
typedef struct GsrView {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    int32_t sh_degree;       /* ACTIVE degree, 0..3 */
    int32_t prefiltered;     /* accepted, ignored (reference passes False) */
    int32_t debug;           /* !=0: synchronise + check after every launch */
    const float* bg;         /* [3]  device */
    const float* viewmatrix; /* [16] device */
    const float* projmatrix; /* [16] device */
    const float* campos;     /* [3]  device */
    int32_t raw_activations; /* !=0: `opacities`, `scales`, `rotations` are DreamGaussian's RAW parameters
                              * (_opacity, _scaling, _rotation): sigmoid / exp / normalise (gs_renderer.py:134-142,
                              * 196-216) and their backward run inside the per-Gaussian kernels */
    int32_t flags;           /* GSR_VIEW_* bits below (0 = the drop-in behaviour) */
    /* Split SH input (the `cat`-free half): when shs_rest != NULL the `shs` argument of
     * gsr_forward / gsr_backward is DreamGaussian's `_features_dc` [N,1,3] and shs_rest its `_features_rest`
     * [N,K-1,3] (K still counts all coefficients): the kernels read the two tensors where they are instead of
     * the torch.cat copy of `get_features` (gs_renderer.py:209-212). In the backward dL_dshs then receives
     * the [N,1,3] part and dL_dshs_rest (required) the [N,K-1,3] part. NULL = the drop-in layout. */
    const float* shs_rest;
    float* dL_dshs_rest;
    /* Optional (gsr_forward only; NULL / 0 = off): ONE caller-owned fp32 array of grad_clear_floats elements (16-byte aligned, a
     * multiple of 4 elements) that gsr_backward's gradient outputs will be carved from. Where the backward would otherwise clear its
     * outputs under its compositing kernel (the "live Gaussians only" per-Gaussian backward of large scenes: one view, concatenated
     * SH layout, >= 64 MB of gradients) the FORWARD's per-tile compositing kernel stores the zeros instead, a slice per workgroup
     * behind its own work -- it leaves HBM idle for longer. GsrStats.bwd_prepared comes back 2 when that happened; the backward is
     * then handed the SAME view struct and outputs that lie inside the array, and clears nothing. */
    float* grad_clear;
    int64_t grad_clear_floats;
} GsrView;

/* GsrView.flags */
#define GSR_VIEW_VIEWMATRIX_T 1  /* viewmatrix points at the TRANSPOSE of the layout above (x' = m[0]*x + m[1]*y + m[2]*z + m[3]): what the
                                  * storage of the reference's `world_view_transform` holds -- a `.transpose(0, 1)` VIEW of the row-major
                                  * w2c (gs_renderer.py:662-664) -- so that binding needs no `.contiguous()` copy kernel per render */
#define GSR_VIEW_PROJMATRIX_T 2  /* the same for projmatrix */
#define GSR_VIEW_NO_BACKWARD 4   /* no gsr_backward will follow this forward (inference): the backward's accumulators are neither allocated nor
                                  * cleared and the serial-walk forward keeps neither checkpoints nor quad masks (118 MB of stores and 0.4 GB of
                                  * scratch at 1M Gaussians / 800^2). GsrStats.bwd_prepared = -1; gsr_backward on such a state returns -1 */
#define GSR_VIEW_DETERMINISTIC 16 /* gsr_backward only (ABI 6): the compositing backward adds its per-Gaussian sums in 64-bit FIXED POINT (integer atomics: the
                                  * order of the additions no longer matters) instead of with float atomics -- gradients bit-identical from run to run, at
                                  * the price of two small extra kernels, N x 80 bytes of `tmp` scratch and integer instead of float atomics. Same values
                                  * as the default up to fp32 rounding of the sums (tests/test_parity_gpu.py::test_deterministic_backward_*) */
#define GSR_VIEW_ASYNC_STATS 8   /* OPT-IN (ABI 6): a gsr_forward that could size its list scratch from earlier calls of the same shape returns as soon as
                                  * its kernels are enqueued -- the host does not wait for the instance counters (no synchronisation in
                                  * the steady state). GsrStats then comes back with pending != 0, bin_capacity / seg_shift / bwd_prepared final (all
                                  * gsr_backward needs) and the four counts = -1 until gsr_forward_complete(stats) -- or the thread's next gsr_forward --
                                  * has collected them. The capacity is taken from the LARGEST of the shape's recent calls + 50 % (instead of the last
                                  * call + 25 %). Should the lists still not fit, nobody is left to repeat the tail: the images of that call are filled
                                  * with NaN, its backward yields zero gradients, and the thread's next gsr_forward (or gsr_forward_complete) returns -6
                                  * without doing anything else. First call of a shape, "speculate" = 0 or stats == NULL: the blocking path as before. */

/* Scratch allocator: resize(ctx, bytes) must return a device pointer, 256-byte aligned, to
 * at least `bytes` bytes that stay alive until the matching backward has run. */
typedef void* (*GsrResizeFn)(void* ctx, size_t bytes);
typedef struct GsrAlloc {
    void* ctx;
    GsrResizeFn resize;
} GsrAlloc;

/* Statistics of the last forward on this thread (scene statistics V and M of SURVEY 8(d)). */
typedef struct GsrStats {
    int64_t num_instances;      /* M_emit: (tile,Gaussian) pairs actually binned */
    int64_t num_instances_ref;  /* M: pairs under the reference's 3-sigma rect rule */
    int64_t num_visible;        /* V: Gaussians with radii > 0 */
    int64_t max_tile_count;     /* longest per-tile list */
    int64_t bin_capacity;       /* instances the `bin` scratch was laid out for (>= num_instances): gsr_forward sizes it
                                 * from the previous call (+25 %) so that nothing waits for the host; the backward needs it */
    int64_t seg_shift;          /* log2 of the depth-segment length (6..8) the forward cut the tile lists with; the backward
                                 * walks the same segments */
    int64_t bwd_prepared;       /* 1: the forward has cleared the backward's per-Gaussian accumulators inside `geom` (every workgroup
                                 * of its per-tile compositing kernel stores a slice of zeros behind its own work) -- gsr_backward
                                 * then neither allocates `tmp` nor clears anything. One-shot: a caller that runs a SECOND backward from the same forward state must pass
                                 * 0 (the first one has accumulated into them); 0 also without GsrStats. 2: as 1, and the forward has also cleared
                                 * GsrView.grad_clear (the array the backward's outputs are carved from; same one-shot rule). -1: the
                                 * forward ran with GSR_VIEW_NO_BACKWARD and left no state for a backward */
    int64_t speculated;         /* 1: the list scratch and the choice of sort kernel came from the thread's earlier calls of the same (N, H, W, views) -- an
                                 * 8-entry table, least recently used shape replaced -- and binning / sort / compositing ran without waiting for the host;
                                 * 0: first call of the shape, a prediction that turned out too small (tail repeated), or "speculate" = 0 */
    int64_t pending;            /* != 0: an asynchronous forward (GSR_VIEW_ASYNC_STATS) whose counts have not been collected: gsr_forward_complete */
} GsrStats;

/* Collects the counts of the calling thread's asynchronous forward `stats` came from (blocking until its counters have arrived; a no-op
 * when stats->pending == 0). Must run on the thread that called gsr_forward, before that thread's forward after next. Returns 0, or -6
 * when the forward's lists did not fit its speculative capacity (see GSR_VIEW_ASYNC_STATS). */
int gsr_forward_complete(GsrStats* stats);

/* Bumped whenever a struct of this header changes size or meaning (GsrStats grew in 3 and 4, GsrView.reserved became flags in 4, GsrView grew in 5; 6: GsrStats grew by `speculated` / `pending`, gsr_backward takes NULL incoming gradients, gsr_forward_complete). A caller built against another
 * value must not call the library: dreamgaussian_amd/_lib.py checks gsr_abi_version() at load. */
#define GSR_ABI_VERSION 6
int gsr_abi_version(void);

/* TEST HOOK -- not part of the drop-in surface. Forces one of the choices the library otherwise makes from the problem shape
 * (process-wide; value -1 = the library decides again). Nothing is read from the process environment: "fwd_mode" and "seg_shift"
 * decide where the per-pixel sums are cut, i.e. the rounding of the results.
 *   "fwd_mode"     1 = serial walk (gsr_render_fwd_serial), 2 = depth-segmented forward (K5a/b/c), 3 = serial walk with a tester
 *                  and a blender wave per 8x8 block (gsr_render_fwd_pair: the library's choice for ONE view of 1 024 .. 2 047 tiles, or of
 *                  any size when its launch also clears GsrView.grad_clear; the same bits as 1)
 *   "seg_shift"    6..8: log2 of the depth-segment length
 *   "fwd_lists"    1 = 8x8 block lists, 2 = quad lists in the forward compositing
 *   "fwd_hints"    1 = off, 2 = every segment behind a tile's first skipped (the chaining kernel walks them all)
 *   "speculate"    0 = gsr_forward waits for the instance count before binning
 *   "hist_max" "k1_grid" "fwd_grid" "k6_grid"   launch geometry (A/B measurements, multi-round paths; "k1_grid" is the grid of
 *                  K1 AND of the scatter kernel, which continues K1's per-workgroup list ranges)
 *   "fwd_lds_kb"   KiB of unused dynamic LDS requested by the serial walk = how many of its workgroups share a CU (0 = none)
 *   "k6_compact"   how the per-Gaussian backward of a single view runs: 0 = it streams every Gaussian, lane = Gaussian
 *                  (gsr_preprocess_bwd); 1 = it visits only those the compositing kernel marked as carrying a gradient
 *                  (gsr_preprocess_bwd_compact, every gradient array cleared under the compositing kernel) whenever the layout
 *                  allows; the library picks 1 from 64 MB of gradient arrays on
 *   "scan_fold"    0 = the scan of the tile counts (K2) always runs as a kernel of its own, 1 = inside the scatter's launch whenever
 *                  the forward is speculative and composites from the tile order (the library: from 2M predicted instances on)
 *   "grad_clear"   0 = the forward ignores GsrView.grad_clear (the backward's compositing kernel clears its outputs, as in ABI 4)
 *   "bwd_grid"     caps the grid of the backward's compositing kernel (TIMING experiments only: work beyond the cap is dropped)
 * Returns 0, or -1 for an unknown name. dreamgaussian_amd/_testing.py wraps it. */
int gsr_testing_override(const char* name, int32_t value);


/* Forward.
 *   N                number of Gaussians; K = shs.shape[1] (max coefficients per Gaussian)
 *   means3D [N,3]    shs [N,K,3] or NULL    colors_precomp [N,3] or NULL (exactly one)
 *   opacities [N]    scales [N,3] + rotations [N,4] (r,x,y,z) or cov3D_precomp [N,6]
 *   out_color [3,H,W] out_depth [H,W] out_alpha [H,W] radii [N] int32
 *   geom/bin/img     scratch; the three buffers must be kept and handed to gsr_backward
 *   stats            [host] optional
 * The host returns once the instance counters of THIS call have arrived (GsrStats is exact), but nothing on the GPU waits for
 * the host: the list scratch is sized from the previous call of the thread on the same (N, H, W) (+25 %), binning / sort /
 * compositing are enqueued before the wait, and a prediction that turns out too small (more instances than the scratch was laid
 * out for; the per-tile sort takes lists of any length) is detected on the device by every kernel that touches the lists and the tail repeated
 * (first call of a shape, or test hook "speculate" = 0: counters first, then the tail, like the reference ext's blocking read of num_rendered).
 * The result does not depend on the prediction: the depth-segment length (GsrStats.seg_shift) follows N and the image size only.
 * Returns 0, or a negative code with gsr_last_error() set. N==0 renders the background. */
int gsr_forward(const GsrView* view, int32_t N, int32_t K,
                const float* means3D, const float* shs, const float* colors_precomp,
                const float* opacities, const float* scales, const float* rotations,
                const float* cov3D_precomp,
                float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                GsrAlloc geom, GsrAlloc bin, GsrAlloc img,
                GsrStats* stats, gsr_stream_t stream);

/* Backward. Same inputs as the forward plus the incoming gradients
 *   dL_dcolor [3,H,W]  dL_ddepth [H,W]  dL_dalpha [H,W]     (each may be NULL = zeros, ABI 6: DreamGaussian's stage 1 never differentiates
 *   depth, main.py:198-275 -- no zero image has to be built for it)
 * and the three scratch buffers of the matching forward, plus (optional, [host]) the GsrStats
 * that forward returned -- without it the backward reads the two counters it needs back from
 * the device (one blocking copy). Outputs (dense, exact zeros for
 * culled Gaussians; a NULL output is skipped where that is meaningful):
 *   dL_dmeans3D [N,3]  dL_dmeans2D [N,3] (x,y in NDC units = pixel grad * 0.5*(W,H); z=0)
 *   dL_dshs [N,K,3] | dL_dcolors [N,3]   dL_dopacities [N]
 *   dL_dscales [N,3] + dL_drotations [N,4] | dL_dcov3D [N,6]
 *   tmp               scratch for the per-Gaussian screen-space gradient accumulators */
int gsr_backward(const GsrView* view, int32_t N, int32_t K,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* opacities, const float* scales, const float* rotations,
                 const float* cov3D_precomp, const int32_t* radii,
                 const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                 const void* geom, const void* bin, const void* img,
                 const GsrStats* fwd_stats,
                 float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors,
                 float* dL_dopacities, float* dL_dscales, float* dL_drotations,
                 float* dL_dcov3D, GsrAlloc tmp, gsr_stream_t stream);

/* B cameras of the same Gaussians in ONE launch chain (replaces the serial per-view render loop main.py:219-255,
 * `for _ in range(batch_size): out = renderer.render(cam)`): every kernel of gsr_forward / gsr_backward is launched
 * once over the B views (per-Gaussian kernels: grid.y = view; per-tile kernels: B * tiles tiles, heaviest first across
 * views), one host round trip for the whole batch, one list array. `views` [host] = B structs, 1 <= B <= GSR_MAX_VIEWS,
 * which must agree in image size, sh_degree, raw_activations and shs_rest / dL_dshs_rest (cameras, tan-fov, bg and
 * scale_modifier may differ).
 *   out_color [B,3,H,W]  out_depth [B,H,W]  out_alpha [B,H,W]  radii [B,N]       (contiguous, view-major)
 *   dL_dcolor [B,3,H,W]  dL_ddepth [B,H,W]  dL_dalpha [B,H,W]  dL_dmeans2D [B,N,3]
 *   dL_dmeans3D, dL_dshs, ... [N,...]: the SUM over the views, accumulated in the order autograd adds the gradients
 *   of B separate calls (last view first), without a [B,N,...] intermediate
 *   stats: totals over the B views. Images and per-view outputs are bit-identical to B gsr_forward calls. */
#define GSR_MAX_VIEWS 16
int gsr_forward_views(const GsrView* views, int32_t B, int32_t N, int32_t K,
                      const float* means3D, const float* shs, const float* colors_precomp,
                      const float* opacities, const float* scales, const float* rotations,
                      const float* cov3D_precomp,
                      float* out_color, float* out_depth, float* out_alpha, int32_t* radii,
                      GsrAlloc geom, GsrAlloc bin, GsrAlloc img,
                      GsrStats* stats, gsr_stream_t stream);
int gsr_backward_views(const GsrView* views, int32_t B, int32_t N, int32_t K,
                       const float* means3D, const float* shs, const float* colors_precomp,
                       const float* opacities, const float* scales, const float* rotations,
                       const float* cov3D_precomp, const int32_t* radii,
                       const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                       const void* geom, const void* bin, const void* img,
                       const GsrStats* fwd_stats,
                       float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors,
                       float* dL_dopacities, float* dL_dscales, float* dL_drotations,
                       float* dL_dcov3D, GsrAlloc tmp, gsr_stream_t stream);

/* Frustum test only: visible[i] = (view-space z of means3D[i] > 0.2). */
int gsr_mark_visible(const GsrView* view, int32_t N, const float* means3D,
                     uint8_t* visible, gsr_stream_t stream);

/* distCUDA2: out[i] = mean of the squared distances from point i to its 3 nearest
 * neighbours (self excluded by index; fewer than 3 neighbours count as FLT_MAX each,
 * as in simple_knn.cu:142-182). points [P,3], out [P]. No host synchronisation. */
int gsr_dist2(int32_t P, const float* points, float* out, GsrAlloc tmp, gsr_stream_t stream);

/* Density grid of the Gaussians: GaussianModel.extract_fields (gs_renderer.py:218-294) with
 * gaussian_3d_coeff (gs_renderer.py:64-83).
 *   xyz [N,3], opacity [N] (activated), scaling [N,3] (activated), rotation_raw [N,4] (the raw
 *   `_rotation`; it is normalised inside, gs_renderer.py:86-88) -- what the reference reads.
 *   axis [resolution]     the grid coordinates: torch.linspace(-1, 1, resolution)   (:251)
 *   split_size            samples per chunk and axis: resolution // num_blocks      (:224)
 *   num_chunks            chunks per axis: ceil(resolution / split_size)  (<= 255)
 *   box_lo/hi [num_chunks] chunk bounds grown by block_size * relax_ratio           (:259-262)
 *   occ [resolution^3]    out, x-major (occ[ix][iy][iz], :286-288); every element is written
 *   norm_out [4]          out: center.xyz and the extent max(mx - mn) of the kept means; the
 *                         reference's `self.center`, and `self.scale` = 1.8 / extent (:237-240)
 * All pointers are device pointers. Gaussians with opacity <= 0.005 are ignored (:230); at least
 * one must remain (the reference fails on an empty reduction there; the caller checks).
 * No host synchronisation. */
int gsr_extract_fields(int32_t N, const float* xyz, const float* opacity, const float* scaling,
                       const float* rotation_raw, int32_t resolution, int32_t split_size,
                       int32_t num_chunks, const float* axis, const float* box_lo, const float* box_hi,
                       float* occ, float* norm_out, GsrAlloc tmp, gsr_stream_t stream);

/* Per-step densification bookkeeping (main.py:279-281 + GaussianModel.add_densification_stats,
 * gs_renderer.py:625-627), in place, for every Gaussian with radii[i] > 0:
 *   max_radii2D[i] = max(max_radii2D[i], radii[i]);
 *   xyz_gradient_accum[i] += |grad_means2D[i, 0:2]|;   denom[i] += 1.
 * grad_means2D [N,3] is the gradient the backward left in the means2D holder; radii [N] int32 the
 * forward's output; the three accumulators are [N] float32 (the reference's [N,1] / [N]).
 * One launch, no host synchronisation. */
int gsr_densify_stats(int32_t N, const float* grad_means2D, const int32_t* radii,
                      float* xyz_gradient_accum, float* denom, float* max_radii2D, gsr_stream_t stream);

/* Optimiser step of GaussianModel.training_setup's Adam (gs_renderer.py:356-374; torch.optim.Adam, amsgrad off,
 * no weight decay) for up to 8 tensors in ONE launch, in torch's arithmetic order:
 *   exp_avg += (1 - beta1) (grad - exp_avg);  exp_avg_sq = exp_avg_sq beta2 + (1 - beta2) grad grad;
 *   param -= lr / (1 - beta1^step) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - beta2^step) + eps).
 * `step` counts from 1 (torch increments state["step"] before the update). All pointers are device pointers to
 * contiguous fp32; the state tensors are torch's own (`optimizer.state[p]["exp_avg"]`, ...), so the reference's
 * optimiser-state surgery in densify_and_prune (gs_renderer.py:464-545) keeps working on them. */
typedef struct GsrAdamTensor {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    int64_t n;      /* elements */
    double lr;      /* the group's learning rate (a Python float) */
} GsrAdamTensor;
int gsr_adam_step(int32_t count, const GsrAdamTensor* tensors /* [host] */, int32_t step, double beta1, double beta2, double eps,
                  gsr_stream_t stream);

/* prune_points / densify selections (gs_renderer.py:479-609) without one nonzero() per tensor:
 * gsr_mask_compact: stable compaction of a byte mask [N] (0 / non-0): idx[j] = index of the j-th set element
 *   (idx must hold N entries), count[0] (device, 8 bytes) = number of set elements. No host synchronisation; the
 *   caller reads `count` once to size the new tensors.
 * gsr_gather_rows: dst_t[j, :] = src_t[idx[j], :], j < rows, for up to 24 fp32 tensors of row width width_t, in one
 *   launch (the six parameters, their two Adam moments each and the three densification accumulators). */
int gsr_mask_compact(int32_t N, const uint8_t* mask, uint32_t* idx, uint64_t* count, GsrAlloc tmp, gsr_stream_t stream);
typedef struct GsrGatherTensor { const float* src; float* dst; int32_t width; int32_t reserved; } GsrGatherTensor;
int gsr_gather_rows(int32_t count, const GsrGatherTensor* tensors /* [host] */, int32_t rows, const uint32_t* idx,
                    gsr_stream_t stream);

/* densification_postfix / cat_tensors_to_optimizer (gs_renderer.py:513-552) in one launch: dst_t = [a_t ; b_t] -- rows_a rows
 * of a_t followed by rows_b rows of b_t -- for up to 24 fp32 tensors of row width width_t. A NULL source stands for zeros: the
 * six parameters are extended by the new Gaussians (b = the new rows), their Adam moments by zeros (b = NULL), the three
 * densification accumulators are reset (a = b = NULL). dst must hold rows_a + rows_b rows and may not alias a source. */
typedef struct GsrConcatTensor { const float* a; const float* b; float* dst; int32_t width; int32_t reserved; } GsrConcatTensor;
int gsr_concat_rows(int32_t count, const GsrConcatTensor* tensors /* [host] */, int32_t rows_a, int32_t rows_b, gsr_stream_t stream);

/* Bytes of scratch the forward will request for geom / img (bin is data dependent). */
size_t gsr_geom_bytes(int32_t N, int32_t image_height, int32_t image_width);
size_t gsr_img_bytes(int32_t image_height, int32_t image_width);

/* Per-kernel timing, a measurement aid for bench.py: while enabled (process
 * wide) every kernel launched by gsr_forward / gsr_backward / gsr_dist2 is bracketed by a
 * hipEvent pair recorded on the caller's stream. gsr_profile_read waits for the recorded
 * events, folds them into per-kernel totals and copies up to `cap` rows out:
 * names[i] (static strings), total_ms[i], launches[i]. Returns the number of rows.
 * gsr_profile_enable(0) stops recording; gsr_profile_reset() clears the totals. */
int gsr_profile_enable(int on);
int gsr_profile_read(int cap, const char** names, float* total_ms, int* launches);
int gsr_profile_reset(void);

const char* gsr_last_error(void);
const char* gsr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H */
