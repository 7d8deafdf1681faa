// gsr_torch.cpp -- the torch binding of libgsr.so's rasterizer entry points as a C++ autograd function (round 6).
//
// The reference binds its rasterizer the same way: `diff_gaussian_rasterization` is a pybind11 module whose `_C.rasterize_gaussians` /
// `_C.rasterize_gaussians_backward` a thin Python autograd.Function calls (gs_renderer.py:10-13, 800-809). The shipped ctypes binding
// (dreamgaussian_amd/rasterizer.py) does everything around the two C-ABI calls in Python -- tensor allocations, three scratch callbacks
// that re-enter Python, the view struct, the gradient views -- and at DreamGaussian's own sizes (5k-11k Gaussians, 128^2-512^2, two
// renders per iteration) that Python IS the step: 97 + 67 us of host per forward + backward around 0.09 ms of kernels
// (profiles/r06_host_phases_5k.txt). This file is the same host logic in C++: same checks, same layouts, same calls into the C ABI
// (include/gsr.h stays the boundary; nothing here computes). rasterizer.py uses it for the drop-in call when the module has been built
// (dreamgaussian_amd/build.py) and falls back to its own Python otherwise; tests/test_binding_gpu.py holds the two to the same bits.
//
// The library is NOT linked: rasterizer.py hands over the path it loaded (GSR_LIB= A/B builds included) and the entry points are
// looked up in that same library instance (dlopen of an already loaded file returns it).
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <c10/core/DeviceGuard.h>
#include <dlfcn.h>

#include "../../include/gsr.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

struct Api {
    decltype(&gsr_forward) forward = nullptr;
    decltype(&gsr_backward) backward = nullptr;
    decltype(&gsr_forward_complete) forward_complete = nullptr;
    decltype(&gsr_last_error) last_error = nullptr;
    decltype(&gsr_abi_version) abi_version = nullptr;
} g_api;

bool g_pass_fwd_stats = true;          // (tests: fwd_stats = NULL exercises the ABI's other documented mode)
thread_local GsrStats g_last_stats = {};
thread_local int64_t g_last_dims[4] = {0, 0, 0, 0};    // N, H, W, K of the thread's last forward

void check(int rc, const char* what) {
    if (rc == 0) return;
    const char* msg = g_api.last_error ? g_api.last_error() : "";
    TORCH_CHECK(false, what, " failed (code ", rc, "): ", msg);
}

struct Scratch {
    Tensor t;
    c10::Device dev;
    explicit Scratch(c10::Device d) : dev(d) {}
    static void* resize(void* ctx, size_t bytes) {
        auto* s = static_cast<Scratch*>(ctx);
        try {
            s->t = at::empty({(int64_t)(bytes ? bytes : 1)}, at::TensorOptions().dtype(at::kByte).device(s->dev));
            return s->t.data_ptr();
        } catch (...) { return nullptr; }              // surfaces as "scratch allocation failed" on the C side
    }
    GsrAlloc alloc() { return GsrAlloc{this, &Scratch::resize}; }
};

// None / empty -> undefined; else fp32 contiguous on `dev` (as it is when it already is)
Tensor f32c(const Tensor& t, const c10::Device& dev) {
    if (!t.defined() || t.numel() == 0) return Tensor();
    TORCH_CHECK(t.device() == dev, "all rasterizer inputs must be on ", dev, ", got ", t.device());
    if (t.scalar_type() == at::kFloat && t.is_contiguous()) return t;
    return t.detach().to(at::kFloat).contiguous();
}
const float* fptr(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
float* fptr_mut(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

struct Cam { Tensor bg, view, proj, campos; int flags = 0; };
// a camera constant as the kernels read it: no copy when it already is fp32 on the device and contiguous -- or the reference's
// `.transpose(0, 1)` VIEW of a row-major 4x4, whose storage is passed as it is with GSR_VIEW_*_T (gs_renderer.py:662-664)
Tensor cam_tensor(const Tensor& x, const c10::Device& dev, int tbit, int& flags) {
    const bool ok = x.defined() && x.scalar_type() == at::kFloat && x.device() == dev;
    if (ok && x.is_contiguous()) return x;
    if (ok && tbit && x.dim() == 2 && x.size(0) == 4 && x.size(1) == 4 && x.stride(0) == 1 && x.stride(1) == 4) { flags |= tbit; return x; }
    return x.to(at::TensorOptions().dtype(at::kFloat).device(dev)).contiguous().reshape({-1});
}

// floats per Gaussian of each gradient, in the order they are carved (rasterizer._gradient_widths): parameter gradients first, the
// per-view means2D gradient last
void gradient_widths(int64_t K, int64_t k_rest, bool has_sh, bool has_col, bool has_sr, bool has_cov, int64_t w[9]) {
    const int64_t k_sh = K - k_rest;
    w[0] = 3; w[1] = 1; w[2] = has_sh ? 3 * k_sh : 0; w[3] = has_col ? 3 : 0; w[4] = has_sr ? 3 : 0; w[5] = has_sr ? 4 : 0;
    w[6] = has_cov ? 6 : 0; w[7] = 3 * k_rest; w[8] = 3;
}
// ONE allocation, gradient i = N x w[i] floats at a 64-element (256-byte) boundary (rasterizer.carve_gradients)
Tensor carve(int64_t N, const int64_t w[9], int64_t offs[9], const c10::Device& dev) {
    int64_t total = 0;
    for (int i = 0; i < 9; ++i) { offs[i] = total; total += (N * w[i] + 63) & ~int64_t(63); }
    auto opt = at::TensorOptions().dtype(at::kFloat).device(dev);
    return N > 0 ? at::empty({total > 0 ? total : 1}, opt) : at::zeros({total > 0 ? total : 1}, opt);
}

GsrView make_view(int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier, int64_t sh_degree, bool prefiltered,
                  bool debug, const Cam& cam, bool raw, int flags) {
    GsrView v;
    memset(&v, 0, sizeof(v));
    v.image_height = (int32_t)H; v.image_width = (int32_t)W; v.tanfovx = (float)tanfovx; v.tanfovy = (float)tanfovy;
    v.scale_modifier = (float)scale_modifier; v.sh_degree = (int32_t)sh_degree; v.prefiltered = prefiltered ? 1 : 0; v.debug = debug ? 1 : 0;
    v.bg = cam.bg.data_ptr<float>(); v.viewmatrix = cam.view.data_ptr<float>(); v.projmatrix = cam.proj.data_ptr<float>();
    v.campos = cam.campos.data_ptr<float>();
    v.raw_activations = raw ? 1 : 0; v.flags = flags | cam.flags;
    return v;
}

class Rasterize : public torch::autograd::Function<Rasterize> {
public:
    // tensors first (their positions are the gradient positions), then the settings' scalars
    static variable_list forward(AutogradContext* ctx, const Tensor& means3D, const Tensor& means2D, const Tensor& sh, const Tensor& colors_precomp,
                                 const Tensor& opacities, const Tensor& scales, const Tensor& rotations, const Tensor& cov3Ds_precomp,
                                 const Tensor& sh_rest, const Tensor& bg, const Tensor& viewmatrix, const Tensor& projmatrix, const Tensor& campos,
                                 int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier, int64_t sh_degree,
                                 bool prefiltered, bool debug, bool raw, int64_t extra_flags, bool grad_mode, bool force_grad_block) {
        TORCH_CHECK(means3D.is_cuda(), "dreamgaussian_amd rasterizer runs on an MI355X (torch device 'cuda' on ROCm) only; got a tensor on '",
                    means3D.device(), "'. There is no CPU fallback.");
        TORCH_CHECK(g_api.forward, "gsr_torch.init(path of libgsr.so) has not been called");
        const c10::Device dev = means3D.device();
        const int64_t N = means3D.size(0);
        TORCH_CHECK(N == 0 || (means3D.dim() == 2 && means3D.size(1) == 3), "means3D must have dimensions (num_points, 3)");
        const Tensor m3 = f32c(means3D, dev), shc = f32c(sh, dev), col = f32c(colors_precomp, dev), op = f32c(opacities, dev),
                     sc = f32c(scales, dev), rot = f32c(rotations, dev), cov = f32c(cov3Ds_precomp, dev), rest = f32c(sh_rest, dev);
        int64_t K = shc.defined() ? shc.size(1) : 0;
        TORCH_CHECK(!shc.defined() || (shc.dim() == 3 && shc.size(0) == N && shc.size(2) == 3), "shs must have dimensions (num_points, num_coeffs, 3)");
        if (rest.defined()) {
            TORCH_CHECK(shc.defined() && K == 1 && rest.dim() == 3 && rest.size(0) == N && rest.size(2) == 3,
                        "split SH input needs features_dc (num_points, 1, 3) and features_rest (num_points, K-1, 3)");
            K = 1 + rest.size(1);
        }
        auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
        Tensor color = at::empty({3, H, W}, fopt), depth = at::empty({1, H, W}, fopt), alpha = at::empty({1, H, W}, fopt);
        Tensor radii = at::empty({N}, at::TensorOptions().dtype(at::kInt).device(dev));
        bool any_grad = false;
        for (const Tensor* t : {&means3D, &means2D, &sh, &colors_precomp, &opacities, &scales, &rotations, &cov3Ds_precomp, &sh_rest})
            any_grad = any_grad || (t->defined() && t->requires_grad());
        const bool want_bwd = grad_mode && any_grad;
        Cam cam;
        cam.bg = cam_tensor(bg, dev, 0, cam.flags); cam.view = cam_tensor(viewmatrix, dev, GSR_VIEW_VIEWMATRIX_T, cam.flags);
        cam.proj = cam_tensor(projmatrix, dev, GSR_VIEW_PROJMATRIX_T, cam.flags); cam.campos = cam_tensor(campos, dev, 0, cam.flags);
        TORCH_CHECK(cam.bg.numel() == 3 && cam.view.numel() == 16 && cam.proj.numel() == 16 && cam.campos.numel() == 3,
                    "bg/campos must have 3 elements and viewmatrix/projmatrix 16");
        const int flags = (want_bwd ? 0 : GSR_VIEW_NO_BACKWARD) | (int)extra_flags;
        GsrView view = make_view(H, W, tanfovx, tanfovy, scale_modifier, sh_degree, prefiltered, debug, cam, raw, flags);
        if (rest.defined()) view.shs_rest = rest.data_ptr<float>();
        // the backward's gradient block, allocated HERE where the library uses it (its rule, gsr_api.hip k6_compact_for: one view,
        // concatenated SH layout, >= 64 MB of gradients): GsrView.grad_clear
        const bool has_sh = shc.defined(), has_col = col.defined(), has_sr = sc.defined(), has_cov = cov.defined();
        const int64_t k_rest = rest.defined() ? rest.size(1) : 0;
        Tensor grad_flat;
        int64_t w[9], offs[9] = {0};
        gradient_widths(K, k_rest, has_sh, has_col, has_sr, has_cov, w);
        if (want_bwd && N > 0 && !rest.defined() && (N * (3 * (has_sh ? K : 1) + 14) * 4 >= (int64_t(64) << 20) || force_grad_block)) {
            grad_flat = carve(N, w, offs, dev);
            view.grad_clear = grad_flat.data_ptr<float>();
            view.grad_clear_floats = grad_flat.numel();
        }
        Scratch geom(dev), binb(dev), img(dev);
        GsrStats stats;
        memset(&stats, 0, sizeof(stats));
        int rc;
        {
            c10::OptionalDeviceGuard guard(dev);
            auto stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
            rc = g_api.forward(&view, (int32_t)N, (int32_t)K, fptr(m3), fptr(shc), fptr(col), fptr(op), fptr(sc), fptr(rot), fptr(cov),
                               color.data_ptr<float>(), depth.data_ptr<float>(), alpha.data_ptr<float>(), N > 0 ? radii.data_ptr<int32_t>() : nullptr,
                               geom.alloc(), binb.alloc(), img.alloc(), &stats, (gsr_stream_t)stream);
        }
        check(rc, "gsr_forward");
        g_last_stats = stats;
        g_last_dims[0] = N; g_last_dims[1] = H; g_last_dims[2] = W; g_last_dims[3] = K;
        const Tensor empty = at::empty({0}, fopt);
        auto keep = [&](const Tensor& t) { return t.defined() ? t : empty; };
        // (the camera constants travel as raw pointers: saved too, so that an in-place edit before the backward raises)
        ctx->save_for_backward({keep(m3), keep(shc), keep(col), keep(op), keep(sc), keep(rot), keep(cov), keep(rest), radii,
                                keep(geom.t), keep(binb.t), keep(img.t), cam.bg, cam.view, cam.proj, cam.campos, keep(grad_flat)});
        auto& sd = ctx->saved_data;
        sd["N"] = N; sd["K"] = K; sd["H"] = H; sd["W"] = W; sd["k_rest"] = k_rest;
        sd["present"] = (int64_t)((has_sh ? 1 : 0) | (has_col ? 2 : 0) | (has_sr ? 4 : 0) | (has_cov ? 8 : 0) | (grad_flat.defined() ? 16 : 0));
        sd["tanfovx"] = tanfovx; sd["tanfovy"] = tanfovy; sd["scale_modifier"] = scale_modifier; sd["sh_degree"] = sh_degree;
        sd["prefiltered"] = prefiltered; sd["debug"] = debug; sd["raw"] = raw; sd["flags"] = (int64_t)flags; sd["cam_flags"] = (int64_t)cam.flags;
        const int64_t* st = reinterpret_cast<const int64_t*>(&stats);
        static_assert(sizeof(GsrStats) == 9 * sizeof(int64_t), "GsrStats: nine int64 (ABI 6)");
        std::vector<int64_t> stv(st, st + 9);
        sd["stats"] = stv;
        sd["m2_is_holder"] = means2D.defined() && means2D.dim() == 2 && means2D.size(0) == N && means2D.size(1) == 3;
        // shapes of the gradients the caller's tensors want
        auto shape = [](const Tensor& t) { return t.defined() && t.numel() > 0 ? t.sizes().vec() : std::vector<int64_t>(); };
        sd["s_m3"] = means3D.sizes().vec(); sd["s_sh"] = shape(sh); sd["s_col"] = shape(colors_precomp); sd["s_op"] = opacities.sizes().vec();
        sd["s_sc"] = shape(scales); sd["s_rot"] = shape(rotations); sd["s_cov"] = shape(cov3Ds_precomp); sd["s_rest"] = shape(sh_rest);
        ctx->mark_non_differentiable({radii});
        ctx->set_materialize_grads(false);       // no zero tensors for outputs the loss does not use (and no int32 "gradient" of radii)
        return {color, radii, depth, alpha};
    }

    static variable_list backward(AutogradContext* ctx, variable_list grad_outputs) {
        auto saved = ctx->get_saved_variables();
        const Tensor &m3 = saved[0], &shc = saved[1], &col = saved[2], &op = saved[3], &sc = saved[4], &rot = saved[5], &cov = saved[6],
                     &rest = saved[7], &radii = saved[8], &geom = saved[9], &binb = saved[10], &img = saved[11];
        Cam cam;
        cam.bg = saved[12]; cam.view = saved[13]; cam.proj = saved[14]; cam.campos = saved[15];
        auto& sd = ctx->saved_data;
        const int64_t N = sd["N"].toInt(), K = sd["K"].toInt(), H = sd["H"].toInt(), W = sd["W"].toInt(), k_rest = sd["k_rest"].toInt();
        const int64_t present = sd["present"].toInt();
        const bool has_sh = present & 1, has_col = present & 2, has_sr = present & 4, has_cov = present & 8;
        cam.flags = 0;
        const c10::Device dev = radii.device();
        auto grad_in = [&](size_t i) -> Tensor {              // None -> NULL = zeros (ABI 6)
            const Tensor& g = grad_outputs[i];
            if (!g.defined()) return Tensor();
            return (g.scalar_type() == at::kFloat && g.is_contiguous()) ? g : g.to(at::kFloat).contiguous();
        };
        const Tensor gc = grad_in(0), gd = grad_in(2), ga = grad_in(3);
        int64_t w[9], offs[9] = {0};
        gradient_widths(K, k_rest, has_sh, has_col, has_sr, has_cov, w);
        auto stv = sd["stats"].toIntVector();
        GsrStats stats;
        memcpy(&stats, stv.data(), sizeof(stats));
        Tensor flat;
        if (present & 16) {                                   // the forward's allocation: taken by the first backward
            flat = saved[16];
            int64_t total = 0;
            for (int i = 0; i < 9; ++i) { offs[i] = total; total += (N * w[i] + 63) & ~int64_t(63); }
            sd["present"] = present & ~int64_t(16);
        } else {                                              // the usual case, or a second backward of this forward (retain_graph)
            flat = carve(N, w, offs, dev);
            if (stats.bwd_prepared == 2) stats.bwd_prepared = 1;
        }
        auto part2 = [&](int i, int64_t a, int64_t b) { return flat.as_strided({a, b}, {b, 1}, offs[i]); };
        auto part3 = [&](int i, int64_t a, int64_t b, int64_t c) { return flat.as_strided({a, b, c}, {b * c, c, 1}, offs[i]); };
        const int64_t k_sh = K - k_rest;
        Tensor d_m3 = part2(0, N, 3), d_op = part2(1, N, 1), d_m2 = part2(8, N, 3);
        Tensor d_sh = has_sh ? part3(2, N, k_sh, 3) : Tensor(), d_rest = k_rest ? part3(7, N, k_rest, 3) : Tensor();
        Tensor d_col = has_col ? part2(3, N, 3) : Tensor(), d_sc = has_sr ? part2(4, N, 3) : Tensor(), d_rot = has_sr ? part2(5, N, 4) : Tensor();
        Tensor d_cov = has_cov ? part2(6, N, 6) : Tensor();
        if (N > 0) {
            GsrView view = make_view(H, W, sd["tanfovx"].toDouble(), sd["tanfovy"].toDouble(), sd["scale_modifier"].toDouble(), sd["sh_degree"].toInt(),
                                     sd["prefiltered"].toBool(), sd["debug"].toBool(), cam, sd["raw"].toBool(),
                                     (int)sd["flags"].toInt() | (int)sd["cam_flags"].toInt());
            if (rest.numel() > 0) view.shs_rest = rest.data_ptr<float>();
            if (d_rest.defined()) view.dL_dshs_rest = d_rest.data_ptr<float>();
            if (stats.bwd_prepared == 2) { view.grad_clear = flat.data_ptr<float>(); view.grad_clear_floats = flat.numel(); }
            Scratch tmp(dev);
            int rc;
            {
                c10::OptionalDeviceGuard guard(dev);
                auto stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
                rc = g_api.backward(&view, (int32_t)N, (int32_t)K, fptr(m3), has_sh ? fptr(shc) : nullptr, has_col ? fptr(col) : nullptr, fptr(op),
                                    has_sr ? fptr(sc) : nullptr, has_sr ? fptr(rot) : nullptr, has_cov ? fptr(cov) : nullptr, radii.data_ptr<int32_t>(),
                                    fptr(gc), fptr(gd), fptr(ga), geom.data_ptr(), binb.data_ptr(), img.data_ptr(), g_pass_fwd_stats ? &stats : nullptr,
                                    fptr_mut(d_m3), fptr_mut(d_m2), fptr_mut(d_sh), fptr_mut(d_col), fptr_mut(d_op), fptr_mut(d_sc), fptr_mut(d_rot),
                                    fptr_mut(d_cov), tmp.alloc(), (gsr_stream_t)stream);
            }
            stv[6] = 0;                                        // bwd_prepared is one-shot: a second backward clears its own accumulators
            sd["stats"] = stv;
            check(rc, "gsr_backward");
        }
        auto as = [&](const Tensor& g, const char* key) -> Tensor {
            if (!g.defined()) return Tensor();
            auto s = sd[key].toIntVector();
            if (s.empty()) return Tensor();
            return g.sizes().vec() == s ? g : g.reshape(s);
        };
        // means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, sh_rest, 4 camera tensors, 12 scalars
        variable_list out(25);
        out[0] = as(d_m3, "s_m3"); out[1] = sd["m2_is_holder"].toBool() ? d_m2 : Tensor(); out[2] = as(d_sh, "s_sh"); out[3] = as(d_col, "s_col");
        out[4] = as(d_op, "s_op"); out[5] = as(d_sc, "s_sc"); out[6] = as(d_rot, "s_rot"); out[7] = as(d_cov, "s_cov"); out[8] = as(d_rest, "s_rest");
        return out;
    }
};

std::vector<Tensor> rasterize(const Tensor& means3D, const Tensor& means2D, const c10::optional<Tensor>& sh, const c10::optional<Tensor>& colors_precomp,
                              const Tensor& opacities, const c10::optional<Tensor>& scales, const c10::optional<Tensor>& rotations,
                              const c10::optional<Tensor>& cov3Ds_precomp, const c10::optional<Tensor>& sh_rest, const Tensor& bg, const Tensor& viewmatrix,
                              const Tensor& projmatrix, const Tensor& campos, int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier,
                              int64_t sh_degree, bool prefiltered, bool debug, bool raw, int64_t extra_flags, bool force_grad_block) {
    // (an absent input travels as ONE shared empty tensor: Function::apply records the device of every tensor argument, an undefined one has none)
    static const Tensor kNone = at::empty({0}, at::TensorOptions().dtype(at::kFloat));
    auto o = [&](const c10::optional<Tensor>& t) { return t.has_value() && t->defined() ? *t : kNone; };
    const bool grad_mode = at::GradMode::is_enabled();        // (inside forward() grad mode is always off)
    return Rasterize::apply(means3D, means2D, o(sh), o(colors_precomp), opacities, o(scales), o(rotations), o(cov3Ds_precomp), o(sh_rest), bg, viewmatrix,
                            projmatrix, campos, H, W, tanfovx, tanfovy, scale_modifier, sh_degree, prefiltered, debug, raw, extra_flags, grad_mode,
                            force_grad_block);
}

void init(const std::string& path) {
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    TORCH_CHECK(h, "gsr_torch: cannot open ", path, ": ", dlerror());
    g_api.forward = reinterpret_cast<decltype(g_api.forward)>(dlsym(h, "gsr_forward"));
    g_api.backward = reinterpret_cast<decltype(g_api.backward)>(dlsym(h, "gsr_backward"));
    g_api.forward_complete = reinterpret_cast<decltype(g_api.forward_complete)>(dlsym(h, "gsr_forward_complete"));
    g_api.last_error = reinterpret_cast<decltype(g_api.last_error)>(dlsym(h, "gsr_last_error"));
    g_api.abi_version = reinterpret_cast<decltype(g_api.abi_version)>(dlsym(h, "gsr_abi_version"));
    TORCH_CHECK(g_api.forward && g_api.backward && g_api.forward_complete && g_api.last_error && g_api.abi_version, "gsr_torch: ", path,
                " does not export the rasterizer's entry points");
    TORCH_CHECK(g_api.abi_version() == GSR_ABI_VERSION, "gsr_torch was built for ABI ", GSR_ABI_VERSION, ", ", path, " reports ", g_api.abi_version());
}

// (M, M_ref, V, max_tile, seg_shift, bwd_prepared, speculated, pending, N, H, W, K) of the calling thread's last forward; complete = true
// collects a pending asynchronous forward's counts first (gsr_forward_complete)
std::vector<int64_t> last_stats(bool complete) {
    if (complete && g_last_stats.pending) check(g_api.forward_complete(&g_last_stats), "gsr_forward_complete");
    const GsrStats& s = g_last_stats;
    return {s.num_instances, s.num_instances_ref, s.num_visible, s.max_tile_count, s.seg_shift, s.bwd_prepared, s.speculated, s.pending,
            g_last_dims[0], g_last_dims[1], g_last_dims[2], g_last_dims[3]};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("init", &init, "bind to the libgsr.so at this path (the one dreamgaussian_amd._lib loaded)");
    m.def("rasterize", &rasterize, "GaussianRasterizer.forward through a C++ autograd function: (color, radii, depth, alpha)");
    m.def("last_stats", &last_stats);
    m.def("set_pass_fwd_stats", [](bool on) { g_pass_fwd_stats = on; });
    m.def("abi_version", []() { return (int)GSR_ABI_VERSION; });
}
