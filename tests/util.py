"""Shared helpers for the parity tests: run the HIP path and the oracle on the same inputs."""
import torch

from oracle import gs_oracle as O
import dreamgaussian_amd as D


def settings_to(S, dev, dtype=torch.float32):
    return D.GaussianRasterizationSettings(*[x.to(device=dev, dtype=dtype) if torch.is_tensor(x) else x for x in S])


def run_hip(sc, S, dev, weights=None, means2D=True):
    t = {k: v.detach().to(dev).requires_grad_(True) for k, v in sc.items()}
    N = t["means3D"].shape[0]
    m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
    out = D.GaussianRasterizer(raster_settings=settings_to(S, dev))(
        means3D=t["means3D"], means2D=m2d, shs=t.get("shs"), colors_precomp=t.get("colors_precomp"),
        opacities=t["opacities"], scales=t.get("scales"), rotations=t.get("rotations"),
        cov3D_precomp=t.get("cov3D_precomp"))
    grads = None
    if weights is not None:
        torch.autograd.backward([out[0], out[2], out[3]], [w.to(dev) for w in weights])
        grads = {k: v.grad.detach().cpu() for k, v in t.items()}
        grads["means2D"] = m2d.grad.detach().cpu()
    return [o.detach().cpu() for o in out], grads, D.last_stats()


def run_oracle(sc, S, weights=None, dtype=torch.float32):
    t = {k: v.detach().to(dtype).requires_grad_(True) for k, v in sc.items()}
    N = t["means3D"].shape[0]
    m2d = torch.zeros(N, 3, dtype=dtype, requires_grad=True)
    S2 = O.Settings(*[x.to(dtype) if torch.is_tensor(x) else x for x in S])
    c, r, d, a, aux = O.rasterize(t["means3D"], m2d, t["opacities"], S2, shs=t.get("shs"),
                                  colors_precomp=t.get("colors_precomp"), scales=t.get("scales"),
                                  rotations=t.get("rotations"), cov3D_precomp=t.get("cov3D_precomp"),
                                  return_aux=True)
    grads = None
    if weights is not None:
        torch.autograd.backward([c, d, a], [w.to(dtype) for w in weights])
        grads = {k: v.grad.detach() for k, v in t.items()}
        grads["means2D"] = m2d.grad.detach()
    return [c.detach(), r, d.detach(), a.detach()], grads, aux


def weights_for(H, W, seed=1):
    g = torch.Generator().manual_seed(seed)
    return [torch.rand(3, H, W, generator=g), torch.rand(1, H, W, generator=g), torch.rand(1, H, W, generator=g)]


# Stated tolerances (fp32 HIP kernels vs the oracle; BASELINE.json: grads within 1e-4 rel):
FWD_ATOL = 2e-5          # color / alpha, absolute (values are O(1)); depth relative to its max
GRAD_RTOL = 1e-4         # per attribute, relative to that attribute's max |grad| (float64 oracle as arbiter)


# Row-relative bound: the max-normalised tolerance above lets a Gaussian whose gradient is small
# against the attribute's maximum be badly wrong. For every row (Gaussian) with |ref|_inf >= ROW_REL_FLOOR * max |ref| the
# ratio |err|_inf / |ref|_inf is taken; its 99.9th percentile over those rows must stay below ROW_REL_P999 and its
# median below ROW_REL_MEDIAN. (Rows under the floor are covered by the max-normalised bound only: their relative error is
# dominated by the cancellation of thousands of per-pixel terms.)
ROW_REL_FLOOR = 1e-3
ROW_REL_P999 = 5e-3         # small scenes (a few thousand rows: the 99.9th percentile is the second-worst row); fp32-vs-fp64 of the
                            # ORACLE ITSELF reaches 1.2e-3 there (tests/test_oracle.py): fp32 rounding, not a kernel property
ROW_REL_P999_FULL = 7e-4    # BASELINE configs[1]-[3] at full size (>= 20k rows): observed on the MI355X <= 5.2e-4 (round 6: the bound follows what is
                            # observed -- it was 1e-3; north_star's literal "1e-4 rel" holds for the MEDIAN row 5x over and for ~97-99 % of the rows, not
                            # for the last few in a thousand: the histogram below is printed with every full-size comparison)
ROW_REL_MEDIAN = 5e-5       # observed on the MI355X: <= 1.8e-5 (BASELINE configs[3], means2D)
REPORT = {}              # what the last assert_* calls observed (printed by the full-size tests)


# A pixel with an ambiguous discrete decision may differ by ONE flipped decision. alpha >= 1/255: the pair contributes
# alpha T <= (1/255) / (1 - 1/255) ~ 4e-3. T (1 - alpha) >= 1e-4 (the stop): the pair at the stop contributes alpha T with
# T (1 - alpha) ~ 1e-4, i.e. up to 1e-4 * 0.99 / 0.01 = 9.9e-3 for a near-opaque Gaussian (alpha is clamped at 0.99) -- the
# trained stage-1 model has those (tools/run_stage1.py's oracle check: 2.4e-2 of depth = 9.9e-3 * z 2.4).
FRAGILE_ABS = 1.1e-2
FRAGILE_GRAD_REL = 5e-2  # ... and the Gaussian of that pair by its single-pair gradient share
ONE_FLIP_ALPHA = (1.0 / 255.0) / (1.0 - 1.0 / 255.0) * 1.02     # |d alpha| of ONE flipped alpha >= 1/255 decision (T <= 1)


def fragile_abs_for(aux):
    """What ONE flipped decision can move a pixel by in THIS scene (round 5: the 1.1e-2 above belongs to opacity-0.99 pairs and was
    applied to every scene): the alpha >= 1/255 flip is worth <= 4.0e-3, the stop flip 1e-4 * a / (1 - a) with a = the scene's
    largest opacity, clamped at 0.99 like alpha itself -- 9e-4 at opacity 0.9, 9.9e-3 at 0.99. Without the oracle's per-Gaussian
    state (the committed golden vector): the near-opaque bound."""
    if aux is None or "pre" not in aux or "opacity" not in aux["pre"]:
        return FRAGILE_ABS
    pre = aux["pre"]
    op = pre["opacity"].detach().double().reshape(-1)
    if "valid" in pre:
        op = op[pre["valid"].reshape(-1)]
    a = min(0.99, float(op.max())) if op.numel() else 0.0
    return max(ONE_FLIP_ALPHA, 1.02e-4 * a / (1.0 - a)) + FWD_ATOL


def fragile_caps(n_pixels, n_gauss):
    """How much of the oracle-flagged set may actually differ: 2e-4 of the pixels / Gaussians (at least 8)."""
    return max(8, int(2e-4 * n_pixels)), max(8, int(2e-4 * n_gauss))


def assert_forward_close(ho, oo, aux=None, atol=FWD_ATOL):
    """radii equal up to boundary flips; color/depth/alpha within atol (depth: atol * max depth)
    on every pixel, except pixels where the ORACLE flags an ambiguous discrete decision
    (aux["fragile_pixels"]: an alpha >= 1/255 or T(1-alpha) >= 1e-4 test within fp32 evaluation
    noise of its threshold -- any two fp32 implementations may decide those differently), which
    must be within one minimal contribution (FRAGILE_ABS)."""
    dr = (ho[1].to(torch.int64) - oo[1].to(torch.int64)).abs()
    # radius = ceil(3 sqrt(lambda)) is discontinuous: fp32 rounding differences (fma contraction,
    # sqrt) may flip a value sitting on an integer boundary, by one, on a handful of Gaussians
    if dr.numel():
        nbad = int((dr != 0).sum())
        REPORT["radii_mismatch"] = nbad
        assert int(dr.max()) <= 1 and nbad <= max(1, dr.numel() // 5000), \
            f"radii mismatch on {nbad} Gaussians (max |diff| {int(dr.max())})"
        if nbad and aux is not None and "pre" in aux and "radius_raw" in aux["pre"]:
            # ... and every one of them IS a boundary case of ceil(): the oracle's own 3 sqrt(lambda) lies within fp32 rounding of
            # the projection chain (a few 1e-6 relative) of an integer -- or the radius differs because the fp32 validity decision
            # (culled <-> radius 0) sits on its own threshold, which the fragile set names
            raw = aux["pre"]["radius_raw"].double()[dr != 0]
            dist = (raw - torch.round(raw)).abs() / raw.clamp_min(1.0)
            culled_flip = ((ho[1] == 0) != (oo[1] == 0))[dr != 0]
            REPORT["radii_mismatch_worst_boundary_distance"] = float(dist[~culled_flip].max()) if (~culled_flip).any() else 0.0
            assert bool(((dist <= 2e-5) | culled_flip).all()), \
                f"a radius differs away from a ceil() boundary: relative distance to the next integer {dist.max().item():.2e}"
    frag = None if aux is None else torch.as_tensor(aux["fragile_pixels"]).bool()
    fabs = fragile_abs_for(aux)
    different = None
    for name, i in (("color", 0), ("depth", 2), ("alpha", 3)):
        ref = oo[i].double()
        err = (ho[i].double() - ref).abs()
        scale = max(1.0, ref.abs().max().item())
        strict = err if frag is None else err.masked_fill(frag[None].expand_as(err), 0.0)
        assert strict.max().item() <= atol * scale, \
            f"{name}: max abs err {strict.max().item():.3e} > {atol * scale:.1e} on a pixel without ambiguous decisions"
        if name == "color" and aux is not None and "pre" in aux and "color" in aux["pre"] and aux["pre"]["color"].numel():
            scale = max(scale, float(aux["pre"]["color"].max()))          # (a flip moves the pixel by alpha T x the GAUSSIAN's colour)
        # a flagged pixel: within TWO flipped decisions of this scene's kind at worst ...
        assert err.max().item() <= 2 * fabs * scale, f"{name}: max abs err {err.max().item():.3e} on a fragile pixel (one flip here: {fabs * scale:.2e})"
        d = (err > atol * max(1.0, ref.abs().max().item())).any(0)
        different = d if different is None else (different | d)
        two = (err > fabs * scale).any(0)
        # ... and all but a handful within one
        assert int(two.sum()) <= max(2, int(1e-5 * two.numel())), f"{name}: {int(two.sum())} pixels beyond one flipped decision"
    if frag is not None and different is not None:
        # ... and FEW of the flagged pixels differ at all (round 5: counted in every oracle comparison, not only at the BASELINE sizes)
        cap_p, _ = fragile_caps(different.numel(), 0)
        REPORT["flagged_pixels"] = int(frag.sum())
        REPORT["flagged_pixels_different"] = int(different.sum())
        assert int(different.sum()) <= cap_p, f"{int(different.sum())} of {int(frag.sum())} flagged pixels differ (cap {cap_p})"


def assert_counts_explained(st, aux):
    """V (visible Gaussians) equal; M_ref (tile instances under the reference's 3-sigma rect rule) equal up to the Gaussians whose
    rect sits ON a discontinuity of an fp32 implementation: ceil(3 sqrt(lambda)) within 2e-5 (relative) of an integer, or a rect
    edge within 2e-5 tiles of a tile boundary. Each such Gaussian can move the count by at most one row or column of its rect."""
    assert st["V"] == aux["V"], (st["V"], aux["V"])
    d = int(st["M_ref"]) - int(aux["M"])
    REPORT["M_ref_mismatch"] = d
    if d == 0:
        return
    pre = aux["pre"]
    raw = pre["radius_raw"].double()
    near = (((raw - torch.round(raw)).abs() / raw.clamp_min(1.0)) <= 2e-5) | (pre["edge_margin"].double() <= 2e-5)
    near = near & pre["valid"]
    rect = pre["rect"]
    slack = int((torch.maximum(rect[:, 2] - rect[:, 0], rect[:, 3] - rect[:, 1]) + 1)[near].sum())
    REPORT["M_ref_boundary_gaussians"] = int(near.sum())
    assert abs(d) <= slack, f"M_ref differs by {d} but the {int(near.sum())} Gaussians on a rect discontinuity explain at most {slack}"


def og32_if_near_opaque(sc, S, w):
    """The fp32 oracle's gradients for assert_grads_close(og32=) -- ONLY for scenes that hold near-opaque Gaussians (opacity >= 0.99:
    two of them in a row sit exactly on the 1e-4 stop, see assert_grads_close). Every other scene is held to the fp64 oracle alone."""
    if float(sc["opacities"].max()) < 0.99:
        return None
    _, og32, _ = run_oracle(sc, S, w, torch.float32)
    return og32


def grad_floors(sc, og):
    """Natural magnitude of each gradient, for attributes whose true gradient may cancel to ~0:
    dL/dq is a sum of dL/dM * s terms (exactly 0 for isotropic Gaussians), so its rounding noise
    scales with |dL/ds| * |s|, not with |dL/dq|."""
    floors = {}
    if "rotations" in og and "scales" in og:
        floors["rotations"] = (og["scales"].double().abs().max() * sc["scales"].double().abs().max()).item()
    return floors


def assert_grads_close(hg, og, aux=None, rtol=GRAD_RTOL, floors=None, row_rel_p999=ROW_REL_P999, og32=None, og32_cap=0.01):
    """Each attribute's gradient within rtol * max|ref| (row-wise strict), except the rows of
    Gaussians the oracle flags as part of an ambiguous discrete decision (see above).
    og32 (optional): the SAME oracle evaluated in fp32. An ambiguous decision at a pixel also moves the gradients of the
    other Gaussians blending there -- behind it through T, in front of it through the composited-behind term / (1 - alpha),
    x100 next to a near-opaque Gaussian (the trained stage-1 model; two alpha = 0.99 Gaussians in a row sit exactly ON the stop:
    (1 - 0.99)^2 = 1e-4, fp32 drops the second, fp64 keeps it) -- rows the per-pair flags cannot name. With og32 a row that
    misses the fp64 oracle must instead match the fp32 oracle within the same rtol * max|ref| (it took the decision the way
    plain fp32 arithmetic takes it), and at most og32_cap of the rows may need that."""
    floors = floors or {}
    fg = None if aux is None else torch.as_tensor(aux["fragile_gaussians"]).bool()
    for k, ref in og.items():
        ref = ref.double()
        got = hg[k].double().reshape(ref.shape)
        scale = max(ref.abs().max().item(), floors.get(k, 0.0))
        err = (got - ref).abs().reshape(ref.shape[0], -1).max(1).values if ref.shape[0] else torch.zeros(0)
        strict = err if fg is None else err[~fg]
        if og32 is not None and ref.shape[0]:
            err32 = (got - og32[k].double().reshape(ref.shape)).abs().reshape(ref.shape[0], -1).max(1).values
            miss = err > rtol * scale + 1e-9
            if fg is not None:
                miss = miss & ~fg
            REPORT[f"fp32_decided_{k}"] = int(miss.sum())
            if int(miss.sum()):
                print(f"[fp32-decided] d{k}: {int(miss.sum())} of {ref.shape[0]} rows miss the fp64 oracle and are held to the fp32 oracle")
            assert int(miss.sum()) <= max(3, int(og32_cap * ref.shape[0])), f"d{k}: {int(miss.sum())} rows miss the fp64 oracle"
            if miss.any():
                assert err32[miss].max().item() <= rtol * scale + 1e-9, \
                    f"d{k}: a row misses the fp64 oracle by {err[miss].max().item():.3e} and the fp32 oracle by {err32[miss].max().item():.3e} (tol {rtol * scale:.3e})"
            strict = strict[strict <= rtol * scale + 1e-9]
            err = torch.where(miss, torch.zeros_like(err), err)
        if strict.numel():
            assert strict.max().item() <= rtol * scale + 1e-9, \
                f"d{k}: max abs err {strict.max().item():.3e} vs {rtol:.0e} * scale {scale:.3e}"
        if err.numel():
            assert err.max().item() <= FRAGILE_GRAD_REL * scale + 1e-9, f"d{k}: fragile row err {err.max().item():.3e}"
            if fg is not None:      # few of the flagged Gaussians differ at all (round 5: counted in every oracle comparison)
                nbad = int(((err > rtol * scale + 1e-9) & fg).sum())
                REPORT[f"flagged_rows_different_{k}"] = nbad
                assert nbad <= fragile_caps(0, ref.shape[0])[1], f"d{k}: {nbad} of {int(fg.sum())} flagged Gaussians differ"
        # row-relative statistic over the rows that carry a gradient worth the name
        if ref.shape[0] and k not in floors:
            rn = ref.abs().reshape(ref.shape[0], -1).max(1).values
            sel = rn >= ROW_REL_FLOOR * scale
            if fg is not None:
                sel = sel & ~fg
            if int(sel.sum()) >= 10:
                rel = (err[sel] / rn[sel]).sort().values
                p999 = rel[min(rel.numel() - 1, int(0.999 * rel.numel()))].item()
                med = rel[rel.numel() // 2].item()
                # how the rows above the floor are distributed against north_star's literal 1e-4 (fractions of rows at or below each bound)
                hist = {f"le_{b:g}": round(float((rel <= b).double().mean()), 5) for b in (1e-5, 3e-5, 1e-4, 3e-4, 1e-3)}
                REPORT[f"rowrel_{k}"] = dict(rows=int(sel.sum()), median=med, p999=p999, max=rel[-1].item(), **hist)
                assert p999 <= row_rel_p999 and med <= ROW_REL_MEDIAN, \
                    f"d{k}: row-relative error median {med:.2e} / p99.9 {p999:.2e} over {int(sel.sum())} rows"


# ---- bounded fragile set -----------------------------------------------------


def fragile_report(ho, oo, hg, og, aux, sc=None, floors=None):
    """How much of the oracle-flagged (ambiguous-decision) set actually differs, and by how much.
    Returns a dict; the callers assert caps on it. A flagged pixel that misses the strict bound must look
    like ONE flipped decision: |d alpha| <= ONE_FLIP_ALPHA and |d colour| <= ONE_FLIP_ALPHA * max colour."""
    floors = floors or {}
    fp = torch.as_tensor(aux["fragile_pixels"]).bool()
    fg = torch.as_tensor(aux["fragile_gaussians"]).bool()
    err_c = (ho[0].double() - oo[0].double()).abs().amax(0)
    err_a = (ho[3].double() - oo[3].double()).abs()[0]
    bad = ((err_c > FWD_ATOL) | (err_a > FWD_ATOL)) & fp
    cmax = max(1.0, float(aux["pre"]["color"].max())) if "pre" in aux else 1.0
    worst_a = float(err_a[bad].max()) if bad.any() else 0.0
    worst_c = float(err_c[bad].max()) if bad.any() else 0.0
    beyond_one = int((bad & ((err_a > ONE_FLIP_ALPHA + FWD_ATOL) | (err_c > ONE_FLIP_ALPHA * cmax + FWD_ATOL))).sum())
    rows_bad = 0
    worst_rel = 0.0
    for k, ref in og.items():
        ref = ref.double()
        got = hg[k].double().reshape(ref.shape)
        scale = max(ref.abs().max().item(), floors.get(k, 0.0)) + 1e-300
        err = (got - ref).abs().reshape(ref.shape[0], -1).max(1).values / scale
        rows_bad = max(rows_bad, int(((err > GRAD_RTOL) & fg).sum()))
        if (fg & (err > GRAD_RTOL)).any():
            worst_rel = max(worst_rel, float(err[fg].max()))
    return dict(flagged_pixels=int(fp.sum()), flagged_gaussians=int(fg.sum()),
                flagged_pixels_different=int(bad.sum()), beyond_one_flip=beyond_one, worst_dalpha=worst_a, worst_dcolor=worst_c, cmax=cmax,
                flagged_gaussians_different=rows_bad, worst_grad_rel=worst_rel)


def assert_fragile_bounded(rep, n_pixels, n_gauss):
    """caps: at most 2e-4 of the pixels / Gaussians (>= 8) may be flagged AND different; a different pixel is
    within ONE flipped decision, except for at most max(2, 1e-5 P) pixels that may hold two."""
    cap_p = max(8, int(2e-4 * n_pixels))
    cap_g = max(8, int(2e-4 * n_gauss))
    assert rep["flagged_pixels_different"] <= cap_p, rep
    assert rep["flagged_gaussians_different"] <= cap_g, rep
    assert rep["beyond_one_flip"] <= max(2, int(1e-5 * n_pixels)), rep
    assert rep["worst_dalpha"] <= 2 * ONE_FLIP_ALPHA + FWD_ATOL, rep
    assert rep["worst_dcolor"] <= 2 * ONE_FLIP_ALPHA * rep["cmax"] + FWD_ATOL, rep
