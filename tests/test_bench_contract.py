"""bench.py's bookkeeping: algorithmic bytes of SURVEY 8(d), workload table = BASELINE.json."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_alg_bytes_match_survey_table():
    # SURVEY 8(d) "Config sizes": #3 N=1M K=16 800^2 M=4.22M -> ~1.07 GB ; #2 -> ~154 MB ; #1 -> ~6.7 MB
    ab = bench.alg_bytes(1_000_000, 16, 1_000_000, 4_221_561, 800 * 800)
    assert abs(ab["total"] / 1e9 - 1.07) < 0.01
    ab = bench.alg_bytes(100_000, 16, 100_000, 1_040_470, 800 * 800)
    assert abs(ab["total"] / 1e6 - 154) < 2
    ab = bench.alg_bytes(5_000, 1, 5_000, 43_824, 256 * 256)
    assert abs(ab["total"] / 1e6 - 6.7) < 0.2
    assert set(ab) == {"preprocess_fwd", "scatter", "tile_sort", "render_fwd", "render_bwd", "preprocess_bwd", "total"}


def test_workloads_are_baseline_configs():
    cfgs = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    assert "1M Gaussians, SH degree 3, 800" in cfgs[2] and bench.WORKLOADS["1M-800-sh3"]["cfg"] == 2
    assert "100k Gaussians, SH degree 3, 800" in cfgs[1] and bench.WORKLOADS["100k-800-sh3"]["cfg"] == 1
    assert "250k Gaussians, 512" in cfgs[3] and bench.WORKLOADS["250k-512-sh0"]["N"] == 250_000
    assert bench.kernel_family("tile_sort_large") == "tile_sort" and bench.kernel_family("render_bwd") == "render_bwd"


def test_traffic_stamp_follows_the_device_code_only(tmp_path, monkeypatch):
    """profiles/pmc_traffic.json is trusted by bench.py only when it was collected with the same DEVICE code
    (build.kernel_digest): a host-side edit of gsr_api.hip / gsr.h must not invalidate it, a kernel edit must."""
    import shutil
    from dreamgaussian_amd import build
    src = os.path.dirname(build.CSRC)
    root = tmp_path / "pkg"
    shutil.copytree(build.CSRC, root / "dreamgaussian_amd" / "csrc")
    (root / "include").mkdir()
    shutil.copy(os.path.join(os.path.dirname(src), "include", "gsr.h"), root / "include" / "gsr.h")
    monkeypatch.setattr(build, "HERE", str(root / "dreamgaussian_amd"))
    monkeypatch.setattr(build, "CSRC", str(root / "dreamgaussian_amd" / "csrc"))
    full0, dev0 = build._digest(), build.kernel_digest()
    with open(root / "dreamgaussian_amd" / "csrc" / "gsr_api.hip", "a") as fh:
        fh.write("\n// host-side edit\n")
    assert build._digest() != full0 and build.kernel_digest() == dev0
    with open(root / "dreamgaussian_amd" / "csrc" / "gsr_render.hip", "a") as fh:
        fh.write("\n// kernel edit\n")
    assert build.kernel_digest() != dev0


def test_committed_traffic_file_uses_the_calibrated_read_factors():
    import json
    from dreamgaussian_amd import build
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    doc = json.load(open(path))
    # a stale file FAILS here (round-4 review: it used to skip, and the driver's bench line lost its `roofline.traffic`): whoever edits
    # a kernel re-collects the PMC passes in the same GPU call (`bash tools/gpu_r5.sh pmc`) and commits profiles/pmc_traffic.json
    assert doc["source_digest"] == build.kernel_digest(), \
        "profiles/pmc_traffic.json was collected with other kernel sources: bench.py would report roofline.traffic = null; " \
        "re-collect with `bash tools/gpu_r5.sh pmc` on the GPU box and commit the file"
    rb = doc["1M-800-sh3/blob"]["render_bwd"]
    assert rb["read_factor"] == 1.0 and abs(rb["hbm_bytes"] - (rb["read_bytes_raw"] + rb["write_bytes"])) < 1.0
    k1 = doc["1M-800-sh3/blob"]["preprocess_fwd<false>"]
    assert k1["read_factor"] == 2.0                       # a streaming kernel: coalesced reads are counted at half
    k6 = doc["1M-800-sh3/blob"]["preprocess_bwd_compact<false>"]
    assert k6["read_factor"] == 1.0                       # gathers of the live Gaussians' rows


def test_order_morton_permutes_the_render_inputs():
    """`bench.py --order morton` must hand the RENDER step a permuted scene (round 3 permuted it in run_sds only and the
    render lines stamped "order": "morton" measured the given order). build_inputs is what both step kinds call."""
    import torch
    wl = dict(N=3000, deg=0, W=64, H=64)
    dev = torch.device("cpu")
    given, _, _, _ = bench.build_inputs(wl, "blob", dev, 0.0)
    mort, _, _, _ = bench.build_inputs(wl, "blob", dev, 0.0, "morton")
    assert not torch.equal(given["means3D"], mort["means3D"])
    # the same Gaussians, rows moved together
    from dreamgaussian_amd.densify import morton_order
    perm = morton_order(given["means3D"]).long()
    for k in given:
        assert torch.equal(given[k][perm], mort[k])
    # and main() routes a.order into it for the render step
    src = open(bench.__file__).read()
    assert src.count("build_inputs(wl, a.kind, dev, azimuth, a.order)") == 2


def test_alg_bytes_attribution_keeps_the_total():
    """Whoever stores the zeros of the gradient arrays (the per-Gaussian backward, or -- round 5 -- a compositing kernel on the side),
    the compulsory bytes of SURVEY 8(d) add up to the same total, and the default split is the survey's."""
    N, K, V, M, P = 1_000_000, 16, 1_000_000, 4_221_565, 640_000
    base = bench.alg_bytes(N, K, V, M, P)
    assert base["render_bwd"] == 84 * V + 8 * M + 28 * P == 135_692_520        # what BENCH_r0N's roofline.frac is computed from
    for who in ("render_fwd", "render_bwd", "preprocess_bwd"):
        ab = bench.alg_bytes(N, K, V, M, P, who)
        assert ab["total"] == base["total"]
        assert ab[who] - (base[who] - (N * (44 + 12 * K + 12) if who == "preprocess_bwd" else 0)) == N * (44 + 12 * K + 12)


def test_issue_budget_classes():
    """tools/issue_budget.py: the issue class of a vector instruction as profiles/r04_valu_rates.txt groups them."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("issue_budget", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "issue_budget.py"))
    ib = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ib)
    c = ib.classify
    assert c("v_fmac_f32_e32 v61, v62, v59") == "plain" and c("v_lshl_add_u32 v70, v111, 4, v3") == "plain"
    assert c("v_lshl_add_u32 v70, v111, 4, s73") == "second"                    # an SGPR source
    assert c("v_med3_f32 v112, v84, 0, v103") == "second" and c("v_cmp_le_f32_e64 s[8:9], s76, v58") == "second"
    assert c("v_cndmask_b32_e32 v84, 0, v58, vcc") == "second"
    assert c("v_exp_f32 v58, v61") == "trans" and c("v_rcp_f32 v119, v114") == "trans"
    assert c("v_mov_b32_dpp v82, v68 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1") == "dpp"
    assert c("ds_read_b128 v[26:29], v52 offset:1056") == "lds" and c("s_waitcnt lgkmcnt(0)") == "wait" and c("s_and_b64 s[8:9], vcc, s[8:9]") == "salu"
