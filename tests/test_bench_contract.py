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
