#!/usr/bin/env python
"""How many Gaussians of a bench scene receive a gradient at all, and how coherent the dead ones are per wave (64 rows) and per
K6 batch (256 rows), in the given order and along the Z-order curve -- the numbers behind K6's "skip the SH rows of Gaussians
whose 2D gradient is zero" (round-3 verdict, item 5).   python tools/grad_stats.py [--workload 1M-800-sh3] [--kind blob]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="1M-800-sh3")
    ap.add_argument("--kind", default="blob")
    a = ap.parse_args()
    import dreamgaussian_amd as D
    dev = torch.device("cuda", 0)
    wl = bench.WORKLOADS[a.workload]
    out = {"workload": a.workload, "kind": a.kind}
    for order in ("given", "morton"):
        sc, _, rs, grads = bench.build_inputs(wl, a.kind, dev, 0.0, order)
        t = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
        m2d = torch.zeros(wl["N"], 3, device=dev, requires_grad=True)
        rast = D.GaussianRasterizer(raster_settings=rs)
        c, r, d, al = rast(means3D=t["means3D"], means2D=m2d, shs=t["shs"], colors_precomp=None, opacities=t["opacities"],
                           scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
        torch.autograd.backward([c, d, al], [g.to(dev) for g in grads])
        live = (t["shs"].grad.abs().amax(dim=(1, 2)) > 0) | (t["opacities"].grad.abs().reshape(-1) > 0)
        N = wl["N"]
        st = D.last_stats()
        row = {"N": N, "visible": int((r > 0).sum()), "with_gradient": int(live.sum()), "frac_with_gradient": round(float(live.float().mean()), 4),
               "M_emitted": st.get("M"), "M_ref": st.get("M_ref")}
        for w in (64, 256):
            pad = (-N) % w
            lv = torch.cat([live, torch.zeros(pad, dtype=torch.bool, device=dev)]).view(-1, w)
            row[f"groups_of_{w}_all_dead"] = round(float((~lv.any(dim=1)).float().mean()), 4)
            row[f"mean_live_rows_per_group_of_{w}"] = round(float(lv.float().sum(1).mean()), 2)
        out[order] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
