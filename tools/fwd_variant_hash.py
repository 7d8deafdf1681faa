#!/usr/bin/env python
"""Checksums of the forward's outputs on a few seeded scenes: run once per GSR_FWD setting (the switch is read once per
process) and compare the lines -- equal lines = the two compositing kernels give bit-identical images.

  GSR_FWD=block python tools/fwd_variant_hash.py; GSR_FWD=q python tools/fwd_variant_hash.py
"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import dreamgaussian_amd as D  # noqa: E402
from dreamgaussian_amd import synthetic as S  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for N, deg, size, kind, radius in ((800, 1, 80, "trained", 3.5), (800, 1, 80, "trained", 2.0), (20000, 0, 256, "trained", 2.0),
                                        (100000, 3, 512, "blob", 2.0), (250000, 0, 512, "trained", 2.5)):
        sc = S.make_scene(N, deg, 0, kind)
        rs = S.make_settings(S.orbit_pose(-10.0, 40.0, radius), size, size, sh_degree=deg, device=dev)
        t = {k: v.to(dev) for k, v in sc.items()}
        c, r, d, a = D.GaussianRasterizer(raster_settings=rs)(means3D=t["means3D"], means2D=torch.zeros(N, 3, device=dev), shs=t["shs"],
                                                              colors_precomp=None, opacities=t["opacities"], scales=t["scales"],
                                                              rotations=t["rotations"], cov3D_precomp=None)
        h = hashlib.sha256()
        for x in (c, d, a):
            h.update(x.detach().cpu().numpy().tobytes())
        st = D.last_stats()
        print(f"{N} deg{deg} {size}px {kind} r={radius}: {h.hexdigest()[:16]}  M_ref/V={st['M_ref'] / max(st['V'], 1):.2f}")


if __name__ == "__main__":
    main()
