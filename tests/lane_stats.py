"""Lane utilisation of the compositing backward, from the oracle (CPU; analysis aid, not a test).

    python tests/lane_stats.py --n 1000000 --size 800 --kind blob --tiles 80

For a sample of tiles it recomputes alpha for every (pixel, listed Gaussian) pair the backward visits
(list positions up to the block's deepest contributor) and counts, per 8x8 block (= one wave of K5b):
  iterations now     entries with at least one blended pixel in the 8x8 block (the wave loops over these)
  blended lanes      pixels that actually blend, per iteration (of 64)
  iterations, quads  if each 16-lane row owned a 4x4 quad with its OWN list: max over the 4 quads of the
                     entries blending in that quad (rows iterate independently, the wave loops to the longest)
DESIGN.md section 7 uses these numbers to price a quad-row backward."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gs_oracle as O


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--kind", default="blob")
    ap.add_argument("--tiles", type=int, default=80)
    a = ap.parse_args()
    sc = O.make_scene(a.n, 0, 0, a.kind)
    S = O.make_settings(O.orbit_pose(0.0, 0.0, 2.0), a.size, a.size, sh_degree=0)
    with torch.no_grad():
        _, _, _, _, aux = O.rasterize(sc["means3D"], None, sc["opacities"], S, shs=sc["shs"], scales=sc["scales"],
                                      rotations=sc["rotations"], return_aux=True)
    pre, ids, ranges = aux["pre"], aux["ids"], aux["ranges"]
    nc = aux["n_contrib"].reshape(a.size, a.size).numpy()
    gx = (a.size + 15) // 16
    nonempty = [t for t in range(gx * gx) if ranges[t + 1] > ranges[t]]
    rs = np.random.RandomState(0)
    sample = rs.choice(nonempty, size=min(a.tiles, len(nonempty)), replace=False)
    xy, conic, opac = pre["xy"].double().numpy(), pre["conic"].double().numpy(), pre["opacity"].double().numpy().reshape(-1)
    it_now = it_quad = it_half = blended = visited_entries = it_quad_bbox = it_now_bbox = 0
    for t in sample:
        g = ids[ranges[t]:ranges[t + 1]]
        ty, tx = divmod(int(t), gx)
        ys, xs = np.meshgrid(np.arange(16) + ty * 16, np.arange(16) + tx * 16, indexing="ij")
        inside = (ys < a.size) & (xs < a.size)
        last = np.where(inside, nc[np.minimum(ys, a.size - 1), np.minimum(xs, a.size - 1)], 0)      # 1-based last contributor
        depth = int(last.max())
        if depth == 0:
            continue
        g = g[:depth]
        dx = xy[g, 0][:, None, None] - xs[None]
        dy = xy[g, 1][:, None, None] - ys[None]
        power = -0.5 * (conic[g, 0][:, None, None] * dx * dx + conic[g, 2][:, None, None] * dy * dy) - conic[g, 1][:, None, None] * dx * dy
        alpha = np.minimum(0.99, opac[g][:, None, None] * np.exp(np.minimum(power, 0)))
        pos = np.arange(1, depth + 1)[:, None, None]
        ok = (power <= 0) & (alpha >= 1 / 255) & (pos <= last[None]) & inside[None]             # pairs the backward blends
        # what a kernel can know before evaluating pixels: the axis-aligned box of the alpha >= 1/255 ellipse
        c = 2.0 * np.log(np.maximum(255.0 * opac[g], 1.0 + 1e-12))
        det = conic[g, 0] * conic[g, 2] - conic[g, 1] ** 2
        ex = np.sqrt(c * conic[g, 2] / det) + 0.5
        ey = np.sqrt(c * conic[g, 0] / det) + 0.5
        vis = (np.abs(dx) <= ex[:, None, None]) & (np.abs(dy) <= ey[:, None, None]) & (pos <= last.max()) & inside[None]
        for by in (0, 8):
            for bx in (0, 8):
                blk = ok[:, by:by + 8, bx:bx + 8]
                hit = blk.any((1, 2))
                it_now += int(hit.sum())
                blended += int(blk.sum())
                q = [blk[:, qy:qy + 4, qx:qx + 4].any((1, 2)).sum() for qy in (0, 4) for qx in (0, 4)]
                it_quad += int(max(q))
                h = [blk[:, hy:hy + 4, :].any((1, 2)).sum() for hy in (0, 4)]
                it_half += int(max(h))
                vb = vis[:, by:by + 8, bx:bx + 8]
                it_now_bbox += int(vb.any((1, 2)).sum())
                it_quad_bbox += int(max(vb[:, qy:qy + 4, qx:qx + 4].any((1, 2)).sum() for qy in (0, 4) for qx in (0, 4)))
        visited_entries += depth
    print(f"{len(sample)} tiles of {len(nonempty)}; list positions visited {visited_entries}")
    print(f"iterations now (8x8 block, one entry per trip): {it_now}; blended lanes per iteration {blended / max(it_now, 1):.1f} of 64")
    print(f"iterations with two 8x4 halves owning their own lists: {it_half} ({it_now / max(it_half, 1):.2f}x fewer)")
    print(f"iterations with four 4x4 quads owning their own lists: {it_quad} ({it_now / max(it_quad, 1):.2f}x fewer)")
    print(f"with a bounding-box cull instead of the ideal one: 8x8 block {it_now_bbox}, quads {it_quad_bbox} "
          f"({it_now_bbox / max(it_quad_bbox, 1):.2f}x fewer; vs today's exact block cull {it_now / max(it_quad_bbox, 1):.2f}x)")


if __name__ == "__main__":
    main()
