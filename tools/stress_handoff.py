#!/usr/bin/env python
# tools/stress_handoff.py -- stress of K1's group hand-off (gsr_preprocess_fwd's histogram flush, round 6): many forwards at the headline size, every image word compared with the first (k1_group = 4 default) and
# with the one-range-per-workgroup scheme (k1_group = 1)
import sys, torch, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle.gs_oracle as O
from util import run_hip
from dreamgaussian_amd import _testing as hooks
gpu = torch.device('cuda:0')
for kind, N, size in (("blob", 1_000_000, 800), ("trained", 1_000_000, 800), ("trained", 250_000, 512)):
    sc = O.make_scene(N, 1, 0, kind)
    S = O.make_settings(O.orbit_pose(0.0, 30.0, 2.0), size, size, sh_degree=1)
    hooks.reset(); hooks.set("k1_group", 1)
    base, _, st0 = run_hip(sc, S, gpu, None)
    hooks.reset()
    bad = 0
    t0 = time.time()
    for rep in range(300):
        ho, _, st = run_hip(sc, S, gpu, None)
        ok = all(torch.equal(ho[i], base[i]) for i in range(4)) and st["M"] == st0["M"] and st["max_tile"] == st0["max_tile"]
        bad += 0 if ok else 1
    print(kind, N, size, "300 forwards,", bad, "different from the k1_group=1 result; M", st0["M"], "max_tile", st0["max_tile"], "%.1fs" % (time.time() - t0), flush=True)
