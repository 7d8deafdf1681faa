#!/usr/bin/env python
"""Join the two runs of tools/valu_ceiling.hip: wall times (plain run, JSON lines) and instruction counts
(rocprofv3 --pmc run of the same binary) -> wave-instructions per second per SIMD for the compositing kernels' own mix.

  python tools/valu_ceiling.py gpurun_out/valu_plain.jsonl gpurun_out/valu_pmc [--real profiles/r03_1M-800-sh3_blob_pmc.txt]

Counter values are summed over the chip per dispatch; of the dispatches of one (kernel, grid) the largest SQ_INSTS_VALU is the
timed launch (the warm-up runs fewer rounds). Clock for "cycles": 2.4 GHz (MI355X_MICROARCH.md) -- an upper bound of what
the part sustains, so the cycles per instruction printed here are upper bounds too."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

CLK = 2.4e9
SIMDS = 1024


def main():
    plain = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
    cnt = defaultdict(lambda: defaultdict(float))          # (kernel, grid) -> counter -> value of the timed dispatch
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(lambda: defaultdict(float))
        meta = {}
        for row in csv.DictReader(open(f)):
            d = row["Dispatch_Id"]
            per[d][row["Counter_Name"]] += float(row["Counter_Value"])
            meta[d] = (row["Kernel_Name"].split("(")[0], int(row["Grid_Size"]) // 256)
        for d, c in per.items():
            key = meta[d]
            for name, v in c.items():
                cnt[key][name] = max(cnt[key][name], v)
    print(f"{'kernel':8s} {'w/SIMD':>6s} {'ms':>8s} {'VALU/launch':>12s} {'VALU/batch/wave':>15s} {'Ginstr/s/SIMD':>13s} {'cyc/VALU@2.4GHz':>15s} "
          f"{'ACTIVE_VALU/INSTS':>17s} {'SALU/VALU':>9s} {'LDS/VALU':>8s}")
    for p in plain:
        key = next((k for k in cnt if k[0].endswith(p["kernel"]) and k[1] == p["grid"]), None)
        c = cnt.get(key, {})
        valu = c.get("SQ_INSTS_VALU", 0.0)
        t = p["ms_min"] * 1e-3
        waves = p["grid"] * 4
        rate = valu / t / SIMDS if valu else 0.0
        print(f"{p['kernel']:8s} {p['waves_per_simd']:6d} {p['ms_min']:8.4f} {valu:12.4g} {valu / waves / max(p['batches_per_wave'], 1):15.1f} "
              f"{rate / 1e9:13.3f} {CLK / rate if rate else 0:15.2f} "
              f"{(c.get('SQ_ACTIVE_INST_VALU', 0) / valu if valu else 0):17.3f} {(c.get('SQ_INSTS_SALU', 0) / valu if valu else 0):9.3f} "
              f"{(c.get('SQ_INSTS_LDS', 0) / valu if valu else 0):8.3f}")


if __name__ == "__main__":
    main()
