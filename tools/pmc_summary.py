#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (csv output) per kernel: mean counter value per launch.

  python tools/pmc_summary.py gpurun_out [--json profiles/pmc_traffic.json --key 1M-800-sh3/blob]

Looks for */*counter_collection.csv under the given directory (one sub-directory per pass, as
tools/gpu_r3.sh writes them). HBM traffic per launch follows MI355X_MICROARCH.md's HBM
section: FETCH_SIZE / WRITE_SIZE are in KiB-equivalents of 1024 B... (rocprofv3 reports
FETCH_SIZE and WRITE_SIZE in kilobytes); on gfx950 FETCH_SIZE counts 64 B per 128-B request for
wide coalesced reads, so the read side is reported both raw and doubled (the correction the
guide prescribes for 16 B/lane streaming reads; gathers of <=64 B are not doubled by hardware,
so the truth lies between the two)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0]
    n = n.replace("void ", "")
    for fam in ("gsr_tile_sort", "gsr_render_fwd_combine", "gsr_render_fwd_seg", "gsr_render_fwd_serial", "gsr_render_fwd_fix", "gsr_render_bwd"):
        if fam in n and "v0" not in n:
            return fam
    return n.strip()


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
    files = sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True))
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))     # kernel -> counter -> [sum, dispatches]
    for f in files:
        with open(f) as fh:
            rd = csv.DictReader(fh)
            per_dispatch = defaultdict(float)
            names = {}
            for row in rd:
                key = (row.get("Dispatch_Id"), row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                names[row.get("Dispatch_Id")] = row["Kernel_Name"]
            for (did, cname), v in per_dispatch.items():
                k = short(names[did])
                if not k.startswith("gsr_"):
                    continue
                e = acc[k][cname]
                e[0] += v
                e[1] += 1
    out = {}
    for k in sorted(acc):
        row = {c: acc[k][c][0] / max(acc[k][c][1], 1) for c in sorted(acc[k])}
        out[k] = row
        print(k, {c: (round(v, 1) if v < 1e6 else f"{v:.4g}") for c, v in row.items()})
    if "--json" in sys.argv:
        path = sys.argv[sys.argv.index("--json") + 1]
        key = sys.argv[sys.argv.index("--key") + 1]
        traffic = {}
        for k, row in out.items():
            if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
                fam = k.replace("gsr_", "")
                rd_raw, wr = row["FETCH_SIZE"] * 1024.0, row["WRITE_SIZE"] * 1024.0
                # calibrated on this repository's access patterns (tools/fetch_calib.hip, profiles/r02_fetch_calibration.txt):
                # coalesced reads (4 or 16 B per lane) are counted at 1/2; random 64-byte record gathers at >= 1 (the counter
                # tallies 64-B requests and a record read as 3-4 loads draws 1.8-2.1 of them); writes at 1 (streaming) and at
                # their 32-byte sectors (scattered 8-byte stores)
                # (gsr_preprocess_bwd_compact gathers the rows of the live Gaussians: 4..192-byte accesses at random rows)
                gather = any(t in fam for t in ("render_fwd", "render_bwd", "preprocess_bwd_compact"))
                f = 1.0 if gather else 2.0
                traffic[fam] = {"read_bytes_raw": rd_raw, "read_bytes_x2": 2 * rd_raw, "write_bytes": wr, "read_factor": f,
                                "hbm_bytes": f * rd_raw + wr}
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from dreamgaussian_amd import build as _build
        dig = _build.kernel_digest()
        doc = {}
        if os.path.exists(path):
            doc = json.load(open(path))
            if doc.get("source_digest") != dig:       # counters of other kernel sources: start over
                doc = {}
        doc["source_digest"] = dig                    # bench.py reports `traffic` only when this matches its own build
        doc["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), per launch; read side x2 for the streaming "
                       "kernels (gfx950 counts 64 B per 128-B request on coalesced reads), x1 for the record-gathering "
                       "compositing kernels (calibration: profiles/r02_fetch_calibration.txt)")
        doc[key] = traffic
        json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
        print("wrote", path)


if __name__ == "__main__":
    main()
