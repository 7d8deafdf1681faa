"""Attribute one steady-state step of `bench.py --step sds --force-collectives` from a rocprofv3 kernel + HIP API trace
(tools/gpu_r5.sh sdstrace): the kernels of the step on a time line (which are RCCL's / torch's / this library's, the idle time in
front of each) and the host's time inside the collective calls."""
import csv
import glob
import os
import sys


def rows(d, suffix):
    f = glob.glob(os.path.join(d, "**", f"*{suffix}"), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def main(d):
    ks = rows(d, "kernel_trace.csv")
    ks.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a step starts at gsr_preprocess_fwd; take the steps of the second half of the run
    starts = [i for i, r in enumerate(ks) if r["Kernel_Name"].startswith("void gsr_preprocess_fwd") or r["Kernel_Name"].startswith("gsr_preprocess_fwd")]
    if len(starts) < 6:
        print("too few steps in the trace", len(starts)); return
    sel = starts[len(starts) // 2:-1]
    agg = {}
    order = []
    span = 0.0
    for a, b in zip(sel[:-1], sel[1:]):
        t0 = int(ks[a]["Start_Timestamp"]); prev_end = t0
        span += (int(ks[b]["Start_Timestamp"]) - t0) / 1e3
        for j, r in enumerate(ks[a:b]):
            name = r["Kernel_Name"].split("(")[0][:60]
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            key = (j, name)
            if key not in agg:
                agg[key] = [0.0, 0.0, 0]; order.append(key)
            agg[key][0] += (e - s) / 1e3; agg[key][1] += max(0, s - prev_end) / 1e3; agg[key][2] += 1
            prev_end = max(prev_end, e)
    n = len(sel) - 1
    print(f"# {n} steps, {span / n:.1f} us per step (start of K1 to start of the next K1)")
    print(f"# {'kernel':60s} {'us':>8s} {'idle in front':>14s}")
    tot_k = tot_gap = 0.0
    for key in order:
        dur, gap, c = agg[key]
        if c < n // 2:
            continue
        print(f"  {key[1]:60s} {dur / c:8.1f} {gap / c:14.1f}")
        tot_k += dur / c; tot_gap += gap / c
    print(f"# kernels {tot_k:.1f} us, idle between them {tot_gap:.1f} us")
    api = rows(d, "hip_api_trace.csv")
    if api:
        t = {}
        for r in api:
            fn = r.get("Function", r.get("Name", "?"))
            t.setdefault(fn, [0.0, 0])
            t[fn][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; t[fn][1] += 1
        print("# host time inside HIP calls (whole run): call, total ms, count")
        for fn, (us, c) in sorted(t.items(), key=lambda kv: -kv[1][0])[:8]:
            print(f"  {fn:40s} {us / 1e3:9.2f} {c:7d}")


if __name__ == "__main__":
    main(sys.argv[1])
