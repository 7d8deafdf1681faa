#!/usr/bin/env python
"""Per-step timeline of a rocprofv3 --kernel-trace run of bench.py: kernel durations and the idle time in front of each kernel
(median over the steps), i.e. what the per-kernel hipEvent sums do not show.   python tools/trace_gaps.py <kernel_trace.csv>"""
import csv
import statistics as st
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "")[-48:]
names = [short(r["Kernel_Name"]) for r in rows]
starts = [i for i, n in enumerate(names) if "preprocess_fwd" in n]
steps = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
seq = None
acc = {}
tot, busy = [], []
for lo, hi in steps[2:]:
    ks = [names[i] for i in range(lo, hi)]
    if seq is None:
        seq = ks
    if ks != seq:
        continue
    prev_end = None
    b = 0
    for j, i in enumerate(range(lo, hi)):
        s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        acc.setdefault(j, []).append(((e - s) / 1e3, gap))
        prev_end = max(e, prev_end or 0)
        b += (e - s) / 1e3
    tot.append((int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3)
    busy.append(b)
print(f"{len(tot)} steps; step (start of K1 to start of the next K1) median {st.median(tot):.1f} us, kernels {st.median(busy):.1f} us")
for j, k in enumerate(seq):
    d = st.median(x[0] for x in acc[j]); g = st.median(x[1] for x in acc[j])
    print(f"  {k:48s} {d:8.1f} us   idle in front {g:7.1f} us")
