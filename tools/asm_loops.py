#!/usr/bin/env python
"""Instruction mix per basic block of one kernel in a hipcc -S dump (finding the hot loops' cost).
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -S --cuda-device-only csrc/gsr_api.hip -o /tmp/gsr.s
  python tools/asm_loops.py /tmp/gsr.s gsr_render_bwd_q2 [min_instructions]"""
import re, sys
from collections import Counter
path, name = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^\w*%s\w*:" % re.escape(name), l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blk, blocks = None, []
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", t):
            blk = [t.rstrip(":").split(":")[0], Counter(), 0]
            blocks.append(blk)
        continue
    if blk is None:
        blk = ["entry", Counter(), 0]
        blocks.append(blk)
    op = t.split()[0]
    c = blk[1]
    blk[2] += 1
    if op.startswith("v_"):
        c["valu"] += 1
        if re.match(r"v_(exp|rcp|log|sqrt|rsq|sin|cos)", op): c["trans"] += 1
        if "dpp" in t or "permlane" in op or "readlane" in op or "readfirstlane" in op: c["xlane"] += 1
        if op.startswith("v_pk_"): c["pk"] += 1
        if op.startswith("v_cmp"): c["cmp"] += 1
        if op.startswith("v_cndmask"): c["cnd"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
        if op.startswith("s_waitcnt"): c["wait"] += 1
        if op.startswith("s_cbranch") or op.startswith("s_branch"): c["br"] += 1
    elif op.startswith("ds_"): c["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
    m = re.search(r"(s_cbranch\w*|s_branch)\s+(\.LBB\d+_\d+)", t)
    if m:
        blk.append(m.group(2))
print(f"{name}: {end - start} lines")
for b in blocks:
    if b[2] >= minn:
        tg = [x for x in b[3:]]
        print(f"{b[0]:12s} n={b[2]:4d} {dict(b[1])}  -> {tg}")
