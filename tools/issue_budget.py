#!/usr/bin/env python
"""Prices the vector instructions of named basic blocks of one kernel by ISSUE CLASS, with the per-class costs measured on this part
(tools/valu_rates.hip -> profiles/r04_valu_rates.txt, 4 waves per SIMD, cycles of a SIMD's issue per wave64 instruction):

  plain      VOP1/2/3 arithmetic on VGPRs, moves, integer ops                           2.35
  second     v_max/min/med3, v_cmp*, v_cndmask, v_bfi, v_pk_*, an SGPR or VCC source    2.35 alone, 4.5 right behind another of its class
  dpp        any DPP / permlane / readlane                                              10.4 alone ("4 fmac + 1 dpp": 3.93 avg), 4.4 behind another
  trans      v_exp / v_rcp / v_log / v_sqrt / v_rsq / v_sin / v_cos                     16 alone ("4 fmac + 1 exp": 5.03 avg), 8.5 behind another

LDS, scalar, wait and branch instructions issue on other ports and are counted, not priced. "behind another" is decided on the
sequence of VECTOR-ALU instructions of the block (scalar / LDS instructions in between do not separate two of them).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize -S --cuda-device-only dreamgaussian_amd/csrc/gsr_api.hip -o /tmp/gsr.s
  python tools/issue_budget.py /tmp/gsr.s gsr_render_bwd_q2 .LBB19_61 .LBB19_63 ...      (no labels: every block of >= 40 instructions)
"""
import re, sys

COST = {"plain": (2.35, 2.35), "second": (2.35, 4.5), "dpp": (10.4, 4.4), "trans": (16.0, 8.5)}


def classify(t):
    op = t.split()[0]
    if not op.startswith("v_"):
        if op.startswith("ds_"): return "lds"
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
        if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
        if op.startswith(("s_cbranch", "s_branch")): return "branch"
        return "salu"
    if "dpp" in t or "permlane" in op or "readlane" in op or "readfirstlane" in op: return "dpp"
    if re.match(r"v_(exp|rcp|log|sqrt|rsq|sin|cos)_", op): return "trans"
    if re.match(r"v_(max|min|med3|cmp|cmpx|cndmask|bfi|pk_)", op): return "second"
    ops = t.split(None, 1)[1] if " " in t else ""
    srcs = ops.split(",")[1:]                              # everything behind the destination
    if any(re.match(r"\s*(-|\|)?(s\d+|s\[|vcc|exec|ttmp)", s) for s in srcs): return "second"
    return "plain"


def main():
    path, name, want = sys.argv[1], sys.argv[2], sys.argv[3:]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^\w*%s\w*:" % re.escape(name), l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur = {}, None
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            cur = blocks.setdefault(m.group(1), [])
            continue
        if not t or t.startswith(";") or t.startswith(".") or cur is None:
            continue
        cur.append(t.split(";")[0].strip())
    if not want:
        want = [b for b, ins in blocks.items() if len(ins) >= 40]
    tot_c = tot_n = 0.0
    tot = {}
    print(f"{name}: blocks {' '.join(want)}")
    print(f"{'block':12s} {'valu':>5s} {'cycles':>8s} {'cyc/instr':>9s}   plain second(behind) dpp(alone) trans(alone)  lds salu wait")
    for b in want:
        ins = blocks[b]
        prev, cyc, n = None, 0.0, 0
        cnt = {}
        for t in ins:
            c = classify(t)
            cnt[c] = cnt.get(c, 0) + 1
            if c not in COST:
                continue
            alone, behind = COST[c]
            follows = prev == c and c != "plain"
            cyc += behind if follows else alone
            if c == "second" and follows: cnt["second_behind"] = cnt.get("second_behind", 0) + 1
            if c in ("dpp", "trans") and not follows: cnt[c + "_alone"] = cnt.get(c + "_alone", 0) + 1
            prev, n = c, n + 1
        for k, v in cnt.items(): tot[k] = tot.get(k, 0) + v
        tot_c, tot_n = tot_c + cyc, tot_n + n
        g = cnt.get
        print(f"{b:12s} {n:5d} {cyc:8.0f} {cyc / max(n, 1):9.2f}   {g('plain', 0):5d} {g('second', 0):4d}({g('second_behind', 0):3d})   {g('dpp', 0):3d}({g('dpp_alone', 0):2d})   {g('trans', 0):3d}({g('trans_alone', 0):2d})   "
              f"{g('lds', 0):4d} {g('salu', 0):4d} {g('wait', 0):4d}")
    g = tot.get
    print(f"{'total':12s} {int(tot_n):5d} {tot_c:8.0f} {tot_c / max(tot_n, 1):9.2f}   {g('plain', 0):5d} {g('second', 0):4d}({g('second_behind', 0):3d})   {g('dpp', 0):3d}({g('dpp_alone', 0):2d})   "
          f"{g('trans', 0):3d}({g('trans_alone', 0):2d})   {g('lds', 0):4d} {g('salu', 0):4d} {g('wait', 0):4d}")


if __name__ == "__main__":
    main()
