#!/usr/bin/env python
"""Loads that are waited for on the spot, per kernel of libgsr.so's device code (rule 1).

Cross-compiles csrc/gsr_api.hip to gfx950 assembly (no GPU needed, ~1 min) and reports, for every kernel, each
`global_load*` (or returning global atomic) that is followed within WINDOW instructions by an `s_waitcnt vmcnt(n)` with
n <= the number of vector-memory operations issued in between -- a wait that covers the load just issued, i.e. a memory
round trip taken in series instead of overlapped. Dependent pairs (list entry -> record) and returning atomics show up by
construction; what must NOT show up are the loads a kernel was written to keep in flight.

    python tools/wait_scan.py [kernel-substring] [--asm file.s] [--keep file.s]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WINDOW = 9


def device_asm(out_path: str) -> str:
    sys.path.insert(0, ROOT)
    from dreamgaussian_amd import build as B
    flags = [f for f in B.FLAGS if f not in ("-shared", "-Wall", "-Wno-unused-function")]
    cmd = ["/opt/rocm/bin/hipcc", f"--offload-arch={B.ARCH}", *flags, "-S", "--cuda-device-only",
           os.path.join(B.CSRC, "gsr_api.hip"), "-I", os.path.join(ROOT, "include"), "-o", out_path]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc -S failed:\n" + r.stderr[-2000:])
    return out_path


def scan(asm_path: str):
    """{kernel name: [(load line, load text, wait line, wait text), ...]} with line numbers relative to the kernel's label."""
    lines = open(asm_path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^(_Z\w+|gsr_\w+):", l)]
    out = {}
    for st, name in starts:
        end = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
        ins = [(i, lines[i].strip()) for i in range(st, end)
               if lines[i].startswith("\t") and not lines[i].strip().startswith((".", ";"))]
        found = []
        for k, (i, l) in enumerate(ins):
            if l.startswith("global_load") or (l.startswith("global_atomic") and " sc0" in l):
                younger = 0
                for j, m in ins[k + 1:k + 1 + WINDOW]:
                    if m.startswith(("global_", "buffer_")):
                        younger += 1
                    w = re.search(r"vmcnt\((\d+)\)", m)
                    if w and int(w.group(1)) <= younger:
                        found.append((i - st, l, j - st, m))
                        break
        out[name] = found
    return out


def demangle(n: str) -> str:
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception:
        return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("filter", nargs="?", default="")
    ap.add_argument("--asm", default=None, help="scan this assembly file instead of compiling")
    ap.add_argument("--keep", default=None, help="keep the generated assembly here")
    a = ap.parse_args()
    path = a.asm
    if path is None:
        path = a.keep or os.path.join(tempfile.mkdtemp(prefix="gsr_asm_"), "gsr.s")
        device_asm(path)
    for name, found in scan(path).items():
        nice = demangle(name)
        if a.filter and a.filter not in nice:
            continue
        print(f"== {nice}: {len(found)}")
        for f in found:
            print(f"    +{f[0]:<5d} {f[1][:70]:70s} -> +{f[2]} {f[3]}")


if __name__ == "__main__":
    main()
