#!/usr/bin/env python
"""Per-kernel register / LDS / occupancy table of libgsr.so's device code (hipcc
-Rpass-analysis=kernel-resource-usage; cross-compiles, no GPU needed).
  python tools/kernel_resources.py [filter-substring]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "dreamgaussian_amd", "csrc", "gsr_api.hip")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fno-slp-vectorize",
                      "-fPIC", "-shared", "-Rpass-analysis=kernel-resource-usage", src, "-o", "/tmp/_kr.so"],
                     capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: (?:[^:]+:\d+:\d+: )?\s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" [")[0]] = v
flt = sys.argv[1] if len(sys.argv) > 1 else ""
def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception:
        return n
print(f"{'kernel':58s} {'VGPR':>5s} {'SGPR':>5s} {'LDS':>7s} {'scr':>4s} {'occ':>4s} {'vspill':>6s} {'sspill':>6s}")
for r in rows:
    n = demangle(r["name"])
    if flt and flt not in n:
        continue
    print(f"{n[:58]:58s} {r.get('VGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} {r.get('LDS Size','?'):>7s} {r.get('ScratchSize','?'):>4s} "
          f"{r.get('Occupancy','?'):>4s} {r.get('VGPRs Spill','?'):>6s} {r.get('SGPRs Spill','?'):>6s}")
