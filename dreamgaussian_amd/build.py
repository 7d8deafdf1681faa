"""Build libgsr.so (the C-ABI library of include/gsr.h) for gfx950 with hipcc.

One translation unit (csrc/gsr_api.hip includes the kernel files); cross-compiles without a
GPU. The .so is written next to this package so that it travels with the source tree.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgsr.so")
STAMP = os.path.join(HERE, ".libgsr.stamp")
ARCH = "gfx950"
# -fno-slp-vectorize: hipcc packs adjacent scalar fp32 ops into v_pk_* (+ v_mov shuffles); on gfx950 a v_pk_fma_f32
# issues in 4.4 cycles against 2.2 for v_fma_f32 (profiles/r02_ubench_lds_valu.txt): no gain, only the shuffles
FLAGS = ["-O3", "-std=c++17", "-munsafe-fp-atomics", "-fno-slp-vectorize", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]


def _sources():
    root = os.path.dirname(HERE)
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))
             if f.endswith((".hip", ".h"))]
    files.append(os.path.join(root, "include", "gsr.h"))
    return files


def _digest() -> str:
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + [ARCH]).encode())
    for f in _sources():
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def kernel_digest() -> str:
    """Digest of the DEVICE code only (kernel files and gsr_device.h, build flags): what hardware-counter profiles of the
    kernels belong to. Host-side changes in gsr_api.hip / include/gsr.h leave it unchanged (profiles/pmc_traffic.json)."""
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + [ARCH]).encode())
    for f in _sources():
        if os.path.basename(f) in ("gsr_api.hip", "gsr.h"):
            continue
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libgsr.so)")


def build(force: bool = False, verbose: bool = True) -> str:
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == dig:
                return LIB
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", *FLAGS,
           os.path.join(CSRC, "gsr_api.hip"), "-o", LIB]
    if verbose:
        print("[dreamgaussian_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


# ---- the torch binding (csrc/gsr_torch.cpp -> _gsr_torch.so): a pybind11 / C++ autograd module compiled by g++ against torch's
# headers (no device code: it only calls the C ABI). Optional at run time -- rasterizer.py falls back to its ctypes path without it.
BIND_SRC = os.path.join(CSRC, "gsr_torch.cpp")
BIND_LIB = os.path.join(HERE, "_gsr_torch.so")
BIND_STAMP = os.path.join(HERE, ".gsr_torch.stamp")


def _bind_cmd():
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    inc = ce.include_paths() + [sysconfig.get_paths()["include"], "/opt/rocm/include"]
    return (["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_gsr_torch",
             "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-deprecated-declarations",
             BIND_SRC, "-o", BIND_LIB] + [f"-I{i}" for i in inc] + [f"-L{p}" for p in ce.library_paths()]
            + ["-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10_hip", "-ltorch_hip", "-ldl"])


def _bind_digest() -> str:
    import torch
    h = hashlib.sha256()
    h.update(torch.__version__.encode())
    for f in (BIND_SRC, os.path.join(os.path.dirname(HERE), "include", "gsr.h")):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build_binding(force: bool = False, verbose: bool = True) -> str:
    dig = _bind_digest()
    if not force and os.path.exists(BIND_LIB) and os.path.exists(BIND_STAMP):
        with open(BIND_STAMP) as fh:
            if fh.read().strip() == dig:
                return BIND_LIB
    cmd = _bind_cmd()
    if verbose:
        print("[dreamgaussian_amd.build]", " ".join(cmd[:14]), "...", flush=True)
    subprocess.run(cmd, check=True)
    with open(BIND_STAMP, "w") as fh:
        fh.write(dig)
    return BIND_LIB


def is_current() -> bool:
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as fh:
        return fh.read().strip() == _digest()


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
    build_binding(force="--force" in sys.argv)
    print(BIND_LIB)
