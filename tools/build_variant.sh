#!/bin/bash
# Build a VARIANT of libgsr.so for same-box A/B runs (GSR_LIB=..., tools/gpu_r5.sh ab): the working tree's csrc/ + include/ are
# copied to _exp/src_<name>/, edited there by the commands on stdin (cwd = that copy: `sed -i ... dreamgaussian_amd/csrc/x.hip`,
# `git -C $ROOT show <ref>:<path> > <path>`, ...) and compiled with build.py's flags into _exp/libgsr_<name>.so.
# The product tree is never touched; nothing under _exp/ is committed.
#   tools/build_variant.sh oldscan <<< 'git -C $ROOT show HEAD:dreamgaussian_amd/csrc/gsr_binning.hip > dreamgaussian_amd/csrc/gsr_binning.hip'
set -e
name=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export ROOT
dst=$ROOT/_exp/src_$name
rm -rf "$dst"; mkdir -p "$dst/dreamgaussian_amd" "$dst/include"
cp -r "$ROOT/dreamgaussian_amd/csrc" "$dst/dreamgaussian_amd/csrc"
cp "$ROOT/include/gsr.h" "$dst/include/gsr.h"
(cd "$dst" && bash -e /dev/stdin)
flags=$(python -c "import sys; sys.path.insert(0, '$ROOT'); from dreamgaussian_amd import build as b; print(' '.join(b.FLAGS))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 $flags "$dst/dreamgaussian_amd/csrc/gsr_api.hip" -o "$ROOT/_exp/libgsr_$name.so"
echo "_exp/libgsr_$name.so"
