#!/bin/bash
# Build libgsr.so from another git revision into tools/<name>.so for same-box A/B runs:
#   tools/build_branch_lib.sh next/round2-prototypes proto      -> tools/libgsr_proto.so
#   gpurun -- 'GSR_LIB=tools/libgsr_proto.so GSR_BWD=quad python -m pytest tests/test_parity_gpu.py -x -q'
# (built .so files are git-ignored but travel with the gpurun snapshot; delete them afterwards)
set -e
REV=${1:?revision}; NAME=${2:?name}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" dreamgaussian_amd/csrc include | tar -x -C "$TMP"
FLAGS=$(python - <<PY
import sys; sys.path.insert(0, "$ROOT")
from dreamgaussian_amd import build as b
print(f"--offload-arch={b.ARCH} " + " ".join(b.FLAGS))
PY
)
/opt/rocm/bin/hipcc $FLAGS "$TMP/dreamgaussian_amd/csrc/gsr_api.hip" -o "$ROOT/tools/libgsr_$NAME.so"
rm -rf "$TMP"
echo "$ROOT/tools/libgsr_$NAME.so"
