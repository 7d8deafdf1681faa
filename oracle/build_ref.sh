#!/bin/bash
# TEST INFRASTRUCTURE ONLY. Builds oracle/_ref/libsimple_knn_ref.so: the REFERENCE's own distCUDA2 (simple-knn/simple_knn.cu, the
# code behind `from simple_knn._C import distCUDA2`, gs_renderer.py:14,341) for gfx950, straight from the sources where they lie under
# /root/reference -- hipify-perl's translation (plus three fix-ups it does not make: two CUDA-only includes, an empty include, the
# spaced `<< < ... >> >` launch syntax, FLT_MAX without <cfloat>) and a five-line extern "C" wrapper. Nothing of the reference is copied
# into this repository: the translated file and the library land in oracle/_ref/ (git-ignored; they travel to the GPU box with the
# work tree like any built .so). Only tests/test_knn_gpu.py loads it, as the CHECKER of gsr_dist2; the product never does.
#   bash oracle/build_ref.sh [/root/reference]
set -e
REF=${1:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
[ -f "$REF/simple-knn/simple_knn.cu" ] || { echo "[oracle/build_ref] no reference at $REF: skipped"; exit 0; }
mkdir -p "$OUT"
/opt/rocm/bin/hipify-perl "$REF/simple-knn/simple_knn.cu" > "$OUT/simple_knn.hip" 2>/dev/null
sed -i '/#include ""/d; /cub\/device\/device_radix_sort.cuh/d; /cooperative_groups\/reduce.h/d; s/<< </<<</g; s/>> >/>>>/g' "$OUT/simple_knn.hip"
cat > "$OUT/wrap.hip" <<'W'
#include "hip/hip_runtime.h"
#include "simple_knn.h"
// points [P,3] and out [P] are device pointers; the reference's SimpleKNN::knn (simple_knn.cu:185-221) as spatial.cu:15-26 calls it
extern "C" int ref_dist2(int P, const float* points_dev, float* out_dev) {
    SimpleKNN::knn(P, (float3*)points_dev, out_dev);
    return (int)hipDeviceSynchronize();
}
W
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -fPIC -shared -w -include cfloat -I"$REF/simple-knn" "$OUT/simple_knn.hip" "$OUT/wrap.hip" -o "$OUT/libsimple_knn_ref.so"
echo "[oracle/build_ref] $OUT/libsimple_knn_ref.so"
