#!/bin/bash
# Dev container only: copy the reference files tools/run_stage1.py drives (UNMODIFIED) into ./_ref_stage so that
# they travel to the GPU box with the gpurun snapshot. _ref_stage/ is git-ignored: reference sources never enter
# this repository's history. It STAYS in the work tree between GPU calls: the GPU suite's test_reference_trainer_through_libgsr
# (tests/test_parity_gpu.py) runs the reference's trainer from it on whichever box the work tree is sent to (`rm -rf _ref_stage`
# to drop it: the test then skips loudly).
set -e
SRC=${1:-/root/reference}
DST="$(cd "$(dirname "$0")/.." && pwd)/_ref_stage"
mkdir -p "$DST/configs" "$DST/data"
for f in main.py gs_renderer.py sh_utils.py cam_utils.py grid_put.py; do cp "$SRC/$f" "$DST/$f"; done
cp "$SRC/configs/image.yaml" "$DST/configs/"
cp "$SRC/data/catstatue_rgba.png" "$DST/data/"
echo "staged $(ls "$DST" | tr '\n' ' ')-> $DST"
