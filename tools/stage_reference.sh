#!/bin/bash
# Dev container only: copy the reference files tools/run_stage1.py drives (UNMODIFIED) into ./_ref_stage so that
# they travel to the GPU box with the gpurun snapshot. _ref_stage/ is git-ignored: reference sources never enter
# this repository's history. Remove it again after the gpurun call (rm -rf _ref_stage): it is scratch, not part of the tree.
set -e
SRC=${1:-/root/reference}
DST="$(cd "$(dirname "$0")/.." && pwd)/_ref_stage"
mkdir -p "$DST/configs" "$DST/data"
for f in main.py gs_renderer.py sh_utils.py cam_utils.py grid_put.py; do cp "$SRC/$f" "$DST/$f"; done
cp "$SRC/configs/image.yaml" "$DST/configs/"
cp "$SRC/data/catstatue_rgba.png" "$DST/data/"
echo "staged $(ls "$DST" | tr '\n' ' ')-> $DST"
