#!/bin/bash
# One gpurun call: GPU tests, smoke, bench lines, rocprofv3 kernel trace. Outputs -> gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench 1M"; timeout 600 python bench.py --steps 10 --warmup 3 2> gpurun_out/bench_1M.err | tee gpurun_out/bench_1M.json
tail -5 gpurun_out/bench_1M.err
for wl in 100k-800-sh3 250k-512-sh0 5k-256-sh0; do
  echo "== bench $wl"; timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --cpu-budget 0 2> gpurun_out/bench_$wl.err | tee gpurun_out/bench_$wl.json
done
echo "== bench 1M trained"; timeout 300 python bench.py --kind trained --steps 10 --warmup 3 --cpu-budget 0 2> gpurun_out/bench_1M_trained.err | tee gpurun_out/bench_1M_trained.json
echo "== rocprofv3 kernel trace (1M)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_1M -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-budget 0 --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_1M.log 2>&1)
tail -3 gpurun_out/prof_1M.log
find gpurun_out/prof_1M -name "*stats*" | head
