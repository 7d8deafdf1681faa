#!/bin/bash
# gpurun driver (round 6): sections chosen by arguments; shared sections live in tools/gpu_r5.sh. Outputs -> gpurun_out/.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_r6.sh hostpath ab'
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], json.dumps(d['kernels_ms_per_step']))"; }
for sec in "$@"; do
case $sec in
hostpath)
  echo "== pytest: host path (ABI 6) + K6c"
  timeout ${QUICK_TIMEOUT:-600} python -m pytest tests/test_hostpath_gpu.py tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider --tb=short -rf \
      -k "${NEW_K:-hostpath or k6_compact or speculative or gradient_arrays_cleared or backward_without_forward_stats or forward_backward_match_oracle}" > gpurun_out/pytest_hostpath.log 2>&1
  grep -a "passed\|failed\|FAILED\|Error\|assert" gpurun_out/pytest_hostpath.log | cut -c1-300 | tail -20;;
ab)
  # same-box A/B of library builds: AB_LIBS="none _exp/libgsr_x.so", AB_WL="1M-800-sh3 1M-800-sh3:trained", AB_REP (default 2)
  for rep in $(seq 1 ${AB_REP:-2}); do for wl in ${AB_WL:-1M-800-sh3}; do
    kind=blob; [ "${wl#*:}" != "$wl" ] && kind=${wl#*:}
    for l in $AB_LIBS; do
      echo "== [$l] $wl"
      if [ $l = none ]; then python bench.py --workload ${wl%%:*} --kind $kind --cpu-budget 0 --steps 60 --warmup 10 $BENCH_ARGS 2>>gpurun_out/ab_err.log | tee -a gpurun_out/ab.jsonl | line
      else GSR_LIB=$R/$l python bench.py --workload ${wl%%:*} --kind $kind --cpu-budget 0 --steps 60 --warmup 10 $BENCH_ARGS 2>>gpurun_out/ab_err.log | tee -a gpurun_out/ab.jsonl | line; fi
    done
  done; done;;
order)
  rm -f gpurun_out/bench_order.jsonl
  for wl in ${ORDER_WL:-1M-800-sh3 1M-800-sh3:trained}; do
    kind=blob; [ "${wl#*:}" != "$wl" ] && kind=${wl#*:}
    for o in given morton; do
      echo "== bench $wl --order $o"; timeout 300 python bench.py --workload ${wl%%:*} --kind $kind --order $o --cpu-budget 0 --steps 60 --warmup 10 2>> gpurun_out/bench_order.err | tee -a gpurun_out/bench_order.jsonl | line
    done
  done;;
floor5k)
  # 5k / 256^2: ten launches of the bench line, blocking and with the forward that does not wait for its counters; the host phases
  python tools/host_overhead.py 2>&1 | grep -a "ms/step\|floor" | tee gpurun_out/host_floor_5k.txt
  for mode in "--binding cpp" "--binding ctypes" "--binding cpp --async-forward" "--binding ctypes --async-forward"; do
    echo "== 5k-256-sh0 $mode" | tee -a gpurun_out/host_floor_5k.txt
    for i in 1 2 3 4 5 6 7 8 9 10; do python bench.py --workload 5k-256-sh0 --cpu-budget 0 --no-roofline --steps 200 --warmup 20 $mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('5k-256-sh0 ms_per_step', d['ms_per_step'])"; done | tee -a gpurun_out/host_floor_5k.txt
  done
  python tools/host_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/host_phases_5k.txt;;
place)
  GSR_LIB=$R/_exp/libgsr_place.so python tools/placement_report.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/placement_1M.txt | head -40;;
reduce)
  # the gradient exchange of the "local" SDS step through a ONE-rank RCCL group: dense all-reduce / sharded Adam / live rows, at
  # BASELINE configs[3] and at 1M / SH 3 (62 MB of gradients)
  rm -f gpurun_out/sds_reduce.jsonl
  for wl in 250k-512-sh0 1M-800-sh3; do for m in "--reduce dense" "--reduce dense --sds-adam" "--reduce sharded" "--reduce live" "--reduce live --sds-adam"; do
    echo "== sds local $wl $m"; timeout 300 python bench.py --step sds --sds-mode local --sds-workload $wl $m --force-collectives --cpu-budget 0 --steps 40 --warmup 10 2>>gpurun_out/sds_reduce.err | grep -a "^{" | tee -a gpurun_out/sds_reduce.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['config'].get('reduce'), d['config'].get('live_rows'))"
  done; done;;
group)
  # K1's group reservation + the four-slice scatter: the tests, then a same-process-order A/B of the hook (k1_group = 1 / 4)
  timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider --tb=short -rf -k "${GROUP_K:-group_reservation or scatter_with_several or speculative or heavy_tile}" > gpurun_out/pytest_group.log 2>&1
  grep -a "passed\|failed\|FAILED\|Error\|assert" gpurun_out/pytest_group.log | cut -c1-300 | tail -20
  for rep in 1 2; do for wl in ${GROUP_WL:-1M-800-sh3 1M-800-sh3:trained 250k-512-sh0 100k-800-sh3}; do
    kind=blob; [ "${wl#*:}" != "$wl" ] && kind=${wl#*:}
    for g in 1 4; do
      echo "== k1_group=$g $wl"; timeout 300 python bench.py --workload ${wl%%:*} --kind $kind --hook k1_group=$g --cpu-budget 0 --steps 60 --warmup 10 2>>gpurun_out/group_err.log | tee -a gpurun_out/group_ab.jsonl | line
    done
  done; done;;
*) bash tools/gpu_r5.sh $sec;;
esac
done
