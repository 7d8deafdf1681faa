#!/bin/bash
# Round-3 gpurun driver: sections chosen by arguments. Outputs -> gpurun_out/.
#   quick | tests | bench | benchall | morton | libs | shifts | prof | pmc | stage1 | views | sds | rccl
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
benchline() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms shift', d['config'].get('seg_shift'), d.get('kernels_ms_per_step_raw') or d['kernels_ms_per_step'])"; }
for s in "$@"; do case $s in
quick)
  echo "== pytest quick (parity without the big fp64 cases, fuzz, views)"
  timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_views_gpu.py tests/test_optim_gpu.py tests/test_densify_gpu.py -m gpu -q -p no:cacheprovider --tb=short -rf -k "not baseline_config and not cfg3 and not full_size" > gpurun_out/pytest_quick.log 2>&1
  grep -a "passed\|failed\|FAILED\|Error\|assert" gpurun_out/pytest_quick.log | cut -c1-400 | tail -40;;
tests)
  echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider --tb=short -rf > gpurun_out/pytest_gpu.log 2>&1
  grep -a "fragile:\|observed:\|passed\|failed\|FAILED\|Error" gpurun_out/pytest_gpu.log | cut -c1-700 | tail -40;;
bench)
  echo "== bench 1M"; timeout 900 python bench.py $BENCH_ARGS 2> gpurun_out/bench_1M.err | tee gpurun_out/bench_1M.json
  tail -3 gpurun_out/bench_1M.err;;
benchall)
  for wl in 1M-800-sh3 100k-800-sh3 250k-512-sh0 5k-256-sh0; do
    echo "== bench $wl"; timeout 300 python bench.py --workload $wl --cpu-budget 0 2> gpurun_out/bench_$wl.err | tee gpurun_out/bench_$wl.json | benchline
  done
  echo "== bench 1M trained"; timeout 300 python bench.py --kind trained --cpu-budget 0 2> gpurun_out/bench_1M_trained.err | tee gpurun_out/bench_1M_trained.json | benchline;;
morton)
  # the same scenes with the rows permuted along a Z-order curve (dreamgaussian_amd.reorder_gaussians)
  for wl in 1M-800-sh3 100k-800-sh3 250k-512-sh0; do
    echo "== bench $wl --order morton"; timeout 300 python bench.py --workload $wl --order morton --cpu-budget 0 2> gpurun_out/bench_${wl}_morton.err | tee gpurun_out/bench_${wl}_morton.json | benchline
  done
  echo "== bench 1M trained --order morton"; timeout 300 python bench.py --kind trained --order morton --cpu-budget 0 2> gpurun_out/bench_1M_trained_morton.err | tee gpurun_out/bench_1M_trained_morton.json | benchline;;
libs)
  # same-box A/B of alternative builds of the library (GSR_LIB=<path>, same ABI): LIBS="path1 path2", WLS="..."
  for l in ${LIBS}; do
    for wl in ${WLS:-1M-800-sh3 100k-800-sh3}; do
      echo "== [$l] $wl"; GSR_LIB=$R/$l timeout 300 python bench.py --cpu-budget 0 --workload $wl $BENCH_ARGS 2>>gpurun_out/ab_err.log | benchline
    done
  done;;
shifts)
  # same-box A/B of the env-selectable variants of the segment forward
  for v in ${VARIANTS:-"GSR_SEG_SHIFT=6" "GSR_SEG_SHIFT=7" "GSR_SEG_SHIFT=8" "GSR_FWD_HINTS=off" "GSR_FWD=q" "GSR_FWD=block"}; do
    for wl in ${WLS:-1M-800-sh3 250k-512-sh0 100k-800-sh3 5k-256-sh0}; do
      echo "== [$v] $wl"; env $v timeout 300 python bench.py --cpu-budget 0 --workload $wl 2>>gpurun_out/ab_err.log | benchline
    done
    echo "== [$v] 1M trained"; env $v timeout 300 python bench.py --cpu-budget 0 --kind trained 2>>gpurun_out/ab_err.log | benchline
  done;;
prof)
  echo "== rocprofv3 kernel trace (1M)"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_1M -o r03 -- python $R/bench.py --steps 20 --warmup 5 --cpu-budget 0 --no-roofline > $R/gpurun_out/prof_1M.log 2>&1)
  tail -2 gpurun_out/prof_1M.log
  f=$(find gpurun_out/prof_1M -name "r03*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" | cut -c1-60,200-320;;
prof5k)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_5k -o r03 -- python $R/bench.py --workload 5k-256-sh0 --steps 20 --warmup 5 --cpu-budget 0 --no-roofline > $R/gpurun_out/prof_5k.log 2>&1)
  f=$(find gpurun_out/prof_5k -name "r03*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" | cut -c1-60,200-320;;
pmc)
  echo "== rocprofv3 PMC passes (1M)"
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
    tag=$(echo $c | cut -d" " -f1)
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag -o r03 -- python $R/bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-roofline > $R/gpurun_out/pmc_$tag.log 2>&1)
    tail -1 gpurun_out/pmc_$tag.log | cut -c1-200
  done
  python tools/pmc_summary.py gpurun_out --json gpurun_out/pmc_traffic.json --key 1M-800-sh3/blob 2>&1 | tee gpurun_out/pmc_summary.txt | tail -40;;
stage1)
  echo "== stage-1 (BASELINE configs[4]) through libgsr.so"
  timeout 1200 python tools/run_stage1.py --out gpurun_out/stage1.json $STAGE1_ARGS 2>&1 | tail -25;;
views)
  for m in "--views 8 --views-serial" "--views 8"; do
    timeout 300 python bench.py --workload 250k-512-sh0 --cpu-budget 0 $m 2>gpurun_out/views_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['views_mode'], d['value'], 'Mrays/s', d['ms_per_step'], 'ms per 8 views')"
    timeout 300 python bench.py --workload 5k-256-sh0 --cpu-budget 0 $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('5k', d['config']['views_mode'], d['value'], 'Mrays/s', d['ms_per_step'], 'ms per 8 views')"
  done;;
viewsprof)
  echo "== rocprofv3 kernel trace, 8 views in one chain (250k / 512^2)"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_views -o r03v -- python $R/bench.py --workload 250k-512-sh0 --views 8 --steps 5 --warmup 2 --cpu-budget 0 --no-roofline > $R/gpurun_out/prof_views.log 2>&1)
  f=$(find gpurun_out/prof_views -name "r03*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" | cut -c1-50,180-330;;
sds)
  echo "== bench --gpus 2 on a 1-GPU box must fail loudly"; timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 2>&1 | tail -3
  echo "== bench --step sds (1 GPU, no collectives)"; timeout 300 python bench.py --step sds --cpu-budget 0 2>&1 | grep -a "^{" | tee gpurun_out/sds_plain.json | cut -c1-400
  echo "== bench --step sds --force-collectives (1 GPU, 1-rank RCCL group)"; timeout 300 python bench.py --step sds --cpu-budget 0 --force-collectives 2>&1 | grep -a "^{" | tee gpurun_out/sds_rccl.json | cut -c1-400;;
smoke)
  echo "== __graft_entry__ smoke"; timeout 600 python __graft_entry__.py --smoke 2>&1 | tail -3;;
cpubase)
  for wl in 100k-800-sh3 5k-256-sh0; do
    echo "== bench $wl with the CPU oracle"; timeout 900 python bench.py --workload $wl --cpu-budget ${CPU_BUDGET:-10} 2> gpurun_out/benchcpu_$wl.err | tee gpurun_out/benchcpu_$wl.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['cpu_baseline'])"
  done;;
esac; done
