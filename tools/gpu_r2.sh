#!/bin/bash
# Round-2 gpurun driver: sections chosen by arguments. Outputs -> gpurun_out/.
#   tests | variants | bench | benchall | prof | pmc | stage1 | views | sds
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
benchline() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms', d['kernels_ms_per_step'])"; }
for s in "$@"; do case $s in
quick)
  echo "== pytest quick (views, optim, parity without the big fp64 cases)"
  timeout 900 python -m pytest tests/test_views_gpu.py tests/test_optim_gpu.py tests/test_parity_gpu.py tests/test_fuzz_gpu.py -m gpu -q -p no:cacheprovider --tb=short -rf -k "not baseline_config" > gpurun_out/pytest_quick.log 2>&1
  grep -a "passed\|failed\|FAILED\|Error\|assert" gpurun_out/pytest_quick.log | tail -30;;
viewstest)
  timeout 600 python -m pytest tests/test_views_gpu.py -m gpu -q -p no:cacheprovider --tb=short -rf > gpurun_out/pytest_views.log 2>&1
  grep -a "passed\|failed\|FAILED\|Error\|assert\|differing" gpurun_out/pytest_views.log | cut -c1-600 | tail -20;;
tests)
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider --tb=short -rf > gpurun_out/pytest_gpu.log 2>&1
  grep -a "fragile:\|passed\|failed\|FAILED\|Error" gpurun_out/pytest_gpu.log | tail -30;;
variants)
  # parity of every env-selectable kernel variant + same-box 1M bench
  for v in ${VARIANTS:-"GSR_FWD=q" "GSR_FWD=block"}; do
    echo "== variant [$v] parity"; env $v timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_views_gpu.py -m gpu -q -s -p no:cacheprovider --tb=short -x 2>&1 | grep -a "fragile:\|passed\|failed\|FAILED\|Error\|assert" | tail -12
  done
  for v in "GSR_X=0" ${VARIANTS:-"GSR_FWD=q" "GSR_FWD=block"}; do
    echo "== variant [$v] bench 1M blob"; env $v timeout 300 python bench.py --cpu-budget 0 2>gpurun_out/ab_err.log | benchline
    echo "== variant [$v] bench 1M trained"; env $v timeout 300 python bench.py --cpu-budget 0 --kind trained 2>>gpurun_out/ab_err.log | benchline
    echo "== variant [$v] bench 100k"; env $v timeout 300 python bench.py --cpu-budget 0 --workload 100k-800-sh3 2>>gpurun_out/ab_err.log | benchline
  done;;
bench)
  echo "== bench 1M"; timeout 900 python bench.py $BENCH_ARGS 2> gpurun_out/bench_1M.err | tee gpurun_out/bench_1M.json
  tail -3 gpurun_out/bench_1M.err;;
benchall)
  for wl in 100k-800-sh3 250k-512-sh0 5k-256-sh0; do
    echo "== bench $wl"; timeout 300 python bench.py --workload $wl --cpu-budget 0 2> gpurun_out/bench_$wl.err | tee gpurun_out/bench_$wl.json
  done
  echo "== bench 1M trained"; timeout 300 python bench.py --kind trained --cpu-budget 0 2> gpurun_out/bench_1M_trained.err | tee gpurun_out/bench_1M_trained.json;;
prof)
  echo "== rocprofv3 kernel trace (1M)"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_1M -o r02 -- python $R/bench.py --steps 20 --warmup 5 --cpu-budget 0 --no-roofline > $R/gpurun_out/prof_1M.log 2>&1)
  tail -2 gpurun_out/prof_1M.log
  f=$(find gpurun_out/prof_1M -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-60,200-320;;
pmc)
  echo "== rocprofv3 PMC passes (1M)"
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
    tag=$(echo $c | cut -d" " -f1)
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag -o r02 -- python $R/bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-roofline > $R/gpurun_out/pmc_$tag.log 2>&1)
    tail -1 gpurun_out/pmc_$tag.log | cut -c1-200
  done
  python tools/pmc_summary.py gpurun_out 2>&1 | tail -40;;
stage1)
  echo "== stage-1 (BASELINE configs[4]) through libgsr.so"
  timeout 1200 python tools/run_stage1.py --out gpurun_out/stage1.json $STAGE1_ARGS 2>&1 | tail -25;;
views)
  for m in "--views 8 --views-serial" "--views 8 --views-mode streams" "--views 8"; do
    timeout 300 python bench.py --workload 250k-512-sh0 --cpu-budget 0 $m 2>gpurun_out/views_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['views_mode'], d['value'], 'Mrays/s', d['ms_per_step'], 'ms per 8 views')"
    timeout 300 python bench.py --workload 5k-256-sh0 --cpu-budget 0 $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('5k', d['config']['views_mode'], d['value'], 'Mrays/s', d['ms_per_step'], 'ms per 8 views')"
  done;;
pmcv)
  # SQ / LDS / L2 counters per env-selectable variant (one bench run per pass)
  for v in ${VARIANTS:-"GSR_SEG_SHIFT=6" "GSR_SEG_SHIFT=7"}; do
    tagv=$(echo $v | tr '= ' '__')
    i=0
    for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      (cd /tmp && env $v timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcv_$tagv/p$i -o r02 -- python $R/bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-roofline $PMCV_ARGS > $R/gpurun_out/pmcv_${tagv}_p$i.log 2>&1)
    done
    echo "== PMC variant [$v]"; python tools/pmc_summary.py gpurun_out/pmcv_$tagv 2>&1 | grep "render\|preprocess\|scatter\|tile_sort" | cut -c1-900
  done;;
abq)
  for v in ${VARIANTS:-"GSR_SEG_SHIFT=8"}; do
    echo "== variant [$v] bench 1M blob"; env $v timeout 300 python bench.py --cpu-budget 0 2>gpurun_out/ab_err.log | benchline
    echo "== variant [$v] bench 100k"; env $v timeout 300 python bench.py --cpu-budget 0 --workload 100k-800-sh3 2>>gpurun_out/ab_err.log | benchline
  done;;
viewsprof)
  echo "== rocprofv3 kernel trace, 8 views in one chain (250k / 512^2)"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_views -o r02v -- python $R/bench.py --workload 250k-512-sh0 --views 8 --steps 5 --warmup 2 --cpu-budget 0 --no-roofline > $R/gpurun_out/prof_views.log 2>&1)
  f=$(find gpurun_out/prof_views -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" | cut -c1-50,180-330;;
sds)
  echo "== bench --gpus 2 on a 1-GPU box must fail loudly"; timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 2>&1 | tail -3
  echo "== bench --step sds (1 GPU)"; timeout 300 python bench.py --step sds --cpu-budget 0 2>&1 | tail -2;;
esac; done
