#!/bin/bash
# gpurun driver (rounds 4 and 5): sections chosen by arguments. Outputs -> gpurun_out/.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_r5.sh new bench1m'
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
benchline() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms', json.dumps(d['kernels_ms_per_step']))"; }
for sec in "$@"; do
case $sec in
valu)
  echo "== VALU ceiling of the compositing kernels' own mix"
  timeout 120 _exp/valu_ceiling 400 24 | tee gpurun_out/valu_plain.jsonl | cut -c1-200
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/valu_pmc -o v -- $R/_exp/valu_ceiling 400 24 > $R/gpurun_out/valu_pmc.log 2>&1)
  tail -2 gpurun_out/valu_pmc.log | cut -c1-200
  python tools/valu_ceiling.py gpurun_out/valu_plain.jsonl gpurun_out/valu_pmc | tee gpurun_out/valu_ceiling.txt;;
bench1m)
  echo "== bench 1M"; timeout 900 python bench.py --cpu-budget 0 $BENCH_ARGS 2> gpurun_out/bench_1M.err | tee gpurun_out/bench_1M.json | benchline
  tail -3 gpurun_out/bench_1M.err;;
benchcpu)
  echo "== bench 1M with the CPU baseline"; timeout 900 python bench.py 2> gpurun_out/bench_1M_cpu.err | tee gpurun_out/bench_1M_cpu.json | benchline;;
others)
  for wl in 100k-800-sh3 250k-512-sh0 5k-256-sh0; do
    echo "== bench $wl"; timeout 300 python bench.py --workload $wl --cpu-budget 0 2> gpurun_out/bench_$wl.err | tee gpurun_out/bench_$wl.json | benchline
  done
  echo "== bench 1M trained"; timeout 300 python bench.py --kind trained --cpu-budget 0 2> gpurun_out/bench_1M_trained.err | tee gpurun_out/bench_1M_trained.json | benchline;;
morton)
  rm -f gpurun_out/bench_order.jsonl
  for wl in 1M-800-sh3 100k-800-sh3 250k-512-sh0; do
    for o in given morton; do
      echo "== bench $wl --order $o"; timeout 300 python bench.py --workload $wl --order $o --cpu-budget 0 2>> gpurun_out/bench_order.err | tee -a gpurun_out/bench_order.jsonl | benchline
    done
  done
  for o in given morton; do
    echo "== bench 1M trained --order $o"; timeout 300 python bench.py --kind trained --order $o --cpu-budget 0 2>> gpurun_out/bench_order.err | tee -a gpurun_out/bench_order.jsonl | benchline
  done;;
gradstats)
  echo "== gradient statistics"; timeout 300 python tools/grad_stats.py 2>&1 | tail -1 | tee gpurun_out/grad_stats.json
  timeout 300 python tools/grad_stats.py --kind trained 2>&1 | tail -1 | tee -a gpurun_out/grad_stats.json;;
ab)
  # A/B of library builds on the same box: AB_LIBS="_exp/a.so _exp/b.so", AB_WL="1M-800-sh3 ..."
  for wl in ${AB_WL:-1M-800-sh3}; do
    for l in $AB_LIBS; do
      echo "== [$l] $wl"; GSR_LIB=$R/$l timeout 300 python bench.py --cpu-budget 0 --workload $wl $BENCH_ARGS 2>>gpurun_out/ab_err.log | tee -a gpurun_out/ab.jsonl | benchline
    done
  done;;
hooks)
  # A/B of test hooks on the same box and library: AB_HOOKS="none fwd_lds_kb=44 fwd_lds_kb=44,fwd_prio=1", AB_WL="1M-800-sh3 1M-800-sh3:trained"
  for wl in ${AB_WL:-1M-800-sh3}; do
    kind=blob; [ "${wl#*:}" != "$wl" ] && kind=${wl#*:}
    for h in $AB_HOOKS; do
      args=""; [ "$h" != none ] && for x in ${h//,/ }; do args="$args --hook $x"; done
      echo "== [$h] $wl"; timeout 300 python bench.py --cpu-budget 0 --workload ${wl%%:*} --kind $kind $args $BENCH_ARGS 2>>gpurun_out/ab_err.log | tee -a gpurun_out/ab_hooks.jsonl | benchline
    done
  done;;
abquick)
  # the parity subset a variant library must pass before it is adopted (GSR_LIB=... in the environment of the call)
  echo "== pytest parity subset [${GSR_LIB:-in-tree library}]"
  timeout ${QUICK_TIMEOUT:-240} python -m pytest ${QUICK_FILES:-tests/test_parity_gpu.py tests/test_fuzz_gpu.py} -m gpu -q -x -p no:cacheprovider --tb=short -rf --durations=8 \
      -k "${QUICK_K:-forward_backward_match_oracle or committed_golden or depth_ties or segment_lengths or stage1_trained or cfg1_100k_blob or (test_fuzz and not large)}" > gpurun_out/pytest_abquick.log 2>&1
  grep -a "passed\|failed\|FAILED\|Error\|assert\|s call" gpurun_out/pytest_abquick.log | cut -c1-300 | tail -24;;
new)
  # the tests of this round's kernel changes, first (fail fast), then the parity subset
  echo "== pytest: round-5 tests"
  timeout ${QUICK_TIMEOUT:-400} python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider --tb=short -rf -k "${NEW_K:-pair or k6_compact or scan_folded or speculative or inference}" > gpurun_out/pytest_new.log 2>&1
  grep -a "passed\|failed\|FAILED\|Error\|assert" gpurun_out/pytest_new.log | cut -c1-300 | tail -12;;
quick)
  # the parity subset a variant library must pass before it is adopted (GSR_LIB=... in the environment of the call)
  echo "== pytest parity subset [${GSR_LIB:-in-tree library}]"
  timeout ${QUICK_TIMEOUT:-240} python -m pytest ${QUICK_FILES:-tests/test_parity_gpu.py tests/test_fuzz_gpu.py} -m gpu -q -x -p no:cacheprovider --tb=short -rf --durations=8 \
      -k "${QUICK_K:-forward_backward_match_oracle or committed_golden or depth_ties or segment_lengths or stage1_trained or cfg1_100k_blob or (test_fuzz and not large)}" > gpurun_out/pytest_abquick.log 2>&1
  grep -a "passed\|failed\|FAILED\|Error\|assert\|s call" gpurun_out/pytest_abquick.log | cut -c1-300 | tail -24;;
pair)
  # the two-waves-per-block forward (gsr_render_fwd_pair, test hook fwd_mode = 3): bit-identity test first, then the A/B against the serial walk
  echo "== pytest -k pair"
  timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider --tb=short -rf -k pair > gpurun_out/pytest_pair.log 2>&1
  tail -15 gpurun_out/pytest_pair.log | cut -c1-300
  if grep -q " passed" gpurun_out/pytest_pair.log && ! grep -q "failed" gpurun_out/pytest_pair.log; then
    for wl in 1M-800-sh3 1M-800-sh3:trained 100k-800-sh3 250k-512-sh0; do
      kind=blob; [ "${wl#*:}" != "$wl" ] && kind=${wl#*:}
      for h in none fwd_mode=3 none fwd_mode=3; do
        args=""; [ "$h" != none ] && args="--hook $h"
        echo "== [$h] $wl"; timeout 300 python bench.py --cpu-budget 0 --workload ${wl%%:*} --kind $kind --steps 60 --warmup 10 $args 2>>gpurun_out/ab_err.log | tee -a gpurun_out/ab_pair.jsonl | benchline
      done
    done
  fi;;
quick)
  echo "== pytest quick (GPU)"
  timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_views_gpu.py tests/test_optim_gpu.py tests/test_densify_gpu.py -m gpu -q -p no:cacheprovider --tb=short -rf -k "not baseline_config and not cfg3 and not full_size" > gpurun_out/pytest_quick.log 2>&1
  grep -a "passed\|failed\|FAILED\|Error\|assert" gpurun_out/pytest_quick.log | cut -c1-400 | tail -40;;
pytest)
  echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider --tb=short -rf > gpurun_out/pytest_gpu.log 2>&1
  grep -a "fragile:\|observed:\|passed\|failed\|FAILED\|Error" gpurun_out/pytest_gpu.log | cut -c1-700 | tail -40;;
prof)
  echo "== rocprofv3 kernel trace (1M)"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_1M -o r05 -- python $R/bench.py --steps 20 --warmup 5 --cpu-budget 0 --no-roofline > $R/gpurun_out/prof_1M.log 2>&1)
  tail -2 gpurun_out/prof_1M.log
  f=$(find gpurun_out/prof_1M -name "r05*kernel_stats.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Name'][:70].ljust(70), r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
  ;;
prof5k)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_5k -o r05 -- python $R/bench.py --workload 5k-256-sh0 --steps 20 --warmup 5 --cpu-budget 0 --no-roofline > $R/gpurun_out/prof_5k.log 2>&1)
  f=$(find gpurun_out/prof_5k -name "r05*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" | cut -c1-60,200-320;;
pmc)
  echo "== rocprofv3 PMC passes (1M)"
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
    tag=$(echo $c | cut -d" " -f1)
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag -o r05 -- python $R/bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-roofline $PMC_ARGS > $R/gpurun_out/pmc_$tag.log 2>&1)
    tail -1 gpurun_out/pmc_$tag.log | cut -c1-200
  done
  python tools/pmc_summary.py gpurun_out --json gpurun_out/pmc_traffic.json --key ${PMC_KEY:-1M-800-sh3/blob} 2>&1 | tee gpurun_out/pmc_summary.txt | tail -40;;
stage1)
  echo "== stage-1 (BASELINE configs[4]) through libgsr.so"
  timeout 1200 python tools/run_stage1.py --out gpurun_out/stage1.json $STAGE1_ARGS 2>&1 | tail -25;;
views)
  for m in "--views 8" "--views 8 --views-serial"; do
    timeout 300 python bench.py --workload 250k-512-sh0 --cpu-budget 0 $m 2>gpurun_out/views_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['views_mode'], d['value'], 'Mrays/s', d['ms_per_step'], 'ms per 8 views')"
  done;;
sds)
  rm -f gpurun_out/sds_one_gpu.jsonl
  for m in local gather; do
    echo "== bench --step sds --sds-mode $m (1 GPU, no collectives)"; timeout 300 python bench.py --step sds --sds-mode $m --cpu-budget 0 $SDS_ARGS 2>&1 | grep -a "^{" | tee -a gpurun_out/sds_one_gpu.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['config']['parallelism'])"
    echo "== bench --step sds --sds-mode $m --force-collectives (1 GPU, 1-rank RCCL group)"; timeout 300 python bench.py --step sds --sds-mode $m --cpu-budget 0 --force-collectives $SDS_ARGS 2>&1 | grep -a "^{" | tee -a gpurun_out/sds_one_gpu.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['config']['parallelism'])"
  done;;
sdstrace)
  # where the exchange's fixed cost sits: kernel + HIP API trace of the forced-collectives step (no counters in this run)
  for m in gather local; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --hip-trace --output-format csv -d $R/gpurun_out/sdstrace_$m -o t -- python $R/bench.py --step sds --sds-mode $m --force-collectives --cpu-budget 0 --steps 20 --warmup 5 > $R/gpurun_out/sdstrace_$m.log 2>&1)
    tail -1 gpurun_out/sdstrace_$m.log | cut -c1-200
    python tools/sds_trace.py gpurun_out/sdstrace_$m | tee gpurun_out/sds_trace_$m.txt | tail -40
  done;;
cpubase)
  for wl in 5k-256-sh0 100k-800-sh3; do
    echo "== bench $wl with the CPU oracle and the naive-GPU point"; timeout 900 python bench.py --workload $wl --cpu-budget ${CPU_BUDGET:-10} --naive-gpu 2> gpurun_out/benchcpu_$wl.err | tee gpurun_out/benchcpu_$wl.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['cpu_baseline'], d.get('naive_gpu'))"
  done;;
pmcmorton)
  echo "== rocprofv3 PMC passes, --order morton (1M)"
  mkdir -p gpurun_out/pmcm
  for c in "FETCH_SIZE" "WRITE_SIZE"; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmcm/pmc_$c -o r05 -- python $R/bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-roofline --order morton > $R/gpurun_out/pmcm/pmc_$c.log 2>&1)
  done
  python tools/pmc_summary.py gpurun_out/pmcm 2>&1 | tee gpurun_out/pmc_morton_summary.txt | cut -c1-300;;
floor5k)
  # what an EMPTY autograd.Function with the rasterizer's signature costs per fwd+bwd on this host, beside the library at 5k / 256^2
  python tools/host_overhead.py 2>&1 | grep -a "ms/step\|floor" | tee gpurun_out/host_floor_5k.txt
  for i in 1 2 3 4 5 6 7 8 9 10; do python bench.py --workload 5k-256-sh0 --cpu-budget 0 --no-roofline --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('5k-256-sh0 ms_per_step', d['ms_per_step'])"; done | tee -a gpurun_out/host_floor_5k.txt;;
smoke)
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3;;
*) echo "unknown section $sec";;
esac
done
