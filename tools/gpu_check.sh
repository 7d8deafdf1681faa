#!/bin/bash
# One gpurun call; sections chosen by arguments (default: all). Outputs -> gpurun_out/.
#   tests smoke bench benchall prof pmc fields views ab
mkdir -p gpurun_out
export TMPDIR=/tmp
SECTIONS="${@:-tests smoke bench benchall prof}"
R=$GRAFT_REPO_ROOT
for s in $SECTIONS; do case $s in
tests)
  echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -rf > gpurun_out/pytest_gpu.log 2>&1
  tail -25 gpurun_out/pytest_gpu.log;;
smoke)
  echo "== smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log;;
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
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_1M -o r01 -- python $R/bench.py --steps 5 --warmup 2 --cpu-budget 0 --no-roofline > $R/gpurun_out/prof_1M.log 2>&1)
  tail -2 gpurun_out/prof_1M.log; find gpurun_out/prof_1M -name "*stats*" | head
  f=$(find gpurun_out/prof_1M -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f";;
fields)
  echo "== extract_fields bench + kernel trace"
  timeout 200 python tools/fields_bench.py 2> gpurun_out/fields_bench.err | tee gpurun_out/fields_bench.jsonl
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fields -o r01 -- python $R/tools/fields_bench.py --sizes 100000 --reps 5 > $R/gpurun_out/prof_fields.log 2>&1)
  f=$(find gpurun_out/prof_fields -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_fields -o r01 -- python $R/tools/fields_bench.py --sizes 100000 --reps 2 > $R/gpurun_out/pmc_fields.log 2>&1)
  f=$(find gpurun_out/pmc_fields -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "fields" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    print(k, {a: round(b) for a, b in c.items()})
    if c.get("SQ_BUSY_CYCLES"): print("   VALU busy share of SQ busy cycles:", round(c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_BUSY_CYCLES"] , 3))
PY
  ;;
ab)
  # A/B of the env-selectable variants: parity suite + 1M bench for each
  for v in ${AB_VARIANTS:-"GSR_DEFAULT=1" "GSR_BWD=b2f" "GSR_RECORDS=copy"}; do
    if [ -z "$AB_NOTEST" ]; then echo "== variant [$v] tests"; env $v timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=line -x 2>&1 | tail -4; fi
    echo "== variant [$v] bench 1M"; env $v timeout 300 python bench.py --cpu-budget 0 --trace-steps 2>gpurun_out/ab_err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_with_events'], d['kernels_ms_per_step'])"
    grep "step ms" gpurun_out/ab_err.log
    if [ -n "$AB_TRAINED" ]; then
      echo "== variant [$v] bench 1M trained"; env $v timeout 300 python bench.py --cpu-budget 0 --kind trained 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['M'], d['config']['M_emitted'], d['kernels_ms_per_step'])"
    fi
  done;;
views)
  echo "== batched views on one GPU (BASELINE configs[3] scene: 250k, 512^2, 8 cameras)"
  for m in "--views 8 --views-serial" "--views 8"; do
    timeout 300 python bench.py --workload 250k-512-sh0 --cpu-budget 0 $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['views_mode'], d['value'], 'Mrays/s', d['ms_per_step'], 'ms per 8 views')"
  done
  for m in "--views 8 --views-serial" "--views 8"; do
    timeout 300 python bench.py --workload 5k-256-sh0 --cpu-budget 0 $m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('5k', d['config']['views_mode'], d['value'], 'Mrays/s', d['ms_per_step'], 'ms per 8 views')"
  done;;
pmc)
  echo "== rocprofv3 PMC passes (1M)"
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"; do
    tag=$(echo $c | cut -d" " -f1)
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag -o r01 -- python $R/bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-roofline > $R/gpurun_out/pmc_$tag.log 2>&1)
    tail -1 gpurun_out/pmc_$tag.log | cut -c1-200
  done
  python tools/pmc_summary.py gpurun_out 2>&1 | tail -30;;
esac; done
