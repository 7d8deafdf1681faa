# same-box A/B of library builds: AB_LIBS="none _exp/libgsr_x.so ..." AB_WL="250k-512-sh0 ..." [BENCH_ARGS]
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], json.dumps(d['kernels_ms_per_step']))"; }
R=$(pwd)
for rep in 1 2; do for wl in ${AB_WL:-250k-512-sh0}; do for l in $AB_LIBS; do
  echo "== [$l] $wl"; if [ $l = none ]; then python bench.py --workload $wl --cpu-budget 0 --steps 60 --warmup 10 $BENCH_ARGS 2>/dev/null | line; else GSR_LIB=$R/$l python bench.py --workload $wl --cpu-budget 0 --steps 60 --warmup 10 $BENCH_ARGS 2>/dev/null | line; fi
done; done; done
