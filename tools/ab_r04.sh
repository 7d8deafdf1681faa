# HEAD against the round-4 tree (git worktree add _exp/r04tree 7aad645 && (cd _exp/r04tree && python -m dreamgaussian_amd.build)) on one box
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], json.dumps(d['kernels_ms_per_step']))"; }
for rep in 1 2; do
for wl in "250k-512-sh0 --views 8" "250k-512-sh0" "100k-800-sh3" "1M-800-sh3" "1M-800-sh3 --kind trained"; do
  echo "== HEAD $wl"; python bench.py --workload $wl --cpu-budget 0 --steps 60 --warmup 10 2>/dev/null | line
  echo "== r04  $wl"; (cd _exp/r04tree && python bench.py --workload $wl --cpu-budget 0 --steps 60 --warmup 10 2>/dev/null | line)
done
done
