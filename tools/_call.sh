cd $GRAFT_REPO_ROOT
echo "== GSR_FWD=q parity"; GSR_FWD=q timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_views_gpu.py -m gpu -q -p no:cacheprovider --tb=short -k "not cfg2_1M_blob and not cfg1_100k_trained" 2>&1 | tail -3
benchline() { python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print(d['value'], 'Mrays/s', d['ms_per_step'], 'ms', 'K1', k['preprocess_fwd'], 'scatter', k['scatter'], 'sort', k['tile_sort'], 'fwd', k['render_fwd'], 'bwd', k['render_bwd'], 'K6', k['preprocess_bwd'])"; }
for v in "GSR_X=0" "GSR_FWD=q" "GSR_SEG_SHIFT=7" "GSR_SCATTER_GRID=256" "GSR_SCATTER_GRID=128"; do
for w in "--workload 1M-800-sh3" "--workload 1M-800-sh3 --kind trained" "--workload 100k-800-sh3" "--workload 250k-512-sh0"; do
  echo "== [$v] $w"; env $v timeout 300 python bench.py --cpu-budget 0 $w 2>>gpurun_out/ab_err.log | benchline
done; done
