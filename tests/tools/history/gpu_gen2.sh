mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/pytest_gpu.log
for C in background imap_plumbing stress_256x64; do
  timeout 300 python bench.py --config $C --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_$C.log 2>&1; echo "bench $C rc=$?"; tail -1 gpurun_out/bench_$C.log | cut -c1-200
done
timeout 600 python tests/tools/query_bench.py > gpurun_out/query_bench.json 2> gpurun_out/query_bench.err; python -c "
import json
d=json.load(open('gpurun_out/query_bench.json'))
for g in d['grids']: print(g['hidden'], g['grid_dim'], round(g['hip_ms'],3), 'ms; eager', round(g['eager_torch_ms'],2), 'frac', round(g['frac_of_fp32_mfma_peak'],3))"
