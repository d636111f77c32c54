mkdir -p gpurun_out; export TMPDIR=/tmp
for w in 1 2 3; do
  VMAPSTEP_QUERY_WPE=$w timeout 300 python tests/tools/query_bench.py > gpurun_out/query_bench_wpe$w.json 2> gpurun_out/query_bench_wpe$w.err; echo "wpe=$w rc=$?"
  python -c "
import json;d=json.load(open('gpurun_out/query_bench_wpe$w.json'))
for g in d['grids']: print(g['grid_dim'], round(g['hip_ms'],3),'ms', round(g['frac_of_fp32_mfma_peak'],3), g['max_abs_diff_occ'])"
done
