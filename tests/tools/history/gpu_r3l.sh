# round 3, call L: the hidden-32 inference query on the bf16 matrix pipe (field_query_s32): tests + the mesh-grid bench
set -x
mkdir -p gpurun_out/r3l
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3l
( timeout 600 python -m pytest tests/test_query.py tests/test_sampler.py -m gpu -q -s ) > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -E "passed|failed|query 4M" $O/pytest_gpu.log
timeout 600 python tests/tools/query_bench.py > $O/query_bench.json 2> $O/query_bench.err < /dev/null; tail -1 $O/query_bench.json | python -c "
import sys,json
j=json.loads(sys.stdin.read())
for g in j['grids']: print(g['hidden'], g['grid_dim'], round(g['hip_ms'],3), 'ms', round(g['eager_torch_ms'],1), round(g['tflops_fp32'],1), 'TF', g['max_abs_diff_occ'], g['max_abs_diff_rgb'])"
true
