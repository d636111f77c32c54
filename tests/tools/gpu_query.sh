mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/pytest_gpu.log
timeout 600 python tests/tools/query_bench.py > gpurun_out/query_bench.json 2> gpurun_out/query_bench.err; echo "query bench rc=$?"; cat gpurun_out/query_bench.json; tail -3 gpurun_out/query_bench.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_query -o q -- python $GRAFT_REPO_ROOT/tests/tools/query_bench.py > /dev/null 2>&1; echo "rocprof rc=$?"
find $GRAFT_REPO_ROOT/gpurun_out/prof_query -name "*kernel_stats.csv" | head -1 | xargs -r head -8
