mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/pytest_gpu.log
timeout 600 python tests/tools/query_bench.py > gpurun_out/query_bench.json 2> gpurun_out/query_bench.err; echo "query bench rc=$?"; cat gpurun_out/query_bench.json; tail -3 gpurun_out/query_bench.err
