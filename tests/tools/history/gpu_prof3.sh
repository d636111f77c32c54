mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -q -x ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu.log | head -1
timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1; tail -1 gpurun_out/bench_quick.log | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']),d['ms_per_step'],d['roofline']['kernel_ms'])"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $R/gpurun_out/rocprof_bench.log 2>&1; echo "rocprof rc=$?"
cd $R; find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs head -4
