# short bench + per-kernel durations (rocprofv3 kernel trace)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1; tail -1 gpurun_out/bench_quick.log | cut -c1-200
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $R/gpurun_out/rocprof_bench.log 2>&1; echo "rocprof rc=$?"
cd $R; cat gpurun_out/prof/*kernel_stats.csv | head -6
