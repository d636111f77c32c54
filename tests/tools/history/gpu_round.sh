set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time python __graft_entry__.py --smoke ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/summary.txt
( time timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/summary.txt
tail -25 gpurun_out/pytest_gpu.log
timeout 300 python tests/tools/phase_profile.py > gpurun_out/phases.txt 2>&1; cat gpurun_out/phases.txt
( time timeout 600 python bench.py ) > gpurun_out/bench.log 2>&1; echo "bench rc=$?" | tee -a gpurun_out/summary.txt
tail -2 gpurun_out/bench.log
for C in scannet0024_vmap background imap_plumbing; do
  timeout 300 python bench.py --config $C --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_$C.log 2>&1; echo "bench $C rc=$?" | tee -a gpurun_out/summary.txt
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $R/gpurun_out/rocprof_bench.log 2>&1; echo "rocprof rc=$?" | tee -a $R/gpurun_out/summary.txt
cd $R; cat gpurun_out/prof/*kernel_stats.csv | head -6
