mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/pytest_gpu.log
for C in background imap_plumbing stress_256x64; do
  timeout 300 python bench.py --config $C --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_$C.log 2>&1; echo "bench $C rc=$?"; tail -1 gpurun_out/bench_$C.log | cut -c1-200
done
