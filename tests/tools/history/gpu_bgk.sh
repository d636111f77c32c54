mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/pytest_gpu.log
for K in gen wide2 auto; do
  timeout 300 python bench.py --config background --kernel $K --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_bg_$K.log 2>&1; echo "bg $K rc=$?"; tail -1 gpurun_out/bench_bg_$K.log | cut -c1-200
done
for K in wide wide2 auto; do
  timeout 300 python bench.py --config imap_plumbing --kernel $K --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_imap_$K.log 2>&1; echo "imap $K rc=$?"; tail -1 gpurun_out/bench_imap_$K.log | cut -c1-200
done
