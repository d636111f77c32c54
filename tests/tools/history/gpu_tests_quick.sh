mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1; tail -1 gpurun_out/bench_quick.log | cut -c1-220
timeout 300 python bench.py --config scannet0024_vmap --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_quick_scannet.log 2>&1; tail -1 gpurun_out/bench_quick_scannet.log | cut -c1-220
