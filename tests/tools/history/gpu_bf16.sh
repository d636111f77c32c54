mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python bench.py --config scannet0024_vmap --weights bf16 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_scannet_bf16.log 2>&1; echo rc=$?; tail -1 gpurun_out/bench_scannet_bf16.log | cut -c1-300
timeout 300 python bench.py --config stress_256x64 --weights bf16 --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/bench_stress_bf16.log 2>&1; echo rc=$?; tail -1 gpurun_out/bench_stress_bf16.log | cut -c1-300
