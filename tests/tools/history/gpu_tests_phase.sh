mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -x ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu.log | head -1
timeout 300 python tests/tools/phase_profile.py > gpurun_out/phases.txt 2>&1; tail -18 gpurun_out/phases.txt
timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
timeout 300 python bench.py --config scannet0024_vmap --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
