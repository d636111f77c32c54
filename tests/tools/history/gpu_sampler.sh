mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/pytest_gpu.log
timeout 600 python tests/tools/sampler_bench.py > gpurun_out/sampler_bench.json 2> gpurun_out/sampler_bench.err; echo "sampler bench rc=$?"; cat gpurun_out/sampler_bench.json; tail -3 gpurun_out/sampler_bench.err
