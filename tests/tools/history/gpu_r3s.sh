# round 3, call S: final tree - whole GPU suite, smoke, the driver's bench command, background line (floor_us, traffic), with-background
set -x
mkdir -p gpurun_out/r3s
export TMPDIR=/tmp
O=$PWD/gpurun_out/r3s
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err < /dev/null; tail -1 $O/bench_20_5.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['value'], j['ms_per_step'], r['frac'], r['frac_of_executed_pipe'], r['floor_us'], r['kernel_ms'], r['traffic'], j['value_exact_fp32_kernel']['value'], j['gpu_eager_baseline']['value'], j['cpu_baseline']['value'], j['cpu_baseline']['cores'])"
timeout 300 python bench.py --config background --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $O/bench_background.json 2>&1 < /dev/null; tail -1 $O/bench_background.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['value'], j['ms_per_step'], r['frac'], r['frac_of_executed_pipe'], r['floor_us'], r['kernel_ms'], r['traffic'])"
timeout 300 python bench.py --with-background --no-cpu-baseline --no-gpu-baseline > $O/bench_withbg.json 2> $O/bench_withbg.err < /dev/null; tail -1 $O/bench_withbg.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['with_background']['ms_per_step'])"
true
