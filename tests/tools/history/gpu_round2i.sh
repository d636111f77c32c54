# round 2, call I: full GPU suite with step_main_ws as the hidden-128 default + headline / background benches
set -x
mkdir -p gpurun_out/r2i
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r2i
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --no-gpu-baseline > $O/bench.json 2> $O/bench.err < /dev/null; tail -1 $O/bench.json | head -c 300; echo
timeout 300 python bench.py --with-background --no-cpu-baseline --no-gpu-baseline > $O/bench_withbg.json 2> $O/bench_withbg.err < /dev/null
tail -1 $O/bench_withbg.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j.get('with_background'))"
timeout 120 python bench.py --config background --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $O/bench_background.json 2>&1 < /dev/null; tail -1 $O/bench_background.json | head -c 250; echo
true
