# round 2, call M: full GPU suite with step_main_wp as the hidden-64 / 128 default + benches
set -x
mkdir -p gpurun_out/r2m
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r2m
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --no-gpu-baseline > $O/bench.json 2> $O/bench.err < /dev/null; tail -1 $O/bench.json | head -c 250; echo
timeout 300 python bench.py --with-background --no-cpu-baseline --no-gpu-baseline > $O/bench_withbg.json 2> $O/bench_withbg.err < /dev/null
tail -1 $O/bench_withbg.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j.get('with_background'))"
for k in auto ws1; do
timeout 120 python bench.py --config background --kernel $k --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $O/bench_background_$k.json 2>&1 < /dev/null; tail -1 $O/bench_background_$k.json | head -c 230; echo
done
true
