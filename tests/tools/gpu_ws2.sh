set -x
mkdir -p gpurun_out/wp
export TMPDIR=/tmp
O=$PWD/gpurun_out/wp
( timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_all.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest_all.log | tail -2
timeout 300 python bench.py --with-background --no-cpu-baseline --no-gpu-baseline > $O/bench_withbg.json 2> $O/bench_withbg.err < /dev/null
tail -1 $O/bench_withbg.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j.get('with_background'))"
timeout 120 python bench.py --config background --no-cpu-baseline --no-gpu-baseline --steps 200 --warmup 20 > $O/bench_bg_auto.json 2> $O/bench_bg_auto.err < /dev/null; tail -1 $O/bench_bg_auto.json | head -c 230; echo
