set -x
mkdir -p gpurun_out/wp
export TMPDIR=/tmp
O=$PWD/gpurun_out/wp
( timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_all.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest_all.log | tail -2
for k in auto ws1; do
timeout 200 python bench.py --config stress_256x64 --kernel $k --steps 40 --warmup 5 --no-cpu-baseline --no-gpu-baseline > $O/bench_stress_$k.json 2> $O/bench_stress_$k.err < /dev/null; tail -1 $O/bench_stress_$k.json | head -c 230; echo
done
timeout 200 python bench.py --config stress_256x64 --weights bf16 --steps 40 --warmup 5 --no-cpu-baseline --no-gpu-baseline > $O/bench_stress_bf16.json 2> $O/bench_stress_bf16.err < /dev/null; tail -1 $O/bench_stress_bf16.json | head -c 230; echo
timeout 120 python bench.py --config background --kernel wp --no-cpu-baseline --no-gpu-baseline --steps 200 --warmup 20 > $O/bench_bg_wp.json 2> $O/bench_bg_wp.err < /dev/null; tail -1 $O/bench_bg_wp.json | head -c 230; echo
