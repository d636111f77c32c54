set -x
mkdir -p gpurun_out/wp
export TMPDIR=/tmp
O=$PWD/gpurun_out/wp
R=$PWD
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "generic_width or parameter_image or shared_background or hidden128 or bitwise" ) > $O/pytest_wp.log 2>&1 < /dev/null; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest_wp.log | tail -2
timeout 120 python bench.py --config background --no-cpu-baseline --no-gpu-baseline --steps 200 --warmup 20 > $O/bench_bg_auto.json 2> $O/bench_bg_auto.err < /dev/null; tail -1 $O/bench_bg_auto.json | head -c 230; echo
timeout 120 python tests/tools/phase_profile.py background > $O/phases_bg_ws.txt 2>&1 < /dev/null; tail -17 $O/phases_bg_ws.txt
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_auto -o bg -- python $R/bench.py --config background --timed-only --steps 200 --warmup 20 > $O/prof_run_auto.log 2>&1 < /dev/null
for f in $O/prof_auto/*kernel_stats.csv; do [ -f "$f" ] && head -3 "$f" | cut -c1-140; done
