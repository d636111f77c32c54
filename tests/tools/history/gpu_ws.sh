# step_main_ws (hidden 128): parity tests, bench of the background config, phase stamps, kernel trace (csv)
set -x
mkdir -p gpurun_out/ws
export TMPDIR=/tmp
O=$PWD/gpurun_out/ws
R=$PWD
( timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "generic_width or parameter_image or shared_background or driver_background" ) > $O/pytest_ws.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_ws.log
timeout 120 python bench.py --config background --no-cpu-baseline --no-gpu-baseline --steps 200 --warmup 20 > $O/bench_bg_auto.json 2> $O/bench_bg_auto.err < /dev/null; tail -1 $O/bench_bg_auto.json | head -c 330; echo
timeout 120 python tests/tools/phase_profile.py background > $O/phases_bg.txt 2>&1 < /dev/null; tail -18 $O/phases_bg.txt
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bg -- python $R/bench.py --config background --timed-only --steps 100 --warmup 10 > $O/prof_run.log 2>&1 < /dev/null
cd $R
for f in $O/prof/*kernel_stats.csv; do [ -f "$f" ] && head -5 "$f"; done
true
