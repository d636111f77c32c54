# round 3, call E: step_main_ws with single-tile tail rounds (200 workgroups x 3 tiles at the background shape) vs the even split
set -x
mkdir -p gpurun_out/r3e
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r3e
( timeout 900 python -m pytest tests -m gpu -q --maxfail=20 ) > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
for v in "" "--ws-no-tail"; do
timeout 200 python bench.py --config background --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline $v > $O/bench_background$v.json 2>&1 < /dev/null; tail -1 $O/bench_background$v.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms'])"
done
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bg -o bg -- python $R/bench.py --config background --steps 200 --warmup 20 --timed-only > $O/prof_bg.log 2>&1 < /dev/null
cd $R
head -4 $O/prof_bg/bg_kernel_stats.csv | cut -c1-170
timeout 100 python tests/tools/phase_profile.py background > $O/phases_background.txt 2>&1; tail -20 $O/phases_background.txt
true
