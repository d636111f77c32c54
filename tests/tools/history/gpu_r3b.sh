# round 3, call B: GPU suite after the tolerance / kink-accounting fixes + kernel stats of the background step
set -x
mkdir -p gpurun_out/r3b
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r3b
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 ) > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bg -o bg -- python $R/bench.py --config background --steps 200 --warmup 20 --timed-only > $O/prof_bg.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_wbg -o wbg -- python $R/bench.py --with-background --steps 200 --warmup 20 --timed-only > $O/prof_wbg.log 2>&1 < /dev/null
cd $R
for f in $O/prof_bg/*kernel_stats.csv $O/prof_wbg/*kernel_stats.csv; do [ -f "$f" ] && head -8 "$f" | cut -c1-170; done
true
