# round 3, call C: finalize kernels with all partial-row loads of a batch in flight (headline + background), kernel stats
set -x
mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r3c
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fixture or adamw or trajectory or image or repeatable" ) > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline > $O/bench_20_5.json 2> $O/bench_20_5.err < /dev/null; tail -1 $O/bench_20_5.json | head -c 300; echo
timeout 300 python bench.py --no-cpu-baseline --no-gpu-baseline > $O/bench_400.json 2> $O/bench_400.err < /dev/null; tail -1 $O/bench_400.json | head -c 300; echo
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof205 -o hl -- python $R/bench.py --steps 20 --warmup 5 --timed-only > $O/prof_run205.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bg -o bg -- python $R/bench.py --config background --steps 200 --warmup 20 --timed-only > $O/prof_bg.log 2>&1 < /dev/null
cd $R
for f in $O/prof205/*kernel_stats.csv $O/prof_bg/*kernel_stats.csv; do [ -f "$f" ] && head -4 "$f" | cut -c1-170; done
true
