# round 3, call J: step_main_ws with single-tile rounds (NT = 1) where every tile gets its own compute unit: the per-rank shapes of the
# ray-sharded background model at 4 and 8 GPUs, against two-tile rounds (tuning.ws_flags = 1); whole GPU suite
set -x
mkdir -p gpurun_out/r3j
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r3j
( timeout 900 python -m pytest tests -m gpu -q --maxfail=20 ) > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
for c in background_rank8 background_rank4 background; do
for v in "" "--ws-two-tile"; do
timeout 200 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline $v > $O/bench_$c$v.json 2>&1 < /dev/null; tail -1 $O/bench_$c$v.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$c $v', j['ms_per_step'], j['roofline']['kernel_ms'])"
done
done
cd /tmp
for c in background_rank8 background_rank4; do
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -o p -- python $R/bench.py --config $c --steps 200 --warmup 20 --timed-only > $O/prof_$c.log 2>&1 < /dev/null
head -4 $O/prof_$c/p_kernel_stats.csv | cut -c1-170
done
cd $R
true
