# round 3, call R: step_main_ws with streaming (nontemporal) partial-row stores: bench + HBM counters
set -x
mkdir -p gpurun_out/r3r
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r3r
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bg or 128 or h64 or hidden" ) > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
for c in background background_rank8; do
timeout 200 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $O/bench_$c.json 2>&1 < /dev/null; tail -1 $O/bench_$c.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$c', j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline'].get('floor_us'))"
done
cd /tmp
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $R/gpurun_out/pmc/$C
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc/$C -o p -- python $R/tests/tools/run_steps.py background 40 > $O/pmc_$C.log 2>&1 < /dev/null
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bg -o bg -- python $R/bench.py --config background --steps 200 --warmup 20 --timed-only > $O/prof_bg.log 2>&1 < /dev/null
cd $R
head -3 $O/prof_bg/bg_kernel_stats.csv | cut -c1-160
python tests/tools/pmc_summary.py > $O/pmc_counters_background.json 2>$O/pmc_summary.err; tail -7 $O/pmc_counters_background.json
