# round 3, call A: the whole GPU suite on the re-structured library (ABI v5, one translation unit per kernel family), the
# driver's bench command, its rocprofv3 kernel stats, HBM counters of the headline step, background / bf16 shapes
set -x
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r3a
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 ) > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -30 $O/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err < /dev/null; tail -1 $O/bench_20_5.json | head -c 400; echo
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof205 -o hl -- python $R/bench.py --steps 20 --warmup 5 --timed-only > $O/prof_run205.log 2>&1 < /dev/null
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc/$C -o p -- python $R/tests/tools/run_steps.py replica_room0_vmap 40 > $O/pmc_$C.log 2>&1 < /dev/null
  echo "pmc $C rc=$?"
done
cd $R
for f in $O/prof205/*kernel_stats.csv; do [ -f "$f" ] && head -6 "$f" | cut -c1-160; done
python tests/tools/pmc_summary.py > $O/pmc_counters.json 2>$O/pmc_summary.err; tail -8 $O/pmc_counters.json
timeout 200 python bench.py --config background --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $O/bench_background.json 2>&1 < /dev/null; tail -1 $O/bench_background.json | head -c 300; echo
timeout 100 python tests/tools/phase_profile.py background > $O/phases_background.txt 2>&1; tail -22 $O/phases_background.txt
timeout 200 python bench.py --config scannet0024_vmap --weights bf16 --no-cpu-baseline --no-gpu-baseline > $O/bench_scannet_bf16.json 2>&1 < /dev/null; tail -1 $O/bench_scannet_bf16.json | head -c 200; echo
timeout 200 python bench.py --config stress_256x64 --weights bf16 --steps 60 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $O/bench_stress_bf16.json 2>&1 < /dev/null; tail -1 $O/bench_stress_bf16.json | head -c 200; echo
timeout 200 python bench.py --with-background --no-cpu-baseline --no-gpu-baseline > $O/bench_withbg.json 2> $O/bench_withbg.err < /dev/null
tail -1 $O/bench_withbg.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j.get('with_background'))"
true
