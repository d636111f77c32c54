# round 3, call X: HBM counters of the background step with three-tile rounds; the whole GPU suite; bench lines
set -x
mkdir -p gpurun_out/r3x
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r3x
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest_gpu_tail.txt; cat $O/pytest_gpu_tail.txt
cd /tmp
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $R/gpurun_out/pmc/$C
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc/$C -o p -- python $R/tests/tools/run_steps.py background 40 > $O/pmc_$C.log 2>&1 < /dev/null
done
cd $R
python tests/tools/pmc_summary.py > $O/pmc_counters_background.json 2>$O/pmc_summary.err; tail -9 $O/pmc_counters_background.json
rm -rf gpurun_out/pmc
python bench.py --config background > $O/bench_background.json 2> $O/bench_background.err; tail -1 $O/bench_background.json | cut -c1-600
python bench.py --steps 20 --warmup 5 --with-background > $O/bench_20_5_withbg.json 2> $O/bench_withbg.err; tail -1 $O/bench_20_5_withbg.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['with_background'])"
true
