# round 3, call I: the bound frame call replayed as a hipGraph (device-resident AdamW step count): whole GPU suite, headline bench
# as the driver runs it with and without the graph, kernel stats
set -x
mkdir -p gpurun_out/r3i
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r3i
( timeout 900 python -m pytest tests -m gpu -q --maxfail=20 ) > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
for v in "" "--no-graph" "" "--no-graph"; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline $v > $O/bench_20_5$v.json 2> $O/bench_20_5$v.err < /dev/null; tail -1 $O/bench_20_5$v.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['frame_call'][:60])"
done
for v in "" "--no-graph"; do
timeout 300 python bench.py --no-cpu-baseline --no-gpu-baseline $v > $O/bench_400$v.json 2> $O/bench_400$v.err < /dev/null; tail -1 $O/bench_400$v.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['frame_call'][:60])"
done
timeout 300 python bench.py --config scannet0024_vmap --no-cpu-baseline --no-gpu-baseline > $O/bench_scannet.json 2>&1 < /dev/null; tail -1 $O/bench_scannet.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
timeout 300 python bench.py --config background --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $O/bench_background.json 2>&1 < /dev/null; tail -1 $O/bench_background.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof205 -o hl -- python $R/bench.py --steps 20 --warmup 5 --timed-only > $O/prof_run205.log 2>&1 < /dev/null
cd $R
head -5 $O/prof205/hl_kernel_stats.csv | cut -c1-170; tail -2 $O/prof_run205.log | cut -c1-300
true
