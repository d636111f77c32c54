# round 3, call 4B: tree after the single-round specialisation: GPU suite, smoke, background lines, kernel trace
set -x
mkdir -p gpurun_out/r4b
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r4b
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest_gpu_tail.txt; cat $O/pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --config background --no-cpu-baseline > $O/bench_background.json 2> $O/bench_background.err; tail -1 $O/bench_background.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('background', j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['kernel'][:30])"
python bench.py --config background --weights bf16 --no-cpu-baseline --no-gpu-baseline > $O/bench_background_bf16.json 2>/dev/null; tail -1 $O/bench_background_bf16.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('background bf16', j['ms_per_step'], j['roofline']['kernel_ms'])"
for c in background_rank4 background_rank8; do python bench.py --config $c --no-cpu-baseline --no-gpu-baseline > $O/bench_$c.json 2>/dev/null; tail -1 $O/bench_$c.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$c', j['ms_per_step'], j['roofline']['kernel_ms'])"; done
python tests/tools/bg_chain_bench.py 2>/dev/null | tail -1 > $O/bg_chain_bench.json; cat $O/bg_chain_bench.json | cut -c1-700
python tests/tools/frame_bench.py 2>/dev/null | tail -1 > $O/frame_bench.json; cat $O/frame_bench.json | cut -c1-400
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bg -- python $R/bench.py --config background --steps 400 --warmup 40 --timed-only > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_background.csv; head -4 $O/kernel_stats_background.csv | cut -c1-150; rm -rf $O/prof
true
