# round 3, call Z: finalize with 8 row groups x 128 quads per block (2 KiB contiguous per row): GPU suite, background / stress / headline lines
set -x
mkdir -p gpurun_out/r3z
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r3z
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/pytest_gpu_tail.txt; cat $O/pytest_gpu_tail.txt
python bench.py --config background --no-cpu-baseline > $O/bench_background.json 2> $O/bench_background.err; tail -1 $O/bench_background.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('background', j['ms_per_step'], j['roofline']['kernel_ms'])"
python bench.py --config background --weights bf16 --no-cpu-baseline --no-gpu-baseline > $O/bench_background_bf16.json 2>/dev/null; tail -1 $O/bench_background_bf16.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('background bf16', j['ms_per_step'], j['roofline']['kernel_ms'])"
python bench.py --config stress_256x64 --steps 60 --warmup 10 --no-cpu-baseline --no-gpu-baseline > $O/bench_stress.json 2>/dev/null; tail -1 $O/bench_stress.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('stress', j['ms_per_step'], j['roofline']['kernel_ms'])"
python bench.py --config stress_256x64 --weights bf16 --steps 60 --warmup 10 --no-cpu-baseline --no-gpu-baseline > $O/bench_stress_bf16.json 2>/dev/null; tail -1 $O/bench_stress_bf16.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('stress bf16', j['ms_per_step'], j['roofline']['kernel_ms'])"
python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2>/dev/null; tail -1 $O/bench_20_5.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('headline', j['value'], j['ms_per_step'])"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bg -- python $R/bench.py --config background --steps 400 --warmup 40 --timed-only > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_background.csv; head -4 $O/kernel_stats_background.csv | cut -c1-150; rm -rf $O/prof
true
