# round 3, final tree: GPU suite, smoke, the scored line (20/5 as the driver runs it), background / iMAP lines
set -x
mkdir -p gpurun_out/r4m
O=$PWD/gpurun_out/r4m
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest_gpu_tail.txt; cat $O/pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2>/dev/null; tail -1 $O/bench_20_5.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('headline 20/5', j['value'], j['ms_per_step'], j['roofline']['frac'])"
python bench.py > $O/bench_default.json 2>/dev/null; tail -1 $O/bench_default.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('headline default', j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['cpu_baseline']['value'])"
python bench.py --config background --no-cpu-baseline > $O/bench_background.json 2>/dev/null; tail -1 $O/bench_background.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('background', j['ms_per_step'], j['roofline']['kernel_ms'])"
python bench.py --config imap_plumbing --no-cpu-baseline --no-gpu-baseline > $O/bench_imap.json 2>/dev/null; tail -1 $O/bench_imap.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('imap', j['ms_per_step'], j['roofline']['kernel_ms'])"
python bench.py --steps 20 --warmup 5 --with-background > $O/bench_withbg.json 2>/dev/null; tail -1 $O/bench_withbg.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('with bg', j['value'], j['with_background']['ms_per_step'])"
true
