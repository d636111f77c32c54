# round 3, call U: three-tile rounds of step_main_ws (hidden 128): parity, then the background step A/B (new plan vs ws_flags = 4)
set -x
mkdir -p gpurun_out/r3u
O=$PWD/gpurun_out/r3u
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "three_tile or full_size_properties_of_the_other or seeded_shapes or generic_width_kernel or shared_background or background_on_second" 2>&1 | tail -15 > $O/pytest_three_tile.txt; cat $O/pytest_three_tile.txt
for rep in 1 2; do
python bench.py --config background --steps 400 --warmup 40 --no-cpu-baseline --no-gpu-baseline > $O/bench_bg_nt3_$rep.json 2>$O/bench_bg_nt3_$rep.err; tail -1 $O/bench_bg_nt3_$rep.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('nt3', j['ms_per_step'], j['roofline']['kernel'][:40], j['roofline']['kernel_ms'], j['roofline'].get('floor_us'))"
python bench.py --config background --steps 400 --warmup 40 --no-cpu-baseline --no-gpu-baseline --ws-flags 4 > $O/bench_bg_nt2_$rep.json 2>$O/bench_bg_nt2_$rep.err; tail -1 $O/bench_bg_nt2_$rep.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('nt2', j['ms_per_step'], j['roofline']['kernel'][:40], j['roofline']['kernel_ms'], j['roofline'].get('floor_us'))"
done
python bench.py --config background --weights bf16 --steps 400 --warmup 40 --no-cpu-baseline --no-gpu-baseline > $O/bench_bg_nt3_bf16.json 2>/dev/null; tail -1 $O/bench_bg_nt3_bf16.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('nt3 bf16', j['ms_per_step'], j['roofline']['kernel_ms'])"
python bench.py --config background --weights bf16 --steps 400 --warmup 40 --no-cpu-baseline --no-gpu-baseline --ws-flags 4 > $O/bench_bg_nt2_bf16.json 2>/dev/null; tail -1 $O/bench_bg_nt2_bf16.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('nt2 bf16', j['ms_per_step'], j['roofline']['kernel_ms'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_nt3 -o bg -- python $GRAFT_REPO_ROOT/bench.py --config background --steps 400 --warmup 40 --timed-only > $O/prof_nt3.log 2>&1
python - <<'PY'
import csv,glob,os
for f in glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r3u/prof_nt3/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:8]: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
true
