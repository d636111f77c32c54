# round 3, call 4E: hidden 256 on the bf16 pipe (step_main_ws<8>, eight waves): parity tests, the iMAP config line, kernel trace
set -x
mkdir -p gpurun_out/r4e
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r4e
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_query.py -x -q -k "hidden256 or imap or generic_width or parameter_image_kept or query" 2>&1 | tail -8 > $O/pytest_h256.txt; cat $O/pytest_h256.txt
python bench.py --config imap_plumbing --no-cpu-baseline > $O/bench_imap.json 2> $O/bench_imap.err; tail -1 $O/bench_imap.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('imap ws8', j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['kernel'][:60])"; tail -3 $O/bench_imap.err
python bench.py --config imap_plumbing --kernel wide --no-cpu-baseline --no-gpu-baseline > $O/bench_imap_wide.json 2>/dev/null; tail -1 $O/bench_imap_wide.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('imap wide', j['ms_per_step'], j['roofline']['kernel_ms'])"
python bench.py --config imap_plumbing --weights bf16 --no-cpu-baseline --no-gpu-baseline > $O/bench_imap_bf16.json 2>/dev/null; tail -1 $O/bench_imap_bf16.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('imap ws8 bf16', j['ms_per_step'], j['roofline']['kernel_ms'])"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bg -- python $R/bench.py --config imap_plumbing --steps 200 --warmup 20 --timed-only > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_imap.csv; head -4 $O/kernel_stats_imap.csv | cut -c1-150; rm -rf $O/prof
true
