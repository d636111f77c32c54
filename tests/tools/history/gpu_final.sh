# end-of-round check: what the driver runs (smoke, default bench, the 20 / 5 command)
set -x
mkdir -p gpurun_out/final
export TMPDIR=/tmp
O=$PWD/gpurun_out/final
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1 < /dev/null; tail -3 $O/smoke.log
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err < /dev/null; tail -1 $O/bench_default.json | head -c 400; echo; tail -4 $O/bench_default.err
( time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_20_5.json 2> $O/bench_20_5.err < /dev/null; tail -1 $O/bench_20_5.json | head -c 300; echo; tail -4 $O/bench_20_5.err
