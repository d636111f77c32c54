# round 3, call K: split sampler (many workgroups per object) vs one workgroup per object; whole GPU suite; dist path with one rank
set -x
mkdir -p gpurun_out/r3k
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r3k
( timeout 900 python -m pytest tests -m gpu -q --maxfail=20 ) > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 300 python tests/tools/sampler_bench.py > $O/sampler_bench.json 2> $O/sampler_bench.err < /dev/null; tail -1 $O/sampler_bench.json | cut -c1-500
VMAP_BENCH_FORCE_DIST=1 MASTER_PORT=29511 timeout 300 python bench.py --with-background --steps 40 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $O/bench_forcedist_withbg.json 2> $O/bench_forcedist_withbg.err < /dev/null; tail -1 $O/bench_forcedist_withbg.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['with_background'], j['world'])"
tail -3 $O/bench_forcedist_withbg.err
true
