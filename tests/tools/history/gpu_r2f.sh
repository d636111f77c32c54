# round 2, call F: full GPU suite after the ABI additions (prepared step index, adamw_apply, sampler sizes) + bench with the shared background model
set -x
mkdir -p gpurun_out/r2f
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2f
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -15 $O/pytest_gpu.log
timeout 300 python bench.py --with-background --no-cpu-baseline --no-gpu-baseline > $O/bench_bg.json 2> $O/bench_bg.err; echo "bench bg rc=$?" | tee -a $O/summary.txt
VMAP_BENCH_FORCE_DIST=1 MASTER_PORT=29611 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --with-background --no-cpu-baseline --no-gpu-baseline > $O/bench_bg_dist1.json 2> $O/bench_bg_dist1.err; echo "bench bg dist1 rc=$?" | tee -a $O/summary.txt
timeout 300 python bench.py --config background --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $O/bench_background.json 2>&1
cat $O/summary.txt
tail -1 $O/bench_bg.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['with_background'], j['world'])"
tail -1 $O/bench_bg_dist1.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['with_background'], j['world'])"
tail -3 $O/bench_bg.err
