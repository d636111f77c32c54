# round 2, call D: hand-interleaved matrix / VALU streams
set -x
mkdir -p gpurun_out/r2e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2e
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -40 $O/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --no-gpu-baseline > $O/bench_400_auto.json 2> $O/bench_400_auto.err; echo "bench rc=$?" | tee -a $O/summary.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_5_auto.json 2> $O/bench_20_5_auto.err
timeout 300 python bench.py --config scannet0024_vmap --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $O/bench_scannet_auto.json 2>&1
timeout 300 python bench.py --config scannet0024_vmap --weights bf16 --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $O/bench_scannet_bf16_auto.json 2>&1
timeout 300 python tests/tools/phase_profile.py replica_room0_vmap split > $O/phases_auto.txt 2>&1
cat $O/summary.txt
tail -1 $O/bench_400_auto.json | head -c 300; echo
cat $O/phases_auto.txt
