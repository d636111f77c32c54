# round 2, call B: the split-bf16 kernel on the device: parity, bench A/B against the exact-fp32 kernel, phase clocks, rocprof stats
set -x
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2b
( time timeout 600 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -40 $O/pytest_gpu.log
( time python __graft_entry__.py --smoke ) > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
for K in auto f32; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --kernel $K --no-cpu-baseline --no-gpu-baseline > $O/bench_20_5_$K.json 2> $O/bench_20_5_$K.err; echo "bench 20/5 $K rc=$?" | tee -a $O/summary.txt
  timeout 300 python bench.py --kernel $K --no-cpu-baseline --no-gpu-baseline > $O/bench_400_$K.json 2> $O/bench_400_$K.err; echo "bench 400 $K rc=$?" | tee -a $O/summary.txt
  timeout 300 python bench.py --config scannet0024_vmap --steps 200 --warmup 20 --kernel $K --no-cpu-baseline --no-gpu-baseline > $O/bench_scannet_$K.json 2>&1
  timeout 300 python bench.py --config scannet0024_vmap --weights bf16 --steps 200 --warmup 20 --kernel $K --no-cpu-baseline --no-gpu-baseline > $O/bench_scannet_bf16_$K.json 2>&1
  timeout 300 python tests/tools/phase_profile.py replica_room0_vmap $([ $K = f32 ] && echo f32 || echo split) > $O/phases_$K.txt 2>&1
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $O/rocprof_bench.log 2>&1; echo "rocprof rc=$?" | tee -a $O/summary.txt
cd $R
F=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $F $O/kernel_stats.csv; rm -rf $O/prof
head -8 $O/kernel_stats.csv
cat $O/summary.txt
for K in auto f32; do tail -1 $O/bench_400_$K.json | head -c 400; echo; tail -1 $O/bench_20_5_$K.json | head -c 300; echo; done
cat $O/phases_auto.txt
