# round 2, call A: design probes for the split-bf16 kernel + where a short (--steps 20 --warmup 5) bench run loses its time
set -x
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2a
( time timeout 120 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
timeout 300 $R/tests/tools/bf16_probe.out > $O/bf16_probe.jsonl 2> $O/bf16_probe.err; echo "bf16_probe rc=$?" | tee -a $O/summary.txt
timeout 300 python tests/tools/gap_probe.py > $O/gap_probe.json 2> $O/gap_probe.err; echo "gap_probe rc=$?" | tee -a $O/summary.txt
for i in 1 2; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5_run$i.json 2> $O/bench_20_5_run$i.err; echo "bench 20/5 run$i rc=$?" | tee -a $O/summary.txt
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --unbound --no-cpu-baseline --no-gpu-baseline > $O/bench_20_5_unbound.json 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --preheat-ms 300 --no-cpu-baseline --no-gpu-baseline > $O/bench_20_5_preheat300.json 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --preheat-ms 50 --no-cpu-baseline --no-gpu-baseline > $O/bench_20_5_preheat50.json 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-gpu-baseline > $O/bench_400_40.json 2>&1
cd /tmp
for V in cold preheat; do
  PH=0; [ $V = preheat ] && PH=300
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$V -o t -- python $R/bench.py --steps 20 --warmup 5 --timed-only --preheat-ms $PH > $O/trace_$V.log 2>&1
  F=$(find $O/trace_$V -name '*kernel_trace.csv' | head -1)
  python $R/tests/tools/trace_gaps.py $F 20 > $O/trace_gaps_$V.json 2> $O/trace_gaps_$V.err
  rm -rf $O/trace_$V
done
cd $R
cat $O/summary.txt
head -c 1500 $O/gap_probe.json
tail -3 $O/bench_20_5_run1.json | head -c 600
