# round 3, call M: end-of-round verification - whole GPU suite, smoke, the driver's bench command (+ kernel stats, HBM counters), the
# N > 1 code path with one rank over RCCL, the other shapes, a whole frame
set -x
mkdir -p gpurun_out/r3m
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r3m
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err < /dev/null; tail -1 $O/bench_20_5.json | head -c 300; echo
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err < /dev/null; tail -1 $O/bench_default.json | head -c 200; echo
VMAP_BENCH_FORCE_DIST=1 timeout 300 python bench.py --with-background --steps 40 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $O/bench_forcedist_withbg.json 2> $O/bench_forcedist_withbg.err < /dev/null; tail -1 $O/bench_forcedist_withbg.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['with_background']['ms_per_step'], j['world'])"; tail -2 $O/bench_forcedist_withbg.err
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err < /dev/null; tail -1 $O/bench_torchrun1.json | head -c 160; echo
for c in scannet0024_vmap stress_256x64 imap_plumbing; do
for w in f32 bf16; do
timeout 300 python bench.py --config $c --weights $w --steps 60 --warmup 20 --no-cpu-baseline --no-gpu-baseline > $O/bench_${c}_$w.json 2>&1 < /dev/null; tail -1 $O/bench_${c}_$w.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$c $w', j['value'], j['ms_per_step'], j['roofline']['kernel'][:40], j['roofline']['frac'])"
done
done
timeout 200 python tests/tools/frame_bench.py > $O/frame_bench.json 2>&1; tail -1 $O/frame_bench.json | cut -c1-400
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof205 -o hl -- python $R/bench.py --steps 20 --warmup 5 --timed-only > $O/prof_run205.log 2>&1 < /dev/null
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $R/gpurun_out/pmc/$C
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc/$C -o p -- python $R/tests/tools/run_steps.py replica_room0_vmap 40 > $O/pmc_$C.log 2>&1 < /dev/null
done
cd $R
head -5 $O/prof205/hl_kernel_stats.csv | cut -c1-170
python tests/tools/pmc_summary.py > $O/pmc_counters.json 2>$O/pmc_summary.err; tail -6 $O/pmc_counters.json
true
