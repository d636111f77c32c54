# quick perf loop: phase clocks + short bench (no CPU baseline) + kernel-level rocprof durations
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python tests/tools/phase_profile.py > gpurun_out/phases.txt 2>&1; cat gpurun_out/phases.txt
timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1; tail -1 gpurun_out/bench_quick.log
timeout 300 python bench.py --config scannet0024_vmap --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_quick_scannet.log 2>&1; tail -1 gpurun_out/bench_quick_scannet.log
