# round 3, call W: step_finalize_ws with 4 / 8 / 16 row groups (waves) per block at 200 rows (the background step's new plan)
set -x
mkdir -p gpurun_out/r3w
O=$PWD/gpurun_out/r3w
cp vmap_amd/libvmapstep.so /tmp/lib_fg4.so
cd /tmp && export TMPDIR=/tmp
for fg in 4 8 16; do
  if [ $fg = 4 ]; then cp /tmp/lib_fg4.so $GRAFT_REPO_ROOT/vmap_amd/libvmapstep.so; else cp $GRAFT_REPO_ROOT/tests/tools/exp/_variants/libvmapstep_fg$fg.so $GRAFT_REPO_ROOT/vmap_amd/libvmapstep.so; fi
  python $GRAFT_REPO_ROOT/bench.py --config background --steps 400 --warmup 40 --timed-only > $O/bench_fg$fg.json 2>/dev/null; tail -1 $O/bench_fg$fg.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fg$fg -o bg -- python $GRAFT_REPO_ROOT/bench.py --config background --steps 400 --warmup 40 --timed-only > $O/prof_fg$fg.log 2>&1
  find $O/prof_fg$fg -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_fg$fg.csv
  head -3 $O/kernel_stats_fg$fg.csv | cut -c1-160
  rm -rf $O/prof_fg$fg
done
cp /tmp/lib_fg4.so $GRAFT_REPO_ROOT/vmap_amd/libvmapstep.so
cd $GRAFT_REPO_ROOT && timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "generic_width_kernel or shared_background" 2>&1 | tail -3
true
