# round 3, call Y: step_finalize_ws geometry variants at 200 rows: quads per block 64 / 128 / 256 (contiguous bytes per row and block),
# row groups 4 / 8, nontemporal row loads
set -x
mkdir -p gpurun_out/r3y
O=$PWD/gpurun_out/r3y
cp vmap_amd/libvmapstep.so /tmp/lib_base.so
cd /tmp && export TMPDIR=/tmp
for v in base fg4_q128 fg8_q128 fg4_q256 fg4_q64nt; do
  if [ $v = base ]; then cp /tmp/lib_base.so $GRAFT_REPO_ROOT/vmap_amd/libvmapstep.so; else cp $GRAFT_REPO_ROOT/tests/tools/exp/_variants/libvmapstep_$v.so $GRAFT_REPO_ROOT/vmap_amd/libvmapstep.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o bg -- python $GRAFT_REPO_ROOT/bench.py --config background --steps 400 --warmup 40 --timed-only > $O/prof_$v.log 2>&1
  find $O/prof_$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$v.csv
  echo "== $v"; grep finalize $O/kernel_stats_$v.csv | cut -c1-140; tail -1 $O/prof_$v.log | cut -c1-120
  rm -rf $O/prof_$v
done
cp /tmp/lib_base.so $GRAFT_REPO_ROOT/vmap_amd/libvmapstep.so
true
