# forward-only prototype on 16-point tiles (two tiles per SIMD) against step_main_s32's forward-only instantiation
set -x
mkdir -p gpurun_out/s16
export TMPDIR=/tmp
O=$PWD/gpurun_out/s16
R=$PWD
cd /tmp
for cfg in replica_room0_vmap scannet0024_vmap; do
for k in auto s16; do
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${cfg}_$k -o r -- python $R/tests/tools/render_steps.py $cfg 200 $k > $O/run_${cfg}_$k.log 2>&1 < /dev/null
grep "max |depth" $O/run_${cfg}_$k.log
for f in $O/prof_${cfg}_$k/*kernel_stats.csv; do [ -f "$f" ] && grep "step_main" "$f" | cut -c1-150; done
done
done
true
