# round 6, call 2: ablation probes of step_main_s32's backward (measurement builds, results wrong on purpose), the stand-alone
# erratum reproducer, phase clocks of the product kernels
set -x
mkdir -p gpurun_out/r6b
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6b
for lib in vmap_amd/libvmapstep.so tests/tools/libvmapstep_abl1.so tests/tools/libvmapstep_abl2.so tests/tools/libvmapstep_abl3.so; do
  VMAPSTEP_LIBRARY=$PWD/$lib timeout 300 python tests/tools/abl_probe.py replica_room0_vmap f32 2>&1 | grep "^{" >> $O/abl_s32.jsonl
done
cat $O/abl_s32.jsonl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_erratum_min tests/tools/pk_erratum_min.hip 2>/dev/null && timeout 120 /tmp/pk_erratum_min > $O/pk_erratum_min.txt; cat $O/pk_erratum_min.txt
timeout 300 python tests/tools/phase_profile.py > $O/phase_clocks_s32.txt 2>&1; tail -30 $O/phase_clocks_s32.txt
true
