set -x
mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2h
for V in VS_EXP_NOBARRIER VS_EXP_NOFIN; do
  VMAPSTEP_LIBRARY=$R/vmap_amd/libvmapstep_$V.so timeout 300 python bench.py --no-cpu-baseline --no-gpu-baseline > $O/bench_$V.json 2> $O/bench_$V.err
  VMAPSTEP_LIBRARY=$R/vmap_amd/libvmapstep_$V.so timeout 300 python tests/tools/phase_profile.py replica_room0_vmap split > $O/phases_$V.txt 2>&1
  tail -1 $O/bench_$V.json | head -c 250; echo; cat $O/phases_$V.txt | tail -18
done
