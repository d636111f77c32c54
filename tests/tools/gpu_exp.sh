# item 8 measurement: partial gradients through the normal L2 write policy instead of nontemporal stores (libvmapstep_VS_EXP_TEMPORAL.so)
set -x
mkdir -p gpurun_out/r2k
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r2k
for V in base VS_EXP_TEMPORAL; do
  if [ $V = base ]; then export VMAPSTEP_LIBRARY=$R/vmap_amd/libvmapstep.so; else export VMAPSTEP_LIBRARY=$R/vmap_amd/libvmapstep_$V.so; fi
  timeout 200 python bench.py --no-cpu-baseline --no-gpu-baseline > $O/bench_$V.json 2> $O/bench_$V.err < /dev/null; tail -1 $O/bench_$V.json | head -c 250; echo
  cd /tmp
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$V -o p -- python $R/bench.py --timed-only --steps 400 --warmup 40 > $O/prof_$V.log 2>&1 < /dev/null
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $C | tr ' ' '_' | cut -c1-30)
    timeout 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/pmc_$V/$tag -o p -- python $R/tests/tools/run_steps.py replica_room0_vmap 40 > $O/pmc_${V}_$tag.log 2>&1 < /dev/null
  done
  cd $R
  for f in $O/prof_$V/*kernel_stats.csv; do [ -f "$f" ] && head -4 "$f" | cut -c1-150; done
done
true
