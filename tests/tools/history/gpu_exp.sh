# A/B: ray ground truth prefetched in front of the MLP forward (shipped) vs loaded at the compositing (-DVS_NO_RAY_PREFETCH)
set -x
mkdir -p gpurun_out/r2q
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/r2q
for rep in 1 2; do
for V in base VS_NO_RAY_PREFETCH; do
  if [ $V = base ]; then export VMAPSTEP_LIBRARY=$R/vmap_amd/libvmapstep.so; else export VMAPSTEP_LIBRARY=$R/vmap_amd/libvmapstep_$V.so; fi
  timeout 200 python bench.py --no-cpu-baseline --no-gpu-baseline > $O/bench_${V}_$rep.json 2> $O/bench_${V}_$rep.err < /dev/null; tail -1 $O/bench_${V}_$rep.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$V', j['value'], j['ms_per_step'], j['roofline']['kernel_ms'])"
done
done
export VMAPSTEP_LIBRARY=$R/vmap_amd/libvmapstep.so
timeout 120 python tests/tools/phase_profile.py replica_room0_vmap split > $O/phases_base.txt 2>&1 < /dev/null; tail -17 $O/phases_base.txt
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x ) > $O/pytest.log 2>&1 < /dev/null; grep -n "passed\|failed" $O/pytest.log | tail -1
