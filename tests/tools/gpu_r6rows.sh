# round 6d: block-native gradient rows (step_main_ws / _wp + step_finalize_ws).  GPU suite on the new product library, then whole-step A/B
# against the library of the commit before (tests/tools/libvmapstep_base.so) on every shape these kernels serve.
set -x
mkdir -p gpurun_out/r6rows
O=$PWD/gpurun_out/r6rows
export TMPDIR=/tmp
[ -n "$SKIP_TESTS" ] || { timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/pytest_gpu_tail.txt; cat $O/pytest_gpu_tail.txt; }
: > $O/ab.jsonl
for rep in 1 2; do
for cw in background:f32 background:bf16 stress_rank8:bf16 stress_256x64:bf16 imap_full:f32 imap_plumbing:f32 background_rank8:f32; do
  for lib in tests/tools/libvmapstep_base.so vmap_amd/libvmapstep.so; do
    VMAPSTEP_LIBRARY=$PWD/$lib timeout 300 python bench.py --config ${cw%%:*} --weights ${cw##*:} --steps 200 --warmup 20 --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(json.dumps({'lib': '$lib', 'config': '${cw%%:*}', 'weights': '${cw##*:}', 'ms_per_step': j['ms_per_step'], 'min': j['repeats']['ms_per_step_min'], 'max': j['repeats']['ms_per_step_max']}))" >> $O/ab.jsonl
  done
done
done
cat $O/ab.jsonl
: > $O/kern.jsonl
for lib in tests/tools/libvmapstep_base.so vmap_amd/libvmapstep.so tests/tools/libvmapstep_ah2.so; do VMAPSTEP_LIBRARY=$PWD/$lib timeout 300 python tests/tools/abl_probe.py imap_full f32 2>&1 | grep "^{" >> $O/kern.jsonl; VMAPSTEP_LIBRARY=$PWD/$lib timeout 300 python tests/tools/abl_probe.py imap_full bf16 2>&1 | grep "^{" >> $O/kern.jsonl; done
for cw in background:f32 background:bf16 stress_rank8:bf16 stress_256x64:bf16; do
  for lib in tests/tools/libvmapstep_base.so vmap_amd/libvmapstep.so; do
    VMAPSTEP_LIBRARY=$PWD/$lib timeout 300 python tests/tools/abl_probe.py ${cw%%:*} ${cw##*:} 2>&1 | grep "^{" >> $O/kern.jsonl
  done
done
cat $O/kern.jsonl
