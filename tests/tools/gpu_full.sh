mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/pytest_gpu.log
for v in 0 1; do
  for C in replica_room0_vmap scannet0024_vmap; do
  VMAPSTEP_CARRY=$v timeout 300 python bench.py --config $C --steps 600 --warmup 60 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('carry=$v', '$C', round(d['value']/1e6,2), 'M rays/s', round(d['ms_per_step']*1e3,2), 'us/step', 'kernel', round(d['roofline']['kernel_ms']*1e3,2), 'us')"
  done
done
