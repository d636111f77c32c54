export TMPDIR=/tmp
for rep in 1 2 3; do
for v in 0 1; do
  VMAPSTEP_CARRY=$v timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('carry=$v', round(d['value']/1e6,2), 'M rays/s', round(d['ms_per_step']*1e3,2), 'us/step', 'kernel', round(d['roofline']['kernel_ms']*1e3,2), 'us')"
done
done
