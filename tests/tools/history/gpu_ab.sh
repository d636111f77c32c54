mkdir -p gpurun_out; export TMPDIR=/tmp
cp vmap_amd/libvmapstep.so /tmp/keep.so
for rep in 1 2 3; do
for v in ${VARIANTS:-base cold}; do
  cp vmap_amd/_exp/$v.so vmap_amd/libvmapstep.so
  timeout 300 python bench.py --steps 600 --warmup 60 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v', round(d['value']/1e6,2), d['ms_per_step'])"
done
done
cp /tmp/keep.so vmap_amd/libvmapstep.so
