# whole-step A/B of libraries on several shapes: gpu_ab_steps.sh <out-name> "<cfg:weights> ..." lib1 lib2 ...   (two passes, interleaved)
set -x
mkdir -p gpurun_out/abl
O=$PWD/gpurun_out/abl/$1.jsonl; CFGS=$2; shift 2
: > $O
for rep in 1 2; do
for cw in $CFGS; do
  for lib in "$@"; do
    VMAPSTEP_LIBRARY=$PWD/$lib timeout 300 python bench.py --config ${cw%%:*} --weights ${cw##*:} --steps 200 --warmup 20 --timed-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(json.dumps({'lib': '$lib', 'config': '${cw%%:*}', 'weights': '${cw##*:}', 'ms_per_step': j['ms_per_step'], 'min': j['repeats']['ms_per_step_min'], 'max': j['repeats']['ms_per_step_max']}))" >> $O
  done
done
done
cat $O
