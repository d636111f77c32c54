# several configs x several libraries: gpu_abl2.sh <out-name> "<cfg:weights> ..." lib1 lib2 ...
set -x
mkdir -p gpurun_out/abl
O=$PWD/gpurun_out/abl/$1.jsonl; CFGS=$2; shift 2
: > $O
for cw in $CFGS; do
  for lib in "$@"; do
    VMAPSTEP_LIBRARY=$PWD/$lib timeout 300 python tests/tools/abl_probe.py ${cw%%:*} ${cw##*:} 2>&1 | grep "^{" >> $O
  done
done
cat $O
