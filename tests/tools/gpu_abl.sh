# ablation probes: gpu_abl.sh <out-name> <config> <weights> lib1 lib2 ...
set -x
mkdir -p gpurun_out/abl
O=$PWD/gpurun_out/abl/$1.jsonl; CFG=$2; WTS=$3; shift 3
: > $O
for lib in "$@"; do
  VMAPSTEP_LIBRARY=$PWD/$lib timeout 300 python tests/tools/abl_probe.py $CFG $WTS 2>&1 | grep "^{" >> $O
done
cat $O
