# A/B of measurement builds (tests/tools/build_variant.py) on one box: bench.py of one configuration per library
# usage: gpu_variants.sh <config> <weights> <tag> [<tag> ...]   ("product" = vmap_amd/libvmapstep.so)
CFG=$1; WTS=$2; shift 2
mkdir -p gpurun_out/variants
for rep in 1 2; do
for tag in "$@"; do
  if [ "$tag" = product ]; then unset VMAPSTEP_LIBRARY; else export VMAPSTEP_LIBRARY=$PWD/tests/tools/libvmapstep_$tag.so; fi
  python bench.py --config $CFG --weights $WTS --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline --no-frame --profile-reps 100 > gpurun_out/variants/${CFG}_${WTS}_$tag.json 2>gpurun_out/variants/${CFG}_${WTS}_$tag.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/variants/${CFG}_${WTS}_$tag.json").read().strip().splitlines()[-1])
print("$CFG $WTS $tag rep$rep ms/step %.4f kernel_ms %.4f" % (j["ms_per_step"], j["roofline"]["kernel_ms"]))
PY
done
done
