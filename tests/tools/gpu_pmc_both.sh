# fresh counter passes for the two kernels that bound a frame (round 6): headline step_main_s32 and the background's step_main_ws<4>
set -x
export TMPDIR=/tmp
for cfg in replica_room0_vmap background; do
  rm -rf gpurun_out/pmc
  PMC_CONFIG=$cfg bash tests/tools/gpu_pmc.sh > gpurun_out/pmc_$cfg.log 2>&1
  PMC_WORKLOAD="tests/tools/run_steps.py $cfg 40 f32" python tests/tools/pmc_summary.py > gpurun_out/pmc_counters_$cfg.json
  rm -rf gpurun_out/pmc
done
python - <<'PY'
import json
for cfg in ("replica_room0_vmap", "background"):
    j = json.load(open(f"gpurun_out/pmc_counters_{cfg}.json"))
    print(cfg, json.dumps({k: v for k, v in j.items() if k.startswith("step_main") or k == "_notes"})[:1500])
PY
