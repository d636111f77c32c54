set -x
mkdir -p gpurun_out/r6f
O=$PWD/gpurun_out/r6f
timeout 900 python -m pytest tests/test_gpu_bwd6.py -m gpu -q -x -s 2>&1 | grep -v Warning | tail -25 > $O/pytest_bwd6.txt; cat $O/pytest_bwd6.txt
for k in auto bwd6 f32; do timeout 300 python tests/tools/abl_probe.py replica_room0_vmap f32 $k 2>&1 | grep "^{" | tee -a $O/probe.jsonl; done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-pmc --no-other-configs --no-frame > $O/bench_20_5.json 2> $O/bench_20_5.err; echo rc=$?
python - <<PY
import json
j = json.loads([l for l in open("gpurun_out/r6f/bench_20_5.json").read().splitlines() if l.startswith("{")][-1])
print("value %.2f M ms/step %.5f kernel_ms %.5f" % (j["value"]/1e6, j["ms_per_step"], j["roofline"]["kernel_ms"]))
print("exact", j.get("value_exact_fp32_kernel")); print("bwd6", j.get("value_fp32_equivalent_backward")); print("precision", j.get("precision"))
PY
true
