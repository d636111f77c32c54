# the usual check of a kernel change: GPU tier, the headline kernel back to back, the driver's command (no baselines)
set -x
T=${1:-chk}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
O=$PWD/gpurun_out/$T
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/pytest_gpu_tail.txt; cat $O/pytest_gpu_tail.txt
timeout 300 python tests/tools/abl_probe.py replica_room0_vmap f32 2>&1 | grep "^{" | tee $O/probe.jsonl
timeout 300 python tests/tools/abl_probe.py scannet0024_vmap bf16 2>&1 | grep "^{" | tee -a $O/probe.jsonl
timeout 300 python tests/tools/abl_probe.py background f32 2>&1 | grep "^{" | tee -a $O/probe.jsonl
timeout 300 python tests/tools/abl_probe.py imap_plumbing f32 2>&1 | grep "^{" | tee -a $O/probe.jsonl
timeout 300 python tests/tools/abl_probe.py stress_rank8 bf16 2>&1 | grep "^{" | tee -a $O/probe.jsonl
timeout 300 python tests/tools/abl_probe.py stress_256x64 bf16 2>&1 | grep "^{" | tee -a $O/probe.jsonl
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-baseline --no-pmc > $O/bench_20_5.json 2> $O/bench_20_5.err; echo rc=$?
python - <<PY
import json
j = json.loads([l for l in open("gpurun_out/$T/bench_20_5.json").read().splitlines() if l.startswith("{")][-1])
print("other", {k: (round(v.get("ms_per_step", 0), 5), round(v.get("kernel_ms", 0), 5)) for k, v in (j.get("other_configs") or {}).items()}, "frame", {k: round(v, 4) for k, v in (j.get("frame") or {}).items() if k.endswith("per_frame")})
print("value %.2f M ms/step %.5f kernel_ms %.5f frac %.3f" % (j["value"]/1e6, j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"]["frac"]), j.get("precision"), j.get("value_exact_fp32_kernel"))
PY
true
