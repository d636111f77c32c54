set -x
mkdir -p gpurun_out/r6i
O=$PWD/gpurun_out/r6i
# the N > 1 code path of bench.py on the REAL RCCL backend with one rank (the only way on a 1-GPU box), background legs included
VMAP_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --with-background > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err; echo rc=$?
python - <<PY
import json
j = json.loads([l for l in open("gpurun_out/r6i/bench_rccl_world1.json").read().splitlines() if l.startswith("{")][-1])
print("n_gpus", j["n_gpus"], "value %.2f M" % (j["value"]/1e6), "world", {k: j["world"][k] for k in ("rccl", "backend", "region_costs")})
print("with_background", json.dumps(j.get("with_background"))[:1200])
PY
for cw in "background bf16" "background f32" "background_rank8 bf16"; do timeout 300 python tests/tools/abl_probe.py $cw 2>&1 | grep "^{"; done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bg or 128 or bf16" 2>&1 | tail -3
true
