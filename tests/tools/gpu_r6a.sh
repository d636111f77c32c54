# round 6, first call: the GPU tier on the fresh tree, the driver's command, plain `bench.py --gpus 2` (gloo dry run) and 8 ranks the same way
set -x
mkdir -p gpurun_out/r6a
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6a
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/pytest_gpu_tail.txt; cat $O/pytest_gpu_tail.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; echo rc=$?
VMAP_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_plain_gloo2.json 2> $O/bench_plain_gloo2.err; echo rc=$?
VMAP_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 > $O/bench_plain_gloo8.json 2> $O/bench_plain_gloo8.err; echo rc=$?
timeout 100 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_plain_rccl2_refused.out 2> $O/bench_plain_rccl2_refused.err; echo rc=$? | tee $O/bench_plain_rccl2_refused.rc
python - <<'PY'
import json
for name in ("bench_20_5", "bench_plain_gloo2", "bench_plain_gloo8"):
    try:
        j = json.loads([l for l in open(f"gpurun_out/r6a/{name}.json").read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        print(name, "no line", e); continue
    print(name, "n_gpus", j["n_gpus"], "value %.2f M ms/step %.5f" % (j["value"]/1e6, j["ms_per_step"]), j["repeats"].get("ms_per_step_incl_closing_barrier"), j["world"].get("region_costs"))
    print("  summary", json.dumps(j.get("summary"))[:1500])
PY
true
