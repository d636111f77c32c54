# round 5, first GPU pass: the GPU suite on the reorganised tests (product library first), smoke, the scored command with the
# reference as live baseline, its kernel trace
set -x
mkdir -p gpurun_out/r5a
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5a
R=$PWD
ls -la oracle/_ref > $O/ref_files.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu_tail.txt; cat $O/pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; tail -3 $O/bench_20_5.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof205 -o hl -- python $R/bench.py --steps 20 --warmup 5 --timed-only > $O/prof_run205.log 2>&1 < /dev/null
cd $R
for f in $O/prof205/*/*kernel_stats.csv $O/prof205/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" $O/kernel_stats_20_5.csv && head -5 "$f" | cut -c1-160; done
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r5a/bench_20_5.json").read().strip().splitlines()[-1])
print("value %.2f M  ms/step %.5f  repeats %s" % (j["value"] / 1e6, j["ms_per_step"], j["repeats"]["ms_per_step"]))
print("vs_baseline", j["vs_baseline"], "gpu_reference", j.get("gpu_reference_baseline"))
print("cpu_baseline", j.get("cpu_baseline"))
print("precision", j.get("precision"))
for k, v in (j.get("other_configs") or {}).items():
    print(k, {x: v.get(x) for x in ("ms_per_step", "rays_per_s", "kernel", "kernel_ms", "frac", "frac_of_executed_pipe", "error")})
print("frame", {k: v for k, v in (j.get("frame") or {}).items() if k.endswith("per_frame")})
print("traffic", j["roofline"]["traffic"], j["roofline"]["traffic_source"][:80])
PY
true
