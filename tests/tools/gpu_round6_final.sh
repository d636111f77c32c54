# round 6, verification on the final tree: GPU suite, smoke, the scored command (plain + under rocprofv3 --stats), the default run, the
# background step's kernel trace.  Everything the documents quote comes from the files this writes under gpurun_out/r6final/.
set -x
mkdir -p gpurun_out/r6final
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6final
R=$PWD
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu_tail.txt; cat $O/pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
VMAP_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_plain_gloo2.json 2> $O/bench_plain_gloo2.err; echo plain_gloo2 rc=$?
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof205 -o hl -- python $R/bench.py --steps 20 --warmup 5 --timed-only > $O/prof_run205.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profbg -o hl -- python $R/bench.py --config background --steps 200 --warmup 20 --timed-only > $O/prof_runbg.log 2>&1 < /dev/null
cd $R
for d in prof205 profbg; do for f in $O/$d/*/*kernel_stats.csv $O/$d/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" $O/kernel_stats_$d.csv && head -4 "$f" | cut -c1-160; done; done
python - <<'PY'
import json
for name in ("bench_20_5", "bench_default"):
    j = json.loads(open(f"gpurun_out/r6final/{name}.json").read().strip().splitlines()[-1])
    r = j["roofline"]
    print(name, "value %.2f M  ms/step %.5f (min %.5f max %.5f)  kernel_ms %.5f frac %.3f  vs_baseline %s" % (
        j["value"] / 1e6, j["ms_per_step"], j["repeats"]["ms_per_step_min"], j["repeats"]["ms_per_step_max"], r["kernel_ms"], r["frac"], j["vs_baseline"]))
    print("   exact fp32:", j.get("value_exact_fp32_kernel"))
    print("   six-product backward:", j.get("value_fp32_equivalent_backward"))
    print("   gpu_reference", {k: v for k, v in (j.get("gpu_reference_baseline") or {}).items() if k != "sample"})
    print("   gpu_port", {k: v for k, v in (j.get("gpu_eager_baseline") or {}).items() if k != "sample"})
    print("   cpu", {k: v for k, v in (j.get("cpu_baseline") or {}).items() if k != "sample"})
    print("   precision", j.get("precision"))
    for k, v in (j.get("other_configs") or {}).items():
        print("   ", k, {x: v.get(x) for x in ("ms_per_step", "rays_per_s", "kernel", "kernel_ms", "frac", "frac_of_executed_pipe", "error")})
    print("   frame", {k: v for k, v in (j.get("frame") or {}).items() if k.endswith("per_frame")})
    print("   traffic", r["traffic"], (r["traffic_source"] or "")[:60])
PY
true
