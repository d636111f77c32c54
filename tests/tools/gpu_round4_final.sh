# round 4, end-of-round verification on the final tree: GPU suite, smoke, the scored command (plain, under rocprofv3 --stats, with
# traffic observed in the same run), the default run, the other shapes the documents quote
set -x
mkdir -p gpurun_out/r4final
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4final
R=$PWD
python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu_tail.txt; cat $O/pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof205 -o hl -- python $R/bench.py --steps 20 --warmup 5 --timed-only > $O/prof_run205.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profdef -o hl -- python $R/bench.py --timed-only > $O/prof_rundef.log 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profbg -o hl -- python $R/bench.py --config background --steps 200 --warmup 20 --timed-only > $O/prof_runbg.log 2>&1 < /dev/null
cd $R
for d in prof205 profdef profbg; do for f in $O/$d/*/*kernel_stats.csv $O/$d/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" $O/kernel_stats_$d.csv && head -4 "$f" | cut -c1-160; done; done
for cfg in "background f32" "background bf16" "stress_256x64 bf16" "stress_256x64 f32" "stress_rank8 bf16" "imap_plumbing f32" "imap_full f32" "scannet0024_vmap f32" "scannet0024_vmap bf16" "background_rank8 f32"; do
  set -- $cfg
  timeout 300 python bench.py --config $1 --weights $2 --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-baseline --no-frame > $O/bench_$1_$2.json 2>$O/bench_$1_$2.err
done
bash tests/tools/gpu_bench_with_pmc.sh replica_room0_vmap f32 20 5 > $O/bench_pmc_run.log 2>&1
cp gpurun_out/bench_pmc/bench.json $O/bench_20_5_observed_traffic.json; cp gpurun_out/bench_pmc/pmc_counters.json $O/pmc_counters_fetch_write_same_run.json
python tests/tools/frame_priority_probe.py > $O/frame_priority_probe.txt 2>&1; cp gpurun_out/frame_priority_probe.json $O/
python - <<'PY'
import json, glob, os
O = "gpurun_out/r4final"
for f in sorted(glob.glob(O + "/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        r = j.get("roofline", {})
        print(os.path.basename(f), "ms/step %.4f" % j["ms_per_step"], "value %.2f M" % (j["value"] / 1e6), "kernel_ms %.4f" % r.get("kernel_ms", 0), "frac %.3f" % r.get("frac", 0),
              "frame", (j.get("frame") or {}).get("two_streams_ms_per_frame"))
    except Exception as e:
        print(f, "unreadable", e)
PY
true
