# round 5: fresh counter pass of configs[4] on one GPU (step_main_wp<2>, bf16 weights): fabric-side bytes (FETCH_SIZE / WRITE_SIZE, separate
# passes), L2 hit / miss, matrix-pipe busy; the kernel's own duration from a kernel trace of the same command
set -x
mkdir -p gpurun_out/r5b
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5b
R=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/tests/tools/run_steps.py stress_256x64 40 bf16 > $O/trace.log 2>&1 < /dev/null
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$tag -o p -- python $R/tests/tools/run_steps.py stress_256x64 40 bf16 > $O/$tag.log 2>&1 < /dev/null
  echo "$tag rc=$?"
done
cd $R
python - <<'PY'
import csv, glob, json, os
from collections import defaultdict
O = "gpurun_out/r5b"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(O + "/*/p_counter_collection.csv") + glob.glob(O + "/*/*/p_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        if "step_" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
for f in glob.glob(O + "/trace/*kernel_stats.csv") + glob.glob(O + "/trace/*/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        if "step_" in row["Name"]:
            out.setdefault(row["Name"].split("(")[0][:60], {})["AverageNs"] = float(row["AverageNs"])
json.dump(out, open(O + "/pmc_counters_stress_wp_bf16.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
true
