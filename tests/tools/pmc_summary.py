"""Fold the rocprofv3 --pmc passes written by tests/tools/gpu_pmc.sh (gpurun_out/pmc/*/p_counter_collection.csv) into
one JSON: per kernel, the mean of every counter per dispatch.  FETCH_SIZE / WRITE_SIZE are KiB per dispatch; the HBM
traffic of step_main applies the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts the 128-byte requests of
wide reads as 64 B: doubled; WRITE_SIZE taken as is).  Usage: python tests/tools/pmc_summary.py > profiles/<tag>_pmc_counters.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc", "*", "p_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        key = ("step_main_wp" if "step_main_wp" in name else "step_main_ws" if "step_main_ws" in name else "step_finalize_ws" if "step_finalize_ws" in name
               else "step_prep_ws" if "step_prep_ws" in name
               else "step_main_s32" if "step_main_s32" in name else "step_finalize_s32" if "step_finalize_s32" in name
               else "step_prep_s32" if "step_prep_s32" in name
               else "step_main_h32" if "step_main_h32" in name else "step_finalize_h32" if "step_finalize_h32" in name
               else "step_finalize" if "step_finalize" in name
               else "step_prep" if "step_prep" in name else None)
        if key:
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in acc.items()}
m = out.get("step_main_wp", out.get("step_main_ws", out.get("step_main_s32", out.get("step_main_h32", {}))))
import hashlib
try:
    lib_sha = hashlib.sha256(open(os.path.join(ROOT, "vmap_amd", "libvmapstep.so"), "rb").read()).hexdigest()
except OSError:
    lib_sha = None
notes = {"library_sha256": lib_sha, "workload": os.environ.get("PMC_WORKLOAD"), "units": "FETCH_SIZE/WRITE_SIZE in KiB per dispatch (rocprofv3), other counters summed over the chip per dispatch",
         "collection": "rocprofv3 --kernel-trace --pmc <group> in 7 separate passes over tests/tools/run_steps.py replica_room0_vmap 40"}
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    notes["hbm_traffic_bytes_per_launch_step_main"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
    if "step_main_ws" in out or "step_main_wp" in out:
        notes["hbm_traffic_bytes_per_launch_step_main_ws"] = notes["hbm_traffic_bytes_per_launch_step_main"]
        f = out.get("step_finalize_ws", {})
        if "FETCH_SIZE" in f and "WRITE_SIZE" in f:
            notes["hbm_traffic_bytes_per_launch_step_finalize_ws"] = (2.0 * f["FETCH_SIZE"] + f["WRITE_SIZE"]) * 1024.0
    notes["correction"] = "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950), WRITE_SIZE uncorrected"
if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
    notes["mfma_busy_fraction_of_occupied_simds"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * m["SQ_WAVE_CYCLES"]) if m.get("SQ_WAVE_CYCLES") else None
if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
    notes["lds_bank_conflict_fraction"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
out["_notes"] = notes
json.dump(out, sys.stdout, indent=1)
