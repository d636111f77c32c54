"""Measurement tool (not product): reads a rocprofv3 --kernel-trace CSV of `bench.py --timed-only` and prints, for the
last N step_main launches (the timed region), each launch's duration, the idle gap in front of it and the spacing of
consecutive steps - where the time of a short (--steps 20 --warmup 5) run goes.
Usage: python tests/tools/trace_gaps.py <kernel_trace.csv> [n_steps] > summary.json"""
import csv
import json
import sys

path = sys.argv[1]
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
mains = [i for i, r in enumerate(rows) if "step_main" in r[2]]
sel = mains[-n_steps:]
first = sel[0]
# the step_prep in front of the first timed step_main belongs to the timed call
start_i = first - 1 if first > 0 and "step_prep" in rows[first - 1][2] else first
t_begin = rows[start_i][0]
t_end = rows[-1][1] if "step_finalize" in rows[-1][2] else rows[sel[-1] + 1][1]
steps = []
prev_end = rows[start_i - 1][1] if start_i > 0 else t_begin
for k, i in enumerate(sel):
    s, e, name = rows[i]
    fin = rows[i + 1] if i + 1 < len(rows) else None
    steps.append({"step": k, "main_us": (e - s) / 1e3, "gap_before_main_us": (s - rows[i - 1][1]) / 1e3,
                  "finalize_us": (fin[1] - fin[0]) / 1e3 if fin else None,
                  "gap_before_finalize_us": (fin[0] - e) / 1e3 if fin else None,
                  "t_rel_us": (s - t_begin) / 1e3})
spacing = [(rows[sel[k + 1]][0] - rows[sel[k]][0]) / 1e3 for k in range(len(sel) - 1)]
out = {"trace": path, "timed_region_gpu_us": (t_end - t_begin) / 1e3,
       "idle_before_timed_region_us": (t_begin - prev_end) / 1e3,
       "prep_us": (rows[start_i][1] - rows[start_i][0]) / 1e3 if start_i != first else None,
       "step_spacing_us": {"first": spacing[0] if spacing else None, "median": sorted(spacing)[len(spacing) // 2] if spacing else None,
                           "last": spacing[-1] if spacing else None, "all": spacing},
       "main_us": {"first": steps[0]["main_us"], "median": sorted(s["main_us"] for s in steps)[len(steps) // 2], "last": steps[-1]["main_us"]},
       "steps": steps}
print(json.dumps(out, indent=1))
