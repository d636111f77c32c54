"""Measurement tool: the per-step launch chain of the ray-sharded background model (parallel.SharedBackgroundHip.step_prepared:
vmapstep_fwd_bwd_prepared -> [all_reduce: skipped at world size 1] -> vmapstep_adamw_apply) at the per-rank ray counts of
1 / 2 / 4 / 8 GPUs: device time per step (events) and host time per step (how long Python needs to ENQUEUE a step)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vmap_amd import fields, parallel, synth  # noqa: E402

dev = torch.device("cuda:0")
STEPS, S, H = 20, 14, 128
out = {"tool": "bg_chain_bench", "hidden": H, "samples": S, "steps_per_frame": STEPS, "rows": []}
for ranks in (1, 2, 4, 8):
    R = 1200 // ranks
    torch.manual_seed(7)
    fc = fields.OccupancyMap(hidden_size=H)
    fc.apply(fields.init_weights)
    pe = fields.UniDirsEmbed(max_deg=5, scale=5.0)
    b = synth.make_batch(1, R * STEPS, S, seed=77)
    loc = tuple(torch.from_numpy(b[k][0]).to(dev) for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask"))
    bg = parallel.SharedBackgroundHip(fc, pe, R, S, dev, max_steps=STEPS)
    for _ in range(3):
        bg.train_frame(*loc, n_steps=STEPS)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    frames = 10
    bg.prepare_frame(*loc, n_steps=STEPS)
    torch.cuda.synchronize()
    host = 0.0
    e0.record()
    for _ in range(frames):
        t0 = time.perf_counter()
        for i in range(STEPS):
            bg.step_prepared(i)
        host += time.perf_counter() - t0
    e1.record()
    torch.cuda.synchronize()
    out["rows"].append({"ranks": ranks, "rays_per_rank": R, "device_us_per_step": e0.elapsed_time(e1) / (frames * STEPS) * 1e3,
                        "host_enqueue_us_per_step": host / (frames * STEPS) * 1e6})
print(json.dumps(out))
