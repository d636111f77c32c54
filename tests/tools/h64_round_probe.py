"""Measurement tool (round 4): what ONE 64-point round of the hidden-64 kernels costs, alone on a compute unit and next to a second
workgroup - the step of `n_obj` objects x 256 rays x 10 samples timed for several workgroups-per-object settings (256 workgroups = one
per compute unit, 512 = two) and kernels; per-round time = kernel time / rounds of the busiest workgroup."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vmap_amd import _lib, step, synth  # noqa: E402

dev = "cuda:0"
R, S, H = 256, 10, 64
out = []
CASES = [(4, 8, "auto", "bf16"), (4, 16, "auto", "bf16"), (8, 8, "auto", "bf16"), (16, 8, "auto", "bf16"), (4, 8, "ws1", "bf16"), (4, 8, "auto", "f32")] if len(sys.argv) > 1 and sys.argv[1] == "small" else None
for n_obj, nw, kern, wts in CASES or [(32, 8, "auto", "bf16"), (32, 15, "auto", "bf16"), (32, 16, "auto", "bf16"), (32, 4, "auto", "bf16"),
                             (32, 8, "ws1", "bf16"), (32, 8, "auto", "f32"), (32, 15, "auto", "f32"),
                             (256, 1, "auto", "bf16"), (256, 2, "auto", "bf16"), (256, 1, "ws1", "bf16")]:
    fc, B, sc = synth.make_params(n_obj, H, scale=2.0, seed=0)
    frame = synth.make_batch(n_obj, R * 20, S, seed=1)
    t = lambda a: torch.from_numpy(a).to(dev)
    tuning = {"workgroups_per_object": nw}
    if kern == "ws1":
        tuning["kernel"] = _lib.KERNEL_WS1
    op = step.VmapStep(n_obj, R, S, H, device=dev, max_steps=20, weights=wts, tuning=tuning)
    opt = step.FusedAdamWState(n_obj, H, dev)
    args = ([t(a) for a in fc], t(B), t(sc), *[t(frame[k]) for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask")])
    b = op.bind(*args, opt=opt)
    for _ in range(3):
        b.train_steps(20)
    torch.cuda.synchronize()
    pairs = [op.profile_train_steps(*args, opt=opt, n_steps=20) for _ in range(3)]
    k_ms = sum(p[0] for p in pairs) / len(pairs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.train_steps(20)
    e1.record()
    torch.cuda.synchronize()
    plan = op.plan()
    rounds_busiest = -(-plan["rounds_per_object"] // plan["workgroups_per_object"])
    rec = {"n_obj": n_obj, "workgroups_per_object": plan["workgroups_per_object"], "workgroups": n_obj * plan["workgroups_per_object"], "kernel": plan["kernel"],
           "weights": wts, "rounds_of_busiest_workgroup": rounds_busiest, "main_kernel_ms": k_ms, "us_per_round": k_ms * 1e3 / rounds_busiest,
           "step_ms": e0.elapsed_time(e1) / 100}
    print(json.dumps(rec), flush=True)
    out.append(rec)
    del op, opt, b, args
    torch.cuda.empty_cache()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "h64_round_probe.json"), "w"), indent=1)
