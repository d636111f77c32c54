"""Measurement tool: whole-step time of a hidden-64 / 128 shape with the library's choice of step_finalize_ws form and with the grouped form
forced (tuning.generic_finalize = 1), and for several workgroups_per_object:   python tests/tools/fin_form_probe.py [config] [weights]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vmap_amd import step, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "stress_rank8"
weights = sys.argv[2] if len(sys.argv) > 2 else "bf16"
cfg = synth.CONFIGS[name]
n, R, S, H = cfg["n_obj"], cfg["R"], cfg["S"], cfg["H"]
ipf = 20
dev = torch.device("cuda:0")
fc, B, sc = synth.make_params(n, H, scale=cfg["scale"], seed=0)
fr = synth.make_batch(n, R * ipf, S, seed=1)
t = lambda a: torch.from_numpy(a).to(dev)
args = ([t(a) for a in fc], t(B), t(sc), *[t(fr[k]) for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask")])
for rep in range(2):
    for tuning in (None, {"generic_finalize": 1}, {"workgroups_per_object": 8}, {"workgroups_per_object": 8, "generic_finalize": 1}, {"workgroups_per_object": 11}, {"workgroups_per_object": 22}):
        try:
            op = step.VmapStep(n, R, S, H, device=dev, max_steps=ipf, weights=weights, tuning=tuning)
        except Exception as e:
            print(json.dumps({"tuning": tuning, "error": str(e)[:200]})); continue
        opt = step.FusedAdamWState(n, H, dev)
        fn = lambda: op.train_steps(*args, opt=opt, n_steps=ipf)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(json.dumps({"config": name, "weights": weights, "tuning": tuning, "plan": op.plan(), "ms_per_step": e0.elapsed_time(e1) / 10 / ipf}), flush=True)
