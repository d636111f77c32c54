"""Measurement tool: per-step relative loss deviation of vmapstep_train_steps from the reference's own step loop (the whole-frame
fixtures of tests/golden), for the default kernels and the exact-fp32 A/B kernels - how fast a trajectory leaves the reference's."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cases  # noqa: E402
from vmap_amd import _lib, step  # noqa: E402

DEV = "cuda:0"
out = {}
for name in ("cfg2_frame20", "scannet50_frame", "h64_r256_frame", "bg128_frame", "scannet50_frame_bf16", "h64_r256_frame_bf16", "bg128_frame_bf16"):
    bf16 = name.endswith("_bf16")
    c = cases.build_frame_case(name[:-5] if bf16 else name)
    g = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
    for kern, tuning in (("default", None), ("exact_fp32", {"kernel": _lib.KERNEL_H32_F32 if c["H"] == 32 else _lib.KERNEL_GEN})):
        fc = [torch.from_numpy(a).to(DEV) for a in c["fc"]]
        B, sc = torch.from_numpy(c["B"]).to(DEV), torch.from_numpy(c["scale"]).to(DEV)
        fr = {k: torch.from_numpy(v).to(DEV) for k, v in c["frame"].items()}
        op = step.VmapStep(c["n"], c["R"], c["S"], c["H"], device=DEV, max_steps=c["n_steps"], tuning=tuning, weights="bf16" if bf16 else "f32")
        st = step.FusedAdamWState(c["n"], c["H"], DEV)
        res = op.train_steps(fc, B, sc, fr["pcs"], fr["z"], fr["gt_depth"], fr["gt_rgb"], fr["sem"], fr["depth_mask"], opt=st,
                             n_steps=c["n_steps"], ray_step=c["R"])
        torch.cuda.synchronize()
        losses = res.loss.cpu().numpy().astype(np.float64)
        out[f"{name}/{kern}"] = (np.abs(losses - g["losses"]) / np.abs(g["losses"])).tolist()
    out[f"{name}/reference_f64_vs_f32"] = (np.abs(g["f64_losses"] - g["losses"]) / np.abs(g["losses"])).tolist()
print(json.dumps(out))
