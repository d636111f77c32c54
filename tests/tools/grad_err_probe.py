"""Measurement tool: per-tensor gradient error of the hidden-32 kernels against a reference fixture (tests/golden/<case>.npz):
    python tests/tools/grad_err_probe.py [case ...]        -> one JSON line per (case, kernel)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cases  # noqa: E402
from conftest import GRAD_KEYS, load_golden, relerr  # noqa: E402
from vmap_amd import _lib, step  # noqa: E402

dev = "cuda:0"
for name in (sys.argv[1:] or ["cfg2", "tiny", "ragged"]):
    c = cases.build_case(name)
    g = load_golden(name)
    fc = [torch.from_numpy(a).to(dev) for a in c["fc"]]
    B, sc = torch.from_numpy(c["B"]).to(dev), torch.from_numpy(c["scale"]).to(dev)
    b = {k: torch.from_numpy(v).to(dev) for k, v in c["batch"].items()}
    for label, tuning in (("default", None), ("bwd6", {"kernel": _lib.KERNEL_S32_BWD6}), ("exact_fp32", {"kernel": _lib.KERNEL_H32_F32})):
        op = step.VmapStep(c["n"], c["R"], c["S"], c["H"], device=dev, tuning=tuning)
        gfc = [torch.zeros_like(t) for t in fc]
        gB = torch.zeros_like(B)
        res = op.fwd_bwd(fc, B, sc, b["pcs"], b["z"], b["gt_depth"], b["gt_rgb"], b["sem"], b["depth_mask"], grads_fc=gfc, grad_B=gB, render=True)
        torch.cuda.synchronize()
        out = {f"g_fc{t}": gfc[t].cpu().numpy() for t in range(14)}
        out["g_B"] = gB.cpu().numpy()
        errs = {k: float("%.3g" % relerr(out[k], g[k])) for k in GRAD_KEYS}
        rend = {k: float("%.3g" % relerr(getattr(res, a).cpu().numpy(), g[k])) for k, a in (("render_depth", "render_depth"), ("render_color", "render_color"), ("opacity", "opacity"))}
        print(json.dumps({"case": name, "kernel": label, "loss_rel_err": float("%.3g" % (abs(float(res.loss[0]) - float(g["loss"])) / abs(float(g["loss"])))),
                          "renders": rend, "grads": errs, "worst": max(errs.values())}))
