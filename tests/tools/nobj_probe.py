"""Measurement tool: the hidden-32 main kernel launched back to back for several object counts at the headline's rays / samples - does a
kernel whose XCDs hold 30 workgroups (20 objects: three objects on four of the eight XCDs, two on the others) run longer than one whose
XCDs all hold 20 (16 objects)?   python tests/tools/nobj_probe.py [n ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vmap_amd import step, synth  # noqa: E402

dev = torch.device("cuda:0")
R, S, H = 120, 10, 32
for rep in range(2):
    for n in [int(x) for x in sys.argv[1:]] or [8, 16, 20, 24]:
        fc, B, sc = synth.make_params(n, H, scale=2.0, seed=0)
        b = synth.make_batch(n, R, S, seed=1)
        tfc = [torch.from_numpy(a).to(dev) for a in fc]
        tB, tsc = torch.from_numpy(B).to(dev), torch.from_numpy(sc).to(dev)
        tb = [torch.from_numpy(b[k]).to(dev) for k in ("pcs", "z", "gt_depth", "gt_rgb", "sem", "depth_mask")]
        op = step.VmapStep(n, R, S, H, device=dev, max_steps=20)
        for _ in range(3):
            ms = op.profile_main_kernel(tfc, tB, tsc, *tb, reps=300)
        print(json.dumps({"n_obj": n, "plan": op.plan(), "kernel_us": ms * 1e3}), flush=True)
