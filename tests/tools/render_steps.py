"""Measurement tool: N forward-only (render) calls of a BASELINE config (used under rocprofv3 --kernel-trace --stats)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vmap_amd import step, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "background"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
kernel = sys.argv[3] if len(sys.argv) > 3 else "auto"        # auto | s16 (hidden 32: forward prototype on 16-point tiles)
cfg = synth.CONFIGS[name]
n, R, S, H = cfg["n_obj"], cfg["R"], cfg["S"], cfg["H"]
fc, B, sc = synth.make_params(n, H, scale=cfg["scale"], seed=0)
b = synth.make_batch(n, R, S, seed=1)
dev = "cuda:0"
tfc = [torch.from_numpy(a).to(dev) for a in fc]
tB, tsc = torch.from_numpy(B).to(dev), torch.from_numpy(sc).to(dev)
tb = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
from vmap_amd import _lib  # noqa: E402
op = step.VmapStep(n, R, S, H, device=dev, tuning={"kernel": _lib.KERNEL_S16_FWD} if kernel == "s16" else None)
ref = step.VmapStep(n, R, S, H, device=dev)
ra = op.render(tfc, tB, tsc, tb["pcs"], tb["z"], tb["gt_depth"], tb["gt_rgb"], tb["sem"], tb["depth_mask"])
rb = ref.render(tfc, tB, tsc, tb["pcs"], tb["z"], tb["gt_depth"], tb["gt_rgb"], tb["sem"], tb["depth_mask"])
torch.cuda.synchronize()
print("max |depth diff| vs the default kernel", float((ra.render_depth - rb.render_depth).abs().max()), "loss", float(ra.loss[0]), float(rb.loss[0]))
for _ in range(reps):
    op.render(tfc, tB, tsc, tb["pcs"], tb["z"], tb["gt_depth"], tb["gt_rgb"], tb["sem"], tb["depth_mask"])
torch.cuda.synchronize()
print("done", reps)
